#!/bin/bash
# round-2 GPU job A: new parity tests, counter passes over the dominant kernel (torch-free harness), host-boundness probe
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out/r2a
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_gpu_round2.py -x -q -m gpu > gpurun_out/r2a/pytest_round2.log 2>&1
echo "pytest round2 rc=$?" | tee -a gpurun_out/r2a/summary.txt
for mode in fwd dgrad wgrad; do
  ./tools/pmc_harness.bin $mode 249000 5 >> gpurun_out/r2a/summary.txt 2>&1
done
for set in "FETCH_SIZE" "WRITE_SIZE" "SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_INST_LDS SQ_INST_CYCLES_VMEM" "SQ_INSTS_VALU_MFMA_MOPS_F32 SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_VALU SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_INSTS_VMEM SQ_ACTIVE_INST_MISC"; do
  tag=$(echo $set | cut -d' ' -f1)
  for mode in fwd dgrad wgrad; do
    ( cd /tmp && timeout 150 rocprofv3 --pmc $set --kernel-trace --output-format csv -d "$GRAFT_REPO_ROOT/gpurun_out/r2a/pmc_${tag}_${mode}" -o p -- "$GRAFT_REPO_ROOT/tools/pmc_harness.bin" $mode 249000 3 > "$GRAFT_REPO_ROOT/gpurun_out/r2a/pmc_${tag}_${mode}.log" 2>&1 )
    echo "pmc $tag $mode rc=$?" >> gpurun_out/r2a/summary.txt
  done
done
python tools/pmc_parse.py gpurun_out/r2a/pmc_* > gpurun_out/r2a/pmc_table.txt 2>&1
for r in 256 1024 4096; do
  timeout 300 python bench.py --rays $r --inst-rays $((r/4)) --no-cpu-baseline --steps 20 --warmup 5 > gpurun_out/r2a/bench_rays$r.json 2> gpurun_out/r2a/bench_rays$r.err
  echo "bench rays=$r rc=$?" >> gpurun_out/r2a/summary.txt
done
cat gpurun_out/r2a/summary.txt
tail -5 gpurun_out/r2a/pytest_round2.log
