#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
out=gpurun_out/r2m
mkdir -p $out
export TMPDIR=/tmp
python tools/host_overhead_probe.py 64 2>&1 | grep "ms per step" >> $out/summary.txt
timeout 300 python bench.py --rays 1024 --inst-rays 1024 --classes 2 --no-cpu-baseline --no-extras --steps 20 --warmup 5 > $out/bench_rays1024.json 2> $out/bench_rays1024.err
timeout 600 python bench.py --no-cpu-baseline --no-extras > $out/bench.json 2> $out/bench.err
for mode in fwd gen outv dgrad wgrad; do ./tools/pmc_harness.bin $mode 249000 5 >> $out/summary.txt 2>&1; done
for set in "FETCH_SIZE" "WRITE_SIZE" "SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_INST_LDS SQ_INST_CYCLES_VMEM"; do
  tag=$(echo $set | cut -d' ' -f1)
  for mode in gen outv wgrad; do
    ( cd /tmp && timeout 150 rocprofv3 --pmc $set --kernel-trace --output-format csv -d "$GRAFT_REPO_ROOT/$out/pmc_${tag}_${mode}" -o p -- "$GRAFT_REPO_ROOT/tools/pmc_harness.bin" $mode 249000 3 > "$GRAFT_REPO_ROOT/$out/pmc_${tag}_${mode}.log" 2>&1 )
  done
done
python tools/pmc_parse.py $out/pmc_* 2>/dev/null | grep -v rocclr > $out/pmc_table.txt
rm -rf $out/pmc_*/ 
timeout 600 python -m pytest tests/test_gpu_trainer_modes.py -q -m gpu --timeout 300 > $out/pytest.log 2>&1; tail -2 $out/pytest.log >> $out/summary.txt
cat $out/summary.txt; cat $out/pmc_table.txt
