#!/bin/bash
# round-2 GPU job B: whole GPU suite + counter passes (HBM bytes) over the bench command itself
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out/r2b
export TMPDIR=/tmp
timeout 1500 python -m pytest tests -x -q -m gpu > gpurun_out/r2b/pytest_gpu.log 2>&1
echo "pytest gpu rc=$?" | tee -a gpurun_out/r2b/summary.txt
tail -3 gpurun_out/r2b/pytest_gpu.log >> gpurun_out/r2b/summary.txt
for set in FETCH_SIZE WRITE_SIZE; do
  ( cd /tmp && timeout 400 rocprofv3 --pmc $set --kernel-trace --output-format csv -d "$GRAFT_REPO_ROOT/gpurun_out/r2b/pmc_bench_${set}" -o p -- python "$GRAFT_REPO_ROOT/bench.py" --steps 3 --warmup 1 --no-cpu-baseline > "$GRAFT_REPO_ROOT/gpurun_out/r2b/pmc_bench_${set}.log" 2>&1 )
  echo "pmc bench $set rc=$?" >> gpurun_out/r2b/summary.txt
done
python tools/pmc_parse.py gpurun_out/r2b/pmc_bench_* > gpurun_out/r2b/pmc_bench_table.txt 2>&1
rm -f gpurun_out/r2b/pmc_bench_*/p_kernel_trace.csv
cat gpurun_out/r2b/summary.txt
