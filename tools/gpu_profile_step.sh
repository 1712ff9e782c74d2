#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out/r2h
export TMPDIR=/tmp
timeout 600 python bench.py --no-cpu-baseline --no-extras > gpurun_out/r2h/bench.json 2> gpurun_out/r2h/bench.err
echo "bench rc=$?" >> gpurun_out/r2h/summary.txt
( cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats -d "$GRAFT_REPO_ROOT/gpurun_out/r2h/prof" -o p -- python "$GRAFT_REPO_ROOT/bench.py" --no-cpu-baseline --no-extras --steps 20 --warmup 3 > "$GRAFT_REPO_ROOT/gpurun_out/r2h/bench_prof.json" 2> "$GRAFT_REPO_ROOT/gpurun_out/r2h/prof.log" )
echo "prof rc=$?" >> gpurun_out/r2h/summary.txt
db=$(find gpurun_out/r2h/prof -name "*.db" | head -1)
python tools/rocprof_summary.py "$db" gpurun_out/r2h/kernel_stats.txt >> gpurun_out/r2h/summary.txt 2>&1
rm -rf gpurun_out/r2h/prof
timeout 900 python -m pytest tests/test_gpu_round2.py tests/test_gpu_parity.py -q -m gpu --timeout 400 -x > gpurun_out/r2h/pytest.log 2>&1
echo "pytest rc=$?" >> gpurun_out/r2h/summary.txt
tail -4 gpurun_out/r2h/pytest.log >> gpurun_out/r2h/summary.txt
cat gpurun_out/r2h/summary.txt; head -50 gpurun_out/r2h/kernel_stats.txt
