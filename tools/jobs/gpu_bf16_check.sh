#!/bin/bash
# bf16 mode: named tests + the bf16 step time / kernel table (configs[2])
cd "$GRAFT_REPO_ROOT" || exit 1
out=gpurun_out/${1:-bf16}; mkdir -p $out
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_gpu_round4b.py tests/test_gpu_round3.py -q -x --timeout 600 -k "bf16 or out_layer or output_layer" 2>&1 | tail -5 | cut -c1-200
for i in 1 2; do python bench.py --dtype bf16 --no-cpu-baseline --no-extras 2>/dev/null | python -c "import json,sys;d=json.loads(sys.stdin.read());print('bf16 step', d['ms_per_step'], d['step_ms_median'])"; done
( cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats -d "$GRAFT_REPO_ROOT/$out/prof" -o p -- python "$GRAFT_REPO_ROOT/bench.py" --dtype bf16 --no-cpu-baseline --no-extras > /dev/null 2> "$GRAFT_REPO_ROOT/$out/prof.log" )
db=$(find $out/prof -name "*.db" | head -1)
python tools/rocprof_summary.py "$db" $out/kernel_stats_bf16.txt
rm -rf $out/prof
head -32 $out/kernel_stats_bf16.txt | cut -c1-150; tail -1 $out/kernel_stats_bf16.txt
