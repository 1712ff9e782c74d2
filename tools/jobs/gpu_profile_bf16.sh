#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
out=gpurun_out/${1:-bf16prof}
mkdir -p $out
export TMPDIR=/tmp
( cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats -d "$GRAFT_REPO_ROOT/$out/prof" -o p -- python "$GRAFT_REPO_ROOT/bench.py" --dtype bf16 --no-cpu-baseline --no-extras --steps 20 --warmup 3 > "$GRAFT_REPO_ROOT/$out/bench_bf16.json" 2> "$GRAFT_REPO_ROOT/$out/prof.log" )
db=$(find $out/prof -name "*.db" | head -1)
python tools/rocprof_summary.py "$db" $out/kernel_stats_bf16.txt
rm -rf $out/prof
( cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats -d "$GRAFT_REPO_ROOT/$out/prof2" -o p -- python "$GRAFT_REPO_ROOT/tools/inference_probe.py" > "$GRAFT_REPO_ROOT/$out/inference.txt" 2> "$GRAFT_REPO_ROOT/$out/prof2.log" )
db=$(find $out/prof2 -name "*.db" | head -1)
python tools/rocprof_summary.py "$db" $out/kernel_stats_inference.txt
rm -rf $out/prof2
head -30 $out/kernel_stats_bf16.txt; cat $out/inference.txt | tail -3; head -24 $out/kernel_stats_inference.txt
