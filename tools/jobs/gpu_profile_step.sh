#!/bin/bash
# kernel-trace profile of the default bench command: summary -> gpurun_out/<tag>/kernel_stats.txt (copy into profiles/ to keep it)
cd "$GRAFT_REPO_ROOT" || exit 1
tag=${1:-prof}
out=gpurun_out/$tag
mkdir -p $out
export TMPDIR=/tmp
( cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats -d "$GRAFT_REPO_ROOT/$out/prof" -o p -- python "$GRAFT_REPO_ROOT/bench.py" --no-cpu-baseline --no-extras --steps 20 --warmup 3 > "$GRAFT_REPO_ROOT/$out/bench_under_rocprof.json" 2> "$GRAFT_REPO_ROOT/$out/prof.log" )
echo "prof rc=$?" > $out/summary.txt
db=$(find $out/prof -name "*.db" | head -1)
python tools/rocprof_summary.py "$db" $out/kernel_stats.txt >> $out/summary.txt 2>&1
rm -rf $out/prof
cat $out/summary.txt; head -60 $out/kernel_stats.txt
