#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
out=gpurun_out/r2p
mkdir -p $out
export TMPDIR=/tmp
( cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats -d "$GRAFT_REPO_ROOT/$out/prof" -o p -- python "$GRAFT_REPO_ROOT/bench.py" --rays 1024 --inst-rays 1024 --classes 2 --no-cpu-baseline --no-extras --steps 20 --warmup 3 > "$GRAFT_REPO_ROOT/$out/bench1024.json" 2> "$GRAFT_REPO_ROOT/$out/prof.log" )
db=$(find $out/prof -name "*.db" | head -1)
python tools/rocprof_summary.py "$db" $out/kernel_stats_1024.txt
rm -rf $out/prof
head -45 $out/kernel_stats_1024.txt | cut -c1-150; tail -2 $out/kernel_stats_1024.txt
python -c "
import json; d=json.loads(open('$out/bench1024.json').read().strip().splitlines()[-1]); print('ms/step under rocprof', d['ms_per_step'])"
