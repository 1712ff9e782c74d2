#!/bin/bash
# kernel table of the configs[3] per-GPU shape (1024 + 256 rays): where the fixed costs sit
cd "$GRAFT_REPO_ROOT" || exit 1
out=gpurun_out/${1:-rays1024}
mkdir -p $out
export TMPDIR=/tmp
( cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats -d "$GRAFT_REPO_ROOT/$out/prof" -o p -- python "$GRAFT_REPO_ROOT/bench.py" --rays 1024 --inst-rays ${INST:-1024} --no-cpu-baseline --no-extras --steps 20 --warmup 3 > "$GRAFT_REPO_ROOT/$out/bench.json" 2> "$GRAFT_REPO_ROOT/$out/prof.log" )
db=$(find $out/prof -name "*.db" | head -1)
python tools/rocprof_summary.py "$db" $out/kernel_stats.txt
rm -rf $out/prof
head -45 $out/kernel_stats.txt | cut -c1-140
python -c "
import json; d=json.loads(open('$out/bench.json').read().strip().splitlines()[-1]); print('ms_per_step', d['ms_per_step'])"
