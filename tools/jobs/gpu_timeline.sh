#!/bin/bash
# timeline of one bench step (kernel order, durations, gaps): usage gpu_timeline.sh <tag> [bench.py arguments]
cd "$GRAFT_REPO_ROOT" || exit 1
tag=${1:-timeline}; shift
out=gpurun_out/$tag; mkdir -p $out
export TMPDIR=/tmp
( cd /tmp && timeout 600 rocprofv3 --kernel-trace --output-format csv -d "$GRAFT_REPO_ROOT/$out/trace" -o p -- python "$GRAFT_REPO_ROOT/bench.py" --no-cpu-baseline --no-extras --steps 8 --warmup 3 "$@" > "$GRAFT_REPO_ROOT/$out/line.json" 2> "$GRAFT_REPO_ROOT/$out/trace.log" )
python tools/rocprof_timeline.py $out/trace $out/timeline.txt ${STEP:-6}
rm -rf $out/trace
tail -1 $out/timeline.txt
