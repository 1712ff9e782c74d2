#!/bin/bash
# round 6: whole GPU suite + smoke + default bench line, then kernel tables (fp32x6 / bf16 / inference) of the library as it stands
cd "$GRAFT_REPO_ROOT" || exit 1
out=gpurun_out/${1:-r06_base}
mkdir -p $out
export TMPDIR=/tmp
[ "$2" = "notests" ] || bash tools/jobs/gpu_full_tests.sh $(basename $out)
prof() {  # name, command...
  name=$1; shift
  ( cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats -d "$GRAFT_REPO_ROOT/$out/prof_$name" -o p -- "$@" > "$GRAFT_REPO_ROOT/$out/$name.out" 2> "$GRAFT_REPO_ROOT/$out/$name.log" )
  db=$(find $out/prof_$name -name "*.db" | head -1)
  python tools/rocprof_summary.py "$db" $out/kernel_stats_$name.txt
  rm -rf $out/prof_$name
}
prof fp32x6 python "$GRAFT_REPO_ROOT/bench.py" --no-cpu-baseline --no-extras --steps 20 --warmup 3
prof bf16 python "$GRAFT_REPO_ROOT/bench.py" --dtype bf16 --no-cpu-baseline --no-extras --steps 20 --warmup 3
prof inference python "$GRAFT_REPO_ROOT/tools/inference_probe.py" fp32x6 32768
for n in fp32x6 bf16; do python -c "
import json; d=json.loads(open('$out/$n.out').read().strip().splitlines()[-1]); print('$n ms_per_step', d['ms_per_step'], 'median', d.get('step_ms_median'), 'frac', d['roofline']['frac'])"; done
tail -1 $out/inference.out
