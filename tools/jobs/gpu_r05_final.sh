#!/bin/bash
# round 5, the round-end record: whole GPU suite + smoke + the default bench line (with the CPU baseline and the extras), then the kernel tables of the final library
cd "$GRAFT_REPO_ROOT" || exit 1
out=gpurun_out/${1:-r05_final}
mkdir -p $out
export TMPDIR=/tmp
bash tools/jobs/gpu_full_tests.sh $(basename $out)
prof() {  # name, command...
  name=$1; shift
  ( cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats -d "$GRAFT_REPO_ROOT/$out/prof_$name" -o p -- "$@" > "$GRAFT_REPO_ROOT/$out/$name.out" 2> "$GRAFT_REPO_ROOT/$out/$name.log" )
  db=$(find $out/prof_$name -name "*.db" | head -1)
  python tools/rocprof_summary.py "$db" $out/kernel_stats_$name.txt
  rm -rf $out/prof_$name
}
prof fp32x6 python "$GRAFT_REPO_ROOT/bench.py" --no-cpu-baseline --no-extras --steps 20 --warmup 3
prof fp32 python "$GRAFT_REPO_ROOT/bench.py" --dtype fp32 --no-cpu-baseline --no-extras --steps 20 --warmup 3
prof bf16 python "$GRAFT_REPO_ROOT/bench.py" --dtype bf16 --no-cpu-baseline --no-extras --steps 20 --warmup 3
prof inference python "$GRAFT_REPO_ROOT/tools/inference_probe.py" fp32x6 32768
for n in fp32x6 fp32 bf16; do python -c "
import json; d=json.loads(open('$out/$n.out').read().strip().splitlines()[-1]); print('$n ms_per_step', d['ms_per_step'], 'median', d.get('step_ms_median'), 'frac', d['roofline']['frac'])"; done
tail -1 $out/inference.out
