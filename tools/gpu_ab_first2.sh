#!/bin/bash
# A/B of the fused first-two-layers backward inside the training step, same box: off / on with the activation kept / on without it
cd "$GRAFT_REPO_ROOT" || exit 1
out=gpurun_out/${1:-ab_first2}
mkdir -p $out
export TMPDIR=/tmp
run() { # name, env...
  name=$1; shift
  for rep in 1 2; do
    env "$@" timeout 600 python bench.py --no-cpu-baseline --no-extras --steps 40 --warmup 5 2> $out/$name.$rep.err | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$name', $rep, 'ms_per_step %.4f' % d['ms_per_step'], 'frac %.4f' % d['roofline']['frac'])" | tee -a $out/summary.txt
  done
}
run off CLIFT_FUSE_FIRST2_BWD=0
run on_keep CLIFT_DROP_FIRST_ACT=0
run on_drop CLIFT_DROP_FIRST_ACT=1
run off CLIFT_FUSE_FIRST2_BWD=0
run on_drop CLIFT_DROP_FIRST_ACT=1
