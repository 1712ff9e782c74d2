#!/bin/bash
# A/B of the XCD-private gradient shards inside the training step, same box, at the bench shape and at the configs[3] per-GPU shape
cd "$GRAFT_REPO_ROOT" || exit 1
out=gpurun_out/${1:-ab_shards}
mkdir -p $out
export TMPDIR=/tmp
run() { name=$1; shift
  for rep in 1 2; do
    env "$@" timeout 600 python bench.py --no-cpu-baseline --no-extras --steps 40 --warmup 8 $ARGS 2> $out/$name.$rep.err | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$name', $rep, 'ms_per_step %.4f' % d['ms_per_step'])" | tee -a $out/summary.txt
  done
}
ARGS=""
run off4096 CLIFT_GRAD_SHARDS=0
run on4096 CLIFT_GRAD_SHARDS=1
run off4096 CLIFT_GRAD_SHARDS=0
run on4096 CLIFT_GRAD_SHARDS=1
ARGS="--rays 1024 --inst-rays 256"
run off1024 CLIFT_GRAD_SHARDS=0
run on1024 CLIFT_GRAD_SHARDS=1
