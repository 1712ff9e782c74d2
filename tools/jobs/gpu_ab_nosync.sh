#!/bin/bash
# default (one host read-back of the active-sample count per chunk) against --nosync, at the bench shape and at the configs[3] per-GPU shape
cd "$GRAFT_REPO_ROOT" || exit 1
out=gpurun_out/${1:-ab_nosync}
mkdir -p $out
export TMPDIR=/tmp
run() { name=$1; shift
  for rep in 1 2; do
    timeout 600 python bench.py --no-cpu-baseline --no-extras --steps 40 --warmup 8 "$@" 2> $out/$name.$rep.err | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$name', $rep, 'ms_per_step %.4f' % d['ms_per_step'])" | tee -a $out/summary.txt
  done
}
run default4096
run nosync4096 --nosync
run default1024 --rays 1024 --inst-rays 256
run nosync1024 --rays 1024 --inst-rays 256 --nosync
