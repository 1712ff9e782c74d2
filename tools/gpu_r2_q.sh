#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
out=gpurun_out/r2q
mkdir -p $out
export TMPDIR=/tmp
run() {  # tag, extra env, bench args
  tag=$1; shift
  ( cd /tmp && timeout 600 env $ENVX rocprofv3 --kernel-trace --stats -d "$GRAFT_REPO_ROOT/$out/prof_$tag" -o p -- python "$GRAFT_REPO_ROOT/bench.py" --no-cpu-baseline --no-extras --steps 10 --warmup 3 "$@" > "$GRAFT_REPO_ROOT/$out/bench_$tag.json" 2> "$GRAFT_REPO_ROOT/$out/prof_$tag.log" )
  db=$(find $out/prof_$tag -name "*.db" | head -1)
  python tools/rocprof_summary.py "$db" $out/ks_$tag.txt
  rm -rf $out/prof_$tag
  echo "== $tag" >> $out/summary.txt
  grep -E "k_app_gather_bwd|k_density_bwd|k_app_gather_fwd|k_density_fwd" $out/ks_$tag.txt | cut -c1-110 >> $out/summary.txt
  python -c "
import json; d=json.loads(open('$out/bench_$tag.json').read().strip().splitlines()[-1]); print('   ms/step', round(d['ms_per_step'],3), 'S', d['samples_per_ray'], 'f_active', round(d['f_active'],4), 'M', round(d['f_active']*4096*d['samples_per_ray']))" >> $out/summary.txt
}
ENVX="X=1" run g128
ENVX="X=1" run g64 --grid 64
ENVX="X=1" run g192 --grid 192
ENVX="CLIFT_APP_SPLIT=1" run g128split
ENVX="CLIFT_XCD_PRIVATE=0" run g128noxcd
cat $out/summary.txt
