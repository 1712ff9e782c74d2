#!/bin/bash
# round 6: one library variant against the build, inside the bench step under rocprofv3.  usage: gpu_r06_variant.sh <variant.so> <kernel regex>
cd "$GRAFT_REPO_ROOT" || exit 1
out=gpurun_out/r06_variant; mkdir -p $out; export TMPDIR=/tmp
for v in base variant base variant; do
  lib=""; [ $v = variant ] && lib="$GRAFT_REPO_ROOT/$1"
  ( cd /tmp && CLIFT_LIB_PATH=$lib timeout 300 rocprofv3 --kernel-trace --stats -d "$GRAFT_REPO_ROOT/$out/p_$v" -o p -- python "$GRAFT_REPO_ROOT/bench.py" --no-cpu-baseline --no-extras --steps 10 --warmup 3 > /dev/null 2>&1 )
  python tools/rocprof_summary.py "$(find $out/p_$v -name '*.db' | head -1)" $out/ks_$v.txt > /dev/null 2>&1
  echo "$v: $(grep -E "$2" $out/ks_$v.txt | awk '{printf "%s ", $3}') | ref $(grep -E 'k_layer_x6<false, true, 0, false, false>' $out/ks_$v.txt | awk '{print $3}')"
  rm -rf $out/p_$v
done
