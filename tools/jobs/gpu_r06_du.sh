#!/bin/bash
# round 6: k_density_bwd_u variants (tools/_scratch/abl/libclift_du*.so, DU_ABL in csrc/march.hip) timed inside the bench step under rocprofv3
cd "$GRAFT_REPO_ROOT" || exit 1
out=gpurun_out/r06_du; mkdir -p $out; export TMPDIR=/tmp
for v in base 1 2 4; do
  lib=""; [ $v != base ] && lib="$GRAFT_REPO_ROOT/tools/_scratch/abl/libclift_du$v.so"
  ( cd /tmp && CLIFT_LIB_PATH=$lib timeout 300 rocprofv3 --kernel-trace --stats -d "$GRAFT_REPO_ROOT/$out/p_$v" -o p -- python "$GRAFT_REPO_ROOT/bench.py" --no-cpu-baseline --no-extras --steps 10 --warmup 3 > /dev/null 2>&1 )
  python tools/rocprof_summary.py "$(find $out/p_$v -name '*.db' | head -1)" $out/ks_$v.txt > /dev/null 2>&1
  echo "DU_ABL=$v: $(grep -E 'k_density_bwd_u' $out/ks_$v.txt | head -1 | cut -c1-60)  |  $(grep -E 'k_app_gather_bwd_u' $out/ks_$v.txt | head -1 | cut -c1-50)"
  rm -rf $out/p_$v
done
