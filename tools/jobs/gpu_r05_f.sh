#!/bin/bash
# round 5, job F: row blocks against the memory-side cache (frame render), the n6 fragment-prefetch variant
cd "$GRAFT_REPO_ROOT" || exit 1
out=gpurun_out/${1:-r05_f}
mkdir -p $out
export TMPDIR=/tmp
timeout 600 python tools/mall_block_probe.py 4194304 > $out/mall_block_probe.txt 2>&1
cat $out/mall_block_probe.txt | cut -c1-200
for v in 0 2048 0 2048; do timeout 200 python tools/x6_fixed_probe.py $v 2>/dev/null | grep -E "X6_ABL|n6" ; done > $out/n6_prefetch_variant.txt
cat $out/n6_prefetch_variant.txt | cut -c1-200
