#!/bin/bash
# round 6: the determinism soak on the round's FINAL library: 4000 renders and 1500 step replays per fp32 arithmetic, one process
cd "$GRAFT_REPO_ROOT" || exit 1
out=gpurun_out/${1:-r06_soak_head}
mkdir -p $out
export TMPDIR=/tmp
git rev-parse HEAD > $out/head.txt 2>/dev/null
timeout 1500 python tools/determinism_soak.py single fp32x6 4000 1500 > $out/single_fp32x6.log 2>&1
echo "single fp32x6 rc=$?"; grep -E "RESULT|iter" $out/single_fp32x6.log | tail -6 | cut -c1-500
timeout 1900 python tools/determinism_soak.py single fp32 4000 1500 > $out/single_fp32.log 2>&1
echo "single fp32 rc=$?"; grep -E "RESULT|iter" $out/single_fp32.log | tail -6 | cut -c1-500
timeout 300 python tools/determinism_soak.py single bf16 1000 1000 > $out/single_bf16.log 2>&1
echo "single bf16 rc=$?"; grep -E "RESULT|iter" $out/single_bf16.log | tail -4 | cut -c1-500
