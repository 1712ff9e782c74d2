#!/bin/bash
# round 5: the determinism soak again, on the library with the fused output layers' consumers pinned behind their waits
cd "$GRAFT_REPO_ROOT" || exit 1
out=gpurun_out/r05_soak2
mkdir -p $out
export TMPDIR=/tmp
timeout 2000 python tools/determinism_soak.py single fp32 ${1:-8000} 2000 > $out/single_fp32.log 2>&1
echo "single fp32 rc=$?"; grep -E "RESULT|iter" $out/single_fp32.log | tail -8 | cut -c1-400
timeout 900 python tools/determinism_soak.py single fp32x6 ${2:-2000} 2000 > $out/single_fp32x6.log 2>&1
echo "single fp32x6 rc=$?"; grep -E "RESULT|iter" $out/single_fp32x6.log | tail -8 | cut -c1-400
timeout 600 python tools/determinism_soak.py single bf16 500 2000 > $out/single_bf16.log 2>&1
echo "single bf16 rc=$?"; grep -E "RESULT|iter" $out/single_bf16.log | tail -8 | cut -c1-400
R2=${3:-2000}
timeout 1500 python tools/determinism_soak.py A fp32 $R2 0 > $out/two_A_fp32.log 2>&1 &
pa=$!
timeout 1500 python tools/determinism_soak.py B fp32 $R2 0 > $out/two_B_fp32.log 2>&1 &
pb=$!
wait $pa; wait $pb
echo "two-process fp32:"; grep -E "RESULT|iter" $out/two_A_fp32.log $out/two_B_fp32.log | tail -10 | cut -c1-400
