#!/bin/bash
# round 5: single-process determinism soak + the two-process experiment with the instrumented compare
cd "$GRAFT_REPO_ROOT" || exit 1
out=gpurun_out/r05_soak
mkdir -p $out
export TMPDIR=/tmp
R=${1:-8000}; S=${2:-2000}; R2=${3:-2000}
timeout 600 python tools/determinism_soak.py sanity fp32x6 5 5 > $out/sanity.log 2>&1 || { echo "sanity failed"; tail -30 $out/sanity.log; exit 1; }
tail -3 $out/sanity.log
timeout 2400 python tools/determinism_soak.py single fp32 $R $S > $out/single_fp32.log 2>&1
echo "single fp32 rc=$?"; grep -E "RESULT|iter" $out/single_fp32.log | tail -20
timeout 2400 python tools/determinism_soak.py single fp32x6 $R $S > $out/single_fp32x6.log 2>&1
echo "single fp32x6 rc=$?"; grep -E "RESULT|iter" $out/single_fp32x6.log | tail -20
timeout 900 python tools/determinism_soak.py single bf16 2000 $S > $out/single_bf16.log 2>&1
echo "single bf16 rc=$?"; grep -E "RESULT|iter" $out/single_bf16.log | tail -20
if [ "$R2" -gt 0 ]; then
  timeout 1800 python tools/determinism_soak.py A fp32 $R2 0 > $out/two_A_fp32.log 2>&1 &
  pa=$!
  timeout 1800 python tools/determinism_soak.py B fp32 $R2 0 > $out/two_B_fp32.log 2>&1 &
  pb=$!
  wait $pa; wait $pb
  echo "two-process fp32:"; grep -E "RESULT|iter" $out/two_A_fp32.log $out/two_B_fp32.log | tail -30
fi
