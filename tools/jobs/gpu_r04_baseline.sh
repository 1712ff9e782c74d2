#!/bin/bash
# round-4 opening job: (1) exact-vs-exact two-process render stress, (2) the GPU suite with fp32x6 forced
cd "$GRAFT_REPO_ROOT" || exit 1
out=gpurun_out/r04_baseline
mkdir -p $out
export TMPDIR=/tmp
N=${1:-2000}
( timeout 1500 python tools/shared_gpu_render_stress.py A fp32 $N > $out/stress_A.log 2>&1 & 
  timeout 1500 python tools/shared_gpu_render_stress.py B fp32 $N > $out/stress_B.log 2>&1 &
  wait )
tail -3 $out/stress_A.log $out/stress_B.log
CLIFT_MLP_DTYPE=fp32x6 timeout 1500 python -m pytest tests -q -m gpu --timeout 900 > $out/pytest_forced_x6.log 2>&1
echo "forced x6 rc=$?"
tail -15 $out/pytest_forced_x6.log
