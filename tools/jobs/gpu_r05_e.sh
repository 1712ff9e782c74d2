#!/bin/bash
# round 5, job E: the synchronous fused output layers of the exact-fp32 kernels: the tests of the exact mode, the micro-soak, the render soak
cd "$GRAFT_REPO_ROOT" || exit 1
out=gpurun_out/${1:-r05_e}
mkdir -p $out
export TMPDIR=/tmp
CLIFT_MLP_DTYPE=fp32 timeout 1500 python -m pytest tests/test_gpu_parity.py tests/test_gpu_round2.py tests/test_gpu_round4.py tests/test_gpu_round4b.py tests/test_gpu_round5.py -q -m gpu --timeout 900 -x > $out/pytest_exact.log 2>&1
echo "pytest (exact fp32 forced) rc=$?"; tail -5 $out/pytest_exact.log | cut -c1-300
timeout 600 python -m pytest tests/test_gpu_round5.py tests/test_gpu_round4b.py tests/test_gpu_round3.py -q -m gpu --timeout 600 -x > $out/pytest_default.log 2>&1
echo "pytest (default) rc=$?"; tail -3 $out/pytest_default.log | cut -c1-300
timeout 500 python tools/last2_soak.py 200 exact_infer > $out/last2_soak_sync.log 2>&1
timeout 300 python tools/last2_soak.py 100 exact_keep >> $out/last2_soak_sync.log 2>&1
echo "last2 soak"; grep -E "RESULT|launch" $out/last2_soak_sync.log | cut -c1-400
timeout 1200 python tools/determinism_soak.py single fp32 ${2:-3000} 300 > $out/soak_fp32_sync.log 2>&1
echo "soak fp32 rc=$?"; grep -E "RESULT|iter" $out/soak_fp32_sync.log | grep -v "step iter" | tail -6 | cut -c1-500
