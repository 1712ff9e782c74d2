#!/bin/bash
# round 5, job D: the drain-barrier fix under the micro-soak, the tests that failed in job C, the fixed cost of the persistent split launches and what it is made of
cd "$GRAFT_REPO_ROOT" || exit 1
out=gpurun_out/${1:-r05_d}
mkdir -p $out
export TMPDIR=/tmp
timeout 600 python -m pytest tests/test_gpu_round5.py tests/test_gpu_round5b.py::test_fp32x6_appearance_chain_against_the_exact_chain -q -m gpu --timeout 600 > $out/pytest.log 2>&1
echo "pytest rc=$?"; tail -4 $out/pytest.log | cut -c1-300
timeout 400 python tools/last2_soak.py 150 exact_infer,exact_keep > $out/last2_soak_after_fix.log 2>&1
echo "last2 soak rc=$?"; grep -E "RESULT|launch" $out/last2_soak_after_fix.log | cut -c1-400
VARIANTS="1024 512 576 704 768 960 64 256" bash tools/jobs/x6_ablation.sh fixed > $out/x6_fixed.txt 2>&1
cat $out/x6_fixed.txt | cut -c1-220
