#!/bin/bash
# round 5, job C: the fp32x6 appearance kernels (layer_n6.hip): their tests, the tests of everything they sit under, a same-box A/B against the exact
# arrangement, then the kernel tables of the round (default / bf16 / 1024 rays / frame render)
cd "$GRAFT_REPO_ROOT" || exit 1
out=gpurun_out/${1:-r05_c}
mkdir -p $out
export TMPDIR=/tmp
timeout 1200 python -m pytest tests/test_gpu_round5b.py -q -m gpu --timeout 600 > $out/pytest_n6.log 2>&1
echo "pytest n6 rc=$?"; tail -15 $out/pytest_n6.log | cut -c1-300
timeout 1500 python -m pytest "tests/test_gpu_parity.py" tests/test_gpu_round3.py::test_fp32x6_mode_full_forward_backward_vs_oracle \
   tests/test_gpu_round3.py::test_full_size_gradients_hip_and_fp32_oracle_against_fp64_oracle tests/test_gpu_round4b.py tests/test_gpu_round4.py tests/test_gpu_round5.py \
   -q -m gpu --timeout 900 -x > $out/pytest_around.log 2>&1
echo "pytest around rc=$?"; tail -8 $out/pytest_around.log | cut -c1-300
timeout 800 python tools/last2_soak.py 90 exact_infer,exact_infer,exact_keep,x6_infer,x6_keep > $out/last2_soak.log 2>&1
echo "last2 soak rc=$?"; grep -E "RESULT|launch" $out/last2_soak.log | cut -c1-400
bash tools/jobs/gpu_ab_flags.sh $(basename $out)/ab 2 "" "APP_X6=False" > /dev/null 2>&1
cat $out/ab/summary.txt
prof() {  # name, command...
  name=$1; shift
  ( cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats -d "$GRAFT_REPO_ROOT/$out/prof_$name" -o p -- "$@" > "$GRAFT_REPO_ROOT/$out/$name.out" 2> "$GRAFT_REPO_ROOT/$out/$name.log" )
  db=$(find $out/prof_$name -name "*.db" | head -1)
  python tools/rocprof_summary.py "$db" $out/kernel_stats_$name.txt
  rm -rf $out/prof_$name
}
prof fp32x6 python "$GRAFT_REPO_ROOT/bench.py" --no-cpu-baseline --no-extras --steps 20 --warmup 3
prof bf16 python "$GRAFT_REPO_ROOT/bench.py" --dtype bf16 --no-cpu-baseline --no-extras --steps 20 --warmup 3
prof rays1024 python "$GRAFT_REPO_ROOT/bench.py" --rays 1024 --inst-rays 1024 --no-cpu-baseline --no-extras --steps 20 --warmup 3
prof inference python "$GRAFT_REPO_ROOT/tools/inference_probe.py" fp32x6 32768
bash tools/jobs/gpu_timeline.sh $(basename $out)/tl1024 --rays 1024 --inst-rays 1024 > /dev/null 2>&1; tail -1 $out/tl1024/timeline.txt
bash tools/jobs/gpu_timeline.sh $(basename $out)/tl1024_nosync --rays 1024 --inst-rays 1024 --nosync > /dev/null 2>&1; tail -1 $out/tl1024_nosync/timeline.txt
for n in fp32x6 bf16 rays1024; do python -c "
import json; d=json.loads(open('$out/$n.out').read().strip().splitlines()[-1]); print('$n ms_per_step', d['ms_per_step'], 'median', d.get('step_ms_median'), 'frac', d['roofline']['frac'])"; done
tail -2 $out/inference.out
head -40 $out/kernel_stats_fp32x6.txt | cut -c1-150
timeout 1200 python tools/determinism_soak.py single fp32x6 ${2:-4000} 500 > $out/soak_fp32x6_n6.log 2>&1
echo "soak fp32x6 (n6 appearance kernels) rc=$?"; grep -E "RESULT|iter" $out/soak_fp32x6_n6.log | tail -6 | cut -c1-400
