#!/bin/bash
# round 5, job A: the round's new tests, then kernel tables of the bf16 step, the 1024-ray step and the frame render
cd "$GRAFT_REPO_ROOT" || exit 1
out=gpurun_out/${1:-r05_a}
mkdir -p $out
export TMPDIR=/tmp
timeout 1500 python -m pytest tests/test_gpu_round5.py "tests/test_gpu_parity.py::test_g12_reference_training_steps_on_gpu" \
   "tests/test_gpu_round3.py::test_bf16_mode_against_the_oracle_at_the_bench_shape" "tests/test_gpu_round3.py::test_bf16_mode_against_the_oracle_at_the_messy_rooms_shape" \
   tests/test_gpu_round4b.py tests/test_gpu_trainer_modes.py -q -m gpu --timeout 900 -x > $out/pytest.log 2>&1
echo "pytest rc=$?"; tail -25 $out/pytest.log
prof() {  # name, command...
  name=$1; shift
  ( cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats -d "$GRAFT_REPO_ROOT/$out/prof_$name" -o p -- "$@" > "$GRAFT_REPO_ROOT/$out/$name.out" 2> "$GRAFT_REPO_ROOT/$out/$name.log" )
  db=$(find $out/prof_$name -name "*.db" | head -1)
  python tools/rocprof_summary.py "$db" $out/kernel_stats_$name.txt
  rm -rf $out/prof_$name
}
prof bf16 python "$GRAFT_REPO_ROOT/bench.py" --dtype bf16 --no-cpu-baseline --no-extras --steps 20 --warmup 3
prof rays1024 python "$GRAFT_REPO_ROOT/bench.py" --rays 1024 --inst-rays 1024 --no-cpu-baseline --no-extras --steps 20 --warmup 3
prof inference python "$GRAFT_REPO_ROOT/tools/inference_probe.py" fp32x6 32768
for n in bf16 rays1024; do python -c "
import json; d=json.loads(open('$out/$n.out').read().strip().splitlines()[-1]); print('$n ms_per_step', d['ms_per_step'], 'frac', d['roofline']['frac'])"; done
tail -2 $out/inference.out
head -32 $out/kernel_stats_bf16.txt | cut -c1-150
