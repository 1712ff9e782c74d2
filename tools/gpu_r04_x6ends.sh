#!/bin/bash
# round 4, job 2: the fp32x6 fused ends (ABI 14): unit tests, the forced-mode suite pieces that run through them, A/B of the bench step, rocprof table
cd "$GRAFT_REPO_ROOT" || exit 1
tag=${1:-r04_x6ends}
out=gpurun_out/$tag
mkdir -p $out
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_gpu_round4.py -q -x --timeout 600 > $out/pytest_round4.log 2>&1
echo "round4 tests rc=$?" | tee -a $out/summary.txt
tail -n 25 $out/pytest_round4.log >> $out/summary.txt
for fe in 0 1; do
  CLIFT_X6_FUSED_ENDS=$fe timeout 300 python bench.py --dtype fp32x6 --no-cpu-baseline --no-extras --steps 40 --warmup 8 > $out/bench_x6_fused$fe.json 2> $out/bench_x6_fused$fe.err
  echo "bench fused_ends=$fe rc=$? $(python -c "import json;d=json.load(open('$out/bench_x6_fused$fe.json'));print(d['ms_per_step'])")" >> $out/summary.txt
done
( cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats -d "$GRAFT_REPO_ROOT/$out/prof_x6" -o p -- python "$GRAFT_REPO_ROOT/bench.py" --dtype fp32x6 --no-cpu-baseline --no-extras --steps 20 --warmup 3 > "$GRAFT_REPO_ROOT/$out/bench_fp32x6_under_rocprof.json" 2> "$GRAFT_REPO_ROOT/$out/prof_x6.log" )
echo "prof rc=$?" >> $out/summary.txt
db=$(find $out/prof_x6 -name "*.db" | head -1)
python tools/rocprof_summary.py "$db" $out/kernel_stats_fp32x6.txt >> $out/summary.txt 2>&1
rm -rf $out/prof_x6
CLIFT_FORCE_MLP_DTYPE=fp32x6 timeout 1500 python -m pytest tests -q -m gpu --timeout 900 -x > $out/pytest_forced_x6.log 2>&1
echo "forced x6 suite rc=$?" >> $out/summary.txt
tail -n 6 $out/pytest_forced_x6.log >> $out/summary.txt
cat $out/summary.txt; head -n 40 $out/kernel_stats_fp32x6.txt
