#!/bin/bash
# round 4, job: sign-byte masks (ABI 15): unit tests, the suite, bench A/B is implicit (compare with profiles/r04_kernel_stats_fp32x6.txt)
cd "$GRAFT_REPO_ROOT" || exit 1
out=gpurun_out/${1:-r04_bm}
mkdir -p $out
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_gpu_round4.py -q --timeout 600 > $out/pytest_round4.log 2>&1
echo "round4 tests rc=$?" | tee -a $out/summary.txt
tail -n 4 $out/pytest_round4.log >> $out/summary.txt
timeout 2400 python -m pytest tests -q -m gpu --timeout 900 > $out/pytest_gpu.log 2>&1
echo "suite rc=$?" >> $out/summary.txt
tail -n 8 $out/pytest_gpu.log | grep -E "passed|failed|FAILED" >> $out/summary.txt
for i in 1 2; do
timeout 300 python bench.py --no-cpu-baseline --no-extras --steps 20 --warmup 3 > $out/bench_$i.json 2> $out/bench_$i.err
echo "bench rc=$? $(python -c "import json;d=json.load(open('$out/bench_$i.json'));print(d['ms_per_step'], d['step_ms_median'])")" >> $out/summary.txt
done
( cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats -d "$GRAFT_REPO_ROOT/$out/prof" -o p -- python "$GRAFT_REPO_ROOT/bench.py" --no-cpu-baseline --no-extras --steps 20 --warmup 3 > "$GRAFT_REPO_ROOT/$out/bench_under_rocprof.json" 2> "$GRAFT_REPO_ROOT/$out/prof.log" )
db=$(find $out/prof -name "*.db" | head -1)
python tools/rocprof_summary.py "$db" $out/kernel_stats.txt >> $out/summary.txt 2>&1
rm -rf $out/prof
cat $out/summary.txt; head -n 24 $out/kernel_stats.txt | cut -c1-150
