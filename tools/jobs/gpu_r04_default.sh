#!/bin/bash
# round 4, job 3: fp32x6 as the DEFAULT arithmetic: whole GPU suite, the default bench line, per-kernel probe, rocprof table, HBM counters, fp64 table
cd "$GRAFT_REPO_ROOT" || exit 1
tag=${1:-r04_default}
out=gpurun_out/$tag
mkdir -p $out
export TMPDIR=/tmp
timeout 2400 python -m pytest tests -q -m gpu --timeout 900 --durations=15 > $out/pytest_gpu.log 2>&1
echo "pytest gpu (default fp32x6) rc=$?" | tee -a $out/summary.txt
tail -n 30 $out/pytest_gpu.log | grep -E "passed|failed|FAILED|ERROR" >> $out/summary.txt
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -n 1 >> $out/summary.txt
timeout 900 python bench.py > $out/bench_default.json 2> $out/bench_default.err
echo "bench rc=$? $(python -c "import json;d=json.load(open('$out/bench_default.json'));print(d['ms_per_step'], d['roofline']['frac'], d['roofline']['hbm']['frac'], d.get('exact_fp32_ms_per_step'), d.get('bf16_ms_per_step'), d.get('rays1024_ms_per_step'), d.get('rays8192_ms_per_step'))")" >> $out/summary.txt
timeout 300 python tools/x6_ends_probe.py 249000 62000 > $out/x6_ends_probe.txt 2>&1
( cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats -d "$GRAFT_REPO_ROOT/$out/prof_x6" -o p -- python "$GRAFT_REPO_ROOT/bench.py" --no-cpu-baseline --no-extras --steps 20 --warmup 3 > "$GRAFT_REPO_ROOT/$out/bench_under_rocprof.json" 2> "$GRAFT_REPO_ROOT/$out/prof_x6.log" )
echo "prof rc=$?" >> $out/summary.txt
db=$(find $out/prof_x6 -name "*.db" | head -1)
python tools/rocprof_summary.py "$db" $out/kernel_stats_fp32x6.txt >> $out/summary.txt 2>&1
rm -rf $out/prof_x6
for set in FETCH_SIZE WRITE_SIZE; do
  ( cd /tmp && timeout 400 rocprofv3 --pmc $set --kernel-trace --output-format csv -d "$GRAFT_REPO_ROOT/$out/pmc_bench_${set}" -o p -- python "$GRAFT_REPO_ROOT/bench.py" --steps 3 --warmup 1 --no-cpu-baseline --no-extras > "$GRAFT_REPO_ROOT/$out/pmc_bench_${set}.log" 2>&1 )
  echo "pmc bench $set rc=$?" >> $out/summary.txt
done
python tools/pmc_parse.py $out/pmc_bench_* 2>/dev/null | grep -v "rocclr\|at::native" > $out/pmc_bench_table.txt
rm -rf $out/pmc_bench_FETCH_SIZE $out/pmc_bench_WRITE_SIZE
timeout 900 python -m pytest "tests/test_gpu_round3.py" -q -s -k "test_full_size_gradients_hip_and_fp32_oracle_against_fp64_oracle" > $out/fp64_table.log 2>&1
echo "fp64 table rc=$?" >> $out/summary.txt
cat $out/summary.txt; cat $out/x6_ends_probe.txt; grep -i "k_layer_x6\|k_wgrad_x6\|out_sum" $out/pmc_bench_table.txt | cut -c1-200
