#!/bin/bash
# Round-3 evidence job: rocprofv3 --kernel-trace --stats tables of the bench step in its three precisions (exact fp32 = the headline, fp32x6,
# bf16), and the HBM-byte counter passes over the exact-fp32 bench.  Summaries land in gpurun_out/<tag>/ (copy into profiles/ to keep them).
cd "$GRAFT_REPO_ROOT" || exit 1
tag=${1:-r03_prof}
out=gpurun_out/$tag
mkdir -p $out
export TMPDIR=/tmp
for dt in fp32 fp32x6 bf16; do
  ( cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats -d "$GRAFT_REPO_ROOT/$out/prof_$dt" -o p -- python "$GRAFT_REPO_ROOT/bench.py" --dtype $dt --no-cpu-baseline --no-extras --steps 20 --warmup 3 > "$GRAFT_REPO_ROOT/$out/bench_${dt}_under_rocprof.json" 2> "$GRAFT_REPO_ROOT/$out/prof_$dt.log" )
  echo "prof $dt rc=$?" >> $out/summary.txt
  db=$(find $out/prof_$dt -name "*.db" | head -1)
  python tools/rocprof_summary.py "$db" $out/kernel_stats_$dt.txt >> $out/summary.txt 2>&1
  rm -rf $out/prof_$dt
done
for set in FETCH_SIZE WRITE_SIZE; do
  ( cd /tmp && timeout 400 rocprofv3 --pmc $set --kernel-trace --output-format csv -d "$GRAFT_REPO_ROOT/$out/pmc_bench_${set}" -o p -- python "$GRAFT_REPO_ROOT/bench.py" --steps 3 --warmup 1 --no-cpu-baseline --no-extras > "$GRAFT_REPO_ROOT/$out/pmc_bench_${set}.log" 2>&1 )
  echo "pmc bench $set rc=$?" >> $out/summary.txt
done
python tools/pmc_parse.py $out/pmc_bench_* 2>/dev/null | grep -v "rocclr\|at::native" > $out/pmc_bench_table.txt
rm -rf $out/pmc_bench_FETCH_SIZE $out/pmc_bench_WRITE_SIZE
python bench.py > $out/bench_default.json 2> $out/bench_default.err
echo "bench rc=$?" >> $out/summary.txt
cat $out/summary.txt; head -30 $out/kernel_stats_fp32.txt
