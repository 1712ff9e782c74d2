#!/bin/bash
# round 4 iteration job: named test files first (fail fast), then optionally the whole suite, two bench lines, rocprof kernel table
# usage: gpu_r04_iter.sh <tag> <suite:0|1> <test files / -k expressions ...>
cd "$GRAFT_REPO_ROOT" || exit 1
tag=${1:-r04_iter}; suite=${2:-0}; shift 2
out=gpurun_out/$tag
mkdir -p $out
export TMPDIR=/tmp
if [ $# -gt 0 ]; then
  timeout 1200 python -m pytest "$@" -q -x --timeout 600 > $out/pytest_named.log 2>&1
  echo "named tests rc=$?" | tee -a $out/summary.txt
  tail -n 25 $out/pytest_named.log | cut -c1-300 >> $out/summary.txt
fi
if [ "$suite" = "1" ]; then
  timeout 2400 python -m pytest tests -q -m gpu --timeout 900 > $out/pytest_gpu.log 2>&1
  echo "suite rc=$?" >> $out/summary.txt
  tail -n 12 $out/pytest_gpu.log | grep -E "passed|failed|FAILED|ERROR" | cut -c1-300 >> $out/summary.txt
fi
for i in 1 2; do
timeout 300 python bench.py --no-cpu-baseline --no-extras --steps 20 --warmup 3 > $out/bench_$i.json 2> $out/bench_$i.err
echo "bench rc=$? $(python -c "import json;d=json.load(open('$out/bench_$i.json'));print(d['ms_per_step'], d['step_ms_median'])")" >> $out/summary.txt
done
( cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats -d "$GRAFT_REPO_ROOT/$out/prof" -o p -- python "$GRAFT_REPO_ROOT/bench.py" --no-cpu-baseline --no-extras --steps 20 --warmup 3 > "$GRAFT_REPO_ROOT/$out/bench_under_rocprof.json" 2> "$GRAFT_REPO_ROOT/$out/prof.log" )
db=$(find $out/prof -name "*.db" | head -1)
python tools/rocprof_summary.py "$db" $out/kernel_stats.txt >> $out/summary.txt 2>&1
rm -rf $out/prof
cat $out/summary.txt; head -n 44 $out/kernel_stats.txt | cut -c1-150
