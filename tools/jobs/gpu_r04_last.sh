#!/bin/bash
# end of round 4: the default bench line (with cpu baseline and extras), twice, and the rocprofv3 kernel table of the same command
cd "$GRAFT_REPO_ROOT" || exit 1
out=gpurun_out/${1:-r04_last}; mkdir -p $out
export TMPDIR=/tmp
for i in 1 2; do
  timeout 900 python bench.py > $out/bench_$i.json 2> $out/bench_$i.err
  echo "bench $i rc=$? $(python -c "import json;d=json.load(open('$out/bench_$i.json'));r=d['roofline'];print(d['ms_per_step'], d['step_ms_median'], d['step_ms_min_max'], r['frac'], r['hbm']['frac'], d.get('exact_fp32_ms_per_step'), d.get('bf16_ms_per_step'), d.get('rays1024_ms_per_step'), d.get('rays8192_ms_per_step'), d['cpu_baseline']['value'])")" | tee -a $out/summary.txt
done
( cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats -d "$GRAFT_REPO_ROOT/$out/prof" -o p -- python "$GRAFT_REPO_ROOT/bench.py" --no-cpu-baseline --no-extras > "$GRAFT_REPO_ROOT/$out/bench_under_rocprof.json" 2> "$GRAFT_REPO_ROOT/$out/prof.log" )
db=$(find $out/prof -name "*.db" | head -1)
python tools/rocprof_summary.py "$db" $out/kernel_stats.txt
rm -rf $out/prof
tail -1 $out/kernel_stats.txt
