#!/bin/bash
# same-box A/B of engine switches as rocprofv3 kernel tables: usage gpu_ab_prof.sh <tag> "<flags A>" "<flags B>" ...   ("" = defaults)
cd "$GRAFT_REPO_ROOT" || exit 1
tag=$1; shift
out=gpurun_out/$tag; mkdir -p $out
export TMPDIR=/tmp
i=0
for f in "$@"; do
  i=$((i+1))
  ( cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats -d "$GRAFT_REPO_ROOT/$out/prof$i" -o p -- python "$GRAFT_REPO_ROOT/tools/ab_flag.py" "$f" --no-cpu-baseline --no-extras --steps 40 --warmup 8 > "$GRAFT_REPO_ROOT/$out/line$i.json" 2> "$GRAFT_REPO_ROOT/$out/prof$i.log" )
  db=$(find $out/prof$i -name "*.db" | head -1)
  python tools/rocprof_summary.py "$db" $out/kernel_stats$i.txt > /dev/null 2>&1
  rm -rf $out/prof$i
  echo "[$f] line: $(python -c "import json;d=json.load(open('$out/line$i.json'));print(d['ms_per_step'], d['step_ms_median'])")  kernels: $(tail -1 $out/kernel_stats$i.txt)" | tee -a $out/summary.txt
done
