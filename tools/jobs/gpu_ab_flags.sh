#!/bin/bash
# same-box A/B of engine switches: usage gpu_ab_flags.sh <tag> <reps> "<flags A>" "<flags B>" ...   ("" = defaults)
cd "$GRAFT_REPO_ROOT" || exit 1
tag=$1; reps=$2; shift 2
out=gpurun_out/$tag; mkdir -p $out
for rep in $(seq 1 $reps); do
  for f in "$@"; do
    timeout 300 python tools/ab_flag.py "$f" --no-cpu-baseline --no-extras --steps 40 --warmup 8 > $out/line.json 2> $out/err.txt
    echo "rep $rep [$f] $(python -c "import json;d=json.load(open('$out/line.json'));print(d['ms_per_step'], d['step_ms_median'])")" | tee -a $out/summary.txt
  done
done
