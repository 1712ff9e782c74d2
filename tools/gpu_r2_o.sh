#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
out=gpurun_out/r2o
mkdir -p $out
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_gpu_round2.py -q -m gpu --timeout 600 -x -k "sync_free" > $out/pytest.log 2>&1
echo "pytest sync_free rc=$?" | tee -a $out/summary.txt
tail -25 $out/pytest.log | cut -c1-300 >> $out/summary.txt
for flag in "" "--nosync"; do
  timeout 600 python bench.py --no-cpu-baseline --no-extras $flag > $out/bench$flag.json 2> $out/bench$flag.err; echo "bench $flag rc=$?" >> $out/summary.txt
  timeout 600 python bench.py --rays 1024 --inst-rays 1024 --classes 2 --no-cpu-baseline --no-extras --steps 20 --warmup 5 $flag > $out/bench1024$flag.json 2> $out/bench1024$flag.err; echo "bench1024 $flag rc=$?" >> $out/summary.txt
done
python - <<'P' >> $out/summary.txt
import json,glob
for f in sorted(glob.glob('gpurun_out/r2o/bench*.json')):
    try:
        d=json.loads(open(f).read().strip().splitlines()[-1]); print(f, round(d['ms_per_step'],3), d.get('main_pass_ms'), d.get('instance_pass_ms'), d.get('f_active'))
    except Exception as e: print(f, 'ERR', e)
P
cat $out/summary.txt
