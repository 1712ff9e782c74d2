#!/bin/bash
# job T: two-phase density backward -- parity, then kernel time
cd "$GRAFT_REPO_ROOT" || exit 1
out=gpurun_out/r2t; mkdir -p $out; export TMPDIR=/tmp
timeout 1200 python -m pytest tests/test_gpu_parity.py tests/test_gpu_round2.py -q -m gpu -k "wave_walk or four_channel or golden_g6 or full_size or mid or lds_lines or LDS or large or density" --timeout 600 2>&1 | tail -8
cd /tmp
for mode in wave walk; do
rm -rf $GRAFT_REPO_ROOT/$out/prof
CLIFT_DENS_SCATTER=$mode rocprofv3 --kernel-trace --stats -d $GRAFT_REPO_ROOT/$out/prof -o step -- python $GRAFT_REPO_ROOT/bench.py --steps 5 --warmup 2 --no-extras --no-cpu-baseline > $GRAFT_REPO_ROOT/$out/bench_$mode.json 2>/dev/null
python - <<P
import sqlite3, glob, os, json
f = glob.glob(os.environ["GRAFT_REPO_ROOT"] + "/$out/prof/**/*.db", recursive=True)[0]
c = sqlite3.connect(f)
for r in c.execute("select name,total_calls,average,percentage from top_kernels limit 60"):
    if any(k in r[0] for k in ("density_bwd", "app_gather_bwd")): print("$mode", r[0][:40], r[1], round(r[2], 1))
print(json.loads(open(os.environ["GRAFT_REPO_ROOT"] + "/$out/bench_$mode.json").read())["ms_per_step"])
P
done
