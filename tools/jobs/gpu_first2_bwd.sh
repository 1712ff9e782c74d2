#!/bin/bash
# fused first-two-layers backward: unit tests, timing probe, a bench line, then the round-3 file with durations
cd "$GRAFT_REPO_ROOT" || exit 1
out=gpurun_out/${1:-first2_bwd}
mkdir -p $out
export TMPDIR=/tmp
python - <<'PY' | tee $out/summary.txt
import os, torch
print("affinity", len(os.sched_getaffinity(0)), "cpu_count", os.cpu_count(), "torch threads", torch.get_num_threads())
try: print("cpu.max", open("/sys/fs/cgroup/cpu.max").read().strip())
except OSError as e: print("cpu.max", e)
PY
timeout 900 python -m pytest tests/test_gpu_round3.py -q -m gpu -k "first2 or without_the_first" > $out/pytest_first2.log 2>&1; echo "first2 tests rc=$?" | tee -a $out/summary.txt
tail -4 $out/pytest_first2.log | tee -a $out/summary.txt
timeout 300 python tools/first2_bwd_probe.py 2>&1 | tee $out/probe.txt
timeout 900 python bench.py --no-cpu-baseline > $out/bench.json 2> $out/bench.err; echo "bench rc=$?" | tee -a $out/summary.txt
python - $out/bench.json <<'PY' | tee -a $out/summary.txt
import json,sys
d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
print("ms_per_step", d["ms_per_step"], "frac", d["roofline"]["frac"], d["roofline"]["instantiations"])
PY
timeout 1500 python -m pytest tests/test_gpu_round3.py -q -m gpu --durations=30 > $out/pytest_r3.log 2>&1; echo "round3 rc=$?" | tee -a $out/summary.txt
tail -45 $out/pytest_r3.log | tee -a $out/summary.txt
