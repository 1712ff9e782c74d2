#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out/r2d
export TMPDIR=/tmp
timeout 1500 python -m pytest tests/test_gpu_round2.py -q -m gpu > gpurun_out/r2d/pytest_round2.log 2>&1
echo "pytest round2 rc=$?" | tee -a gpurun_out/r2d/summary.txt
tail -8 gpurun_out/r2d/pytest_round2.log >> gpurun_out/r2d/summary.txt
timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_trainer_modes.py -q -m gpu -x > gpurun_out/r2d/pytest_parity.log 2>&1
echo "pytest parity+modes rc=$?" | tee -a gpurun_out/r2d/summary.txt
tail -4 gpurun_out/r2d/pytest_parity.log >> gpurun_out/r2d/summary.txt
python tools/layer_probe.py 249000 >> gpurun_out/r2d/summary.txt 2>&1
python tools/layer_probe.py 62000 >> gpurun_out/r2d/summary.txt 2>&1
timeout 600 python bench.py --no-cpu-baseline > gpurun_out/r2d/bench.json 2> gpurun_out/r2d/bench.err
echo "bench rc=$?" >> gpurun_out/r2d/summary.txt
CLIFT_FUSE_FIRST2=0 timeout 600 python bench.py --no-cpu-baseline --no-extras > gpurun_out/r2d/bench_nofuse.json 2> gpurun_out/r2d/bench_nofuse.err
cat gpurun_out/r2d/summary.txt
