#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out/r2e
export TMPDIR=/tmp
python tools/ab_persistent.py > gpurun_out/r2e/ab.txt 2>&1
timeout 1500 python -m pytest tests/test_gpu_round2.py tests/test_gpu_parity.py tests/test_gpu_trainer_modes.py -q -m gpu > gpurun_out/r2e/pytest.log 2>&1
echo "pytest rc=$?" | tee -a gpurun_out/r2e/summary.txt
tail -8 gpurun_out/r2e/pytest.log >> gpurun_out/r2e/summary.txt
python tools/layer_probe.py 249000 >> gpurun_out/r2e/summary.txt 2>&1
timeout 600 python bench.py --no-cpu-baseline --no-extras > gpurun_out/r2e/bench.json 2> gpurun_out/r2e/bench.err
echo "bench rc=$?" >> gpurun_out/r2e/summary.txt
cat gpurun_out/r2e/ab.txt gpurun_out/r2e/summary.txt
