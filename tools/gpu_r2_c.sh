#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out/r2c
export TMPDIR=/tmp
timeout 2400 python -m pytest tests -q -m gpu --deselect tests/test_gpu_parity.py > gpurun_out/r2c/pytest_gpu_rest.log 2>&1
echo "pytest gpu (all but test_gpu_parity) rc=$?" | tee -a gpurun_out/r2c/summary.txt
tail -15 gpurun_out/r2c/pytest_gpu_rest.log >> gpurun_out/r2c/summary.txt
timeout 1200 python -m pytest tests/test_gpu_parity.py -q -m gpu > gpurun_out/r2c/pytest_gpu_parity.log 2>&1
echo "pytest test_gpu_parity rc=$?" | tee -a gpurun_out/r2c/summary.txt
tail -8 gpurun_out/r2c/pytest_gpu_parity.log >> gpurun_out/r2c/summary.txt
python -c "import __graft_entry__ as g; g.smoke()" >> gpurun_out/r2c/summary.txt 2>&1
cat gpurun_out/r2c/summary.txt
