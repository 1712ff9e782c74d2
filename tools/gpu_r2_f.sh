#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out/r2f
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_gpu_parity.py -v -m gpu --timeout 200 -k "g12 or g13 or g14 or g15 or g3 or g10 or pq_scene or training_step or psnr" > gpurun_out/r2f/pytest_late.log 2>&1
echo "late parity rc=$?" | tee -a gpurun_out/r2f/summary.txt
grep -E "PASSED|FAILED|ERROR|Timeout" gpurun_out/r2f/pytest_late.log | tail -30 >> gpurun_out/r2f/summary.txt
timeout 600 python -m pytest tests/test_gpu_round2.py -q -m gpu --timeout 300 -k "full_size or last_two or appearance_head or weight_gradient" > gpurun_out/r2f/pytest_r2.log 2>&1
echo "round2 subset rc=$?" | tee -a gpurun_out/r2f/summary.txt
tail -5 gpurun_out/r2f/pytest_r2.log >> gpurun_out/r2f/summary.txt
timeout 300 python -m pytest tests/test_gpu_trainer_modes.py -q -m gpu --timeout 200 > gpurun_out/r2f/pytest_modes.log 2>&1
echo "modes rc=$?" | tee -a gpurun_out/r2f/summary.txt
tail -5 gpurun_out/r2f/pytest_modes.log >> gpurun_out/r2f/summary.txt
timeout 300 python tools/layer_probe.py 249000 2>&1 | tail -6 >> gpurun_out/r2f/summary.txt
timeout 600 python bench.py --no-cpu-baseline --no-extras > gpurun_out/r2f/bench.json 2> gpurun_out/r2f/bench.err
echo "bench rc=$?" >> gpurun_out/r2f/summary.txt
cat gpurun_out/r2f/summary.txt
