#!/bin/bash
# round 5, job G: the weight-gradient kernel's plane writes without bank conflicts: tests, time against rows, LDS counters, the bench line
cd "$GRAFT_REPO_ROOT" || exit 1
out=gpurun_out/${1:-r05_g}
mkdir -p $out
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_gpu_round3.py tests/test_gpu_round4.py tests/test_gpu_round5b.py -q -m gpu --timeout 600 > $out/pytest.log 2>&1
echo "pytest rc=$?"; tail -3 $out/pytest.log | cut -c1-300
timeout 300 python tools/x6_fixed_probe.py 0 2>/dev/null | grep -E "X6_ABL|wgrad" | tee $out/wgrad_after.txt | cut -c1-200
( cd /tmp && timeout 400 rocprofv3 --pmc SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_WAVE_CYCLES SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d "$GRAFT_REPO_ROOT/$out/pmc" -o p -- python "$GRAFT_REPO_ROOT/bench.py" --steps 3 --warmup 1 --no-cpu-baseline --no-extras > "$GRAFT_REPO_ROOT/$out/pmc.log" 2>&1 )
python tools/pmc_parse.py $out/pmc/ 2>/dev/null | grep -E "k_wgrad_x6|k_wgrad_n6" > $out/pmc_wgrad_after.txt; rm -rf $out/pmc
cat $out/pmc_wgrad_after.txt | cut -c1-200
for i in 1 2; do timeout 300 python bench.py --no-cpu-baseline --no-extras > $out/bench_$i.json 2>/dev/null; python -c "
import json; d=json.loads(open('$out/bench_$i.json').read().strip().splitlines()[-1]); print('bench', d['ms_per_step'], d['step_ms_median'], d['roofline']['frac'])"; done
