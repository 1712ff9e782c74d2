#!/bin/bash
# round 5: counter evidence for the claims DESIGN.md makes about three kernels of the default step:
#   k_wgrad_x6 (what bounds it: matrix pipe / LDS / memory?), k_density_bwd_u + k_app_gather_bwd_u ("instruction-bound": VALU busy, atomics),
# and a clock / power trace of the device while the default bench step runs (the "power-bound" claim).
# One counter SET per pass, --kernel-trace only (gpurun refuses --pmc with the other trace domains).
cd "$GRAFT_REPO_ROOT" || exit 1
out=gpurun_out/${1:-r05_counters}
mkdir -p $out
export TMPDIR=/tmp
i=0
for set in "SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_INST_LDS SQ_INST_CYCLES_VMEM GRBM_GUI_ACTIVE" \
           "SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_LDS SQ_INSTS_SALU SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS" \
           "SQ_INSTS_VMEM SQ_INSTS_FLAT SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_FLAT SQ_WAVES SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_MISC GRBM_GUI_ACTIVE" \
           "TCC_EA0_ATOMIC_sum TCC_ATOMIC_sum TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum" "FETCH_SIZE" "WRITE_SIZE"; do
  i=$((i+1))
  ( cd /tmp && timeout 400 rocprofv3 --pmc $set --kernel-trace --output-format csv -d "$GRAFT_REPO_ROOT/$out/pmc_$i" -o p -- python "$GRAFT_REPO_ROOT/bench.py" --steps 3 --warmup 1 --no-cpu-baseline --no-extras > "$GRAFT_REPO_ROOT/$out/pmc_$i.log" 2>&1 )
  echo "pmc set $i ($set) rc=$?" >> $out/summary.txt
done
python tools/pmc_parse.py $out/pmc_*/ 2>/dev/null | grep -E "k_wgrad_x6|k_density_bwd_u|k_app_gather_bwd_u|k_layer_x6<false, false, 0, false, false>|k_layer_n128<40" > $out/pmc_table.txt
rm -rf $out/pmc_*/
# clock / power trace: sample the device every 50 ms while 200 steps of the default bench run (and 3 s of idle before / after)
( for k in $(seq 1 400); do echo "$(date +%s.%N) $(rocm-smi --showclocks --showpower --csv 2>/dev/null | tr '\n' ' ')"; sleep 0.05; done ) > $out/smi_trace.txt 2>&1 &
smi=$!
sleep 3
python bench.py --steps 600 --warmup 20 --no-cpu-baseline --no-extras > $out/bench_600.json 2> $out/bench_600.err
sleep 3
kill $smi 2>/dev/null
wc -l $out/smi_trace.txt; head -3 $out/smi_trace.txt | cut -c1-400
cat $out/summary.txt; cat $out/pmc_table.txt | cut -c1-200
