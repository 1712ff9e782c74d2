#!/bin/bash
# Counter passes over the fp32x6 persistent layer kernels (csrc/layer_x6.hip) through the torch-free harness.  One counter set per pass.
cd "$GRAFT_REPO_ROOT" || exit 1
out=gpurun_out/${1:-pmc_x6}
mkdir -p $out
export TMPDIR=/tmp LD_LIBRARY_PATH=$GRAFT_REPO_ROOT/contrastive_lift_amd:$LD_LIBRARY_PATH
: > $out/summary.txt
for mode in fwd dgrad; do ./tools/pmc_harness.bin $mode 249000 20 2 >> $out/summary.txt 2>&1; ./tools/pmc_harness.bin $mode 249000 20 0 >> $out/summary.txt 2>&1; done
i=0
for set in "SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_INST_LDS SQ_INST_CYCLES_VMEM" \
           "SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_LDS SQ_INSTS_SALU SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS" \
           "GRBM_GUI_ACTIVE SQ_ACTIVE_INST_MISC SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_VMEM SQ_INSTS_VMEM SQ_WAVES" "FETCH_SIZE" "WRITE_SIZE"; do
  i=$((i+1))
  for mode in ${MODES:-fwd}; do
    ( cd /tmp && timeout 150 rocprofv3 --pmc $set --kernel-trace --output-format csv -d "$GRAFT_REPO_ROOT/$out/pmc_${i}_${mode}" -o p -- "$GRAFT_REPO_ROOT/tools/pmc_harness.bin" $mode 249000 3 2 > "$GRAFT_REPO_ROOT/$out/pmc_${i}_${mode}.log" 2>&1 )
  done
done
python tools/pmc_parse.py $out/pmc_*/ 2>/dev/null | grep -v rocclr > $out/pmc_table.txt
rm -rf $out/pmc_*/
cat $out/summary.txt; cat $out/pmc_table.txt
