#!/bin/bash
# Counter passes over the torch-free single-kernel harness (tools/pmc_harness.cpp): HBM bytes and matrix-pipe occupancy of the persistent
# 256 x 256 kernels at 249 000 rows.  One counter set per pass, --kernel-trace only.
cd "$GRAFT_REPO_ROOT" || exit 1
out=gpurun_out/${1:-pmc_harness}
mkdir -p $out
export TMPDIR=/tmp LD_LIBRARY_PATH=$GRAFT_REPO_ROOT/contrastive_lift_amd:$LD_LIBRARY_PATH
for mode in fwd gen outv dgrad wgrad; do ./tools/pmc_harness.bin $mode 249000 5 >> $out/summary.txt 2>&1; done
for set in "FETCH_SIZE" "WRITE_SIZE" "SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_INST_LDS SQ_INST_CYCLES_VMEM"; do
  tag=$(echo $set | cut -d' ' -f1)
  for mode in fwd gen outv dgrad wgrad; do
    ( cd /tmp && timeout 150 rocprofv3 --pmc $set --kernel-trace --output-format csv -d "$GRAFT_REPO_ROOT/$out/pmc_${tag}_${mode}" -o p -- "$GRAFT_REPO_ROOT/tools/pmc_harness.bin" $mode 249000 3 > "$GRAFT_REPO_ROOT/$out/pmc_${tag}_${mode}.log" 2>&1 )
  done
done
python tools/pmc_parse.py $out/pmc_* 2>/dev/null | grep -v rocclr > $out/pmc_table.txt
rm -rf $out/pmc_*/
cat $out/summary.txt; cat $out/pmc_table.txt
