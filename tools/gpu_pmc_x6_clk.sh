#!/bin/bash
# (Round-3 record: CLIFT_X6_ABLATE selected timing-probe variants of the four-wave kernel that are no longer in the tree; results in
#  profiles/r03_pmc_x6_clock_with_without_stores.txt.)
# Cycles vs wall time of the fp32x6 forward kernel with and without its HBM writes (timing probes CLIFT_X6_ABLATE = 6 / 8): is the cost of the
# output stores cycles (stalls) or clock (power)?
cd "$GRAFT_REPO_ROOT" || exit 1
out=gpurun_out/${1:-pmc_x6_clk}
mkdir -p $out
export TMPDIR=/tmp LD_LIBRARY_PATH=$GRAFT_REPO_ROOT/contrastive_lift_amd:$LD_LIBRARY_PATH CLIFT_X6_W4=1
for abl in 0 6 8; do
  if [ $abl != 0 ]; then export CLIFT_X6_ABLATE=$abl; else unset CLIFT_X6_ABLATE; fi
  echo "ablate=$abl: $(./tools/pmc_harness.bin fwd 249000 20 2)" >> $out/summary.txt
  ( cd /tmp && timeout 150 rocprofv3 --pmc GRBM_GUI_ACTIVE SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY --kernel-trace --stats --output-format csv -d "$GRAFT_REPO_ROOT/$out/pmc_a${abl}" -o p -- "$GRAFT_REPO_ROOT/tools/pmc_harness.bin" fwd 249000 5 2 > "$GRAFT_REPO_ROOT/$out/pmc_a${abl}.log" 2>&1 )
  grep -h "k_layer_x6" $out/pmc_a${abl}/*/*kernel_stats.csv 2>/dev/null | cut -c1-160 >> $out/summary.txt
done
python tools/pmc_parse.py $out/pmc_*/ 2>/dev/null | grep -v rocclr > $out/pmc_table.txt
rm -rf $out/pmc_*/
cat $out/summary.txt; cat $out/pmc_table.txt
