#!/bin/bash
# HBM write-request counters of the fp32x6 forward kernel (are the output lines leaving L2 as whole 64-byte requests?)
cd "$GRAFT_REPO_ROOT" || exit 1
out=gpurun_out/${1:-pmc_x6_wr}
mkdir -p $out
export TMPDIR=/tmp LD_LIBRARY_PATH=$GRAFT_REPO_ROOT/contrastive_lift_amd:$LD_LIBRARY_PATH
i=0
for set in "TCC_EA0_WRREQ_sum TCC_EA0_WRREQ_64B_sum TCC_EA0_WRREQ_DRAM_CREDIT_STALL_sum TCC_EA0_WRREQ_LEVEL_sum" "TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum TCC_EA0_RDREQ_DRAM_CREDIT_STALL_sum TCC_EA0_RDREQ_LEVEL_sum"; do
  i=$((i+1))
  for prec in 2 0; do
    ( cd /tmp && timeout 150 rocprofv3 --pmc $set --kernel-trace --output-format csv -d "$GRAFT_REPO_ROOT/$out/pmc_${i}_p${prec}" -o p -- "$GRAFT_REPO_ROOT/tools/pmc_harness.bin" fwd 249000 3 $prec > "$GRAFT_REPO_ROOT/$out/pmc_${i}_p${prec}.log" 2>&1 )
  done
done
python tools/pmc_parse.py $out/pmc_*/ 2>/dev/null | grep -v rocclr > $out/pmc_table.txt
rm -rf $out/pmc_*/
cat $out/pmc_table.txt
