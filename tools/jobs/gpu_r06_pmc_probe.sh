#!/bin/bash
# round 6: counters of ONE kernel under a probe script.  usage: gpu_r06_pmc_probe.sh <outdir> <kernel-name-substring> <probe command...>
cd "$GRAFT_REPO_ROOT" || exit 1
out=gpurun_out/$1; kern=$2; shift 2
mkdir -p $out
export TMPDIR=/tmp
i=0
for set in "SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_INST_LDS SQ_INST_CYCLES_VMEM SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE" \
           "SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_LDS SQ_INSTS_SALU SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS" \
           "SQ_INSTS_VMEM SQ_INSTS_FLAT SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_FLAT SQ_WAVES SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_MISC SQ_INSTS_SMEM" \
           "SQ_INST_CYCLES_SALU SQ_INSTS_BRANCH SQ_INSTS_CBRANCH_TAKEN SQ_IFETCH SQ_WAIT_ANY SQ_LDS_ADDR_CONFLICT SQ_LDS_UNALIGNED_STALL SQ_INSTS_GDS" \
           "TCC_EA0_ATOMIC_sum TCC_ATOMIC_sum TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum TCP_TCC_ATOMIC_WITHOUT_RET_REQ_sum TCP_PENDING_STALL_CYCLES_sum TCP_TA_TCP_STATE_READ_sum"; do
  i=$((i+1))
  ( cd /tmp && timeout 300 rocprofv3 --pmc $set --kernel-trace --output-format csv -d "$GRAFT_REPO_ROOT/$out/pmc_$i" -o p -- "$@" > "$GRAFT_REPO_ROOT/$out/pmc_$i.log" 2>&1 )
  echo "pmc set $i rc=$?" >> $out/summary.txt
done
python tools/pmc_parse.py $out/pmc_*/ 2>/dev/null | grep -E "$kern" > $out/pmc_table.txt
rm -rf $out/pmc_*/
cat $out/summary.txt; cut -c1-170 $out/pmc_table.txt
