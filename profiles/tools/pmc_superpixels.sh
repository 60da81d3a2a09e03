#!/bin/bash
# usage (GPU box): bash profiles/tools/pmc_superpixels.sh <outdir-under-gpurun_out> -- PMC passes over profiles/tools/superpixel_timing.py
export TMPDIR=/tmp; R=$GRAFT_REPO_ROOT; OUT=$R/gpurun_out/$1; mkdir -p $OUT; cd /tmp
i=0
for pass in "SQ_WAVES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY" \
            "SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_WAIT_INST_LDS" \
            "TA_TA_BUSY TD_TD_BUSY GRBM_GUI_ACTIVE TCP_TOTAL_CACHE_ACCESSES"; do
  i=$((i+1))
  timeout -k 10 300 rocprofv3 --kernel-trace --pmc $pass --output-format csv -d $OUT/p$i -- python3 $R/profiles/tools/superpixel_timing.py > $OUT/p$i.log 2>&1
  echo "pass $i exit=$? ($pass)" | tee -a $OUT/progress.txt
done
python3 $R/profiles/pmc_summary.py $OUT > $OUT/pmc_summary.txt
