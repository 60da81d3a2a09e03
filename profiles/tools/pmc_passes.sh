#!/bin/bash
# usage: pmc_passes.sh <outdir-under-gpurun_out> -- runs rocprofv3 --pmc passes over a short bench run.
# (counter groups that fit one pass each; TA_ADDR_STALLED_* hung rocprofv3 on this pool and is left out)
export TMPDIR=/tmp; R=$GRAFT_REPO_ROOT; OUT=$R/gpurun_out/$1; mkdir -p $OUT; cd /tmp
i=0
for pass in "SQ_WAVES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY" \
            "SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_WAIT_INST_LDS" \
            "TA_TA_BUSY TD_TD_BUSY GRBM_GUI_ACTIVE TCP_TOTAL_CACHE_ACCESSES" \
            "FETCH_SIZE" "WRITE_SIZE"; do
  i=$((i+1))
  rocprofv3 --kernel-trace --pmc $pass --output-format csv -d $OUT/p$i -- python3 $R/bench.py --steps 2 --warmup 1 --no-cpu-baseline > $OUT/p$i.log 2>&1
  echo "pass $i exit=$? ($pass)" | tee -a $OUT/progress.txt
done
