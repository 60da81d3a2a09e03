#!/bin/bash
# usage: r02_pmc_mem.sh <tag> <bench args...>: memory-pipeline / address-translation counters of the bench's kernels (separate --pmc passes)
export TMPDIR=/tmp; R=$GRAFT_REPO_ROOT; T=$1; shift; OUT=$R/gpurun_out/$T; mkdir -p $OUT; cd /tmp
i=0
for pass in "TCP_UTCL1_REQUEST TCP_UTCL1_TRANSLATION_HIT TCP_UTCL1_TRANSLATION_MISS TCP_UTCL1_PERMISSION_MISS" \
            "TCP_UTCL1_STALL_INFLIGHT_MAX TCP_UTCL1_STALL_MULTI_MISS TCP_UTCL1_SERIALIZATION_STALL TCP_UTCL1_THRASHING_STALL" \
            "SQ_VMEM_TA_ADDR_FIFO_FULL SQ_VMEM_TA_CMD_FIFO_FULL SQ_VMEM_WR_TA_DATA_FIFO_FULL SQ_ACTIVE_INST_VMEM SQ_INST_LEVEL_VMEM SQ_LEVEL_WAVES SQ_BUSY_CYCLES SQ_INST_CYCLES_VMEM_WR" \
            "TCP_PENDING_STALL_CYCLES TCP_TCR_TCP_STALL_CYCLES TCP_TD_TCP_STALL_CYCLES TCP_TCP_TA_DATA_STALL_CYCLES" \
            "TCP_TCC_READ_REQ_LATENCY TCP_TCC_WRITE_REQ_LATENCY TCP_TCC_READ_REQ TCP_TCC_WRITE_REQ" \
            "TCC_EA0_WRREQ_STALL TCC_TOO_MANY_EA_WRREQS_STALL TCC_EA0_WRREQ_DRAM_CREDIT_STALL TCC_TAG_STALL" \
            "TCC_BUSY TCC_REQ TCC_HIT TCC_MISS" "GRBM_GUI_ACTIVE TD_TC_STALL TD_SPI_STALL TD_TD_BUSY"; do
  i=$((i+1))
  timeout -k 10 200 rocprofv3 --kernel-trace --pmc $pass --output-format csv -d $OUT/pmc/p$i -- python3 $R/bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-pcie --no-overlap "$@" > $OUT/p$i.log 2>&1
  echo "pass $i exit=$? ($pass)"
done
python3 $R/profiles/pmc_summary.py $OUT/pmc > $OUT/pmc_summary.txt; rm -rf $OUT/pmc
grep -A45 "aggregate_kernel" $OUT/pmc_summary.txt | head -48
