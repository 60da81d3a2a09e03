#!/bin/bash
# usage (GPU box): r02_profile_config.sh <tag> <pmc: 0|1> <bench args...>
# bench line + rocprofv3 kernel stats of the same command (+ PMC passes) for one configuration; outputs under gpurun_out/<tag>/
export TMPDIR=/tmp; R=$GRAFT_REPO_ROOT; T=$1; PMC=$2; shift 2; OUT=$R/gpurun_out/$T; mkdir -p $OUT
cd $R && timeout -k 10 300 python3 bench.py --no-cpu-baseline --no-pcie "$@" > $OUT/bench.json 2> $OUT/bench.err || { tail -5 $OUT/bench.err; exit 1; }
python3 -c 'import json,sys; d=json.loads(open(sys.argv[1]).read()); print(d["value"], d["ms_per_step"], d["stages_ms_per_launch"], d["roofline"]["frac"])' $OUT/bench.json
cd /tmp && timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats -o s -- python3 $R/bench.py --no-cpu-baseline --no-pcie "$@" > $OUT/stats.log 2>&1
echo "kernel stats exit=$?"; f=$(ls $OUT/stats/*kernel_stats.csv 2>/dev/null | head -1); [ -n "$f" ] && cp $f $OUT/kernel_stats.csv && python3 $R/profiles/tools/kernel_avgs.py $f | head -12
rm -rf $OUT/stats
if [ "$PMC" = 1 ]; then
  i=0
  for pass in "SQ_WAVES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY" \
              "SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_WAIT_INST_LDS" \
              "TA_TA_BUSY TD_TD_BUSY GRBM_GUI_ACTIVE TCP_TOTAL_CACHE_ACCESSES" "FETCH_SIZE" "WRITE_SIZE"; do
    i=$((i+1))
    timeout -k 10 300 rocprofv3 --kernel-trace --pmc $pass --output-format csv -d $OUT/pmc/p$i -- python3 $R/bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-pcie "$@" > $OUT/pmc_p$i.log 2>&1
    echo "pass $i exit=$?"
  done
  python3 $R/profiles/pmc_summary.py $OUT/pmc > $OUT/pmc_summary.txt; rm -rf $OUT/pmc
  grep -A40 "aggregate_kernel" $OUT/pmc_summary.txt | head -45
fi
