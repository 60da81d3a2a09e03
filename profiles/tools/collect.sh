#!/bin/bash
# usage (GPU box): collect.sh <tag> <bench args...>: bench line (with CPU baseline / PCIe legs when FULL=1), rocprofv3 kernel stats of the
# same command, SQ / FETCH_SIZE / WRITE_SIZE passes; everything lands in gpurun_out/<tag>/ for copying into profiles/
export TMPDIR=/tmp; R=$GRAFT_REPO_ROOT; T=$1; shift; OUT=$R/gpurun_out/$T; mkdir -p $OUT
EXTRA="--no-cpu-baseline --no-pcie"; [ "$FULL" = 1 ] && EXTRA="--latency"
cd $R && timeout -k 10 400 python3 bench.py $EXTRA "$@" > $OUT/bench.json 2> $OUT/bench.err || { tail -5 $OUT/bench.err; exit 1; }
python3 -c 'import json,sys; d=json.loads(open(sys.argv[1]).read()); print(sys.argv[2], d["value"], d["ms_per_step"], d["stages_ms_per_launch"], d["roofline"]["frac"], d.get("value_pcie_inclusive"))' $OUT/bench.json $T
cd /tmp && timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats -o s -- python3 $R/bench.py --no-cpu-baseline --no-pcie --no-bgr "$@" > $OUT/bench_under_rocprof.json 2> $OUT/stats.log
f=$(ls $OUT/stats/*kernel_stats.csv 2>/dev/null | head -1); [ -n "$f" ] && cp $f $OUT/kernel_stats.csv && python3 $R/profiles/tools/kernel_avgs.py $f | head -4
# the HIP-event stage times of the SAME process the kernel stats come from (the two must agree; the unprofiled line above may sit a few per cent
# away from both: the profiler's host overhead opens gaps between the launches and the chip clocks higher inside them)
python3 -c 'import json,sys; d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1]); print("under rocprofv3:", d["value"], d["ms_per_step"], d["stages_ms_per_launch"])' $OUT/bench_under_rocprof.json
rm -rf $OUT/stats
i=0
for pass in "SQ_WAVES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY GRBM_GUI_ACTIVE" "FETCH_SIZE" "WRITE_SIZE"; do
  i=$((i+1))
  timeout -k 10 200 rocprofv3 --kernel-trace --pmc $pass --output-format csv -d $OUT/pmc/p$i -- python3 $R/bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-pcie --no-bgr --no-overlap "$@" > $OUT/pmc_p$i.log 2>&1 || echo "pass $i failed"
done
python3 $R/profiles/pmc_summary.py $OUT/pmc > $OUT/pmc_summary.txt; rm -rf $OUT/pmc $OUT/pmc_p*.log $OUT/stats.log
echo "collected $T"
# afterwards, in the repo: cp gpurun_out/<tag>/{bench.json,kernel_stats.csv,pmc_summary.txt} profiles/rNN_{bench,kernel_stats,pmc}_<cfg>.{json,csv,txt}
# and python3 profiles/tools/make_traffic.py profiles/rNN_pmc_<cfg>.txt W H D P <frames per launch> <plan> <round>
