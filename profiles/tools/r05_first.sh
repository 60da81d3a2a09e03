#!/bin/bash
# GPU box, round 5, first call: baseline bench line of this box, the format of rocprofv3's JSON counter records (per-instance values?), and the
# DRAM-destined read counters of the 1080p aggregation launch.
export TMPDIR=/tmp; R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r05_first; mkdir -p $O; cd $R
timeout -k 10 400 python3 bench.py > $O/bench_default.json 2> $O/bench_default.err; echo "bench default rc=$?" | tee -a $O/progress.txt
cd /tmp
timeout -k 10 300 rocprofv3 --kernel-trace --pmc TCC_EA0_WRREQ_STALL TCC_EA0_WRREQ --kernel-include-regex "aggregate_kernel" --output-format json csv -d $O/pmc_json -o j -- python3 $R/bench.py --steps 2 --warmup 1 --repeats 1 --no-cpu-baseline --no-pcie --no-bgr --no-overlap --placement-tries 8 > $O/pmc_json.log 2>&1; echo "pmc json rc=$?" | tee -a $O/progress.txt
ls -la $O/pmc_json/* | tee -a $O/progress.txt
C3="--width 1920 --height 1080 --disparities 256 --paths 8 --batch 4"
i=0
for pass in "TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_DRAM_sum TCC_BUBBLE_sum" "TCC_EA0_RDREQ_32B_sum TCC_EA0_RDREQ_DRAM_32B_sum TCC_HIT_sum TCC_MISS_sum" "TCC_EA0_RDREQ_GMI_32B_sum TCC_EA0_RDREQ_IO_32B_sum TCC_READ_sum TCC_REQ_sum"; do
  i=$((i+1))
  timeout -k 10 300 rocprofv3 --kernel-trace --pmc $pass --output-format csv -d $O/c3/p$i -- python3 $R/bench.py --steps 2 --warmup 1 --repeats 1 --no-cpu-baseline --no-pcie --no-bgr --no-overlap --placement-tries 1 $C3 > $O/c3_p$i.log 2>&1; echo "c3 pass $i rc=$? ($pass)" | tee -a $O/progress.txt
done
python3 $R/profiles/pmc_summary.py $O/c3 > $O/c3_pmc_summary.txt; rm -rf $O/c3
head -c 3000 $O/pmc_json/*results.json > $O/json_head.txt 2>/dev/null
echo done | tee -a $O/progress.txt
