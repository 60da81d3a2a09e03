#!/bin/bash
# Round 3, Stage A continued: (a) the pair sweep without its hand-over wait (ablation build, results wrong) at 16 / 32 / 48 frames
# per launch -- the gate for Stage B is <= 0.40 ms per 16 frames; (b) SQ counters of the sweeps at 48 frames per launch.
export TMPDIR=/tmp; R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r03_stageA2; mkdir -p $O
for v in nopoll nopoll_nobar; do for B in 16 32 48; do
  CART_ENGINE_LIB=$R/cart-slam_amd/build/ab/$v/libcart_engine.so timeout -k 10 240 python3 $R/bench.py --no-cpu-baseline --no-pcie --no-overlap --steps 10 --plan pairs --batch $B --chunk $B > $O/${v}_b$B.json 2> $O/${v}_b$B.err || { echo "$v $B failed"; continue; }
  python3 -c 'import json,sys; d=json.loads(open(sys.argv[1]).read()); s=d["stages_ms_per_launch"]; B=int(sys.argv[3]); print(sys.argv[2], "B", B, "ms/16", round(d["ms_per_step"]*16/B,3), {k: round(v*16/B,3) for k,v in s.items()})' $O/${v}_b$B.json $v $B | tee -a $O/summary.txt
done; done
cd /tmp
for plan in pairs fused_up slabs; do
  timeout -k 10 300 rocprofv3 --kernel-trace --pmc SQ_WAVES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY --output-format csv -d $O/pmc_$plan/p1 -- python3 $R/bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-pcie --no-overlap --plan $plan --batch 48 --chunk 48 > $O/pmc_$plan.log 2>&1
  echo "pmc $plan p1 exit=$?"
  timeout -k 10 300 rocprofv3 --kernel-trace --pmc GRBM_GUI_ACTIVE --output-format csv -d $O/pmc_$plan/p2 -- python3 $R/bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-pcie --no-overlap --plan $plan --batch 48 --chunk 48 >> $O/pmc_$plan.log 2>&1
  echo "pmc $plan p2 exit=$?"
  python3 $R/profiles/pmc_summary.py $O/pmc_$plan > $O/pmc_${plan}_b48.txt; rm -rf $O/pmc_$plan
done
