#!/bin/bash
# GPU box: A/B of engine builds (cart-slam_amd/build/ab/<name>/libcart_engine.so, made by build_variant.sh) over bench configurations.
# usage: r04_ab.sh <tag> "<variant names>" [rounds]   ("base" = the product build)   -> gpurun_out/<tag>/summary.txt
# Every variant first passes the fused-sweep parity tests; the configurations alternate between the variants inside one process-per-run series.
export TMPDIR=/tmp; R=$GRAFT_REPO_ROOT; T=$1; VARS=$2; ROUNDS=${3:-2}; O=$R/gpurun_out/$T; mkdir -p $O; cd $R
# a GPU fault ends the whole call at once: no further GPU step after one (the box may be unusable, and repeating a fault can reset the host's GPUs)
faulted() { grep -qs "Memory access fault\|HSA_STATUS_ERROR\|core dumped" "$@" && { echo "GPU FAULT in $*: stopping" | tee -a $O/summary.txt; exit 9; }; return 0; }
lib() { [ "$1" = base ] && echo $R/cart-slam_amd/build/libcart_engine.so || echo $R/cart-slam_amd/build/ab/$1/libcart_engine.so; }
for v in ${PARITY_VARS-$VARS}; do   # PARITY_VARS="" skips the parity step (ablation builds compute wrong results by design)
  CART_ENGINE_LIB=$(lib $v) timeout -k 10 600 python3 -m pytest tests/test_gpu_parity.py -x -q -k "fused_wta_path or launch_plans_agree or xcd_placed or full_size_oracle_1080p or randomized_configurations" > $O/parity_$v.log 2>&1 \
    && echo "$v parity: $(tail -1 $O/parity_$v.log)" >> $O/summary.txt || { echo "$v PARITY FAILED" >> $O/summary.txt; tail -15 $O/parity_$v.log; faulted $O/parity_$v.log; exit 8; }
done
line() { python3 -c 'import json,sys; d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1]); s=d["stages_ms_per_launch"]; print("%-10s %-12s %9.1f pairs/s  step %.4f  agg %.4f  wta %.4f  frac_moved %.4f  plan %s" % (sys.argv[2], sys.argv[3], d["value"], d["ms_per_step"], s.get("aggregate",0), s.get("wta",0), d["roofline"]["frac_moved"], d["config"]["launch_plan"]["plan"]))' $1 $2 $3 >> $O/summary.txt; tail -1 $O/summary.txt; }
declare -A CFG
CFG[ref]="--disparities 256 --paths 4"
CFG[c3]="--width 1920 --height 1080 --disparities 256 --paths 8 --batch 4"
CFG[c2f]="--plan fused_up"
CFG[c2]=""
CFG[d256p8]="--disparities 256 --paths 8"
CFG[c2f32]="--plan fused_up --batch 32 --chunk 32"
CFG[c3p4]="--width 1920 --height 1080 --disparities 256 --paths 4 --batch 4"
CFG[ref6]="--disparities 256 --paths 4 --batch 6"
CFG[c2b12]="--batch 12"
CFG[c1]="--disparities 64 --paths 4"
CFG[c2b4]="--batch 4"
CFG[c1b4]="--disparities 64 --paths 4 --batch 4"
CFG[ref8]="--disparities 256 --paths 4 --batch 8"
CFG[ref12]="--disparities 256 --paths 4 --batch 12"
CFG[c2b8]="--batch 8"
CFG[d128p4]="--disparities 128 --paths 4"
CFG[c1b8]="--disparities 64 --paths 4 --batch 8"
CFG[d64p8]="--disparities 64 --paths 8"
CFG[d64p8b8]="--disparities 64 --paths 8 --batch 8"
CFG[c1b32]="--disparities 64 --paths 4 --batch 32 --chunk 32"
CFG[c1b24]="--disparities 64 --paths 4 --batch 24 --chunk 24"
CFG[d128p4b8]="--disparities 128 --paths 4 --batch 8"
CFG[ref32]="--disparities 256 --paths 4 --batch 32 --chunk 32"
for r in $(seq $ROUNDS); do
  for c in ${CONFIGS:-ref c3 c2f d256p8}; do
    for v in $VARS; do
      CART_ENGINE_LIB=$(lib $v) timeout -k 10 240 python3 bench.py --no-cpu-baseline --no-pcie --no-bgr --steps 20 --repeats 3 ${CFG[$c]} > $O/${c}_${v}_$r.json 2> $O/${c}_${v}_$r.err && line $O/${c}_${v}_$r.json $c $v || { echo "$c $v FAILED" >> $O/summary.txt; tail -3 $O/${c}_${v}_$r.err; faulted $O/${c}_${v}_$r.err; exit 8; }
    done
  done
done
echo done >> $O/summary.txt
