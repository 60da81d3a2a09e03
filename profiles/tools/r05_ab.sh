#!/bin/bash
# GPU box: A/B of engine builds (cart-slam_amd/build/ab/<name>/libcart_engine.so, made by r05_variants.py) over bench configurations.
# usage: r05_ab.sh <tag> "<variant names>" [rounds]   ("base" = the product build)   -> gpurun_out/<tag>/summary.txt
# env: CONFIGS="c2 c1 ..." (default c2), PARITY_VARS (default: all variants; "" = none), PARITY_K (pytest -k filter), PMC_VARS (variants that also get
# a counter pass at the first configuration; PMC_PASSES="A B;C" = the counter sets, one pass each, default WRITE_SIZE and the write-stall counters)
export TMPDIR=/tmp; R=$GRAFT_REPO_ROOT; T=$1; VARS=$2; ROUNDS=${3:-2}; O=$R/gpurun_out/$T; mkdir -p $O; cd $R
# a GPU fault ends the whole call at once: no further GPU step after one (the box may be unusable, and repeating a fault can reset the host's GPUs)
faulted() { grep -qs "Memory access fault\|HSA_STATUS_ERROR\|core dumped" "$@" && { echo "GPU FAULT in $*: stopping" | tee -a $O/summary.txt; exit 9; }; return 0; }
lib() { [ "$1" = base ] && echo $R/cart-slam_amd/build/libcart_engine.so || echo $R/cart-slam_amd/build/ab/$1/libcart_engine.so; }
K=${PARITY_K:-"launch_plans_agree or xcd_placed or randomized_configurations or full_size_against_oracle or split_horizontal"}
for v in ${PARITY_VARS-$VARS}; do
  [ $v = base ] && continue
  CART_ENGINE_LIB=$(lib $v) timeout -k 10 600 python3 -m pytest tests/test_gpu_parity.py -x -q -k "$K" > $O/parity_$v.log 2>&1 \
    && echo "$v parity: $(tail -1 $O/parity_$v.log)" | tee -a $O/summary.txt || { echo "$v PARITY FAILED" | tee -a $O/summary.txt; tail -15 $O/parity_$v.log; faulted $O/parity_$v.log; exit 8; }
done
line() { python3 -c 'import json,sys; d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1]); s=d["stages_ms_per_launch"]; p=d.get("placement_tuning") or {}; print("%-8s %-10s %9.1f pairs/s  step %.4f  agg %.4f  wta %.4f  post+interp %.4f  frac %.4f  placement %s kept %.3f" % (sys.argv[2], sys.argv[3], d["value"], d["ms_per_step"], s.get("aggregate",0), s.get("wta",0), s.get("post",0)+s.get("interpolate",0), d["roofline"]["frac"], p.get("mode"), p.get("launch_pair_ms_kept",0)))' $1 $2 $3 | tee -a $O/summary.txt; }
declare -A CFG
CFG[c2]=""
CFG[c1]="--disparities 64 --paths 4"
CFG[ref]="--disparities 256 --paths 4"
CFG[c3]="--width 1920 --height 1080 --disparities 256 --paths 8 --batch 4"
CFG[c3p4]="--width 1920 --height 1080 --disparities 256 --paths 4 --batch 4"
CFG[c2b8]="--batch 8"
CFG[d128p4f]="--disparities 128 --paths 4 --plan fused_up"
CFG[d64p4f]="--disparities 64 --paths 4 --plan fused_up"
CFG[ref8]="--disparities 256 --paths 4 --batch 8"
CFG[c2b32]="--batch 32 --chunk 32"
for r in $(seq $ROUNDS); do
  for c in ${CONFIGS:-c2}; do
    for v in $VARS; do
      CART_ENGINE_LIB=$(lib $v) timeout -k 10 240 python3 bench.py --no-cpu-baseline --no-pcie --no-bgr --steps 20 --repeats 3 ${CFG[$c]} > $O/${c}_${v}_$r.json 2> $O/${c}_${v}_$r.err && line $O/${c}_${v}_$r.json $c $v || { echo "$c $v FAILED" | tee -a $O/summary.txt; tail -3 $O/${c}_${v}_$r.err; faulted $O/${c}_${v}_$r.err; exit 8; }
    done
  done
done
c=$(echo ${CONFIGS:-c2} | cut -d' ' -f1)
cd /tmp
for v in $PMC_VARS; do
  IFS=';' read -ra PASSES <<< "${PMC_PASSES:-WRITE_SIZE;TCC_EA0_WRREQ_STALL_sum TCC_EA0_WRREQ_sum TCC_TOO_MANY_EA_WRREQS_STALL_sum GRBM_GUI_ACTIVE}"
  for pass in "${PASSES[@]}"; do
    n=$(echo $pass | cut -d' ' -f1)
    CART_ENGINE_LIB=$(lib $v) timeout -k 10 300 rocprofv3 --kernel-trace --pmc $pass --kernel-include-regex "aggregate_kernel" --output-format csv -d $O/pmc_$v/$n -- python3 $R/bench.py --steps 3 --warmup 1 --repeats 1 --no-cpu-baseline --no-pcie --no-bgr --no-overlap ${CFG[$c]} > $O/pmc_${v}_$n.log 2>&1 || echo "pmc $v $n failed" | tee -a $O/summary.txt
  done
  echo "== counters $v ($c; bench steps only = the dispatches of a 16-frame batch after the placement probes are mixed in: see n)" >> $O/summary.txt
  python3 $R/profiles/pmc_summary.py $O/pmc_$v aggregate >> $O/summary.txt; rm -rf $O/pmc_$v
done
echo done | tee -a $O/summary.txt
