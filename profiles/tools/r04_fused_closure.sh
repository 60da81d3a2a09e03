#!/bin/bash
# GPU box: the numbers behind DESIGN.md section 8's fused-sweep closure (round 4).
#  (1) stage times of plan FUSED_UP at the headline size, 16 and 32 frames per launch: product / 128-VGPR builds / ablations
#  (2) SQ counters of wta_fused_kernel<8,8> in the product build and in the 128-VGPR build (one slab row in flight), 16 and 32 frames per launch
export TMPDIR=/tmp; R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r04_closure; mkdir -p $O; cd $R
lib() { [ "$1" = base ] && echo $R/cart-slam_amd/build/libcart_engine.so || echo $R/cart-slam_amd/build/ab/$1/libcart_engine.so; }
faulted() { grep -qs "Memory access fault\|HSA_STATUS_ERROR" "$@" && { echo "GPU FAULT: stopping" | tee -a $O/summary.txt; exit 9; }; return 0; }
for r in 1 2 3; do for v in base s2w4 s1w4 abl2 abl4 abl512 abl2048; do for n in 16 32; do
  [ $n = 32 ] && case $v in abl*) continue;; esac
  CART_ENGINE_LIB=$(lib $v) timeout -k 10 240 python3 bench.py --no-cpu-baseline --no-pcie --no-bgr --steps 20 --repeats 3 --plan fused_up --batch $n --chunk $n > $O/t.json 2> $O/t.err || { faulted $O/t.err; echo "$v $n FAILED"; exit 8; }
  python3 -c 'import json,sys; d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1]); s=d["stages_ms_per_launch"]; print("%-8s frames %2s  %8.1f pairs/s  agg %.4f  wta %.4f" % (sys.argv[2], sys.argv[3], d["value"], s["aggregate"], s["wta"]))' $O/t.json $v $n | tee -a $O/summary.txt
done; done; done
cd /tmp
for v in base s1w4; do for n in 16 32; do
  i=0
  for pass in "SQ_WAVES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY" "SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_INSTS_VMEM_RD GRBM_GUI_ACTIVE"; do
    i=$((i+1))
    CART_ENGINE_LIB=$(lib $v) timeout -k 10 240 rocprofv3 --kernel-trace --pmc $pass --output-format csv -d $O/pmc_${v}_$n/p$i -- python3 $R/bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-pcie --no-bgr --no-overlap --placement-tries 1 --plan fused_up --batch $n --chunk $n > $O/pmc_${v}_${n}_p$i.log 2>&1 || { faulted $O/pmc_${v}_${n}_p$i.log; echo "pmc $v $n pass $i failed"; }
  done
  echo "== $v, $n frames per launch" >> $O/pmc_summary.txt; python3 $R/profiles/pmc_summary.py $O/pmc_${v}_$n wta_fused >> $O/pmc_summary.txt; rm -rf $O/pmc_${v}_$n
done; done
rm -f $O/t.json $O/t.err $O/*.log
echo done >> $O/summary.txt
