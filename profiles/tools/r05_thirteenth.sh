#!/bin/bash
# GPU box, round 5: slab groups from hipExtMallocWithFlags (uncached / fine-grained) against plain hipMalloc; placement search OFF so that the runs sample the modes
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r05_memkind; mkdir -p $O; cd $R
for v in uc fg; do
  CART_ENGINE_LIB=$R/cart-slam_amd/build/ab/$v/libcart_engine.so timeout -k 10 300 python3 -m pytest tests/test_gpu_parity.py -x -q -k "launch_plans_agree_at_full_size or stage_by_stage" > $O/parity_$v.log 2>&1 && echo "$v parity: $(tail -1 $O/parity_$v.log)" | tee -a $O/summary.txt || { echo "$v PARITY FAILED"; tail -15 $O/parity_$v.log; exit 8; }
done
for r in 1 2 3 4 5; do for v in base uc fg; do
  L=$R/cart-slam_amd/build/ab/$v/libcart_engine.so; [ $v = base ] && L=$R/cart-slam_amd/build/libcart_engine.so
  CART_ENGINE_LIB=$L timeout -k 10 200 python3 bench.py --no-cpu-baseline --no-pcie --no-bgr --steps 20 --repeats 3 --placement-tries 1 > $O/${v}_$r.json 2> $O/${v}_$r.err || { tail -3 $O/${v}_$r.err; exit 1; }
  python3 -c 'import json,sys; d=json.loads(open(sys.argv[1]).read()); s=d["stages_ms_per_launch"]; print("%-6s %8.1f pairs/s  agg %.4f  wta %.4f" % (sys.argv[2], d["value"], s["aggregate"], s["wta"]))' $O/${v}_$r.json $v | tee -a $O/summary.txt
done; done
