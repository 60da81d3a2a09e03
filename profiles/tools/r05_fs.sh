#!/bin/bash
# GPU box, round 5: what the fused sweep's memory access alone costs (timing builds, results wrong by design): plan FUSED_UP at the headline
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r05_fs; mkdir -p $O; cd $R
for r in 1 2 3; do for v in base fs_noagg fs_nocensus; do
  L=$R/cart-slam_amd/build/ab/$v/libcart_engine.so; [ $v = base ] && L=$R/cart-slam_amd/build/libcart_engine.so
  CART_ENGINE_LIB=$L timeout -k 10 200 python3 bench.py --no-cpu-baseline --no-pcie --no-bgr --steps 20 --repeats 3 --plan fused_up > $O/${v}_$r.json 2> $O/${v}_$r.err || { tail -3 $O/${v}_$r.err; exit 1; }
  python3 -c 'import json,sys; d=json.loads(open(sys.argv[1]).read()); s=d["stages_ms_per_launch"]; print("%-12s %8.1f pairs/s  agg %.4f  sweep %.4f" % (sys.argv[2], d["value"], s["aggregate"], s["wta"]))' $O/${v}_$r.json $v | tee -a $O/summary.txt
done; done
