#!/bin/bash
# GPU box: does a placement probe under sustained launches (A/B builds sus / sus2) pick placements that bench faster than the product's isolated probe?
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r04_probe_ab; mkdir -p $O; cd $R
lib() { [ "$1" = base ] && echo $R/cart-slam_amd/build/libcart_engine.so || echo $R/cart-slam_amd/build/ab/$1/libcart_engine.so; }
for r in $(seq ${ROUNDS:-5}); do for v in ${VARS:-base sus sus2}; do
  CART_ENGINE_LIB=$(lib $v) timeout -k 10 240 python3 bench.py --no-cpu-baseline --no-pcie --no-bgr --steps 20 --repeats 3 $ARGS > $O/${v}_$r.json 2> $O/${v}_$r.err || { tail -3 $O/${v}_$r.err; exit 1; }
  python3 -c 'import json,sys; d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1]); s=d["stages_ms_per_launch"]; p=d["placement_tuning"]; print("%-5s %8.1f pairs/s  agg %.4f  wta %.4f   tuning %5.2f s  %.3f -> %.3f ms" % (sys.argv[2], d["value"], s["aggregate"], s["wta"], p["seconds"], p["launch_pair_ms_first"], p["launch_pair_ms_kept"]))' $O/${v}_$r.json $v
done; done
