#!/bin/bash
# GPU box, round 5: the fused sweep without its path arithmetic at the reference default (D=256, 4 paths) and at 1080p (timing builds)
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r05_fs2; mkdir -p $O; cd $R
for r in 1 2; do for c in "ref --disparities 256 --paths 4" "c3 --width 1920 --height 1080 --disparities 256 --paths 8 --batch 4"; do set -- $c; n=$1; shift; for v in base fs_noagg; do
  L=$R/cart-slam_amd/build/ab/$v/libcart_engine.so; [ $v = base ] && L=$R/cart-slam_amd/build/libcart_engine.so
  CART_ENGINE_LIB=$L timeout -k 10 200 python3 bench.py --no-cpu-baseline --no-pcie --no-bgr --steps 20 --repeats 3 "$@" > $O/${n}_${v}_$r.json 2> $O/err.txt || { tail -3 $O/err.txt; exit 1; }
  python3 -c 'import json,sys; d=json.loads(open(sys.argv[1]).read()); s=d["stages_ms_per_launch"]; print("%-4s %-10s %8.1f pairs/s  agg %.4f  sweep %.4f" % (sys.argv[3], sys.argv[2], d["value"], s["aggregate"], s["wta"]))' $O/${n}_${v}_$r.json $v $n | tee -a $O/summary.txt
done; done; done
