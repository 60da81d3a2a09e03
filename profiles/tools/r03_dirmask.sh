#!/bin/bash
# Per-direction launch times of the aggregation kernel alone (experiment build, CART_DEBUG_DIRMASK; results wrong by construction):
# what do the W-step horizontal scans cost when nothing shares their SIMDs?
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r03_dirmask; mkdir -p $O
export CART_ENGINE_LIB=$R/cart-slam_amd/build/ab/exp/libcart_engine.so
for cfg in "64 4" "128 8" "256 4"; do set -- $cfg
  for mask in 0xff 0x3 0x1 0xc 0xfc; do
    for B in 16 1; do
      D=$1 P=$2 B=$B TAG="D=$1 P=$2 B=$B mask=$mask" CART_DEBUG_DIRMASK=$mask timeout -k 10 120 python3 $R/profiles/tools/disparity_only.py 2>/dev/null | tee -a $O/summary.txt
    done
  done
done
