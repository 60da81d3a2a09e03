# two-kernel WTA (HBM-bound): workgroups resident per CU capped with unused dynamic LDS (experiment build)
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/resid; mkdir -p $O
export CART_ENGINE_LIB=$R/cart-slam_amd/build/ab/exp/libcart_engine.so
for rep in 1 2; do for cfg in "" "--disparities_64_--paths_4"; do a=${cfg//_/ }; for pad in 0 8600 14000 22000 35900 46800; do
  CART_WTA_DYNLDS=$pad timeout -k 10 200 python3 $R/bench.py --no-cpu-baseline --no-pcie --steps 40 $a > $O/x.json 2> $O/x.err || { echo failed; continue; }
  python3 -c 'import json,sys; d=json.loads(open(sys.argv[1]).read()); print(sys.argv[2], "wta pad", sys.argv[3], d["value"], d["ms_per_step"], d["stages_ms_per_launch"]["aggregate"], d["stages_ms_per_launch"]["wta"])' $O/x.json "$a" $pad
done; done; done
