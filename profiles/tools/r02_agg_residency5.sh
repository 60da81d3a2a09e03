# headline configuration with the second stream on: all resident / 6 / 5 / 4 workgroups per CU
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/resid; mkdir -p $O
export CART_ENGINE_LIB=$R/cart-slam_amd/build/ab/exp/libcart_engine.so
for rep in 1 2 3 4; do for pad in 0 19456 25600 22528; do
  CART_AGG_DYNLDS=$pad timeout -k 10 200 python3 $R/bench.py --no-cpu-baseline --no-pcie --steps 50 > $O/x.json 2> $O/x.err || { echo failed; continue; }
  python3 -c 'import json,sys; d=json.loads(open(sys.argv[1]).read()); print("pad", sys.argv[2], d["value"], d["ms_per_step"], d["stages_ms_per_launch"]["aggregate"], d["stages_ms_per_launch"]["wta"])' $O/x.json $pad
done; done
