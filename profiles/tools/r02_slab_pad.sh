# distance between the slabs of two paths (stride = W*H*D + pad): does the HBM channel mapping of the 8 streams the WTA reads matter?
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/pad; mkdir -p $O
export CART_ENGINE_LIB=$R/cart-slam_amd/build/ab/exp/libcart_engine.so
for rep in 1 2; do for pad in 0 256 512 1024 2048 4096 8192 16384 65536 131328 1048832; do
  CART_SLAB_PAD=$pad timeout -k 10 200 python3 $R/bench.py --no-cpu-baseline --no-pcie --steps 40 > $O/x.json 2> $O/x.err || { echo failed $pad; tail -2 $O/x.err; continue; }
  python3 -c 'import json,sys; d=json.loads(open(sys.argv[1]).read()); print("pad", sys.argv[2], d["value"], d["ms_per_step"], d["stages_ms_per_launch"]["aggregate"], d["stages_ms_per_launch"]["wta"])' $O/x.json $pad
done; done
