# D=64 / 4 paths with the second stream on: 3 or 2 four-wave workgroups per CU (and D=128 / 4 paths)
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/resid; mkdir -p $O
export CART_ENGINE_LIB=$R/cart-slam_amd/build/ab/exp/libcart_engine.so
for rep in 1 2 3 4; do for pad in 48213 75520; do
  CART_AGG_DYNLDS=$pad timeout -k 10 200 python3 $R/bench.py --no-cpu-baseline --no-pcie --steps 50 --disparities 64 --paths 4 > $O/x.json 2> $O/x.err || { echo failed; continue; }
  python3 -c 'import json,sys; d=json.loads(open(sys.argv[1]).read()); print("D64P4 pad", sys.argv[2], d["value"], d["ms_per_step"], d["stages_ms_per_launch"]["aggregate"])' $O/x.json $pad
done; done
for rep in 1 2 3; do for pad in 44117 71424; do
  CART_AGG_DYNLDS=$pad timeout -k 10 200 python3 $R/bench.py --no-cpu-baseline --no-pcie --steps 50 --disparities 128 --paths 4 > $O/x.json 2> $O/x.err || { echo failed; continue; }
  python3 -c 'import json,sys; d=json.loads(open(sys.argv[1]).read()); print("D128P4 pad", sys.argv[2], d["value"], d["ms_per_step"], d["stages_ms_per_launch"]["aggregate"])' $O/x.json $pad
done; done
