# three instead of two rows of slab bytes in flight per wave in the 4-path fused sweeps (-DCART_FUSED_SETS_P4=3, build "s3") against two ("exp")
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/s3; mkdir -p $O
CART_ENGINE_LIB=$R/cart-slam_amd/build/ab/s3/libcart_engine.so timeout -k 10 400 python3 -m pytest $R/tests/test_gpu_parity.py -q -m gpu -k "fused or randomized or plans_agree or 1080p" 2>&1 | tail -3 || exit 1
run() { v=$1; name=$2; shift 2; CART_ENGINE_LIB=$R/cart-slam_amd/build/ab/$v/libcart_engine.so timeout -k 10 300 python3 $R/bench.py --no-cpu-baseline --no-pcie --steps 40 "$@" > $O/x.json 2> $O/x.err || { echo "$name $v failed"; tail -3 $O/x.err; return; }
  python3 -c 'import json,sys; d=json.loads(open(sys.argv[1]).read()); print(sys.argv[2], sys.argv[3], d["value"], d["ms_per_step"], d["stages_ms_per_launch"]["wta"])' $O/x.json $name $v; }
for rep in 1 2 3; do for v in exp s3; do
run $v ref --disparities 256 --paths 4
run $v d128p4_fused --disparities 128 --paths 4 --plan fused_up
run $v 1080p_p4 --width 1920 --height 1080 --disparities 256 --paths 4 --batch 4
run $v c1_fused --disparities 64 --paths 4 --plan fused_up
done; done
