# A/B on one box: right-view partial rows as u16 keys (exp = working tree) against u32 (prev = the commit before)
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/rv16; mkdir -p $O
run() { v=$1; name=$2; shift 2; CART_ENGINE_LIB=$R/cart-slam_amd/build/ab/$v/libcart_engine.so timeout -k 10 300 python3 $R/bench.py --no-cpu-baseline --no-pcie --steps 40 "$@" > $O/x.json 2> $O/x.err || { echo "$name failed"; tail -3 $O/x.err; return; }
  python3 -c 'import json,sys; d=json.loads(open(sys.argv[1]).read()); print(sys.argv[2], sys.argv[3], d["value"], d["ms_per_step"], d["stages_ms_per_launch"]["wta"])' $O/x.json $name $v; }
for rep in 1 2 3; do for v in prev exp; do
run $v ref --disparities 256 --paths 4
run $v c3 --width 1920 --height 1080 --disparities 256 --batch 4
run $v d256p8 --disparities 256 --paths 8
run $v c2_fused --plan fused_up
done; done
