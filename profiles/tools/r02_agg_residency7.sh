R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/resid; mkdir -p $O
timeout -k 10 600 python3 -m pytest $R/tests -q -m gpu 2>&1 | tail -2
for rep in 1 2 3; do for pad in product 157440; do
  if [ $pad = product ]; then unset CART_AGG_DYNLDS CART_ENGINE_LIB; else export CART_AGG_DYNLDS=$pad CART_ENGINE_LIB=$R/cart-slam_amd/build/ab/exp/libcart_engine.so; fi
  timeout -k 10 200 python3 $R/bench.py --no-cpu-baseline --no-pcie --steps 50 --disparities 64 --paths 4 > $O/x.json 2> $O/x.err || { echo failed $pad; tail -2 $O/x.err; continue; }
  python3 -c 'import json,sys; d=json.loads(open(sys.argv[1]).read()); print("D64P4", sys.argv[2], d["value"], d["ms_per_step"], d["stages_ms_per_launch"]["aggregate"])' $O/x.json $pad
done; done
unset CART_AGG_DYNLDS CART_ENGINE_LIB
for a in "--disparities 64 --paths 4 --deferred" "--disparities 64 --paths 4 --batch 32" "--disparities 64 --paths 4 --batch 8"; do
  timeout -k 10 200 python3 $R/bench.py --no-cpu-baseline --no-pcie --steps 50 $a 2>/dev/null | python3 -c 'import json,sys; d=json.loads(sys.stdin.read()); print(sys.argv[1], d["value"], d["ms_per_step"], d["stages_ms_per_launch"]["aggregate"])' "$a"
done
