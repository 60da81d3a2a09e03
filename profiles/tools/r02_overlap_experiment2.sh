R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/ovl2; mkdir -p $O
for v in base w6 w8 nopoll; do
  lib=$R/cart-slam_amd/build/ab/$v/libcart_engine.so; [ "$v" = base ] && lib=$R/cart-slam_amd/build/libcart_engine.so
  echo "--- $v"; PLANS=pairs MODES="1x16 2x8 1x32" CART_ENGINE_LIB=$lib timeout -k 10 300 python3 $R/profiles/tools/stream_overlap.py 2>&1 | grep -v amdgpu.ids | tee $O/$v.txt
done
PLANS=slabs MODES="1x16" timeout -k 10 300 python3 $R/profiles/tools/stream_overlap.py 2>&1 | grep -v amdgpu.ids | tee $O/slabs.txt
timeout -k 10 600 python3 -m pytest $R/tests/test_gpu_parity.py -m gpu -q -k "pairs or plans or randomized" > $O/pytest.log 2>&1; tail -3 $O/pytest.log
