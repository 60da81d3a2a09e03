R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/ovl1; mkdir -p $O
timeout -k 10 600 python3 $R/profiles/tools/stream_overlap.py > $O/base.txt 2>&1; cat $O/base.txt
echo "--- nopoll build (upper bound without hand-over waits; results wrong)"
PLANS=pairs MODES="1x16 2x8 3x8" CART_ENGINE_LIB=$R/cart-slam_amd/build/ab/nopoll/libcart_engine.so timeout -k 10 300 python3 $R/profiles/tools/stream_overlap.py > $O/nopoll.txt 2>&1; cat $O/nopoll.txt
timeout -k 10 600 python3 -m pytest $R/tests/test_gpu_sequence.py -m gpu -q > $O/pytest.log 2>&1; tail -3 $O/pytest.log
