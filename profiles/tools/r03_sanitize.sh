#!/bin/bash
# Sanitizer runs of the GPU-less code (dev container, NEVER on a GPU box): builds with `make -C cart-slam_amd sanitize`, then
#   * host runtime (worker pool, coalescer, frame order, System, retention ring) under ThreadSanitizer and under ASan + UBSan,
#   * the PNG and JSON readers over mutated / random inputs under ASan + UBSan,
#   * the oracle's tests under ASan + UBSan, and its OpenMP loops under ThreadSanitizer (clang + libomp + archer).
# Logs go to profiles/${ROUND:-r03}_sanitizers/ (ROUND=r04 for this round's host code).
set -u
R=$(cd "$(dirname "$0")/../.." && pwd); O=$R/profiles/${ROUND:-r03}_sanitizers; T=$(mktemp -d); mkdir -p $O
make -C $R/cart-slam_amd sanitize > $T/build.log 2>&1 || { tail -20 $T/build.log; exit 1; }
S=$R/cart-slam_amd/build/san
run() { name=$1; shift; ( "$@" > $T/$name.log 2>&1; echo "exit code $?" >> $T/$name.log ); tail -n 3 $T/$name.log | sed "s/^/[$name] /"; grep -E "^(SUMMARY|WARNING: ThreadSanitizer|ERROR: AddressSanitizer|.*runtime error:)" $T/$name.log | sort | uniq -c > $O/$name.txt; tail -n 4 $T/$name.log >> $O/$name.txt; }
run runtime_test_tsan timeout 900 $S/runtime_test_tsan
run runtime_test_asan_ubsan env ASAN_OPTIONS=detect_leaks=1 timeout 900 $S/runtime_test_asan
python3 $R/tools/ref_pin/export_inputs.py $T/in > /dev/null && cp $T/in/road_200x120_d128_p8_bgr_left.png $T/seed.png
cat > $T/modules.json <<'JSON'
{"modules": [{"type": "disparity", "min_disparity": 4, "num_disparities": 128, "smoothing_radius": 2, "smoothing_iterations": 1},
             {"type": "disparity_planeseg", "parameter_provider": {"type": "histogram_peak"}, "update_interval": 30, "reset_interval": 10,
              "use_temporal_smoothing": false}]}
JSON
run fuzz_readers_asan_ubsan timeout 1800 $S/fuzz_readers_asan $T/seed.png 600 $T/modules.json
ASAN=$(gcc -print-file-name=libasan.so)
run oracle_tests_asan_ubsan env CART_ORACLE_LIB=$R/oracle/_build/libcart_oracle_asan.so LD_PRELOAD=$ASAN ASAN_OPTIONS=detect_leaks=0 timeout 1800 python3 -m pytest $R/tests/test_oracle.py -x -q -p no:cacheprovider
TSAN=/opt/rocm/lib/llvm/lib/clang/$(ls /opt/rocm/lib/llvm/lib/clang | head -1)/lib/linux/libclang_rt.tsan-x86_64.so
run oracle_tests_tsan env CART_ORACLE_LIB=$R/oracle/_build/libcart_oracle_tsan.so LD_PRELOAD=$TSAN OMP_TOOL_LIBRARIES=/opt/rocm/lib/llvm/lib/libarcher.so TSAN_OPTIONS="ignore_noninstrumented_modules=1 halt_on_error=0" OMP_NUM_THREADS=4 timeout 2400 python3 -m pytest $R/tests/test_oracle.py -x -q -p no:cacheprovider
rm -rf $T
