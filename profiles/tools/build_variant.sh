#!/bin/bash
# usage: build_variant.sh <name> <extra hipcc flags...>   ->  cart-slam_amd/build/ab/<name>/libcart_engine.so
# A/B builds of the engine with development knobs (-D...); select one at run time with CART_ENGINE_LIB=<path> (cartslam/_lib.py).
# Only sgm_kernels.hip is rebuilt, the other objects come from build/.
set -e
ROOT=$(cd "$(dirname "$0")/../.." && pwd); PKG=$ROOT/cart-slam_amd; name=$1; shift
OUT=$PKG/build/ab/$name; mkdir -p $OUT
/opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -ffp-contract=off -Wall -Wno-unused-result -I$ROOT/include -I$PKG/csrc "$@" -c $PKG/csrc/sgm_kernels.hip -o $OUT/sgm_kernels.o
/opt/rocm/bin/hipcc -shared -fPIC --offload-arch=gfx950 -o $OUT/libcart_engine.so $OUT/sgm_kernels.o $PKG/build/cart_engine.o $PKG/build/post_kernels.o $PKG/build/flow_kernels.o $PKG/build/superpixel_kernels.o
rm -f $OUT/sgm_kernels.o
echo built $OUT/libcart_engine.so
