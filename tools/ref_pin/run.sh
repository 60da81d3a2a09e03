#!/bin/bash
# One command from an OpenCV-CUDA install to a pinned oracle:   tools/ref_pin/run.sh [<OpenCV prefix, e.g. /usr/local>]
#   1. exports the golden + full-size inputs as PNG (numpy + zlib only),
#   2. builds ref_pin.cpp against that OpenCV (needs cudastereo + cudaimgproc: an OpenCV built WITH_CUDA + opencv_contrib -- what the reference
#      itself links, CMakeLists.txt:19; README.md here has a build recipe),
#   3. runs it on the machine's NVIDIA GPU: tests/golden/ref/ref_disparity_<case>.bin + OPENCV_VERSION.txt,
#   4. compares them with the CPU oracle under all eight settings of the three open choices and prints what to do (verdict.py):
#      nothing, or the cart_engine_set_option defaults to flip, or that the difference is none of the three.
# Nothing of the reference repository is read, built or copied: ref_pin.cpp makes the reference's three OpenCV calls on our inputs.
set -euo pipefail
HERE=$(cd "$(dirname "$0")" && pwd); ROOT=$(cd "$HERE/../.." && pwd); PREFIX=${1:-}
if [ -n "$PREFIX" ]; then export PKG_CONFIG_PATH="$PREFIX/lib/pkgconfig:$PREFIX/lib64/pkgconfig:${PKG_CONFIG_PATH:-}"; export LD_LIBRARY_PATH="$PREFIX/lib:$PREFIX/lib64:${LD_LIBRARY_PATH:-}"; fi
command -v pkg-config > /dev/null || { echo "run.sh: pkg-config not found (it locates the OpenCV the reference itself links)" >&2; exit 1; }
if ! pkg-config --exists opencv4; then echo "run.sh: no opencv4.pc found (give the install prefix of an OpenCV built with -DOPENCV_GENERATE_PKGCONFIG=ON)" >&2; exit 1; fi
if ! pkg-config --libs opencv4 | tr ' ' '\n' | grep -q cudastereo; then echo "run.sh: this OpenCV ($(pkg-config --modversion opencv4)) has no cudastereo module: it needs WITH_CUDA + opencv_contrib (README.md)" >&2; exit 1; fi
python3 "$HERE/export_inputs.py" "$HERE/inputs"
BIN=$(mktemp -d)/ref_pin
g++ -O2 "$HERE/ref_pin.cpp" -o "$BIN" $(pkg-config --cflags --libs opencv4)
mkdir -p "$ROOT/tests/golden/ref"
"$BIN" "$HERE/inputs" "$ROOT/tests/golden/ref"
rc=0; python3 "$HERE/verdict.py" "$ROOT/tests/golden/ref" || rc=$?
echo "next: python -m pytest tests/test_ref_pin.py -q   (add -m gpu on an MI355X for the engine)   and   git add tests/golden/ref"
exit $rc
