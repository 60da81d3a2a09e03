#!/bin/bash
# baseline sweep at round-2 start
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/sweep1; mkdir -p $O
run() { name=$1; shift; echo "== $name: $*"; env "$@" python3 $R/bench.py --no-cpu-baseline --no-pcie --steps 10 ${ARGS} > $O/$name.json 2> $O/$name.err; python3 -c 'import json,sys; d=json.loads(open(sys.argv[1]).read()); print(d["value"], d["ms_per_step"], d["stages_ms_per_launch"], d["roofline"]["frac"])' $O/$name.json; }
ARGS="" run c2_b16 X=1
ARGS="" run c2_b16_fused CART_FUSED_WTA=1
ARGS="--batch 32" run c2_b32 X=1
ARGS="--batch 32" run c2_b32_fused CART_FUSED_WTA=1
ARGS="--disparities 64 --paths 4" run c1_b16 X=1
ARGS="--disparities 64 --paths 4 --batch 32" run c1_b32 X=1
ARGS="--disparities 64 --paths 4 --batch 64" run c1_b64 X=1
ARGS="--disparities 64 --paths 4 --batch 64" run c1_b64_chunk64 CART_CHUNK_FRAMES=64
ARGS="--disparities 64 --paths 4 --batch 64" run c1_b64_chunk32 CART_CHUNK_FRAMES=32
ARGS="--disparities 64 --paths 4 --batch 32" run c1_b32_fused CART_FUSED_WTA=1
ARGS="--disparities 64 --paths 4 --batch 32" run c1_b32_chunk32 CART_CHUNK_FRAMES=32
ARGS="--disparities 64 --paths 4 --batch 32" run c1_b32_chunk32_fused CART_CHUNK_FRAMES=32 CART_FUSED_WTA=1
