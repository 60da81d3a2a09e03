# configs[1] (D=64, 4 paths): batch size, frames per launch sequence, split stages
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/c1b; mkdir -p $O
run() { name="$*"; timeout -k 10 300 python3 $R/bench.py --no-cpu-baseline --no-pcie --steps 40 --disparities 64 --paths 4 "$@" > $O/x.json 2> $O/x.err || { echo "$name failed"; tail -3 $O/x.err; return; }
  python3 -c 'import json,sys; d=json.loads(open(sys.argv[1]).read()); print(sys.argv[2], "|", d["value"], d["ms_per_step"], d["config"]["launch_plan"]["frames_per_launch"], d["stages_ms_per_launch"])' $O/x.json "$name"; }
for rep in 1 2; do
run --batch 16
run --batch 16 --split
run --batch 32
run --batch 32 --chunk 32
run --batch 32 --chunk 32 --split
run --batch 64 --chunk 32
run --batch 64 --chunk 64
run --batch 24 --chunk 24
done
