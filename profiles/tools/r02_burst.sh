# What the fused sweep's burst (16 buffered rows -> HBM between two block barriers) costs: experiment builds
# fa0 (all on), fa512 (no burst), fa2048 (burst without its barriers), rb32 / rb8 (32 / 8 rows per burst), plan fused_up.
R=$GRAFT_REPO_ROOT; export PLAN=fused_up
for cfg in "" "--disparities 256 --paths 4"; do
  echo "==== bench args: $cfg (plan fused_up; fa512 / fa2048 give wrong results: timing only)"
  BENCH_ARGS="$cfg" bash $R/profiles/tools/r02_variants.sh burst base fa0 fa512 fa2048 rb32 rb8 | sed "s/'census.*'aggregate/ aggregate/"
done
echo "==== frame loop, reference defaults + D=128"
N=960 ONLY=0,1 timeout -k 10 400 python3 $R/profiles/tools/host_loop_throughput.py 2>&1 | tail -12
echo "==== full GPU suite"
timeout -k 10 900 python3 -m pytest $R/tests -q -m gpu 2>&1 | tail -4
