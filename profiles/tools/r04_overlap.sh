cd $GRAFT_REPO_ROOT; O=gpurun_out/r04_ovl; mkdir -p $O
for c in "c1 --disparities 64 --paths 4" "c2 " "ref --disparities 256 --paths 4"; do set -- $c; n=$1; shift
 for m in overlap no-overlap; do for r in 1 2; do
  timeout -k 10 200 python3 bench.py --no-cpu-baseline --no-pcie --no-bgr --steps 20 --repeats 3 --$m "$@" > $O/${n}_${m}_$r.json 2>$O/err.txt || { tail -3 $O/err.txt; exit 1; }
  python3 -c 'import json,sys; d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1]); s=d["stages_ms_per_launch"]; print(sys.argv[2], sys.argv[3], d["value"], d["ms_per_step"], {k:round(v,4) for k,v in s.items()}, round(sum(s.values()),4))' $O/${n}_${m}_$r.json $n $m
 done; done; done
