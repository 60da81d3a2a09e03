R=$GRAFT_REPO_ROOT; export PLAN=auto
timeout -k 10 600 python3 -m pytest $R/tests/test_gpu_parity.py -m gpu -q -x > $R/gpurun_out/vd_pytest.log 2>&1; tail -2 $R/gpurun_out/vd_pytest.log
echo "==== D=64 P=4"; BENCH_ARGS="--disparities 64 --paths 4" bash $R/profiles/tools/r02_variants.sh vd1 base v2 v6 v4pf4 | sed "s/'census.*'aggregate/ aggregate/"
echo "==== D=128 P=8"; BENCH_ARGS="" bash $R/profiles/tools/r02_variants.sh vd2 base v4all | sed "s/'census.*'aggregate/ aggregate/"
echo "==== D=256 P=4"; BENCH_ARGS="--disparities 256 --paths 4" bash $R/profiles/tools/r02_variants.sh vd3 base v4all | sed "s/'census.*'aggregate/ aggregate/"
echo "==== D=128 P=4"; BENCH_ARGS="--disparities 128 --paths 4" bash $R/profiles/tools/r02_variants.sh vd4 base v4all | sed "s/'census.*'aggregate/ aggregate/"
