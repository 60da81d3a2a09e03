R=$GRAFT_REPO_ROOT
timeout -k 10 600 python3 -m pytest $R/tests/test_host.py $R/tests/test_gpu_sequence.py -q -m gpu 2>&1 | tail -3
N=960 ONLY=0,1,2,3 timeout -k 10 500 python3 $R/profiles/tools/host_loop_throughput.py 2>&1 | grep -v "amdgpu.ids" | tail -24
