R=$GRAFT_REPO_ROOT
echo "cpu.max: $(cat /sys/fs/cgroup/cpu.max 2>&1)"; echo "cfs: $(cat /sys/fs/cgroup/cpu/cpu.cfs_quota_us 2>&1) / $(cat /sys/fs/cgroup/cpu/cpu.cfs_period_us 2>&1)"; echo "nproc: $(nproc)  cpuset: $(cat /sys/fs/cgroup/cpuset.cpus.effective 2>/dev/null | cut -c1-80)"
cd $R && python3 - <<'PY'
import os, sys, json, subprocess
sys.path.insert(0, ".")
import bench
for n, bind in ((16, "false"), (32, "false"), (64, "false"), (128, "false"), (256, "false"), (16, "close"), (64, "close")):
    env = dict(os.environ, OMP_NUM_THREADS=str(n), OMP_DYNAMIC="false", OMP_PROC_BIND=bind)
    if bind == "close": env["OMP_PLACES"] = "cores"
    r = subprocess.run([sys.executable, "-c", bench.CPU_BASELINE_CHILD, bench.ROOT, "1242", "375", "128", "8", "4"], env=env, capture_output=True, text=True)
    print(n, bind, r.stdout.strip()[-200:] or r.stderr[-300:], flush=True)
print(bench.cpu_baseline(1242, 375, 128, 8, 4.0))
PY
timeout -k 10 600 python3 $R/bench.py --no-cpu-baseline > $R/gpurun_out/pcie2.json 2> $R/gpurun_out/pcie2.err; python3 -c 'import json,sys; d=json.loads(open(sys.argv[1]).read()); print("narrow+prefetch", d["value"], d["ms_per_step"], d.get("value_pcie_inclusive"))' $R/gpurun_out/pcie2.json
timeout -k 10 600 python3 $R/bench.py --no-cpu-baseline --pcie-copy blit > $R/gpurun_out/pcie3.json 2> $R/gpurun_out/pcie3.err; python3 -c 'import json,sys; d=json.loads(open(sys.argv[1]).read()); print("blit+prefetch", d["value"], d["ms_per_step"], d.get("value_pcie_inclusive"))' $R/gpurun_out/pcie3.json
