"""pytest configuration: registers the `gpu` marker and puts the package + test helpers on sys.path."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (os.path.join(ROOT, "cart-slam_amd"), os.path.join(ROOT, "tests"), ROOT):
    if p not in sys.path:
        sys.path.insert(0, p)


def _cpu_share():
    """CPUs this container may actually run: the cgroup quota when there is one (a GPU box shows 256 CPUs and grants 16)."""
    try:
        visible = len(os.sched_getaffinity(0))
    except Exception:
        visible = os.cpu_count() or 1
    for path in ("/sys/fs/cgroup/cpu.max", "/sys/fs/cgroup/cpu/cpu.cfs_quota_us"):
        try:
            txt = open(path).read().split()
            quota = float(txt[0]) if txt[0] != "max" else -1.0
            period = float(txt[1]) if len(txt) > 1 else float(open("/sys/fs/cgroup/cpu/cpu.cfs_period_us").read())
            if quota > 0:
                return max(1, min(visible, int(-(-quota // period))))
            break
        except Exception:
            continue
    return visible


# The OpenMP oracle (the checker) would start one thread per VISIBLE CPU: on a 16-CPU share of a 256-CPU host that made
# the oracle comparisons 4-30x slower (a host-loop test took 15-20 s instead of 1-2).  Has to be set before libgomp loads.
os.environ.setdefault("OMP_NUM_THREADS", str(_cpu_share()))
os.environ.setdefault("OMP_DYNAMIC", "false")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")
