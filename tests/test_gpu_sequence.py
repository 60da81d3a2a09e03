"""GPU tests (-m gpu) of the batched-sequence configuration (BASELINE.json configs[4]) and of the frame-sharded driver
with the real engine in every rank.

* 64 frames, 1242x375, D=128, 8 paths, through StereoPipeline in 4 batches of 16 with the reference's update interval
  (30), so that the plane-parameter refresh at ids 1, 31, 61 (src/modules/planeseg/planeseg.cu:381-395) is crossed at
  full size; disparity, planes, component ids / count / table of a subset of the frames against the oracle.
* two ranks (gloo, both on cuda:0, fresh child processes) against a single-process run of the same frames."""
import os
import socket
import subprocess
import sys

import numpy as np
import pytest

import oracle_lib as O
from cartslam import synth

pytestmark = pytest.mark.gpu
HERE = os.path.dirname(os.path.abspath(__file__))


def test_sequence_64_frames_full_size():
    import torch
    from cartslam import Engine
    from cartslam.pipeline import StereoPipeline
    w, h, D, P, n, B = 1242, 375, 128, 8, 64, 16
    check = [1, 2, 30, 31, 32, 61, 64]
    eng = Engine(w, h, num_disparities=D, paths=P, min_disparity=4, smoothing_radius=2, smoothing_iterations=1, max_inflight=2 * B)
    assert eng.describe_plan(B)["frames_per_launch"] == B
    pipe = StereoPipeline(eng, provider="histogram_peak", with_ccl=True, overlap=True, keep_hists=True)   # the bench's configuration
    outs, frames = [], {}
    for b0 in range(0, n, B):
        # from the second batch on the scene's ground plane is flatter (disparities up to 42 instead of 76): its
        # derivative peak moves, so the refreshes at ids 31 and 61 really change the parameters
        ls, rs = synth.make_batch(B, w, h, D if b0 == 0 else 40, 4, first_frame=b0)
        for fid in check:
            if b0 < fid <= b0 + B:
                frames[fid] = (ls[fid - 1 - b0], rs[fid - 1 - b0])
        got = pipe.process_sequence(torch.from_numpy(ls).cuda(), torch.from_numpy(rs).cuda(), B,
                                    keys=("disparity", "planes", "ids", "n_components", "components", "hists", "params"))
        outs.append(got)
    torch.cuda.synchronize()
    cat = lambda k: np.concatenate([o[k].cpu().numpy() for o in outs])
    disp, planes, ids, ncomp, comps, hists, params = (cat(k) for k in ("disparity", "planes", "ids", "n_components", "components", "hists", "params"))
    assert disp.shape == (n, h, w) and hists.shape == (n, 256)
    # the oracle's disparity / derivative / histogram of the checked frames ...
    exp = {}
    for fid in check:
        l, r = frames[fid]
        ed = O.disparity_module(l, r, D, P, 4, radius=2, iterations=1)
        assert (disp[fid - 1] == ed).all(), f"disparity of frame {fid}: {int((disp[fid - 1] != ed).sum())} pixels differ"
        dd, eh = O.plane_derivative(ed)
        assert (hists[fid - 1] == eh).all(), f"histogram of frame {fid}"
        exp[fid] = dd
    # ... the reference's bookkeeping over ALL 64 per-frame histograms (the GPU's, equal to the oracle's on the checked
    # frames and covered at this size by test_full_size_against_oracle) with the oracle's peak finder ...
    cum = np.zeros(256, np.int64)
    p = (0, 0, 0, 0, 0, 0)
    refreshed = []
    for fid in range(1, n + 1):
        cum += hists[fid - 1]
        if fid % 30 == 1:   # planeseg.cu:381
            h32 = cum.astype(np.int32)
            if fid % 300 == 1:   # :391-394
                cum[:] = 0
            ok, p = O.histogram_peak_params(h32, p)
            refreshed.append((fid, ok, p))
        assert tuple(int(v) for v in params[fid - 1]) == tuple(p), f"plane parameters of frame {fid}"
        # ... and classification, components and the component table of the checked frames
        if fid in exp:
            ep = O.classify(exp[fid], p)
            assert (planes[fid - 1] == ep).all(), f"planes of frame {fid}"
            eids, en = O.ccl(ep)
            assert (ids[fid - 1] == eids).all() and int(ncomp[fid - 1]) == en, f"components of frame {fid}"
            et, _ = O.ccl_stats(ep, eids, max_components=4096)
            m = min(len(et), 4096)
            assert (comps[fid - 1][:m] == et[:m]).all(), f"component table of frame {fid}"
    assert [f for f, _, _ in refreshed] == [1, 31, 61]
    assert any(ok for _, ok, _ in refreshed), "the schedule never produced parameters"
    assert len({tuple(int(v) for v in row) for row in params}) >= 2, "the refresh at id 31 / 61 changed nothing: the test would not see a wrong schedule"
    eng.close()


def _free_port():
    s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
    return port


@pytest.mark.parametrize("device_schedule", [1, 0])
def test_two_ranks_real_engine_equal_single_process(tmp_path, device_schedule):
    """tests/dist_worker.py x 2 (gloo, one GPU) vs one process: every frame of every step, of the sequence call and of four
    pipelined sequences (one of uneven length: rank 0 holds one frame more; one of a single frame: rank 1 holds none)."""
    import torch
    from cartslam import Engine
    from cartslam.pipeline import StereoPipeline
    from dist_worker import frames_of
    world, w, h, D, P, n_local, steps, ui, ri = 2, 256, 96, 64, 8, 3, 3, 4, 2   # 18 + 6 frames: refreshes at ids 1, 5, 9, 13, 17, 21; resets at 9, 17
    port = _free_port()
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0")
    procs = []
    for r in range(world):
        cmd = [sys.executable, os.path.join(HERE, "dist_worker.py"), str(r), str(world), str(port), str(tmp_path / f"rank{r}.npz")] + \
              [str(v) for v in (w, h, D, P, n_local, steps, ui, ri, device_schedule)]
        procs.append(subprocess.Popen(cmd, env=env, stdout=subprocess.PIPE, stderr=subprocess.STDOUT))
    for r, pr in enumerate(procs):
        out, _ = pr.communicate(timeout=540)
        assert pr.returncode == 0, f"rank {r} failed:\n{out.decode(errors='replace')[-3000:]}"
    ranks = [np.load(tmp_path / f"rank{r}.npz") for r in range(world)]
    # single process, same frames in id order, same batch boundaries in ids
    eng = Engine(w, h, num_disparities=D, paths=P, min_disparity=4, smoothing_radius=2, smoothing_iterations=1, max_inflight=2 * n_local * world + 2)
    pipe = StereoPipeline(eng, provider="histogram_peak", update_interval=ui, reset_interval=ri, with_ccl=True, device_schedule=bool(device_schedule))
    n_step = n_local * world
    # the batch boundaries of the ranks' run: `steps` batches, the sequence call, then the three pipelined sequences
    sizes = [n_step] * (steps + 1) + [n_step, n_step + 1, 1, n_step]
    total = sum(sizes)
    seen, first = 0, 1
    for s, size in enumerate(sizes):
        ids = list(range(first, first + size))
        first += size
        ls, rs = frames_of(ids, w, h, D, 4321)
        o = pipe.process_batch(torch.from_numpy(ls).cuda(), torch.from_numpy(rs).cuda())
        torch.cuda.synchronize()
        for k, fid in enumerate(ids):
            d1, p1 = o["disparity"][k].cpu().numpy(), o["planes"][k].cpu().numpy()
            if s < steps:
                z = ranks[(fid - 1) % world]   # frame f -> rank (f - first) mod world
                assert (z[f"disp_{fid}"] == d1).all(), f"disparity frame {fid}"
                assert (z[f"planes_{fid}"] == p1).all(), f"planes frame {fid}"
                assert (z[f"ids_{fid}"] == o["ids"][k].cpu().numpy()).all() and int(z[f"ncomp_{fid}"]) == int(o["n_components"][k].item()), f"ccl frame {fid}"
            elif s == steps:
                z = ranks[0]
                assert int(z["seq_first_id"]) == ids[0]
                assert (z["seq_disp"][k] == d1).all(), f"sequence disparity frame {fid}"
                assert (z["seq_planes"][k] == p1).all(), f"sequence planes frame {fid}"
            else:
                z, q = ranks[0], s - steps - 1
                assert z[f"pseq{q}_disp"].shape[0] == size
                assert (z[f"pseq{q}_disp"][k] == d1).all(), f"pipelined sequence {q} disparity frame {fid}"
                assert (z[f"pseq{q}_planes"][k] == p1).all(), f"pipelined sequence {q} planes frame {fid}"
                assert (z[f"pseq{q}_ids"][k] == o["ids"][k].cpu().numpy()).all() and int(z[f"pseq{q}_ncomp"][k]) == int(o["n_components"][k].item()), f"pipelined sequence {q} components frame {fid}"
            seen += 1
    assert seen == total
    assert "seq_disp" not in ranks[1].files
    eng.close()


def test_rccl_collectives_single_rank(tmp_path):
    """The collectives of the sharded path on the REAL backend: one RCCL rank (all a one-GPU lease can host) pushes the
    histogram all-gather -- issued on the pipeline's side stream like in bench.py -- and the sequence scatter / gather
    through `nccl`, and the results must equal a run without torch.distributed."""
    import torch
    from cartslam import Engine
    from cartslam.pipeline import StereoPipeline
    from dist_worker import frames_of
    out = tmp_path / "rccl.npz"
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0")
    pr = subprocess.run([sys.executable, os.path.join(HERE, "rccl_single_rank_worker.py"), str(_free_port()), str(out)], env=env,
                        stdout=subprocess.PIPE, stderr=subprocess.STDOUT, timeout=540)
    assert pr.returncode == 0, pr.stdout.decode(errors="replace")[-3000:]
    z = np.load(out)
    w, h, D, P, n, ui, ri = 256, 96, 64, 8, 6, 4, 2
    eng = Engine(w, h, num_disparities=D, paths=P, min_disparity=4, smoothing_radius=2, smoothing_iterations=1, max_inflight=2 * n)
    for dev_sched in (1, 0):
        pipe = StereoPipeline(eng, provider="histogram_peak", update_interval=ui, reset_interval=ri, with_ccl=True, device_schedule=bool(dev_sched))
        for s in range(6):   # two batches, one sequence, three pipelined sequences in the worker
            ls, rs = frames_of(range(s * n + 1, (s + 1) * n + 1), w, h, D, 4321)
            o = pipe.process_batch(torch.from_numpy(ls).cuda(), torch.from_numpy(rs).cuda())
            torch.cuda.synchronize()
            assert (z[f"disp_{dev_sched}_{s}"] == o["disparity"].cpu().numpy()).all(), (dev_sched, s)
            assert (z[f"planes_{dev_sched}_{s}"] == o["planes"].cpu().numpy()).all(), (dev_sched, s)
    eng.close()


def test_native_rccl_sharder_single_gpu(tmp_path):
    """cart_shard_amd (host/src/sharder.cpp: the batched-sequence mode below Python -- ncclCommInitAll, grouped ncclSend /
    ncclRecv scatter and gather, ncclAllGather of the histograms, device-side schedule replay) with the one GPU a lease has:
    every collective runs through RCCL with a world of one, and disparity + planes must equal the Python pipeline's."""
    import torch
    from cartslam import Engine
    from cartslam.pipeline import StereoPipeline
    from dist_worker import frames_of
    exe = os.path.join(os.path.dirname(HERE), "cart-slam_amd", "build", "cart_shard_amd")
    assert os.path.exists(exe), "cart_shard_amd not built (make -C cart-slam_amd)"
    w, h, D, P, n, per_call, ui, ri = 256, 96, 64, 8, 12, 4, 5, 2   # refreshes at ids 1, 6, 11; reset at 11
    ls, rs = frames_of(range(1, n + 1), w, h, D, 987)
    ls.tofile(tmp_path / "left.bin"); rs.tofile(tmp_path / "right.bin")
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0")
    pr = subprocess.run([exe, str(tmp_path / "left.bin"), str(tmp_path / "right.bin"), str(w), str(h), str(n), str(D), str(P), "1", str(per_call),
                         str(tmp_path), str(ui), str(ri)], env=env, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, timeout=540)
    assert pr.returncode == 0, pr.stdout.decode(errors="replace")[-3000:]
    assert b"pairs_per_s" in pr.stdout
    disp = np.fromfile(tmp_path / "disparity.bin", np.int16).reshape(n, h, w)
    planes = np.fromfile(tmp_path / "planes.bin", np.uint8).reshape(n, h, w)
    eng = Engine(w, h, num_disparities=D, paths=P, min_disparity=4, smoothing_radius=2, smoothing_iterations=1, max_inflight=per_call)
    pipe = StereoPipeline(eng, provider="histogram_peak", update_interval=ui, reset_interval=ri, with_ccl=False)
    for f0 in range(0, n, per_call):
        o = pipe.process_batch(torch.from_numpy(ls[f0:f0 + per_call]).cuda(), torch.from_numpy(rs[f0:f0 + per_call]).cuda())
        torch.cuda.synchronize()
        assert (disp[f0:f0 + per_call] == o["disparity"].cpu().numpy()).all(), f"disparity of frames {f0 + 1}.."
        assert (planes[f0:f0 + per_call] == o["planes"].cpu().numpy()).all(), f"planes of frames {f0 + 1}.."
    assert len(np.unique(planes)) == 3
    eng.close()
    # a box with one GPU must refuse two
    pr = subprocess.run([exe, str(tmp_path / "left.bin"), str(tmp_path / "right.bin"), str(w), str(h), str(n), str(D), str(P), "2", "4", str(tmp_path)],
                        env=env, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, timeout=120)
    if torch.cuda.device_count() < 2:
        assert pr.returncode != 0 and b"GPUs" in pr.stdout
