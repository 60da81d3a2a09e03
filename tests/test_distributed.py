"""World-size-2 gloo tests (CPU) of the frame-sharded path: id interleaving, the histogram all-gather and the
plane-parameter schedule must give every rank exactly what one process gets feeding the frames in id order.
The per-frame histograms stand in for the GPU's (they come from the oracle here; the GPU parity of the histogram
kernel itself is covered in test_gpu_parity.py)."""
import os
import socket
import sys

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

import oracle_lib as O
from cartslam.pipeline import CollectiveError, PlaneParameterSchedule, SequencePipeliner, gather_sequence, scatter_sequence, share_of, shard_ids

HERE = os.path.dirname(os.path.abspath(__file__))


def make_hists(n, seed=3):
    """n synthetic per-frame derivative histograms whose peaks drift so that parameters really change."""
    rng = np.random.default_rng(seed)
    x = np.arange(256)
    out = []
    for f in range(n):
        h = 6000 * np.exp(-0.5 * ((x - 128) / 1.5) ** 2) + 2500 * np.exp(-0.5 * ((x - (136 + f // 20)) / 2.5) ** 2)
        out.append((h + rng.integers(0, 20, 256)).astype(np.int32))
    return np.stack(out)


def sequential_reference(hists, update_interval, reset_interval):
    """The reference's bookkeeping, one frame at a time (planeseg.cu:271-283, :381-395) with the oracle's peak finder."""
    cum = np.zeros(256, np.int64)
    params = (0, 0, 0, 0, 0, 0)
    out = []
    for k, h in enumerate(hists):
        fid = k + 1
        cum += h
        if fid % update_interval == 1:
            h32 = cum.astype(np.int32)
            if fid % (update_interval * reset_interval) == 1:
                cum[:] = 0
            _, params = O.histogram_peak_params(h32, params)
        out.append(params)
    return out


def test_schedule_matches_sequential_reference():
    hists = make_hists(95)
    for ui, ri in ((30, 10), (7, 2), (4, 3)):
        exp = sequential_reference(hists, ui, ri)
        sch = PlaneParameterSchedule("histogram_peak", update_interval=ui, reset_interval=ri)
        got = []
        for a in range(0, 95, 16):  # arbitrary batch boundaries must not matter
            got += [p.as_tuple() for p in sch.advance(a + 1, hists[a:a + 16])]
        assert got == exp


def test_static_provider_and_unknown_type():
    sch = PlaneParameterSchedule("static", static_params=(6, 18, -5, 6, 12, 0))
    assert [p.as_tuple() for p in sch.advance(1, make_hists(3))] == [(6, 18, -5, 6, 12, 0)] * 3
    with pytest.raises(ValueError):
        PlaneParameterSchedule("nope")


def test_shard_ids_partition():
    world, n_local, first = 4, 5, 17
    ids = sorted(i for r in range(world) for i in shard_ids(first, n_local, r, world))
    assert ids == list(range(first, first + world * n_local))


def _worker(rank, world, port, n_local, steps, ui, ri, q):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        hists = make_hists(world * n_local * steps)
        sch = PlaneParameterSchedule("histogram_peak", update_interval=ui, reset_interval=ri)
        mine = []
        next_id = 1
        for s in range(steps):
            ids = shard_ids(next_id, n_local, rank, world)
            local = torch.from_numpy(np.stack([hists[i - 1] for i in ids]))
            allh = torch.empty((world * n_local, 256), dtype=torch.int32)
            dist.all_gather_into_tensor(allh, local)  # the path's only exchange step
            ordered = allh.view(world, n_local, 256).permute(1, 0, 2).reshape(n_local * world, 256).numpy()
            allp = sch.advance(next_id, ordered)
            mine += [(i, p.as_tuple()) for i, p in zip(ids, allp[rank::world])]
            next_id += n_local * world
        q.put((rank, mine))
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("world,n_local,steps,ui,ri", [(2, 5, 4, 7, 2), (8, 8, 2, 30, 10)])
def test_world_size_2_gloo_equals_single_process(world, n_local, steps, ui, ri):
    """Two ranks, and BASELINE configs[4]'s own partition: EIGHT ranks x 8 frames = a 64-frame step (two of them: ids 1..128 with the
    reference's update interval 30, refreshes at 1 / 31 / 61 / 91 / 121) -- every rank gets, for its frames, the parameters one
    process gets feeding the frames in id order."""
    s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_worker, args=(r, world, port, n_local, steps, ui, ri, q)) for r in range(world)]
    for p in procs:
        p.start()
    results = [q.get(timeout=120) for _ in range(world)]
    for p in procs:
        p.join(timeout=120)
        assert p.exitcode == 0
    exp = sequential_reference(make_hists(world * n_local * steps), ui, ri)
    seen = {}
    for _, mine in results:
        for fid, params in mine:
            seen[fid] = params
    assert sorted(seen) == list(range(1, world * n_local * steps + 1))
    for fid, params in seen.items():
        assert params == exp[fid - 1], f"frame {fid}"


def _spawn(target, world, args, timeout=90, expect_exit=None):
    s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=target, args=(r, world, port) + tuple(args) + (q,)) for r in range(world)]
    for p in procs:
        p.start()
    results = []
    for _ in range(world if expect_exit is None else sum(1 for e in expect_exit if e is not None)):
        results.append(q.get(timeout=timeout))
    for r, p in enumerate(procs):
        p.join(timeout=timeout)
        if expect_exit is None:
            assert p.exitcode == 0, f"rank {r} exited with {p.exitcode}"
    return results, [p.exitcode for p in procs]


def _frames(n_total, h=6, w=10, salt=0):
    return ((torch.arange(n_total * h * w, dtype=torch.int32) * 7 + salt) % 251).reshape(n_total, h, w).to(torch.uint8)


def _sequence_worker(rank, world, port, n_total, q):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        h, w = 6, 10
        full = _frames(n_total, h, w)
        seq = full.clone() if rank == 0 else None
        mine = scatter_sequence(seq, n_total, ((h, w), torch.uint8, "cpu"), root=0)
        n_mine = share_of(n_total, rank, world)
        ids = shard_ids(1, n_mine, rank, world)   # frame k (id k + 1) -> rank k mod world
        ok_scatter = mine.shape[0] == n_mine and all(torch.equal(mine[j], full[i - 1]) for j, i in enumerate(ids))
        out = (mine.to(torch.int16) * 3 + rank)   # stands in for this rank's outputs
        back = gather_sequence(out, n_total, root=0)
        if rank == 0:
            want = full.to(torch.int16) * 3 + (torch.arange(n_total) % world).to(torch.int16)[:, None, None]
            ok_gather = torch.equal(back, want)
        else:
            ok_gather = back is None
        bad = None
        try:
            gather_sequence(out[:0] if n_mine else torch.zeros((1, h, w), dtype=torch.int16), n_total, root=0)   # a share of the wrong length
        except ValueError as e:
            bad = str(e)
        q.put((rank, ok_scatter, ok_gather, bad))
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("n_total", [8, 7, 1])
def test_sequence_scatter_gather_world_2(n_total):
    """BASELINE configs[4] plumbing: a sequence on rank 0 is dealt out frame k -> rank k mod world and the outputs come
    back in sequence order.  A length that does not divide gives the first ranks one frame more (7 -> 4 + 3), a sequence
    shorter than the world leaves a rank with nothing to compute (1 -> 1 + 0); a share of the wrong length is refused
    before any collective is posted."""
    results, _ = _spawn(_sequence_worker, 2, (n_total,))
    for rank, ok_scatter, ok_gather, bad in results:
        assert ok_scatter and ok_gather, rank
        assert bad and "frames of a" in bad


def _fake_compute(rank):
    # stands in for StereoPipeline.process_batch: outputs that depend on both inputs, the rank and a per-call counter
    # (a pipeline that handed a later sequence's frames to an earlier handle would be caught by the counter)
    state = {"calls": 0}
    def compute(l, r, n_total):
        state["calls"] += 1
        return {"disparity": l.to(torch.int16) * 2 - r.to(torch.int16) + 100 * state["calls"],
                "planes": ((l.to(torch.int32) + r.to(torch.int32) + rank) % 3).to(torch.uint8)}
    return compute


def _pipeliner_worker(rank, world, port, lengths, q):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        h, w = 6, 10
        seqs = [(_frames(n, h, w, 3 * i), _frames(n, h, w, 3 * i + 1)) for i, n in enumerate(lengths)]
        got = {}
        for mode in ("serial", "pipelined"):
            sp = SequencePipeliner(_fake_compute(rank), (h, w), device="cpu")
            give = lambda i: (seqs[i][0].clone(), seqs[i][1].clone()) if rank == 0 else (None, None)
            if mode == "serial":
                outs = [sp.submit(*give(i), lengths[i]).result() for i in range(len(lengths))]
            else:
                hs = [sp.submit(*give(i), lengths[i]) for i in range(len(lengths))]   # all submitted before any result is asked for
                outs = [hd.result() for hd in hs]
            got[mode] = outs
        ok = True
        if rank == 0:
            for i, n in enumerate(lengths):
                l, r = seqs[i]
                who = (torch.arange(n) % world)[:, None, None]
                want_d = l.to(torch.int16) * 2 - r.to(torch.int16) + 100 * (i + 1)   # every rank has made i + 1 calls, with or without frames
                want_p = ((l.to(torch.int32) + r.to(torch.int32) + who) % 3).to(torch.uint8)
                for mode in got:
                    ok &= torch.equal(got[mode][i]["disparity"], want_d) and torch.equal(got[mode][i]["planes"], want_p)
        else:
            ok = all(o["disparity"] is None and o["planes"] is None for outs in got.values() for o in outs)
        q.put((rank, bool(ok)))
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("world,lengths", [(2, (8, 7, 1, 6)), (4, (13, 3, 16, 5)), (8, (64, 63, 65, 7))])
def test_pipelined_sequences_equal_one_at_a_time(world, lengths):
    """SequencePipeliner over gloo, two, four and EIGHT ranks (the driver's SCALE shape: configs[4]'s 64 frames dealt 8 per rank,
    then 63 and 65 -- a length that does not divide is NOT rejected: frame k -> rank k mod N, the first n mod N ranks hold one frame
    more, short shares travel padded -- and 7, which leaves rank 7 without a frame): four back-to-back sequences (lengths that divide, that leave a
    remainder, and that are shorter than the world, so that some ranks hold no frame at all) give the same gathered outputs
    whether each is waited for before the next is submitted or all are submitted first -- the order in which scatter(i+1) and
    gather(i) are posted is the same on every rank, so no collective can pair with the wrong one."""
    results, _ = _spawn(_pipeliner_worker, world, (lengths,), timeout=180)
    assert sorted(r for r, _ in results) == list(range(world)) and all(ok for _, ok in results)


def _dead_peer_worker(rank, world, port, q):
    import datetime
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world, timeout=datetime.timedelta(seconds=5))
    if rank == 1:
        os._exit(0)   # the peer dies without joining the exchange
    try:
        scatter_sequence(_frames(4), 4, ((6, 10), torch.uint8, "cpu"), root=0)
        q.put((rank, "no error"))
    except CollectiveError as e:
        q.put((rank, str(e)))
        q.close(); q.join_thread()   # the message has left this process before it goes away
        os._exit(3)   # what bench.py does: a non-zero exit that names the rank, no teardown that waits for the dead peer


def test_dead_peer_ends_the_job_with_the_rank_named():
    """A collective whose peer is gone must not hang: the process group's timeout turns it into a CollectiveError that
    names the rank and the call, and the process exits non-zero."""
    results, codes = _spawn(_dead_peer_worker, 2, (), timeout=60, expect_exit=[3, None])
    (rank, msg), = results
    assert rank == 0 and "rank 0" in msg and "scatter(sequence)" in msg, msg
    assert codes[0] == 3
