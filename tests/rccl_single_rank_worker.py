"""One RCCL rank on cuda:0 (tests/test_gpu_sequence.py starts it as a fresh child process): the collectives of the
frame-sharded path -- histogram all_gather_into_tensor on the pipeline's side stream, scatter / gather of the sequence
mode -- executed by the real backend ("nccl" = RCCL), in a world of one rank, which is all a one-GPU lease can host.

  python rccl_single_rank_worker.py <port> <out.npz>
"""
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(os.path.dirname(HERE), "cart-slam_amd"))
sys.path.insert(0, HERE)


def main():
    port, out_path = sys.argv[1], sys.argv[2]
    import torch
    import torch.distributed as dist
    from cartslam import Engine
    from cartslam.pipeline import StereoPipeline
    from dist_worker import frames_of
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=port)
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    torch.cuda.set_device(0)
    dist.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device("cuda", 0))
    try:
        assert dist.get_backend() == "nccl"
        w, h, D, P, n, ui, ri = 256, 96, 64, 8, 6, 4, 2
        eng = Engine(w, h, num_disparities=D, paths=P, min_disparity=4, smoothing_radius=2, smoothing_iterations=1, max_inflight=4 * n)
        res = {}
        for dev_sched in (True, False):
            pipe = StereoPipeline(eng, provider="histogram_peak", update_interval=ui, reset_interval=ri, with_ccl=True,
                                  device_schedule=dev_sched, overlap=dev_sched, always_exchange=True)
            for s in range(2):
                ls, rs = frames_of(range(s * n + 1, (s + 1) * n + 1), w, h, D, 4321)
                o = pipe.process_batch(torch.from_numpy(ls).cuda(), torch.from_numpy(rs).cuda())
                torch.cuda.synchronize()
                res[f"disp_{int(dev_sched)}_{s}"] = o["disparity"].cpu().numpy()
                res[f"planes_{int(dev_sched)}_{s}"] = o["planes"].cpu().numpy()
            ls, rs = frames_of(range(2 * n + 1, 3 * n + 1), w, h, D, 4321)
            got = pipe.process_sequence(torch.from_numpy(ls).cuda(), torch.from_numpy(rs).cuda(), n)   # scatter + gather over RCCL
            torch.cuda.synchronize()
            res[f"disp_{int(dev_sched)}_2"] = got["disparity"].cpu().numpy()
            res[f"planes_{int(dev_sched)}_2"] = got["planes"].cpu().numpy()
            # pipelined: three sequences submitted back to back (scatter of i+1 and gather of i-1 on the copy stream beside
            # the kernels of i, all through RCCL), results asked for afterwards
            handles = []
            for s in range(3, 6):
                ls, rs = frames_of(range(s * n + 1, (s + 1) * n + 1), w, h, D, 4321)
                handles.append(pipe.submit_sequence(torch.from_numpy(ls).cuda(), torch.from_numpy(rs).cuda(), n))
            for s, hd in zip(range(3, 6), handles):
                got = hd.result()
                res[f"disp_{int(dev_sched)}_{s}"] = got["disparity"].cpu().numpy()
                res[f"planes_{int(dev_sched)}_{s}"] = got["planes"].cpu().numpy()
        t = torch.ones(4, device="cuda")
        dist.all_reduce(t)   # what bench.py does with its elapsed time
        dist.barrier()
        assert float(t.sum()) == 4.0
        np.savez(out_path, **res)
        eng.close()
    finally:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
