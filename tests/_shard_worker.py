"""One RANK of the native row-sharded driver as its own PROCESS (started by tests/test_large_d.py; not a test module).
    python tests/_shard_worker.py RANK WORLD PORT D NP METHOD OUT_DIR [fail]
Several ranks share GPU 0, so RCCL is not available; the vgpa_comm table is HostStagedComm (gloo, staged through the host)."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))


def main():
    rank, world, port, d, n = (int(v) for v in sys.argv[1:6])
    method, out_dir = sys.argv[6], sys.argv[7]
    fail = len(sys.argv) > 8 and sys.argv[8] == "fail"
    os.environ.update(RANK=str(rank), WORLD_SIZE=str(world), MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    import numpy as np
    import torch
    import torch.distributed as dist
    from vgpa_amd.large_d import NativeShardedRecursion, HostStagedComm
    from vgpa_amd._lib import SHARD_OPT_TIMEOUT_MS
    from oracle import vgpa_oracle as vo
    from test_gpu_edge_cases import make_problem
    import datetime
    torch.cuda.set_device(0)
    # (a short collective time-out: the dead-peer case must end in seconds, not in gloo's default half hour)
    dist.init_process_group(backend="gloo", rank=rank, world_size=world, timeout=datetime.timedelta(seconds=30))
    p, x = make_problem("L96", d, n, method=method)
    a_h, b_h = x[:n * d * d].reshape(n, d, d), x[n * d * d:].reshape(n, d)
    comm = HostStagedComm()
    rec = NativeShardedRecursion(method, p.dt, d, n, rank=rank, world=world, device=0, comm=comm.table)
    rec.set_option(SHARD_OPT_TIMEOUT_MS, 20000)
    lo, hi = rec.time_slice
    if fail and rank == world - 1:
        # this rank dies before the sweep: its peers must come back with an error (time-out -> abort), not hang
        os._exit(3)
    try:
        f, ga, gb = rec.sweep_sharded(a_h[lo:hi], b_h[lo:hi], p.theta, np.diag(p.sigma), p.m0, p.s0, p.obs_t, p.obs_y,
                                      np.diag(p.obs_noise), float(np.asarray(vo.kl0(p))))
    except RuntimeError as exc:
        np.savez(os.path.join(out_dir, f"r{rank}.npz"), error=str(exc))
        os._exit(0)
    np.savez(os.path.join(out_dir, f"r{rank}.npz"), f=f, ga=ga.cpu().numpy(), gb=gb.cpu().numpy(), lo=lo, hi=hi,
             chunks=rec.gather_chunks)
    rec.close()
    dist.barrier()
    dist.destroy_process_group()


if __name__ == "__main__":
    main()
