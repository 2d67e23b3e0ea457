"""CPU suite: the N > 1 plumbing (problem sharding, objective gather, max-over-ranks timing) with gloo, world size 2."""
import os
import sys
import socket

import numpy as np
import pytest

from conftest import ROOT, load_golden
from vgpa_amd.parallel import shard_range


def test_shard_range_partitions_exactly():
    for n in (0, 1, 5, 256, 257, 1001):
        for world in (1, 2, 3, 8):
            covered = []
            for r in range(world):
                lo, hi = shard_range(n, r, world)
                assert 0 <= lo <= hi <= n
                covered.extend(range(lo, hi))
            assert covered == list(range(n))
            sizes = [shard_range(n, r, world)[1] - shard_range(n, r, world)[0] for r in range(world)]
            assert max(sizes) - min(sizes) <= 1
    with pytest.raises(ValueError):
        shard_range(4, 2, 2)


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    return port


def _worker(rank, world, port, n_problems, out_dir):
    os.environ.update(RANK=str(rank), WORLD_SIZE=str(world), MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import torch.distributed as dist
    from vgpa_amd import parallel as par
    from oracle import vgpa_oracle as vo
    r, w = par.init_from_env("gloo")
    assert (r, w) == (rank, world)
    z = load_golden("ou_rk4_p")
    prob = vo.Problem.from_fixture(z)
    lo, hi = par.shard_range(n_problems, rank, world)
    # the objective of this rank's problems (the oracle stands in for the GPU sweep on this CPU-only test)
    f_local = [vo.free_energy(prob, z["x"] + 0.01 * i)[0] for i in range(lo, hi)]
    f_all = par.gather_objective(f_local, n_problems)
    t = par.max_over_ranks(1.0 + rank)
    par.barrier()
    np.save(os.path.join(out_dir, f"f_{rank}.npy"), f_all)
    np.save(os.path.join(out_dir, f"t_{rank}.npy"), np.array([t]))
    dist.destroy_process_group()


def test_two_rank_gloo_gather_and_timing(tmp_path):
    import torch.multiprocessing as mp
    world, n_problems = 2, 5
    mp.spawn(_worker, args=(world, _free_port(), n_problems, str(tmp_path)), nprocs=world, join=True)
    from oracle import vgpa_oracle as vo
    z = load_golden("ou_rk4_p")
    prob = vo.Problem.from_fixture(z)
    serial = np.array([vo.free_energy(prob, z["x"] + 0.01 * i)[0] for i in range(n_problems)])
    for rank in range(world):
        f_all = np.load(tmp_path / f"f_{rank}.npy")
        assert np.array_equal(f_all, serial)              # same problems, same order, on every rank
        assert float(np.load(tmp_path / f"t_{rank}.npy")[0]) == 2.0   # MAX over ranks of (1 + rank)
