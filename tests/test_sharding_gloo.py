"""The N > 1 path on CPU: two processes, gloo backend.  The sharding + all-gather plumbing of
radiativetransfer.jl_amd/sharding.py is exercised with the C oracle standing in for the GPU compute
(tests may use the oracle); the GPU version of the same property is tests/test_gpu_rt_run.py::
test_sharded_equals_unsharded_bitwise."""
import os
import socket
import sys
from pathlib import Path

import numpy as np
import pytest

ROOT = Path(__file__).resolve().parents[1]


def _worker(rank, world, port, S, q):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    sys.path.insert(0, str(ROOT))
    sys.path.insert(0, str(ROOT / "tests"))
    import torch.distributed as dist
    import rtamd
    import helpers
    from oracle import cref
    dist.init_process_group("gloo", rank=rank, world_size=world)
    model = rtamd.scenes.make_scene(3, 5, 4, S, seed=13)
    scene = rtamd.prepare_scene(model)
    p_full = cref.pack_scene(helpers.oracle_scene(model))

    def run_local(shard):
        # oracle on this rank's points only, with the GLOBAL ndoubl / iface of the unsharded scene
        lo, hi = rtamd.sharding.shard_bounds(scene.S, world, rank)
        assert np.array_equal(shard.ndoubl, scene.ndoubl) and shard.S == hi - lo
        R, T, info = cref.rt_run(p_full, pts=np.arange(lo, hi, dtype=np.int32), nthreads=2)
        assert info == 0
        return R[:, :, lo:hi], T[:, :, lo:hi]

    R, T = rtamd.sharding.rt_run_sharded(scene, run_local, dist)
    if rank == 0:
        q.put((R, T))
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("S", [10, 7])  # even split and a ragged tail
def test_two_rank_sharded_run_matches_single(S):
    import torch.multiprocessing as mp
    sys.path.insert(0, str(ROOT / "tests"))
    import rtamd
    import helpers
    from oracle import cref
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, S, q)) for r in range(2)]
    for p in procs:
        p.start()
    R, T = q.get(timeout=180)
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    model = rtamd.scenes.make_scene(3, 5, 4, S, seed=13)
    Rr, Tr, _ = cref.rt_run(cref.pack_scene(helpers.oracle_scene(model)))
    assert np.array_equal(R, Rr) and np.array_equal(T, Tr)  # bit for bit


def test_shard_bounds():
    import rtamd
    sb = rtamd.sharding.shard_bounds
    assert [sb(10, 4, r) for r in range(4)] == [(0, 3), (3, 6), (6, 9), (9, 10)]
    assert [sb(2, 4, r) for r in range(4)] == [(0, 1), (1, 2), (2, 2), (2, 2)]
    assert sb(29_944, 8, 7) == (26_201, 29_944)
