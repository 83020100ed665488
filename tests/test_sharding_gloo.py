"""The N > 1 path on CPU: two and EIGHT processes (the node size north_star names), gloo backend.  The sharding + all-gather plumbing of
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
        R, T, info = cref.rt_run(p_full, pts=np.arange(lo, hi, dtype=np.int32), nthreads=1)
        assert info == 0
        return R[:, :, lo:hi], T[:, :, lo:hi]

    R, T = rtamd.sharding.rt_run_sharded(scene, run_local, dist)
    if rank == 0:
        q.put((R, T))
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("S,world", [(10, 2), (7, 2), (32, 8), (29, 8), (6, 8)])  # even split, ragged tail, ranks without points
def test_sharded_run_matches_single(S, world):
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
    procs = [ctx.Process(target=_worker, args=(r, world, port, S, q)) for r in range(world)]
    for p in procs:
        p.start()
    R, T = q.get(timeout=300)
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


def test_rrs_window():
    """Owned slice + halo of max |i_λ₁λ₀| on interior edges, clipped at the ends of the axis; empty tails stay empty."""
    import rtamd
    rw = rtamd.sharding.rrs_window
    offs = [-4, -1, 2, 7, 3]
    assert [rw(40, 5, r, offs) for r in range(5)] == [(0, 8, 0, 15), (8, 16, 1, 23), (16, 24, 9, 31), (24, 32, 17, 39), (32, 40, 25, 40)]
    assert rw(6837, 8, 3, [-5000, 4000]) == (2565, 3420, 0, 6837)   # halo longer than the axis: the whole axis
    assert rw(2, 4, 3, offs) == (2, 2, 2, 2)
    for S, world in ((40, 3), (7, 4), (6837, 8)):
        b = [rw(S, world, r, offs) for r in range(world)]
        assert b[0][0] == 0 and b[-1][1] == S and all(b[i][1] == b[i + 1][0] for i in range(world - 1))
        assert all(w[2] <= w[0] and w[1] <= w[3] for w in b)


def _gather_worker(rank, world, port, S, q):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    sys.path.insert(0, str(ROOT))
    sys.path.insert(0, str(ROOT / "tests"))
    import torch.distributed as dist
    import rtamd
    dist.init_process_group("gloo", rank=rank, world_size=world)
    lo, hi = rtamd.sharding.shard_bounds(S, world, rank)
    n = np.arange(lo, hi, dtype=np.float64)
    loc = [np.stack([[n + 100 * v + 10 * k for k in range(3)] for v in range(2)]), -n, np.stack([n * n, 2 * n])]  # [2,3,.], [.], [2,.]
    full = rtamd.sharding.gather_spectra(loc, S, dist)
    if rank == 1:
        q.put(full)
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("S,world", [(10, 2), (7, 2), (2, 3), (29, 8), (5, 8)])  # even, ragged, ranks with no points, eight ranks
def test_gather_spectra_over_gloo(S, world):
    """The collective of the sharded RRS run: one all-gather of the seven spectra of the return tuple (any leading shape)."""
    import torch.multiprocessing as mp
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_gather_worker, args=(r, world, port, S, q)) for r in range(world)]
    for p in procs:
        p.start()
    full = q.get(timeout=180)
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    n = np.arange(S, dtype=np.float64)
    assert np.array_equal(full[0], np.stack([[n + 100 * v + 10 * k for k in range(3)] for v in range(2)]))
    assert np.array_equal(full[1], -n) and np.array_equal(full[2], np.stack([n * n, 2 * n]))


def test_unpack_rrs_spectra_layout():
    """sharding.unpack_rrs_spectra inverts the packing of mom_get_spectra_rrs_device / mom_allgather_rrs_device: per rank
    [R | T | ieR | ieT | hdr][per, nStokes, nVza] then [bhr_uw | bhr_dw][per, nStokes] (the ABI's order: spectral index
    slowest); ragged tail cut off."""
    import rtamd
    rng = np.random.default_rng(0)
    nV, nS, per, world, S = 3, 4, 5, 3, 13
    full = [rng.standard_normal((nV, nS, world * per)) for _ in range(5)] + [rng.standard_normal((nS, world * per)) for _ in range(2)]
    G = np.stack([np.concatenate([np.ascontiguousarray(x[..., r * per:(r + 1) * per].T).reshape(-1) for x in full])
                  for r in range(world)])
    assert G.shape == (world, (5 * nV * nS + 2 * nS) * per)
    got = rtamd.sharding.unpack_rrs_spectra(G, nV, nS, per, S)
    for x, y in zip(got, full):
        assert np.array_equal(x, y[..., :S])


def _rrs_worker(rank, world, port, S, q):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    sys.path.insert(0, str(ROOT))
    sys.path.insert(0, str(ROOT / "tests"))
    import torch.distributed as dist
    import rtamd
    import helpers
    from oracle import cref, momref as mr, rrsref as rr
    dist.init_process_group("gloo", rank=rank, world_size=world)
    scene, offs, w = _rrs_case(rtamd, helpers, S)
    lo, hi, wlo, whi = rtamd.sharding.rrs_window(S, world, rank, offs)
    assert wlo <= lo and hi <= whi
    nV, nS = len(scene.vza), scene.pol.n
    if hi > lo:   # the C oracle on this rank's owned window (its elastic part covers the halo, like the GPU's window run)
        ora = rr.RRSInputs(offs, w, mr.get_greek_rayleigh(0.2), rrs_strict_reference=False, owned=(lo, hi))
        loc = [x[..., lo:hi] for x in cref.rt_run_rrs(scene, ora, nthreads=1)[:4]]
    else:
        loc = [np.zeros((nV, nS, 0))] * 4
    full = rtamd.sharding.gather_spectra(loc, S, dist)
    if rank == world - 1:
        q.put(full)
    dist.barrier()
    dist.destroy_process_group()


def _rrs_case(rtamd, helpers, S):
    m = rtamd.scenes.make_scene(3, 5, 3, S, seed=31, aerosol_total=0.1, vza=(30.0,), vaz=(20.0,))
    scene = helpers.oracle_scene(m)
    scene.varpi_cabannes = 0.96
    offs = np.array([-5, -2, 3, 6], dtype=np.int64)
    return scene, offs, 0.02 * (1.0 + 0.1 * np.arange(4))


@pytest.mark.parametrize("S,world", [(30, 2), (37, 8)])
def test_rrs_windows_and_gather_over_gloo(S, world):
    """rt_run(::RRS) sharded over 2 and 8 processes: sharding.rrs_window (owned slice + halo of max |i_l1l0|), the C oracle's
    owned-window run standing in for the GPU's, one all-gather of the spectra (sharding.gather_spectra): bitwise the
    single-process run."""
    import torch.multiprocessing as mp
    sys.path.insert(0, str(ROOT / "tests"))
    import rtamd
    import helpers
    from oracle import cref, momref as mr, rrsref as rr
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_rrs_worker, args=(r, world, port, S, q)) for r in range(world)]
    for p in procs:
        p.start()
    full = q.get(timeout=300)
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    scene, offs, w = _rrs_case(rtamd, helpers, S)
    ref = cref.rt_run_rrs(scene, rr.RRSInputs(offs, w, mr.get_greek_rayleigh(0.2), rrs_strict_reference=False), nthreads=1)
    assert np.abs(ref[2]).max() > 0
    for k in range(4):
        assert np.array_equal(full[k], ref[k]), k


def _dual_worker(rank, world, port, S, q):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    sys.path.insert(0, str(ROOT))
    sys.path.insert(0, str(ROOT / "tests"))
    import torch.distributed as dist
    import rtamd
    import helpers
    from oracle import dualref as dr
    dist.init_process_group("gloo", rank=rank, world_size=world)
    model, sc, parts, host = _dual_case(rtamd, helpers, dr, S)
    scene = rtamd.prepare_scene(model)
    L_full = dr.layer_inputs(sc)

    def run_local(shard, shard_parts):
        # the Dual oracle on this rank's points, with the GLOBAL ndoubl / interface codes of the unsharded scene
        lo, hi = rtamd.sharding.shard_bounds(scene.S, world, rank)
        assert np.array_equal(shard.ndoubl, scene.ndoubl) and shard.S == hi - lo and shard_parts[0].dτ.shape[0] == hi - lo
        from dataclasses import replace
        Ls = replace(L_full, tau=L_full.tau[lo:hi], varpi=L_full.varpi[lo:hi], zw=L_full.zw[:, lo:hi])
        scs = replace(sc, tau_rayl=sc.tau_rayl[lo:hi], tau_abs=sc.tau_abs[lo:hi])
        ps = [dr.Partial(dtau=p.dτ, dvarpi=p.dϖ, dzw=p.dzw, dalbedo=p.dalbedo) for p in shard_parts]
        return dr.rt_run_dual(scs, ps, Ls)

    out = rtamd.sharding.rt_run_dual_sharded(scene, host, run_local, dist)
    if rank == 0:
        q.put(out)
    dist.barrier()
    dist.destroy_process_group()


def _dual_case(rtamd, helpers, dr, S):
    model = rtamd.scenes.make_scene(3, 5, 3, S, seed=17, aerosol_total=0.1)
    sc = helpers.oracle_scene(model)
    L = dr.layer_inputs(sc)
    rng = np.random.default_rng(4)
    parts = [dr.Partial(dtau=L.tau * rng.uniform(-1, 1, L.tau.shape), dvarpi=0.1 * L.varpi, dzw=L.zw * rng.uniform(-1, 1, L.zw.shape),
                        dalbedo=float(i + 1)) for i in range(2)]
    host = [rtamd.ScenePartial(dτ=p.dtau, dϖ=p.dvarpi, dzw=p.dzw, dalbedo=p.dalbedo) for p in parts]
    return model, sc, parts, host


@pytest.mark.parametrize("S,world", [(9, 2), (13, 8)])
def test_dual_run_sharded_over_gloo(S, world):
    """rt_run on Dual numbers sharded over 2 and 8 processes (sharding.rt_run_dual_sharded: values and partials of a spectral point
    are independent of every other point; one all-gather of the four arrays), the Dual oracle standing in for the GPU: bitwise
    the single-process run."""
    import torch.multiprocessing as mp
    sys.path.insert(0, str(ROOT / "tests"))
    import rtamd
    import helpers
    from oracle import dualref as dr
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_dual_worker, args=(r, world, port, S, q)) for r in range(world)]
    for p in procs:
        p.start()
    out = q.get(timeout=300)
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    model, sc, parts, host = _dual_case(rtamd, helpers, dr, S)
    ref = dr.rt_run_dual(sc, parts)
    assert np.abs(ref[2]).max() > 0
    for a, b in zip(out, ref):
        assert np.array_equal(a, b)


def test_slice_partials():
    """Per-point partial arrays are cut with the points; per-scene ones (bases, scalar albedo, BRDF matrices) are shared."""
    import rtamd
    rng = np.random.default_rng(0)
    S, Nz, K = 7, 3, 2
    p = rtamd.ScenePartial(dτ=rng.standard_normal((S, Nz)), dϖ=None, dzw=rng.standard_normal((K, S, Nz)), dZpp=np.ones((1, K, 4, 4)),
                           dZmp=np.ones((1, K, 4, 4)), dalbedo=0.5, dalbedo_spec=rng.standard_normal(S))
    q, = rtamd.sharding.slice_partials([p], 2, 5)
    assert np.array_equal(q.dτ, p.dτ[2:5]) and q.dϖ is None and np.array_equal(q.dzw, p.dzw[:, 2:5])
    assert q.dZpp is p.dZpp and q.dalbedo == 0.5 and np.array_equal(q.dalbedo_spec, p.dalbedo_spec[2:5])
