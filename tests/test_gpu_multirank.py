"""The N > 1 path with the REAL HIP kernels on a one-GPU box: two processes share GPU 0, each runs its spectral
shard through libmomcore.so, the spectra are gathered over gloo (RCCL refuses two ranks on one device) -- and the
result is bitwise the one-rank run.  Also bench.py's own launcher logic (--gpus 2 must start two ranks itself)."""
import json
import os
import socket
import subprocess
import sys
from pathlib import Path

import numpy as np
import pytest

pytestmark = pytest.mark.gpu
ROOT = Path(__file__).resolve().parents[1]


def _worker(rank, world, port, S, q):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    sys.path.insert(0, str(ROOT))
    import torch.distributed as dist
    import rtamd
    dist.init_process_group("gloo", rank=rank, world_size=world)
    model = rtamd.scenes.make_scene(3, 33, 5, S, seed=13)     # N = 60: the strip-chained kernels + m = 0 sub-problem
    scene = rtamd.prepare_scene(model)

    def run_local(shard):
        with rtamd.corert.make_handle(model, S=shard.S) as h:  # GPU 0 for every rank
            return rtamd.corert.run_scene(h, shard)

    R, T = rtamd.sharding.rt_run_sharded(scene, run_local, dist)
    if rank == 0:
        q.put((R, T))
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("S", [40, 33])  # even split and a ragged tail
def test_two_ranks_on_one_gpu_match_single_rank_bitwise(rtamd, S):
    import torch.multiprocessing as mp
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, S, q)) for r in range(2)]
    for p in procs:
        p.start()
    R, T = q.get(timeout=600)
    for p in procs:
        p.join(timeout=120)
        assert p.exitcode == 0
    model = rtamd.scenes.make_scene(3, 33, 5, S, seed=13)
    R1, T1 = rtamd.rt_run(model)[:2]
    assert np.array_equal(R, R1) and np.array_equal(T, T1)


def test_bench_launches_two_ranks_itself():
    """`python bench.py --gpus 2` (no launcher, WORLD_SIZE unset) must start two ranks and report n_gpus = 2."""
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK")}
    out = subprocess.run([sys.executable, str(ROOT / "bench.py"), "--gpus", "2", "--backend", "gloo", "--share-device",
                          "--points", "1024", "--steps", "1", "--warmup", "1", "--no-cpu-baseline", "--no-voigt"],
                         env=env, capture_output=True, text=True, timeout=900)
    assert out.returncode == 0, out.stderr[-2000:]
    line = [ln for ln in out.stdout.splitlines() if ln.startswith("{")][-1]
    j = json.loads(line)
    assert j["n_gpus"] == 2 and j["config"]["sharding"].startswith("spectral axis, 2 x 1024")
    assert j["value"] > 0 and j["scaling"] == "weak"


def _rrs_worker(rank, world, port, S, strict, q):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    sys.path.insert(0, str(ROOT))
    import torch.distributed as dist
    import rtamd
    dist.init_process_group("gloo", rank=rank, world_size=world)
    model, RS = _rrs_case(rtamd, S, strict)
    res = rtamd.corert.rt_run_rrs_sharded(RS, model, dist)
    if rank == world - 1:
        q.put(res)
    dist.barrier()
    dist.destroy_process_group()


def _rrs_case(rtamd, S, strict):
    rt = rtamd.corert
    model = rtamd.scenes.make_scene(3, 5, 3, S, seed=31, aerosol_total=0.1, vza=(30.0,), vaz=(20.0,))
    offs = np.array([-5, -2, 3, 6])
    RS = rt.RRS(greek_raman=rt.get_greek_rayleigh(0.2), ϖ_Cabannes=0.96, ϖ_λ1λ0=0.02 * (1.0 + 0.1 * np.arange(4)), i_λ1λ0=offs,
                rrs_strict_reference=strict)
    return model, RS


@pytest.mark.parametrize("S,strict", [(30, False), (23, True)])  # even split / ragged tail; both switch positions
def test_rrs_two_ranks_with_halo_match_single_rank_bitwise(rtamd, S, strict):
    """rt_run(::RRS) sharded over two processes (windows with a recomputed halo of max |i_λ₁λ₀| = 6 points, one all-gather
    of the seven spectra) against the one-rank run."""
    import torch.multiprocessing as mp
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_rrs_worker, args=(r, 2, port, S, strict, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = q.get(timeout=600)
    for p in procs:
        p.join(timeout=120)
        assert p.exitcode == 0
    model, RS = _rrs_case(rtamd, S, strict)
    one = rtamd.corert.rt_run_rrs(RS, model)
    assert np.abs(one[2]).max() > 0
    for got, ref in zip(res, one):
        assert got.shape == ref.shape and np.array_equal(got, ref)


def test_bench_C5_two_ranks():
    """`python bench.py --workload C5 --gpus 2`: the RRS leg shards with halos and reports the whole-job rate."""
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK")}
    out = subprocess.run([sys.executable, str(ROOT / "bench.py"), "--workload", "C5", "--gpus", "2", "--backend", "gloo", "--share-device",
                          "--points", "600", "--steps", "1", "--warmup", "1", "--no-cpu-baseline"],
                         env=env, capture_output=True, text=True, timeout=900)
    assert out.returncode == 0, out.stderr[-2000:]
    j = json.loads([ln for ln in out.stdout.splitlines() if ln.startswith("{")][-1])
    assert j["n_gpus"] == 2 and "1200 in total" in j["config"]["workload"] and "halo" in j["config"]["sharding"]
    assert j["value"] > 0 and j["roofline"]["bound"] == "hbm"


def _bench(args, timeout=1200):
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK")}
    out = subprocess.run([sys.executable, str(ROOT / "bench.py")] + args + ["--no-cpu-baseline", "--no-voigt", "--no-extras"],
                         env=env, capture_output=True, text=True, timeout=timeout)
    assert out.returncode == 0, out.stderr[-3000:]
    return json.loads([ln for ln in out.stdout.splitlines() if ln.startswith("{")][-1])


def test_bench_C3_strong_scaling_two_ranks_gathered_spectra_bitwise(tmp_path):
    """BASELINE configs[2] as bench.py runs it on N GPUs (--workload C3 --scaling strong): the spectra rank 0 holds after
    the all-gather of the 2-rank run are, bit for bit, the spectra of the 1-rank run of the same axis (every rank's block,
    not only the rank's own: global ndoubl / interface codes, slice boundaries, the order of the gathered blocks)."""
    f1, f2 = tmp_path / "one.npy", tmp_path / "two.npy"
    common = ["--workload", "C3", "--points", "2048", "--steps", "1", "--warmup", "0"]
    j1 = _bench(common + ["--gpus", "1", "--dump-spectra", str(f1)])
    j2 = _bench(common + ["--gpus", "2", "--scaling", "strong", "--backend", "gloo", "--share-device", "--dump-spectra", str(f2)])
    assert j1["n_gpus"] == 1 and j2["n_gpus"] == 2 and j2["scaling"] == "strong"
    assert "2 x 1024 points" in j2["config"]["sharding"]
    a, b = np.load(f1), np.load(f2)
    assert a.shape == b.shape and a.shape[0] == 2 and a.shape[-1] == 2048
    assert np.all(np.isfinite(a)) and np.abs(a).max() > 0
    assert np.array_equal(a, b)


def test_bench_C3_strong_scaling_eight_ranks_full_size_bitwise(tmp_path):
    """BASELINE configs[2] at FULL size over EIGHT ranks (north_star's node size; all on GPU 0, gloo: RCCL refuses several
    ranks on one device): 29 944 = 8 x 3 743 points, strong scaling, one all-gather -- every rank's block of the gathered
    spectra is bitwise the 1-rank run of the same axis."""
    f1, f8 = tmp_path / "one.npy", tmp_path / "eight.npy"
    common = ["--workload", "C3", "--steps", "1", "--warmup", "0"]
    j1 = _bench(common + ["--gpus", "1", "--dump-spectra", str(f1)])
    j8 = _bench(common + ["--gpus", "8", "--scaling", "strong", "--backend", "gloo", "--share-device", "--dump-spectra", str(f8)], timeout=1800)
    assert j1["n_gpus"] == 1 and j8["n_gpus"] == 8 and j8["scaling"] == "strong"
    assert "8 x 3743 points" in j8["config"]["sharding"], j8["config"]["sharding"]
    a, b = np.load(f1), np.load(f8)
    assert a.shape == b.shape and a.shape[0] == 2 and a.shape[-1] == 29_944
    assert np.all(np.isfinite(a)) and np.abs(a).max() > 0
    assert np.array_equal(a, b)


def test_bench_C5_eight_ranks_gathered_spectra_bitwise(tmp_path):
    """The RRS leg over EIGHT ranks on one GPU (windows with recomputed halos, one all-gather of the packed owned spectra)
    against the one-rank run of the same 2 400-point axis."""
    f1, f8 = tmp_path / "one.npy", tmp_path / "eight.npy"
    _bench(["--workload", "C5", "--points", "2400", "--steps", "1", "--warmup", "0", "--gpus", "1", "--dump-spectra", str(f1)])
    j8 = _bench(["--workload", "C5", "--points", "300", "--steps", "1", "--warmup", "0", "--gpus", "8", "--backend", "gloo",
                 "--share-device", "--dump-spectra", str(f8)], timeout=1800)
    assert j8["n_gpus"] == 8 and "2400 in total" in j8["config"]["workload"]
    a, b = np.load(f1), np.load(f8)
    assert a.shape == b.shape and a.shape[-1] == 2400 and np.abs(a[2]).max() > 0
    assert np.array_equal(a, b)


def test_bench_C5_two_ranks_gathered_spectra_bitwise(tmp_path):
    """The RRS leg over two ranks (windows with a recomputed halo; device-side pack of the owned slices,
    mom_get_spectra_rrs_device, then one all-gather) against the one-rank run of the same 1 200-point axis: R, T, ieR, ieT,
    hdr bit for bit."""
    f1, f2 = tmp_path / "one.npy", tmp_path / "two.npy"
    j1 = _bench(["--workload", "C5", "--points", "1200", "--steps", "1", "--warmup", "0", "--gpus", "1", "--dump-spectra", str(f1)])
    j2 = _bench(["--workload", "C5", "--points", "600", "--steps", "1", "--warmup", "0", "--gpus", "2", "--backend", "gloo",
                 "--share-device", "--dump-spectra", str(f2)])
    assert "1200 in total" in j1["config"]["workload"].replace("1200/GPU (1200 in total)", "1200 in total") and "1200 in total" in j2["config"]["workload"]
    a, b = np.load(f1), np.load(f2)
    assert a.shape == b.shape == (5,) + a.shape[1:] and a.shape[-1] == 1200
    assert np.abs(a[2]).max() > 0
    assert np.array_equal(a, b)
