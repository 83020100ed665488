#!/usr/bin/env python3
"""bench.py -- spectral points / second of the CoreRT layer-adding sweep on MI355X.

Workload (BASELINE.json configs[1], "C2"): O2 A-band, Stokes IQU, 20 quadrature streams
(N = 60), 40 layers, 3 Fourier moments, 10 000 spectral points per GPU, Float64, seeded synthetic
scene (radiativetransfer.jl_amd/scenes.py).  One "step" = the whole of rt_run.jl:125-215 for the
resident scene: every layer's elemental -> doubling -> interaction, the surface, post-processing,
and (N > 1) the RCCL all-gather of the R/T spectra.  Inputs are in HBM before the timed region.

  python bench.py [--gpus N --steps K --warmup W]        (N > 1: launched by torch.distributed.run)

Prints ONE JSON line on rank 0 (contract in the task description).  Weak scaling: every rank owns
10 000 points of a global axis of N x 10 000; ndoubl / interface codes are computed on the global axis.
"""
import argparse
import json
import os
import sys
import time
from pathlib import Path

import numpy as np

ROOT = Path(__file__).resolve().parent
sys.path.insert(0, str(ROOT))

S_PER_GPU = 10_000
PEAK_FP64_MFMA_TFLOPS = 78.6  # AMD MI355X FP64 matrix spec; v_mfma_f64_16x16x4 issue-rate microbenchmark: 77.5


def cpu_baseline(model, budget_s=12.0):
    """The C oracle (oracle/momref.c, kind "port": the reference is Julia and cannot run here) on all
    host cores over a bounded seeded sample of the same scene's spectral points."""
    sys.path.insert(0, str(ROOT / "tests"))
    import helpers
    from oracle import cref
    cores = os.cpu_count() or 1
    p = cref.pack_scene(helpers.oracle_scene(model))
    rng = np.random.default_rng(0)
    done, t_used, batch = 0, 0.0, cores
    while t_used < budget_s and done < p.S:
        pts = rng.choice(p.S, batch, replace=False).astype(np.int32)
        t0 = time.perf_counter()
        cref.rt_run(p, pts=pts, nthreads=cores)
        t_used += time.perf_counter() - t0
        done += batch
    return {"value": done / t_used, "unit": "spectral points/s", "cores": cores, "kind": "port",
            "sample": f"{done} seeded random spectral points of the same C2 scene, {t_used:.1f} s, OpenMP over points"}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=3)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--points", type=int, default=S_PER_GPU, help="spectral points per GPU")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    a = ap.parse_args()

    import torch
    import rtamd

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    dist = None
    if world > 1:
        import torch.distributed as dist
        torch.cuda.set_device(local)
        dist.init_process_group("nccl", device_id=torch.device("cuda", local))  # "nccl" is RCCL on ROCm
    else:
        torch.cuda.set_device(0)
    dev = torch.device("cuda", local if world > 1 else 0)

    S_loc, S_tot = a.points, a.points * world
    model = rtamd.scenes.scene_C2(S=S_tot, architecture=rtamd.MI355X(dev.index))
    scene = rtamd.prepare_scene(model)          # global axis: ndoubl / iface are global
    shard = scene.spectral_slice(rank * S_loc, (rank + 1) * S_loc) if world > 1 else scene
    h = rtamd.corert.make_handle(model, S=S_loc)
    h.scene_set(shard.Nz, shard.K, shard.M, shard.tau, shard.varpi, shard.zw, shard.Zpp, shard.Zmp, shard.ndoubl,
                shard.iface, shard.tau_sum, shard.albedo, shard.node, shard.cos_mphi, shard.sin_mphi)
    nout = len(shard.node) * shard.nStokes * S_loc
    R = torch.empty(nout, dtype=torch.float64, device=dev)
    T = torch.empty(nout, dtype=torch.float64, device=dev)
    Rg = torch.empty(nout * world, dtype=torch.float64, device=dev) if world > 1 else None
    Tg = torch.empty(nout * world, dtype=torch.float64, device=dev) if world > 1 else None

    def step():
        h.rt_run()
        h.get_RT_device(R.data_ptr(), T.data_ptr())  # synchronises the library's stream
        if world > 1:
            dist.all_gather_into_tensor(Rg, R)
            dist.all_gather_into_tensor(Tg, T)

    def fence():
        h.sync()
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    for _ in range(a.warmup):
        step()
    fence()
    t0 = time.perf_counter()
    for _ in range(a.steps):
        step()
    fence()
    el = time.perf_counter() - t0
    if world > 1:
        t = torch.tensor([el], dtype=torch.float64, device=dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        el = float(t.item())
    tm = h.timers()  # of the last step, HIP events on the library's stream

    if rank == 0:
        N, M = scene.N, scene.M
        nd = scene.ndoubl.astype(np.float64)
        # ALGORITHMIC flop of the layer kernels per spectral point and Fourier moment (the surface
        # interaction runs in k_surface): GEMM = 2N^3, inverse = 2N^3, matvec = 2N^2 -- the reference's op list
        # on the FULL N x N operators, whatever the kernels do internally (Neumann series, m = 0 sub-problem)
        f_pm = (nd.sum() * (12 * N ** 3 + 8 * N ** 2) + (scene.Nz - 1) * (24 * N ** 3 + 8 * N ** 2) + scene.Nz * N * N * 15)
        # dominant kernel = the full-problem layer kernel mom::k_layer<true, 3, 15> (strip-chained, N = 60); it handles moments 1..M-1 when
        # moment 0 runs as the (I,Q) sub-problem in mom4::k_layer (reduced_launches > 0), else all M moments
        m_dom = (M - 1) if tm["reduced_launches"] > 0 else M
        flop_dom = f_pm * m_dom * S_loc
        achieved = flop_dom / (tm["full_layers_ms"] * 1e-3) / 1e12
        whole = f_pm * M * S_loc / (tm["layers_ms"] * 1e-3) / 1e12
        traffic = None
        tf = ROOT / "profiles" / "traffic.json"
        if tf.exists():
            traffic = json.loads(tf.read_text()).get("k_layer_hbm_bytes_per_launch")
        out = {
            "metric": "spectral points/sec (whole node), O2-A band IQU scene", "value": S_tot / (el / a.steps),
            "unit": "spectral points/s", "n_gpus": world, "steps": a.steps, "warmup": a.warmup,
            "ms_per_step": el / a.steps * 1e3, "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": "f64", "data": "synthetic",
            "config": {"workload": f"C2 O2-A IQU: N=60 (20 streams x 3 Stokes), Nz=40, M=3, S={S_loc}/GPU, "
                                   f"sum(ndoubl)={int(nd.sum())}, Lambertian surface, 3 VZA",
                       "sharding": f"spectral axis, {world} x {S_loc} points, RCCL all_gather of R/T" if world > 1 else "none"},
            "roofline": {"bound": "mfma", "achieved": achieved, "peak": PEAK_FP64_MFMA_TFLOPS, "unit": "TFLOP/s",
                         "frac": achieved / PEAK_FP64_MFMA_TFLOPS, "traffic": traffic,
                         "kernel": "mom::k_layer<true, 3, 15>", "avg_launch_ms": tm["full_layers_ms"] / max(tm["full_launches"], 1),
                         "launches_per_step": tm["full_launches"], "moments_per_launch": m_dom,
                         "algorithmic_flop_per_avg_launch": flop_dom / max(tm["full_launches"], 1),
                         "all_layer_kernels_achieved": whole, "all_layer_kernels_frac": whole / PEAK_FP64_MFMA_TFLOPS},
            "stages_ms": {k: tm[k] for k in ("layers_ms", "full_layers_ms", "reduced_layers_ms", "surface_ms",
                                             "postprocess_ms", "total_ms")},
        }
        if world == 1 and not a.no_cpu_baseline:
            out["cpu_baseline"] = cpu_baseline(model)
        print(json.dumps(out), flush=True)
    h.close()
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
