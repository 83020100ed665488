#!/usr/bin/env python3
"""bench.py -- spectral points / second of the CoreRT layer-adding sweep on MI355X.

Workload (BASELINE.json configs[1], "C2"): O2 A-band, Stokes IQU, 20 quadrature streams
(N = 60), 40 layers, 3 Fourier moments, 10 000 spectral points per GPU, Float64, seeded synthetic
scene (radiativetransfer.jl_amd/scenes.py).  One "step" = the whole of rt_run.jl:125-215 for the
resident scene: every layer's elemental -> doubling -> interaction, the surface, post-processing,
and (N > 1) the ONE RCCL all-gather of the R/T spectra.  Inputs are in HBM before the timed region.

  python bench.py [--gpus N --steps K --warmup W]

N > 1: one process per GPU.  The driver's launcher (`python -m torch.distributed.run ... bench.py --gpus N`)
sets WORLD_SIZE/RANK/LOCAL_RANK; started WITHOUT it, `--gpus N` spawns that launcher as a child process before
anything touches a GPU and exits with its code.  Collective back ends (`--backend`):
  rccl   (default) mom_comm_init + mom_allgather_RT_device: RCCL through the C ABI on the library's own stream;
         torch.distributed (gloo) is only the control plane (id exchange, barrier, max over ranks)
  torch  torch.distributed all_gather_into_tensor on the "nccl" (= RCCL) process group
  gloo   host all-gather over gloo; with --share-device every rank uses GPU 0 (the 1-GPU test box)

Other workloads (`--workload`): C1 (scalar, N = 4: HBM-bound wave-per-point kernel), C4 (IQUV, 64 streams,
N = 256).  Prints ONE JSON line on rank 0 (contract in the task description).  Weak scaling: every rank owns
`--points` spectral points of a global axis of N x points; ndoubl / interface codes are computed on the global axis.
"""
import argparse
import json
import os
import socket
import subprocess
import sys
import time
from pathlib import Path

import numpy as np

ROOT = Path(__file__).resolve().parent
sys.path.insert(0, str(ROOT))

PEAK_FP64_MFMA_TFLOPS = 78.6  # AMD MI355X FP64 matrix spec; v_mfma_f64_16x16x4 issue-rate microbenchmark: 77.5
PEAK_HBM_GBS = 8000.0         # MI355X_MICROARCH.md: 8 TB/s spec (6.3 TB/s achievable)
DEFAULT_POINTS = {"C2": 10_000, "C1": 200_000, "C4": 2_000}


def cpu_baseline(model, workload, budget_s=12.0):
    """The C oracle (oracle/momref.c, kind "port": the reference is Julia and cannot run here) on all the
    host cores this process may use over a bounded seeded sample of the same scene's spectral points."""
    sys.path.insert(0, str(ROOT / "tests"))
    import helpers
    from oracle import cref
    import rtamd
    cores = cref.effective_cores()  # affinity mask and cgroup CPU quota, not the machine's logical CPU count
    sc = helpers.oracle_scene(model)
    p = cref.pack_scene(sc)
    rng = np.random.default_rng(0)
    done, t_used, batch = 0, 0.0, cores * (64 if workload == "C1" else 1)
    while t_used < budget_s and done < p.S:
        pts = rng.choice(p.S, min(batch, p.S), replace=False).astype(np.int32)
        t0 = time.perf_counter()
        cref.rt_run(p, pts=pts, nthreads=cores)
        t_used += time.perf_counter() - t0
        done += len(pts)
    flop_pt = rtamd.scenes.work_model_flops(p.N, p.nd, p.M)
    gfs_core = done * flop_pt / t_used / cores / 1e9
    return {"value": done / t_used, "unit": "spectral points/s", "cores": cores, "kind": "port",
            "gflops_per_core": gfs_core,
            "sample": f"{done} seeded random spectral points of the same {workload} scene, {t_used:.1f} s, OpenMP over points, "
                      f"blocked AVX2 GEMM: {gfs_core:.2f} algorithmic GFLOP/s per core"}


def voigt_leg():
    """Second kernel of the north star, reported as `extra.voigt` with its own roofline."""
    import bench_voigt
    return bench_voigt.run(repeats=5)


def spawn_ranks(a):
    """`python bench.py --gpus N` without a launcher: start the driver's launcher as a CHILD (this process has not
    touched the GPU and never will) and pass its exit code on."""
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={a.gpus}",
           "--master-addr", "127.0.0.1", "--master-port", str(port), str(Path(__file__).resolve())] + sys.argv[1:]
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    return subprocess.call(cmd, env=env)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=3)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--workload", choices=["C2", "C1", "C4"], default="C2")
    ap.add_argument("--points", type=int, default=0, help="spectral points per GPU (default: the workload's size)")
    ap.add_argument("--backend", choices=["rccl", "torch", "gloo"], default="rccl")
    ap.add_argument("--share-device", action="store_true", help="all ranks on GPU 0 (needs --backend gloo)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-voigt", action="store_true")
    ap.add_argument("--opt", action="append", default=[], metavar="ID=VALUE", help="mom_set_option(ID, VALUE) before the scene is set (kernel A/B runs)")
    a = ap.parse_args()
    if a.gpus < 1:
        ap.error("--gpus must be >= 1")
    if a.share_device and a.backend != "gloo":
        ap.error("--share-device needs --backend gloo (RCCL refuses two ranks on one GPU)")
    if a.gpus > 1 and "WORLD_SIZE" not in os.environ:
        sys.exit(spawn_ranks(a))

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if world != a.gpus:
        print(f"bench.py: --gpus {a.gpus} but the launcher started {world} rank(s); refusing to report a wrong n_gpus",
              file=sys.stderr)
        sys.exit(2)

    import torch
    import rtamd

    dev_index = 0 if (a.share_device or world == 1) else local
    torch.cuda.set_device(dev_index)
    dev = torch.device("cuda", dev_index)
    dist = None
    nccl_group = None
    collective = "none"
    if world > 1:
        import torch.distributed as dist
        if a.backend == "torch":
            dist.init_process_group("nccl", device_id=dev)  # "nccl" is RCCL on ROCm
            collective = "torch.distributed all_gather_into_tensor (RCCL)"
        else:
            dist.init_process_group("gloo")
            collective = "mom_allgather_RT_device (RCCL through the C ABI)" if a.backend == "rccl" else "gloo host all_gather"

    S_loc = a.points or DEFAULT_POINTS[a.workload]
    S_tot = S_loc * world
    scene_fn = {"C2": rtamd.scenes.scene_C2, "C1": rtamd.scenes.scene_C1, "C4": rtamd.scenes.scene_C4}[a.workload]
    model = scene_fn(S=S_tot, architecture=rtamd.MI355X(dev.index))
    scene = rtamd.prepare_scene(model)          # global axis: ndoubl / iface are global
    shard = scene.spectral_slice(rank * S_loc, (rank + 1) * S_loc) if world > 1 else scene
    h = rtamd.corert.make_handle(model, S=S_loc)
    for kv in a.opt:
        k, v = kv.split("=")
        h.set_option(int(k), int(v))
    h.scene_set(shard.Nz, shard.K, shard.M, shard.tau, shard.varpi, shard.zw, shard.Zpp, shard.Zmp, shard.ndoubl,
                shard.iface, shard.tau_sum, shard.albedo, shard.node, shard.cos_mphi, shard.sin_mphi)
    nout = len(shard.node) * shard.nStokes * S_loc
    RT = torch.empty(2 * nout, dtype=torch.float64, device=dev)           # local R || T
    G = torch.empty(2 * nout * world, dtype=torch.float64, device=dev) if world > 1 else None
    Gh = torch.empty(2 * nout * world, dtype=torch.float64) if (world > 1 and a.backend == "gloo") else None
    if world > 1 and a.backend == "rccl":
        # RCCL through the C ABI.  Should its set-up fail on ANY rank (a mis-matched RCCL/HIP pair in the host process,
        # for instance), every rank falls back -- together -- to torch.distributed's own RCCL group, so that a scaling
        # run still produces its line; the JSON says which collective ran.
        ok = 1
        try:
            ident = [rtamd._lib.comm_unique_id() if rank == 0 else None]
        except Exception as e:
            ident, ok = [None], 0
            print(f"bench.py: mom_comm_unique_id failed on rank {rank}: {e}", file=sys.stderr)
        dist.broadcast_object_list(ident, src=0)
        if ok and ident[0] is not None:
            try:
                h.comm_init(rank, world, ident[0])
            except Exception as e:
                ok = 0
                print(f"bench.py: mom_comm_init failed on rank {rank}: {e}", file=sys.stderr)
        else:
            ok = 0
        flag = torch.tensor([ok], dtype=torch.int32)
        dist.all_reduce(flag, op=dist.ReduceOp.MIN)
        if int(flag.item()) == 0:
            nccl_group = dist.new_group(backend="nccl")
            a.backend = "torch-fallback"
            collective = "torch.distributed all_gather_into_tensor (RCCL; fallback: the C-ABI communicator failed to initialise)"

    def step():
        h.rt_run()
        if world == 1:
            h.get_RT_device(RT.data_ptr(), RT.data_ptr() + 8 * nout)      # asynchronous, library stream
        elif a.backend == "rccl":
            h.allgather_RT_device(G.data_ptr())                           # ONE collective, library stream
        elif a.backend in ("torch", "torch-fallback"):
            h.get_RT_device(RT.data_ptr(), RT.data_ptr() + 8 * nout)
            h.sync()
            dist.all_gather_into_tensor(G, RT, group=nccl_group)          # ONE collective: R and T packed
        else:
            h.get_RT_device(RT.data_ptr(), RT.data_ptr() + 8 * nout)
            h.sync()
            dist.all_gather_into_tensor(Gh, RT.cpu())

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
    h.check_async()  # deferred singular-operator report of the asynchronous steps
    if world > 1:
        t = torch.tensor([el], dtype=torch.float64, device=dev if a.backend == "torch" else "cpu")
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        el = float(t.item())
        # the gathered block of this rank must be its own spectra, and every block finite
        R_loc, T_loc = h.get_RT()
        mine = np.concatenate([np.transpose(R_loc, (2, 1, 0)).reshape(-1), np.transpose(T_loc, (2, 1, 0)).reshape(-1)])
        got = (Gh if a.backend == "gloo" else G.cpu()).numpy().reshape(world, 2 * nout)
        assert np.array_equal(got[rank], mine) and np.all(np.isfinite(got)), "all-gather returned wrong data"
    tm = h.timers()  # of the last step, HIP events on the library's stream

    if rank == 0:
        N, M = scene.N, scene.M
        nd = scene.ndoubl.astype(np.float64)
        # ALGORITHMIC flop of the layer kernels per spectral point and Fourier moment (the surface
        # interaction runs in k_surface): GEMM = 2N^3, inverse = 2N^3, matvec = 2N^2 -- the reference's op list
        # on the FULL N x N operators, whatever the kernels do internally (Neumann series, m = 0 sub-problem)
        f_pm = (nd.sum() * (12 * N ** 3 + 8 * N ** 2) + (scene.Nz - 1) * (24 * N ** 3 + 8 * N ** 2) + scene.Nz * N * N * 15)
        # dominant kernel = the full-problem layer kernel; it handles moments 1..M-1 when moment 0 runs as the (I,Q)
        # sub-problem (reduced_launches > 0), else all M moments
        m_dom = (M - 1) if tm["reduced_launches"] > 0 else M
        flop_dom = f_pm * m_dom * S_loc
        whole = f_pm * M * S_loc / (tm["layers_ms"] * 1e-3) / 1e12
        prof = {}
        tf = ROOT / "profiles" / "traffic.json"
        if tf.exists():
            prof = json.loads(tf.read_text()).get(a.workload, {})
        if a.workload == "C1":
            # N = 4: one spectral point per lane, operators in registers (mom_small.hip).  HBM moves only the per-point
            # inputs and outputs -- algorithmic bytes per point = (tau, varpi, tau_sum + K weights) x Nz x 8
            # + 3 x nVza x nStokes x 8 + 2 x nStokes x 8 -- which at the measured rate is < 1 % of the HBM roofline: the
            # bound of this regime is the FP64 vector-FMA pipe (the kernel issues no MFMA: a 4 x 4 operator would use 1/16
            # of a 16 x 16 tile), whose peak on MI355X equals the FP64 MFMA peak (78.6 TFLOP/s).  `bound` keeps the
            # contract's vocabulary ("mfma" = the FP64 arithmetic peak); the HBM figures are reported next to it.
            bytes_pt = (3 + scene.K) * scene.Nz * 8 + 3 * len(scene.node) * scene.nStokes * 8 + 2 * scene.nStokes * 8
            gbs = bytes_pt * S_loc / (tm["layers_ms"] * 1e-3) / 1e9
            ach = f_pm * M * S_loc / (tm["layers_ms"] * 1e-3) / 1e12
            roof = {"bound": "mfma", "achieved": ach, "peak": PEAK_FP64_MFMA_TFLOPS, "unit": "TFLOP/s",
                    "frac": ach / PEAK_FP64_MFMA_TFLOPS, "traffic": prof.get("hbm_bytes_per_launch"),
                    "kernel": "momsm::k_sweep<4> (FP64 vector FMA, no MFMA issued)",
                    "avg_launch_ms": tm["layers_ms"] / max(tm["layer_launches"], 1), "launches_per_step": tm["layer_launches"],
                    "algorithmic_flop_per_avg_launch": f_pm * M * S_loc,
                    "hbm": {"algorithmic_bytes_per_point": bytes_pt, "achieved_GBps": gbs, "frac_of_8TBps": gbs / PEAK_HBM_GBS}}
        else:
            achieved = flop_dom / (tm["full_layers_ms"] * 1e-3) / 1e12
            roof = {"bound": "mfma", "achieved": achieved, "peak": PEAK_FP64_MFMA_TFLOPS, "unit": "TFLOP/s",
                    "frac": achieved / PEAK_FP64_MFMA_TFLOPS, "traffic": prof.get("hbm_bytes_per_launch"),
                    "kernel": prof.get("kernel", "full-problem layer kernel"),
                    "avg_launch_ms": tm["full_layers_ms"] / max(tm["full_launches"], 1),
                    "launches_per_step": tm["full_launches"], "moments_per_launch": m_dom,
                    "algorithmic_flop_per_avg_launch": flop_dom / max(tm["full_launches"], 1),
                    "all_layer_kernels_achieved": whole, "all_layer_kernels_frac": whole / PEAK_FP64_MFMA_TFLOPS,
                    # `achieved` counts the reference's op list (full 2N^3 inverses); the hardware-side figure is the
                    # MFMA-busy share of SIMD cycles from the PMC pass (profiles/), not recomputed here
                    "hw_mfma_busy_frac": prof.get("mfma_busy_frac")}
        out = {
            "metric": "spectral points/sec (whole node), O2-A band IQU scene", "value": S_tot / (el / a.steps),
            "unit": "spectral points/s", "n_gpus": world, "steps": a.steps, "warmup": a.warmup,
            "ms_per_step": el / a.steps * 1e3, "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": "f64", "data": "synthetic",
            "config": {"workload": f"{a.workload}: N={N} ({N // scene.nStokes} streams x {scene.nStokes} Stokes), Nz={scene.Nz}, "
                                   f"M={M}, S={S_loc}/GPU, sum(ndoubl)={int(nd.sum())}, Lambertian surface, {len(scene.node)} VZA",
                       "sharding": f"spectral axis, {world} x {S_loc} points, one all-gather of R||T" if world > 1 else "none",
                       "collective": collective,
                       "devices": "all ranks on GPU 0 (--share-device)" if a.share_device else "one GPU per rank"},
            "roofline": roof,
            "stages_ms": {k: tm[k] for k in ("layers_ms", "full_layers_ms", "reduced_layers_ms", "surface_ms",
                                             "postprocess_ms", "total_ms")},
        }
        if a.workload != "C2":
            out["metric"] = f"spectral points/sec (whole node), {a.workload} scene"
        if world == 1 and not a.no_cpu_baseline:
            out["cpu_baseline"] = cpu_baseline(model, a.workload)
        if world == 1 and not a.no_voigt and a.workload == "C2":
            try:
                out["extra"] = {"voigt": voigt_leg()}
            except Exception as e:  # the headline number must not depend on the second kernel's leg
                out["extra"] = {"voigt": {"error": repr(e)}}
        print(json.dumps(out), flush=True)
    h.close()
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
