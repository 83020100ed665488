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

Other workloads (`--workload`): C1 (scalar, N = 4: lane-per-point kernel), C3 (BASELINE configs[2]: three bands,
29 944 points; `--scaling strong` splits that FIXED axis over the ranks), C4 (IQUV, 64 streams, N = 256), C5 (rotational
Raman, N = 15, 6 837 points x 178 Raman lines: HBM-bound pair kernels).  Prints ONE JSON line on rank 0 (contract in
the task description).  Weak scaling (default): every rank owns `--points` spectral points of a global axis of
N x points; ndoubl / interface codes are computed on the global axis.  The default run (C2, one GPU) also carries
`extra.workloads.{C1,C4,C5}` -- each with its own `roofline` and `cpu_baseline` -- and `extra.f32`, so that the other
regimes are driver-timed too (`--no-extras` skips them).
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
PEAK_FP32_MFMA_TFLOPS = 157.3  # MI355X_MICROARCH.md: FP32 matrix
DEFAULT_POINTS = {"C2": 10_000, "C1": 200_000, "C3": 29_944, "C4": 2_000, "C5": 6_837}


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
    out = bench_voigt.run(repeats=5)
    try:
        out["operating_point"] = bench_voigt.run_profile()   # the reference's setting: 40 cm^-1 wings, 0.015 cm^-1, 40 layers
    except Exception as e:
        out["operating_point"] = {"error": repr(e)}
    return out


def spawn_ranks(a):
    """`python bench.py --gpus N` without a launcher: start the driver's launcher as a CHILD (this process has not
    touched the GPU and never will) and pass its exit code on."""
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={a.gpus}",
           "--master-addr", "127.0.0.1", "--master-port", str(port), str(Path(__file__).resolve())] + sys.argv[1:]
    env = dict(os.environ)
    # RCCL / CUDA-tensor sharing between processes needs dmabuf IPC on this image's host driver (the legacy IPC mode fails
    # with hipIpcGetMemHandle: invalid argument); the variable is already exported on the pool's boxes, kept for others
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    return subprocess.call(cmd, env=env)


class stdout_to_stderr:
    """File descriptor 1 -> 2 for the duration of the block (C-level writers included), see main()."""

    def __enter__(self):
        sys.stdout.flush()
        self.saved = os.dup(1)
        os.dup2(2, 1)
        return self

    def __exit__(self, *exc):
        sys.stdout.flush()
        os.dup2(self.saved, 1)
        os.close(self.saved)
        return False


def init_collective(h, backend, rank, world, dist, collective):
    """The communicator of the run's ONE collective.  backend "rccl": RCCL through the C ABI (mom_comm_init).  Should its
    set-up fail on ANY rank (a mis-matched RCCL/HIP pair in the host process, for instance), every rank falls back -- together
    -- to torch.distributed's own RCCL group, so that a scaling run still produces its line; the JSON says which collective
    ran.  Returns (backend, torch group or None, description)."""
    import torch
    import rtamd
    group = None
    if backend != "rccl":
        return backend, group, collective
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
        with stdout_to_stderr():
            group = dist.new_group(backend="nccl")
        backend = "torch-fallback"
        collective = "torch.distributed all_gather_into_tensor (RCCL; fallback: the C-ABI communicator failed to initialise)"
    return backend, group, collective


def elastic_leg(a, workload, S_loc, steps, warmup, world, rank, dev, dist, with_cpu=True, cpu_budget=12.0):
    """One elastic workload (C1..C4) on this rank's GPU: returns the JSON dict of the leg on rank 0 (None elsewhere)."""
    import torch
    import rtamd
    strong = a.scaling == "strong" and world > 1
    S_tot = S_loc if strong else S_loc * world
    if strong:
        if S_tot % world:
            raise SystemExit(f"--scaling strong: {S_tot} points do not split evenly over {world} ranks")
        S_loc = S_tot // world
    nccl_group = None
    collective = "none"
    backend = a.backend
    if world > 1:
        collective = {"torch": "torch.distributed all_gather_into_tensor (RCCL)",
                      "rccl": "mom_allgather_RT_device (RCCL through the C ABI)", "gloo": "gloo host all_gather"}[backend]
    scene_fn = {"C2": rtamd.scenes.scene_C2, "C1": rtamd.scenes.scene_C1, "C3": rtamd.scenes.scene_C3,
                "C4": rtamd.scenes.scene_C4}[workload]
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
    Gh = torch.empty(2 * nout * world, dtype=torch.float64) if (world > 1 and backend == "gloo") else None
    if world > 1:
        backend, nccl_group, collective = init_collective(h, a.backend, rank, world, dist, collective)

    def step():
        h.rt_run()
        if world == 1:
            h.get_RT_device(RT.data_ptr(), RT.data_ptr() + 8 * nout)      # asynchronous, library stream
        elif backend == "rccl":
            h.allgather_RT_device(G.data_ptr())                           # ONE collective, library stream
        elif backend in ("torch", "torch-fallback"):
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

    for _ in range(warmup):
        step()
    fence()
    t0 = time.perf_counter()
    for _ in range(steps):
        step()
    fence()
    el = time.perf_counter() - t0
    h.check_async()  # deferred singular-operator report of the asynchronous steps
    if world > 1:
        t = torch.tensor([el], dtype=torch.float64, device=dev if backend == "torch" else "cpu")
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        el = float(t.item())
        # the gathered block of this rank must be its own spectra, and every block finite
        R_loc, T_loc = h.get_RT()
        mine = np.concatenate([np.transpose(R_loc, (2, 1, 0)).reshape(-1), np.transpose(T_loc, (2, 1, 0)).reshape(-1)])
        got = (Gh if backend == "gloo" else G.cpu()).numpy().reshape(world, 2 * nout)
        assert np.array_equal(got[rank], mine) and np.all(np.isfinite(got)), "all-gather returned wrong data"
        if a.dump_spectra and rank == 0:     # [world][R|T][S_loc, nStokes, nVza] -> R, T [nVza, nStokes, S_tot]
            g5 = got.reshape(world, 2, S_loc, shard.nStokes, len(shard.node))
            RT_full = np.transpose(g5, (1, 4, 3, 0, 2)).reshape(2, len(shard.node), shard.nStokes, S_tot)
            np.save(a.dump_spectra, RT_full)
    elif a.dump_spectra:
        R_loc, T_loc = h.get_RT()
        np.save(a.dump_spectra, np.stack([R_loc, T_loc]))
    tm = h.timers()  # of the last step, HIP events on the library's stream
    kernel_timing = "HIP events of the last timed step on the library's stream"
    if tm["reduced_launches"] > 0 and tm["full_launches"] > 0 and "9=0" not in a.opt:
        # MOM_OPT_OVERLAP (default): the m = 0 sub-problem's launches run on the handle's second stream CONCURRENTLY with the
        # launch of moments 1..M-1, so in the timed steps the event pairs around either launch span both (and a rocprofv3
        # kernel trace shows the same overlapping intervals).  Per-kernel durations -- the denominator of `roofline` -- are
        # therefore taken from three further steps with the overlap switched off, directly after the timed region; `value`
        # and `ms_per_step` are the overlapped, timed steps.  profiles/ holds the kernel trace of the same serialized form
        # (bench.py --opt 9=0).
        h.set_option(rtamd._lib.MOM_OPT_OVERLAP, 0)
        acc = None
        for _ in range(3):
            h.rt_run()
            t = h.timers()
            acc = t if acc is None else {k: acc[k] + t[k] for k in t}
        h.set_option(rtamd._lib.MOM_OPT_OVERLAP, 1)
        tm_ov = tm
        tm = {k: (acc[k] / 3 if k.endswith("_ms") else acc[k] // 3) for k in acc}
        kernel_timing = ("HIP events of 3 steps with MOM_OPT_OVERLAP = 0 run directly after the timed region (in the timed steps the "
                         f"two problem sizes run concurrently on two streams: layer sweep of the last timed step {tm_ov['layers_ms']:.2f} ms "
                         f"against {tm['layers_ms']:.2f} ms serialized)")
    out = None
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
            prof = json.loads(tf.read_text()).get(workload, {})
        if workload == "C1":
            # N = 4: one spectral point per lane, operators in registers (mom_small.hip).  HBM moves only the per-point
            # inputs and outputs -- algorithmic bytes per point = (tau, varpi, tau_sum + K weights) x Nz x 8
            # + 3 x nVza x nStokes x 8 + 2 x nStokes x 8 -- which at the measured rate is < 1 % of the HBM roofline: the
            # bound of this regime is the FP64 vector-FMA pipe (the kernel issues no MFMA: a 4 x 4 operator would use 1/16
            # of a 16 x 16 tile), whose peak on MI355X equals the FP64 MFMA peak (78.6 TFLOP/s).
            bytes_pt = (3 + scene.K) * scene.Nz * 8 + 3 * len(scene.node) * scene.nStokes * 8 + 2 * scene.nStokes * 8
            gbs = bytes_pt * S_loc / (tm["layers_ms"] * 1e-3) / 1e9
            ach = f_pm * M * S_loc / (tm["layers_ms"] * 1e-3) / 1e12
            # counter traffic of the committed PMC collection next to the algorithmic bytes (the r4 kernel moved
            # 74 x its algorithmic bytes as spill traffic while this block reported 0.006 of the HBM roof from the latter)
            ctr = prof.get("hbm_bytes_per_launch")
            alg_launch = bytes_pt * S_loc
            launch_s = tm["layers_ms"] * 1e-3 / max(tm["layer_launches"], 1)
            roof = {"bound": "fp64-valu", "achieved": ach, "peak": PEAK_FP64_MFMA_TFLOPS, "unit": "TFLOP/s",
                    "frac": ach / PEAK_FP64_MFMA_TFLOPS, "traffic": ctr,
                    "kernel": "momsm::k_sweep<4, true> + k_sum (FP64 vector FMA, one (point, moment) per lane, no MFMA issued)",
                    "avg_launch_ms": tm["layers_ms"] / max(tm["layer_launches"], 1), "launches_per_step": tm["layer_launches"],
                    "algorithmic_flop_per_avg_launch": f_pm * M * S_loc,
                    "pmc_source": ("profiles/traffic.json: " + prof.get("source", prof.get("round", "?"))) if prof else None,
                    "hbm": {"algorithmic_bytes_per_point": bytes_pt, "algorithmic_bytes_per_launch": alg_launch,
                            "achieved_GBps": gbs, "frac_of_8TBps": gbs / PEAK_HBM_GBS,
                            "counter_bytes_per_launch": ctr,
                            "counter_over_algorithmic": (ctr / alg_launch) if ctr else None,
                            "counter_GBps_at_this_run": (ctr / launch_s / 1e9) if ctr else None,
                            "counter_frac_of_8TBps": (ctr / launch_s / 1e9 / PEAK_HBM_GBS) if ctr else None}}
        else:
            achieved = flop_dom / (tm["full_layers_ms"] * 1e-3) / 1e12
            # counter traffic of the committed PMC pass, brought to THIS run's size: the dominant launch moves the composite
            # state (and, N > 64, the slabs) of its own units, so its bytes are proportional to the spectral points of the
            # launch (the S = 512 figure was printed next to the duration of an S = 256 launch)
            traffic, traffic_note = prof.get("hbm_bytes_per_launch"), None
            if traffic and prof.get("points") and prof["points"] != S_loc:
                traffic = traffic * S_loc / prof["points"]
                traffic_note = (f"scaled by {S_loc} / {prof['points']} from the PMC pass at {prof['points']} points "
                                f"({prof['hbm_bytes_per_launch']:.4g} B per launch there)")
            roof = {"bound": "mfma", "achieved": achieved, "peak": PEAK_FP64_MFMA_TFLOPS, "unit": "TFLOP/s",
                    "frac": achieved / PEAK_FP64_MFMA_TFLOPS, "traffic": traffic, "traffic_note": traffic_note,
                    "kernel": prof.get("kernel", "full-problem layer kernel"),
                    "avg_launch_ms": tm["full_layers_ms"] / max(tm["full_launches"], 1),
                    "launches_per_step": tm["full_launches"], "moments_per_launch": m_dom,
                    "algorithmic_flop_per_avg_launch": flop_dom / max(tm["full_launches"], 1),
                    "all_layer_kernels_achieved": whole, "all_layer_kernels_frac": whole / PEAK_FP64_MFMA_TFLOPS,
                    # `achieved` counts the reference's op list (full 2N^3 inverses); the hardware-side figure is the
                    # MFMA-busy share of SIMD cycles from the PMC pass (profiles/), not recomputed here
                    "hw_mfma_busy_frac": prof.get("mfma_busy_frac"),
                    # `traffic` and `hw_mfma_busy_frac` are NOT measured in this run: they are the PMC figures of the
                    # committed collection named here (rocprofv3 --pmc passes cannot run inside the timed bench)
                    "pmc_source": ("profiles/traffic.json: " + prof.get("source", prof.get("round", "?"))) if prof else None,
                    "traffic_all_launches_of_the_pmc_pass": prof.get("hbm_bytes_all_launches_of_the_pass"),
                    "kernel_timing": kernel_timing}
        names = {"C2": "O2-A band IQU scene", "C3": "OCO-2-style 3-band IQU scene (BASELINE configs[2])"}
        out = {
            "metric": f"spectral points/sec (whole node), {names.get(workload, workload + ' scene')}",
            "value": S_tot / (el / steps),
            "unit": "spectral points/s", "n_gpus": world, "steps": steps, "warmup": warmup,
            "ms_per_step": el / steps * 1e3, "higher_is_better": True, "scaling": "strong" if strong else "weak",
            "vs_baseline": None, "dtype": "f64", "data": "synthetic",
            "config": {"workload": f"{workload}: N={N} ({N // scene.nStokes} streams x {scene.nStokes} Stokes), Nz={scene.Nz}, "
                                   f"M={M}, S={S_loc}/GPU ({S_tot} in total), sum(ndoubl)={int(nd.sum())}, Lambertian surface, "
                                   f"{len(scene.node)} VZA",
                       "sharding": f"spectral axis, {world} x {S_loc} points, one all-gather of R||T" if world > 1 else "none",
                       "collective": collective,
                       "devices": "all ranks on GPU 0 (--share-device)" if a.share_device else "one GPU per rank"},
            "roofline": roof,
            "stages_ms": {k: tm[k] for k in ("layers_ms", "full_layers_ms", "reduced_layers_ms", "surface_ms",
                                             "postprocess_ms", "total_ms")},
            "stages_timing": kernel_timing,
        }
        if world == 1 and with_cpu:
            out["cpu_baseline"] = cpu_baseline(model, workload, budget_s=cpu_budget)
    h.close()
    return out


def c5_leg(a, S, steps, warmup, dev, with_cpu=True, world=1, rank=0, dist=None):
    """BASELINE configs[4]: rt_run(::RRS) on scene_C5 (N = 15, 5 layers, 178 Raman lines), corrected switch position.
    Bound: HBM -- the pair kernels read and write every block of the 4-D inelastic operators once per doubling step /
    interaction.  Algorithmic bytes of the dominant kernel (the doubling pair kernel) per (n1, dn) pair with its source
    point on the grid: ier-+, iet++ in and out + ieJ0+- in and out = (4 N^2 + 4 N) x 8 B.
    world > 1 (weak scaling): a global axis of S x world points, every rank owns S of them and runs the window widened by
    the halo max |i_l1l0| (mom_rrs_set_shard; the halo is recomputed, not exchanged); one all-gather of the five spectra."""
    import torch
    import rtamd
    rt = rtamd.corert
    S_tot = S * world
    m, RS = rtamd.scenes.scene_C5(S=S_tot, architecture=rtamd.MI355X(dev.index))
    m = rt._with_cabannes(RS, m)
    sc_full = rtamd.prepare_scene(m)
    lo, hi, wlo, whi = rtamd.sharding.rrs_window(S_tot, world, rank, RS.i_λ1λ0)
    sc = sc_full if world == 1 else sc_full.spectral_slice(wlo, whi)
    Zr_pp, Zr_mp = rt.raman_z(RS, m)
    h = rt.make_handle(m, S=whi - wlo)
    h.set_option(rtamd._lib.MOM_OPT_STRIP_PAD, 0)
    h.rrs_set(RS.i_λ1λ0, RS.ϖ_λ1λ0, RS.rrs_strict_reference)
    h.rrs_set_shard(S_tot, wlo, lo - wlo, hi - wlo)
    rt.scene_set(h, sc)
    h.scene_set_rrs(np.ascontiguousarray(rt.fscatt_rayleigh(m)[wlo:whi].T), rt._abi_mats(Zr_pp), rt._abi_mats(Zr_mp))
    backend = a.backend if world > 1 else "none"
    group, collective = None, "none"
    per = hi - lo                                  # owned points of this rank (weak scaling: S on every rank)
    assert per == S
    cnt = h.rrs_spectra_count(per)                 # (5 nVza nStokes + 2 nStokes) x per: the seven spectra of the return tuple
    loc = G = Gh = None
    if world > 1:
        collective = {"torch": "torch.distributed all_gather_into_tensor (RCCL) of the packed owned spectra",
                      "rccl": "mom_allgather_rrs_device (RCCL through the C ABI): R, T, ieR, ieT, hdr, bhr_uw, bhr_dw in one all-gather",
                      "gloo": "gloo host all_gather of the packed owned spectra"}[backend]
        backend, group, collective = init_collective(h, backend, rank, world, dist, collective)
        loc = torch.empty(cnt, dtype=torch.float64, device=dev)
        G = torch.empty(world * cnt, dtype=torch.float64, device=dev)
        Gh = torch.empty(world * cnt, dtype=torch.float64) if backend == "gloo" else None

    def step():
        h.rt_run_rrs()
        if world == 1:
            return
        if backend == "rccl":
            h.allgather_rrs_device(per, G.data_ptr())                 # pack + ONE collective on the library's stream
        else:
            h.get_spectra_rrs_device(per, loc.data_ptr())             # device-side pack of the owned slices
            h.sync()
            if backend == "gloo":
                dist.all_gather_into_tensor(Gh, loc.cpu())
            else:
                dist.all_gather_into_tensor(G, loc, group=group)

    def fence():
        h.sync()
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    for _ in range(warmup):
        step()
    fence()
    t0 = time.perf_counter()
    for _ in range(steps):
        step()
    fence()
    el = time.perf_counter() - t0
    if world > 1:
        # a torch / torch-fallback run's default (or fallback) group is NCCL-only: the tensor must live on the device
        t = torch.tensor([el], dtype=torch.float64, device=dev if backend in ("torch", "torch-fallback") else "cpu")
        dist.all_reduce(t, op=dist.ReduceOp.MAX, group=group)
        el = float(t.item())
        got = (Gh if backend == "gloo" else G.cpu()).numpy().reshape(world, cnt)
        # this rank's block of the gathered buffer must be its own owned spectra; every block finite
        res = h.get_RT_rrs()[:4] + h.get_hdr_rrs()
        own = [np.ascontiguousarray(r[..., lo - wlo:hi - wlo].T).reshape(-1) for r in res]   # ABI order: spectral index slowest
        assert np.array_equal(got[rank], np.concatenate(own)) and np.all(np.isfinite(got)), "all-gather returned wrong data"
        if a.dump_spectra and rank == 0:
            np.save(a.dump_spectra, np.stack(rtamd.sharding.unpack_rrs_spectra(got, len(sc.node), sc.nStokes, per)[:5]))
    elif a.dump_spectra:
        np.save(a.dump_spectra, np.stack(h.get_RT_rrs()[:4] + h.get_hdr_rrs()[:1]))
    tk = h.rrs_timers()
    res = h.get_RT_rrs()
    assert np.all(np.isfinite(res[2][..., lo - wlo:hi - wlo])) and np.abs(res[2]).max() > 0
    N, nR = sc.N, RS.n_Raman
    pairs = int(sum(max(0, min(hi, S_tot - int(o)) - max(lo, -int(o))) for o in RS.i_λ1λ0))  # owned n1 with n1 + offset on the grid
    bytes_pair = (4 * N * N + 4 * N) * 8
    ms, nl = tk["dbl_pair"]
    avg = ms / max(nl, 1)
    gbs = pairs * bytes_pair / (avg * 1e-3) / 1e9
    # the interaction pair kernel (interface 11; VERDICT r5: it had no roofline entry): per pair the four composite blocks ieR-+,
    # ieR+-, ieT++, ieT-- in and out, the added layer's ier-+, iet++ in (ier+-, iet-- are derived from them in the corrected
    # position, read as well in the strict one), ieJ0+- of the composite in and out and of the added layer in
    blocks_int = 8 + (4 if RS.rrs_strict_reference else 2)
    bytes_pair_int = (blocks_int * N * N + 6 * N) * 8
    ms_i, nl_i = tk["int_pair"]
    avg_i = ms_i / max(nl_i, 1)
    gbs_i = pairs * bytes_pair_int / (avg_i * 1e-3) / 1e9 if nl_i else 0.0
    out = None
    if rank == 0:
        prof = {}
        tf = ROOT / "profiles" / "traffic.json"
        if tf.exists():
            prof = json.loads(tf.read_text()).get("C5", {})
        out = {"metric": "spectral points/sec (whole node), C5 rotational-Raman scene", "value": S_tot / (el / steps),
               "unit": "spectral points/s", "n_gpus": world, "steps": steps, "warmup": warmup, "ms_per_step": el / steps * 1e3,
               "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "f64", "data": "synthetic",
               "config": {"workload": f"C5: RRS, N={N} (5 streams x 3 Stokes), Nz={sc.Nz}, M={sc.M}, S={S}/GPU ({S_tot} in total), "
                                      f"nRaman={nR}, sum(ndoubl)={int(sc.ndoubl.sum())}, "
                                      f"rrs_strict_reference={int(RS.rrs_strict_reference)}, Lambertian surface, 1 VZA",
                          "sharding": "none" if world == 1 else
                          f"contiguous spectral slices + recomputed halo of max|i_l1l0| = {int(np.abs(RS.i_λ1λ0).max())} points "
                          f"(rank 0 window: {whi - wlo} points)",
                          "collective": collective},
               "roofline": {"bound": "hbm", "achieved": gbs, "peak": PEAK_HBM_GBS, "unit": "GB/s", "frac": gbs / PEAK_HBM_GBS,
                            "traffic": prof.get("hbm_bytes_per_launch"), "kernel": "momr::k_dbl_pair1<FUSE, MODE> (one-tile doubling pair kernel)",
                            "pmc_source": ("profiles/traffic.json: " + prof.get("source", prof.get("round", "?"))) if prof else None,
                            "avg_launch_ms": avg, "launches_per_step": nl, "pairs_per_launch": pairs,
                            "algorithmic_bytes_per_pair": bytes_pair, "algorithmic_bytes_per_avg_launch": pairs * bytes_pair},
               "roofline_int_pair": {"bound": "hbm", "achieved": gbs_i, "peak": PEAK_HBM_GBS, "unit": "GB/s", "frac": gbs_i / PEAK_HBM_GBS,
                                     "kernel": "momr::k_int_pair1<...> (one-tile interaction pair kernel, interface 11)",
                                     "avg_launch_ms": avg_i, "launches_per_step": nl_i, "pairs_per_launch": pairs,
                                     "algorithmic_bytes_per_pair": bytes_pair_int},
               "stages_ms": {"dbl_pair_ms": tk["dbl_pair"][0], "int_pair_ms": tk["int_pair"][0],
                             "ie_elemental_ms": tk["ie_elemental"][0], "total_ms": tk["total"][0]}}
    h.close()
    if with_cpu and rank == 0 and world == 1:
        out["cpu_baseline"] = c5_cpu_baseline(S)
    return out


def c5_cpu_baseline(S, budget_s=20.0):
    """The C port of the Raman restatement (oracle/momref.c ora_rt_run_rrs, OpenMP over points and (n1, dn) pairs) TIMED on
    all usable host cores on the same C5 scene with ALL its Raman lines, on an owned window of n1 in the middle of the grid
    sized to the time budget (a 64-point pilot window gives the rate; the elastic operators cover all S points in either
    run, as in the full problem).  value = owned points / wall time of the window run: nothing is extrapolated over lines."""
    sys.path.insert(0, str(ROOT / "tests"))
    import helpers
    import rtamd
    from oracle import cref, momref as mr, rrsref as rr
    m, RS = rtamd.scenes.scene_C5(S=S)
    scene = helpers.oracle_scene(m)
    scene.varpi_cabannes = RS.ϖ_Cabannes
    g = mr.get_greek_rayleigh(0.75)
    p = cref.pack_scene(scene)
    cores = cref.effective_cores()
    offs = np.asarray(RS.i_λ1λ0, dtype=np.int64)

    def run(lo, hi):
        ora = rr.RRSInputs(offs, RS.ϖ_λ1λ0, g, rrs_strict_reference=bool(RS.rrs_strict_reference), owned=(lo, hi))
        t0 = time.perf_counter()
        out = cref.rt_run_rrs(scene, ora, nthreads=cores, p=p)
        assert out[4] == 0 and np.all(np.isfinite(out[2]))
        return time.perf_counter() - t0

    mid = S // 2
    w1, w2 = 1, min(256, S)                             # two pilot windows: t = a (elastic part over all S points) + b W
    t1, t2 = run(mid, mid + w1), run(mid, mid + w2)
    b = max((t2 - t1) / max(w2 - w1, 1), 1e-6)
    a_fix = max(t1 - b * w1, 0.0)
    W = int(min(S, max(w2, 0.6 * (budget_s - a_fix) / b)))   # 0.6: the pair arrays leave the caches as W grows
    lo = max(0, mid - W // 2)
    t_w = run(lo, lo + W)
    return {"value": W / t_w, "unit": "spectral points/s", "cores": cores, "kind": "port",
            "sample": f"C port of the Raman restatement (oracle/momref.c ora_rt_run_rrs), {cores} OpenMP threads, the same C5 scene "
                      f"with all {RS.n_Raman} Raman lines, rrs_strict_reference={int(RS.rrs_strict_reference)}: owned window of {W} "
                      f"of {S} points [{lo}, {lo + W}) TIMED at {t_w:.1f} s, of which ~{a_fix:.1f} s are the elastic operators of all "
                      f"{S} points (pilot windows of {w1} / {w2} points: {t1:.1f} / {t2:.1f} s); the reference itself is Julia and "
                      f"not runnable here"}


def f32_leg(dev, S=20_000, n_streams=8, nd=5, steps=3):
    """The reference's own GPU micro-benchmark shape (test/gpu_tests/gpu_cpu_tests.jl:21-43, gpu_batched_interaction.jl:20-22:
    Float32, n = 32, nSpec = 20 000, ndoubl = 5): one layer's elemental + 5 doublings on a Float32 handle (dtype = 1),
    IQUV with 8 streams = operators of edge 32, against the FP32 matrix roof."""
    import torch
    import rtamd
    rt = rtamd.corert
    # 8 streams = 4 Gauss nodes (l_trunc 7) + 3 views + the Sun at 50 deg (at the default 60 deg it merges with the third view
    # under rt_set_streams' `unique`: 7 streams, N = 28 -- what this leg ran on up to round 4)
    m = rtamd.scenes.make_scene(4, 7, 1, S, sza=50.0, aerosol_total=0.0, absorption=False, architecture=rtamd.MI355X(dev.index))
    sc = rtamd.prepare_scene(m)
    assert sc.N == 4 * n_streams, (sc.N, n_streams)
    sc.ndoubl[:] = nd   # the micro-benchmark's fixed doubling count (the scene's own would be larger)
    h = rt.make_handle(m, float_type="Float32")
    h.scene_set(sc.Nz, sc.K, sc.M, sc.tau, sc.varpi, sc.zw, sc.Zpp, sc.Zmp, sc.ndoubl, sc.iface, sc.tau_sum,
                sc.albedo, sc.node, sc.cos_mphi, sc.sin_mphi)
    h.rt_run()
    h.sync()
    best = 1e30
    for _ in range(steps):
        h.rt_run()
        h.sync()
        best = min(best, h.timers()["layers_ms"])
    N, M = sc.N, sc.M
    flop = M * S * (nd * (12 * N ** 3 + 8 * N ** 2) + N * N * 15)
    ach = flop / (best * 1e-3) / 1e12
    h.close()
    # the BASELINE scene itself (C2: IQU, N = 60, 40 layers, M = 3) on a Float32 handle: strip-chained Float32 images (4-wave
    # builds, two workgroups per CU), moment 0 on the (I,Q) sub-scene -- whole runs incl. surface and post-processing
    c2 = {}
    try:
        S2 = 4096
        m2 = rtamd.scenes.scene_C2(S=S2, architecture=rtamd.MI355X(dev.index))
        sc2 = rtamd.prepare_scene(m2)
        t = {}
        for ft in ("Float64", "Float32"):
            with rt.make_handle(m2, float_type=ft) as h2:
                rt.scene_set(h2, sc2)
                h2.rt_run(); h2.sync()
                tb = 1e30
                for _ in range(2):
                    t0 = time.perf_counter(); h2.rt_run(); h2.sync(); tb = min(tb, time.perf_counter() - t0)
                t[ft] = tb
        c2 = {"workload": f"C2 scene at S = {S2} (N = {sc2.N}, Nz = {sc2.Nz}, M = {sc2.M})", "float64_ms": t["Float64"] * 1e3,
              "float32_ms": t["Float32"] * 1e3, "float32_points_per_s": S2 / t["Float32"], "speedup_over_float64": t["Float64"] / t["Float32"]}
    except Exception as e:
        c2 = {"error": repr(e)}
    return {"metric": "Float32 doubling, reference GPU micro-benchmark shape", "value": S * M / (best * 1e-3), "c2_scene_float32": c2,
            "unit": "(spectral point, moment) units/s", "dtype": "f32", "layers_ms": best,
            "config": {"workload": f"N={N} ({n_streams} streams x 4 Stokes), S={S}, one layer, ndoubl={nd}, M={M}, dtype=1 (Float32)"},
            "roofline": {"bound": "mfma", "achieved": ach, "peak": PEAK_FP32_MFMA_TFLOPS, "unit": "TFLOP/s",
                         "frac": ach / PEAK_FP32_MFMA_TFLOPS, "kernel": "momwf::k_wsweep<2, 7> (wave per spectral point, Float32 build of mom_wave.hip)",
                         "algorithmic_flop_per_avg_launch": flop}}


def dual_cpu_baseline(P, budget_s=20.0):
    """The Dual run of the numpy oracle (oracle/dualref.py: complex-step run of the restatement, one run per partial) on a
    bounded sample of the same shape: 2 spectral points, the value run + ONE partial timed, scaled to P partials."""
    import rtamd
    sys.path.insert(0, str(ROOT / "tests"))
    import helpers
    from oracle import dualref as dr
    m = rtamd.scenes.scene_C2(S=2)
    sc = helpers.oracle_scene(m)
    L = dr.layer_inputs(sc)
    p = dr.Partial(dtau=L.tau.copy(), dvarpi=0.1 * L.varpi, dzw=L.zw.copy(), dalbedo=1.0)
    t0 = time.perf_counter(); dr._run(sc, L, None); tv = time.perf_counter() - t0
    t0 = time.perf_counter(); dr._run(sc, L, p); tp = time.perf_counter() - t0
    return {"value": 2.0 / (tv + P * tp), "unit": "spectral points/s", "cores": os.cpu_count(), "kind": "port",
            "sample": f"2 spectral points of the same scene: value run {tv:.2f} s + one complex-step partial {tp:.2f} s on numpy (threaded BLAS), "
                      f"scaled to {P} partials"}


def dual_leg(dev, S=2000, P=3, with_cpu=True):
    """rt_run on ForwardDiff.Dual numbers (mom_rt_run_dual) on the BASELINE scene's shape (C2: IQU, N = 60, 40 layers, M = 3) with P
    partials, next to the value run of the same scene: whole runs incl. surface and post-processing.  The Dual run streams its
    operators (values + partials) through HBM in batched MFMA products (csrc/mom_dual.hip); its model is (1 + 2 P) value runs."""
    import rtamd
    rt = rtamd.corert
    m = rtamd.scenes.scene_C2(S=S, architecture=rtamd.MI355X(dev.index))
    sc = rtamd.prepare_scene(m)
    L = rt.construct_layer_inputs(m)
    rng = np.random.default_rng(0)
    parts = [rtamd.ScenePartial(dτ=L.τ * rng.uniform(-1, 1, L.τ.shape), dϖ=0.1 * L.ϖ * rng.uniform(-1, 1, L.ϖ.shape),
                                dzw=L.zw * rng.uniform(-1, 1, L.zw.shape), dalbedo=1.0) for _ in range(P)]
    with rt.make_handle(m) as h:
        rt.scene_set(h, sc)
        h.rt_run(); h.sync()
        t0 = time.perf_counter(); h.rt_run(); h.sync(); tv = time.perf_counter() - t0
        rt.scene_set_partials(h, sc, parts)
        h.rt_run_dual(); h.sync()
        t0 = time.perf_counter(); h.rt_run_dual(); h.sync(); td = time.perf_counter() - t0
        dR, _ = h.get_RT_partials()
    N, Nz, M = sc.N, sc.Nz, sc.M
    nd_sum = int(np.sum(sc.ndoubl))
    # products of the tangent-linear sweep: 5 value + 10 P per doubling step, 10 value + 20 P per interaction (interface 11)
    flop = 2.0 * N ** 3 * M * S * (nd_sum * (5 + 10 * P) + Nz * (10 + 20 * P))
    return {"metric": "rt_run on Dual numbers (values + P partials), C2 operator shape", "value": S / td, "unit": "spectral points/s",
            "dtype": "f64", "ms_per_run": td * 1e3, "value_run_ms": tv * 1e3, "ratio_to_value_run": td / tv, "ideal_ratio": 1 + 2 * P,
            "config": {"workload": f"C2 shape: N = {N}, Nz = {Nz}, M = {M}, S = {S}, P = {P} partials (tau, varpi, phase weights, albedo)"},
            "finite": bool(np.isfinite(dR).all()), "cpu_baseline": dual_cpu_baseline(P) if with_cpu else None,
            "roofline": {"bound": "mfma", "achieved": flop / td / 1e12, "peak": PEAK_FP64_MFMA_TFLOPS, "unit": "TFLOP/s",
                         "frac": flop / td / 1e12 / PEAK_FP64_MFMA_TFLOPS, "traffic": None,
                         "note": "product flops of the tangent-linear sweep (2 N^3 per term-product) over the whole run; SQ counters: "
                                 "matrix pipe 43-49 % busy in momd::k_dgemm, HBM 1.5-2.5 TB/s (profiles/r06_dual_ab.txt)"}}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=3)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--workload", choices=["C2", "C1", "C3", "C4", "C5"], default="C2")
    ap.add_argument("--scaling", choices=["weak", "strong"], default="weak",
                    help="strong: the workload's spectral axis (--points or its default size) is split over the ranks")
    ap.add_argument("--points", type=int, default=0, help="spectral points per GPU (strong scaling: in total); default: the workload's size")
    ap.add_argument("--backend", choices=["rccl", "torch", "gloo"], default="rccl")
    ap.add_argument("--share-device", action="store_true", help="all ranks on GPU 0 (needs --backend gloo)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-voigt", action="store_true")
    ap.add_argument("--no-extras", action="store_true", help="skip extra.workloads / extra.f32 of the default run")
    ap.add_argument("--dump-spectra", default="", metavar="FILE.npy",
                    help="rank 0 saves the gathered spectra of the last step (parity of an N-rank run against the 1-rank run)")
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

    dev_index = 0 if (a.share_device or world == 1) else local
    torch.cuda.set_device(dev_index)
    dev = torch.device("cuda", dev_index)
    dist = None
    if world > 1:
        import torch.distributed as dist
        # The JSON line of rank 0 must be the ONLY thing on stdout: gloo's transport logs "[Gloo] Rank r is connected to ..." on
        # the C-level stdout of every rank while a group connects.  Ranks > 0 send their stdout to stderr for good, rank 0
        # while process groups are being set up (stdout_to_stderr).
        if rank != 0:
            sys.stdout.flush()
            os.dup2(2, 1)
        with stdout_to_stderr():
            if a.backend == "torch":
                dist.init_process_group("nccl", device_id=dev)  # "nccl" is RCCL on ROCm
            else:
                dist.init_process_group("gloo")

    S_loc = a.points or DEFAULT_POINTS[a.workload]
    if a.workload == "C5":
        out = c5_leg(a, S_loc, a.steps, a.warmup, dev, with_cpu=not a.no_cpu_baseline, world=world, rank=rank, dist=dist)
    else:
        out = elastic_leg(a, a.workload, S_loc, a.steps, a.warmup, world, rank, dev, dist, with_cpu=not a.no_cpu_baseline)
    if rank == 0:
        default_run = a.workload == "C2" and world == 1 and not a.points
        if world == 1 and a.workload == "C2":
            extra = {}
            if not a.no_voigt:
                try:
                    extra["voigt"] = voigt_leg()
                except Exception as e:  # the headline number must not depend on the second kernel's leg
                    extra["voigt"] = {"error": repr(e)}
            if default_run and not a.no_extras:
                legs = {}
                for wl, pts, st in (("C1", DEFAULT_POINTS["C1"], 3), ("C4", 256, 1)):
                    try:
                        legs[wl] = elastic_leg(a, wl, pts, st, 1, 1, 0, dev, None, with_cpu=not a.no_cpu_baseline, cpu_budget=6.0)
                    except Exception as e:
                        legs[wl] = {"error": repr(e)}
                try:
                    legs["C5"] = c5_leg(a, DEFAULT_POINTS["C5"], 2, 1, dev, with_cpu=not a.no_cpu_baseline)
                except Exception as e:
                    legs["C5"] = {"error": repr(e)}
                extra["workloads"] = legs
                try:
                    extra["f32"] = f32_leg(dev)
                except Exception as e:
                    extra["f32"] = {"error": repr(e)}
                try:
                    extra["dual"] = dual_leg(dev, with_cpu=not a.no_cpu_baseline)
                except Exception as e:
                    extra["dual"] = {"error": repr(e)}
            if extra:
                out["extra"] = extra
        print(json.dumps(out), flush=True)
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
