#!/usr/bin/env python3
"""Throughput of mom_rt_run over the operator edge N (20 layers, M = 3, two view angles): which kernel family serves
which size and what it reaches.  Wall clock of one resident-scene run including the result download; F = the
algorithmic flop count of SURVEY 8d with the FULL N.  usage: python tools/size_sweep.py > profiles/rNN_size_sweep.txt"""
import os
import sys
import time

ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import conftest  # noqa: E402,F401  (registers the package under its import name)
import rtamd  # noqa: E402

CASES = [(1, 3, 200000), (1, 9, 20000), (3, 3, 20000), (4, 3, 20000), (1, 21, 20000), (3, 7, 8192), (4, 5, 8192), (4, 7, 8192),
         (4, 9, 4096), (4, 11, 4096), (4, 13, 4096), (3, 21, 4096), (4, 15, 4096), (3, 25, 4096), (3, 27, 4096), (4, 19, 4096),
         (3, 29, 4096), (4, 21, 4096), (3, 31, 4096), (3, 33, 4096), (4, 23, 4096), (3, 35, 4096), (4, 25, 4096),
         (3, 37, 2048), (3, 49, 2048), (4, 41, 1024), (4, 57, 1024), (4, 65, 512), (4, 85, 512), (4, 121, 256)]

only = {int(x) for x in os.environ.get("MOM_SWEEP_ONLY", "").split(",") if x}  # operator edges to keep (default: all)
print(f"{'nStokes':>7} {'N':>4} {'S':>7} {'ms':>9} {'points/s':>11} {'TFLOP/s':>8} {'of FP64 peak':>12}  kernel path")
for nS, lt, S in CASES:
    m = rtamd.scenes.make_scene(nS, lt, 20, S, vza=(0.0, 30.0), vaz=(0.0, 20.0))
    sc = rtamd.prepare_scene(m)
    N = sc.N
    if only and N not in only:
        continue
    with rtamd.corert.make_handle(m) as h:
        rtamd.corert.run_scene(h, sc)
        t0 = time.time()
        h.rt_run()
        h.get_RT()
        dt = time.time() - t0
    nd = int(sum(sc.ndoubl))
    F = sc.M * (nd * (12 * N**3 + 8 * N**2) + (sc.Nz - 1) * (24 * N**3 + 8 * N**2))
    pad = next((p for p in (36, 40, 44, 52, 56, 60) if N <= p <= N + 4), None) if N > 32 else None
    path = ("lane per point (mom_small)" if N <= 4 else "wave per point (mom_wave)" if N <= 32 else
            f"strip chains{'' if pad == N else f', padded to {pad}'}" if pad else
            "workgroup per unit, LDS operators" if N <= 64 else "workgroup per unit, panel GEMM from the L2 slab")
    print(f"{nS:7d} {N:4d} {S:7d} {dt * 1e3:9.1f} {S / dt:11.0f} {F * S / dt / 1e12:8.2f} {F * S / dt / 78.6e12:12.3f}  {path}", flush=True)
