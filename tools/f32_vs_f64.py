#!/usr/bin/env python3
"""Float32 (dtype = 1) against Float64 handles on the scenes of tools/size_sweep.py: mom_rt_run wall time incl. the result
download, 20 layers, M = 3, two view angles.  usage: python tools/f32_vs_f64.py > profiles/rNN_f32_vs_f64.txt"""
import os
import sys
import time

ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import conftest  # noqa: E402,F401
import rtamd  # noqa: E402

CASES = [(1, 3, 200000), (4, 3, 20000), (4, 7, 8192), (4, 9, 4096), (4, 13, 4096), (4, 15, 4096), (3, 25, 4096), (4, 19, 4096),
         (3, 29, 4096), (4, 21, 4096), (3, 31, 4096), (3, 33, 4096), (4, 23, 4096), (4, 25, 4096)]


def run(m, sc, ft, opts=()):
    with rtamd.corert.make_handle(m, float_type=ft) as h:
        for o, v in opts:
            h.set_option(o, v)
        rtamd.corert.run_scene(h, sc)
        best = 1e9
        for _ in range(3):
            t0 = time.time()
            h.rt_run()
            h.get_RT()
            best = min(best, time.time() - t0)
    return best * 1e3


L = rtamd._lib
print(f"{'nStokes':>7} {'N':>4} {'S':>7} {'Float64 ms':>11} {'Float32 ms':>11} {'f64/f32':>8} {'f32, no m=0 reduction / padding':>32} {'f32, 8-wave strip images':>25}")
for nS, lt, S in CASES:
    m = rtamd.scenes.make_scene(nS, lt, 20, S, vza=(0.0, 30.0), vaz=(0.0, 20.0))
    sc = rtamd.prepare_scene(m)
    t64 = run(m, sc, "Float64")
    t32 = run(m, sc, "Float32")
    t32n = run(m, sc, "Float32", ((L.MOM_OPT_M0_REDUCTION, 0), (L.MOM_OPT_STRIP_PAD, 0)))
    t32w = run(m, sc, "Float32", ((L.MOM_OPT_SMALL_WG, 0),))
    print(f"{nS:7d} {sc.N:4d} {S:7d} {t64:11.1f} {t32:11.1f} {t64 / t32:8.2f} {t32n:32.1f} {t32w:25.1f}", flush=True)
