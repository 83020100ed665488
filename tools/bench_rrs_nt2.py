#!/usr/bin/env python3
"""Timing of the multi-tile RRS kernels: mom_rt_run_rrs on seeded scenes of edge 24, 30, 32 (2 x 2 tiles), 42 (3 x 3) and 56,
60 (4 x 4: scratch-resident operators), 40 Raman offsets, corrected switch position.
usage: python tools/bench_rrs_nt2.py > profiles/rNN_rrs_nt2.txt"""
import os
import sys
import time
from pathlib import Path

import numpy as np

sys.path.insert(0, str(Path(__file__).resolve().parents[1]))
import rtamd  # noqa: E402

rt = rtamd.corert
print(f"{'nStokes':>7} {'N':>3} {'S':>5} {'nRaman':>6} {'sum nd':>6} {'ms/run':>8} {'points/s':>9} {'dbl_pair ms':>11} {'int_pair ms':>11}  pairs/launch  GB/s (alg.) of 8 TB/s")
for nS, lt, vza, S in ((4, 7, (0.0,), 2000), (3, 13, (0.0, 30.0), 2000), (4, 9, (0.0, 30.0), 2000), (3, 21, (0.0, 30.0, 60.0), 500),
                       (4, 21, (0.0, 30.0, 60.0), 500), (3, 33, (0.0, 30.0, 60.0), 500)):
    nR = 40
    m = rtamd.scenes.make_scene(nS, lt, 5, S, seed=3, aerosol_total=0.1, vza=vza, vaz=tuple(20.0 * i for i in range(len(vza))))
    rng = np.random.default_rng(1)
    offs = np.unique(rng.integers(-min(600, S // 2), min(600, S // 2), 2 * nR))[:nR]
    offs = offs[offs != 0]
    RS = rt.RRS(greek_raman=rt.get_greek_rayleigh(0.2), ϖ_Cabannes=0.96, ϖ_λ1λ0=np.full(len(offs), 0.04 / len(offs)), i_λ1λ0=offs,
                rrs_strict_reference=False)
    model = rt._with_cabannes(RS, m)
    sc = rtamd.prepare_scene(model)
    if os.environ.get("MOM_NT_ONLY") and str(sc.N) not in os.environ["MOM_NT_ONLY"].split(","):   # e.g. MOM_NT_ONLY=42,56,60
        continue
    Zr_pp, Zr_mp = rt.raman_z(RS, model)
    with rt.make_handle(model) as h:
        h.set_option(rtamd._lib.MOM_OPT_STRIP_PAD, 0)
        h.rrs_set(RS.i_λ1λ0, RS.ϖ_λ1λ0, False)
        rt.scene_set(h, sc)
        h.scene_set_rrs(np.ascontiguousarray(rt.fscatt_rayleigh(model).T), rt._abi_mats(Zr_pp), rt._abi_mats(Zr_mp))
        h.rt_run_rrs()
        best = 1e30
        for _ in range(3):
            t0 = time.perf_counter()
            h.rt_run_rrs()
            h.get_RT_rrs()
            best = min(best, time.perf_counter() - t0)
        tk = h.rrs_timers()
        assert h.rrs_check_padding() == 0
    N = sc.N
    pairs = int(sum(max(0, min(S, S - int(o)) - max(0, -int(o))) for o in offs))
    ms, nl = tk["dbl_pair"]
    gbs = pairs * (4 * N * N + 4 * N) * 8 / (ms / max(nl, 1) * 1e-3) / 1e9
    print(f"{nS:7d} {N:3d} {S:5d} {len(offs):6d} {int(sc.ndoubl.sum()):6d} {best * 1e3:8.1f} {S / best:9.0f} {ms:11.1f} {tk['int_pair'][0]:11.1f}  "
          f"{pairs:12d}  {gbs:8.1f}  {gbs / 8000:.3f}", flush=True)
