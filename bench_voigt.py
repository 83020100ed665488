#!/usr/bin/env python3
"""The Voigt line-by-line kernel (csrc/voigt.hip) as a benchmark: bench.py reports `run()` as `extra.voigt`; run as
a script it prints the same dict as one JSON line.

Workload: a line-core pass -- 50 000 O2-A-like lines on a 400 000-point grid (0.00086 cm^-1 spacing), wing cut-off
0.3 cm^-1, at 150 hPa / 220 K: every window is 700 points wide and 37 % of all (line, grid point) evaluations fall in
the |x| + y < 8 core (half width (8 - y) gamma_d / sqrt(ln 2) = 0.11 cm^-1 here) where w(z) is the 32-term Weideman
rational; the rest take the Humlicek region-II branch.  (The r1 workload, 40 cm^-1 wings, had 99.7 % far-wing
evaluations and barely exercised the core branch.)

Flop model per evaluation of Re w(z): Weideman-32 branch (32-term complex Horner = 31 x 8 flop, 2 complex
divisions ~ 2 x 28, ~26 for z, the prefactor and the sum) about 330 flop; Humlicek branch about 70 flop; the branch mix
is counted exactly on the host from the prefactors.  Bound: FP64 VALU (no data reuse, negligible HBM bytes)."""
import json
import sys
import time
from pathlib import Path

import numpy as np

sys.path.insert(0, str(Path(__file__).resolve().parent))

FLOP_CORE, FLOP_WING = 330.0, 70.0
PEAK_FP64_VALU_TFLOPS = 78.6  # MI355X FP64 vector spec


def workload(n_lines=50_000, n_grid=400_000, wing=0.3, p=150.0, T=220.0):
    import rtamd
    ab = rtamd.absorption
    tab = ab.synthetic_o2a_lines(n_lines)
    grid = np.linspace(12903.0, 13245.0, n_grid)
    pf = ab.line_prefactors(tab, grid, p, T, vmr=0.21, wing_cutoff=wing)
    width = np.maximum(pf.ind_stop - pf.ind_start + 1, 0)
    evals = int(width.sum())
    # |x| + y < 8 with x = sqrt(ln 2) (nu - nu0) / gamma_d: counted exactly per line on the grid
    half = np.maximum(8.0 - pf.y, 0.0) * pf.γ_d / 0.8325546111577
    lo = np.searchsorted(grid, pf.ν - half, side="right")
    hi = np.searchsorted(grid, pf.ν + half, side="left")
    core = int(np.sum(np.clip(np.minimum(hi, pf.ind_stop) - np.maximum(lo, pf.ind_start - 1), 0, None)))
    return pf, grid, evals, core


def run(repeats=5, cpu_lines=4000):
    import rtamd
    pf, grid, evals, core = workload()
    flops = core * FLOP_CORE + (evals - core) * FLOP_WING
    best = 1e30
    for _ in range(repeats):
        rtamd.voigt_xsec(pf.ν, pf.γ_d, pf.y, pf.S, pf.ind_start, pf.ind_stop, grid)
        best = min(best, rtamd._lib.voigt_last_kernel_ms())
    ach = flops / (best * 1e-3) / 1e12
    out = {"metric": "Voigt line-shape evaluations/s", "value": evals / (best * 1e-3), "unit": "evaluations/s",
           "kernel": "k_voigt", "kernel_ms": best, "evaluations": evals, "lines": int(len(pf.ν)), "grid_points": int(len(grid)),
           "weideman32_fraction": core / evals,
           "roofline": {"bound": "fp64-valu", "achieved": ach, "peak": PEAK_FP64_VALU_TFLOPS, "unit": "TFLOP/s",
                        "frac": ach / PEAK_FP64_VALU_TFLOPS, "flop_model": {"weideman32": FLOP_CORE, "humlicek2": FLOP_WING}}}
    sys.path.insert(0, str(Path(__file__).resolve().parent / "tests"))
    from oracle import cref
    n = cpu_lines
    cores = cref.effective_cores()
    t0 = time.perf_counter()
    cref.voigt_xsec(pf.ν[:n], pf.γ_d[:n], pf.y[:n], pf.S[:n], pf.ind_start[:n], pf.ind_stop[:n], grid)
    dt = time.perf_counter() - t0
    ev = int(np.sum(np.maximum(pf.ind_stop[:n] - pf.ind_start[:n] + 1, 0)))
    out["cpu_baseline"] = {"value": ev / dt, "unit": "evaluations/s", "kind": "port", "cores": cores,
                           "sample": f"first {n} lines of the same workload, C oracle, OpenMP over grid points"}
    return out


if __name__ == "__main__":
    print(json.dumps(run()))
