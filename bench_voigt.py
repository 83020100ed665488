#!/usr/bin/env python3
"""Secondary benchmark: the Voigt line-by-line kernel (csrc/voigt.hip).  Not the driver's bench
(that is bench.py); prints one JSON line with line-shape evaluations per second and the FP64-VALU
roofline fraction.  Flop model per evaluation of Re w(z): Weideman-32 branch (|x|+y < 8: 32-term complex
Horner + 2 complex divisions) about 330 flop; Humlicek region-II branch (far wings, the common case) about
70 flop; the mix is counted from the prefactors on the host."""
import json
import sys
import time
from pathlib import Path

import numpy as np

sys.path.insert(0, str(Path(__file__).resolve().parent))
import rtamd  # noqa: E402

FLOP_CORE, FLOP_WING = 330.0, 70.0
PEAK_FP64_VALU_TFLOPS = 78.6  # MI355X FP64 vector spec; v_fma_f64 microbenchmark: 60


def main():
    ab = rtamd.absorption
    tab = ab.synthetic_o2a_lines(5000)
    grid = np.linspace(12903.0, 13245.0, 200_000)
    pf = ab.line_prefactors(tab, grid, 500.0, 250.0, vmr=0.21, wing_cutoff=40.0)
    evals = int(np.sum(np.maximum(pf.ind_stop - pf.ind_start + 1, 0)))
    dnu = grid[1] - grid[0]
    half = np.maximum(8.0 - pf.y, 0.0) * pf.γ_d / 0.8325546111577  # |nu - nu0| below which the Weideman branch runs
    core = int(np.sum(np.minimum(2 * half / dnu, pf.ind_stop - pf.ind_start + 1)))
    flops = core * FLOP_CORE + (evals - core) * FLOP_WING
    best = 1e30
    for _ in range(4):
        rtamd.voigt_xsec(pf.ν, pf.γ_d, pf.y, pf.S, pf.ind_start, pf.ind_stop, grid)
        best = min(best, rtamd._lib.voigt_last_kernel_ms())
    out = {"metric": "Voigt line-shape evaluations/s", "value": evals / (best * 1e-3), "unit": "evaluations/s",
           "kernel_ms": best, "evaluations": evals, "lines": len(pf.ν), "grid_points": len(grid),
           "core_fraction": core / evals,
           "roofline": {"bound": "fp64-valu", "achieved": flops / (best * 1e-3) / 1e12,
                        "peak": PEAK_FP64_VALU_TFLOPS, "unit": "TFLOP/s",
                        "frac": flops / (best * 1e-3) / 1e12 / PEAK_FP64_VALU_TFLOPS}}
    t0 = time.perf_counter()
    sys.path.insert(0, str(Path(__file__).resolve().parent / "tests"))
    from oracle import cref
    n = 40
    cref.voigt_xsec(pf.ν[:n], pf.γ_d[:n], pf.y[:n], pf.S[:n], pf.ind_start[:n], pf.ind_stop[:n], grid)
    dt = time.perf_counter() - t0
    ev = int(np.sum(np.maximum(pf.ind_stop[:n] - pf.ind_start[:n] + 1, 0)))
    out["cpu_baseline"] = {"value": ev / dt, "unit": "evaluations/s", "kind": "port", "sample": f"first {n} lines, OpenMP over grid points"}
    print(json.dumps(out))


if __name__ == "__main__":
    main()
