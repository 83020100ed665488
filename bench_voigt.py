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


def profile_workload(n_lines=300, dnu=0.015, wing=40.0, Nz=40):
    """The reference's operating point of compute_absorption_profile! (atmo_prof.jl:427-449): the O2 A-band window
    12 903 ... 13 245 cm^-1 at 0.015 cm^-1 (DefaultParameters.yaml: 22 801 grid points), wing cut-off 40 cm^-1
    (parameters_from_yaml.jl), 40 layers (log-spaced pressures 0.1 ... 1000 hPa), a seeded O2-A-like list of 300 lines
    (SURVEY section 8d).  Evaluations = sum over layers and lines of the window width; the |x| + y < 8 core is counted per
    layer from the host-side prefactors."""
    import rtamd
    ab = rtamd.absorption
    tab = ab.synthetic_o2a_lines(n_lines)
    grid = np.arange(12903.0, 13245.0 + 0.5 * dnu, dnu)
    p_full = np.exp(np.linspace(np.log(0.1), np.log(1000.0), Nz))
    T = 216.0 + (288.0 - 216.0) * (p_full / 1000.0) ** 0.19
    vcd = np.gradient(p_full) * 2.1e22
    evals = core = 0
    for iz in range(Nz):
        pf = ab.line_prefactors(tab, grid, p_full[iz], T[iz], vmr=0.21, wing_cutoff=wing)
        evals += int(np.maximum(pf.ind_stop - pf.ind_start + 1, 0).sum())
        half = np.maximum(8.0 - pf.y, 0.0) * pf.γ_d / 0.8325546111577
        lo = np.searchsorted(grid, pf.ν - half, side="right")
        hi = np.searchsorted(grid, pf.ν + half, side="left")
        core += int(np.sum(np.clip(np.minimum(hi, pf.ind_stop) - np.maximum(lo, pf.ind_start - 1), 0, None)))
    return tab, grid, p_full, T, vcd, evals, core


def run_profile(repeats=5):
    """extra.voigt.operating_point: the f1 product path -- resident line table, device-side prefactors, all 40 layers in two
    launches (mom_voigt_tau_abs_profile) accumulating into the resident tau_abs -- next to the same work issued layer by
    layer (mom_voigt_tau_abs_layer: two host round trips per layer, 90 workgroups per launch)."""
    import rtamd
    ab = rtamd.absorption
    tab, grid, p_full, T, vcd, evals, core = profile_workload()
    flops = core * FLOP_CORE + (evals - core) * FLOP_WING
    m = rtamd.scenes.make_scene(1, 3, len(p_full), grid.size)
    best, wall, wall_layers = 1e30, 1e30, 1e30
    with rtamd.corert.make_handle(m) as h:
        for _ in range(repeats):
            t0 = time.perf_counter()
            ms = ab.compute_absorption_profile(h, tab, grid, p_full, T, vcd, 0.21, wing_cutoff=40.0, model_vmr=0.21, device_prefactors=True)
            wall = min(wall, time.perf_counter() - t0)
            best = min(best, ms)
        tau = h.absorption_get()
        for _ in range(2):
            t0 = time.perf_counter()
            ab.compute_absorption_profile(h, tab, grid, p_full, T, vcd, 0.21, wing_cutoff=40.0, model_vmr=0.21, device_prefactors=True,
                                          layer_by_layer=True)
            wall_layers = min(wall_layers, time.perf_counter() - t0)
        assert np.array_equal(tau, h.absorption_get()) and np.all(np.isfinite(tau)) and tau.max() > 0
    ach = flops / (best * 1e-3) / 1e12
    return {"workload": f"O2 A-band 12903-13245 cm^-1 at 0.015 cm^-1 ({grid.size} points), wing cut-off 40 cm^-1, {len(p_full)} layers, "
                        f"{len(tab.Sᵢ)} lines: compute_absorption_profile! through mom_voigt_tau_abs_profile (device-side prefactors)",
            "value": evals / (best * 1e-3), "unit": "evaluations/s", "kernels": "k_line_prefactors_profile + k_voigt_profile",
            "kernels_ms": best, "evaluations": evals, "weideman32_fraction": core / evals,
            "call_wall_ms": wall * 1e3, "layer_by_layer_wall_ms": wall_layers * 1e3,
            "roofline": {"bound": "fp64-valu", "achieved": ach, "peak": PEAK_FP64_VALU_TFLOPS, "unit": "TFLOP/s",
                         "frac": ach / PEAK_FP64_VALU_TFLOPS, "flop_model": {"weideman32": FLOP_CORE, "humlicek2": FLOP_WING}}}


if __name__ == "__main__":
    out = run()
    out["operating_point"] = run_profile()
    print(json.dumps(out))
