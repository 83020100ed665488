"""
oracle/absref.py -- TEST INFRASTRUCTURE ONLY (imported by tests/ and tests/golden/make_golden.py, never by the
product).  CPU restatement of the HOST side of the reference's line-by-line absorption path, written independently
of radiativetransfer.jl_amd/absorption.py (no shared code; the only shared thing is the extracted TIPS-2017 /
isotopologue DATA table):

  compute_absorption_cross_section(model::HitranModel, grid, p, T)   src/Absorption/compute_absorption_cross_section.jl:19-130
  qoft!(M, I, T, T_ref, result)                                     :197-214
  get_TT / get_TQ, mol_weight                                       constants/TIPS_2017.jl:15-27, mol_weights.jl:19-26
  DataInterpolations.CubicSpline (third-party, compat "4"; restated from its published constructor/_interpolate)

Parity status: the reference's own goldens for this path (test/test_profiles/Voigt_*.csv, from HAPI) need HITRAN
line lists that are Pkg artifacts downloaded at install time -- not available here: Voigt-vs-HAPI parity UNPINNED.
What is pinned: the partition sums against TIPS-2017's published Q(296 K) values (tests/test_oracle_absorption.py),
read_hitran against the reference's exact-value tests, w(z) against scipy.special.wofz.
"""
from __future__ import annotations

import math
from pathlib import Path

import numpy as np

from . import momref as mr

C2 = 1.4387769            # constants/constants.jl:7-17
C_MASS_MOL = 1.66053873e-27
C_LN2 = 0.6931471805599
C_SQRT2LN2 = 1.1774100225
CC = 2.99792458e8
C_BOLTZ = 1.3806503e-23
P_REF = 1013.25
T_REF = 296.0

_DATA = Path(__file__).resolve().parents[1] / "radiativetransfer.jl_amd" / "data" / "tips_2017_subset.npz"
_tab = None


def tables():
    global _tab
    if _tab is None:
        _tab = np.load(_DATA)
    return _tab


def spline_second_derivatives(u32: np.ndarray, t32: np.ndarray) -> np.ndarray:
    """z of DataInterpolations.CubicSpline: tA z = d with tA = Tridiagonal(h[2:n+1], 2(h[1:n+1] + h[2:n+2]), h[2:n+1]),
    h = [0, diff(t), 0], d[1] = d[n+1] = 0.  Solved here as a DENSE Float32 system (LAPACK sgesv: LU with partial
    pivoting like Julia's `\\` on a Tridiagonal; the matrix is diagonally dominant, so no interchange happens)."""
    u, t = u32.astype(np.float32), t32.astype(np.float32)
    n = t.size - 1
    h = np.zeros(n + 2, dtype=np.float32)
    h[1:n + 1] = t[1:] - t[:-1]
    A = np.zeros((n + 1, n + 1), dtype=np.float32)
    rhs = np.zeros(n + 1, dtype=np.float32)
    for i in range(n + 1):
        A[i, i] = np.float32(2) * (h[i] + h[i + 1])
        if i > 0:
            A[i, i - 1] = h[i]
        if i < n:
            A[i, i + 1] = h[i + 1]
        if 0 < i < n:
            rhs[i] = np.float32(6) * (u[i + 1] - u[i]) / h[i + 1] - np.float32(6) * (u[i] - u[i - 1]) / h[i]
    return np.linalg.solve(A, rhs).astype(np.float32), h


def spline_eval(u32, t32, z, h, x: float) -> float:
    """_interpolate(A::CubicSpline, t): Float32 data and z, Float64 abscissa."""
    idx = int(np.searchsorted(t32, np.float64(x), side="right"))  # searchsortedlast, 1-based
    idx = max(1, min(idx, t32.size - 1))
    i = idx - 1
    x = np.float64(x)
    hi = h[idx]
    term_i = z[i] * (t32[i + 1] - x) ** 3 / (6 * hi) + z[i + 1] * (x - t32[i]) ** 3 / (6 * hi)
    term_c = (u32[i + 1] / hi - z[i + 1] * hi / 6) * (x - t32[i])
    term_d = (u32[i] / hi - z[i] * hi / 6) * (t32[i + 1] - x)
    return float(term_i + term_c + term_d)


def qoft(M: int, I: int, T: float, T_ref: float = T_REF) -> float:
    tab = tables()
    TT, TQ = tab[f"T_{M}_{I}"], tab[f"Q_{M}_{I}"]
    assert TT.min() < T < TT.max(), f"TIPS2017: T ({T}) must be between {TT.min()} K and {TT.max()} K."
    z, h = spline_second_derivatives(TQ, TT)
    return spline_eval(TQ, TT, z, h, T_ref) / spline_eval(TQ, TT, z, h, T)


def mol_weight(M: int, I: int) -> np.float32:
    tab = tables()
    w = tab["mol_weight"][list(tab["molecules"]).index(M), I - 1]
    assert w != -1, "No matching (mol, iso) pair"
    return np.float32(w)


def julia_round(x: float) -> int:
    """Base.round(x) -> nearest, ties to even."""
    return int(np.rint(x))


def line_parameters(hit: dict, grid: np.ndarray, pressure: float, temperature: float, vmr: float, wing_cutoff: float):
    """The host loop of compute_absorption_cross_section.jl:73-116, line by line.  `hit`: read_hitran-style columns
    (mol, iso, νᵢ, Sᵢ, γ_air, γ_self, E_lower, n_air, δ_air).  Returns ν, γ_d, y, S, ind_start, ind_stop (1-based)."""
    grid = np.asarray(grid, dtype=np.float64)
    grid_max, grid_min = grid.max() + wing_cutoff, grid.min() - wing_cutoff
    nG = grid.size
    out = [[] for _ in range(6)]
    temperature = float(temperature)
    for j in range(len(hit["Sᵢ"])):
        nu_j = float(hit["νᵢ"][j])
        if not (grid_min < nu_j < grid_max):
            continue
        nu = nu_j + pressure / P_REF * float(hit["δ_air"][j])
        gamma_l = (float(hit["γ_air"][j]) * (1 - vmr) * pressure / P_REF + float(hit["γ_self"][j]) * vmr * pressure / P_REF) * \
                  (T_REF / temperature) ** float(hit["n_air"][j])
        sq = np.sqrt(mol_weight(int(hit["mol"][j]), int(hit["iso"][j])))  # Float32 sqrt of a Float32
        gamma_d = ((C_SQRT2LN2 / CC) * math.sqrt(C_BOLTZ / C_MASS_MOL) * math.sqrt(temperature) * nu_j / np.float64(sq))
        y = math.sqrt(C_LN2) * gamma_l / gamma_d
        S = float(hit["Sᵢ"][j])
        E = float(hit["E_lower"][j])
        if E != -1:
            rate = qoft(int(hit["mol"][j]), int(hit["iso"][j]), temperature, T_REF)
            S = S * rate * math.exp(C2 * E * (1 / T_REF - 1 / temperature)) * \
                (1 - math.exp(-C2 * nu_j / temperature)) / (1 - math.exp(-C2 * nu_j / T_REF))
        if nG > 1:
            # grid_idx_interp_low / _high (:60-61): LinearInterpolation(grid, 1:n, extrapolation_bc = 1) and (… = n) -- each
            # returns ITS constant on both sides of the grid (a pressure-shifted line whose nu - wing lies beyond the last
            # grid point starts at 1; one whose nu + wing lies before the first grid point stops at n)
            lo = np.interp(nu - wing_cutoff, grid, np.arange(1, nG + 1), left=1, right=1)
            hi = np.interp(nu + wing_cutoff, grid, np.arange(1, nG + 1), left=nG, right=nG)
            i0, i1 = julia_round(lo), julia_round(hi)
        else:
            i0 = i1 = 1
        for lst, v in zip(out, (nu, gamma_d, y, S, i0, i1)):
            lst.append(v)
    f = lambda k: np.array(out[k], dtype=np.float64)
    return f(0), f(1), f(2), f(3), np.array(out[4], dtype=np.int32), np.array(out[5], dtype=np.int32)


def absorption_cross_section(hit: dict, grid, pressure, temperature, vmr=0.0, wing_cutoff=40.0) -> np.ndarray:
    nu, gd, y, S, i0, i1 = line_parameters(hit, grid, pressure, temperature, vmr, wing_cutoff)
    return mr.voigt_xsec(nu, gd, y, S, i0, i1, np.asarray(grid, dtype=np.float64))
