"""Host side of the Voigt line-by-line path (src/Absorption of the reference).

The per-line prefactors are cheap O(nLines) host work and stay on the host exactly as in
compute_absorption_cross_section.jl:73-107; the O(nLines x window) line-shape sum -- the
reference's one-kernel-launch-per-line hot loop (:118-124) -- runs as ONE launch in
libmomcore.so (csrc/voigt.hip).
"""
from __future__ import annotations

from dataclasses import dataclass
from typing import Callable, Optional

import numpy as np

from . import _lib

# src/Absorption/constants/constants.jl:7-17
c2 = 1.4387769
cMassMol = 1.66053873e-27
cLn2 = 0.6931471805599
cSqrt2Ln2 = 1.1774100225
cc_ = 2.99792458e8
cBolts_ = 1.3806503e-23
p_ref = 1013.25
t_ref = 296.0


@dataclass
class HitranTable:
    """The columns of read_hitran's table (read_hitran.jl:14-68) that the line shape needs."""
    νᵢ: np.ndarray
    Sᵢ: np.ndarray
    γ_air: np.ndarray
    γ_self: np.ndarray
    E_lower: np.ndarray  # E″ ; -1 means "no temperature correction"
    n_air: np.ndarray
    δ_air: np.ndarray
    mol_weight: np.ndarray  # g/mol per line (mol_weight(mol, iso), constants/mol_weights.jl)


_HITRAN_FIELDS = [("mol", 2, int), ("iso", 1, int), ("νᵢ", 12, float), ("Sᵢ", 10, float), ("Aᵢ", 10, float),
                  ("γ_air", 5, float), ("γ_self", 5, float), ("E_lower", 10, float), ("n_air", 4, float),
                  ("δ_air", 8, float), ("global_upper_quanta", 15, str), ("global_lower_quanta", 15, str),
                  ("local_upper_quanta", 15, str), ("local_lower_quanta", 15, str), ("ierr", 6, str), ("iref", 12, str),
                  ("line_mixing_flag", 1, str), ("g_upper", 7, float), ("g_lower", 7, float)]


def read_hitran(filepath, mol: int = -1, iso: int = -1, ν_min: float = 0.0, ν_max: float = float("inf"),
                min_strength: float = 0.0) -> dict:
    """read_hitran (read_hitran.jl:14-68): parse a fixed-width HITRAN .par file into columns,
    keeping rows that match molecule / isotopologue / wavenumber range / minimum strength
    (-1 = any).  Unparseable numeric fields become 0 like the reference's `something(tryparse, 0)`."""
    cols = {name: [] for name, _, _ in _HITRAN_FIELDS}
    with open(filepath, "r") as fh:
        for ln in fh:
            ln = ln.rstrip("\n")
            vals, pos = [], 0
            for name, width, typ in _HITRAN_FIELDS:
                txt = ln[pos:pos + width]
                pos += width
                if typ is str:
                    vals.append(txt)
                else:
                    try:
                        vals.append(typ(txt))
                    except ValueError:
                        vals.append(typ(0))
            if (vals[0] == mol or mol == -1) and (vals[1] == iso or iso == -1) and (ν_min <= vals[2] <= ν_max) \
                    and vals[3] >= min_strength:
                for (name, _, _), v in zip(_HITRAN_FIELDS, vals):
                    cols[name].append(v)
    if not cols["mol"]:
        raise ValueError("No matching records found in the HITRAN file")
    return {k: (np.array(v) if not isinstance(v[0], str) else v) for k, v in cols.items()}


def linear_rotor_qratio(T: float) -> float:
    """Q(T_ref)/Q(T) for a rigid linear rotor (stand-in for the TIPS-2017 spline of qoft!,
    compute_absorption_cross_section.jl:197-214, whose NetCDF tables are not shipped here)."""
    return t_ref / T


@dataclass
class LinePrefactors:
    ν: np.ndarray
    γ_d: np.ndarray
    y: np.ndarray
    S: np.ndarray
    ind_start: np.ndarray  # 1-based inclusive
    ind_stop: np.ndarray


def line_prefactors(h: HitranTable, grid: np.ndarray, pressure: float, temperature: float, vmr: float = 0.0,
                    wing_cutoff: float = 40.0, qratio: Optional[Callable[[float], float]] = None) -> LinePrefactors:
    """compute_absorption_cross_section.jl:54-107: selection of lines inside the padded grid,
    pressure shift, Lorentz and Doppler half widths, y, temperature-corrected strength and the
    index window each line touches (linear interpolation of grid -> index, clamped, rounded
    half-to-even like Julia's `round`)."""
    grid = np.asarray(grid, dtype=np.float64)
    qratio = qratio or linear_rotor_qratio
    keep = (grid.min() - wing_cutoff < h.νᵢ) & (h.νᵢ < grid.max() + wing_cutoff)
    ν0, S0 = h.νᵢ[keep], h.Sᵢ[keep]
    ν = ν0 + pressure / p_ref * h.δ_air[keep]
    γ_l = (h.γ_air[keep] * (1 - vmr) * pressure / p_ref + h.γ_self[keep] * vmr * pressure / p_ref) * \
          (t_ref / temperature) ** h.n_air[keep]
    γ_d = (cSqrt2Ln2 / cc_) * np.sqrt(cBolts_ / cMassMol) * np.sqrt(temperature) * ν0 / np.sqrt(h.mol_weight[keep])
    y = np.sqrt(cLn2) * γ_l / γ_d
    E = h.E_lower[keep]
    corr = qratio(temperature) * np.exp(c2 * E * (1 / t_ref - 1 / temperature)) * \
           (1 - np.exp(-c2 * ν0 / temperature)) / (1 - np.exp(-c2 * ν0 / t_ref))
    S = np.where(E != -1, S0 * corr, S0)
    idx = np.arange(1, grid.size + 1, dtype=np.float64)
    if grid.size > 1:
        i0 = np.rint(np.interp(ν - wing_cutoff, grid, idx)).astype(np.int32)
        i1 = np.rint(np.interp(ν + wing_cutoff, grid, idx)).astype(np.int32)
    else:
        i0 = np.ones(ν.size, dtype=np.int32)
        i1 = np.ones(ν.size, dtype=np.int32)
    return LinePrefactors(ν, γ_d, y, S, i0, i1)


def compute_absorption_cross_section(h: HitranTable, grid, pressure: float, temperature: float, vmr: float = 0.0,
                                     wing_cutoff: float = 40.0, qratio=None, device: int = 0) -> np.ndarray:
    """compute_absorption_cross_section(model::HitranModel, grid, p, T) with Voigt broadening and
    the HumlicekWeidemann32SD error function (the validated default, parameters_from_yaml.jl:115).
    Returns σ[nGrid] in cm²/molecule, computed on the GPU."""
    pf = line_prefactors(h, grid, pressure, temperature, vmr, wing_cutoff, qratio)
    return _lib.voigt_xsec(pf.ν, pf.γ_d, pf.y, pf.S, pf.ind_start, pf.ind_stop, np.asarray(grid, dtype=np.float64),
                           device=device)


def synthetic_o2a_lines(n_lines: int = 300, ν_lo: float = 12903.0, ν_hi: float = 13245.0, seed: int = 1234) -> HitranTable:
    """Seeded O2-A-like line list of SURVEY section 8d."""
    rng = np.random.default_rng(seed)
    ν = np.sort(rng.uniform(ν_lo, ν_hi, n_lines))
    return HitranTable(νᵢ=ν, Sᵢ=10.0 ** rng.uniform(-27, -23, n_lines), γ_air=rng.uniform(0.03, 0.06, n_lines),
                       γ_self=rng.uniform(0.03, 0.06, n_lines), E_lower=rng.uniform(0.0, 2000.0, n_lines),
                       n_air=np.full(n_lines, 0.7), δ_air=np.full(n_lines, -0.005), mol_weight=np.full(n_lines, 31.98983))
