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
    """The columns of read_hitran's table (read_hitran.jl:14-68, types.jl:24-61) that the line shape needs."""
    mol: np.ndarray      # HITRAN molecule id per line
    iso: np.ndarray      # HITRAN isotopologue id per line
    νᵢ: np.ndarray
    Sᵢ: np.ndarray
    γ_air: np.ndarray
    γ_self: np.ndarray
    E_lower: np.ndarray  # E″ ; -1 means "no temperature correction"
    n_air: np.ndarray
    δ_air: np.ndarray


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


def hitran_table(cols: dict) -> HitranTable:
    """read_hitran's column dict -> HitranTable (make_hitran_model's input, make_model_helpers.jl)."""
    f = lambda k: np.asarray(cols[k], dtype=np.float64)
    return HitranTable(mol=np.asarray(cols["mol"], dtype=np.int64), iso=np.asarray(cols["iso"], dtype=np.int64),
                       νᵢ=f("νᵢ"), Sᵢ=f("Sᵢ"), γ_air=f("γ_air"), γ_self=f("γ_self"), E_lower=f("E_lower"),
                       n_air=f("n_air"), δ_air=f("δ_air"))


# ------------------------------------------------------------------------------------------
# TIPS-2017 partition sums and isotopologue weights (constants/TIPS_2017.jl, mol_weights.jl).
# The tables are the reference's NetCDF files, extracted for every HITRAN molecule they cover (ids 1-49, 157 (molecule,
# isotopologue) pairs; rounds 1-3 shipped ids 1-7) by tools/extract_tips.py into data/tips_2017_subset.npz (Float32 like the originals).
# ------------------------------------------------------------------------------------------

_TIPS = None


def _tips():
    global _TIPS
    if _TIPS is None:
        f = _lib.PKG_DIR / "data" / "tips_2017_subset.npz"
        if not f.exists():
            raise FileNotFoundError(f"{f} missing: run /opt/conda/bin/python3.9 tools/extract_tips.py")
        _TIPS = dict(np.load(f))
    return _TIPS


def mol_weight(mol: int, iso: int) -> np.float32:
    """mol_weight(mol, iso) (mol_weights.jl:23): Float32 g/mol; raises like check_exists for unfilled pairs."""
    t = _tips()
    mols = list(t["molecules"])
    if int(mol) not in mols or not (1 <= int(iso) <= t["mol_weight"].shape[1]):
        raise KeyError(f"No matching (mol, iso) pair ({mol}, {iso}) in the extracted tables (tools/extract_tips.py)")
    w = t["mol_weight"][mols.index(int(mol)), int(iso) - 1]
    if w == -1:
        raise KeyError("No matching (mol, iso) pair")
    return np.float32(w)


def get_TT(mol: int, iso: int) -> np.ndarray:
    return _tips()[f"T_{int(mol)}_{int(iso)}"]


def get_TQ(mol: int, iso: int) -> np.ndarray:
    return _tips()[f"Q_{int(mol)}_{int(iso)}"]


class CubicSpline:
    """DataInterpolations.CubicSpline(u, t) as `qoft!` uses it (compute_absorption_cross_section.jl:208-210; compat
    DataInterpolations 4, third-party, restated from its published source): second derivatives z from the
    tridiagonal system with rows [2(h_i + h_{i+1})] and right-hand side 0 at BOTH ends (first row 2 h_1 z_1 + h_1 z_2 = 0,
    not z_1 = 0), all in the element type of the data -- Float32 for the TIPS tables -- and a Float64 evaluation."""

    def __init__(self, u: np.ndarray, t: np.ndarray):
        FT = np.result_type(u.dtype, t.dtype).type
        u, t = u.astype(FT), t.astype(FT)
        n = len(t) - 1
        h = np.concatenate(([FT(0)], (t[1:] - t[:-1]).astype(FT), [FT(0)])).astype(FT)
        dl = h[1:n + 1].copy()
        dg = (FT(2) * (h[0:n + 1] + h[1:n + 2])).astype(FT)
        du = h[1:n + 1].copy()
        d = np.zeros(n + 1, dtype=FT)
        for i in range(1, n):
            d[i] = FT(6) * (u[i + 1] - u[i]) / h[i + 1] - FT(6) * (u[i] - u[i - 1]) / h[i]
        # LU of the (diagonally dominant: no interchange) tridiagonal matrix and the two substitutions, in FT
        for i in range(n):
            f = FT(dl[i] / dg[i])
            dg[i + 1] = FT(dg[i + 1] - f * du[i])
            d[i + 1] = FT(d[i + 1] - f * d[i])
        z = np.zeros(n + 1, dtype=FT)
        z[n] = FT(d[n] / dg[n])
        for i in range(n - 1, -1, -1):
            z[i] = FT((d[i] - du[i] * z[i + 1]) / dg[i])
        self.u, self.t, self.h, self.z = u, t, h[:n + 1], z

    def __call__(self, x: float) -> float:
        t, u, h, z = self.t, self.u, self.h, self.z
        i = int(np.searchsorted(t, x, side="right")) - 1        # searchsortedlast
        i = max(0, min(i, len(t) - 2))
        x = np.float64(x)
        I = z[i] * (t[i + 1] - x) ** 3 / (6 * h[i + 1]) + z[i + 1] * (x - t[i]) ** 3 / (6 * h[i + 1])
        C = (u[i + 1] / h[i + 1] - z[i + 1] * h[i + 1] / 6) * (x - t[i])
        D = (u[i] / h[i + 1] - z[i] * h[i + 1] / 6) * (t[i + 1] - x)
        return float(I + C + D)


_SPLINES = {}


def qoft(mol: int, iso: int, T: float, T_ref: float = t_ref) -> float:
    """qoft!(M, I, T, T_ref, result) (compute_absorption_cross_section.jl:197-214): Q(T_ref)/Q(T) from the
    TIPS-2017 table of the isotopologue by cubic-spline interpolation."""
    key = (int(mol), int(iso))
    if key not in _SPLINES:
        TT, TQ = get_TT(*key), get_TQ(*key)
        _SPLINES[key] = (CubicSpline(TQ, TT), float(TT.min()), float(TT.max()))
    sp, Tmin, Tmax = _SPLINES[key]
    if not (Tmin < T < Tmax):
        raise AssertionError(f"TIPS2017: T ({T}) must be between {Tmin} K and {Tmax} K.")
    return sp(T_ref) / sp(T)


def linear_rotor_qratio(T: float) -> float:
    """Q(T_ref)/Q(T) of a rigid linear rotor -- NOT what the reference uses (that is `qoft`); kept only as an
    explicit `qratio=` choice for synthetic line lists of molecules outside the extracted tables."""
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
                    wing_cutoff: float = 40.0, qratio: Optional[Callable[[float], float]] = None,
                    mol_weights: Optional[dict] = None) -> LinePrefactors:
    """compute_absorption_cross_section.jl:54-107: selection of lines inside the padded grid,
    pressure shift, Lorentz and Doppler half widths (Float32 square root of the Float32 isotopologue weight, as
    `sqrt(mol_weight(mol, iso))` evaluates), y, the TIPS-2017 temperature correction of the strength (`qoft!`) and the
    index window each line touches (linear interpolation of grid -> index with the reference's constant fill values outside
    the grid -- 1 for the start, n for the stop, on either side -- rounded half-to-even like Julia's `round`).  `qratio` overrides the partition-sum ratio (a callable of T); default = the reference's qoft.
    `mol_weights` {(mol, iso): g/mol} supplies isotopologue weights for (molecule, isotopologue) pairs the bundled tables do
    not hold (they cover HITRAN molecules 1-49 as the reference's NetCDF files do; tools/extract_tips.py)."""
    grid = np.asarray(grid, dtype=np.float64)
    temperature = float(temperature)
    keep = (grid.min() - wing_cutoff < h.νᵢ) & (h.νᵢ < grid.max() + wing_cutoff)
    ν0, S0 = h.νᵢ[keep], h.Sᵢ[keep]
    mol, iso = h.mol[keep], h.iso[keep]
    ν = ν0 + pressure / p_ref * h.δ_air[keep]
    γ_l = (h.γ_air[keep] * (1 - vmr) * pressure / p_ref + h.γ_self[keep] * vmr * pressure / p_ref) * \
          (t_ref / temperature) ** h.n_air[keep]
    pairs = sorted(set(zip(mol.tolist(), iso.tolist())))
    sqw = np.empty(ν0.size)
    rate = np.empty(ν0.size)
    E = h.E_lower[keep]
    for (M, I) in pairs:
        sel = (mol == M) & (iso == I)
        if mol_weights is not None and (M, I) in mol_weights:
            w = np.float32(mol_weights[(M, I)])
        else:
            try:
                w = mol_weight(M, I)
            except KeyError as e:
                raise KeyError(f"no isotopologue weight for HITRAN molecule {M}, isotopologue {I}: the bundled TIPS-2017 tables (HITRAN "
                               "molecules 1-49, as in the reference) do not hold this pair; pass mol_weights={(mol, iso): g_per_mol} "
                               "(and qratio=)") from e
        sqw[sel] = np.float64(np.sqrt(w))  # Float32 sqrt, then promoted
        if np.any(E[sel] != -1):
            rate[sel] = qratio(temperature) if qratio is not None else qoft(M, I, temperature, t_ref)
        else:
            rate[sel] = 1.0
    γ_d = ((cSqrt2Ln2 / cc_) * np.sqrt(cBolts_ / cMassMol) * np.sqrt(temperature) * ν0 / sqw)
    y = np.sqrt(cLn2) * γ_l / γ_d
    corr = rate * np.exp(c2 * E * (1 / t_ref - 1 / temperature)) * \
           (1 - np.exp(-c2 * ν0 / temperature)) / (1 - np.exp(-c2 * ν0 / t_ref))
    S = np.where(E != -1, S0 * corr, S0)
    idx = np.arange(1, grid.size + 1, dtype=np.float64)
    if grid.size > 1:
        # LinearInterpolation(grid, 1:n, extrapolation_bc = 1) / (… = n) (:60-61): the constant on BOTH sides of the grid
        n = float(grid.size)
        i0 = np.rint(np.interp(ν - wing_cutoff, grid, idx, left=1.0, right=1.0)).astype(np.int32)
        i1 = np.rint(np.interp(ν + wing_cutoff, grid, idx, left=n, right=n)).astype(np.int32)
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


def resident_line_table(h, table: HitranTable, grid, wing_cutoff: float = 40.0):
    """Uploads ONE resident table of the absorber for mom_voigt_tau_abs_layer (device-side prefactors): the lines inside the
    padded grid (compute_absorption_cross_section.jl:54-72), sqrt(mol_weight) per line (Float32 square root, :88) and the
    TIPS-2017 spline tables of the isotopologues in use (qoft!, :197-214).  Returns the number of resident lines."""
    grid = np.asarray(grid, dtype=np.float64)
    keep = (grid.min() - wing_cutoff < table.νᵢ) & (table.νᵢ < grid.max() + wing_cutoff)
    mol, iso = table.mol[keep], table.iso[keep]
    E = table.E_lower[keep]
    pairs = sorted(set(zip(mol.tolist(), iso.tolist())))
    sqw = np.empty(mol.size)
    iso_index = np.full(mol.size, -1, dtype=np.int32)
    splines = []
    for (M, I) in pairs:
        sel = (mol == M) & (iso == I)
        sqw[sel] = np.float64(np.sqrt(mol_weight(M, I)))
        if np.any(E[sel] != -1):
            iso_index[sel] = len(splines)
            splines.append(CubicSpline(get_TQ(M, I), get_TT(M, I)))
    nTmax = max([len(sp.t) for sp in splines], default=0)
    tabs = np.zeros((3, len(splines), nTmax))
    for k, sp in enumerate(splines):
        n = len(sp.t)
        tabs[0, k, :n], tabs[1, k, :n], tabs[2, k, :n] = sp.t, sp.u, sp.z   # Float32 values, widened exactly
    cols = [table.νᵢ[keep], table.Sᵢ[keep], table.γ_air[keep], table.γ_self[keep], E, table.n_air[keep], table.δ_air[keep]]
    h.absorption_set_lines(cols, sqw, iso_index, [len(sp.t) for sp in splines] or [0], tabs[0], tabs[1], tabs[2])
    return int(mol.size)


def compute_absorption_profile(h, table: HitranTable, grid, p_full, T, vcd_dry, vmr, wing_cutoff: float = 40.0,
                               model_vmr: float = 0.0, qratio=None, begin: bool = True, device_prefactors: bool = False,
                               layer_by_layer: bool = False):
    """compute_absorption_profile!(τ_abs, absorption_model, grid, vmr, profile) (atmo_prof.jl:427-449) on the handle's
    resident τ_abs table: per layer the host builds the line prefactors (O(nLines)), the GPU adds
    σ(ν; p[iz], T[iz]) * vcd_dry[iz] * vmr[iz] into τ_abs[:, iz] (mom_voigt_tau_abs).  `vmr` scalar or per layer (the
    profile's mixing ratio); `model_vmr` is HitranModel.vmr, the self-broadening fraction of the line shape.
    begin=False adds another absorber to the same table (the reference's `+=` over molecules).  device_prefactors=True
    forms the per-line prefactors on the GPU from one resident line table (mom_absorption_set_lines) and runs ALL layers in
    two launches (mom_voigt_tau_abs_profile; returns their GPU time in ms); layer_by_layer=True keeps one
    mom_voigt_tau_abs_layer call per layer (same arithmetic, bitwise)."""
    p_full, T, vcd_dry = (np.asarray(x, dtype=np.float64) for x in (p_full, T, vcd_dry))
    Nz = p_full.size
    assert T.size == Nz and vcd_dry.size == Nz
    vmr_arr = np.full(Nz, float(vmr)) if np.ndim(vmr) == 0 else np.asarray(vmr, dtype=np.float64)
    assert vmr_arr.size == Nz, "Length of VMR array has to match profile size or be uniform"
    if begin:
        h.absorption_begin(Nz, grid)
    if device_prefactors:   # one resident table, per layer only (p, T, vmr, vcd) cross the bus
        if qratio is not None:
            raise ValueError("device_prefactors uses the reference's qoft! (TIPS-2017); qratio overrides are host-route only")
        resident_line_table(h, table, grid, wing_cutoff)
        if layer_by_layer:
            for iz in range(Nz):
                h.voigt_tau_abs_layer(iz + 1, p_full[iz], T[iz], model_vmr, wing_cutoff, vcd_dry[iz] * vmr_arr[iz])
            return None
        return h.voigt_tau_abs_profile(p_full, T, model_vmr, wing_cutoff, vcd_dry * vmr_arr)   # all layers in two launches
    for iz in range(Nz):
        pf = line_prefactors(table, grid, p_full[iz], T[iz], vmr=model_vmr, wing_cutoff=wing_cutoff, qratio=qratio)
        h.voigt_tau_abs(iz + 1, pf.ν, pf.γ_d, pf.y, pf.S, pf.ind_start, pf.ind_stop, vcd_dry[iz] * vmr_arr[iz])


def synthetic_o2a_lines(n_lines: int = 300, ν_lo: float = 12903.0, ν_hi: float = 13245.0, seed: int = 1234) -> HitranTable:
    """Seeded O2-A-like line list of SURVEY section 8d."""
    rng = np.random.default_rng(seed)
    ν = np.sort(rng.uniform(ν_lo, ν_hi, n_lines))
    return HitranTable(mol=np.full(n_lines, 7), iso=np.full(n_lines, 1), νᵢ=ν, Sᵢ=10.0 ** rng.uniform(-27, -23, n_lines),
                       γ_air=rng.uniform(0.03, 0.06, n_lines), γ_self=rng.uniform(0.03, 0.06, n_lines),
                       E_lower=rng.uniform(0.0, 2000.0, n_lines), n_air=np.full(n_lines, 0.7),
                       δ_air=np.full(n_lines, -0.005))
