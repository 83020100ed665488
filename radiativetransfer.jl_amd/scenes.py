"""Seeded synthetic scenes for benchmarks and parity tests (no external data): SURVEY.md
section 8d / BASELINE.json `configs`.  They stand in for what the reference's
model_from_parameters() derives from YAML + HITRAN + Mie, with the same shapes:

  C1  scalar I, 4 streams, 10 layers, S = 100                  ("default_parameters-like" plumbing)
  C2  O2 A-band, IQU, 20 streams (N = 60), 40 layers, S = 10 000   <- headline metric
  C3  3-band OCO-2-like, S = 29 944, sharded over the GPUs of a node
  C4  aerosol + cloud, IQUV, 64 streams (N = 256)
"""
from __future__ import annotations

import math
from typing import Optional

import numpy as np

from . import corert as rt


def hg_like_greek(g: float, lmax: int) -> rt.GreekCoefs:
    """Henyey-Greenstein-like expansion: β_l = (2l+1) g^l, polarised terms scaled from it."""
    l = np.arange(lmax, dtype=np.float64)
    β = (2 * l + 1) * g ** l
    hi = (l >= 2).astype(np.float64)
    return rt.GreekCoefs(α=0.9 * β * hi, β=β, γ=0.08 * β * hi * g, δ=0.8 * β, ϵ=0.02 * β * hi * g, ζ=0.9 * β * hi)


def pressure_grid(Nz: int) -> np.ndarray:
    """Nz layers between log-spaced half levels 0.1 ... 1000 hPa (TOA first)."""
    return np.exp(np.linspace(math.log(0.1), math.log(1000.0), Nz + 1))


def rayleigh_tau(ν: np.ndarray, p_half: np.ndarray, depol: float) -> np.ndarray:
    """getRayleighLayerOptProp (atmo_prof.jl:210-224) with vcd ∝ Δp.  ν in cm⁻¹ -> [S, Nz]."""
    λ = 1.0e4 / ν  # μm
    tau = 0.00864 * (p_half[-1] / 1013.25) * λ ** (-(3.916 + 0.074 * λ + 0.05 / λ))
    tau = tau * (6.0 + 3.0 * depol) / (6.0 - 7.0 * depol)
    dp = np.diff(p_half)
    return tau[:, None] * (dp / dp.sum())[None, :]


def aerosol_profile(total: float, p0: float, σp: float, p_half: np.ndarray) -> np.ndarray:
    """Gaussian-in-pressure aerosol optical depth per layer (like getAerosolLayerOptProp)."""
    pc = 0.5 * (p_half[1:] + p_half[:-1])
    w = np.exp(-0.5 * ((pc - p0) / σp) ** 2) * np.diff(p_half)
    return total * w / w.sum()


def lognormal_absorption(S: int, p_half: np.ndarray, seed: int) -> np.ndarray:
    """Line-like gas absorption: column optical depth log-uniform-ish over 1e-3 ... 50 along the
    spectral axis (smooth random walk + spikes), split over layers ∝ Δp·p (pressure broadening)."""
    rng = np.random.default_rng(seed)
    x = np.cumsum(rng.normal(0.0, 0.35, S))
    x = (x - x.min()) / max(x.max() - x.min(), 1e-300)
    col = 10.0 ** (-3.0 + x * (math.log10(50.0) + 3.0))
    pc = 0.5 * (p_half[1:] + p_half[:-1])
    w = np.diff(p_half) * pc
    return col[:, None] * (w / w.sum())[None, :]


def make_scene(nStokes: int, l_trunc: int, Nz: int, S: int, ν_lo: float = 12903.0, ν_hi: float = 13245.0,
               sza: Optional[float] = None, vza=(0.0, 30.0, 60.0), vaz=(0.0, 45.0, 120.0), aerosol_total: float = 0.2,
               aerosol_p0: float = 800.0, aerosol_σp: float = 50.0, g: float = 0.7, albedo: float = 0.2,
               depol: float = 0.03, max_m: int = 3, seed: int = 1234, absorption: bool = True,
               architecture=None) -> rt.vSmartMOM_Model:
    pol = {1: rt.Stokes_I, 3: rt.Stokes_IQU, 4: rt.Stokes_IQUV}[nStokes]()
    if sza is None:
        # Sun at 60 deg -- unless the Gauss rule itself has a node at cosd(60) = 0.5 (odd number of nodes on [0, 1]):
        # then mu0 would merge with that node under rt_set_streams' `unique`; 50 deg keeps the Sun a zero-weight stream
        # of its own, so every scene has (l_trunc + 1) // 2 + 3 streams for the default three view angles
        sza = 50.0 if ((l_trunc + 1) // 2) % 2 == 1 else 60.0
    params = rt.vSmartMOM_Parameters(
        polarization_type=pol, quadrature_type="GaussQuadHemisphere", max_m=max_m, l_trunc=l_trunc, depol=depol,
        sza=sza, vza=np.asarray(vza, dtype=np.float64), vaz=np.asarray(vaz, dtype=np.float64), brdf_albedo=albedo,
        architecture=architecture or rt.default_architecture())
    p_half = pressure_grid(Nz)
    ν = np.linspace(ν_lo, ν_hi, S)
    τ_rayl = rayleigh_tau(ν, p_half, depol)
    τ_abs = lognormal_absorption(S, p_half, seed) if absorption else np.zeros((S, Nz))
    aer, τ_aer = [], None
    if aerosol_total > 0:
        aer = [rt.AerosolOptics(hg_like_greek(g, l_trunc), 0.95, 0.0)]
        τ_aer = aerosol_profile(aerosol_total, aerosol_p0, aerosol_σp, p_half)[None, :]
    return rt.model_from_parameters(params, τ_rayl, τ_abs, τ_aer, aer)


def scene_C1(S: int = 100, **kw):
    """scalar I, Nquad_eff = 4 (2 Gauss nodes + μ=1 + μ₀), 10 layers."""
    return make_scene(1, 3, 10, S, vza=(0.0,), vaz=(0.0,), **kw)


def scene_C2(S: int = 10_000, Nz: int = 40, **kw):
    """O2 A-band IQU: 17 Gauss nodes (one of them at 0.5 = cos 60°, the third view angle) + {1, cos 30°} + μ₀ = cos 50°
    = 20 streams, N = 60."""
    return make_scene(3, 33, Nz, S, **kw)


def scene_C3(S: int = 29_944, Nz: int = 40, **kw):
    """three concatenated bands (13 672 + 6 402 + 9 870 points) on one spectral axis."""
    m = make_scene(3, 33, Nz, S, **kw)
    n1, n2 = 13_672 * S // 29_944, 6_402 * S // 29_944
    ν = np.concatenate([np.linspace(12903.0, 13245.0, n1), np.linspace(6170.0, 6290.0, n2),
                        np.linspace(4800.0, 4900.0, S - n1 - n2)])
    m.τ_rayl[:] = rayleigh_tau(ν, pressure_grid(Nz), m.params.depol)
    return m


def scene_C4(S: int = 2_000, Nz: int = 40, **kw):
    """aerosol + cloud, IQUV, 61 Gauss nodes (incl. 0.5 = cos 60°) + {1, cos 30°, μ₀ = cos 50°} = 64 streams, N = 256."""
    kw.setdefault("aerosol_total", 5.0)
    kw.setdefault("aerosol_p0", 700.0)
    return make_scene(4, 121, Nz, S, **kw)


def raman_lines(Δν_grid: float, T: float = 250.0, Jmax: int = 30, ϖ_Cabannes: float = 0.96, S: Optional[int] = None):
    """Synthetic rotational-Raman line list in the form the CoreRT hot path consumes (RS_type.i_λ₁λ₀, ϖ_λ₁λ₀): the N₂ and O₂
    J -> J ± 2 Stokes / anti-Stokes lines (Δν̃ = ± 4B (J + 3/2), Jmax = 30 like inelastic_cross_section.jl:142-160; N₂: all
    J with 6:3 nuclear-spin weights, B = 1.98957 cm⁻¹; O₂: odd J only, B = 1.43768 cm⁻¹; vmr 0.79 / 0.21), Boltzmann
    populations at T times the Placzek-Teller factors, each line split 50/50 onto its two neighbouring grid points
    (apply_gridlines!, src/Inelastic/inelastic_helper.jl:146-216), weights normalised to Σ ϖ_λ₁λ₀ = 1 - ϖ_Cabannes.
    Stands in for getRamanSSProp! (raman_atmo_prop.jl:57-73), which needs the molecular constants (host set-up, out of scope).
    Returns (i_λ₁λ₀ [nRaman] int, ϖ_λ₁λ₀ [nRaman]); offsets with |offset| >= S are dropped."""
    kT = 0.6950348 * T  # cm⁻¹
    acc = {}
    for B, vmr, odd_only, gJ in ((1.98957, 0.79, False, lambda J: 6.0 if J % 2 == 0 else 3.0), (1.43768, 0.21, True, lambda J: 1.0)):
        Js = [J for J in range(Jmax + 1) if (J % 2 == 1 or not odd_only)]
        pop = np.array([gJ(J) * (2 * J + 1) * math.exp(-B * J * (J + 1) / kT) for J in Js])
        pop /= pop.sum()
        for J, pJ in zip(Js, pop):
            # Stokes J -> J + 2 (scattered light at lower wavenumber: the source index n₀ lies ABOVE n₁) and anti-Stokes J -> J - 2
            for dJ in (+2, -2):
                if J + dJ < 0:
                    continue
                if dJ > 0:
                    pt = 3.0 * (J + 1) * (J + 2) / (2.0 * (2 * J + 1) * (2 * J + 3))
                    shift = 4.0 * B * (J + 1.5)
                else:
                    pt = 3.0 * J * (J - 1) / (2.0 * (2 * J + 1) * (2 * J - 1))
                    shift = -4.0 * B * (J - 0.5)
                x = shift / Δν_grid
                lo = math.floor(x)
                for k in (lo, lo + 1):
                    acc[k] = acc.get(k, 0.0) + 0.5 * vmr * pJ * pt
    offs = np.array(sorted(k for k in acc if k != 0 and (S is None or abs(k) < S)), dtype=np.int32)
    w = np.array([acc[int(k)] for k in offs])
    return offs, (1.0 - ϖ_Cabannes) * w / w.sum()


def scene_C5(S: int = 6_837, Nz: int = 5, nRaman: Optional[int] = None, rrs_strict_reference: bool = False, **kw):
    """BASELINE config 5, the reference's own RRS driver shape (test/benchmarks/prototype_inelastic.jl:8-93 on
    test/test_parameters/O2Parameters.yaml): O₂ A-band at 0.05 cm⁻¹ (6 837 points), Stokes IQU, l_trunc 5 -> 3 Gauss nodes +
    Sun + one view = 5 streams (N = 15), 5 layers (profile_reduction 5), Rayleigh + gas absorption, Lambertian surface;
    rotational-Raman lines from raman_lines (≈ 190 grid offsets; nRaman keeps the strongest ones -- parity tests use few).
    Returns (model, RS_type)."""
    kw.setdefault("aerosol_total", 0.0)
    kw.setdefault("vza", (30.0,))
    kw.setdefault("vaz", (20.0,))
    Δν = 0.05
    m = make_scene(3, 5, Nz, S, ν_lo=12950.0, ν_hi=12950.0 + Δν * (S - 1), **kw)
    offs, w = raman_lines(Δν, S=S)
    if nRaman is not None and nRaman < len(offs):
        keep = np.sort(np.argsort(-w)[:nRaman])
        offs, w = offs[keep], w[keep] * (w.sum() / w[keep].sum())
    RS = rt.RRS(greek_raman=rt.get_greek_rayleigh(0.75), ϖ_Cabannes=0.96, ϖ_λ1λ0=w, i_λ1λ0=offs,
                rrs_strict_reference=rrs_strict_reference)
    return m, RS


def work_model_flops(N: int, ndoubl, M: int) -> float:
    """ALGORITHMIC flop per spectral point: the reference's op list, GEMM =
    2N³, inverse = 2N³, matvec = 2N², regardless of how the kernels restructure it."""
    Nz = len(ndoubl)
    per_m = float(np.sum(ndoubl)) * (12 * N ** 3 + 8 * N ** 2) + Nz * (24 * N ** 3 + 8 * N ** 2)
    return M * per_m + M * Nz * N * N * (3 + 12)
