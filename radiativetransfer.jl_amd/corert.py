"""Host-side mirror of the reference's CoreRT interface for the elastic path.

The reference (vSmartMOM.jl) keeps this logic in Julia; no Julia toolchain exists in this
image, so the same interface is restated in Python above the C ABI (include/momcore.h):
same names, argument meaning and error behaviour, so that tests read like the reference's
own (test/test_CoreRT.jl).  Everything numerical on the hot path -- elemental, doubling,
interaction, surface, post-processing -- runs in libmomcore.so on the GPU; this module only
prepares the inputs the reference's host code prepares (streams, phase-matrix Fourier
moments, layer optics, doubling numbers, interface codes) and replays rt_run's loops.

Reference files followed (relative to the reference root):
  src/CoreRT/rt_run.jl:41-230                       rt_run
  src/CoreRT/tools/rt_set_streams.jl:24-170         rt_set_streams
  src/CoreRT/tools/rt_helper_functions.jl:8-57      interface state machine, doubling_number
  src/CoreRT/CoreKernel/rt_kernel.jl:238-275        get_dtau_ndoubl / init_layer
  src/CoreRT/LayerOpticalProperties/compEffectiveLayerProperties.jl:1-135, types.jl:632-678
  src/Scattering/compute_Z_matrices.jl:5-84, legendre_functions.jl:17-178,
  src/Scattering/mie_helper_functions.jl:237-251,287-348
  src/Architectures.jl:20-55                        architecture dispatch seam
"""
from __future__ import annotations

import math
from dataclasses import dataclass, field
from typing import List, Optional, Sequence

import numpy as np

from . import _lib

EPS = float(np.finfo(np.float64).eps)

# ------------------------------------------------------------------------------------------
# Architectures (src/Architectures.jl:20-55): CPU()/GPU() exist in the reference; MI355X is
# the architecture this package adds.  There is deliberately no CPU execution path here.
# ------------------------------------------------------------------------------------------


class AbstractArchitecture:
    pass


@dataclass(frozen=True)
class MI355X(AbstractArchitecture):
    device: int = 0


def default_architecture() -> MI355X:
    return MI355X(0)


# ------------------------------------------------------------------------------------------
# Polarization types (src/Scattering/types.jl:82-123)
# ------------------------------------------------------------------------------------------


@dataclass(frozen=True)
class PolarizationType:
    n: int
    D: tuple
    I0: tuple


def Stokes_I():
    return PolarizationType(1, (1.0,), (1.0,))


def Stokes_IQU():
    return PolarizationType(3, (1.0, 1.0, -1.0), (1.0, 0.0, 0.0))


def Stokes_IQUV():
    return PolarizationType(4, (1.0, 1.0, -1.0, -1.0), (1.0, 0.0, 0.0, 0.0))


# ------------------------------------------------------------------------------------------
# trig in degrees like Julia's sind/cosd (base/special/trig.jl): the argument is reduced in DEGREES
# (exact) to [-45, 45], converted to radians in double-double (deg2rad_ext) and fed to the sine or
# cosine kernel with its low word -- so cosd(60) == 0.5 and sind(30) == 0.5 EXACTLY, not
# 0.5000000000000001.  This matters beyond an ulp: rt_set_streams removes duplicate nodes with
# `unique`, and a Gauss node at 0.5 next to mu0 = cosd(60) must collapse into one stream
# (near-duplicate nodes make t++ depend on the rounding of exp).
# ------------------------------------------------------------------------------------------

def _deg2rad_ext(x: float):
    """deg2rad_ext (Julia base/special/trig.jl): x * (pi/180 as a double) as an exact double-double (Dekker split)."""
    m, m_hi, m_lo = 0.017453292519943295, 0.01745329238474369, 1.3519960527851425e-10
    u = 134217729.0 * x  # 0x1p27 + 1
    x_hi = u - (u - x)
    x_lo = x - x_hi
    y_hi = m * x
    y_lo = x_hi * m_lo + (x_lo * m_hi + ((x_hi * m_hi - y_hi) + x_lo * m_lo))
    return y_hi, y_lo


# fdlibm's __kernel_sin / __kernel_cos on a (hi, lo) argument, |hi| <= pi/4 -- the kernels Julia's sin_kernel /
# cos_kernel(::DoubleFloat64) evaluate
_S = (-1.66666666666666324348e-01, 8.33333333332248946124e-03, -1.98412698298579493134e-04,
      2.75573137070700676789e-06, -2.50507602534068634195e-08, 1.58969099521155010221e-10)
_C = (4.16666666666666019037e-02, -1.38888888888741095749e-03, 2.48015872894767294178e-05,
      -2.75573143513906633035e-07, 2.08757232129817482790e-09, -1.13596475577881948265e-11)


def _sin_k(deg: float) -> float:  # |deg| <= 45
    x, y = _deg2rad_ext(deg)
    z = x * x
    v = z * x
    r = _S[1] + z * (_S[2] + z * (_S[3] + z * (_S[4] + z * _S[5])))
    return x - ((z * (0.5 * y - v * r) - y) - v * _S[0])


def _cos_k(deg: float) -> float:  # |deg| <= 45
    x, y = _deg2rad_ext(deg)
    z = x * x
    w = z * z
    r = z * (_C[0] + z * (_C[1] + z * _C[2])) + w * w * (_C[3] + z * (_C[4] + z * _C[5]))
    hz = 0.5 * z
    w = 1.0 - hz
    return w + (((1.0 - w) - hz) + (z * r - x * y))


def cosd(x: float) -> float:
    rx = abs(math.fmod(float(x), 360.0))
    if rx <= 45.0:
        return _cos_k(rx)
    if rx < 135.0:
        return _sin_k(90.0 - rx)
    if rx <= 225.0:
        return -_cos_k(180.0 - rx)
    if rx < 315.0:
        return _sin_k(rx - 270.0)
    return _cos_k(360.0 - rx)


def sind(x: float) -> float:
    rx = math.fmod(float(x), 360.0)
    arx = abs(rx)
    sg = math.copysign(1.0, rx)
    if rx == 0.0:
        return rx
    if arx < 45.0:
        return _sin_k(rx)
    if arx <= 135.0:
        return math.copysign(_cos_k(90.0 - arx), rx)
    if arx == 180.0:
        return math.copysign(0.0, rx)
    if arx < 225.0:
        return _sin_k((180.0 - arx) * sg)
    if arx <= 315.0:
        return -math.copysign(_cos_k(270.0 - arx), rx)
    return _sin_k(rx - math.copysign(360.0, rx))


# ------------------------------------------------------------------------------------------
# Quadrature streams
# ------------------------------------------------------------------------------------------


@dataclass
class QuadPoints:
    """types.jl:456-473; iμ₀ and iμ₀Nstart are 1-based like the reference."""
    μ0: float
    iμ0: int
    iμ0Nstart: int
    qp_μ: np.ndarray
    wt_μ: np.ndarray
    qp_μN: np.ndarray
    wt_μN: np.ndarray
    Nquad: int


def _julia_unique(values: Sequence[float]) -> np.ndarray:
    out: List[float] = []
    for v in values:
        if float(v) not in out:
            out.append(float(v))
    return np.asarray(out, dtype=np.float64)


def _gauss_radau(n: int):
    """n-point Gauss-Radau rule on [-1, 1], fixed node -1 first (what
    FastGaussQuadrature.gaussradau returns at rt_set_streams.jl:115)."""
    from scipy.special import roots_jacobi

    if n == 1:
        return np.array([-1.0]), np.array([2.0])
    x, v = roots_jacobi(n - 1, 0.0, 1.0)
    return np.concatenate(([-1.0], x)), np.concatenate(([2.0 / (n * n)], v / (1.0 + x)))


def rt_set_streams(quadrature_type: str, Ltrunc: int, sza: float, vza: Sequence[float],
                   pol_type: PolarizationType) -> QuadPoints:
    """rt_set_streams(::GaussQuadHemisphere | ::GaussQuadFullSphere | ::RadauQuad, ...)."""
    Nq = (Ltrunc + 1) // 2
    μ0 = cosd(sza)
    cams = [cosd(v) for v in vza]
    if quadrature_type == "GaussQuadHemisphere":
        x, w = np.polynomial.legendre.leggauss(Nq)
        nodes, weights = 0.5 * x + 0.5, 0.5 * w
        qp = _julia_unique(list(nodes) + cams + [μ0])
    elif quadrature_type == "GaussQuadFullSphere":
        x, w = np.polynomial.legendre.leggauss(2 * Nq)
        nodes, weights = x[Nq:], w[Nq:]
        qp = _julia_unique(list(nodes) + cams + [μ0])
    elif quadrature_type == "RadauQuad":
        tx, tw = _gauss_radau(Nq)
        x0, w0 = -tx[::-1], tw[::-1]
        if μ0 in x0:
            nodes, weights = (1.0 + x0) / 2.0, w0.copy()
        else:
            nodes = np.concatenate([(μ0 + μ0 * x0) / 2.0, ((1.0 + μ0) + (1.0 - μ0) * x0) / 2.0])
            weights = np.concatenate([μ0 * w0 / 2.0, (1.0 - μ0) * w0 / 2.0])
        qp = _julia_unique(list(nodes) + cams)
    else:
        raise ValueError(f"unknown quadrature type {quadrature_type!r}")
    wt = np.concatenate([weights, np.zeros(len(qp) - len(weights))])
    iμ0 = int(np.argmin(np.abs(qp - μ0))) + 1
    n = pol_type.n
    return QuadPoints(μ0, iμ0, n * (iμ0 - 1) + 1, qp, wt, np.repeat(qp, n), np.repeat(wt, n), len(qp))


# ------------------------------------------------------------------------------------------
# Greek coefficients and phase-matrix Fourier moments
# ------------------------------------------------------------------------------------------


@dataclass
class GreekCoefs:
    """src/Scattering/types.jl:198-211"""
    α: np.ndarray
    β: np.ndarray
    γ: np.ndarray
    δ: np.ndarray
    ϵ: np.ndarray
    ζ: np.ndarray


def get_greek_rayleigh(depol: float) -> GreekCoefs:
    """mie_helper_functions.jl:237-251"""
    p = (1 - depol) / (1 + depol / 2)
    r = (1 - 2 * depol) / (1 - depol)
    z3 = np.zeros(3)
    return GreekCoefs(α=np.array([0.0, 0.0, 3 * p]), β=np.array([1.0, 0.0, 0.5 * p]),
                      γ=np.array([0.0, 0.0, p * math.sqrt(1.5)]), δ=np.array([0.0, p * r * 1.5, 0.0]),
                      ϵ=z3.copy(), ζ=z3.copy())


def _prt_for_moment(μ: np.ndarray, lmax: int, m: int):
    """Generalised spherical functions P, R, T for ONE Fourier index m and l = m..lmax-1 at the
    nodes μ (legendre_functions.jl:17-178, normalisation built in).  Returns arrays [lmax, nμ]
    (rows l < m are zero) with T sign-flipped as the reference returns it."""
    μ = np.asarray(μ, dtype=np.float64)
    s = np.sqrt(1.0 - μ * μ)
    P = np.zeros((lmax, μ.size))
    R = np.zeros_like(P)
    T = np.zeros_like(P)
    for l in range(m, lmax):
        if m == 0:
            if l == 0:
                P[l] = 1.0
            elif l == 1:
                P[l] = μ
            elif l == 2:
                P[l] = 0.5 * (3.0 * μ * μ - 1.0)
                R[l] = 0.5 * math.sqrt(1.5) * s * s
            else:
                P[l] = (P[l - 1] * (2 * l - 1) * μ - P[l - 2] * (l - 1)) / l
                R[l] = (R[l - 1] * (2 * l - 1) * μ - R[l - 2] * math.sqrt((l + 1) * (l - 3))) / math.sqrt(l * l - 4)
            continue
        if m == 1 and l == 1:
            P[l] = math.sqrt(0.5) * s
            continue
        if m == 1 and l == 2:
            c = math.sqrt(1 / 6)
            P[l] = c * (3 * μ * s)
            R[l] = -c * μ * (math.sqrt(1.5) * s)
            T[l] = c * (math.sqrt(1.5) * s)
            continue
        if l == m:  # m >= 2 seed (eq. 36, 37)
            f1 = np.ones_like(μ)
            f2 = np.ones_like(μ)
            for i in range(1, m + 1):
                f1 = f1 * ((2 * i - 1) * s) / math.sqrt(i * (i + m))
                f2 = f2 * (s / 2) * (math.sqrt((m + i) / (i - 2)) if i > 2 else 1.0)
            ok = s > 1e-8
            ss = np.where(ok, s * s, 1.0)
            edge = 0.5 if m == 2 else 0.0
            P[l] = f1
            R[l] = np.where(ok, f2 * (1.0 + μ * μ) / ss, edge)
            T[l] = -np.where(ok, f2 * (2 * μ) / ss, edge)
            continue
        # three-term recurrences (eq. 34-35, 38)
        if l == m + 1 and m >= 2:
            m1, m2 = math.sqrt(1 / (l + m)), 0.0
        elif m == 1:
            m1 = math.sqrt((l - 1) / (l + 1))
            m2 = m1 * math.sqrt((l - 2) / l)
        else:
            m1 = math.sqrt((l - m) / (l + m))
            m2 = m1 * math.sqrt((l - m - 1) / (l + m - 1))
        Zc = (2 * m * (2 * l - 1)) / (l * (l - 1))
        Yp, Xp = (l - 1 + m), (l - m)
        Yr = ((l + m - 1) / (l - 1)) * math.sqrt((l - 3) * (l + 1))
        Xr = ((l - m) / l) * math.sqrt(l * l - 4)
        Pm2 = P[l - 2] if l - 2 >= 0 else 0.0
        P[l] = (m1 * P[l - 1] * (2 * l - 1) * μ - m2 * Pm2 * Yp) / Xp
        R[l] = (m1 * R[l - 1] * (2 * l - 1) * μ - m2 * R[l - 2] * Yr + m1 * T[l - 1] * Zc) / Xr
        T[l] = (m1 * T[l - 1] * (2 * l - 1) * μ - m2 * T[l - 2] * Yr + m1 * R[l - 1] * Zc) / Xr
    return P, R, -T


def compute_Z_moments(pol_type: PolarizationType, μ: np.ndarray, greek: GreekCoefs, m: int):
    """Scattering.compute_Z_moments (compute_Z_matrices.jl:5-84): Z⁺⁺, Z⁻⁺ of shape [N, N]
    (N = n*len(μ); row = outgoing, column = incoming) for Fourier moment m."""
    μ = np.asarray(μ, dtype=np.float64)
    if not np.all((0 < μ) & (μ <= 1)):
        raise AssertionError("all μ's within compute_Z_moments have to be ∈ ]0,1]")
    n, nμ, lmax = pol_type.n, μ.size, len(greek.β)
    fact = 0.5 if m == 0 else 1.0
    Pp, Rp, Tp = _prt_for_moment(μ, lmax, m)
    Pn, Rn, Tn = _prt_for_moment(-μ, lmax, m)

    def pi_stack(P, R, T):  # [lmax, nμ, n, n]  (construct_Π_matrix, mie_helper_functions.jl:287-323)
        Π = np.zeros((lmax, nμ, n, n))
        Π[:, :, 0, 0] = P
        if n >= 3:
            Π[:, :, 1, 1] = R
            Π[:, :, 2, 2] = R
            Π[:, :, 1, 2] = -T
            Π[:, :, 2, 1] = -T
        if n == 4:
            Π[:, :, 3, 3] = P
        return Π

    B = np.zeros((lmax, n, n))  # construct_B_matrix (:334-348)
    B[:, 0, 0] = greek.β
    if n >= 3:
        B[:, 0, 1] = greek.γ
        B[:, 1, 0] = greek.γ
        B[:, 1, 1] = greek.α
        B[:, 2, 2] = greek.ζ
    if n == 4:
        B[:, 2, 3] = greek.ϵ
        B[:, 3, 2] = -greek.ϵ
        B[:, 3, 3] = greek.δ
    Πp, Πn = pi_stack(Pp, Rp, Tp), pi_stack(Pn, Rn, Tn)
    App = np.zeros((nμ, nμ, n, n))
    Amp = np.zeros((nμ, nμ, n, n))
    for l in range(m, lmax):  # sequential over l like the reference's accumulation
        left = Πp[l] @ B[l]  # [nμ, n, n]
        App += np.einsum("iab,jbc->ijac", left, Πp[l])
        Amp += np.einsum("iab,jbc->ijac", left, Πn[l])
    sign = np.ones((n, n))
    if n >= 3:
        sign[:2, 2:] = -1.0
        sign[2:, :2] = -1.0
    N = n * nμ
    Zpp = (2 * fact * App).transpose(0, 2, 1, 3).reshape(N, N)
    Zmp = (2 * fact * Amp * sign).transpose(0, 2, 1, 3).reshape(N, N)
    return Zpp, Zmp


# ------------------------------------------------------------------------------------------
# Surface types (types.jl:300-344) and their Fourier-moment reflectance matrices
# ------------------------------------------------------------------------------------------


@dataclass(frozen=True)
class LambertianSurfaceScalar:
    albedo: float


@dataclass(frozen=True)
class LambertianSurfaceLegendre:
    """albedo(ν) = Σ legendre_coeff[k] P_k(x), x = linspace(-1, 1, nSpec) (lambertian_surface.jl:90-96)"""
    legendre_coeff: tuple


@dataclass(frozen=True)
class rpvSurfaceScalar:
    """Rahman-Pinty-Verstraete (types.jl:320-329)"""
    ρ0: float
    ρ_c: float
    k: float
    Θ: float


@dataclass(frozen=True)
class RossLiSurfaceScalar:
    """Ross-thick / Li-sparse kernels (types.jl:331-338)"""
    fvol: float
    fgeo: float
    fiso: float


def gauleg(n: int, a: float, b: float):
    """Gauss-Legendre nodes/weights on [a, b] (CanopyOptics.gauleg as used at rpv_surface.jl:116)."""
    x, w = np.polynomial.legendre.leggauss(n)
    return 0.5 * (b - a) * x + 0.5 * (b + a), 0.5 * (b - a) * w


def _brdf_scalar(brdf, μi: np.ndarray, μr: np.ndarray, dϕ: float) -> np.ndarray:
    """reflectance(brdf, n = 1, μᵢ, μᵣ, dϕ) broadcast over μᵢ (rows) x μᵣ (columns)."""
    if isinstance(brdf, rpvSurfaceScalar):  # rpv_surface.jl:69-95
        θi, θr = np.arccos(μi), np.arccos(μr)
        cosg = -μi * μr + np.sin(θi) * np.sin(θr) * math.cos(dϕ)
        G = (np.tan(θi) ** 2 + np.tan(θr) ** 2 + 2 * np.tan(θi) * np.tan(θr) * math.cos(dϕ)) ** 0.5
        Mf = (μi * μr) ** (brdf.k - 1) / (μi + μr) ** (1 - brdf.k)
        θ = -brdf.Θ
        F = (1 - θ ** 2) / (1 + θ ** 2 + 2 * θ * cosg) ** 1.5
        H = 1 + (1 - brdf.ρ_c) / (1 + G)
        return brdf.ρ0 * Mf * F * H
    if isinstance(brdf, RossLiSurfaceScalar):  # rossli_surface.jl:1-56
        dϕ = math.pi - dϕ
        θi, θr = np.arccos(μi), np.arccos(μr)
        ξ = np.arccos(np.cos(θi) * np.cos(θr) + np.sin(θi) * np.sin(θr) * math.cos(dϕ))
        K_vol = ((math.pi / 2 - ξ) * np.cos(ξ) + np.sin(ξ)) / (np.cos(θi) + np.cos(θr)) - (math.pi / 4)
        h_by_b, b_by_r = 2.0, 1.0
        θip, θrp = np.arctan(np.tan(θi) * b_by_r), np.arctan(np.tan(θr) * b_by_r)
        ξp = np.arccos(np.cos(θip) * np.cos(θrp) + np.sin(θip) * np.sin(θrp) * math.cos(dϕ))
        D = np.sqrt(np.tan(θip) ** 2 + np.tan(θrp) ** 2 - 2 * np.tan(θip) * np.tan(θrp) * math.cos(dϕ))
        sec = lambda a: 1.0 / np.cos(a)
        ct = h_by_b * np.sqrt(D ** 2 + (np.tan(θip) * np.tan(θrp) * math.sin(dϕ)) ** 2) / (sec(θip) + sec(θrp))
        ct = np.clip(ct, -1.0, 1.0)
        t = np.arccos(ct)
        O = (1 / math.pi) * (t - np.sin(t) * np.cos(t)) * (sec(θip) + sec(θrp))
        K_geo = O - (sec(θip) + sec(θrp)) + 0.5 * (1 + np.cos(ξp)) * sec(θip) * sec(θrp)
        return brdf.fiso * 1.0 + brdf.fvol * K_vol + brdf.fgeo * K_geo
    raise TypeError(f"no reflectance() for {type(brdf).__name__}")


def reflectance(brdf, pol_type: PolarizationType, μ: np.ndarray, m: int, nQuad: int = 100) -> np.ndarray:
    """reflectance(brdf, pol_type, μ, m) (rpv_surface.jl:104-134): Fourier moment m of the BRDF on the quadrature
    streams, (1/π) ∫₀^π ρ(μᵢ, μⱼ, φ) cos(mφ) dφ by nQuad-point Gauss-Legendre, expanded to the Stokes layout (only the
    I component is populated by the scalar BRDF types), times 1 (m = 0) or 2 (m > 0)."""
    μ = np.asarray(μ, dtype=np.float64)
    n, nμ = pol_type.n, μ.size
    R = np.zeros((n * nμ, n * nμ))
    ff = 1.0 if m == 0 else 2.0
    ϕ, w = gauleg(nQuad, 0.0, math.pi)
    c = np.zeros((nμ, nμ))
    for ϕi, wi in zip(ϕ, w):
        with np.errstate(invalid="ignore", divide="ignore"):
            c += wi * (_brdf_scalar(brdf, μ[:, None], μ[None, :], float(ϕi)) * math.cos(m * ϕi))
    R[0::n, 0::n] = c / math.pi
    return ff * R


def compute_legendre_poly(x: np.ndarray, nmax: int) -> np.ndarray:
    """Scattering.compute_legendre_poly(x, nmax)[1]: P_0 .. P_{nmax-1} at x, shape [len(x), nmax]."""
    x = np.asarray(x, dtype=np.float64)
    P = np.zeros((x.size, nmax))
    P[:, 0] = 1.0
    if nmax > 1:
        P[:, 1] = x
    for l in range(2, nmax):
        P[:, l] = ((2 * l - 1) * x * P[:, l - 1] - (l - 1) * P[:, l - 2]) / l
    return P


def surface_inputs(brdf, pol_type: PolarizationType, qp_μ: np.ndarray, max_m: int, nSpec: int):
    """What mom_scene_set_surface needs for a surface type: (kind, Rsurf [M, N, N] or None, albedo_spec [nSpec] or None).
    create_surface_layer! multiplies the m = 0 BRDF moment by 2 (rpv_surface.jl:39-43)."""
    if isinstance(brdf, LambertianSurfaceScalar):
        return 0, None, None
    if isinstance(brdf, LambertianSurfaceLegendre):
        x = np.linspace(-1.0, 1.0, nSpec)
        coef = np.asarray(brdf.legendre_coeff, dtype=np.float64)
        return 2, None, compute_legendre_poly(x, coef.size) @ coef
    Rs = np.array([(2.0 if m == 0 else 1.0) * reflectance(brdf, pol_type, qp_μ, m) for m in range(max_m)])
    return 1, Rs, None


# ------------------------------------------------------------------------------------------
# model containers
# ------------------------------------------------------------------------------------------


@dataclass
class AerosolOptics:
    """The fields of the reference's AerosolOptics that the RT core consumes."""
    greek_coefs: GreekCoefs
    ω̃: float
    fᵗ: float = 0.0


@dataclass
class vSmartMOM_Parameters:
    """Subset of types.jl:394-446 consumed by the elastic hot path."""
    polarization_type: PolarizationType
    quadrature_type: str
    max_m: int
    l_trunc: int
    depol: float
    sza: float
    vza: np.ndarray
    vaz: np.ndarray
    brdf_albedo: float = 0.0  # LambertianSurfaceScalar(albedo)
    brdf: Optional[object] = None  # any surface type above; None = LambertianSurfaceScalar(brdf_albedo)
    architecture: AbstractArchitecture = field(default_factory=default_architecture)
    strict_reference_indexing: bool = True


@dataclass
class vSmartMOM_Model:
    """types.jl:476-520, restricted to one concatenated band: τ arrays are [nSpec, Nz]."""
    params: vSmartMOM_Parameters
    quad_points: QuadPoints
    greek_rayleigh: GreekCoefs
    τ_rayl: np.ndarray
    τ_abs: np.ndarray
    τ_aer: np.ndarray  # [nAer, Nz]
    aerosol_optics: List[AerosolOptics]
    ϖ_Cabannes: float = 1.0


def model_from_parameters(params: vSmartMOM_Parameters, τ_rayl, τ_abs, τ_aer=None,
                          aerosol_optics: Optional[List[AerosolOptics]] = None) -> vSmartMOM_Model:
    """model_from_parameters(params) (model_from_parameters.jl:12-194) for callers that bring
    their own optical-depth tables (the reference computes them from profiles/HITRAN/Mie, which
    stays host-side Julia and is outside this package's scope)."""
    qp = rt_set_streams(params.quadrature_type, params.l_trunc, params.sza, params.vza, params.polarization_type)
    τ_rayl = np.asarray(τ_rayl, dtype=np.float64)
    τ_abs = np.asarray(τ_abs, dtype=np.float64)
    if τ_rayl.shape != τ_abs.shape or τ_rayl.ndim != 2:
        raise ValueError("τ_rayl and τ_abs must both be [nSpec, Nz]")
    aerosol_optics = list(aerosol_optics or [])
    τ_aer = np.zeros((0, τ_rayl.shape[1])) if τ_aer is None else np.asarray(τ_aer, dtype=np.float64)
    if τ_aer.shape != (len(aerosol_optics), τ_rayl.shape[1]):
        raise ValueError("τ_aer must be [nAer, Nz]")
    return vSmartMOM_Model(params, qp, get_greek_rayleigh(params.depol), τ_rayl, τ_abs, τ_aer, aerosol_optics)


# ------------------------------------------------------------------------------------------
# layer optics, doubling numbers, interface codes
# ------------------------------------------------------------------------------------------


def doubling_number(dτ_max: float, τ_end: float):
    """rt_helper_functions.jl:31-57 (log10 arithmetic and the eps test kept verbatim)."""
    if τ_end <= dτ_max:
        return τ_end, 0
    q1, q2, q3 = math.log10(2.0), math.log10(dτ_max), math.log10(τ_end)
    tlimit = (q3 - q2) / q1
    nlimit = math.floor(tlimit)
    if tlimit - nlimit < EPS:
        return dτ_max, int(nlimit)
    nd = int(nlimit) + 1
    return 10.0 ** (q3 - q1 * nd), nd


def get_dtau_ndoubl(τ: np.ndarray, ϖ: np.ndarray, qp_μ: np.ndarray):
    """rt_kernel.jl:238-246: both maxima run over the WHOLE spectral axis."""
    mx = float(np.max(τ * ϖ))
    _, nd = doubling_number(min(mx, 0.001 * float(np.min(qp_μ))), mx)
    return τ / 2 ** nd, nd


@dataclass
class LayerInputs:
    """What constructCoreOpticalProperties + extractEffectiveProps + get_dtau_ndoubl deliver
    for all layers, in the native form of the C ABI (basis + weights instead of N×N×nSpec)."""
    τ: np.ndarray  # [S, Nz]
    ϖ: np.ndarray  # [S, Nz]
    zw: np.ndarray  # [K, S, Nz]
    ndoubl: np.ndarray  # [Nz] int32
    iface: np.ndarray  # [Nz] int32, 0..3 = ScatteringInterface_00/01/10/11
    τ_sum: np.ndarray  # [S, Nz+1]


def construct_layer_inputs(model: vSmartMOM_Model) -> LayerInputs:
    """constructCoreOpticalProperties (compEffectiveLayerProperties.jl:1-78) with the `+` of
    types.jl:632-678, createAero (:80-85), extractEffectiveProps (:88-111) and
    get_scattering_interface (rt_helper_functions.jl:8-27).  None of it depends on m."""
    S, Nz = model.τ_rayl.shape
    K = 1 + len(model.aerosol_optics)
    τ = np.empty((S, Nz))
    ϖ = np.empty((S, Nz))
    zw = np.zeros((K, S, Nz))
    for z in range(Nz):
        τz = model.τ_rayl[:, z].copy()
        ϖz = np.full(S, float(model.ϖ_Cabannes))
        w = np.zeros((K, S))
        w[0] = 1.0
        for a, aer in enumerate(model.aerosol_optics, start=1):
            τy = (1 - aer.fᵗ * aer.ω̃) * model.τ_aer[a - 1, z]
            ϖy = (1 - aer.fᵗ) * aer.ω̃ / (1 - aer.fᵗ * aer.ω̃)
            wx, wy = τz * ϖz, np.full(S, τy * ϖy)
            tot = wx + wy
            τn = τz + τy
            ϖn = tot / τn
            if np.all(wx == 0.0):
                w[:] = 0.0
                w[a] = 1.0
            elif not np.all(wy == 0.0):
                w *= (wx / tot)[None, :]
                w[a] = wy / tot
            τz, ϖz = τn, ϖn
        τn = τz + model.τ_abs[:, z]
        ϖz = (τz * ϖz) / τn
        τ[:, z], ϖ[:, z], zw[:, :, z] = τn, ϖz, w
    iface = np.zeros(Nz, dtype=np.int32)
    nd = np.zeros(Nz, dtype=np.int32)
    τ_sum = np.zeros((S, Nz + 1))
    prev = 0
    for z in range(Nz):
        scatter = bool(np.max(τ[:, z] * ϖ[:, z]) > 2 * EPS)
        if z == 0:
            prev = 3 if scatter else 0
        elif prev == 0:
            prev = 1 if scatter else 0
        else:
            prev = 3 if scatter else 2
        iface[z] = prev
        τ_sum[:, z + 1] = τ_sum[:, z] + 1.0 * τ[:, z]
        nd[z] = get_dtau_ndoubl(τ[:, z], ϖ[:, z], model.quad_points.qp_μ)[1]
    return LayerInputs(τ, ϖ, zw, nd, iface, τ_sum)


def z_bases(model: vSmartMOM_Model):
    """Z⁺⁺/Z⁻⁺ bases [M, K, N, N] (k = 0 Rayleigh, then aerosols) for m = 0..max_m-1."""
    pol, μ = model.params.polarization_type, model.quad_points.qp_μ
    greeks = [model.greek_rayleigh] + [a.greek_coefs for a in model.aerosol_optics]
    Zpp = np.array([[compute_Z_moments(pol, μ, g, m)[0] for g in greeks] for m in range(model.params.max_m)])
    Zmp = np.array([[compute_Z_moments(pol, μ, g, m)[1] for g in greeks] for m in range(model.params.max_m)])
    return Zpp, Zmp


def _abi_mats(Z: np.ndarray) -> np.ndarray:
    """[..., i, j] -> flat buffer with i fastest (Julia column-major per matrix)."""
    return np.ascontiguousarray(np.swapaxes(Z, -1, -2)).reshape(-1)


def view_nodes(model: vSmartMOM_Model) -> np.ndarray:
    """postprocessing_vza.jl:28: nearest quadrature node (1-based) per viewing zenith angle."""
    qp = model.quad_points.qp_μ
    return np.array([int(np.argmin(np.abs(qp - cosd(v)))) + 1 for v in model.params.vza], dtype=np.int32)


# ------------------------------------------------------------------------------------------
# rt_run
# ------------------------------------------------------------------------------------------


@dataclass
class SceneInputs:
    """Everything mom_scene_set consumes, already in ABI memory order."""
    N: int
    nStokes: int
    S: int
    Nz: int
    K: int
    M: int
    tau: np.ndarray
    varpi: np.ndarray
    zw: np.ndarray
    Zpp: np.ndarray
    Zmp: np.ndarray
    ndoubl: np.ndarray
    iface: np.ndarray
    tau_sum: np.ndarray
    albedo: float
    node: np.ndarray
    cos_mphi: np.ndarray
    sin_mphi: np.ndarray
    surf_kind: int = 0
    Rsurf: Optional[np.ndarray] = None        # ABI order [N, N, M]
    albedo_spec: Optional[np.ndarray] = None  # [S]

    def spectral_slice(self, lo: int, hi: int) -> "SceneInputs":
        """Shard [lo, hi) of the spectral axis.  ndoubl / iface stay the GLOBAL ones
        (rt_kernel.jl:241-242 takes maxima over the whole axis): SURVEY section 8e."""
        S, Nz, K = self.S, self.Nz, self.K
        sl = slice(lo, hi)
        return SceneInputs(self.N, self.nStokes, hi - lo, Nz, K, self.M,
                           np.ascontiguousarray(self.tau.reshape(Nz, S)[:, sl]).reshape(-1),
                           np.ascontiguousarray(self.varpi.reshape(Nz, S)[:, sl]).reshape(-1),
                           np.ascontiguousarray(self.zw.reshape(Nz, S, K)[:, sl, :]).reshape(-1),
                           self.Zpp, self.Zmp, self.ndoubl, self.iface,
                           np.ascontiguousarray(self.tau_sum.reshape(Nz + 1, S)[:, sl]).reshape(-1),
                           self.albedo, self.node, self.cos_mphi, self.sin_mphi, self.surf_kind, self.Rsurf,
                           None if self.albedo_spec is None else np.ascontiguousarray(self.albedo_spec[sl]))


def prepare_scene(model: vSmartMOM_Model) -> SceneInputs:
    """Host preparation of rt_run.jl:43-138 for every Fourier moment at once."""
    p, qp = model.params, model.quad_points
    L = construct_layer_inputs(model)
    Zpp, Zmp = z_bases(model)
    M = p.max_m
    S, Nz = L.τ.shape
    cm = np.array([[cosd(m * a) for a in p.vaz] for m in range(M)])
    sm = np.array([[sind(m * a) for a in p.vaz] for m in range(M)])
    brdf = p.brdf if p.brdf is not None else LambertianSurfaceScalar(float(p.brdf_albedo))
    kind, Rs, alb = surface_inputs(brdf, p.polarization_type, qp.qp_μ, M, S)
    return SceneInputs(
        surf_kind=kind, Rsurf=None if Rs is None else _abi_mats(Rs), albedo_spec=alb,
        N=len(qp.qp_μN), nStokes=p.polarization_type.n, S=S, Nz=Nz, K=L.zw.shape[0], M=M,
        tau=np.ascontiguousarray(L.τ.T).reshape(-1), varpi=np.ascontiguousarray(L.ϖ.T).reshape(-1),
        zw=np.ascontiguousarray(L.zw.transpose(2, 1, 0)).reshape(-1),  # [z][n][k]
        Zpp=_abi_mats(Zpp), Zmp=_abi_mats(Zmp), ndoubl=L.ndoubl, iface=L.iface,
        tau_sum=np.ascontiguousarray(L.τ_sum.T).reshape(-1),
        albedo=float(brdf.albedo) if isinstance(brdf, LambertianSurfaceScalar) else 0.0, node=view_nodes(model),
        cos_mphi=cm.reshape(-1), sin_mphi=sm.reshape(-1))


def make_handle(model: vSmartMOM_Model, S: Optional[int] = None, float_type: str = "Float64") -> _lib.Handle:
    """float_type: the reference's `float_type` (parameters_from_yaml.jl:160): "Float64" or "Float32"."""
    p, qp = model.params, model.quad_points
    if not isinstance(p.architecture, MI355X):
        raise TypeError("this package only executes on Architectures.MI355X (no CPU path)")
    h = _lib.Handle(len(qp.qp_μN), p.polarization_type.n, S if S is not None else model.τ_rayl.shape[0], p.max_m,
                    device=p.architecture.device, dtype={"Float64": 0, "Float32": 1}[float_type])
    h.set_streams(qp.qp_μN, qp.wt_μN, qp.iμ0, qp.μ0, p.polarization_type.I0, p.polarization_type.D,
                  p.strict_reference_indexing)
    return h


def run_scene(h: _lib.Handle, sc: SceneInputs):
    h.scene_set(sc.Nz, sc.K, sc.M, sc.tau, sc.varpi, sc.zw, sc.Zpp, sc.Zmp, sc.ndoubl, sc.iface, sc.tau_sum,
                sc.albedo, sc.node, sc.cos_mphi, sc.sin_mphi)
    if sc.surf_kind != 0:
        h.scene_set_surface(sc.surf_kind, sc.M, sc.Rsurf, sc.albedo_spec)
    h.rt_run()
    return h.get_RT()


@dataclass
class ScenePartial:
    """Partials of the hot path's inputs with respect to ONE parameter, in the reference's array shapes -- what the Dual
    numbers of model_from_parameters carry into rt_run (rt_run.jl:89-96).  None = no dependence."""
    dτ: Optional[np.ndarray] = None             # [nSpec, Nz]
    dϖ: Optional[np.ndarray] = None             # [nSpec, Nz]
    dzw: Optional[np.ndarray] = None            # [K, nSpec, Nz] weights of the phase-matrix bases
    dZpp: Optional[np.ndarray] = None           # [M, K, N, N] partials of the bases themselves (aerosol microphysics)
    dZmp: Optional[np.ndarray] = None
    dalbedo: float = 0.0                        # LambertianSurfaceScalar
    dRsurf: Optional[np.ndarray] = None         # [M, N, N] BRDF surfaces (with the factor 2 of m = 0 like Rsurf)
    dalbedo_spec: Optional[np.ndarray] = None   # [nSpec] LambertianSurfaceLegendre


def scene_set_partials(h: _lib.Handle, sc: SceneInputs, partials: Sequence[ScenePartial]):
    """Pack the partials into the ABI layout of mom_scene_set_partials (partial index slowest) and upload them."""
    P = len(partials)

    def pack(get, conv):
        xs = [get(p) for p in partials]
        if all(x is None for x in xs):
            return None
        ref = next(x for x in xs if x is not None)
        return np.concatenate([conv(np.zeros_like(ref) if x is None else np.asarray(x, dtype=np.float64)) for x in xs])

    col = lambda a: np.ascontiguousarray(a.T).reshape(-1)
    h.scene_set_partials(
        P, dtau=pack(lambda p: p.dτ, col), dvarpi=pack(lambda p: p.dϖ, col),
        dzw=pack(lambda p: p.dzw, lambda a: np.ascontiguousarray(a.transpose(2, 1, 0)).reshape(-1)),
        dZpp=pack(lambda p: p.dZpp, _abi_mats), dZmp=pack(lambda p: p.dZmp, _abi_mats),
        dalbedo=np.array([float(p.dalbedo) for p in partials]) if sc.surf_kind == 0 and P else None,
        dRsurf=pack(lambda p: p.dRsurf, _abi_mats) if sc.surf_kind == 1 else None,
        dalbedo_spec=pack(lambda p: p.dalbedo_spec, lambda a: a.reshape(-1)) if sc.surf_kind == 2 else None)


def rt_run_dual(model: vSmartMOM_Model, partials: Sequence[ScenePartial], workspace_mb: int = 0, full: bool = False):
    """rt_run on ForwardDiff.Dual inputs (rt_run.jl:41-230 with FT_dual element types): returns
    (R_SFI, T_SFI [nVza, nStokes, nSpec], dR_SFI, dT_SFI [P, nVza, nStokes, nSpec]); full = True: the rest of the reference's
    tuple as well -- ((R, T, hdr, bhr_uw, bhr_dw), (dR, dT, dhdr, dbhr_uw, dbhr_dw)), bhr_* [nStokes, nSpec]."""
    sc = prepare_scene(model)
    with make_handle(model) as h:
        if workspace_mb:
            h.set_option(_lib.MOM_OPT_DUAL_WORKSPACE_MB, int(workspace_mb))
        scene_set(h, sc)
        scene_set_partials(h, sc, partials)
        h.rt_run_dual()
        R, T = h.get_RT()
        if len(partials) == 0:
            return R, T, np.zeros((0,) + R.shape), np.zeros((0,) + T.shape)
        dR, dT = h.get_RT_partials()
        if full:
            hdr, up, dw = h.get_hdr()
            dhdr, dup, ddw = h.get_hdr_partials()
            return (R, T, hdr, up, dw), (dR, dT, dhdr, dup, ddw)
    return R, T, dR, dT


def run_scene_device_optics(h: _lib.Handle, model: vSmartMOM_Model, upload_tau_abs: bool = True):
    """The same run with the layer optics assembled on the GPU (mom_scene_set_optics): the host hands over the
    INGREDIENTS (τ_rayl, aerosol columns, Z bases) and the gas absorption is the handle's resident τ_abs table --
    uploaded here from model.τ_abs, or (upload_tau_abs=False) already accumulated by mom_voigt_tau_abs."""
    p, qp = model.params, model.quad_points
    Zpp, Zmp = z_bases(model)
    M = p.max_m
    S, Nz = model.τ_rayl.shape
    if upload_tau_abs:
        h.absorption_set(model.τ_abs)
    cm = np.array([[cosd(m * a) for a in p.vaz] for m in range(M)])
    sm = np.array([[sind(m * a) for a in p.vaz] for m in range(M)])
    brdf = p.brdf if p.brdf is not None else LambertianSurfaceScalar(float(p.brdf_albedo))
    kind, Rs, alb = surface_inputs(brdf, p.polarization_type, qp.qp_μ, M, S)
    albedo = float(brdf.albedo) if isinstance(brdf, LambertianSurfaceScalar) else 0.0
    h.scene_set_optics(Nz, M, model.τ_rayl, model.ϖ_Cabannes, model.τ_aer, [a.ω̃ for a in model.aerosol_optics],
                       [a.fᵗ for a in model.aerosol_optics], _abi_mats(Zpp), _abi_mats(Zmp), albedo,
                       view_nodes(model), cm.reshape(-1), sm.reshape(-1))
    if kind != 0:
        h.scene_set_surface(kind, M, None if Rs is None else _abi_mats(Rs), alb)
    h.rt_run()
    return h.get_RT()


def rt_run(model: vSmartMOM_Model, i_band: int = 1):
    """rt_run(model; i_band) (rt_run.jl:19-21 -> :41-230), SFI = true, noRS.  Returns the
    reference's 7-tuple (rt_run.jl:226):
        (R_SFI, T_SFI, ieR_SFI, ieT_SFI, hdr, bhr_uw[1,:], bhr_dw[1,:])
    R/T/hdr are [nVza, nStokes, nSpec]; the inelastic terms are zero for noRS."""
    sc = prepare_scene(model)
    with make_handle(model) as h:
        R, T = run_scene(h, sc)
        hdr, up, dw = h.get_hdr()
    return R, T, np.zeros_like(R), np.zeros_like(T), hdr, up[0], dw[0]


def scene_set(h: _lib.Handle, sc: SceneInputs):
    h.scene_set(sc.Nz, sc.K, sc.M, sc.tau, sc.varpi, sc.zw, sc.Zpp, sc.Zmp, sc.ndoubl, sc.iface, sc.tau_sum,
                sc.albedo, sc.node, sc.cos_mphi, sc.sin_mphi)
    if sc.surf_kind != 0:
        h.scene_set_surface(sc.surf_kind, sc.M, sc.Rsurf, sc.albedo_spec)


def rt_run_test_ms(sensor_levels, model: vSmartMOM_Model, i_band: int = 1):
    """rt_run_test_ms(RS_type::noRS, sensor_levels, model, iBand) (rt_run_multisensor.jl:14-191): sensors inside the
    atmosphere, labelled from the top (0 = the TOA/BOA pair, L = below layer L), all sharing the view angles.
    Returns the reference's 4-tuple (uwJ, dwJ, uwieJ, dwieJ): lists over the sensors of [nVza, nStokes, nSpec] arrays;
    the inelastic terms are zero for noRS."""
    sc = prepare_scene(model)
    with make_handle(model) as h:
        scene_set(h, sc)
        uw, dw = h.rt_run_multisensor(sensor_levels)
    z = [np.zeros_like(u) for u in uw]
    return list(uw), list(dw), z, [x.copy() for x in z]


def rt_run_operators(model: vSmartMOM_Model):
    """The same run replayed operator by operator through the op-level ABI, exactly in the
    order of rt_run.jl:125-215 / rt_kernel.jl:173-235 (what a Julia shim overloading
    elemental!/doubling!/interaction! one by one would execute)."""
    p, qp = model.params, model.quad_points
    pol = p.polarization_type
    L = construct_layer_inputs(model)
    Zpp, Zmp = z_bases(model)
    S, Nz = L.τ.shape
    nV = len(p.vza)
    R = np.zeros(nV * pol.n * S)  # ABI order [nVza, nStokes, nSpec], v fastest
    T = np.zeros(nV * pol.n * S)
    nodes = view_nodes(model)
    brdf = p.brdf if p.brdf is not None else LambertianSurfaceScalar(float(p.brdf_albedo))
    if not isinstance(brdf, LambertianSurfaceScalar):
        raise NotImplementedError("the operator-level surface entry is mom_surface_lambertian (LambertianSurfaceScalar); "
                                  "use rt_run / run_scene for %s" % type(brdf).__name__)
    albedo = float(brdf.albedo)
    with make_handle(model) as h:
        for m in range(p.max_m):
            weight = 0.5 if m == 0 else 1.0
            for z in range(Nz):
                dτ, nd = get_dtau_ndoubl(L.τ[:, z], L.ϖ[:, z], qp.qp_μ)
                expk = np.exp(-dτ / qp.μ0)
                Zp = np.einsum("ks,kij->sij", L.zw[:, :, z], Zpp[m])
                Zm = np.einsum("ks,kij->sij", L.zw[:, :, z], Zmp[m])
                h.elemental(m, nd, L.τ_sum[:, z], dτ, L.ϖ[:, z], _abi_mats(Zp), _abi_mats(Zm), S)
                h.doubling(nd, expk)
                if z == 0:
                    h.copy_added_to_composite()
                else:
                    h.interaction(int(L.iface[z]))
            h.surface_lambertian(m, albedo, L.τ_sum[:, -1])
            h.interaction(int(L.iface[-1]), with_surface_layer=True)
            h.postprocess(m, nodes, p.vaz, weight, R, T)  # postprocessing_vza! (R_SFI += ..., T_SFI += ...)
    shp = (S, pol.n, nV)
    return np.transpose(R.reshape(shp), (2, 1, 0)).copy(), np.transpose(T.reshape(shp), (2, 1, 0)).copy()


# ------------------------------------------------------------------------------------------
# rotational-Raman scattering: rt_run(RS_type::RRS, model, iBand)
# ------------------------------------------------------------------------------------------


@dataclass
class RRS:
    """InelasticScattering.RRS (src/Inelastic/types.jl:13-33): the fields the CoreRT hot path reads.  They are produced by
    getRamanSSProp! (src/CoreRT/tools/raman_atmo_prop.jl:57-73) from the N2/O2 molecular constants, which stays host-side
    set-up (out of scope, like the Mie code); synthetic line lists for benchmarks: scenes.raman_lines.

    rrs_strict_reference: the reference's RRS text executed as written (True: the default here and in the Julia shim
    integration/MomCoreRT.jl -- a drop-in reproduces the reference) or with the five documented corrections D1..D5 (False:
    what scenes.scene_C5 and therefore every C5 benchmark number of this repository selects EXPLICITLY, and says so) --
    DESIGN.md section 7, include/momcore.h mom_rrs_set.  As written the path is barely usable: any scene with a 00 / 01 / 10
    interface raises (D4: MethodError in the reference -> MOM_EUNSUPPORTED here) and with realistic line counts expk underflows
    in the first doubling step (D1: expk -> expk^(2^nRaman))."""
    greek_raman: GreekCoefs
    ϖ_Cabannes: float          # elastic (Cabannes) fraction of Rayleigh scattering: the Rayleigh ϖ of the elastic layer optics
    ϖ_λ1λ0: np.ndarray         # [nRaman]
    i_λ1λ0: np.ndarray         # [nRaman] grid offsets n₀ - n₁
    rrs_strict_reference: bool = True

    @property
    def n_Raman(self):
        return len(self.i_λ1λ0)


def fscatt_rayleigh(model: vSmartMOM_Model) -> np.ndarray:
    """fScattRayleigh of constructCoreOpticalProperties (compEffectiveLayerProperties.jl:58): rayl.τ ./ combo.τ, combo =
    Rayleigh + aerosols before the gas absorption is merged.  [nSpec, Nz]."""
    combo = model.τ_rayl.astype(np.float64).copy()
    for a, aer in enumerate(model.aerosol_optics):
        combo = combo + ((1 - aer.fᵗ * aer.ω̃) * model.τ_aer[a])[None, :]
    return model.τ_rayl / combo


def raman_z(RS_type: RRS, model: vSmartMOM_Model):
    """computeRamanZλ! (src/Inelastic/inelastic_helper.jl:457-464) for m = 0..max_m-1: Z⁺⁺_λ₁λ₀, Z⁻⁺_λ₁λ₀ [M, N, N]."""
    pol, μ = model.params.polarization_type, model.quad_points.qp_μ
    Z = [compute_Z_moments(pol, μ, RS_type.greek_raman, m) for m in range(model.params.max_m)]
    return np.array([z[0] for z in Z]), np.array([z[1] for z in Z])


def _with_cabannes(RS_type: RRS, model: vSmartMOM_Model) -> vSmartMOM_Model:
    import dataclasses
    return dataclasses.replace(model, ϖ_Cabannes=float(RS_type.ϖ_Cabannes))  # compEffectiveLayerProperties.jl:27


def rt_run_rrs(RS_type: RRS, model: vSmartMOM_Model, i_band: int = 1, kernels: Optional[int] = None):
    """rt_run(RS_type::RRS, model, iBand) (rt_run.jl:41-230), SFI = true.  Returns the reference's 7-tuple (rt_run.jl:226):
        (R_SFI, T_SFI, ieR_SFI, ieT_SFI, hdr, bhr_uw[1,:], bhr_dw[1,:])
    R/T/ieR/ieT/hdr are [nVza, nStokes, nSpec]; every surface type of `params.brdf`.  With RS_type.rrs_strict_reference =
    True the run raises MomError (MOM_EUNSUPPORTED) for scenes with a 00 / 01 / 10 scattering interface, as the reference's
    text does (D4); False runs them."""
    S = model.τ_rayl.shape[0]
    return rt_run_rrs_window(RS_type, model, 0, S, kernels=kernels)


def rt_run_rrs_window(RS_type: RRS, model: vSmartMOM_Model, lo: int, hi: int, window=None, kernels: Optional[int] = None):
    """The owned slice [lo, hi) of rt_run(::RRS): runs the window `window` = (wlo, whi) ⊇ [lo, hi) (default: [lo, hi) widened
    by max |i_λ₁λ₀| and clipped, sharding.rrs_window) with the GLOBAL ndoubl / interface codes and returns the 7-tuple
    restricted to the owned points (last axis hi - lo).  kernels: MOM_OPT_RRS_KERNELS mask (None: the library's default forms)."""
    model = _with_cabannes(RS_type, model)
    sc_full = prepare_scene(model)
    S = sc_full.S
    if window is None:
        H = int(np.max(np.abs(RS_type.i_λ1λ0)))
        window = (max(0, lo - H), min(S, hi + H))
    wlo, whi = window
    sc = sc_full if (wlo, whi) == (0, S) else sc_full.spectral_slice(wlo, whi)
    Zr_pp, Zr_mp = raman_z(RS_type, model)
    fs = fscatt_rayleigh(model)[wlo:whi]
    with make_handle(model, S=whi - wlo) as h:
        h.set_option(_lib.MOM_OPT_STRIP_PAD, 0)
        if kernels is not None:
            h.set_option(_lib.MOM_OPT_RRS_KERNELS, int(kernels))
        h.rrs_set(RS_type.i_λ1λ0, RS_type.ϖ_λ1λ0, RS_type.rrs_strict_reference)
        h.rrs_set_shard(S, wlo, lo - wlo, hi - wlo)
        scene_set(h, sc)
        h.scene_set_rrs(np.ascontiguousarray(fs.T), _abi_mats(Zr_pp), _abi_mats(Zr_mp))
        h.rt_run_rrs()
        R, T, ieR, ieT = h.get_RT_rrs()[:4]
        hdr, up, dw = h.get_hdr_rrs()
    own = slice(lo - wlo, hi - wlo)
    return R[..., own], T[..., own], ieR[..., own], ieT[..., own], hdr[..., own], up[0][..., own], dw[0][..., own]


def rt_run_rrs_sharded(RS_type: RRS, model: vSmartMOM_Model, dist, device=None):
    """rt_run(::RRS) over the ranks of an initialised torch.distributed group (one process per GPU): every rank runs its
    window (sharding.rrs_window) and one all-gather assembles the 7-tuple on every rank.  No exchange during the run."""
    from . import sharding
    S = model.τ_rayl.shape[0]
    lo, hi, wlo, whi = sharding.rrs_window(S, dist.get_world_size(), dist.get_rank(), RS_type.i_λ1λ0)
    p = model.params
    nV, nS = len(p.vza), p.polarization_type.n
    if hi > lo:
        loc = list(rt_run_rrs_window(RS_type, model, lo, hi, (wlo, whi)))
    else:
        loc = [np.zeros((nV, nS, 0))] * 5 + [np.zeros((0,))] * 2
    return tuple(sharding.gather_spectra(loc, S, dist, device))
