"""
oracle/momref.py -- TEST INFRASTRUCTURE ONLY (numpy twin of the CPU oracle).

A CPU restatement, in plain numpy, of the reference's (vSmartMOM.jl) elastic
Matrix-Operator hot path: streams -> Z moments -> layer optics -> per layer
{elemental, doubling, interaction} -> Lambertian surface -> post-processing.
Every function cites the reference file:line it follows (paths relative to the
reference's root).  Nothing in the product path (radiativetransfer.jl_amd/,
csrc/) may import this file; only tests/, __graft_entry__.smoke() and the
cpu_baseline leg of bench.py do, and only as the checker.

Pinned against the reference's own known-answer tests (test/test_CoreRT.jl:3-83:
6SV1 and Natraj tables, committed as data under tests/golden/) by
tests/test_oracle_reference_tables.py.  Per-op values are NOT pinned by anything in
the reference (no live Julia here) -- see DESIGN.md "oracle pinning".

Layout used inside this twin: batched matrices are numpy arrays M[n, i, j]
(spectral index first, then row, column); sources are J[n, i].  `to_abi`/`from_abi`
convert to the reference's column-major [i, j, n] memory order that the C-ABI uses.

All the reference's index quirks are reproduced when
`strict_reference_indexing=True` (the default).
"""
from __future__ import annotations

import math
from dataclasses import dataclass, field
from typing import List, Optional, Sequence

import numpy as np

# --------------------------------------------------------------------------------------
# layout helpers
# --------------------------------------------------------------------------------------

def to_abi(M: np.ndarray) -> np.ndarray:
    """M[n,i,j] -> flat buffer in Julia column-major [i,j,n] order (i fastest)."""
    if M.ndim == 3:
        return np.ascontiguousarray(np.transpose(M, (0, 2, 1))).reshape(-1)
    if M.ndim == 2:  # J[n,i] -> [i,1,n]
        return np.ascontiguousarray(M).reshape(-1)
    raise ValueError("bad rank")


def from_abi(buf: np.ndarray, N: int, S: int, vec: bool = False) -> np.ndarray:
    buf = np.asarray(buf)
    if vec:
        return buf.reshape(S, N).copy()
    return np.transpose(buf.reshape(S, N, N), (0, 2, 1)).copy()


# --------------------------------------------------------------------------------------
# polarization types  (src/Scattering/types.jl:82-123)
# --------------------------------------------------------------------------------------

@dataclass
class PolType:
    n: int
    D: np.ndarray
    I0: np.ndarray


def Stokes_I():
    return PolType(1, np.array([1.0]), np.array([1.0]))


def Stokes_IQU():
    return PolType(3, np.array([1.0, 1.0, -1.0]), np.array([1.0, 0.0, 0.0]))


def Stokes_IQUV():
    return PolType(4, np.array([1.0, 1.0, -1.0, -1.0]), np.array([1.0, 0.0, 0.0, 0.0]))


def pol_from_n(n: int) -> PolType:
    return {1: Stokes_I, 3: Stokes_IQU, 4: Stokes_IQUV}[n]()


# --------------------------------------------------------------------------------------
# quadrature (third-party FastGaussQuadrature.jl is not vendored in the reference; the
# published Gauss-Legendre / Gauss-Radau rules are restated via scipy's Jacobi roots)
# --------------------------------------------------------------------------------------

def gausslegendre(n: int):
    x, w = np.polynomial.legendre.leggauss(n)
    return x, w


def gaussradau(n: int):
    """n-point Gauss-Radau rule on [-1,1] with the fixed node at x=-1 (first node),
    as FastGaussQuadrature.gaussradau returns it (call site rt_set_streams.jl:115)."""
    from scipy.special import roots_jacobi

    if n == 1:
        return np.array([-1.0]), np.array([2.0])
    xi, vi = roots_jacobi(n - 1, 0.0, 1.0)  # weight (1+x)
    x = np.concatenate([[-1.0], xi])
    w = np.concatenate([[2.0 / n ** 2], vi / (1.0 + xi)])
    return x, w


def _unique_keep_order(a: Sequence[float]) -> np.ndarray:
    """Julia `unique`: first occurrences, original order (exact float comparison)."""
    seen = set()
    out = []
    for v in a:
        fv = float(v)
        if fv not in seen:
            seen.add(fv)
            out.append(fv)
    return np.array(out, dtype=np.float64)


def _deg2rad_dd(x):
    """Julia's deg2rad_ext (base/special/trig.jl): x * m, m = Float64(pi/180), as an unevaluated sum hi + lo."""
    m, mh, ml = 0.017453292519943295, 0.01745329238474369, 1.3519960527851425e-10
    u = 134217729.0 * x
    xh = u - (u - x)
    xl = x - xh
    hi = m * x
    lo = xh * ml + (xl * mh + ((xh * mh - hi) + xl * ml))
    return hi, lo


def _ksin(deg):
    """sin_kernel(::DoubleFloat64) = fdlibm __kernel_sin(x, y, 1) on the double-double radians of |deg| <= 45."""
    x, y = _deg2rad_dd(deg)
    z = x * x
    v = z * x
    r = 8.33333333332248946124e-03 + z * (-1.98412698298579493134e-04 + z * (2.75573137070700676789e-06 + z * (
        -2.50507602534068634195e-08 + z * 1.58969099521155010221e-10)))
    return x - ((z * (0.5 * y - v * r) - y) - v * -1.66666666666666324348e-01)


def _kcos(deg):
    """cos_kernel(::DoubleFloat64) = fdlibm __kernel_cos(x, y)."""
    x, y = _deg2rad_dd(deg)
    z = x * x
    w = z * z
    r = z * (4.16666666666666019037e-02 + z * (-1.38888888888741095749e-03 + z * 2.48015872894767294178e-05)) + \
        w * w * (-2.75573143513906633035e-07 + z * (2.08757232129817482790e-09 + z * -1.13596475577881948265e-11))
    hz = 0.5 * z
    w = 1.0 - hz
    return w + (((1.0 - w) - hz) + (z * r - x * y))


def cosd(x):
    """Julia cosd (base/special/trig.jl): reduction in degrees, kernels on a double-double argument; exact at the
    multiples of 30 and 90 degrees (cosd(60) == 0.5)."""
    x = np.asarray(x, dtype=np.float64)
    rx = np.abs(np.fmod(x, 360.0))
    with np.errstate(all="ignore"):
        out = np.select([rx <= 45.0, rx < 135.0, rx <= 225.0, rx < 315.0],
                        [_kcos(rx), _ksin(90.0 - rx), -_kcos(180.0 - rx), _ksin(rx - 270.0)], _kcos(360.0 - rx))
    return out


def sind(x):
    x = np.asarray(x, dtype=np.float64)
    rx = np.fmod(x, 360.0)
    arx = np.abs(rx)
    sg = np.copysign(1.0, rx)
    with np.errstate(all="ignore"):
        out = np.select([rx == 0.0, arx < 45.0, arx <= 135.0, arx == 180.0, arx < 225.0, arx <= 315.0],
                        [rx, _ksin(rx), np.copysign(_kcos(90.0 - arx), rx), np.copysign(0.0, rx), _ksin((180.0 - arx) * sg),
                         -np.copysign(_kcos(270.0 - arx), rx)], _ksin(rx - np.copysign(360.0, rx)))
    return out


def nearest_point(arr, f) -> int:
    """rt_helper_functions.jl:60 -- 0-based index of first minimum of |arr-f|."""
    return int(np.argmin(np.abs(np.asarray(arr) - f)))


@dataclass
class QuadPoints:
    """types.jl:456-473.  imu0 / imu0Nstart are 1-BASED as in the reference."""
    mu0: float
    imu0: int
    imu0Nstart: int
    qp_mu: np.ndarray
    wt_mu: np.ndarray
    qp_muN: np.ndarray
    wt_muN: np.ndarray
    Nquad: int


def _finish_streams(qp_mu, wt_mu, mu0, n):
    imu0 = nearest_point(qp_mu, mu0) + 1
    qp_muN = np.repeat(qp_mu, n)
    wt_muN = np.repeat(wt_mu, n)
    i_start = n * (imu0 - 1) + 1
    return QuadPoints(float(mu0), imu0, i_start, qp_mu, wt_mu, qp_muN, wt_muN, len(qp_mu))


def rt_set_streams(quadtype: str, Ltrunc: int, sza: float, vza: Sequence[float], nStokes: int) -> QuadPoints:
    """rt_set_streams.jl:24-50 (GaussQuadHemisphere), :63-87 (GaussQuadFullSphere),
    :101-170 (RadauQuad)."""
    vza = np.asarray(vza, dtype=np.float64)
    Nquad = (Ltrunc + 1) // 2
    mu0 = float(cosd(sza))
    if quadtype == "GaussQuadHemisphere":
        xi, w = gausslegendre(Nquad)  # gauleg(Nquad,0,1): mie_helper_functions.jl:177-182
        qp = 0.5 * xi + 0.5
        wt = w * 0.5
        qp_mu = _unique_keep_order(list(qp) + list(cosd(vza)) + [mu0])
        wt_mu = np.concatenate([wt, np.zeros(len(qp_mu) - len(wt))])
        return _finish_streams(qp_mu, wt_mu, mu0, nStokes)
    if quadtype == "GaussQuadFullSphere":
        xi, w = gausslegendre(2 * Nquad)
        qp_mu = _unique_keep_order(list(xi[Nquad:]) + list(cosd(vza)) + [mu0])
        wt_mu = np.concatenate([w[Nquad:], np.zeros(len(qp_mu) - Nquad)])
        return _finish_streams(qp_mu, wt_mu, mu0, nStokes)
    if quadtype == "RadauQuad":
        tq, tw = gaussradau(Nquad)
        q0 = -tq[::-1]
        w0 = tw[::-1]
        if mu0 in q0:
            qp = (1.0 + q0) / 2.0
            wt = w0.copy()
        else:
            qp = np.zeros(2 * Nquad)
            wt = np.zeros(2 * Nquad)
            for i in range(Nquad):
                qp[i] = (mu0 + mu0 * q0[i]) / 2
                wt[i] = mu0 * w0[i] / 2
                qp[Nquad + i] = ((1 + mu0) + (1 - mu0) * q0[i]) / 2
                wt[Nquad + i] = (1 - mu0) * w0[i] / 2
        qp_mu = _unique_keep_order(list(qp) + list(cosd(vza)))
        wt_mu = np.concatenate([wt, np.zeros(len(qp_mu) - len(wt))])
        return _finish_streams(qp_mu, wt_mu, mu0, nStokes)
    raise ValueError(quadtype)


# --------------------------------------------------------------------------------------
# Greek coefficients, generalized spherical functions, Z moments
# --------------------------------------------------------------------------------------

@dataclass
class GreekCoefs:
    """src/Scattering/types.jl:198-211"""
    alpha: np.ndarray
    beta: np.ndarray
    gamma: np.ndarray
    delta: np.ndarray
    epsilon: np.ndarray
    zeta: np.ndarray


def get_greek_rayleigh(depol: float) -> GreekCoefs:
    """mie_helper_functions.jl:237-251"""
    dpl_p = (1 - depol) / (1 + depol / 2)
    dpl_r = (1 - 2 * depol) / (1 - depol)
    a = np.array([0.0, 0.0, 3 * dpl_p])
    b = np.array([1.0, 0.0, 0.5 * dpl_p])
    g = np.array([0.0, 0.0, dpl_p * math.sqrt(1.5)])
    d = np.array([0.0, dpl_p * dpl_r * 1.5, 0.0])
    e = np.zeros(3)
    z = np.zeros(3)
    return GreekCoefs(a, b, g, d, e, z)


def compute_associated_legendre_PRT(mu: np.ndarray, Lmax: int):
    """legendre_functions.jl:17-178.  Returns P,R,T of shape [len(mu),Lmax,Lmax]
    (index [imu, l, m], 0-based l and m), T already sign-flipped as the reference
    returns `-T`."""
    mu = np.asarray(mu, dtype=np.float64)
    n = len(mu)
    P = np.zeros((n, Lmax, Lmax))
    R = np.zeros((n, Lmax, Lmax))
    T = np.zeros((n, Lmax, Lmax))
    smu = np.sqrt(1.0 - mu ** 2)
    cmu = mu
    for m in range(Lmax):
        for l in range(m, Lmax):
            if m == 0:
                if l == 0:
                    P[:, l, m] = 1
                elif l == 1:
                    P[:, l, m] = cmu
                elif l == 2:
                    P[:, l, m] = 0.5 * (3.0 * cmu * cmu - 1.0)
                    R[:, l, m] = 0.5 * math.sqrt(1.5) * smu * smu
                else:
                    P[:, l, m] = (P[:, l - 1, m] * (2 * l - 1) * cmu - P[:, l - 2, m] * (l - 1)) / l
                    Y = math.sqrt((l + 1) * (l - 3))
                    X = math.sqrt(l * l - 4)
                    R[:, l, m] = (R[:, l - 1, m] * (2 * l - 1) * cmu - R[:, l - 2, m] * Y) / X
            elif m == 1:
                if l == 1:
                    P[:, l, m] = math.sqrt(0.5) * smu
                elif l == 2:
                    m1 = math.sqrt(1 / 6)
                    cA = 3 * cmu * smu
                    cB = math.sqrt(1.5) * smu
                    P[:, l, m] = m1 * cA
                    R[:, l, m] = -m1 * cmu * cB
                    T[:, l, m] = m1 * cB
                else:
                    m1 = math.sqrt((l - 1) / (l + 1))
                    m2 = m1 * math.sqrt((l - 2) / l)
                    Y = l - 1 + m
                    X = l - m
                    P[:, l, m] = (m1 * P[:, l - 1, m] * (2 * l - 1) * cmu - m2 * P[:, l - 2, m] * Y) / X
                    Z = (2 * m * (2 * l - 1)) / (l * (l - 1))
                    Y = ((l + m - 1) / (l - 1)) * math.sqrt((l - 3) * (l + 1))
                    X = ((l - m) / l) * math.sqrt(l * l - 4)
                    R[:, l, m] = (m1 * R[:, l - 1, m] * (2 * l - 1) * cmu - m2 * R[:, l - 2, m] * Y
                                  + m1 * T[:, l - 1, m] * Z) / X
                    T[:, l, m] = (m1 * T[:, l - 1, m] * (2 * l - 1) * cmu - m2 * T[:, l - 2, m] * Y
                                  + m1 * R[:, l - 1, m] * Z) / X
            else:
                if l == m:
                    fact1 = np.ones(n)
                    fact2 = np.ones(n)
                    sfull = smu
                    shalf = sfull / 2
                    for i in range(1, m + 1):
                        fact1 = fact1 * ((2 * i - 1) * sfull) / math.sqrt(i * (i + m))
                        if i > 2:
                            fact2 = fact2 * shalf * math.sqrt((m + i) / (i - 2))
                        else:
                            fact2 = fact2 * shalf
                    big = smu > 1e-8
                    safe = np.where(big, smu, 1.0)
                    Aii = np.where(big, fact2 * (1.0 + cmu * cmu) / (safe * safe), 0.5 if m == 2 else 0.0)
                    Aij = np.where(big, fact2 * (2 * cmu) / (safe * safe), 0.5 if m == 2 else 0.0)
                    P[:, l, m] = fact1
                    R[:, l, m] = Aii
                    T[:, l, m] = -Aij
                elif l == m + 1:
                    m1 = math.sqrt(1 / (l + m))
                    X = l - m
                    P[:, l, m] = (m1 * P[:, l - 1, m] * (2 * l - 1) * cmu) / X
                    Z = (2 * m * (2 * l - 1)) / (l * (l - 1))
                    X = ((l - m) / l) * math.sqrt(l * l - 4)
                    R[:, l, m] = (m1 * R[:, l - 1, m] * (2 * l - 1) * cmu + m1 * T[:, l - 1, m] * Z) / X
                    T[:, l, m] = (m1 * T[:, l - 1, m] * (2 * l - 1) * cmu + m1 * R[:, l - 1, m] * Z) / X
                else:
                    m1 = math.sqrt((l - m) / (l + m))
                    m2 = m1 * math.sqrt((l - m - 1) / (l + m - 1))
                    Y = l - 1 + m
                    X = l - m
                    P[:, l, m] = (m1 * P[:, l - 1, m] * (2 * l - 1) * cmu - m2 * P[:, l - 2, m] * Y) / X
                    Z = (2 * m * (2 * l - 1)) / (l * (l - 1))
                    Y = ((l + m - 1) / (l - 1)) * math.sqrt((l - 3) * (l + 1))
                    X = ((l - m) / l) * math.sqrt(l * l - 4)
                    R[:, l, m] = (m1 * R[:, l - 1, m] * (2 * l - 1) * cmu - m2 * R[:, l - 2, m] * Y
                                  + m1 * T[:, l - 1, m] * Z) / X
                    T[:, l, m] = (m1 * T[:, l - 1, m] * (2 * l - 1) * cmu - m2 * T[:, l - 2, m] * Y
                                  + m1 * R[:, l - 1, m] * Z) / X
    return P, R, -T


def _Pi_matrices(n: int, P, R, T, l: int, m: int) -> np.ndarray:
    """construct_Π_matrix, mie_helper_functions.jl:287-323 (sign_change=false branch);
    l, m are 0-based array indices here.  Returns [nmu, n, n]."""
    nmu = P.shape[0]
    Pi = np.zeros((nmu, n, n))
    Pi[:, 0, 0] = P[:, l, m]
    if n >= 3:
        Pi[:, 1, 1] = R[:, l, m]
        Pi[:, 1, 2] = -T[:, l, m]
        Pi[:, 2, 1] = -T[:, l, m]
        Pi[:, 2, 2] = R[:, l, m]
    if n == 4:
        Pi[:, 3, 3] = P[:, l, m]
    return Pi


def _B_matrix(n: int, g: GreekCoefs, l: int) -> np.ndarray:
    """construct_B_matrix, mie_helper_functions.jl:334-348 (l 0-based index)."""
    if n == 1:
        return np.array([[g.beta[l]]])
    if n == 3:
        return np.array([[g.beta[l], g.gamma[l], 0], [g.gamma[l], g.alpha[l], 0], [0, 0, g.zeta[l]]])
    return np.array([[g.beta[l], g.gamma[l], 0, 0], [g.gamma[l], g.alpha[l], 0, 0],
                     [0, 0, g.zeta[l], g.epsilon[l]], [0, 0, -g.epsilon[l], g.delta[l]]])


def compute_Z_moments(nStokes: int, mu: np.ndarray, greek: GreekCoefs, m: int):
    """Scattering/compute_Z_matrices.jl:5-84.  Returns Z++ and Z-+ as [N,N] (row i, col j)."""
    mu = np.asarray(mu, dtype=np.float64)
    assert np.all((0 < mu) & (mu <= 1)), "all mu within compute_Z_moments have to be in ]0,1]"
    n = len(mu)
    fact = 0.5 if m == 0 else 1.0
    l_max = len(greek.beta)
    P, R, T = compute_associated_legendre_PRT(mu, l_max)
    Pm, Rm, Tm = compute_associated_legendre_PRT(-mu, l_max)
    B = nStokes
    App = np.zeros((B, B, n, n))
    Amp = np.zeros((B, B, n, n))
    for l in range(m, l_max):  # reference: l = m+1 : l_max in 1-based
        Bl = _B_matrix(B, greek, l)
        Pi = _Pi_matrices(B, P, R, T, l, m)
        Pim = _Pi_matrices(B, Pm, Rm, Tm, l, m)
        left = np.einsum("iab,bc->iac", Pi, Bl)  # Pi[i]*B
        App += np.einsum("iac,jcd->adij", left, Pi)
        Amp += np.einsum("iac,jcd->adij", left, Pim)
    N = B * n
    Zpp = np.zeros((N, N))
    Zmp = np.zeros((N, N))
    for i in range(B):
        for j in range(B):
            sgn = 1.0
            if (i <= 1 and j >= 2) or (i >= 2 and j <= 1):
                sgn = -1.0
            Zpp[i::B, j::B] = 2 * fact * App[i, j]
            Zmp[i::B, j::B] = sgn * 2 * fact * Amp[i, j]
    return Zpp, Zmp


# --------------------------------------------------------------------------------------
# layer optics  (LayerOpticalProperties/compEffectiveLayerProperties.jl, types.jl:632-678)
# --------------------------------------------------------------------------------------

@dataclass
class AerosolOptics:
    greek: GreekCoefs
    omega: float  # ω̃
    ft: float = 0.0  # fᵗ


@dataclass
class Scene:
    """Inputs of the hot path = outputs of the reference's model_from_parameters()
    (model_from_parameters.jl:12-194) restricted to one concatenated band."""
    pol: PolType
    quad: QuadPoints
    max_m: int
    tau_rayl: np.ndarray  # [S, Nz]
    tau_abs: np.ndarray  # [S, Nz]
    greek_rayleigh: GreekCoefs
    tau_aer: np.ndarray = field(default_factory=lambda: np.zeros((0, 0)))  # [nAer, Nz]
    aerosols: List[AerosolOptics] = field(default_factory=list)
    varpi_cabannes: float = 1.0
    albedo: float = 0.0
    vza: np.ndarray = field(default_factory=lambda: np.zeros(0))
    vaz: np.ndarray = field(default_factory=lambda: np.zeros(0))
    strict_reference_indexing: bool = True
    brdf: Optional[tuple] = None  # None = LambertianSurfaceScalar(albedo); ("rpv", rho0, rho_c, k, Theta);
    #                               ("rossli", fvol, fgeo, fiso); ("legendre", c0, c1, ...)

    @property
    def S(self):
        return self.tau_rayl.shape[0]

    @property
    def Nz(self):
        return self.tau_rayl.shape[1]

    @property
    def N(self):
        return len(self.quad.qp_muN)


@dataclass
class LayerOptics:
    """CoreScatteringOpticalProperties (types.jl:605-614) for one layer: tau[S], varpi[S],
    Z as basis [K,N,N] + weights [K,S] (the reference stores the mixed N x N x S array;
    `Zfull()` rebuilds it with the same chained arithmetic as types.jl:632-661)."""
    tau: np.ndarray
    varpi: np.ndarray
    Zpp_basis: np.ndarray
    Zmp_basis: np.ndarray
    zweights: np.ndarray  # [K,S]

    def Zfull(self):
        Zpp = np.einsum("ks,kij->sij", self.zweights, self.Zpp_basis)
        Zmp = np.einsum("ks,kij->sij", self.zweights, self.Zmp_basis)
        return Zpp, Zmp


def construct_core_optical_properties(scene: Scene, m: int) -> List[LayerOptics]:
    """constructCoreOpticalProperties (compEffectiveLayerProperties.jl:1-78) with the
    `+` algebra of types.jl:632-678 and createAero (:80-85).

    The reference materialises Z[:,:,n] = wx[n]*Zx + wy[n]*Zy per pair, chained over
    aerosols; here the chain is carried on the K weights (w_k[n]) instead -- identical
    up to re-association of two multiplications (<= 1 ulp per term)."""
    S, Nz = scene.tau_rayl.shape
    nS = scene.pol.n
    mu = scene.quad.qp_mu
    Zr_pp, Zr_mp = compute_Z_moments(nS, mu, scene.greek_rayleigh, m)
    Zb_pp = [Zr_pp]
    Zb_mp = [Zr_mp]
    for a in scene.aerosols:
        zp, zm = compute_Z_moments(nS, mu, a.greek, m)
        Zb_pp.append(zp)
        Zb_mp.append(zm)
    Zb_pp = np.array(Zb_pp)
    Zb_mp = np.array(Zb_mp)
    K = len(Zb_pp)
    out = []
    # float64 for every real scene; complex only when oracle/dualref.py pushes a complex-step perturbation through this algebra
    dt = np.result_type(np.float64, *[np.asarray(x).dtype for x in (scene.tau_rayl, scene.tau_abs, scene.tau_aer, scene.varpi_cabannes)],
                        *[np.asarray(v).dtype for a in scene.aerosols for v in (a.omega, a.ft)])
    for iz in range(Nz):
        tau = scene.tau_rayl[:, iz].astype(dt).copy()
        varpi = np.full(S, scene.varpi_cabannes, dtype=dt)
        wts = np.zeros((K, S), dtype=dt)
        wts[0] = 1.0
        only = 0  # index of the single basis in use while Z is still unmixed
        mixed = False
        for ia, a in enumerate(scene.aerosols):
            tau_y = (1 - a.ft * a.omega) * scene.tau_aer[ia, iz]
            varpi_y = (1 - a.ft) * a.omega / (1 - a.ft * a.omega)
            tau_new = tau + tau_y
            wx = tau * varpi
            wy = np.full(S, tau_y * varpi_y)
            w = wx + wy
            varpi_new = w / tau_new
            if np.all(wx == 0.0):  # types.jl:650
                wts[:] = 0.0
                wts[ia + 1] = 1.0
            elif np.all(wy == 0.0):  # types.jl:651
                pass
            else:
                fx = wx / w
                fy = wy / w
                wts *= fx[None, :]
                wts[ia + 1] = fy
            tau, varpi = tau_new, varpi_new
        # + CoreAbsorptionOpticalProperties (types.jl:672-678)
        tau_new = tau + scene.tau_abs[:, iz]
        wx = tau * varpi
        varpi = wx / tau_new
        tau = tau_new
        out.append(LayerOptics(tau, varpi, Zb_pp, Zb_mp, wts))
    return out


def get_scattering_interface(prev: Optional[int], scatter: bool, iz: int) -> int:
    """rt_helper_functions.jl:8-27.  Codes: 0='00', 1='01', 2='10', 3='11'. iz 1-based."""
    if iz == 1:
        return 3 if scatter else 0
    if prev == 0:
        return 0 if not scatter else 1
    return 2 if not scatter else 3


def extract_effective_props(layers: List[LayerOptics]):
    """extractEffectiveProps, compEffectiveLayerProperties.jl:88-111."""
    S = len(layers[0].tau)
    Nz = len(layers)
    tau_sum = np.zeros((S, Nz + 1))
    ifaces = []
    iface = 0  # ScatteringInterface_00()
    eps = np.finfo(np.float64).eps
    for iz in range(Nz):
        scatter = bool(np.max(layers[iz].tau * layers[iz].varpi) > 2 * eps)
        iface = get_scattering_interface(iface, scatter, iz + 1)
        ifaces.append(iface)
        tau_sum[:, iz + 1] = tau_sum[:, iz] + 1.0 * layers[iz].tau
    return ifaces, tau_sum


def doubling_number(dtau_max: float, tau_end: float):
    """rt_helper_functions.jl:31-57"""
    if tau_end <= dtau_max:
        return tau_end, 0
    q1 = math.log10(2.0)
    q2 = math.log10(dtau_max)
    q3 = math.log10(tau_end)
    tlimit = (q3 - q2) / q1
    nlimit = math.floor(tlimit)
    diff = tlimit - nlimit
    if diff < np.finfo(np.float64).eps:
        return dtau_max, int(nlimit)
    nd = int(nlimit) + 1
    return 10.0 ** (q3 - q1 * nd), nd


def get_dtau_ndoubl(tau: np.ndarray, varpi: np.ndarray, qp_mu: np.ndarray):
    """rt_kernel.jl:238-246 -- maxima over the WHOLE spectral axis."""
    mx = float(np.max(tau * varpi))
    dtau_max = min(mx, 0.001 * float(np.min(qp_mu)))
    _, nd = doubling_number(dtau_max, mx)
    dtau = tau / 2 ** nd
    return dtau, nd


# --------------------------------------------------------------------------------------
# added / composite layers
# --------------------------------------------------------------------------------------

@dataclass
class AddedLayer:
    """types.jl:123-142"""
    r_pm: np.ndarray  # r⁺⁻
    r_mp: np.ndarray  # r⁻⁺
    t_mm: np.ndarray  # t⁻⁻
    t_pp: np.ndarray  # t⁺⁺
    j0p: np.ndarray
    j0m: np.ndarray


@dataclass
class CompositeLayer:
    """types.jl:105-121"""
    R_mp: np.ndarray
    R_pm: np.ndarray
    T_pp: np.ndarray
    T_mm: np.ndarray
    J0p: np.ndarray
    J0m: np.ndarray


def make_added_layer(N, S):
    z = lambda: np.zeros((S, N, N))
    return AddedLayer(z(), z(), z(), z(), np.zeros((S, N)), np.zeros((S, N)))


def make_composite_layer(N, S):
    z = lambda: np.zeros((S, N, N))
    return CompositeLayer(z(), z(), z(), z(), np.zeros((S, N)), np.zeros((S, N)))


def stokes_comp(idx0: np.ndarray, n: int, strict: bool) -> np.ndarray:
    """Stokes-component label used by the D-sign kernels.  strict (reference, Q1):
    mod(i_1based, n) -> 1,2,..,n-1,0.  non-strict: 1..n."""
    if strict:
        return np.mod(idx0 + 1, n)
    return np.mod(idx0, n) + 1


def elemental(pol: PolType, quad: QuadPoints, tau_sum: np.ndarray, dtau: np.ndarray, varpi: np.ndarray,
              Zpp: np.ndarray, Zmp: np.ndarray, m: int, nd: int, added: AddedLayer, strict: bool = True):
    """elemental! (elemental.jl:109-162) with kernels get_elem_rt! (:164-207),
    get_elem_rt_SFI! (:209-253), apply_D_elemental! (:255-274).
    Zpp/Zmp: [N,N] or [S,N,N].  Writes into `added` in place."""
    mu = quad.qp_muN
    N = len(mu)
    S = len(dtau)
    wct02 = 0.5 if m == 0 else 0.25
    wct = quad.wt_muN / 2 if m == 0 else quad.wt_muN / 4
    if Zpp.ndim == 2:
        Zpp = Zpp[None]
        Zmp = Zmp[None]
    mui = mu[:, None]
    muj = mu[None, :]
    d = dtau[:, None, None]
    w3 = varpi[:, None, None]
    wj = wct[None, None, :]
    with np.errstate(divide="ignore", invalid="ignore"):
        r = w3 * Zmp * (muj / (mui + muj))[None] * wj * (1 - np.exp(-d * ((1 / mui) + (1 / muj))[None]))
        ei = np.exp(-d / mui[None])
        ej = np.exp(-d / muj[None])
        t_off = w3 * Zpp * (muj / (mui - muj))[None] * wj * (ei - ej)
    eye = np.eye(N, dtype=bool)
    # diagonal formula uses wct[i] and Z[i,i]
    t_diag = ei * (1 + w3 * Zpp * (d / mui[None]) * wct[None, :, None])
    same_mu = (mui == muj)
    t = np.where(same_mu[None], np.where(eye[None], t_diag, 0.0), t_off)
    zero_w = ~(wct > 1e-8)  # columns with zero weight (Q4)
    r = np.where(zero_w[None, None, :], 0.0, r)
    t = np.where(zero_w[None, None, :], np.where(eye[None], ei * np.ones((1, N, N)), 0.0), t)
    added.r_mp[:] = r
    added.t_pp[:] = t

    # SFI
    n = pol.n
    i_start = n * (quad.imu0 - 1)  # 0-based
    i_end = n * quad.imu0  # exclusive
    I0 = pol.I0
    Zpp_I0 = np.einsum("sik,k->si", Zpp[:, :, i_start:i_end], I0)
    Zmp_I0 = np.einsum("sik,k->si", Zmp[:, :, i_start:i_end], I0)
    Zpp_I0 = np.broadcast_to(Zpp_I0, (S, N))
    Zmp_I0 = np.broadcast_to(Zmp_I0, (S, N))
    mu_s = mu[i_start]
    d2 = dtau[:, None]
    w2 = varpi[:, None]
    idx = np.arange(N)
    insun = (idx >= i_start) & (idx < i_end)
    with np.errstate(divide="ignore", invalid="ignore"):
        jp_in = wct02 * w2 * Zpp_I0 * (d2 / mu[None]) * np.exp(-d2 / mu[None])
        jp_out = wct02 * w2 * Zpp_I0 * (mu_s / (mu - mu_s))[None] * (np.exp(-d2 / mu[None]) - np.exp(-d2 / mu_s))
        jp = np.where(insun[None], jp_in, jp_out)
        jm = wct02 * w2 * Zmp_I0 * (mu_s / (mu + mu_s))[None] * (1 - np.exp(-d2 * ((1 / mu) + (1 / mu_s))[None]))
    att = np.exp(-tau_sum / mu_s)[:, None]
    jp = jp * att
    jm = jm * att
    if nd >= 1:
        Dfull = np.tile(pol.D, N // n)
        jm = Dfull[None] * jm
    added.j0p[:] = jp
    added.j0m[:] = jm

    # apply_D_elemental!
    comp = stokes_comp(idx, n, strict)
    if nd < 1:
        ii = comp[:, None]
        jj = comp[None, :]
        same = ((ii <= 2) & (jj <= 2)) | ((ii > 2) & (jj > 2))
        sgn = np.where(same, 1.0, -1.0)[None]
        added.r_pm[:] = sgn * added.r_mp
        added.t_mm[:] = sgn * added.t_pp
    else:
        neg = comp > 2
        added.r_mp[:, neg, :] = -added.r_mp[:, neg, :]
    # apply_D_matrix_elemental_SFI! is a no-op for every nd (Q3, elemental.jl:296-307)


def batch_inv(A: np.ndarray) -> np.ndarray:
    """batch_inv! CPU method (gpu_batched.jl:78-82): A[:,:,i]\\I via LU, partial pivoting."""
    return np.linalg.inv(A)


def elemental_inelastic_rrs(pol: PolType, quad: QuadPoints, i_l1l0, varpi_l1l0, fscatt, tau_sum, dtau, varpi, Zpp, Zmp, m: int,
                            nd: int, strict: bool = True, owned=None):
    """elemental_inelastic!(RS_type::RRS, ...) CoreKernel/elemental_inelastic.jl:23-91 with get_elem_rt_RRS! (:93-160),
    get_elem_rt_SFI_RRS! (:320-382) and apply_D_elemental_RRS! (:384-402; apply_D_elemental_SFI! :404-412 changes nothing).
    i_l1l0 [nR]: grid offsets n0 - n1; Zpp/Zmp [N,N]; dtau, varpi, fscatt, tau_sum [S].  Returns ier_mp, iet_pp, ier_pm,
    iet_mm [nR,S,N,N] and ieJ0p, ieJ0m [nR,S,N] (entries whose n0 is off the grid are zero).  PARITY UNPINNED: the reference
    holds no known-answer test for its Raman path; pinned here only through the elastic limit (tests/test_oracle_twin.py)."""
    n, N, S, nR = pol.n, len(quad.qp_muN), len(dtau), len(i_l1l0)
    mu = np.asarray(quad.qp_muN, dtype=np.float64)
    wct2 = np.asarray(quad.wt_muN) / (2.0 if m == 0 else 4.0)
    wct02 = 0.5 if m == 0 else 0.25
    i_start = n * (quad.imu0 - 1)
    I0 = np.asarray(pol.I0, dtype=np.float64)
    ier = np.zeros((nR, S, N, N)); iet = np.zeros((nR, S, N, N))
    jp = np.zeros((nR, S, N)); jm = np.zeros((nR, S, N))
    comp = stokes_comp(np.arange(N), n, strict)
    # loop invariants (hoisted; the per-(n1, dn) arithmetic below is the reference's expression for expression)
    mi, mj = mu[:, None], mu[None, :]
    eq = mi == mj
    live = (wct2 > 1.e-8)[None, :]
    dZpp = np.diag(Zpp).copy()
    zpI = Zpp[:, i_start:i_start + n] @ I0
    zmI = Zmp[:, i_start:i_start + n] @ I0
    mus = mu[i_start]
    idx = np.arange(N)
    sun = (idx >= i_start) & (idx < i_start + n)
    with np.errstate(divide="ignore", invalid="ignore"):
        for dn in range(nR):
            for n1 in range(*(owned or (0, S))):          # owned: test-side window of n1 (rrsref.RRSInputs.owned)
                n0 = n1 + int(i_l1l0[dn])
                if not 0 <= n0 < S:
                    continue
                d1, d0 = dtau[n1], dtau[n0]
                pre = varpi_l1l0[dn] * varpi[n0] * fscatt[n0]
                r = fscatt[n0] * varpi_l1l0[dn] * varpi[n0] * Zmp * (1 / ((mi / mj) + (d1 / d0))) * \
                    (1 - np.exp(-((d1 / mi) + (d0 / mj)))) * wct2[None, :]
                t_off = pre * Zpp * (1 / ((mi / mj) - (d1 / d0))) * wct2[None, :] * (np.exp(-d1 / mi) - np.exp(-d0 / mj))
                if abs(d0 - d1) > 1.e-6:
                    t_dia = pre * dZpp * wct2 * (np.exp(-d0 / mu) - np.exp(-d1 / mu)) / (1 - (d1 / d0))
                else:
                    t_dia = pre * dZpp * wct2 * (1 - np.exp(-d0 / mu))
                t = np.where(eq, 0.0, t_off)
                t[idx, idx] = t_dia
                ier[dn, n1] = np.where(live, r, 0.0)
                iet[dn, n1] = np.where(live, t, 0.0)
                if abs(d0 - d1) > 1.e-6:
                    jp_sun = (np.exp(-d0 / mu) - np.exp(-d1 / mu)) / ((d1 / d0) - 1) * pre * zpI * wct02
                else:
                    jp_sun = wct02 * pre * zpI * (1 - np.exp(-d0 / mus))
                jp_off = wct02 * pre * zpI * (1 / ((mu / mus) - (d1 / d0))) * (np.exp(-d1 / mu) - np.exp(-d0 / mus))
                att = np.exp(-tau_sum[n0] / mus)
                jp[dn, n1] = np.where(sun, jp_sun, jp_off) * att
                jm[dn, n1] = wct02 * pre * zmI * (1 / ((mu / mus) + (d1 / d0))) * (1 - np.exp(-((d1 / mu) + (d0 / mus)))) * att
    ier_pm = np.zeros_like(ier); iet_mm = np.zeros_like(iet)
    if nd < 1:
        same = ((comp[:, None] <= 2) & (comp[None, :] <= 2)) | ((comp[:, None] > 2) & (comp[None, :] > 2))
        sgn = np.where(same, 1.0, -1.0)
        ier_pm, iet_mm = sgn * ier, sgn * iet
    else:
        ier = np.where((comp > 2)[None, None, :, None], -ier, ier)
        jm = jm * np.tile(np.asarray(pol.D, dtype=np.float64), N // n)[None, None, :]
    return ier, iet, ier_pm, iet_mm, jp, jm


def batched_mul_dual(A, dA, B, dB):
    """batched_mul on ForwardDiff.Dual arrays (gpu_batched.jl:100-110): values [S,N,N], partials [P,S,N,N].
    C = A B, dC_i = A dB_i + dA_i B."""
    return A @ B, A[None] @ dB + dA @ B[None]


def batch_inv_dual(A, dA):
    """batch_inv! on ForwardDiff.Dual arrays (gpu_batched.jl:129-150): X = A^-1, dX_i = -A^-1 dA_i A^-1."""
    X = batch_inv(A)
    return X, -(X[None] @ dA) @ X[None]


def doubling(pol: PolType, expk: np.ndarray, nd: int, added: AddedLayer, strict: bool = True,
             snapshots: Optional[list] = None):
    """doubling_helper! (doubling.jl:13-79) + apply_D! (:93-110) + apply_D_SFI! (:112-118).
    expk is updated in place like the reference's `expk .= expk.^2`."""
    if nd == 0:
        return
    r = added.r_mp
    t = added.t_pp
    jp = added.j0p
    jm = added.j0m
    N = r.shape[1]
    I = np.eye(N)[None]
    for _ in range(nd):
        gp = batch_inv(I - r @ r)
        ttgp = t @ gp
        j1p = jp * expk[:, None]
        j1m = jm * expk[:, None]
        jm_new = jm + np.einsum("sij,sj->si", ttgp, j1m + np.einsum("sij,sj->si", r, jp))
        jp_new = j1p + np.einsum("sij,sj->si", ttgp, jp + np.einsum("sij,sj->si", r, j1m))
        jm[:] = jm_new
        jp[:] = jp_new
        expk[:] = expk ** 2
        r_new = r + (ttgp @ r) @ t
        t_new = ttgp @ t
        r[:] = r_new
        t[:] = t_new
        if snapshots is not None:
            snapshots.append((r.copy(), t.copy(), jp.copy(), jm.copy()))
    n = pol.n
    if n == 1:
        added.r_pm[:] = r
        added.t_mm[:] = t
        return
    comp = stokes_comp(np.arange(N), n, strict)
    neg = comp > 2
    r[:, neg, :] = -r[:, neg, :]
    ii = comp[:, None]
    jj = comp[None, :]
    same = ((ii <= 2) & (jj <= 2)) | ((ii > 2) & (jj > 2))
    sgn = np.where(same, 1.0, -1.0)[None]
    added.r_pm[:] = sgn * r
    added.t_mm[:] = sgn * t
    jm[:, neg] = -jm[:, neg]


def _mv(A, x):
    return np.einsum("sij,sj->si", A, x)


def interaction(iface: int, comp: CompositeLayer, added: AddedLayer):
    """interaction_helper! for the four interfaces (interaction.jl:8-22, 27-43, 49-64,
    69-117), dispatcher interaction_inelastic.jl:474-484.  iface: 0='00',1='01',2='10',3='11'."""
    r_pm, r_mp, t_mm, t_pp, j0p, j0m = added.r_pm, added.r_mp, added.t_mm, added.t_pp, added.j0p, added.j0m
    c = comp
    if iface == 0:
        J0p = j0p + _mv(t_pp, c.J0p)
        J0m = c.J0m + _mv(c.T_mm, j0m)
        c.J0p[:] = J0p
        c.J0m[:] = J0m
        c.T_mm[:] = t_mm @ c.T_mm
        c.T_pp[:] = t_pp @ c.T_pp
    elif iface == 1:
        J0m = c.J0m + _mv(c.T_mm, _mv(r_mp, c.J0p) + j0m)
        c.J0m[:] = J0m
        c.J0p[:] = j0p + _mv(t_pp, c.J0p)
        c.R_mp[:] = (c.T_mm @ r_mp) @ c.T_pp
        c.R_pm[:] = r_pm
        c.T_pp[:] = t_pp @ c.T_pp
        c.T_mm[:] = c.T_mm @ t_mm
    elif iface == 2:
        c.J0p[:] = j0p + _mv(t_pp, c.J0p + _mv(c.R_pm, j0m))
        c.J0m[:] = c.J0m + _mv(c.T_mm, j0m)
        c.T_pp[:] = t_pp @ c.T_pp
        c.T_mm[:] = c.T_mm @ t_mm
        c.R_pm[:] = (t_pp @ c.R_pm) @ t_mm
    elif iface == 3:
        N = r_mp.shape[1]
        I = np.eye(N)[None]
        tmp_inv = batch_inv(I - r_mp @ c.R_pm)
        T01_inv = c.T_mm @ tmp_inv
        c.J0m[:] = c.J0m + _mv(T01_inv, _mv(r_mp, c.J0p) + j0m)
        c.R_mp[:] = c.R_mp + (T01_inv @ r_mp) @ c.T_pp
        c.T_mm[:] = T01_inv @ t_mm
        tmp_inv = batch_inv(I - c.R_pm @ r_mp)
        T21_inv = t_pp @ tmp_inv
        c.J0p[:] = j0p + _mv(T21_inv, c.J0p + _mv(c.R_pm, j0m))
        c.T_pp[:] = T21_inv @ c.T_pp
        c.R_pm[:] = r_pm + (T21_inv @ c.R_pm) @ t_mm
    else:
        raise ValueError(iface)


def create_surface_layer_lambertian(albedo: float, added: AddedLayer, m: int, pol: PolType, quad: QuadPoints,
                                    tau_sum: np.ndarray):
    """create_surface_layer!(::LambertianSurfaceScalar) lambertian_surface.jl:20-75."""
    N = len(quad.qp_muN)
    n = pol.n
    Nquad = N // n
    if m == 0:
        rho = 2 * albedo
        blk = np.zeros((n, n))
        blk[0, 0] = rho
        R_surf = np.tile(blk, (Nquad, Nquad))
        I0N = np.zeros(N)
        I0N[quad.imu0Nstart - 1: n * quad.imu0] = pol.I0
        att = np.exp(-tau_sum / quad.mu0)
        added.j0p[:] = I0N[None, :] * att[:, None]
        added.j0m[:] = (quad.mu0 * (R_surf @ I0N))[None, :] * att[:, None]
        R_surf = R_surf @ np.diag(quad.qp_muN * quad.wt_muN)
        added.r_mp[:] = R_surf[None]
        added.r_pm[:] = 0
        added.t_pp[:] = np.eye(N)[None]
        added.t_mm[:] = np.eye(N)[None]
    else:
        added.r_mp[:] = 0
        added.t_pp[:] = np.eye(N)[None]
        added.t_mm[:] = np.eye(N)[None]
        added.j0p[:] = 0
        added.j0m[:] = 0


def brdf_value(brdf: tuple, mu_i: float, mu_r: float, dphi: float) -> float:
    """reflectance(brdf, n = 1, mu_i, mu_r, dphi): rpv_surface.jl:69-95 (Rahman-Pinty-Verstraete, the signs of the
    reference's vSmartMOM convention) and rossli_surface.jl:1-56 (Ross-thick + Li-sparse, h/b = 2, b/r = 1)."""
    if brdf[0] == "rpv":
        _, rho0, rho_c, k, Theta = brdf
        ti, tr = math.acos(mu_i), math.acos(mu_r)
        cosg = -mu_i * mu_r + math.sin(ti) * math.sin(tr) * math.cos(dphi)
        G = (math.tan(ti) ** 2 + math.tan(tr) ** 2 + 2 * math.tan(ti) * math.tan(tr) * math.cos(dphi)) ** 0.5
        Mf = (mu_i * mu_r) ** (k - 1) / (mu_i + mu_r) ** (1 - k)
        th = -Theta
        F = (1 - th ** 2) / (1 + th ** 2 + 2 * th * cosg) ** 1.5
        H = 1 + (1 - rho_c) / (1 + G)
        return rho0 * Mf * F * H
    if brdf[0] == "rossli":
        _, fvol, fgeo, fiso = brdf
        dphi = math.pi - dphi
        ti, tr = math.acos(mu_i), math.acos(mu_r)
        clip = lambda v: max(-1.0, min(1.0, v))
        xi = math.acos(clip(math.cos(ti) * math.cos(tr) + math.sin(ti) * math.sin(tr) * math.cos(dphi)))
        K_vol = ((math.pi / 2 - xi) * math.cos(xi) + math.sin(xi)) / (math.cos(ti) + math.cos(tr)) - math.pi / 4
        tip, trp = math.atan(math.tan(ti) * 1.0), math.atan(math.tan(tr) * 1.0)
        xip = math.acos(clip(math.cos(tip) * math.cos(trp) + math.sin(tip) * math.sin(trp) * math.cos(dphi)))
        D2 = math.tan(tip) ** 2 + math.tan(trp) ** 2 - 2 * math.tan(tip) * math.tan(trp) * math.cos(dphi)
        D = math.sqrt(max(D2, 0.0)) if D2 > -1e-300 else float("nan")
        sec_sum = 1 / math.cos(tip) + 1 / math.cos(trp)
        ct = 2.0 * math.sqrt(D ** 2 + (math.tan(tip) * math.tan(trp) * math.sin(dphi)) ** 2) / sec_sum
        t = math.acos(clip(ct))
        O = (1 / math.pi) * (t - math.sin(t) * math.cos(t)) * sec_sum
        K_geo = O - sec_sum + 0.5 * (1 + math.cos(xip)) / (math.cos(tip) * math.cos(trp))
        return fiso * 1.0 + fvol * K_vol + fgeo * K_geo
    raise ValueError(brdf[0])


def brdf_fourier_moment(brdf: tuple, n: int, mu: np.ndarray, m: int, nquad: int = 100) -> np.ndarray:
    """reflectance(brdf, pol_type, mu, m) (rpv_surface.jl:104-134): (1/pi) sum_q w_q rho(mu_i, mu_j, phi_q) cos(m phi_q)
    on a 100-point Gauss-Legendre rule over [0, pi], placed on the I component of every stream pair, times 1 (m = 0)
    or 2 (m > 0)."""
    x, w = np.polynomial.legendre.leggauss(nquad)
    phi, w = 0.5 * math.pi * (x + 1.0), 0.5 * math.pi * w
    nmu = len(mu)
    R = np.zeros((n * nmu, n * nmu))
    for i in range(nmu):
        for j in range(nmu):
            acc = 0.0
            for q in range(nquad):
                acc += w[q] * (brdf_value(brdf, float(mu[i]), float(mu[j]), float(phi[q])) * math.cos(m * phi[q]))
            R[i * n, j * n] = acc / math.pi
    return (1.0 if m == 0 else 2.0) * R


def surface_inputs(scene: "Scene"):
    """(kind, Rsurf [M,N,N] | None, albedo_spec [S] | None) of the scene's surface type."""
    b = scene.brdf
    if b is None:
        return 0, None, None
    if b[0] == "legendre":
        coef = np.asarray(b[1:], dtype=np.float64)
        x = np.linspace(-1.0, 1.0, scene.S)
        P = np.polynomial.legendre.legvander(x, len(coef) - 1)  # P_0 .. P_{n-1}
        return 2, None, P @ coef
    Rs = np.array([(2.0 if m == 0 else 1.0) * brdf_fourier_moment(b, scene.pol.n, scene.quad.qp_mu, m)
                   for m in range(scene.max_m)])
    return 1, Rs, None


def create_surface_layer(scene: "Scene", sinp, added: AddedLayer, m: int, tau_sum: np.ndarray):
    """Dispatch of create_surface_layer! on the surface type: LambertianSurfaceScalar (lambertian_surface.jl:20-75),
    LambertianSurfaceLegendre (:77-138), any BRDF type (rpv_surface.jl:20-66)."""
    pol, quad = scene.pol, scene.quad
    kind, Rs, alb = sinp
    if kind == 0:
        return create_surface_layer_lambertian(scene.albedo, added, m, pol, quad, tau_sum)
    N, n = len(quad.qp_muN), pol.n
    I0N = np.zeros(N)
    I0N[quad.imu0Nstart - 1: n * quad.imu0] = pol.I0
    att = np.exp(-tau_sum / quad.mu0)
    if kind == 1:
        R_surf = Rs[m]
        added.j0p[:] = I0N[None, :] * att[:, None]
        added.j0m[:] = (quad.mu0 * (R_surf @ I0N))[None, :] * att[:, None]
        added.r_mp[:] = (R_surf @ np.diag(quad.qp_muN * quad.wt_muN))[None]
        added.r_pm[:] = 0
        added.t_pp[:] = np.eye(N)[None]
        added.t_mm[:] = np.eye(N)[None]
        return
    # LambertianSurfaceLegendre
    if m == 0:
        rho = 2 * alb
        blk = np.zeros((n, n))
        blk[0, 0] = 1.0
        R_surf = np.tile(blk, (N // n, N // n))
        added.j0p[:] = 0
        added.j0m[:] = (quad.mu0 * (R_surf @ I0N))[None, :] * (rho * att)[:, None]
        added.r_mp[:] = rho[:, None, None] * (R_surf @ np.diag(quad.qp_muN * quad.wt_muN))[None]
        added.r_pm[:] = 0
        added.t_pp[:] = np.eye(N)[None]
        added.t_mm[:] = np.eye(N)[None]
    else:
        added.r_mp[:] = 0
        added.t_pp[:] = 0
        added.t_mm[:] = 0
        added.j0p[:] = 0
        added.j0m[:] = 0


def postprocessing_vza(pol: PolType, comp: CompositeLayer, vza, qp_mu, m: int, vaz, weight: float,
                       R_SFI: np.ndarray, T_SFI: np.ndarray):
    """postprocessing_vza!(::noRS) postprocessing_vza.jl:9-60 (SFI branch).
    R_SFI/T_SFI: [nVza, nStokes, S]."""
    n = pol.n
    for i in range(len(vza)):
        imu = nearest_point(qp_mu, float(cosd(vza[i])))
        istart = imu * n
        cs = np.array([float(cosd(m * vaz[i])), float(cosd(m * vaz[i])), float(sind(m * vaz[i])),
                       float(sind(m * vaz[i]))])[:n]
        bigCS = weight * cs
        R_SFI[i] += (bigCS[None, :] * comp.J0m[:, istart:istart + n]).T
        T_SFI[i] += (bigCS[None, :] * comp.J0p[:, istart:istart + n]).T


def rt_kernel(pol, quad, added, comp, lay: LayerOptics, iface, tau_sum, m, iz, strict=True, hook=None):
    """rt_kernel!(::noRS) rt_kernel.jl:173-235.  iz is 1-based."""
    dtau, nd = get_dtau_ndoubl(lay.tau, lay.varpi, quad.qp_mu)
    expk = np.exp(-dtau / quad.mu0)
    Zpp, Zmp = lay.Zfull()
    elemental(pol, quad, tau_sum, dtau, lay.varpi, Zpp, Zmp, m, nd, added, strict)
    if hook:
        hook("elemental", m, iz, added, comp)
    doubling(pol, expk, nd, added, strict)
    if hook:
        hook("doubling", m, iz, added, comp)
    if iz == 1:
        comp.T_pp[:] = added.t_pp
        comp.T_mm[:] = added.t_mm
        comp.R_mp[:] = added.r_mp
        comp.R_pm[:] = added.r_pm
        comp.J0p[:] = added.j0p
        comp.J0m[:] = added.j0m
    else:
        interaction(iface, comp, added)
    if hook:
        hook("interaction", m, iz, added, comp)
    return nd


def interaction_hdrf(surf: AddedLayer, comp: CompositeLayer, m: int, pol: PolType, quad: QuadPoints, bhr_uw, bhr_dw):
    """interaction_hdrf! (CoreKernel/interaction_hdrf.jl:9-45): hemispherical-directional source
    hdr_J0- = r-+_surf J0+ + j0-_surf and, for m = 0, the up/down-welling flux sums of the BHR."""
    n = pol.n
    hdr_J0m = _mv(surf.r_mp, comp.J0p) + surf.j0m
    if m == 0:
        wq = quad.wt_muN * quad.qp_muN
        i0 = quad.imu0Nstart - 1
        for i in range(n):
            bhr_uw[i, :] = (hdr_J0m[:, i::n] * wq[None, i::n]).sum(axis=1)
            bhr_dw[i, :] = (comp.J0p[:, i::n] * wq[None, i::n]).sum(axis=1) + surf.j0p[:, i0] * quad.qp_muN[i0]
    return hdr_J0m


def rt_run_full(scene: Scene, hook=None):
    """rt_run.jl:41-230 with the RAMI extras: (R_SFI, T_SFI, hdr, bhr_uw, bhr_dw); hdr is
    [nVza, nStokes, S] (postprocessing_vza_hdrf!, postprocessing_vza.jl:63-93), bhr_* [nStokes, S]."""
    pol, quad = scene.pol, scene.quad
    S, Nz, N = scene.S, scene.Nz, scene.N
    nV = len(scene.vza)
    R_SFI = np.zeros((nV, pol.n, S))
    T_SFI = np.zeros((nV, pol.n, S))
    hdr = np.zeros((nV, pol.n, S))
    bhr_uw = np.zeros((pol.n, S))
    bhr_dw = np.zeros((pol.n, S))
    added = make_added_layer(N, S)
    surf = make_added_layer(N, S)
    comp = make_composite_layer(N, S)
    strict = scene.strict_reference_indexing
    sinp = surface_inputs(scene)
    for m in range(scene.max_m):
        weight = 0.5 if m == 0 else 1.0
        layers = construct_core_optical_properties(scene, m)
        ifaces, tau_sum_all = extract_effective_props(layers)
        for iz in range(Nz):
            rt_kernel(pol, quad, added, comp, layers[iz], ifaces[iz], tau_sum_all[:, iz], m, iz + 1, strict, hook)
        create_surface_layer(scene, sinp, surf, m, tau_sum_all[:, -1])
        interaction(ifaces[-1], comp, surf)
        hdr_J0m = interaction_hdrf(surf, comp, m, pol, quad, bhr_uw, bhr_dw)
        postprocessing_vza(pol, comp, scene.vza, quad.qp_mu, m, scene.vaz, weight, R_SFI, T_SFI)
        dummy = CompositeLayer(None, None, None, None, np.zeros_like(hdr_J0m), hdr_J0m)
        postprocessing_vza(pol, dummy, scene.vza, quad.qp_mu, m, scene.vaz, weight, hdr, np.zeros_like(hdr))
    return R_SFI, T_SFI, hdr, bhr_uw, bhr_dw


def rt_run(scene: Scene, hook=None):
    """rt_run(::noRS, model, iBand) rt_run.jl:41-230, SFI=true; returns (R_SFI, T_SFI)
    each [nVza, nStokes, S]."""
    pol, quad = scene.pol, scene.quad
    S, Nz, N = scene.S, scene.Nz, scene.N
    nV = len(scene.vza)
    R_SFI = np.zeros((nV, pol.n, S))
    T_SFI = np.zeros((nV, pol.n, S))
    added = make_added_layer(N, S)
    surf = make_added_layer(N, S)
    comp = make_composite_layer(N, S)
    strict = scene.strict_reference_indexing
    sinp = surface_inputs(scene)
    for m in range(scene.max_m):
        weight = 0.5 if m == 0 else 1.0
        layers = construct_core_optical_properties(scene, m)
        ifaces, tau_sum_all = extract_effective_props(layers)
        for iz in range(Nz):
            rt_kernel(pol, quad, added, comp, layers[iz], ifaces[iz], tau_sum_all[:, iz], m, iz + 1, strict, hook)
        create_surface_layer(scene, sinp, surf, m, tau_sum_all[:, -1])
        interaction(ifaces[-1], comp, surf)  # Q6: last layer's interface code
        if hook:
            hook("surface", m, Nz + 1, surf, comp)
        postprocessing_vza(pol, comp, scene.vza, quad.qp_mu, m, scene.vaz, weight, R_SFI, T_SFI)
    return R_SFI, T_SFI


def _copy_added(comp: CompositeLayer, added: AddedLayer):
    comp.T_pp[:] = added.t_pp
    comp.T_mm[:] = added.t_mm
    comp.R_mp[:] = added.r_mp
    comp.R_pm[:] = added.r_pm
    comp.J0p[:] = added.j0p
    comp.J0m[:] = added.j0m


def interlayer_flux(top: CompositeLayer, bot: CompositeLayer):
    """interlayer_flux_helper!(::noRS) CoreKernel/interlayer_flux.jl:7-24: the downwelling and upwelling fields at the
    interface between the composite above (top) and below (bot) a sensor.  Returns (tdwJ, tuwJ) [S, N]."""
    N = top.R_pm.shape[1]
    I = np.eye(N)[None]
    tmpR = batch_inv(I - top.R_pm @ bot.R_mp)
    tdw = _mv(tmpR, top.J0p + _mv(top.R_pm, bot.J0m))
    tmpR = batch_inv(I - bot.R_mp @ top.R_pm)
    tuw = _mv(tmpR, bot.J0m + _mv(bot.R_mp, top.J0p))
    return tdw, tuw


def rt_run_multisensor(scene: Scene, sensor_levels: Sequence[int], hook=None):
    """rt_run_test_ms(::noRS, sensor_levels, model, iBand) rt_run_multisensor.jl:14-191 with
    rt_kernel_multisensor!(::noRS) rt_kernel_multisensor.jl:2-113 and postprocessing_vza_ms!(::noRS)
    tools/postprocessing_vza_ms.jl:9-77.  A sensor at level L >= 1 sits below layer L (counted from the top, L < Nz):
    its top composite holds layers 1..L, its bottom composite layers L+1..Nz and the surface; level 0 is the TOA/BOA
    pair (uwJ = J0- at the top, dwJ = J0+ at the bottom of the whole column).  Returns (uwJ, dwJ), each
    [nSensors, nVza, nStokes, S]."""
    pol, quad = scene.pol, scene.quad
    S, Nz, N = scene.S, scene.Nz, scene.N
    nV, nSens = len(scene.vza), len(sensor_levels)
    for L in sensor_levels:
        if not 0 <= L < Nz:
            raise ValueError("sensor level must be in 0..Nz-1")
    uwJ = np.zeros((nSens, nV, pol.n, S))
    dwJ = np.zeros((nSens, nV, pol.n, S))
    added = make_added_layer(N, S)
    surf = make_added_layer(N, S)
    tops = [make_composite_layer(N, S) for _ in sensor_levels]
    bots = [make_composite_layer(N, S) for _ in sensor_levels]
    strict = scene.strict_reference_indexing
    sinp = surface_inputs(scene)
    for m in range(scene.max_m):
        weight = 0.5 if m == 0 else 1.0
        layers = construct_core_optical_properties(scene, m)
        ifaces, tau_sum_all = extract_effective_props(layers)
        for iz in range(1, Nz + 1):
            lay = layers[iz - 1]
            dtau, nd = get_dtau_ndoubl(lay.tau, lay.varpi, quad.qp_mu)
            expk = np.exp(-dtau / quad.mu0)
            Zpp, Zmp = lay.Zfull()
            elemental(pol, quad, tau_sum_all[:, iz - 1], dtau, lay.varpi, Zpp, Zmp, m, nd, added, strict)
            doubling(pol, expk, nd, added, strict)
            for ims, L in enumerate(sensor_levels):           # rt_kernel_multisensor.jl:51-112
                if iz == 1:
                    _copy_added(bots[ims] if L == 0 else tops[ims], added)
                elif L == 0:
                    interaction(ifaces[iz - 1], bots[ims], added)
                elif L == iz - 1:
                    _copy_added(bots[ims], added)
                elif L < iz - 1:
                    interaction(ifaces[iz - 1], bots[ims], added)
                else:
                    interaction(ifaces[iz - 1], tops[ims], added)
        create_surface_layer(scene, sinp, surf, m, tau_sum_all[:, -1])
        for ims in range(nSens):                              # rt_run_multisensor.jl:150-159
            interaction(ifaces[-1], bots[ims], surf)
        for ims, L in enumerate(sensor_levels):               # postprocessing_vza_ms.jl:33-53
            if L == 0:
                tuw, tdw = bots[ims].J0m, bots[ims].J0p
            else:
                tdw, tuw = interlayer_flux(tops[ims], bots[ims])
            if hook:
                hook(m, ims, tops[ims], bots[ims], tdw, tuw)
            dummy = CompositeLayer(None, None, None, None, tdw, tuw)
            postprocessing_vza(pol, dummy, scene.vza, quad.qp_mu, m, scene.vaz, weight, uwJ[ims], dwJ[ims])
    return uwJ, dwJ


# --------------------------------------------------------------------------------------
# Voigt line shape (src/Absorption)
# --------------------------------------------------------------------------------------

_W32A = np.array([
    2.5722534081245696e+00, 2.2635372999002676e+00, 1.8256696296324824e+00, 1.3455441692345453e+00,
    9.0192548936480144e-01, 5.4601397206393498e-01, 2.9544451071508926e-01, 1.4060716226893769e-01,
    5.7304403529837900e-02, 1.9006155784845689e-02, 4.5195411053501429e-03, 3.9259136070122748e-04,
    -2.4532980269928922e-04, -1.3075449254548613e-04, -2.1409619200870880e-05, 6.8210319440412389e-06,
    4.4015317319048931e-06, 4.2558331390536872e-07, -4.1840763666294341e-07, -1.4813078891201116e-07,
    2.2930439569075392e-08, 2.3797557105844622e-08, 8.1248960947953431e-10, -3.2080150458594088e-09,
    -5.2310170266050247e-10, 4.1537465934749353e-10, 1.1658312885903929e-10, -5.5441820344468828e-11,
    -2.1542618451370239e-11, 8.0314997274316680e-12, 3.7424975634801558e-12, -1.3031797863050087e-12])


def w_hw32sd(z: np.ndarray) -> np.ndarray:
    """w(::HumlicekWeidemann32SDErrorFunction, z) complex_error_functions.jl:226-234:
    humlicek2 (:24-30) where |x|+y >= 8, weideman32a (:170-190) elsewhere."""
    z = np.asarray(z, dtype=np.complex128)
    far = np.abs(z.real) + z.imag >= 8
    t = z.imag - 1j * z.real
    u = t * t
    w_far = (t * (1.410474 + u * (1 / math.sqrt(math.pi)))) / (3 / 4 + (u * (3 + u)))
    L = math.sqrt(32 / math.sqrt(2))
    iz = 1j * z.real - z.imag
    rec = 1 / (L - iz)
    Z = (L + iz) * rec
    poly = np.full(z.shape, _W32A[31], dtype=np.complex128)
    for k in range(30, -1, -1):
        poly = _W32A[k] + poly * Z
    w_near = (1 / math.sqrt(math.pi) + 2 * poly * rec) * rec
    return np.where(far, w_far, w_near)


def voigt_xsec(nu, gamma_d, y, S, ind_start, ind_stop, grid) -> np.ndarray:
    """line_shape!(::Voigt) (compute_absorption_cross_section.jl:179-183) summed over lines in
    line order over each line's 1-based inclusive window (:106-122)."""
    grid = np.asarray(grid, dtype=np.float64)
    out = np.zeros(grid.size)
    cS, cL = 0.469718639319144059835, 0.8325546111577
    for j in range(len(nu)):
        a, b = int(ind_start[j]) - 1, int(ind_stop[j])
        if b <= a:
            continue
        zz = cL / gamma_d[j] * (grid[a:b] - nu[j]) + 1j * y[j]
        out[a:b] += S[j] * cS / gamma_d[j] * w_hw32sd(zz).real
    return out
