"""
oracle/dualref.py -- TEST INFRASTRUCTURE ONLY: the ForwardDiff.Dual run of the elastic hot path.

The reference differentiates rt_run by running the SAME code on ForwardDiff.Dual numbers: rt_run.jl:89-96 allocates
R, T, R_SFI, T_SFI in `FT_dual = typeof(τ_aer[1])`-style element types, the layer operators are Dual arrays, and the two
batched operators have Dual methods (gpu_batched.jl:100-150: C = A B, dC_i = A dB_i + dA_i B; X = A^-1, dX_i = -X dA_i X).
Every other statement of rt_kernel!/elemental!/doubling!/interaction! is generic arithmetic (+, *, exp, division), for
which a Dual carries the exact derivative of each operation.

This twin obtains the same tangents from the numpy restatement (oracle/momref.py, unchanged statement for statement) by
the complex-step method: the run is repeated once per partial with every input x replaced by x + i h x' (h = 1e-20);
all operations on the path are analytic (products, sums, exp, division, LU inverse with magnitude pivoting), so
Im(result) / h is the derivative of each operation propagated exactly like a Dual's partial (the O(h^2) = 1e-40 error is
far below rounding) and Re(result) is bitwise the real run's value up to the rounding of complex arithmetic.  Integer
decisions (ndoubl, interface codes, the zero-weight tests) are taken on the real parts, as ForwardDiff takes them on the values.

PARITY UNPINNED against the reference (no Julia here; the reference holds no Jacobian vector): pinned by properties --
central finite differences of the real oracle, and the explicit Dual rules of gpu_batched.jl applied to one doubling step
(tests/test_oracle_dual.py).

Inputs: a momref.Scene (values) and, per partial, the tangents AT THE BOUNDARY OF THE HOT PATH, i.e. of what
mom_scene_set uploads: d tau [S,Nz], d varpi [S,Nz], d zw [K,S,Nz], d Zpp / d Zmp [M,K,N,N] (phase-matrix bases, e.g. from
aerosol microphysics), d albedo (LambertianSurfaceScalar), d Rsurf [M,N,N] (BRDF surfaces), d albedo_spec [S] (Legendre).
`scene_partials` derives them from a perturbation of the Scene itself (the oracle-side stand-in for the Duals that
model_from_parameters would hand over).
"""
from __future__ import annotations

from dataclasses import dataclass, replace
from typing import List, Optional, Sequence

import numpy as np

from . import momref as mr

H = 1e-20


@dataclass
class Partial:
    """Tangent of the hot path's inputs with respect to ONE parameter (None = zero)."""
    dtau: Optional[np.ndarray] = None         # [S, Nz]
    dvarpi: Optional[np.ndarray] = None       # [S, Nz]
    dzw: Optional[np.ndarray] = None          # [K, S, Nz]
    dZpp: Optional[np.ndarray] = None         # [M, K, N, N]
    dZmp: Optional[np.ndarray] = None
    dalbedo: float = 0.0
    dRsurf: Optional[np.ndarray] = None       # [M, N, N]
    dalbedo_spec: Optional[np.ndarray] = None  # [S]


@dataclass
class LayerInputs:
    """What mom_scene_set uploads (values): the reference's host preparation of one band."""
    tau: np.ndarray      # [S, Nz]
    varpi: np.ndarray    # [S, Nz]
    zw: np.ndarray       # [K, S, Nz]
    Zpp: np.ndarray      # [M, K, N, N]
    Zmp: np.ndarray
    ndoubl: List[int]
    iface: List[int]
    albedo: float
    surf: tuple          # momref.surface_inputs(scene)


def layer_inputs(scene: mr.Scene) -> LayerInputs:
    """constructCoreOpticalProperties + extractEffectiveProps + get_dtau_ndoubl of the real scene (momref)."""
    per_m = [mr.construct_core_optical_properties(scene, m) for m in range(scene.max_m)]
    L0 = per_m[0]
    tau = np.stack([l.tau for l in L0], axis=1)
    varpi = np.stack([l.varpi for l in L0], axis=1)
    zw = np.stack([l.zweights for l in L0], axis=2)
    Zpp = np.stack([per_m[m][0].Zpp_basis for m in range(scene.max_m)])
    Zmp = np.stack([per_m[m][0].Zmp_basis for m in range(scene.max_m)])
    ifaces, _ = mr.extract_effective_props(L0)
    nd = [mr.get_dtau_ndoubl(l.tau, l.varpi, scene.quad.qp_mu)[1] for l in L0]
    return LayerInputs(tau, varpi, zw, Zpp, Zmp, nd, ifaces, scene.albedo, mr.surface_inputs(scene))


def _surface(scene, kind, Rs_m, alb, albedo, added, m, tau_sum):
    """momref.create_surface_layer (lambertian_surface.jl:20-138, rpv_surface.jl:20-66) for complex albedo / Rsurf."""
    pol, quad = scene.pol, scene.quad
    N, n = len(quad.qp_muN), pol.n
    I0N = np.zeros(N)
    I0N[quad.imu0Nstart - 1: n * quad.imu0] = pol.I0
    att = np.exp(-tau_sum / quad.mu0)
    eye = np.eye(N)[None]
    blk = np.zeros((n, n))
    blk[0, 0] = 1.0
    tile = np.tile(blk, (N // n, N // n))
    dmw = np.diag(quad.qp_muN * quad.wt_muN)
    if kind == 1:
        R_surf = Rs_m
        added.j0p[:] = I0N[None, :] * att[:, None]
        added.j0m[:] = (quad.mu0 * (R_surf @ I0N))[None, :] * att[:, None]
        added.r_mp[:] = (R_surf @ dmw)[None]
        added.r_pm[:] = 0
        added.t_pp[:] = eye
        added.t_mm[:] = eye
    elif kind == 0:
        if m == 0:
            R_surf = (2 * albedo) * tile
            added.j0p[:] = I0N[None, :] * att[:, None]
            added.j0m[:] = (quad.mu0 * (R_surf @ I0N))[None, :] * att[:, None]
            added.r_mp[:] = (R_surf @ dmw)[None]
            added.r_pm[:] = 0
            added.t_pp[:] = eye
            added.t_mm[:] = eye
        else:
            added.r_mp[:] = 0
            added.r_pm[:] = 0
            added.t_pp[:] = eye
            added.t_mm[:] = eye
            added.j0p[:] = 0
            added.j0m[:] = 0
    else:  # LambertianSurfaceLegendre
        if m == 0:
            rho = 2 * alb
            added.j0p[:] = 0
            added.j0m[:] = (quad.mu0 * (tile @ I0N))[None, :] * (rho * att)[:, None]
            added.r_mp[:] = rho[:, None, None] * (tile @ dmw)[None]
            added.r_pm[:] = 0
            added.t_pp[:] = eye
            added.t_mm[:] = eye
        else:
            for a in (added.r_mp, added.r_pm, added.t_pp, added.t_mm, added.j0p, added.j0m):
                a[:] = 0


def _batch_inv_gj(A: np.ndarray) -> np.ndarray:
    """Batched Gauss-Jordan with partial pivoting in the array's own precision (numpy.linalg has no extended-precision solver):
    the inverse for the x87 arbiter run below."""
    S, N, _ = A.shape
    M = np.concatenate([A.copy(), np.broadcast_to(np.eye(N, dtype=A.dtype), A.shape).copy()], axis=2)
    idx = np.arange(S)
    for k in range(N):
        p = k + np.argmax(np.abs(M[:, k:, k]), axis=1)
        rk, rp = M[idx, k].copy(), M[idx, p].copy()
        M[idx, k], M[idx, p] = rp, rk
        M[:, k] = M[:, k] / M[:, k, k][:, None]
        f = M[:, :, k].copy()
        f[:, k] = 0
        M -= f[:, :, None] * M[:, k][:, None, :]
    return M[:, :, N:]


def _run(scene: mr.Scene, L: LayerInputs, p: Optional[Partial], hook=None, extended: bool = False, full: bool = False):
    """rt_run.jl:125-215 on momref's operators with x + i H x' inputs (p = None: the plain real run, float64).
    extended: the same run in x87 extended precision (numpy complex256 / longdouble, 64-bit mantissa) -- the ARBITER of the
    thick-layer comparisons, where two Float64 runs differ by rounding amplified over the doublings."""
    cplx = p is not None
    dt = (np.clongdouble if cplx else np.longdouble) if extended else (np.complex128 if cplx else np.float64)
    if extended:
        saved = mr.batch_inv
        mr.batch_inv = _batch_inv_gj
        try:
            return _run_dt(scene, L, p, hook, cplx, dt, full)
        finally:
            mr.batch_inv = saved
    return _run_dt(scene, L, p, hook, cplx, dt, full)


def _run_dt(scene, L, p, hook, cplx, dt, full=False):

    def pert(x, dx):
        x = np.asarray(x, dtype=dt)
        return x if (not cplx or dx is None) else x + 1j * H * np.asarray(dx)

    tau = pert(L.tau, p.dtau if cplx else None)
    varpi = pert(L.varpi, p.dvarpi if cplx else None)
    zw = pert(L.zw, p.dzw if cplx else None)
    Zpp = pert(L.Zpp, p.dZpp if cplx else None)
    Zmp = pert(L.Zmp, p.dZmp if cplx else None)
    albedo = L.albedo + (1j * H * p.dalbedo if cplx else 0.0)
    kind, Rs, alb = L.surf
    if kind == 1:
        Rs = pert(Rs, p.dRsurf if cplx else None)
    if kind == 2:
        alb = pert(alb, p.dalbedo_spec if cplx else None)
    pol, quad = scene.pol, scene.quad
    S, Nz = L.tau.shape
    N = scene.N
    nV = len(scene.vza)
    R_SFI = np.zeros((nV, pol.n, S), dtype=dt)
    T_SFI = np.zeros((nV, pol.n, S), dtype=dt)
    hdr = np.zeros((nV, pol.n, S), dtype=dt)
    bhr_uw = np.zeros((pol.n, S), dtype=dt)
    bhr_dw = np.zeros((pol.n, S), dtype=dt)

    def layer(cls):
        z = lambda: np.zeros((S, N, N), dtype=dt)
        return cls(z(), z(), z(), z(), np.zeros((S, N), dtype=dt), np.zeros((S, N), dtype=dt))

    added, surf, comp = layer(mr.AddedLayer), layer(mr.AddedLayer), layer(mr.CompositeLayer)
    strict = scene.strict_reference_indexing
    tau_sum = np.zeros((S, Nz + 1), dtype=dt)
    for iz in range(Nz):
        tau_sum[:, iz + 1] = tau_sum[:, iz] + 1.0 * tau[:, iz]  # compEffectiveLayerProperties.jl:108
    for m in range(scene.max_m):
        weight = 0.5 if m == 0 else 1.0
        for iz in range(Nz):
            nd = L.ndoubl[iz]
            dtau = tau[:, iz] / 2 ** nd  # rt_kernel.jl:238-246 (the integer is the value run's)
            expk = np.exp(-dtau / quad.mu0)
            Zp = np.einsum("ks,kij->sij", zw[:, :, iz], Zpp[m])
            Zm = np.einsum("ks,kij->sij", zw[:, :, iz], Zmp[m])
            mr.elemental(pol, quad, tau_sum[:, iz], dtau, varpi[:, iz], Zp, Zm, m, nd, added, strict)
            if hook:
                hook("elemental", m, iz + 1, added, comp)
            mr.doubling(pol, expk, nd, added, strict)
            if hook:
                hook("doubling", m, iz + 1, added, comp)
            if iz == 0:
                mr._copy_added(comp, added)
            else:
                mr.interaction(L.iface[iz], comp, added)
            if hook:
                hook("interaction", m, iz + 1, added, comp)
        _surface(scene, kind, Rs[m] if kind == 1 else None, alb, albedo, surf, m, tau_sum[:, -1])
        mr.interaction(L.iface[-1], comp, surf)
        mr.postprocessing_vza(pol, comp, scene.vza, quad.qp_mu, m, scene.vaz, weight, R_SFI, T_SFI)
        # the RAMI extras (momref.rt_run_full: interaction_hdrf!, postprocessing_vza_hdrf!)
        hdr_J0m = mr.interaction_hdrf(surf, comp, m, pol, quad, bhr_uw, bhr_dw)
        dummy = mr.CompositeLayer(None, None, None, None, np.zeros_like(hdr_J0m), hdr_J0m)
        mr.postprocessing_vza(pol, dummy, scene.vza, quad.qp_mu, m, scene.vaz, weight, hdr, np.zeros_like(hdr))
    if full:
        return R_SFI, T_SFI, hdr, bhr_uw, bhr_dw
    return R_SFI, T_SFI


def rt_run_dual(scene: mr.Scene, partials: Sequence[Partial], L: Optional[LayerInputs] = None, hook=None, extended: bool = False):
    """rt_run on Dual inputs: (R_SFI, T_SFI) [nVza, nStokes, S] and their partials dR, dT [P, nVza, nStokes, S].
    extended = True: the x87 extended-precision run (results rounded to Float64 at the end)."""
    L = layer_inputs(scene) if L is None else L
    R, T = _run(scene, L, None, extended=extended)
    R, T = np.asarray(R, dtype=np.float64), np.asarray(T, dtype=np.float64)
    dR = np.zeros((len(partials),) + R.shape)
    dT = np.zeros_like(dR)
    for i, p in enumerate(partials):
        Rc, Tc = _run(scene, L, p, hook=(lambda *a, i=i: hook(i, *a)) if hook else None, extended=extended)
        dR[i] = np.asarray(Rc.imag / H, dtype=np.float64)
        dT[i] = np.asarray(Tc.imag / H, dtype=np.float64)
    return R, T, dR, dT


def rt_run_dual_full(scene: mr.Scene, partials: Sequence[Partial], L: Optional[LayerInputs] = None):
    """The reference's whole return tuple on Duals: values (R, T, hdr, bhr_uw, bhr_dw) and their partials (leading axis P)."""
    L = layer_inputs(scene) if L is None else L
    vals = [np.asarray(x, dtype=np.float64) for x in _run(scene, L, None, full=True)]
    ders = [np.zeros((len(partials),) + v.shape) for v in vals]
    for i, p in enumerate(partials):
        for d, x in zip(ders, _run(scene, L, p, full=True)):
            d[i] = x.imag / H
    return vals, ders


def rt_run_values(scene: mr.Scene, L: LayerInputs):
    """The real run from explicit layer inputs (finite-difference checks of the tangents)."""
    return _run(scene, L, None)


def scene_partial(scene: mr.Scene, perturb) -> Partial:
    """The boundary tangents produced by a perturbation of the Scene: `perturb(scene, eps)` returns the scene with its
    parameter moved by eps (eps complex); constructCoreOpticalProperties (complex-safe) carries it to tau, varpi, zw."""
    sc = perturb(scene, 1j * H)
    per_m = [mr.construct_core_optical_properties(sc, m) for m in range(scene.max_m)]
    L0 = per_m[0]
    tau = np.stack([l.tau for l in L0], axis=1)
    varpi = np.stack([l.varpi for l in L0], axis=1)
    zw = np.stack([l.zweights for l in L0], axis=2)
    Zpp = np.stack([per_m[m][0].Zpp_basis for m in range(scene.max_m)])
    Zmp = np.stack([per_m[m][0].Zmp_basis for m in range(scene.max_m)])
    im = lambda x: np.imag(x) / H
    return Partial(im(tau), im(varpi), im(zw), im(Zpp), im(Zmp), float(np.imag(sc.albedo) / H))
