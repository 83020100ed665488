"""oracle/dualref.py (the ForwardDiff.Dual run of the elastic path as a complex-step run of the numpy twin) pinned by
properties: its values are the real run's, its tangents are the central finite differences of the real run, the
Dual rules of gpu_batched.jl:100-150 reproduce one doubling step's tangents, and tangents are linear in the direction."""
import sys
from dataclasses import replace
from pathlib import Path

import numpy as np
import pytest

ROOT = Path(__file__).resolve().parents[1]
sys.path.insert(0, str(ROOT))
sys.path.insert(0, str(ROOT / "tests"))

import rtamd  # noqa: E402
import helpers  # noqa: E402
from oracle import dualref as dr, momref as mr  # noqa: E402


def _scene(nS=3, nq=3, Nz=4, S=6, seed=5, **kw):
    m = rtamd.scenes.make_scene(nS, nq, Nz, S, seed=seed, **kw)
    return helpers.oracle_scene(m)


def random_partials(L, P, seed=0, with_Z=True, kind=0, M=None):
    rng = np.random.default_rng(seed)
    out = []
    for _ in range(P):
        p = dr.Partial(dtau=L.tau * rng.uniform(-1, 1, L.tau.shape), dvarpi=0.3 * L.varpi * rng.uniform(-1, 1, L.varpi.shape),
                       dzw=L.zw * rng.uniform(-1, 1, L.zw.shape), dalbedo=float(rng.uniform(0.2, 1.0)))
        if with_Z:
            p.dZpp = L.Zpp * rng.uniform(-0.5, 0.5, L.Zpp.shape)
            p.dZmp = L.Zmp * rng.uniform(-0.5, 0.5, L.Zmp.shape)
        if kind == 1:
            p.dRsurf = L.surf[1] * rng.uniform(-0.5, 0.5, L.surf[1].shape)
        if kind == 2:
            p.dalbedo_spec = rng.uniform(-0.2, 0.2, L.surf[2].shape)
        out.append(p)
    return out


def shifted(L, p, eps):
    a = lambda x, dx: x if dx is None else x + eps * dx
    kind, Rs, alb = L.surf
    surf = (kind, a(Rs, p.dRsurf) if kind == 1 else Rs, a(alb, p.dalbedo_spec) if kind == 2 else alb)
    return replace(L, tau=a(L.tau, p.dtau), varpi=a(L.varpi, p.dvarpi), zw=a(L.zw, p.dzw), Zpp=a(L.Zpp, p.dZpp),
                   Zmp=a(L.Zmp, p.dZmp), albedo=L.albedo + eps * p.dalbedo, surf=surf)


@pytest.mark.parametrize("nS,brdf", [(1, None), (3, None), (4, None), (3, "rpv"), (1, "legendre")])
def test_values_and_finite_differences(nS, brdf):
    sc = _scene(nS=nS, aerosol_total=0.2)
    if brdf == "rpv":
        sc.brdf = ("rpv", 0.1, -0.1, 0.8, 0.05)
    if brdf == "legendre":
        sc.brdf = ("legendre", 0.3, 0.05, -0.02)
    sc.albedo = 0.25
    L = dr.layer_inputs(sc)
    ps = random_partials(L, 2, seed=nS, kind=L.surf[0])
    R, T, dR, dT = dr.rt_run_dual(sc, ps, L)
    R0, T0 = mr.rt_run(sc)
    assert np.allclose(R, R0, rtol=1e-12, atol=1e-15) and np.allclose(T, T0, rtol=1e-12, atol=1e-15)
    assert np.abs(dR).max() > 0 and np.abs(dT).max() > 0
    eps = 1e-4   # the difference quotient is rounding-bound (the run's own 1e-13 over eps): measured 1.5e-9 here, 2e-7 at 1e-6
    for i, p in enumerate(ps):
        Rp, Tp = dr.rt_run_values(sc, shifted(L, p, eps))
        Rm, Tm = dr.rt_run_values(sc, shifted(L, p, -eps))
        for d, fp, fm in ((dR[i], Rp, Rm), (dT[i], Tp, Tm)):
            fd = (fp - fm) / (2 * eps)
            assert np.abs(d - fd).max() <= 5e-8 * max(np.abs(d).max(), 1e-12), (i, np.abs(d - fd).max(), np.abs(d).max())


def test_tangents_are_linear_in_the_direction():
    sc = _scene(aerosol_total=0.1)
    L = dr.layer_inputs(sc)
    p1, p2 = random_partials(L, 2, seed=9)
    comb = dr.Partial(**{k: (None if getattr(p1, k) is None else 2.0 * getattr(p1, k) - 0.5 * getattr(p2, k))
                         for k in ("dtau", "dvarpi", "dzw", "dZpp", "dZmp", "dalbedo")})
    _, _, dR, dT = dr.rt_run_dual(sc, [p1, p2, comb], L)
    assert np.allclose(dR[2], 2 * dR[0] - 0.5 * dR[1], rtol=1e-11, atol=1e-14 * np.abs(dR).max())
    assert np.allclose(dT[2], 2 * dT[0] - 0.5 * dT[1], rtol=1e-11, atol=1e-14 * np.abs(dT).max())


def test_doubling_step_equals_the_dual_rules_of_gpu_batched():
    """One doubling step written out with batched_mul / batch_inv on Duals (gpu_batched.jl:100-150: dC = A dB + dA B,
    dX = -X dA X) gives the complex-step tangents of momref.doubling."""
    rng = np.random.default_rng(3)
    S, N = 4, 6
    pol = mr.Stokes_I()
    r = 0.1 * rng.uniform(size=(S, N, N))
    t = np.eye(N)[None] * 0.8 + 0.05 * rng.uniform(size=(S, N, N))
    dr_ = rng.uniform(-1, 1, size=(1, S, N, N)) * 0.1
    dt_ = rng.uniform(-1, 1, size=(1, S, N, N)) * 0.1
    z = lambda: np.zeros((S, N, N), dtype=complex)
    added = mr.AddedLayer(z(), z(), z(), z(), np.zeros((S, N), dtype=complex), np.zeros((S, N), dtype=complex))
    added.r_mp[:] = r + 1j * dr.H * dr_[0]
    added.t_pp[:] = t + 1j * dr.H * dt_[0]
    mr.doubling(pol, np.full(S, 0.9, dtype=complex), 1, added)
    I = np.eye(N)[None]
    rr, drr = mr.batched_mul_dual(r, dr_, r, dr_)
    gp, dgp = mr.batch_inv_dual(I - rr, -drr)
    tg, dtg = mr.batched_mul_dual(t, dt_, gp, dgp)
    x, dx = mr.batched_mul_dual(tg, dtg, r, dr_)
    y, dy = mr.batched_mul_dual(x, dx, t, dt_)
    tn, dtn = mr.batched_mul_dual(tg, dtg, t, dt_)
    assert np.allclose(added.r_mp.real, r + y, rtol=1e-13) and np.allclose(added.t_pp.real, tn, rtol=1e-13)
    assert np.allclose(added.r_mp.imag / dr.H, (dr_ + dy)[0], rtol=1e-12, atol=1e-15)
    assert np.allclose(added.t_pp.imag / dr.H, dtn[0], rtol=1e-12, atol=1e-15)


def test_scene_partial_through_the_layer_optics_algebra():
    """A physical parameter (here the aerosol column and the albedo) moved through constructCoreOpticalProperties: the
    boundary tangents of dualref.scene_partial give the finite difference of the whole run with the parameter moved."""
    sc = _scene(nS=3, aerosol_total=0.15)
    sc.albedo = 0.1

    def move(s, eps):
        return replace(s, tau_aer=s.tau_aer * (1 + eps), albedo=s.albedo + 0.5 * eps)

    p = dr.scene_partial(sc, move)
    assert np.abs(p.dtau).max() > 0 and np.abs(p.dzw).max() > 0 and p.dalbedo == 0.5
    _, _, dR, dT = dr.rt_run_dual(sc, [p])
    eps = 1e-4
    Rp, Tp = mr.rt_run(move(sc, eps))
    Rm, Tm = mr.rt_run(move(sc, -eps))
    assert np.abs(dR[0] - (Rp - Rm) / (2 * eps)).max() <= 5e-8 * np.abs(dR).max()
    assert np.abs(dT[0] - (Tp - Tm) / (2 * eps)).max() <= 5e-8 * np.abs(dT).max()


def test_extended_precision_arbiter_run():
    """The x87 extended-precision run of the same statements (numpy complex256, own Gauss-Jordan inverse) -- the arbiter of the
    thick-layer partials in tests/test_gpu_dual.py: on a thin scene it is the Float64 run to rounding, on a thick one the Float64
    run is measurably away from it (the reason the arbiter exists)."""
    sc = _scene(nS=3, aerosol_total=0.2)
    L = dr.layer_inputs(sc)
    ps = random_partials(L, 1, seed=1)
    a = dr.rt_run_dual(sc, ps, L)
    b = dr.rt_run_dual(sc, ps, L, extended=True)
    for x, y in zip(a, b):
        assert np.abs(x - y).max() <= 2e-12 * np.abs(y).max()
    A = np.random.default_rng(0).uniform(-1, 1, (3, 7, 7)) + 3 * np.eye(7)
    assert np.allclose(dr._batch_inv_gj(A.astype(np.longdouble)).astype(float), np.linalg.inv(A), rtol=1e-13, atol=1e-15)
