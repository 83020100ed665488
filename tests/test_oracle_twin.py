"""C oracle (oracle/momref.c) against its numpy twin (oracle/momref.py) and the golden vectors."""
from pathlib import Path

import numpy as np
import pytest

import helpers
from oracle import momref as mr

GOLD = Path(__file__).parent / "golden"


def _model(rtamd, nS, **kw):
    kw.setdefault("seed", 3)
    return rtamd.scenes.make_scene(nS, 5, 3, 6, **kw)


@pytest.mark.parametrize("nS", [1, 3, 4])
@pytest.mark.parametrize("strict", [True, False])
def test_full_run_c_vs_numpy(cref, rtamd, nS, strict):
    m = _model(rtamd, nS)
    m.params.strict_reference_indexing = strict
    sc = helpers.oracle_scene(m)
    R, T = mr.rt_run(sc)
    Rc, Tc, info = cref.rt_run(cref.pack_scene(sc))
    assert info == 0
    helpers.assert_stokes_close(Rc, R, rtol=1e-11, what="R")
    helpers.assert_stokes_close(Tc, T, rtol=1e-11, what="T")


def test_strict_indexing_matters_for_iqu(cref, rtamd):
    """for Stokes_IQU the reference's 1-based mod() never applies the D similarity."""
    m = _model(rtamd, 3, vaz=(90.0, 90.0, 90.0))
    a = cref.rt_run(cref.pack_scene(helpers.oracle_scene(m)))[0]
    m.params.strict_reference_indexing = False
    b = cref.rt_run(cref.pack_scene(helpers.oracle_scene(m)))[0]
    assert np.max(np.abs(a[:, 2] - b[:, 2])) > 1e-6  # U differs


def test_golden_small_iqu(cref, rtamd):
    g = np.load(GOLD / "small_iqu.npz")
    m = rtamd.scenes.make_scene(3, 3, 3, 4, vza=(0.0,), vaz=(35.0,), seed=7, aerosol_total=0.3, aerosol_p0=500.0,
                                aerosol_σp=300.0)
    sc = helpers.oracle_scene(m)
    p = cref.pack_scene(sc)
    assert np.array_equal(p.nd, g["nd"]) and np.array_equal(p.iface, g["iface"])
    np.testing.assert_allclose(p.Zpp, g["Zpp"], rtol=0, atol=1e-13)
    R, T, _ = cref.rt_run(p)
    helpers.assert_stokes_close(R, g["R"], rtol=1e-11, what="R vs golden")
    helpers.assert_stokes_close(T, g["T"], rtol=1e-11, what="T vs golden")
    # per-operator snapshots: elemental + doubling of every layer for m = 0
    S, N = sc.S, sc.N
    for z in range(sc.Nz):
        nd = int(p.nd[z])
        tau = p.tau.reshape(sc.Nz, S)[z]
        dtau = tau / 2 ** nd
        varpi = p.varpi.reshape(sc.Nz, S)[z]
        ts = p.tau_sum.reshape(sc.Nz + 1, S)[z]
        lay = mr.construct_core_optical_properties(sc, 0)[z]
        Zpp, Zmp = lay.Zfull()
        out = cref.elemental(p, 0, nd, ts, dtau, varpi, mr.to_abi(Zpp), mr.to_abi(Zmp), S, S)
        names = ["r_pm", "r_mp", "t_mm", "t_pp", "j0p", "j0m"]
        for nm, arr in zip(names, out):
            if nd >= 1 and nm in ("r_pm", "t_mm"):
                continue  # left untouched by the reference when nd >= 1
            ref = g[f"elemental_m0_z{z + 1}_{nm}"]
            helpers.assert_op_close(arr, mr.to_abi(ref), what=f"elemental z{z + 1} {nm}")
        expk = np.exp(-dtau / sc.quad.mu0)
        cref.doubling(p, nd, expk, out, S)
        for nm, arr in zip(names, out):
            ref = g[f"doubling_m0_z{z + 1}_{nm}"]
            helpers.assert_op_close(arr, mr.to_abi(ref), rtol=1e-11, what=f"doubling z{z + 1} {nm}")


@pytest.mark.parametrize("iface", [0, 1, 2, 3])
def test_interaction_c_vs_numpy(cref, iface):
    rng = np.random.default_rng(iface)
    N, S = 9, 5
    mk = lambda s: rng.random((S, N, N)) * s / N
    added = mr.AddedLayer(mk(0.5), mk(0.5), mk(0.9) + np.eye(N) * 0.3, mk(0.9) + np.eye(N) * 0.3, rng.random((S, N)),
                          rng.random((S, N)))
    comp = mr.CompositeLayer(mk(0.5), mk(0.5), mk(0.9), mk(0.9), rng.random((S, N)), rng.random((S, N)))
    c_comp = [mr.to_abi(x).copy() for x in (comp.R_mp, comp.R_pm, comp.T_pp, comp.T_mm, comp.J0p, comp.J0m)]
    c_add = [mr.to_abi(x).copy() for x in (added.r_pm, added.r_mp, added.t_mm, added.t_pp, added.j0p, added.j0m)]
    mr.interaction(iface, comp, added)
    assert cref.interaction(N, S, iface, c_comp, c_add) == 0
    for arr, ref in zip(c_comp, (comp.R_mp, comp.R_pm, comp.T_pp, comp.T_mm, comp.J0p, comp.J0m)):
        helpers.assert_op_close(arr, mr.to_abi(ref), what=f"iface {iface}")


def test_batch_inv_and_mul(cref):
    rng = np.random.default_rng(0)
    N, S = 17, 6
    A = rng.normal(size=(S, N, N))
    B = rng.normal(size=(S, N, N))
    X, info = cref.batch_inv(N, S, mr.to_abi(A))
    assert info == 0
    helpers.assert_op_close(X, mr.to_abi(np.linalg.inv(A)), rtol=1e-10)
    helpers.assert_op_close(cref.batched_mul(N, S, mr.to_abi(A), mr.to_abi(B)), mr.to_abi(A @ B), rtol=1e-13)
    Z = A.copy()
    Z[2, :, 3] = 0.0  # singular: zero column
    assert cref.batch_inv(N, S, mr.to_abi(Z))[1] != 0


def test_cef_against_scipy_and_golden(cref):
    from scipy.special import wofz
    g = np.load(GOLD / "cef.npz")
    X, Y = np.meshgrid(g["x"], g["y"], indexing="ij")
    w = mr.w_hw32sd(X + 1j * Y)
    np.testing.assert_allclose(w.real, g["w_re"], rtol=1e-14, atol=0)
    ex = wofz(X + 1j * Y)
    assert np.max(np.abs(w.real - ex.real) / np.abs(ex)) < 1e-4  # the approximation's own error, not parity
    wc = np.array([[cref.lib().ora_w_hw32sd_re(float(a), float(b)) for b in g["y"]] for a in g["x"]])
    np.testing.assert_allclose(wc, g["w_re"], rtol=2e-13, atol=2e-16)  # |w| <= 1; Re w cancels to 1e-5 in places


def test_voigt_c_vs_numpy_and_golden(cref):
    g = np.load(GOLD / "voigt_co2.npz")
    for tag in ("a", "b"):
        args = [g[f"{k}_{tag}"] for k in ("nu", "gamma_d", "y", "S", "ind_start", "ind_stop")]
        sig_np = mr.voigt_xsec(*args, g["grid"])
        sig_c = cref.voigt_xsec(*args, g["grid"])
        scale = g[f"sigma_{tag}"].max()
        assert np.max(np.abs(sig_np - g[f"sigma_{tag}"])) <= 1e-14 * scale
        assert np.max(np.abs(sig_c - g[f"sigma_{tag}"])) <= 1e-12 * scale


@pytest.mark.parametrize("brdf", [("rpv", 0.1, 0.8, 0.7, -0.1), ("rossli", 0.1, 0.05, 0.2), ("legendre", 0.2, 0.05, -0.02)])
def test_surface_types_twin_vs_c(rtamd, cref, brdf):
    """The two oracles (numpy twin / C) agree on the non-Lambertian surface layers (rpv_surface.jl:20-66,
    lambertian_surface.jl:77-138), and the product's host-side BRDF Fourier moments (corert.reflectance, vectorised)
    agree with the oracle's scalar-loop version."""
    import helpers
    from oracle import momref as mr
    m = rtamd.scenes.make_scene(3, 7, 3, 5, seed=3)
    so = helpers.oracle_scene(m)
    so.brdf = brdf
    p = cref.pack_scene(so)
    Rr, Tr, Hr, up, dw, info = cref.rt_run_full(p)
    Rt, Tt, Ht, upt, dwt = mr.rt_run_full(so)
    assert info == 0
    helpers.assert_stokes_close(Rt, Rr, rtol=1e-12, what="R")
    helpers.assert_stokes_close(Tt, Tr, rtol=1e-12, what="T")
    helpers.assert_stokes_close(Ht, Hr, rtol=1e-12, what="hdr")
    rt = rtamd.corert
    prod = {"rpv": lambda b: rt.rpvSurfaceScalar(*b[1:]), "rossli": lambda b: rt.RossLiSurfaceScalar(*b[1:]),
            "legendre": lambda b: rt.LambertianSurfaceLegendre(tuple(b[1:]))}[brdf[0]](brdf)
    kind, Rs, alb = rt.surface_inputs(prod, m.params.polarization_type, m.quad_points.qp_μ, m.params.max_m, 5)
    assert kind == p.surf_kind
    if Rs is not None:
        np.testing.assert_allclose(rt._abi_mats(Rs), p.Rsurf, rtol=0, atol=1e-13)
    if alb is not None:
        np.testing.assert_allclose(alb, p.albedo_spec, rtol=0, atol=1e-15)


@pytest.mark.parametrize("nS,kw", [(1, {}), (3, {}), (4, dict(brdf="rpv")), (3, dict(brdf="legendre"))])
def test_multisensor_twin_vs_c_and_adding_identities(rtamd, cref, nS, kw):
    """rt_run_test_ms (rt_run_multisensor.jl:14-191): the C oracle against the numpy twin, and the twin against the
    standard run through the adding equations themselves -- for a sensor below layer L the TOA field is the top slab's own
    source plus the interface upwelling transmitted through it, J0-(TOA) = topJ0- + topT-- tuwJ, and the BOA field
    J0+(BOA) = botJ0+ + botT++ tdwJ (interaction.jl:87-90,107-110 with the inverses of interlayer_flux.jl:14-23)."""
    m = rtamd.scenes.make_scene(nS, 5, 5, 6, seed=13, aerosol_total=0.4, vaz=(20.0, 70.0, 140.0))
    if kw.get("brdf") == "rpv":
        m.params.brdf = rtamd.corert.rpvSurfaceScalar(0.1, 0.8, 0.7, -0.1)
    elif kw.get("brdf") == "legendre":
        m.params.brdf = rtamd.corert.LambertianSurfaceLegendre((0.2, 0.05, -0.02))
    sc = helpers.oracle_scene(m)
    levels = [0, 1, 3, 4]
    total = {}
    mr.rt_run(sc, hook=lambda what, mm, iz, added, comp: total.__setitem__(mm, (comp.J0p.copy(), comp.J0m.copy()))
              if what == "surface" else None)

    def check(mm, ims, top, bot, tdw, tuw):
        if levels[ims] == 0:
            return
        J0p, J0m = total[mm]
        np.testing.assert_allclose(top.J0m + mr._mv(top.T_mm, tuw), J0m, rtol=1e-9, atol=1e-13)
        np.testing.assert_allclose(bot.J0p + mr._mv(bot.T_pp, tdw), J0p, rtol=1e-9, atol=1e-13)

    uw, dw = mr.rt_run_multisensor(sc, levels, hook=check)
    R, T = mr.rt_run(sc)
    np.testing.assert_allclose(uw[0], R, rtol=1e-12, atol=1e-15)   # level 0 = the TOA/BOA pair of rt_run
    np.testing.assert_allclose(dw[0], T, rtol=1e-12, atol=1e-15)
    uwc, dwc, info = cref.rt_run_multisensor(cref.pack_scene(sc), levels)
    assert info == 0
    for ims in range(len(levels)):
        helpers.assert_stokes_close(uwc[ims], uw[ims], rtol=1e-10, what=f"uwJ sensor {ims}")
        helpers.assert_stokes_close(dwc[ims], dw[ims], rtol=1e-10, what=f"dwJ sensor {ims}")
    assert np.abs(uw[2] - uw[0]).max() > 1e-4   # the sensors see different fields


def test_dual_operators_against_finite_differences():
    """gpu_batched.jl:100-150: the Dual rules of the two batched operators, pinned by central differences of the plain
    operators along a random direction."""
    rng = np.random.default_rng(5)
    S, N, P = 3, 6, 2
    A = rng.normal(size=(S, N, N)) + 3 * np.eye(N)
    B = rng.normal(size=(S, N, N))
    dA, dB = rng.normal(size=(P, S, N, N)), rng.normal(size=(P, S, N, N))
    C, dC = mr.batched_mul_dual(A, dA, B, dB)
    X, dX = mr.batch_inv_dual(A, dA)
    np.testing.assert_allclose(C, A @ B)
    np.testing.assert_allclose(X @ A, np.broadcast_to(np.eye(N), A.shape), atol=1e-12)
    eps = 1e-6
    for i in range(P):
        fdC = ((A + eps * dA[i]) @ (B + eps * dB[i]) - (A - eps * dA[i]) @ (B - eps * dB[i])) / (2 * eps)
        fdX = (mr.batch_inv(A + eps * dA[i]) - mr.batch_inv(A - eps * dA[i])) / (2 * eps)
        np.testing.assert_allclose(dC[i], fdC, rtol=1e-7, atol=1e-8)
        np.testing.assert_allclose(dX[i], fdX, rtol=1e-6, atol=1e-8)


def test_rrs_elemental_restatement_elastic_limit(rtamd):
    """elemental_inelastic.jl:93-160,320-382 restated in oracle/momref.py: with a zero Raman shift, unit Raman albedo and
    unit Rayleigh fraction the inelastic single-scattering layer is the elastic one (elemental.jl:164-253) wherever the
    two share an expression -- r-+ everywhere, t++ off the diagonal, j0- everywhere and j0+ off the solar rows.  (The
    reference has no known-answer test for its Raman path: this limit is the restatement's only pin.)"""
    rt = rtamd.corert
    pol = rt.Stokes_IQU()
    q = rt.rt_set_streams("GaussQuadHemisphere", 7, 50.0, [0.0, 30.0], pol)
    N, S, m, nd = len(q.qp_μN), 9, 1, 2
    rng = np.random.default_rng(3)
    Zpp, Zmp = rt.compute_Z_moments(pol, q.qp_μ, rtamd.scenes.hg_like_greek(0.5, 7), m)
    dtau, varpi, tau_sum = 10.0 ** rng.uniform(-5, -2, S), rng.uniform(0.1, 1.0, S), rng.uniform(0, 2, S)
    mq = mr.QuadPoints(q.μ0, q.iμ0, pol.n * (q.iμ0 - 1) + 1, np.asarray(q.qp_μ), np.asarray(q.wt_μ), np.asarray(q.qp_μN),
                       np.asarray(q.wt_μN), len(q.qp_μ))
    mp = mr.pol_from_n(3)
    ier, iet, _, _, jp, jm = mr.elemental_inelastic_rrs(mp, mq, [0], [1.0], np.ones(S), tau_sum, dtau, varpi, Zpp, Zmp, m, nd)
    added = mr.make_added_layer(N, S)
    mr.elemental(mp, mq, tau_sum, dtau, varpi, Zpp, Zmp, m, nd, added)
    np.testing.assert_allclose(ier[0], added.r_mp, rtol=1e-12, atol=1e-300)
    mu = np.asarray(q.qp_μN)
    off = mu[:, None] != mu[None, :]
    np.testing.assert_allclose(iet[0][:, off], added.t_pp[:, off], rtol=1e-9, atol=1e-300)
    np.testing.assert_allclose(jm[0], added.j0m, rtol=1e-12, atol=1e-300)
    i0 = pol.n * (q.iμ0 - 1)
    rows = np.r_[0:i0, i0 + pol.n:N]
    np.testing.assert_allclose(jp[0][:, rows], added.j0p[:, rows], rtol=1e-9, atol=1e-300)
