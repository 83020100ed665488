"""CPU tests of the rotational-Raman restatement oracle/rrsref.py (BASELINE config 5).

The reference holds no known-answer test for its Raman path, so the restatement is pinned through properties:
  * ScatteringInterface_11 (interaction_inelastic.jl:230-340) is the exact first-order perturbation of the elastic adding
    equations (interaction.jl:69-117) in the zero-shift limit -- all six inelastic outputs against central differences;
  * the corrected 01 / 10 cases (D4) are the same perturbation under the assumptions their text makes (no inelastic
    operators in the non-scattering partner);
  * doubling_inelastic.jl:61-124: the iet++ and ieJ0+ updates of ONE doubling step are the perturbation of doubling.jl:43-68
    (the ier-+ / ieJ0- updates as written are not -- they read the already-updated iet++ / ieJ0+ -- and are kept as written);
  * elastic limit of the whole run, spectrally uniform scenes (index bookkeeping of get_n0_n1), strict-vs-corrected switch.
"""
import copy

import numpy as np
import pytest

from oracle import momref as mr
from oracle import rrsref as rr

import helpers


def small_scene(rtamd, nS, S, Nz, aerosol, seed, lt=None):
    """A seeded synthetic scene through the product's host code, converted to the oracle's own types."""
    lt = lt or {1: 3, 3: 5, 4: 5}[nS]
    m = rtamd.scenes.make_scene(nS, lt, Nz, S, seed=seed, aerosol_total=0.2 if aerosol else 0.0)
    return helpers.oracle_scene(m)


def _fill(o, names4, names3, elastic_m, elastic_v, S, N, rng, nR=1):
    for nm in elastic_m:
        getattr(o, nm)[:] = 0.1 * rng.random((S, N, N))
    for nm in elastic_v:
        getattr(o, nm)[:] = rng.random((S, N))
    for nm in names4:
        getattr(o, nm)[:] = rng.random((nR, S, N, N))
    for nm in names3:
        getattr(o, nm)[:] = rng.random((nR, S, N))


PAIRS_A = [("r_pm", "ier_pm"), ("r_mp", "ier_mp"), ("t_mm", "iet_mm"), ("t_pp", "iet_pp"), ("j0p", "ieJ0p"), ("j0m", "ieJ0m")]
PAIRS_C = [("R_pm", "ieR_pm"), ("R_mp", "ieR_mp"), ("T_mm", "ieT_mm"), ("T_pp", "ieT_pp"), ("J0p", "ieJ0p"), ("J0m", "ieJ0m")]


def _random_layers(S, N, rng):
    a = rr.make_added_layer_rs(N, S, 1)
    c = rr.make_composite_layer_rs(N, S, 1)
    _fill(a, ["ier_pm", "ier_mp", "iet_mm", "iet_pp"], ["ieJ0p", "ieJ0m"], ["r_pm", "r_mp", "t_mm", "t_pp"], ["j0p", "j0m"], S, N, rng)
    _fill(c, ["ieR_pm", "ieR_mp", "ieT_mm", "ieT_pp"], ["ieJ0p", "ieJ0m"], ["R_pm", "R_mp", "T_mm", "T_pp"], ["J0p", "J0m"], S, N, rng)
    for o, ns in ((a, ["t_mm", "t_pp"]), (c, ["T_mm", "T_pp"])):
        for nm in ns:
            getattr(o, nm)[:] += 0.8 * np.eye(N)
    return a, c


@pytest.mark.parametrize("iface", [3, 1, 2])
def test_interaction_is_first_order_perturbation(iface):
    rng = np.random.default_rng(2)
    S, N = 4, 5
    rrs = rr.RRSInputs(np.array([0]), np.array([1.0]), None, rrs_strict_reference=False)
    a, c = _random_layers(S, N, rng)
    if iface == 1:      # composite layer without scattering: no reflection, no inelastic operators
        for nm in ("R_pm", "R_mp", "ieR_pm", "ieR_mp", "ieT_mm", "ieT_pp", "ieJ0p", "ieJ0m"):
            getattr(c, nm)[:] = 0.0
    if iface == 2:      # added layer without scattering
        for nm in ("r_pm", "r_mp", "ier_pm", "ier_mp", "iet_mm", "iet_pp", "ieJ0p", "ieJ0m"):
            getattr(a, nm)[:] = 0.0

    def elastic(eps):
        ea, ec = mr.make_added_layer(N, S), mr.make_composite_layer(N, S)
        for e, i in PAIRS_A:
            getattr(ea, e)[:] = getattr(a, e) + eps * getattr(a, i)[0]
        for e, i in PAIRS_C:
            getattr(ec, e)[:] = getattr(c, e) + eps * getattr(c, i)[0]
        mr.interaction(iface, ec, ea)
        return ec

    eps = 1e-6
    ep, em = elastic(eps), elastic(-eps)
    cc = copy.deepcopy(c)
    rr.interaction_inelastic(rrs, iface, cc, copy.deepcopy(a))
    e0 = elastic(0.0)
    for e, i in PAIRS_C:
        fd = (getattr(ep, e) - getattr(em, e)) / (2 * eps)
        got = getattr(cc, i)[0]
        assert np.abs(fd - got).max() <= 2e-9 * max(np.abs(fd).max(), 1.0), (iface, e)
        assert np.array_equal(getattr(cc, e), getattr(e0, e)), (iface, e)   # the elastic part is interaction.jl's


def test_strict_position_raises_where_the_reference_does():
    rrs = rr.RRSInputs(np.array([0]), np.array([1.0]), None, rrs_strict_reference=True)
    a, c = _random_layers(3, 4, np.random.default_rng(0))
    for iface in (0, 1, 2):
        with pytest.raises(rr.ReferenceRaises):
            rr.interaction_inelastic(rrs, iface, c, a)
    rr.interaction_inelastic(rrs, 3, c, a)


def test_doubling_step_perturbation_identities():
    rng = np.random.default_rng(1)
    S, N = 5, 6
    pol = mr.Stokes_I()
    a = rr.make_added_layer_rs(N, S, 1)
    a.r_mp[:] = 0.05 * rng.random((S, N, N))
    a.t_pp[:] = np.eye(N) * 0.9 + 0.05 * rng.random((S, N, N))
    a.j0p[:] = rng.random((S, N))
    a.j0m[:] = rng.random((S, N))
    a.ier_mp[:] = rng.random((1, S, N, N))
    a.iet_pp[:] = rng.random((1, S, N, N))
    a.ieJ0p[:] = rng.random((1, S, N))
    a.ieJ0m[:] = rng.random((1, S, N))
    rrs = rr.RRSInputs(np.array([0]), np.array([1.0]), None, rrs_strict_reference=False)
    expk0 = rng.random(S) * 0.5 + 0.5

    def elastic(eps):
        e = mr.make_added_layer(N, S)
        e.r_mp[:] = a.r_mp + eps * a.ier_mp[0]
        e.t_pp[:] = a.t_pp + eps * a.iet_pp[0]
        e.j0p[:] = a.j0p + eps * a.ieJ0p[0]
        e.j0m[:] = a.j0m + eps * a.ieJ0m[0]
        mr.doubling(pol, expk0.copy(), 1, e, True)
        return e

    eps = 1e-6
    ep, em, e0 = elastic(eps), elastic(-eps), elastic(0.0)
    b = copy.deepcopy(a)
    ek = expk0.copy()
    rr.doubling_inelastic(pol, rrs, ek, 1, b, True)
    for nm, ie in (("t_pp", "iet_pp"), ("j0p", "ieJ0p")):
        fd = (getattr(ep, nm) - getattr(em, nm)) / (2 * eps)
        assert np.abs(fd - getattr(b, ie)[0]).max() <= 2e-9 * np.abs(fd).max(), nm
    for nm in ("r_mp", "t_pp", "j0p", "j0m", "r_pm", "t_mm"):               # elastic part = doubling.jl (corrected position)
        assert np.array_equal(getattr(b, nm), getattr(e0, nm)), nm
    assert np.array_equal(ek, expk0 ** 2)
    # D1: in the strict position the elastic sources are advanced nRaman times per step and expk squared nRaman times
    rrs3 = rr.RRSInputs(np.array([0, 1, -1]), np.ones(3), None, rrs_strict_reference=True)
    a3 = rr.make_added_layer_rs(N, S, 3)
    for nm in ("r_mp", "t_pp", "j0p", "j0m"):
        getattr(a3, nm)[:] = getattr(a, nm)
    ek3 = expk0.copy()
    rr.doubling_inelastic(pol, rrs3, ek3, 1, a3, True)
    assert np.allclose(ek3, expk0 ** 8, rtol=1e-14)
    assert not np.allclose(a3.j0p, e0.j0p)
    assert np.array_equal(a3.r_mp, e0.r_mp) and np.array_equal(a3.t_pp, e0.t_pp)


def _rrs_for(scene, offsets, strict, amp=0.02):
    nR = len(offsets)
    return rr.RRSInputs(np.asarray(offsets, dtype=np.int64), amp * (1.0 + 0.1 * np.arange(nR)), mr.get_greek_rayleigh(0.2),
                        rrs_strict_reference=strict)


@pytest.mark.parametrize("nS", [1, 3])
def test_elastic_limit_of_the_run(rtamd, nS):
    scene = small_scene(rtamd, nS=nS, S=6, Nz=3, aerosol=False, seed=5)
    scene.varpi_cabannes = 0.97
    rrs = _rrs_for(scene, [-2, 1, 3], strict=False, amp=0.0)
    R, T, ieR, ieT = rr.rt_run_rrs(scene, rrs)
    R0, T0 = mr.rt_run(scene)
    assert np.array_equal(R, R0) and np.array_equal(T, T0)
    assert not ieR.any() and not ieT.any()


def test_uniform_scene_bookkeeping(rtamd):
    """Spectrally uniform optical properties: away from the ends of the grid every Raman line sees the same operators, so
    the inelastic spectrum of nRaman lines with equal weights is nRaman times the single zero-shift line's."""
    scene = small_scene(rtamd, nS=3, S=12, Nz=2, aerosol=False, seed=3)
    scene.tau_rayl[:] = scene.tau_rayl[:1]
    scene.tau_abs[:] = scene.tau_abs[:1]
    scene.varpi_cabannes = 0.96
    one = rr.RRSInputs(np.array([0]), np.array([0.03]), mr.get_greek_rayleigh(0.2), rrs_strict_reference=False)
    offs = [-3, -1, 0, 2]
    many = rr.RRSInputs(np.array(offs), np.full(4, 0.03), mr.get_greek_rayleigh(0.2), rrs_strict_reference=False)
    _, _, ieR1, ieT1 = rr.rt_run_rrs(scene, one)
    R, T, ieR, ieT = rr.rt_run_rrs(scene, many)
    interior = slice(3, 12 - 3)
    assert np.abs(ieR1).max() > 0
    assert np.allclose(ieR[..., interior], 4 * ieR1[..., interior], rtol=1e-12, atol=0)
    assert np.allclose(ieT[..., interior], 4 * ieT1[..., interior], rtol=1e-12, atol=0)
    # at the ends lines whose source index falls off the grid are missing (get_n0_n1)
    assert np.allclose(ieR[..., 0], 2 * ieR1[..., 0], rtol=1e-12)      # offsets 0 and +2 remain at n1 = 0
    assert np.allclose(ieR[..., -1], 3 * ieR1[..., -1], rtol=1e-12)    # -3, -1, 0 at the last point


@pytest.mark.parametrize("nS", [1, 3, 4])
def test_strict_and_corrected_positions(rtamd, nS):
    scene = small_scene(rtamd, nS=nS, S=8, Nz=3, aerosol=True, seed=11)
    scene.varpi_cabannes = 0.96
    offs = [-2, 1, 3]
    Rc, Tc, ieRc, ieTc = rr.rt_run_rrs(scene, _rrs_for(scene, offs, strict=False))
    Rs, Ts, ieRs, ieTs = rr.rt_run_rrs(scene, _rrs_for(scene, offs, strict=True))
    R0, _ = mr.rt_run(scene)
    assert np.allclose(Rc, R0, rtol=1e-13, atol=1e-300)          # corrected: the elastic field is the noRS one
    assert not np.allclose(Rs, R0, rtol=1e-6)                     # strict: D1 advances the elastic sources nRaman times
    assert np.isfinite(ieRs).all() and np.isfinite(ieRc).all()
    assert np.abs(ieRc).max() > 0 and np.abs(ieRs).max() > 0


# ---- the C port of this path (oracle/momref.c ora_rt_run_rrs) against the numpy twin -------------------------------------

def _cmp4(got, ref, own=slice(None), tol=1e-11):   # the bar of the elastic C-vs-numpy twin tests (LU + FMA rounding x 2^nd)
    for k in range(4):
        g, r = got[k][..., own], ref[k][..., own]
        scale = max(float(np.abs(ref[k]).max()), 1e-300)
        assert np.abs(g - r).max() <= tol * scale, (k, np.abs(g - r).max() / scale)


@pytest.mark.parametrize("nS,S,Nz,aer,offs", [(1, 9, 3, False, [-2, 1, 3]), (3, 8, 3, True, [-2, 1, 3]), (4, 8, 2, True, [-5, 0, 2, 7]),
                                               (3, 14, 4, True, [-13, -1, 4, 13])])
@pytest.mark.parametrize("strict", [False, True])
def test_c_port_equals_numpy_twin(rtamd, cref, nS, S, Nz, aer, offs, strict):
    """Whole runs, both switch positions, IQ(U)(V) and scalar, offsets up to S - 1 (a single on-grid pair), zero offset, several
    layers (persistent added layer: D5 reads the previous layer's iet--; stale off-grid ieJ0 entries; D2 / D3 cross-indexing)."""
    scene = small_scene(rtamd, nS=nS, S=S, Nz=Nz, aerosol=aer, seed=7 + nS)
    scene.varpi_cabannes = 0.96
    rrs = _rrs_for(scene, offs, strict=strict)
    ref = rr.rt_run_rrs(scene, rrs)
    got = cref.rt_run_rrs(scene, rrs)
    assert got[4] == 0
    assert np.abs(ref[2]).max() > 0
    _cmp4(got, ref)
    for lo, hi in ((0, 3), (2, S - 1), (S - 2, S)):                        # owned windows: the owned entries of the full run
        rrs_w = _rrs_for(scene, offs, strict=strict)
        rrs_w.owned = (lo, hi)
        w = cref.rt_run_rrs(scene, rrs_w)
        _cmp4(w, ref, own=slice(lo, hi))
        assert not np.any(w[2][..., :lo]) and not np.any(w[2][..., hi:])


def test_c_port_other_interfaces_and_zero_doublings(rtamd, cref):
    """Layers without scattering at the top (interfaces 00 / 01 / 10 of the corrected position, D4) and nd = 0 layers
    (apply_D_elemental_RRS!'s ndoubl < 1 branch); the strict position raises like the twin."""
    m = rtamd.scenes.make_scene(3, 5, 5, 10, seed=3, aerosol_total=0.0)
    for z in (0, 1, 3):
        m.τ_rayl[:, z] = 0.0
    m.τ_rayl[:, 2] *= 1e-4                                                  # below the doubling threshold: nd = 0
    scene = helpers.oracle_scene(m)
    scene.varpi_cabannes = 0.95
    p = cref.pack_scene(scene)
    assert set(p.iface.tolist()) >= {0, 1, 2, 3} or len(set(p.iface.tolist())) >= 3, p.iface
    assert 0 in p.nd.tolist()
    rrs = _rrs_for(scene, [-3, 0, 2], strict=False)
    ref = rr.rt_run_rrs(scene, rrs)
    got = cref.rt_run_rrs(scene, rrs, p=p)
    _cmp4(got, ref)
    with pytest.raises(rr.ReferenceRaises):
        rr.rt_run_rrs(scene, _rrs_for(scene, [-3, 0, 2], strict=True))
    with pytest.raises(cref.ReferenceRaises):
        cref.rt_run_rrs(scene, _rrs_for(scene, [-3, 0, 2], strict=True), p=p)
    with pytest.raises(cref.ReferenceRaises):                              # |offset| >= S: get_n0_n1's BoundsError
        cref.rt_run_rrs(scene, _rrs_for(scene, [10], strict=False), p=p)


def test_c_port_surfaces(rtamd, cref):
    rt = rtamd.corert
    for brdf in (rt.rpvSurfaceScalar(0.1, 0.8, 0.7, -0.1), rt.LambertianSurfaceLegendre((0.2, 0.05, -0.02))):
        m = rtamd.scenes.make_scene(3, 3, 3, 7, seed=2)
        m.params.brdf = brdf
        scene = helpers.oracle_scene(m)
        scene.varpi_cabannes = 0.96
        rrs = _rrs_for(scene, [-1, 2], strict=False)
        _cmp4(cref.rt_run_rrs(scene, rrs), rr.rt_run_rrs(scene, rrs))
