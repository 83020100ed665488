"""rt_run on ForwardDiff.Dual numbers (mom_scene_set_partials / mom_rt_run_dual / mom_get_RT_partials, csrc/mom_dual.hip)
through the C ABI: values against the C oracle to the Stokes bar, partials against oracle/dualref.py (the Dual run of the numpy
twin) to the same bar relative to the largest partial of the view (tangents are propagated by the same products as the values)."""
import sys
from pathlib import Path

import numpy as np
import pytest

ROOT = Path(__file__).resolve().parents[1]
sys.path.insert(0, str(ROOT))
sys.path.insert(0, str(ROOT / "tests"))

import rtamd  # noqa: E402
import helpers  # noqa: E402
from oracle import cref, dualref as dr, momref as mr  # noqa: E402
from test_oracle_dual import random_partials  # noqa: E402

pytestmark = pytest.mark.gpu


def to_host(p: dr.Partial) -> rtamd.ScenePartial:
    return rtamd.ScenePartial(dτ=p.dtau, dϖ=p.dvarpi, dzw=p.dzw, dZpp=p.dZpp, dZmp=p.dZmp, dalbedo=p.dalbedo, dRsurf=p.dRsurf,
                              dalbedo_spec=p.dalbedo_spec)


def assert_partial_close(d, dref, rtol, what):
    """|d - dref| <= rtol * the largest |partial| of the same view (a partial of I crosses zero along the spectrum, so the
    intensity-relative bar of the values is taken against the view's largest partial instead)."""
    assert d.shape == dref.shape and np.all(np.isfinite(d)), what
    scale = np.abs(dref).max(axis=(1, 2), keepdims=True)
    err = np.abs(d - dref)
    worst = float((err / np.maximum(scale, 1e-300)).max())
    helpers._log_parity(what, rtol, worst, err.size)
    assert np.all(err <= rtol * scale + helpers.ATOL_STOKES), f"{what}: {worst:.3e} of the view's largest partial > {rtol:.1e}"


def compare(model, P=2, seed=0, with_Z=True, workspace_mb=0, rtol=None):
    sc = helpers.oracle_scene(model)
    L = dr.layer_inputs(sc)
    ps = random_partials(L, P, seed=seed, with_Z=with_Z, kind=L.surf[0])
    R, T, dR, dT = rtamd.rt_run_dual(model, [to_host(p) for p in ps], workspace_mb=workspace_mb)
    _, _, dRo, dTo = dr.rt_run_dual(sc, ps, L)
    # VALUES against the C oracle, like every other GPU test: on thick layers two Float64 runs with different product kernels
    # (numpy's dgemm / the C port's / the GPU's) are 1e-10 apart from rounding amplified by the doublings (helpers.stokes_rtol
    # was measured against the C port); the numpy run that carries the complex step is only the checker of the PARTIALS
    Ro, To, info = cref.rt_run(cref.pack_scene(sc))
    assert info == 0
    rtol = rtol or helpers.stokes_rtol(max(L.ndoubl))
    helpers.assert_stokes_close(R, Ro, what="dual R", rtol=rtol)
    helpers.assert_stokes_close(T, To, what="dual T", rtol=rtol)
    assert np.abs(dRo).max() > 0 and np.abs(dTo).max() > 0
    if max(L.ndoubl) <= 15:
        for i in range(P):
            assert_partial_close(dR[i], dRo[i], helpers.RTOL_STOKES, f"dR[{i}]")
            assert_partial_close(dT[i], dTo[i], helpers.RTOL_STOKES, f"dT[{i}]")
        return R, T, dR, dT
    # thick layers: the Float64 complex-step run is itself 1e-9 .. 1e-8 away from the exact partials of the same equations
    # (rounding amplified by the doublings, more for a partial than for its value).  Arbiter: the same run in x87 extended
    # precision; the GPU's error must stay within 4 x the Float64 oracle's own (the rule of tests/test_gpu_precision.py).
    _, _, dRx, dTx = dr.rt_run_dual(sc, ps, L, extended=True)
    for name, g, o, x in (("dR", dR, dRo, dRx), ("dT", dT, dTo, dTx)):
        for i in range(P):
            scale = np.abs(x[i]).max(axis=(1, 2), keepdims=True)
            e_gpu = float((np.abs(g[i] - x[i]) / scale).max())
            e_orc = float((np.abs(o[i] - x[i]) / scale).max())
            helpers._log_parity(f"{name}[{i}] vs x87 arbiter (oracle's own error {e_orc:.2e})", max(helpers.RTOL_STOKES, 4 * e_orc), e_gpu, g[i].size)
            assert e_gpu <= max(helpers.RTOL_STOKES, 4 * e_orc), f"{name}[{i}]: GPU {e_gpu:.2e} of the view's largest partial, oracle {e_orc:.2e}"
    return R, T, dR, dT


@pytest.mark.parametrize("nS,ltr,Nz,S", [(1, 3, 3, 9), (3, 5, 4, 7), (4, 5, 3, 5), (3, 13, 3, 6), (3, 21, 3, 4), (1, 40, 2, 3), (1, 75, 2, 3), (1, 130, 2, 2)])
def test_dual_run_matches_the_dual_oracle(nS, ltr, Nz, S):
    """Scalar / IQU / IQUV, edges 5 .. 68: every kernel form (wavefront per item up to edge 16; workgroup tiles of 32, 48, 64, 80 -- 96 in
    test_dual_six_wave_tile --, even and odd edges = 16-byte and 8-byte staging; wavefront / workgroup / two-block inverse), aerosol +
    absorption, all three moments."""
    compare(rtamd.scenes.make_scene(nS, ltr, Nz, S, seed=3 + nS, aerosol_total=0.2), P=2, seed=ltr)


def test_dual_values_equal_the_value_run():
    m = rtamd.scenes.make_scene(3, 9, 5, 12, seed=8)
    R, T, dR, dT = compare(m, P=1)
    Rv, Tv = rtamd.rt_run(m)[:2]
    helpers.assert_stokes_close(R, Rv, what="dual value vs mom_rt_run R")
    helpers.assert_stokes_close(T, Tv, what="dual value vs mom_rt_run T")


def test_dual_no_partials_and_single_partials():
    """P = 0 is the plain run; a partial that only moves the albedo / only tau uses the NULL (no dependence) inputs."""
    m = rtamd.scenes.make_scene(3, 5, 3, 6, seed=2, albedo=0.3)
    R0, T0, dR0, _ = rtamd.rt_run_dual(m, [])
    Rv, Tv = rtamd.rt_run(m)[:2]
    assert dR0.shape[0] == 0
    helpers.assert_stokes_close(R0, Rv, what="P = 0")
    sc = helpers.oracle_scene(m)
    L = dr.layer_inputs(sc)
    for p in (dr.Partial(dalbedo=1.0), dr.Partial(dtau=L.tau.copy()), dr.Partial(dvarpi=-0.1 * L.varpi)):
        _, _, dR, dT = rtamd.rt_run_dual(m, [to_host(p)])
        _, _, dRo, dTo = dr.rt_run_dual(sc, [p], L)
        assert_partial_close(dR[0], dRo[0], helpers.RTOL_STOKES, "single dR")
        assert_partial_close(dT[0], dTo[0], helpers.RTOL_STOKES, "single dT")


def test_dual_interfaces_and_zero_doublings():
    """Non-scattering layers on top (interfaces 00 / 01), in the middle (10) and layers without doublings."""
    m = rtamd.scenes.make_scene(3, 5, 6, 5, seed=11, aerosol_total=0.0, absorption=True)
    m.τ_rayl[:, 0] = 0.0      # 00 then 01
    m.τ_rayl[:, 3] = 0.0      # 10
    m.τ_rayl[:, 1] *= 1e-4    # ndoubl = 0
    sc = helpers.oracle_scene(m)
    L = dr.layer_inputs(sc)
    assert 0 in L.iface and 1 in L.iface and 2 in L.iface and 0 in L.ndoubl
    compare(m, P=2, seed=4, with_Z=False)


@pytest.mark.parametrize("surface", ["rpv", "legendre"])
def test_dual_surface_types(surface):
    m = rtamd.scenes.make_scene(3, 5, 3, 6, seed=5)
    if surface == "rpv":
        m.params.brdf = rtamd.corert.rpvSurfaceScalar(0.1, -0.1, 0.8, 0.05)
    else:
        m.params.brdf = rtamd.corert.LambertianSurfaceLegendre((0.3, 0.05, -0.02))
    compare(m, P=2, seed=6)


def test_dual_chunked_workspace_is_bitwise_the_unchunked_run():
    """MOM_OPT_DUAL_WORKSPACE_MB small enough for several chunks of spectral points: units are independent."""
    m = rtamd.scenes.make_scene(3, 9, 4, 40, seed=9)
    sc = helpers.oracle_scene(m)
    L = dr.layer_inputs(sc)
    ps = [to_host(p) for p in random_partials(L, 2, seed=1)]
    a = rtamd.rt_run_dual(m, ps)
    b = rtamd.rt_run_dual(m, ps, workspace_mb=2)   # 30 x 30 operators, P = 2: 0.3 MB per point -> chunks of 6
    for x, y in zip(a, b):
        assert np.array_equal(x, y)


def test_dual_padded_operator_edge():
    """A scene whose operators carry strip_pad's dummy entries (N = 58 -> 60): the partials of the bases are padded like the bases."""
    m = rtamd.scenes.make_scene(1, 110, 2, 2, seed=1, aerosol_total=0.1)
    assert m.quad_points.qp_μN.size not in (52, 56, 60)
    compare(m, P=1, seed=2)


def test_dual_errors():
    m = rtamd.scenes.make_scene(1, 3, 2, 4, seed=1)
    sc = rtamd.prepare_scene(m)
    with rtamd.corert.make_handle(m) as h:
        with pytest.raises(rtamd.MomError):
            h.scene_set_partials(1)              # no scene yet
        rtamd.corert.scene_set(h, sc)
        with pytest.raises(rtamd.MomError):
            h.scene_set_partials(1, dZpp=np.zeros(sc.Zpp.size))   # dZpp without dZmp
        with pytest.raises(rtamd.MomError):
            h.get_RT_partials()                  # no Dual run yet
        h.scene_set_partials(0)
        h.rt_run_dual()
        with pytest.raises(rtamd.MomError):
            h.get_hdr_partials()                 # a run without partials


def test_dual_C2_operator_shape():
    """The headline's operator shape (IQU, 20 streams, N = 60, three moments, aerosol + absorption) on a short column."""
    m = rtamd.scenes.make_scene(3, 33, 6, 2, seed=21, aerosol_total=0.3)
    assert m.quad_points.qp_μN.size == 60
    compare(m, P=1, seed=7)


@pytest.mark.parametrize("surface", [None, "rpv", "legendre"])
def test_dual_hdr_and_bhr(surface):
    """The rest of the reference's return tuple on Duals: hdr (postprocessing_vza_hdrf!) and the BHR flux sums (interaction_hdrf!)."""
    m = rtamd.scenes.make_scene(3, 5, 3, 6, seed=5, albedo=0.3)
    if surface == "rpv":
        m.params.brdf = rtamd.corert.rpvSurfaceScalar(0.1, -0.1, 0.8, 0.05)
    elif surface == "legendre":
        m.params.brdf = rtamd.corert.LambertianSurfaceLegendre((0.3, 0.05, -0.02))
    sc = helpers.oracle_scene(m)
    L = dr.layer_inputs(sc)
    ps = random_partials(L, 2, seed=3, kind=L.surf[0])
    vals, ders = rtamd.rt_run_dual(m, [to_host(p) for p in ps], full=True)
    vo, do = dr.rt_run_dual_full(sc, ps, L)
    hdr_v = rtamd.rt_run(m)[4]
    helpers.assert_stokes_close(vals[2], hdr_v, what="dual hdr vs mom_rt_run")
    for k, name in enumerate(("R", "T", "hdr")):
        helpers.assert_stokes_close(vals[k], vo[k], what="dual " + name)
        for i in range(2):
            assert_partial_close(ders[k][i], do[k][i], helpers.RTOL_STOKES, f"d{name}[{i}]")
    for k, name in ((3, "bhr_uw"), (4, "bhr_dw")):
        assert np.abs(vo[k]).max() > 0
        assert np.abs(vals[k] - vo[k]).max() <= 1e-10 * np.abs(vo[k]).max(), name
        for i in range(2):
            assert np.abs(ders[k][i] - do[k][i]).max() <= 1e-10 * np.abs(do[k][i]).max(), name


def test_dual_on_a_device_optics_scene():
    """The partials attach to a scene whose layer optics were assembled on the device (mom_scene_set_optics): its tau / varpi / zw
    are bitwise the host route's, so the Dual run is bitwise the one on the host-prepared scene."""
    m = rtamd.scenes.make_scene(3, 5, 4, 8, seed=3, aerosol_total=0.2)
    sc = rtamd.prepare_scene(m)
    L = dr.layer_inputs(helpers.oracle_scene(m))
    ps = [to_host(p) for p in random_partials(L, 2, seed=5)]
    R0, T0, dR0, dT0 = rtamd.rt_run_dual(m, ps)
    with rtamd.corert.make_handle(m) as h:
        rtamd.corert.run_scene_device_optics(h, m)
        rtamd.corert.scene_set_partials(h, sc, ps)
        h.rt_run_dual()
        R1, T1 = h.get_RT()
        dR1, dT1 = h.get_RT_partials()
    assert np.abs(dR0).max() > 0
    for x, y in ((R0, R1), (T0, T1), (dR0, dR1), (dT0, dT1)):
        assert np.array_equal(x, y)


def test_dual_six_wave_tile():
    """IQUV with 24 streams, N = 96: the one-tile 96 x 96 form of the product kernel (six wavefronts) and the two-block inverse."""
    m = rtamd.scenes.make_scene(4, 41, 2, 2, seed=2, aerosol_total=0.1, absorption=False)
    assert m.quad_points.qp_μN.size == 96
    compare(m, P=1, seed=3)
