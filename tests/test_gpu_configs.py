"""The named BASELINE.json configurations on the GPU through the C ABI: C1 (scalar plumbing
scene), C2 at full size with ALL 10 000 points against the oracle, C3 (three bands, 29 944 points; 3 072 of them) and C4
(IQUV, 64 streams, N = 256), each against the C oracle on the same seeded inputs."""
import numpy as np
import pytest

import helpers

pytestmark = pytest.mark.gpu


def _oracle(cref, model, pts=None):
    p = cref.pack_scene(helpers.oracle_scene(model))
    R, T, info = cref.rt_run(p, pts=pts)
    assert info == 0
    return R, T


def _stratified(model, n, seed=0):
    """n spectral points spread evenly over the ORDER of column absorption optical depth (thin ... opaque)."""
    col = model.τ_abs.sum(axis=1)
    order = np.argsort(col, kind="stable")
    pick = order[np.linspace(0, len(order) - 1, n).round().astype(int)]
    return np.unique(pick).astype(np.int32)


def test_config_C1_default_scalar(rtamd, cref):
    """configs[0]: scalar I, 4 streams, 10 layers, 100 spectral points -- every point against the oracle."""
    m = rtamd.scenes.scene_C1()
    sc = rtamd.prepare_scene(m)
    assert (sc.N, sc.nStokes, sc.Nz, sc.S) == (4, 1, 10, 100)
    R, T = rtamd.rt_run(m)[:2]
    Rr, Tr = _oracle(cref, m)
    helpers.assert_stokes_close(R, Rr, what="C1 R")
    helpers.assert_stokes_close(T, Tr, what="C1 T")


@pytest.mark.parametrize("lt,vza,kw", [(3, (0.0,), {}), (3, (0.0,), dict(aerosol_total=0.0, zero_layers=(0, 1, 3))),
                                       (1, (0.0,), dict(albedo=0.35)), (1, (50.0,), {}), (3, (60.0,), dict(aerosol_total=1.5)),
                                       (1, (0.0, 30.0), {}), (1, (0.0, 50.0, 50.0, 0.0), dict(albedo=0.1)),
                                       (1, (0.0, 50.0, 50.0, 0.0), dict(max_m=1)), (1, (0.0, 30.0, 30.0, 0.0), dict(max_m=1))])
def test_small_operator_sweep_kernel(rtamd, cref, lt, vza, kw):
    """N <= 4 (mom_small.hip: one spectral point per lane, the whole sweep in one launch) against the oracle AND against
    the general workgroup-per-point kernels (MOM_OPT_SMALL_N = 0): spectra, hdr and the BHR fluxes; includes all four
    interface cases with zero doublings, N = 2, 3, 4, a thick aerosol layer and max_m = 1 (the UNSPLIT kernel with its view
    accumulators in dynamic LDS, mom_small.hip)."""
    kw = dict(kw)
    zero = kw.pop("zero_layers", ())
    m = rtamd.scenes.make_scene(1, lt, 6, 200, seed=5 + lt, vza=vza, vaz=tuple(30.0 + 40.0 * i for i in range(len(vza))), **kw)
    for z in zero:
        m.τ_rayl[:, z] = 0.0
    sc = rtamd.prepare_scene(m)
    assert sc.N <= 4
    p = cref.pack_scene(helpers.oracle_scene(m))
    Rr, Tr, Hr, upr, dwr, info = cref.rt_run_full(p)
    assert info == 0
    out = {}
    for small in (1, 0):
        with rtamd.corert.make_handle(m) as h:
            h.set_option(rtamd._lib.MOM_OPT_SMALL_N, small)
            R, T = rtamd.corert.run_scene(h, sc)
            out[small] = (R, T) + h.get_hdr() + (h.timers()["layer_launches"],)
    assert out[1][5] == 1                            # the whole sweep in ONE launch
    tol = helpers.stokes_rtol(sc.ndoubl)
    for small in (1, 0):
        R, T, H, up, dw, _ = out[small]
        helpers.assert_stokes_close(R, Rr, rtol=tol, what=f"R small={small}")
        helpers.assert_stokes_close(T, Tr, rtol=tol, what=f"T small={small}")
        helpers.assert_stokes_close(H, Hr, rtol=tol, what=f"hdr small={small}")
        np.testing.assert_allclose(up, upr, rtol=max(tol, 1e-10), atol=helpers.ATOL_STOKES)
        np.testing.assert_allclose(dw, dwr, rtol=max(tol, 1e-10), atol=helpers.ATOL_STOKES)


@pytest.mark.parametrize("nS,lt,vza,kw", [
    (1, 9, (0.0, 30.0), {}),                                    # N = 8
    (1, 21, (0.0, 30.0, 45.0), dict(aerosol_total=0.6)),        # N = 15
    (3, 3, (0.0,), {}),                                         # N = 12, polarized
    (3, 5, (20.0,), dict(albedo=0.6)),                          # N = 15
    (4, 3, (0.0,), {}),                                         # N = 16, IQUV
    (4, 3, (0.0,), dict(aerosol_total=3.0, albedo=0.05)),       # N = 16, thick aerosol: long series / pivoted inverse
    (3, 1, (0.0,), {}),                                         # N = 9
    (1, 27, (0.0, 30.0, 60.0), {}),                             # N = 17: two tiles per edge, one k-step in the second
    (3, 7, (0.0, 40.0), {}),                                    # N = 21
    (4, 7, (0.0,), dict(aerosol_total=0.8)),                    # N = 24
    (1, 51, (0.0, 30.0), {}),                                   # N = 29
    (3, 13, (0.0, 30.0), dict(albedo=0.5)),                     # N = 30
    (4, 9, (0.0, 30.0), dict(aerosol_total=2.5)),               # N = 32, thick aerosol
])
@pytest.mark.parametrize("inverse", [0, 1])
def test_wave_per_point_sweep_kernel(rtamd, cref, nS, lt, vza, kw, inverse):
    """4 < N <= 32 (mom_wave.hip: one spectral point per wavefront, operators as 1 x 1 or 2 x 2 MFMA-layout register tiles,
    the whole sweep in one launch) against the oracle and against the general kernels; inverse = 1 forces the pivoted
    Gauss-Jordan inverse in place of the series."""
    m = rtamd.scenes.make_scene(nS, lt, 8, 301, seed=11 + lt + nS, vza=vza,   # 301: the last workgroup is partly idle
                                vaz=tuple(15.0 + 50.0 * i for i in range(len(vza))), **kw)
    sc = rtamd.prepare_scene(m)
    assert 4 < sc.N <= 32, sc.N
    p = cref.pack_scene(helpers.oracle_scene(m))
    Rr, Tr, Hr, upr, dwr, info = cref.rt_run_full(p)
    assert info == 0
    out = {}
    for small in (1, 0):
        with rtamd.corert.make_handle(m) as h:
            h.set_option(rtamd._lib.MOM_OPT_SMALL_N, small)
            h.set_option(rtamd._lib.MOM_OPT_INVERSE, inverse)
            R, T = rtamd.corert.run_scene(h, sc)
            out[small] = (R, T) + h.get_hdr() + (h.timers()["layer_launches"],)
    assert out[1][5] == 1                            # the whole sweep in ONE launch
    tol = helpers.stokes_rtol(sc.ndoubl)
    for small in (1, 0):
        R, T, H, up, dw, _ = out[small]
        helpers.assert_stokes_close(R, Rr, rtol=tol, what=f"R small={small}")
        helpers.assert_stokes_close(T, Tr, rtol=tol, what=f"T small={small}")
        helpers.assert_stokes_close(H, Hr, rtol=tol, what=f"hdr small={small}")
        np.testing.assert_allclose(up, upr, rtol=max(tol, 1e-10), atol=helpers.ATOL_STOKES)
        np.testing.assert_allclose(dw, dwr, rtol=max(tol, 1e-10), atol=helpers.ATOL_STOKES)


@pytest.mark.parametrize("lt,vza,N,kw", [
    (5, (0.0,), 5, {}),                                          # three points per wavefront (packed edge 15)
    (5, (0.0,), 5, dict(aerosol_total=3.0, albedo=0.05)),        # thick aerosol: long series / pivoted inverse of the packed tile
    (5, (0.0,), 5, dict(brdf="rpv")),
    (5, (0.0, 30.0), 6, {}),                                     # two points per wavefront (packed edge 12)
    (7, (0.0, 30.0), 7, dict(albedo=0.5)),                       # packed edge 14
    (9, (0.0, 30.0), 8, {}),                                     # packed edge 16: a full tile
    (9, (0.0, 30.0), 8, dict(brdf="legendre")),
    (7, (0.0,) * 40, 6, {}),                                     # 40 views x 2 points = 80 outputs per wave: two output passes
])
def test_wave_kernel_packed_points(rtamd, cref, lt, vza, N, kw):
    """N = 5 ... 8: several spectral points per wavefront as diagonal blocks of one MFMA tile (mom_wave.hip, PK = 3 / 2).
    Against the oracle and against the one-point-per-wave form of the same kernel (MOM_OPT_SMALL_N = 2); S = 301 leaves
    the last wave with one point (the tail repeats a point and does not store it)."""
    kw = dict(kw)
    brdf = kw.pop("brdf", None)
    rt = rtamd.corert
    m = rtamd.scenes.make_scene(1, lt, 7, 301, seed=3 + lt + N, vza=vza, vaz=tuple(9.0 * i for i in range(len(vza))), **kw)
    if brdf:
        m.params.brdf = {"rpv": rt.rpvSurfaceScalar(0.1, 0.8, 0.7, -0.1), "legendre": rt.LambertianSurfaceLegendre((0.2, 0.05, -0.02))}[brdf]
    sc = rtamd.prepare_scene(m)
    assert sc.N == N
    Rr, Tr, Hr, upr, dwr, info = cref.rt_run_full(cref.pack_scene(helpers.oracle_scene(m)))
    assert info == 0
    out = {}
    for small in (1, 2):
        with rt.make_handle(m) as h:
            h.set_option(rtamd._lib.MOM_OPT_SMALL_N, small)
            R, T = rt.run_scene(h, sc)
            out[small] = (R, T) + h.get_hdr() + (h.timers()["layer_launches"],)
        assert out[small][5] == 1
    tol = helpers.stokes_rtol(sc.ndoubl)
    for small in (1, 2):
        R, T, H, up, dw, _ = out[small]
        helpers.assert_stokes_close(R, Rr, rtol=tol, what=f"R small={small}")
        helpers.assert_stokes_close(T, Tr, rtol=tol, what=f"T small={small}")
        helpers.assert_stokes_close(H, Hr, rtol=tol, what=f"hdr small={small}")
        np.testing.assert_allclose(up, upr, rtol=max(tol, 1e-10), atol=helpers.ATOL_STOKES)
        np.testing.assert_allclose(dw, dwr, rtol=max(tol, 1e-10), atol=helpers.ATOL_STOKES)
    helpers.assert_stokes_close(out[1][0], out[2][0], rtol=tol, what="packed vs one point per wave")


@pytest.mark.parametrize("surf", ["rpv", "rossli", "legendre"])
@pytest.mark.parametrize("nS,lt", [(3, 3), (4, 7)])
def test_wave_kernel_surfaces_and_many_views(rtamd, cref, surf, nS, lt):
    """RAMI-style scenes on the wave-per-point kernel: BRDF surfaces (a surface interaction and hdr for every moment),
    LambertianSurfaceLegendre (spectral albedo, its j0+ = 0 and T_SFI from m = 0 only) and 30 view directions (90 / 120
    outputs per point: more than one pass of the wave over the output list), N = 15 (1 x 1 tiles) and N = 28 (2 x 2)."""
    rt = rtamd.corert
    vza = (0.0, 40.0) * 15      # two view zenith angles (two extra streams), 30 azimuths
    m = rtamd.scenes.make_scene(nS, lt, 5, 40, seed=29, vza=vza, vaz=tuple(np.linspace(0.0, 350.0, 30)))
    m.params.brdf = {"rpv": rt.rpvSurfaceScalar(0.1, 0.8, 0.7, -0.1), "rossli": rt.RossLiSurfaceScalar(0.1, 0.05, 0.2),
                     "legendre": rt.LambertianSurfaceLegendre((0.2, 0.05, -0.02))}[surf]
    sc = rtamd.prepare_scene(m)
    assert 4 < sc.N <= 32 and len(vza) * nS > 64
    Rr, Tr, Hr, upr, dwr, info = cref.rt_run_full(cref.pack_scene(helpers.oracle_scene(m)))
    assert info == 0
    with rt.make_handle(m) as h:
        R, T = rt.run_scene(h, sc)
        H, up, dw = h.get_hdr()
        assert h.timers()["layer_launches"] == 1
    tol = helpers.stokes_rtol(sc.ndoubl)
    helpers.assert_stokes_close(R, Rr, rtol=tol, what="R")
    helpers.assert_stokes_close(T, Tr, rtol=tol, what="T")
    helpers.assert_stokes_close(H, Hr, rtol=tol, what="hdr")
    np.testing.assert_allclose(up, upr, rtol=max(tol, 1e-10), atol=helpers.ATOL_STOKES)
    np.testing.assert_allclose(dw, dwr, rtol=max(tol, 1e-10), atol=helpers.ATOL_STOKES)


def test_wave_kernel_falls_back_on_other_interfaces(rtamd, cref):
    """Layers without scattering above the first scattering layer give interface codes other than 11: the wave-per-point
    kernel does not apply and the general kernels run."""
    m = rtamd.scenes.make_scene(3, 3, 8, 64, seed=3, vza=(0.0,), vaz=(0.0,), aerosol_total=0.0)
    for z in (0, 1, 3):
        m.τ_rayl[:, z] = 0.0
    sc = rtamd.prepare_scene(m)
    assert 4 < sc.N <= 16
    with rtamd.corert.make_handle(m) as h:
        R, T = rtamd.corert.run_scene(h, sc)
        assert h.timers()["layer_launches"] > 1
    Rr, Tr = _oracle(cref, m)
    helpers.assert_stokes_close(R, Rr, what="R")
    helpers.assert_stokes_close(T, Tr, what="T")


def test_config_C2_full_size_stratified(rtamd, cref):
    """configs[1] at full size (N = 60, 40 layers, S = 10 000): finite, reproducible run to run, and equal to the
    C oracle on ALL 10 000 points (36 s of the box's 16 cores)."""
    m = rtamd.scenes.scene_C2()
    sc = rtamd.prepare_scene(m)
    assert (sc.N, sc.Nz, sc.S, sc.M) == (60, 40, 10_000, 3)
    with rtamd.corert.make_handle(m) as h:
        R, T = rtamd.corert.run_scene(h, sc)
        h.rt_run()
        R2, _ = h.get_RT()
        _, up, dw = h.get_hdr()
    assert np.all(np.isfinite(R)) and np.all(np.isfinite(T)) and np.array_equal(R, R2)
    assert np.all(R[:, 0, :] > 0)
    # size-independent property over ALL 10 000 points: the bihemispherical reflectance of a Lambertian surface is its
    # albedo -- the upward flux at the surface (interaction_hdrf.jl:9-45, through the whole layer sweep: it contains the
    # diffuse downward field J0+) over the downward flux, with sum(w mu) = 1/2 exact for the Gauss rule on [0, 1]
    np.testing.assert_allclose(up[0] / dw[0], m.params.brdf_albedo, rtol=1e-12)
    assert np.all(dw[0] > 0)
    Rr, Tr = _oracle(cref, m)
    helpers.assert_stokes_close(R, Rr, what="C2 all points R")
    helpers.assert_stokes_close(T, Tr, what="C2 all points T")


def test_config_C2_dual_full_size(rtamd):
    """rt_run on Dual numbers at the headline size (N = 60, 40 layers, S = 10 000, three partials) through size-independent
    properties on ALL points: the values are mom_rt_run's, the partials are linear in the direction (the third direction is
    2 x the first - 0.5 x the second), and the partial along the first direction is the central difference quotient of two
    VALUE runs of the fused kernels on scenes moved by +-eps along it (an independent code path: elemental -> doubling ->
    interaction in LDS-resident images against the streamed tangent-linear sweep)."""
    rt = rtamd.corert
    m = rtamd.scenes.scene_C2()
    sc = rtamd.prepare_scene(m)
    L = rt.construct_layer_inputs(m)
    rng = np.random.default_rng(2)
    p1 = rtamd.ScenePartial(dτ=L.τ * rng.uniform(0.0, 1.0, L.τ.shape), dϖ=-0.05 * L.ϖ * rng.uniform(0, 1, L.ϖ.shape), dalbedo=0.3)
    p2 = rtamd.ScenePartial(dτ=L.τ * rng.uniform(-1, 1, L.τ.shape), dzw=L.zw * rng.uniform(-1, 1, L.zw.shape), dalbedo=-1.0)
    comb = rtamd.ScenePartial(dτ=2.0 * p1.dτ - 0.5 * p2.dτ, dϖ=2.0 * p1.dϖ, dzw=-0.5 * p2.dzw, dalbedo=2.0 * 0.3 + 0.5)
    with rt.make_handle(m) as h:
        Rv, Tv = rt.run_scene(h, sc)
        rt.scene_set_partials(h, sc, [p1, p2, comb])
        h.rt_run_dual()
        R, T = h.get_RT()
        dR, dT = h.get_RT_partials()
        # two value runs of the fused kernels on the scene moved along p1 (tau, varpi and the albedo only: same ndoubl / interfaces)
        eps, fd = 1e-4, []
        for sgn in (1.0, -1.0):
            tau = L.τ + sgn * eps * p1.dτ
            varpi = L.ϖ + sgn * eps * p1.dϖ
            tau_sum = np.concatenate([np.zeros((tau.shape[0], 1)), np.cumsum(tau, axis=1)], axis=1)
            col = lambda a: np.ascontiguousarray(a.T).reshape(-1)
            h.scene_set(sc.Nz, sc.K, sc.M, col(tau), col(varpi), sc.zw, sc.Zpp, sc.Zmp, sc.ndoubl, sc.iface, col(tau_sum),
                        sc.albedo + sgn * eps * 0.3, sc.node, sc.cos_mphi, sc.sin_mphi)
            h.rt_run()
            fd.append(h.get_RT())
    assert np.all(np.isfinite(dR)) and np.all(np.isfinite(dT)) and np.abs(dR).max() > 0
    helpers.assert_stokes_close(R, Rv, what="C2 dual values vs mom_rt_run R")
    helpers.assert_stokes_close(T, Tv, what="C2 dual values vs mom_rt_run T")
    for d in (dR, dT):
        scale = np.abs(d[:2]).max(axis=(0, 2, 3), keepdims=True)[0]          # the view's largest partial
        assert np.abs(d[2] - (2.0 * d[0] - 0.5 * d[1])).max() <= 1e-10 * scale.max()
    for d, k in ((dR[0], 0), (dT[0], 1)):
        q = (fd[0][k] - fd[1][k]) / (2 * eps)
        scale = np.abs(d).max(axis=(1, 2), keepdims=True)
        # the quotient's own noise: each value run is within stokes_rtol(nd = 16) = 2.2e-10 of I of the exact result of the equations,
        # so the quotient carries up to 2.2e-10 I / eps = 2e-6 of I (measured 7.6e-7 of the largest partial); truncation is eps^2
        assert np.abs(d - q).max() <= 3e-6 * scale.max(), float(np.abs(d - q).max() / scale.max())


def test_config_C3_three_bands(rtamd, cref):
    """configs[2] on one GPU: 13 672 + 6 402 + 9 870 = 29 944 points on one spectral axis, same kernels as C2;
    3 072 points stratified over absorption depth against the oracle, and the 8-way spectral split of the multi-GPU run (global ndoubl)
    reproduces the first and the last shard bit for bit."""
    m = rtamd.scenes.scene_C3()
    sc = rtamd.prepare_scene(m)
    assert (sc.N, sc.S) == (60, 29_944)
    with rtamd.corert.make_handle(m) as h:
        R, T = rtamd.corert.run_scene(h, sc)
    assert np.all(np.isfinite(R)) and np.all(np.isfinite(T))
    pts = _stratified(m, 3_072)
    assert len(pts) >= 3_000
    Rr, Tr = _oracle(cref, m, pts=pts)
    helpers.assert_stokes_close(R[:, :, pts], Rr[:, :, pts], what="C3 sample R")
    helpers.assert_stokes_close(T[:, :, pts], Tr[:, :, pts], what="C3 sample T")
    for rank in (0, 7):
        lo, hi = rtamd.sharding.shard_bounds(sc.S, 8, rank)
        with rtamd.corert.make_handle(m, S=hi - lo) as h:
            Rs, Ts = rtamd.corert.run_scene(h, sc.spectral_slice(lo, hi))
        assert np.array_equal(Rs, R[:, :, lo:hi]) and np.array_equal(Ts, T[:, :, lo:hi])


def test_config_C4_iquv_64_streams(rtamd, cref):
    """configs[3]: aerosol + cloud, IQUV, 64 streams: 256 x 256 operators (the large-N kernels).  S = 256 points:
    finite, reproducible, ALL 256 points against the oracle (80 s of 16 cores)."""
    m = rtamd.scenes.scene_C4(S=256)
    sc = rtamd.prepare_scene(m)
    assert (sc.N, sc.nStokes, sc.Nz) == (256, 4, 40)
    with rtamd.corert.make_handle(m) as h:
        R, T = rtamd.corert.run_scene(h, sc)
        h.rt_run()
        R2, T2 = h.get_RT()
    assert np.all(np.isfinite(R)) and np.all(np.isfinite(T))
    assert np.array_equal(R, R2) and np.array_equal(T, T2)
    Rr, Tr = _oracle(cref, m)
    tol = helpers.stokes_rtol(sc.ndoubl)  # tau = 5 cloud: 24 doublings of 256 x 256 operators (arbiter: test_gpu_precision.py)
    helpers.assert_stokes_close(R, Rr, rtol=tol, what="C4 all points R")
    helpers.assert_stokes_close(T, Tr, rtol=tol, what="C4 all points T")


def _c5_line_subsets(RS):
    """(A, B): A = 24 of the scene's real Raman offsets -- the two extreme ones, 20 spread over the sorted list and the four
    strongest lines -- B = the rest.  Weights as in the full list (no renormalisation), so the subsets add up to it."""
    offs, w = np.asarray(RS.i_λ1λ0), np.asarray(RS.ϖ_λ1λ0)
    order = np.argsort(offs)
    A = np.unique(np.concatenate([[order[0], order[-1]], order[np.linspace(0, len(offs) - 1, 20).round().astype(int)],
                                  np.argsort(-w)[:4]]))
    B = np.setdiff1d(np.arange(len(offs)), A)
    return A, B


# owned windows of the C5 oracle comparisons: both grid edges (where the source index of half the lines runs off the grid),
# the middle, and the two stretches where the extreme offsets (+5014 / -4696) enter and leave the grid: 1 150 points
C5_WINDOWS = ((0, 300), (1_700, 1_850), (3_300, 3_550), (4_700, 4_850), (6_537, 6_837))


def _c5_compare(cref, scene, p, offs, w, g, strict, got, windows, what):
    """GPU 4-tuple (R, T, ieR, ieT) against the C oracle (oracle/momref.c ora_rt_run_rrs, ALL lines) on owned windows:
    elastic spectra 1e-10 of I; inelastic spectra 1e-10 of the elastic intensity of the same view and point, and 1e-9 of the
    largest inelastic value."""
    from oracle import rrsref as rr
    n = 0
    for lo, hi in windows:
        ora = rr.RRSInputs(offs.astype(np.int64), w, g, rrs_strict_reference=strict, owned=(lo, hi))
        Rr, Tr, ieRr, ieTr, info = cref.rt_run_rrs(scene, ora, p=p)
        assert info == 0
        own = slice(lo, hi)
        helpers.assert_stokes_close(got[0][..., own], Rr[..., own], what=f"{what} R [{lo},{hi})")
        helpers.assert_stokes_close(got[1][..., own], Tr[..., own], what=f"{what} T [{lo},{hi})")
        assert np.abs(ieRr[..., own]).max() > 0
        for x, ref, el in ((got[2], ieRr, Rr), (got[3], ieTr, Tr)):
            scale = np.abs(el[:, 0:1, own])
            d = np.abs(x[..., own] - ref[..., own])
            assert np.all(d <= 1e-10 * scale + 1e-14), (what, lo, hi, d.max())
            helpers.assert_op_close(x[..., own], ref[..., own], 1e-9, f"{what} ie spectra [{lo},{hi})")
        n += hi - lo
    return n


def test_config_C5_rotational_raman(rtamd, cref):
    """configs[4] at FULL size on the GPU: S = 6 837, N = 15, 5 layers, all 178 Raman offsets (|Δn| up to 5 014 grid points;
    804 233 valid (n₁, Δn) pairs).

      (1) the full run -- ALL 178 lines -- against the C oracle (oracle/momref.c ora_rt_run_rrs, the C port of oracle/rrsref.py)
          on 1 150 owned points in five windows incl. both grid edges, corrected switch position;
      (1s) the STRICT position (the reference's text as written, defects D1..D5 included) at full size: finite, and equal to
          the C oracle's strict run on 450 owned points in three windows;
      (2) linearity in the line list (each Δn evolves independently of the others in every operator of the path,
          doubling_inelastic.jl:61-125, interaction_inelastic.jl:249-335): ie spectra of the 178-line run = A-run + B-run,
          and the elastic spectra of all three runs are bitwise equal;
      (3) elastic limit at full size: zero Raman weights give zero inelastic spectra and the elastic spectra of (2), and the
          elastic spectra equal rt_run(::noRS) with the Cabannes albedo to 1e-10;
      (4) the 2-way spectral split (window = owned + halo of 5 014) reproduces the full run bit for bit."""
    from oracle import momref as mr
    rt = rtamd.corert
    m, RS = rtamd.scenes.scene_C5()
    S, nR = m.τ_rayl.shape[0], RS.n_Raman
    offs, w = np.asarray(RS.i_λ1λ0), np.asarray(RS.ϖ_λ1λ0)
    assert (S, nR) == (6_837, 178) and rtamd.prepare_scene(m).N == 15
    assert np.abs(offs).max() == 5_014 and offs.min() == -4_696
    sub = lambda k, strict=False: rt.RRS(greek_raman=RS.greek_raman, ϖ_Cabannes=RS.ϖ_Cabannes, ϖ_λ1λ0=w[k], i_λ1λ0=offs[k],
                                         rrs_strict_reference=strict)
    allk = np.arange(nR)
    A, B = _c5_line_subsets(RS)
    full = rt.rt_run_rrs(sub(allk), m)
    gA = rt.rt_run_rrs(sub(A), m)
    gB = rt.rt_run_rrs(sub(B), m)
    for x in full:
        assert np.all(np.isfinite(x))
    assert np.abs(full[2]).max() > 1e-4 and np.abs(full[3]).max() > 1e-5
    # (2) linearity and the elastic part
    for k in (0, 1, 4, 5, 6):
        assert np.array_equal(full[k], gA[k]) and np.array_equal(full[k], gB[k]), k
    for k in (2, 3):
        helpers.assert_op_close(gA[k] + gB[k], full[k], 1e-12, f"C5 linearity [{k}]")
    # (1) the C oracle, all 178 lines
    scene = helpers.oracle_scene(m)
    scene.varpi_cabannes = RS.ϖ_Cabannes
    g = mr.get_greek_rayleigh(0.75)
    p = cref.pack_scene(scene)
    assert _c5_compare(cref, scene, p, offs, w, g, False, full, C5_WINDOWS, "C5") >= 1_000
    # (1s) the strict position at full size (all interfaces of this scene are 11, so D4 does not raise)
    strict = rt.rt_run_rrs(sub(allk, True), m)
    for x in strict[:4]:
        assert np.all(np.isfinite(x))
    assert not np.allclose(strict[0], full[0], rtol=1e-6)                   # D1 advances the elastic sources nRaman times per step
    _c5_compare(cref, scene, p, offs, w, g, True, strict, ((0, 150), (3_300, 3_450), (6_687, 6_837)), "C5 strict")
    # (3) elastic limit
    zero = rt.RRS(greek_raman=RS.greek_raman, ϖ_Cabannes=RS.ϖ_Cabannes, ϖ_λ1λ0=np.zeros(nR), i_λ1λ0=offs, rrs_strict_reference=False)
    g0 = rt.rt_run_rrs(zero, m)
    assert not np.any(g0[2]) and not np.any(g0[3])
    assert np.array_equal(g0[0], full[0]) and np.array_equal(g0[1], full[1])
    Re, Te = rt.rt_run(rt._with_cabannes(RS, m))[:2]
    helpers.assert_stokes_close(full[0], Re, what="C5 elastic R vs rt_run(noRS)")
    helpers.assert_stokes_close(full[1], Te, what="C5 elastic T vs rt_run(noRS)")
    # (4) two windows
    parts = []
    for rank in range(2):
        lo, hi, wlo, whi = rtamd.sharding.rrs_window(S, 2, rank, offs)
        parts.append(rt.rt_run_rrs_window(sub(allk), m, lo, hi, (wlo, whi)))
    for k in range(7):
        assert np.array_equal(np.concatenate([p_[k] for p_ in parts], axis=-1), full[k]), k
