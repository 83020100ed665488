"""Scene-level parity on the GPU through the C ABI: mom_rt_run (fused per-layer kernels) and the
operator-by-operator replay against the C oracle, the golden vectors and the reference's tables."""
import json
from pathlib import Path

import numpy as np
import pytest

import helpers
from oracle import momref as mr

pytestmark = pytest.mark.gpu
GOLD = Path(__file__).parent / "golden"


def _oracle(cref, model, pts=None):
    p = cref.pack_scene(helpers.oracle_scene(model))
    R, T, info = cref.rt_run(p, pts=pts)
    assert info == 0
    return R, T


def _gpu(rtamd, model, generic=False, force_gj=False, inverse=None):
    sc = rtamd.prepare_scene(model)
    with rtamd.corert.make_handle(model) as h:
        if generic:
            h.set_option(rtamd._lib.MOM_OPT_FORCE_GENERIC, 1)
        if force_gj:
            h.set_option(rtamd._lib.MOM_OPT_INVERSE, 1)
        if inverse is not None:
            h.set_option(rtamd._lib.MOM_OPT_INVERSE, inverse)
        return rtamd.corert.run_scene(h, sc)


@pytest.mark.parametrize("nS,lt", [(1, 3), (3, 9), (4, 7), (3, 33)])
@pytest.mark.parametrize("mode", ["lds", "generic", "gj"])
def test_rt_run_parity(rtamd, cref, nS, lt, mode):
    m = rtamd.scenes.make_scene(nS, lt, 6, 24, seed=nS + lt)
    R, T = _gpu(rtamd, m, generic=mode == "generic", force_gj=mode == "gj")
    Rr, Tr = _oracle(cref, m)
    helpers.assert_stokes_close(R, Rr, what=f"R {mode}")
    helpers.assert_stokes_close(T, Tr, what=f"T {mode}")


# operator edges above 64 (generic mode, panel GEMM from the slab): every register-block variant of wg_gemm_big (4 x 2, 4 x 3,
# 4 x 4 tiles), one and two passes, an odd edge (element copies, k tail of the last panel) and the slab mat-vec / copy helpers
@pytest.mark.parametrize("nS,lt,N", [(3, 43, 75), (4, 43, 100), (3, 79, 129), (4, 65, 144), (4, 93, 200)])
def test_rt_run_parity_generic_sizes(rtamd, cref, nS, lt, N):
    m = rtamd.scenes.make_scene(nS, lt, 3, 4, seed=nS + lt, aerosol_total=0.3)
    sc = rtamd.prepare_scene(m)
    assert sc.N == N
    R, T = _gpu(rtamd, m)
    Rr, Tr = _oracle(cref, m)
    tol = helpers.stokes_rtol(sc.ndoubl)
    helpers.assert_stokes_close(R, Rr, rtol=tol, what=f"R N={N}")
    helpers.assert_stokes_close(T, Tr, rtol=tol, what=f"T N={N}")
    R3, T3 = _gpu(rtamd, m, force_gj=True)
    helpers.assert_stokes_close(R3, Rr, rtol=1e-11, what="R gauss-jordan")
    helpers.assert_stokes_close(T3, Tr, rtol=1e-11, what="T gauss-jordan")


# 64 < N <= 96: the doubling loop runs with register-resident operators (csrc/mom_regdbl.hpp: tiles dealt over the 8 waves,
# operands staged through two LDS slots), elemental and interaction in the slab as before.  5 x 5 tile grids (N = 66, 75,
# 80) and 6 x 6 (N = 84, 96: no spare column / full tiles), N odd, thin and thick layers: Horner (p <= 4), the squaring
# series (p <= 512) and the hand-back to the general path (pivoted inverse beyond).  inverse = 2 switches the register
# path off (the r3 slab path): both against the oracle and against each other.
@pytest.mark.parametrize("nS,lt,N,kw", [(3, 37, 66, {}), (3, 43, 75, {}), (4, 33, 80, {}), (3, 49, 84, {}), (4, 41, 96, {}),
                                        (3, 49, 84, dict(aerosol_total=3.0, absorption=False)),
                                        (4, 41, 96, dict(aerosol_total=8.0, absorption=False, aerosol_p0=600.0, aerosol_σp=250.0)),
                                        (1, 143, 75, dict(aerosol_total=1.0))])
def test_register_resident_doubling_sizes(rtamd, cref, nS, lt, N, kw):
    kw = dict(dict(aerosol_total=0.3), **kw)
    m = rtamd.scenes.make_scene(nS, lt, 4, 6, seed=nS + lt, **kw)
    sc = rtamd.prepare_scene(m)
    assert sc.N == N
    Rr, Tr = _oracle(cref, m)
    tol = helpers.stokes_rtol(sc.ndoubl)
    R, T = _gpu(rtamd, m)
    helpers.assert_stokes_close(R, Rr, rtol=tol, what=f"R N={N}")
    helpers.assert_stokes_close(T, Tr, rtol=tol, what=f"T N={N}")
    R2, T2 = _gpu(rtamd, m, inverse=2)
    helpers.assert_stokes_close(R2, Rr, rtol=tol, what=f"R N={N} slab path")
    helpers.assert_stokes_close(R, R2, rtol=tol, what="register path vs slab path R")
    helpers.assert_stokes_close(T, T2, rtol=tol, what="register path vs slab path T")
    assert not np.array_equal(R, R2)     # the two paths really are different code (summation order of the mat-vecs)


# operator sizes with strip-chained kernels (mom_strip.hpp): N = 52, 56, 60 in the 8-wave build (IQUV with 13, 14,
# 15 streams; scalar with 60), N = 36, 40, 44 in the 4-wave build (scalar scenes; the m = 0 (I,Q) sub-problems of
# IQU scenes with 18, 20, 22 streams -- their full problems have N = 54 (plain LDS path), 60, 66 (generic path))
@pytest.mark.parametrize("nS,lt", [(4, 19), (4, 21), (4, 23), (1, 113), (1, 65), (1, 73), (1, 81), (3, 29), (3, 37)])
def test_rt_run_parity_strip_sizes(rtamd, cref, nS, lt):
    m = rtamd.scenes.make_scene(nS, lt, 5, 10, seed=3 * nS + lt, aerosol_total=0.6)
    R, T = _gpu(rtamd, m)
    Rr, Tr = _oracle(cref, m)
    tol = helpers.stokes_rtol(rtamd.prepare_scene(m).ndoubl)  # aerosol tau 0.6: up to 21 doublings
    helpers.assert_stokes_close(R, Rr, rtol=tol, what="R strip")
    helpers.assert_stokes_close(T, Tr, rtol=tol, what="T strip")
    R2, T2 = _gpu(rtamd, m, inverse=2)  # same kernels with the strip chains switched off
    helpers.assert_stokes_close(R2, Rr, rtol=tol, what="R no strip")
    helpers.assert_stokes_close(R, R2, rtol=tol, what="strip vs plain R")
    helpers.assert_stokes_close(T, T2, rtol=tol, what="strip vs plain T")
    R3, T3 = _gpu(rtamd, m, force_gj=True)  # pivoted Gauss-Jordan on I - B like the oracle's LU: same rounding pattern
    helpers.assert_stokes_close(R3, Rr, rtol=1e-11, what="R gauss-jordan")
    helpers.assert_stokes_close(T3, Tr, rtol=1e-11, what="T gauss-jordan")


@pytest.mark.parametrize("nS,lt,N,kw", [(3, 19, 39, {}), (3, 21, 42, {}), (3, 27, 51, {}), (3, 29, 54, {}), (3, 31, 57, {}),
                                        (1, 69, 38, {}), (1, 91, 49, {}), (3, 25, 48, {}), (4, 9, 32, {}),
                                        (3, 27, 51, dict(brdf="rpv"))])
def test_strip_padding_to_kernel_sizes(rtamd, cref, nS, lt, N, kw):
    """MOM_OPT_STRIP_PAD: operator edges without a strip-chained kernel image (most IQU stream counts, N = 32, 48) run the
    kernels of the next size that has one, with up to 4 decoupled dummy stream entries; the m = 0 sub-problem pads itself the same way (N0 = 34 -> 36,
    38 -> 40).  Same results as the unpadded general path and as the oracle."""
    kw = dict(kw)
    brdf = kw.pop("brdf", None)
    m = rtamd.scenes.make_scene(nS, lt, 5, 12, seed=5 * nS + lt, aerosol_total=0.4, **kw)
    if brdf:
        m.params.brdf = rtamd.corert.rpvSurfaceScalar(0.12, 0.7, -0.15, 0.9)
    sc = rtamd.prepare_scene(m)
    assert sc.N == N
    out = {}
    for pad in (1, 0):
        with rtamd.corert.make_handle(m) as h:
            h.set_option(rtamd._lib.MOM_OPT_STRIP_PAD, pad)
            R, T = rtamd.corert.run_scene(h, sc)
            out[pad] = (R, T) + h.get_hdr()
    Rr, Tr = _oracle(cref, m)
    tol = helpers.stokes_rtol(sc.ndoubl)
    for pad in (1, 0):
        helpers.assert_stokes_close(out[pad][0], Rr, rtol=tol, what=f"R pad={pad}")
        helpers.assert_stokes_close(out[pad][1], Tr, rtol=tol, what=f"T pad={pad}")
    helpers.assert_stokes_close(out[1][2], out[0][2], rtol=tol, what="hdr padded vs unpadded")
    np.testing.assert_allclose(out[1][3], out[0][3], rtol=max(tol, 1e-10), atol=helpers.ATOL_STOKES)
    np.testing.assert_allclose(out[1][4], out[0][4], rtol=max(tol, 1e-10), atol=helpers.ATOL_STOKES)


@pytest.mark.parametrize("nS,lt,N,thick", [(4, 11, 36, False), (4, 13, 40, False), (4, 13, 40, True), (4, 11, 36, True),
                                           (3, 33, 60, False), (3, 33, 60, True)])
def test_lean_strip_image_and_resume(rtamd, cref, nS, lt, N, thick):
    """MOM_OPT_LEAN (r5): operators of edge 36 / 40 -- and the m = 0 (I,Q) sub-problem of an N = 60 IQU scene, N0 = 40 -- run on
    the lean 4-wave strip image (three operator buffers, three workgroups per CU; csrc/mom_lean.hpp) followed by the full
    image's resume launch.  Thin layers: every unit finishes in the lean image.  Thick layers (aerosol optical depth 2: series
    beyond 12 terms in the late doubling steps and the interactions): the lean workgroup leaves the unit at that layer and the
    full image redoes the layer and finishes the unit.  The four-wave lean image (MOM_OPT_LEAN = 1) is BITWISE the full image (the
    chains perform the same operations in the same order); the six-wave one (= 2, default: half-strip doubling chains) agrees with
    the oracle at the suite's tolerance."""
    kw = dict(aerosol_total=2.0, aerosol_p0=600.0, aerosol_σp=200.0, absorption=False) if thick else dict(aerosol_total=0.3)
    m = rtamd.scenes.make_scene(nS, lt, 6, 40, seed=11 * nS + lt, **kw)
    sc = rtamd.prepare_scene(m)
    assert sc.N == N
    out = {}
    for lean in (3, 2, 1, 0):
        with rtamd.corert.make_handle(m) as h:
            h.set_option(rtamd._lib.MOM_OPT_LEAN, lean)
            R, T = rtamd.corert.run_scene(h, sc)
            out[lean] = (R, T) + h.get_hdr() + (h.timers()["layer_launches"],)
            R2, T2 = rtamd.corert.run_scene(h, sc)                       # the resume table is reused: same answer again
            assert np.array_equal(R, R2) and np.array_equal(T, T2)
    for k in range(5):
        assert np.array_equal(out[1][k], out[0][k]), f"four-wave lean vs full image, output {k}"
    assert out[1][5] > out[0][5] and out[2][5] > out[0][5] and out[3][5] > out[0][5]   # the lean launch + the resume launch
    Rr, Tr = _oracle(cref, m)
    tol = helpers.stokes_rtol(sc.ndoubl)
    # the six-wave image sums a contraction in two halves, the quad-block image (= 3: one wavefront per unit, 4 x 4 x 4 MFMA
    # blocks; csrc/mom_q4.hpp) in blocks of four: equal to the oracle's tolerance, not bitwise
    for lean in (3, 2, 1):
        helpers.assert_stokes_close(out[lean][0], Rr, rtol=tol, what=f"R lean={lean}")
        helpers.assert_stokes_close(out[lean][1], Tr, rtol=tol, what=f"T lean={lean}")
    helpers.assert_stokes_close(out[2][2], out[0][2], rtol=tol, what="hdr six-wave lean vs full")
    helpers.assert_stokes_close(out[3][2], out[0][2], rtol=tol, what="hdr quad-block vs full")
    helpers.assert_stokes_close(out[3][0], out[0][0], rtol=min(tol, 1e-10), what="R quad-block vs full")


@pytest.mark.parametrize("nS,lt,N,small,kw", [
    (4, 3, 20, 0, {}), (4, 5, 24, 0, {}), (4, 7, 28, 0, dict(aerosol_total=0.8)), (4, 9, 32, 0, {}), (3, 9, 24, 0, {}),   # full problems
    (4, 17, 48, 1, {}), (4, 21, 56, 1, {}), (4, 23, 60, 1, {}), (4, 19, 52, 1, dict(aerosol_total=2.0, aerosol_p0=600.0, aerosol_σp=200.0)),
    (3, 23, 45, 1, {})])                                                                                                  # m = 0 sub-problems
def test_quad_block_image_small_edges(rtamd, cref, nS, lt, N, small, kw):
    """The quad-block image (csrc/mom_q4.hpp) at the edges 20 .. 32: full problems on the general route (MOM_OPT_SMALL_N = 0; the
    wave-per-point kernel is the default there) and the m = 0 (I,Q) sub-problems of larger IQUV / IQU scenes (N0 = 24, 28, 30 -> 32
    by one dummy stream, 26 -> 28; a thick scene whose late steps go to the finishing launch of the general image), against the
    oracle and against MOM_OPT_LEAN = 1 (no quad-block image)."""
    m = rtamd.scenes.make_scene(nS, lt, 6, 48, seed=5 * nS + lt, **kw)
    sc = rtamd.prepare_scene(m)
    assert sc.N == N
    Rr, Tr, Hr, upr, dwr, info = cref.rt_run_full(cref.pack_scene(helpers.oracle_scene(m)))
    assert info == 0
    out = {}
    for lean in (3, 1):
        with rtamd.corert.make_handle(m) as h:
            h.set_option(rtamd._lib.MOM_OPT_SMALL_N, small)
            h.set_option(rtamd._lib.MOM_OPT_LEAN, lean)
            R, T = rtamd.corert.run_scene(h, sc)
            out[lean] = (R, T) + h.get_hdr() + (h.timers()["layer_launches"],)
    assert out[3][5] >= out[1][5]                     # the quad-block launch + the finishing launch behind it
    tol = helpers.stokes_rtol(sc.ndoubl)
    for lean in (3, 1):
        helpers.assert_stokes_close(out[lean][0], Rr, rtol=tol, what=f"R lean={lean}")
        helpers.assert_stokes_close(out[lean][1], Tr, rtol=tol, what=f"T lean={lean}")
        helpers.assert_stokes_close(out[lean][2], Hr, rtol=tol, what=f"hdr lean={lean}")
        np.testing.assert_allclose(out[lean][4], dwr, rtol=max(tol, 1e-10), atol=helpers.ATOL_STOKES)


@pytest.mark.parametrize("nS,lt,N", [(4, 13, 40), (3, 33, 60), (4, 21, 56)])
def test_quad_block_image_layers_without_doublings(rtamd, cref, nS, lt, N):
    """Layers thin enough for ndoubl = 0 (the elemental layer IS the added layer: no doubling step, no D signs from one) and a
    layer without any scattering at one spectral point (series of one term) through the quad-block image -- as a full problem
    (N = 40) and as the m = 0 sub-problem (N0 = 40, 28)."""
    m = rtamd.scenes.make_scene(nS, lt, 6, 32, seed=3 * nS + lt, aerosol_total=0.05)
    m.τ_rayl[:, :3] *= 1e-6
    m.τ_abs[:, :3] *= 1e-6
    m.τ_aer[:, :3] *= 1e-6
    sc = rtamd.prepare_scene(m)
    assert sc.N == N and sc.ndoubl[0] == 0 and sc.ndoubl[-1] > 0 and np.all(sc.iface == 3)
    Rr, Tr, Hr, upr, dwr, info = cref.rt_run_full(cref.pack_scene(helpers.oracle_scene(m)))
    assert info == 0
    with rtamd.corert.make_handle(m) as h:
        R, T = rtamd.corert.run_scene(h, sc)
        H, up, dw = h.get_hdr()
    tol = helpers.stokes_rtol(sc.ndoubl)
    helpers.assert_stokes_close(R, Rr, rtol=tol, what="R")
    helpers.assert_stokes_close(T, Tr, rtol=tol, what="T")
    helpers.assert_stokes_close(H, Hr, rtol=tol, what="hdr")


def test_strip_chains_thick_layers_fall_back(rtamd, cref):
    """optically thick scattering layers: the series length exceeds the strip chains' limit for part of the
    doubling steps and interactions, which must then take the general path inside the same kernels"""
    m = rtamd.scenes.make_scene(3, 33, 4, 8, seed=77, aerosol_total=2.0, aerosol_p0=600.0, aerosol_σp=200.0, absorption=False)
    R, T = _gpu(rtamd, m)
    Rr, Tr = _oracle(cref, m)
    tol = helpers.stokes_rtol(rtamd.prepare_scene(m).ndoubl)
    helpers.assert_stokes_close(R, Rr, rtol=tol, what="R thick")
    helpers.assert_stokes_close(T, Tr, rtol=tol, what="T thick")


def test_launch_shape_options_do_not_change_results(rtamd, cref):
    """MOM_OPT_STAGGER only delays workgroup starts, MOM_OPT_OVERLAP only moves the m = 0 sub-problem to the handle's second
    stream and MOM_OPT_SMALL_WG only picks the workgroup shape: the staggered and the one-stream runs are bitwise the default
    run (also when repeated on one handle: fork / join ordering), the 8-wave run of the m = 0 sub-problem agrees to rounding."""
    m = rtamd.scenes.make_scene(3, 33, 6, 2400, seed=8)   # N = 60, enough units for the persistent grid to stagger
    sc = rtamd.prepare_scene(m)
    out = {}
    for key, opt, val in (("default", None, None), ("nostagger", rtamd._lib.MOM_OPT_STAGGER, 0),
                          ("onestream", rtamd._lib.MOM_OPT_OVERLAP, 0), ("wg8", rtamd._lib.MOM_OPT_SMALL_WG, 0)):
        with rtamd.corert.make_handle(m) as h:
            if opt is not None:
                h.set_option(opt, val)
            out[key] = rtamd.corert.run_scene(h, sc)
            if key == "default":
                for _ in range(3):          # back-to-back runs without a host sync in between
                    h.rt_run()
                again = h.get_RT()
                np.testing.assert_array_equal(out[key][0], again[0])
                np.testing.assert_array_equal(out[key][1], again[1])
    for other in ("nostagger", "onestream"):
        np.testing.assert_array_equal(out["default"][0], out[other][0])
        np.testing.assert_array_equal(out["default"][1], out[other][1])
    helpers.assert_stokes_close(out["wg8"][0], out["default"][0], rtol=1e-11, what="8-wave vs 4-wave m = 0")
    pts = np.arange(0, 2400, 300)
    Rr, Tr = _oracle(cref, m, pts=pts)
    helpers.assert_stokes_close(out["default"][0][:, :, pts], Rr[:, :, pts], what="R")


@pytest.mark.parametrize("strict", [True, False])
def test_rt_run_iquv_indexing_switch(rtamd, cref, strict):
    m = rtamd.scenes.make_scene(4, 9, 4, 8, seed=9, vaz=(90.0, 10.0, 170.0))
    m.params.strict_reference_indexing = strict
    R, T = _gpu(rtamd, m)
    Rr, Tr = _oracle(cref, m)
    helpers.assert_stokes_close(R, Rr, what="R")
    helpers.assert_stokes_close(T, Tr, what="T")


def test_operator_replay_equals_fused(rtamd, cref):
    """rt_run replayed op by op (elemental!/doubling!/interaction! like a Julia shim would) gives the
    fused kernels' result to rounding, and both match the oracle."""
    m = rtamd.scenes.make_scene(3, 9, 5, 12, seed=21)
    R1, T1 = rtamd.rt_run(m)[:2]
    R2, T2 = rtamd.rt_run_operators(m)
    Rr, Tr = _oracle(cref, m)
    helpers.assert_stokes_close(R1, Rr, what="fused R")
    helpers.assert_stokes_close(R2, Rr, what="replay R")
    helpers.assert_stokes_close(T2, Tr, what="replay T")
    helpers.assert_stokes_close(R1, R2, rtol=1e-11, what="fused vs replay")


def test_all_interface_cases_and_zero_doublings(rtamd, cref):
    """Non-scattering layers drive the interface state machine through 00, 01, 10, 11
    (rt_helper_functions.jl:8-27) and ndoubl = 0 (doubling.jl:28)."""
    m = rtamd.scenes.make_scene(3, 7, 5, 10, aerosol_total=0.0, seed=2)
    for z in (0, 1, 3):
        m.τ_rayl[:, z] = 0.0
    sc = rtamd.prepare_scene(m)
    assert list(sc.iface) == [0, 0, 1, 2, 3] and sc.ndoubl[0] == 0
    for generic in (False, True):
        R, T = _gpu(rtamd, m, generic=generic)
        Rr, Tr = _oracle(cref, m)
        helpers.assert_stokes_close(R, Rr, what="R")
        helpers.assert_stokes_close(T, Tr, what="T")
    # surface after a non-scattering last layer: interface code 10 is used for the surface (Q6)
    m2 = rtamd.scenes.make_scene(1, 5, 3, 6, aerosol_total=0.0, seed=3)
    m2.τ_rayl[:, 2] = 0.0
    assert list(rtamd.prepare_scene(m2).iface) == [3, 3, 2]
    helpers.assert_stokes_close(_gpu(rtamd, m2)[0], _oracle(cref, m2)[0], what="R iface10 surface")


def test_interface_cases_in_strip_kernels(rtamd, cref):
    """The same interface state machine at N = 60 and N0 = 40: the strip-chained kernel images (8-wave and, for the
    m = 0 sub-problem, 4-wave) contain the general code of the 00 / 01 / 10 cases with the pitched composite blocks."""
    m = rtamd.scenes.make_scene(3, 33, 5, 10, aerosol_total=0.0, seed=4)
    for z in (0, 1, 3):
        m.τ_rayl[:, z] = 0.0
    sc = rtamd.prepare_scene(m)
    assert sc.N == 60 and list(sc.iface) == [0, 0, 1, 2, 3]
    R, T = _gpu(rtamd, m)
    Rr, Tr = _oracle(cref, m)
    helpers.assert_stokes_close(R, Rr, what="R")
    helpers.assert_stokes_close(T, Tr, what="T")


def test_golden_small_iqu(rtamd):
    g = np.load(GOLD / "small_iqu.npz")
    m = rtamd.scenes.make_scene(3, 3, 3, 4, vza=(0.0,), vaz=(35.0,), seed=7, aerosol_total=0.3, aerosol_p0=500.0,
                                aerosol_σp=300.0)
    R, T = rtamd.rt_run(m)[:2]
    helpers.assert_stokes_close(R, g["R"], what="R vs golden")
    helpers.assert_stokes_close(T, g["T"], what="T vs golden")
    # per-operator golden: doubling iterations 1 and 2 of the thickest layer
    z = int(g["dbl_layer"]) - 1
    sc = rtamd.prepare_scene(m)
    S, N, Nz = sc.S, sc.N, sc.Nz
    nd = int(sc.ndoubl[z])
    dtau = sc.tau.reshape(Nz, S)[z] / 2 ** nd
    L = rtamd.corert.construct_layer_inputs(m)
    Zpp, Zmp = rtamd.corert.z_bases(m)
    Zp = np.einsum("ks,kij->sij", L.zw[:, :, z], Zpp[0])
    Zm = np.einsum("ks,kij->sij", L.zw[:, :, z], Zmp[0])
    with rtamd.corert.make_handle(m) as h:
        for it in (1, 2):
            h.elemental(0, nd, sc.tau_sum.reshape(Nz + 1, S)[z], dtau, sc.varpi.reshape(Nz, S)[z], mr.to_abi(Zp),
                        mr.to_abi(Zm), S)
            h.doubling(it, np.exp(-dtau / m.quad_points.μ0))
            # undo the D signs the operator applies after its last iteration (identity for strict IQU)
            helpers.assert_op_close(h.download(3), mr.to_abi(g[f"dbl_iter{it}_z{z + 1}_t_pp"]), rtol=1e-12, what="t_pp")
            helpers.assert_op_close(h.download(1), mr.to_abi(g[f"dbl_iter{it}_z{z + 1}_r_mp"]), rtol=1e-12, what="r_mp")
            helpers.assert_op_close(h.download(4), mr.to_abi(g[f"dbl_iter{it}_z{z + 1}_j0p"]), rtol=1e-12, what="j0p")


def test_natraj_on_gpu(rtamd):
    """test/test_CoreRT.jl:40-83 end to end on the GPU (N = 136 -> generic kernels, ndoubl = 18):
    the reference's thresholds against the Natraj tables, and 1e-10 against the oracle's stored output."""
    import test_oracle_reference_tables as t
    g = np.load(GOLD / "natraj.npz")
    rt = rtamd.corert
    m = helpers.one_layer_rayleigh(rt, float(np.degrees(np.arccos(0.2))), g["vza"], g["vaz"], 0.5, 0.0)
    sc = rtamd.prepare_scene(m)
    assert sc.N == 136 and list(sc.ndoubl) == [18]
    np.testing.assert_allclose(sc.Zmp, g["Zmp"], rtol=0, atol=1e-13)
    R, T = rtamd.rt_run(m)[:2]
    helpers.assert_stokes_close(R, g["R"], what="Natraj R vs oracle")
    eI, eQ, eU = t.natraj_errors(R)
    assert eI < 0.002 and eQ < 0.008 and eU < 0.008


@pytest.mark.parametrize("case", [0, 1, 2, 3, 4, 5])
def test_6sv1_on_gpu(rtamd, cref, case):
    """test/test_CoreRT.jl:3-38 on the GPU: R/μ₀ within 0.006 of the 6SV1 tables, and every one of the scenes at the
    1e-10 level against the oracle run of the same scene."""
    G = json.loads((GOLD / "reference_tables.json").read_text())
    c = G["sixsv_cases"][case]
    Rt = np.array(G["sixsv_R"][case])
    vza1 = np.array(G["sixsv_vza"])
    for si, sza in enumerate(c["sza"]):
        m = helpers.one_layer_rayleigh(rtamd.corert, sza, np.tile(vza1, 3), np.repeat(np.array(c["az"], float), 16),
                                       c["tau"], c["rho"])
        R, T = rtamd.rt_run(m)[:2]
        Rm = (R[:, 0, 0] / m.quad_points.μ0).reshape(3, 16)
        assert np.max(np.abs(Rt[si] - Rm) / Rt[si]) < 0.006
        Rr, Tr = _oracle(cref, m)
        nd = int(rtamd.prepare_scene(m).ndoubl.max())
        helpers.assert_stokes_close(R, Rr, rtol=helpers.stokes_rtol(nd), what=f"6SV1 case {case} sza {sza} R vs oracle")
        helpers.assert_stokes_close(T, Tr, rtol=helpers.stokes_rtol(nd), what=f"6SV1 case {case} sza {sza} T vs oracle")


def _natraj_model(rtamd, pol, strict):
    g = np.load(GOLD / "natraj.npz")
    m = helpers.one_layer_rayleigh(rtamd.corert, float(np.degrees(np.arccos(0.2))), g["vza"], g["vaz"], 0.5, 0.0, pol=pol)
    m.params.strict_reference_indexing = strict
    return m


def test_natraj_iqu_nonstrict_on_gpu(rtamd, cref):
    """The nStokes = 3 code path (N = 102: panel-GEMM kernels, ndoubl = 18) on the reference-held Natraj tables: for
    Rayleigh V decouples, so Stokes_IQU with the zero-based Stokes rule must meet the thresholds of test_CoreRT.jl:40-83
    with the IQUV run's own error fingerprints; 1e-10-level against the oracle on the same scene."""
    import test_oracle_reference_tables as t
    m = _natraj_model(rtamd, rtamd.corert.Stokes_IQU(), strict=False)
    sc = rtamd.prepare_scene(m)
    assert sc.N == 102 and list(sc.ndoubl) == [18]
    R = rtamd.rt_run(m)[0]
    eI, eQ, eU = t.natraj_errors(R)
    assert eI < 0.002 and eQ < 0.008 and eU < 0.008
    assert abs(eI - 1.3668e-3) < 2e-7 and abs(eQ - 7.7745e-3) < 2e-7 and abs(eU - 3.7466e-3) < 2e-7
    Rr, _ = _oracle(cref, m)
    helpers.assert_stokes_close(R, Rr, rtol=helpers.stokes_rtol(18), what="Natraj IQU R vs oracle")
    g = np.load(GOLD / "natraj.npz")
    np.testing.assert_allclose(R[:, :3, 0], g["R"][:, :3, 0], rtol=0, atol=5e-10)      # the IQUV oracle's stored I, Q, U


def test_natraj_iqu_strict_q1_fingerprint_on_gpu(rtamd, cref):
    """Stokes_IQU as the reference's text has it (quirk Q1): the fingerprint SURVEY 8a-Q1 recorded, on the GPU."""
    import test_oracle_reference_tables as t
    m = _natraj_model(rtamd, rtamd.corert.Stokes_IQU(), strict=True)
    R3 = rtamd.rt_run(m)[0]
    g = np.load(GOLD / "natraj.npz")
    t.assert_q1_fingerprint(R3, g["R"])
    Rr, _ = _oracle(cref, m)
    helpers.assert_stokes_close(R3, Rr, rtol=helpers.stokes_rtol(18), what="Natraj IQU strict R vs oracle")


@pytest.mark.parametrize("case", [0, 1, 2, 3, 4, 5])
def test_6sv1_iqu_nonstrict_on_gpu(rtamd, cref, case):
    """The six 6SV1 cases as Stokes_IQU (zero-based rule) on the GPU: ε = 0.006 on R/μ₀; cases 2, 4, 6 carry the
    Lambertian surface with ρ = 0.25 (lambertian_surface.jl:20-75) into the nStokes = 3 pin; each scene also at the 1e-10
    level against the oracle run of the same scene."""
    G = json.loads((GOLD / "reference_tables.json").read_text())
    c = G["sixsv_cases"][case]
    Rt = np.array(G["sixsv_R"][case])
    vza1 = np.array(G["sixsv_vza"])
    for si, sza in enumerate(c["sza"]):
        m = helpers.one_layer_rayleigh(rtamd.corert, sza, np.tile(vza1, 3), np.repeat(np.array(c["az"], float), 16),
                                       c["tau"], c["rho"], pol=rtamd.corert.Stokes_IQU())
        m.params.strict_reference_indexing = False
        R, T = rtamd.rt_run(m)[:2]
        Rm = (R[:, 0, 0] / m.quad_points.μ0).reshape(3, 16)
        assert np.max(np.abs(Rt[si] - Rm) / Rt[si]) < 0.006
        Rr, Tr = _oracle(cref, m)
        nd = int(rtamd.prepare_scene(m).ndoubl.max())
        helpers.assert_stokes_close(R, Rr, rtol=helpers.stokes_rtol(nd), what=f"6SV1 IQU case {case} sza {sza} R vs oracle")
        helpers.assert_stokes_close(T, Tr, rtol=helpers.stokes_rtol(nd), what=f"6SV1 IQU case {case} sza {sza} T vs oracle")


def test_sharded_equals_unsharded_bitwise(rtamd):
    """an N-way split of the spectral axis with GLOBAL ndoubl/iface reproduces the
    1-way result bit for bit (the multi-GPU correctness argument, exercised on one device)."""
    m = rtamd.scenes.make_scene(3, 9, 5, 30, seed=8)
    sc = rtamd.prepare_scene(m)
    R, T = rtamd.rt_run(m)[:2]
    parts = []
    for lo, hi in ((0, 11), (11, 19), (19, 30)):
        with rtamd.corert.make_handle(m, S=hi - lo) as h:
            parts.append(rtamd.corert.run_scene(h, sc.spectral_slice(lo, hi)))
    assert np.array_equal(np.concatenate([p[0] for p in parts], axis=2), R)
    assert np.array_equal(np.concatenate([p[1] for p in parts], axis=2), T)


@pytest.mark.parametrize("nS,lt,mode", [(1, 5, "lds"), (3, 9, "lds"), (4, 7, "generic")])
def test_hdrf_bhr_extras(rtamd, cref, nS, lt, mode):
    """The RAMI extras of rt_run's 7-tuple (rt_run.jl:226): hdr, bhr_uw[1,:], bhr_dw[1,:]
    (interaction_hdrf.jl:9-45, postprocessing_vza.jl:63-93)."""
    m = rtamd.scenes.make_scene(nS, lt, 4, 10, seed=31, albedo=0.35)
    p = cref.pack_scene(helpers.oracle_scene(m))
    Rr, Tr, Hr, upr, dwr, info = cref.rt_run_full(p)
    assert info == 0
    sc = rtamd.prepare_scene(m)
    with rtamd.corert.make_handle(m) as h:
        if mode == "generic":
            h.set_option(rtamd._lib.MOM_OPT_FORCE_GENERIC, 1)
        R, T = rtamd.corert.run_scene(h, sc)
        H, up, dw = h.get_hdr()
    helpers.assert_stokes_close(R, Rr, what="R")
    helpers.assert_stokes_close(H, Hr, what="hdr")
    np.testing.assert_allclose(up, upr, rtol=1e-10, atol=1e-300)
    np.testing.assert_allclose(dw, dwr, rtol=1e-10, atol=1e-300)
    np.testing.assert_allclose(up[0] / dw[0], 0.35, rtol=1e-12)  # Lambertian: BHR = albedo
    out = rtamd.rt_run(m)
    assert len(out) == 7 and not out[2].any() and not out[3].any()
    helpers.assert_stokes_close(out[4], H, rtol=1e-11, what="rt_run hdr")
    np.testing.assert_allclose(out[5], up[0], rtol=1e-11)
    np.testing.assert_allclose(out[6], dw[0], rtol=1e-11)


@pytest.mark.parametrize("nS,lt,generic", [(3, 9, False), (4, 9, False), (3, 33, False), (4, 7, True), (3, 53, False), (4, 53, False)])
def test_m0_reduction_matches_full_problem(rtamd, cref, nS, lt, generic):
    """Fourier moment 0 on the (I,Q) sub-problem (include/momcore.h, mom_scene_set) gives the outputs of
    the full nStokes problem: both against the oracle (which always solves the full problem) and against
    each other.  lt = 53: 30 streams, N = 90 / 120, sub-problem N0 = 60 with TWO Stokes components per stream on the 8-wave
    strip image -- the case whose persistent stream-pair tables do not fit the CU (ptab_reals must return 0)."""
    m = rtamd.scenes.make_scene(nS, lt, 5, 12, seed=17, vaz=(10.0, 95.0, 170.0))
    sc = rtamd.prepare_scene(m)
    Rr, Tr, Hr, upr, dwr, _ = cref.rt_run_full(cref.pack_scene(helpers.oracle_scene(m)))
    res = {}
    for red in (1, 0):
        with rtamd.corert.make_handle(m) as h:
            h.set_option(rtamd._lib.MOM_OPT_M0_REDUCTION, red)
            if generic:
                h.set_option(rtamd._lib.MOM_OPT_FORCE_GENERIC, 1)
            R, T = rtamd.corert.run_scene(h, sc)
            res[red] = (R, T) + h.get_hdr()
        helpers.assert_stokes_close(R, Rr, what=f"R red={red}")
        helpers.assert_stokes_close(T, Tr, what=f"T red={red}")
        helpers.assert_stokes_close(res[red][2], Hr, what=f"hdr red={red}")
        np.testing.assert_allclose(res[red][4], dwr, rtol=1e-10, atol=helpers.ATOL_STOKES)
    helpers.assert_stokes_close(res[1][0], res[0][0], rtol=1e-11, what="reduced vs full R")
    helpers.assert_stokes_close(res[1][1], res[0][1], rtol=1e-11, what="reduced vs full T")


def test_m0_reduction_refused_for_polarised_source(rtamd, cref):
    """A source with a U component couples the (U,V) block to the outputs for m = 0: the library must
    detect it (I0[2] != 0) and keep the full problem."""
    rt = rtamd.corert
    m = rtamd.scenes.make_scene(3, 7, 3, 6, seed=5, vaz=(30.0, 60.0, 140.0))
    m.params.polarization_type = rt.PolarizationType(3, (1.0, 1.0, -1.0), (1.0, 0.0, 0.2))
    sc_o = helpers.oracle_scene(m)
    sc_o.pol.I0 = np.array([1.0, 0.0, 0.2])
    Rr, Tr, _ = cref.rt_run(cref.pack_scene(sc_o))
    R, T = rtamd.rt_run(m)[:2]
    helpers.assert_stokes_close(R, Rr, what="R polarised source")
    assert np.abs(Rr[:, 2]).max() > 1e-4  # U really is fed by the source


def test_operator_level_state_rules(rtamd, cref):
    """mom_download of the composite layer after a scene-level run returns Fourier moment 0 without the internal row
    pitch (N = 24: pitch 32), or MOM_ESTATE when moment 0 ran on the (I,Q) sub-problem; the operator-level calls
    refuse to continue from scene-level state; the inverse option reaches the reduced problem's stream set."""
    L = rtamd._lib
    m = rtamd.scenes.make_scene(3, 9, 3, 6, seed=12)
    sc = rtamd.prepare_scene(m)
    N, S = sc.N, sc.S
    assert N == 24
    # operator-level replay of moment 0 -> composite R-+ and J0-
    qp, p = m.quad_points, m.params
    Lin = rtamd.corert.construct_layer_inputs(m)
    Zpp, Zmp = rtamd.corert.z_bases(m)
    with rtamd.corert.make_handle(m) as h:
        for z in range(sc.Nz):
            dtau, nd = rtamd.corert.get_dtau_ndoubl(Lin.τ[:, z], Lin.ϖ[:, z], qp.qp_μ)
            Zp = np.einsum("ks,kij->sij", Lin.zw[:, :, z], Zpp[0])
            Zm = np.einsum("ks,kij->sij", Lin.zw[:, :, z], Zmp[0])
            h.elemental(0, nd, Lin.τ_sum[:, z], dtau, Lin.ϖ[:, z], rtamd.corert._abi_mats(Zp), rtamd.corert._abi_mats(Zm), S)
            h.doubling(nd, np.exp(-dtau / qp.μ0))
            if z == 0:
                h.copy_added_to_composite()
            else:
                h.interaction(int(Lin.iface[z]))
        h.surface_lambertian(0, p.brdf_albedo, Lin.τ_sum[:, -1])
        h.interaction(int(Lin.iface[-1]), with_surface_layer=True)
        R_op, J_op = h.download(L.COMP["R_mp"]), h.download(L.COMP["J0m"])
    with rtamd.corert.make_handle(m) as h:
        rtamd.corert.run_scene(h, sc)   # default at N = 24: the wave-per-point kernel, no composite layer in HBM
        with pytest.raises(rtamd.MomError) as e:
            h.download(L.COMP["R_mp"])
        assert e.value.code == L.MOM_ESTATE and "registers" in str(e.value)
    with rtamd.corert.make_handle(m) as h:
        h.set_option(L.MOM_OPT_M0_REDUCTION, 0)
        h.set_option(L.MOM_OPT_SMALL_N, 0)
        rtamd.corert.run_scene(h, sc)
        helpers.assert_op_close(h.download(L.COMP["R_mp"]), R_op, rtol=1e-11, what="de-pitched R-+ after mom_rt_run")
        helpers.assert_op_close(h.download(L.COMP["J0m"]), J_op, rtol=1e-11, what="J0- after mom_rt_run")
        with pytest.raises(rtamd.MomError) as e:
            h.interaction(3)
        assert e.value.code == L.MOM_ESTATE
    with rtamd.corert.make_handle(m) as h:
        h.set_option(L.MOM_OPT_SMALL_N, 0)
        Rd, _ = rtamd.corert.run_scene(h, sc)  # workgroup kernels: m = 0 on the (I,Q) sub-problem
        with pytest.raises(rtamd.MomError) as e:
            h.download(L.COMP["R_mp"])
        assert e.value.code == L.MOM_ESTATE and "sub-problem" in str(e.value)
        # forcing the pivoted inverse AFTER mom_scene_set must reach the reduced problem too
        h.set_option(L.MOM_OPT_INVERSE, 1)
        h.rt_run()
        Rg, _ = h.get_RT()
    assert not np.array_equal(Rg, Rd)  # a different inverse: equal only to rounding
    helpers.assert_stokes_close(Rg, Rd, rtol=1e-11, what="Gauss-Jordan vs series after scene_set")


def test_rccl_allgather_single_rank(rtamd):
    """mom_comm_unique_id / mom_comm_init / mom_allgather_RT with a world of one rank: the RCCL path behind the C
    ABI executes on the GPU and returns this rank's spectra (the N > 1 layout is rank-major, tests/test_gpu_multirank.py)."""
    m = rtamd.scenes.make_scene(3, 9, 4, 20, seed=5)
    sc = rtamd.prepare_scene(m)
    with rtamd.corert.make_handle(m) as h:
        R, T = rtamd.corert.run_scene(h, sc)
        h.comm_init(0, 1, rtamd._lib.comm_unique_id())
        Rg, Tg = h.allgather_RT()
        # device-side optics with a communicator: the per-layer maxima go through ncclAllReduce(max) before ndoubl is derived
        R2, T2 = rtamd.corert.run_scene_device_optics(h, m)
        nd, iface = h.scene_get_layers(sc.Nz, sc.K, arrays=False)[:2]
        h.comm_destroy()
    assert np.array_equal(Rg, R) and np.array_equal(Tg, T)
    assert np.array_equal(nd, sc.ndoubl) and np.array_equal(iface, sc.iface) and np.array_equal(R2, R)


@pytest.mark.parametrize("nS,lt,aer", [(3, 9, 0.2), (1, 5, 0.0), (4, 7, 0.4), (3, 33, 0.2)])
def test_device_side_optics_bitwise_equals_host_route(rtamd, cref, nS, lt, aer):
    """mom_scene_set_optics assembles tau, varpi, Z weights, tau_sum, ndoubl and the interface codes on the
    GPU from tau_rayl / aerosol columns / the resident tau_abs table; every array and the resulting spectra are
    bitwise those of the host route (reference algebra in corert.construct_layer_inputs + mom_scene_set)."""
    m = rtamd.scenes.make_scene(nS, lt, 6, 40, seed=3 + nS, aerosol_total=aer)
    if aer == 0.0:
        m.τ_rayl[:, 1] = 0.0     # a non-scattering layer drives the interface state machine
    sc = rtamd.prepare_scene(m)
    with rtamd.corert.make_handle(m) as h:
        R0, T0 = rtamd.corert.run_scene(h, sc)
    with rtamd.corert.make_handle(m) as h:
        R1, T1 = rtamd.corert.run_scene_device_optics(h, m)
        nd, iface, tau, varpi, zw, tau_sum = h.scene_get_layers(sc.Nz, sc.K)
    assert np.array_equal(nd, sc.ndoubl) and np.array_equal(iface, sc.iface)
    assert np.array_equal(tau, sc.tau) and np.array_equal(varpi, sc.varpi)
    assert np.array_equal(zw, sc.zw) and np.array_equal(tau_sum, sc.tau_sum)
    assert np.array_equal(R1, R0) and np.array_equal(T1, T0)
    Rr, Tr, info = cref.rt_run(cref.pack_scene(helpers.oracle_scene(m)))
    helpers.assert_stokes_close(R1, Rr, what="device optics R vs oracle")


@pytest.mark.parametrize("nS,lt,aer", [(3, 9, 0.2), (1, 5, 0.0), (3, 33, 0.2)])
def test_device_side_optics_on_a_float32_handle(rtamd, nS, lt, aer):
    """mom_scene_set_optics on a dtype = 1 handle (r6): the layer optics are assembled on the device in Float64 exactly as for a
    Float64 handle and rounded to the scene's Float32 there -- the same Float32 inputs the host route uploads (its Float64
    arrays are bitwise the device's), so the spectra of the two Float32 routes are bitwise equal; tau_abs never visits the host."""
    m = rtamd.scenes.make_scene(nS, lt, 6, 40, seed=3 + nS, aerosol_total=aer)
    if aer == 0.0:
        m.τ_rayl[:, 1] = 0.0
    sc = rtamd.prepare_scene(m)
    with rtamd.corert.make_handle(m, float_type="Float32") as h:
        R0, T0 = rtamd.corert.run_scene(h, sc)
    with rtamd.corert.make_handle(m, float_type="Float32") as h:
        R1, T1 = rtamd.corert.run_scene_device_optics(h, m)
        nd, iface, tau, varpi, zw, tau_sum = h.scene_get_layers(sc.Nz, sc.K)
    assert np.array_equal(nd, sc.ndoubl) and np.array_equal(iface, sc.iface) and np.array_equal(tau, sc.tau)
    assert np.abs(R0).max() > 0 and np.array_equal(R1, R0) and np.array_equal(T1, T0)
    R64 = rtamd.rt_run(m)[0]
    assert np.isfinite(R1).all() and np.abs(R1 - R64).max() > 0    # a Float32 run of this scene, not the Float64 one (its accuracy
    #                                                                    against the Float32 oracle: test_float32_scene_level_path)


def test_voigt_to_spectrum_without_host_tau(rtamd):
    """Line list -> tau_abs (mom_voigt_tau_abs) -> layer optics (mom_scene_set_optics) -> spectrum, with tau_abs
    resident on the GPU throughout; equals the route that downloads tau_abs and feeds it through the host algebra."""
    ab = rtamd.absorption
    m = rtamd.scenes.make_scene(3, 9, 5, 600, seed=11, absorption=False)
    S, Nz = m.τ_rayl.shape
    grid = np.linspace(12903.0, 13245.0, S)
    lines = ab.synthetic_o2a_lines(200, seed=3)
    ph = rtamd.scenes.pressure_grid(Nz)
    p_full = 0.5 * (ph[1:] + ph[:-1]); T = np.linspace(215.0, 288.0, Nz); vcd = np.diff(ph) * 2.1e22
    with rtamd.corert.make_handle(m) as h:
        ab.compute_absorption_profile(h, lines, grid, p_full, T, vcd, 0.2095, model_vmr=0.2095)
        tau_abs = h.absorption_get()
        R1, T1 = rtamd.corert.run_scene_device_optics(h, m, upload_tau_abs=False)
    assert tau_abs.max() > 0.05
    m.τ_abs[:] = tau_abs
    R0, T0 = rtamd.rt_run(m)[:2]
    assert np.array_equal(R1, R0) and np.array_equal(T1, T0)
    assert np.ptp(R1[0, 0]) > 1e-3 * R1[0, 0].max()    # the lines are in the spectrum


def _surfaces(rt):
    return {"rpv": rt.rpvSurfaceScalar(0.1, 0.8, 0.7, -0.1), "rossli": rt.RossLiSurfaceScalar(0.1, 0.05, 0.2),
            "legendre": rt.LambertianSurfaceLegendre((0.2, 0.05, -0.02))}


@pytest.mark.parametrize("surf", ["rpv", "rossli", "legendre"])
@pytest.mark.parametrize("nS,lt,mode", [(3, 9, "lds"), (1, 5, "lds"), (4, 7, "generic"), (3, 33, "lds"), (3, 9, "nored")])
def test_surface_types(rtamd, cref, surf, nS, lt, mode):
    """non-Lambertian surfaces through mom_scene_set_surface -- create_surface_layer!(brdf::AbstractSurfaceType)
    (rpv_surface.jl:20-66: RPV and Ross-Li BRDF Fourier moments, a surface interaction for EVERY moment, hdr over all
    moments) and LambertianSurfaceLegendre (lambertian_surface.jl:77-138, incl. its j0+ = 0 and t = 0 for m > 0) --
    against the oracle's own surface code; N = 60 exercises the m = 0 (I,Q) sub-problem with the reduced BRDF matrix."""
    m = rtamd.scenes.make_scene(nS, lt, 4, 12, seed=41, vaz=(10.0, 95.0, 170.0))
    m.params.brdf = _surfaces(rtamd.corert)[surf]
    sc = rtamd.prepare_scene(m)
    Rr, Tr, Hr, upr, dwr, info = cref.rt_run_full(cref.pack_scene(helpers.oracle_scene(m)))
    assert info == 0
    with rtamd.corert.make_handle(m) as h:
        if mode == "generic":
            h.set_option(rtamd._lib.MOM_OPT_FORCE_GENERIC, 1)
        if mode == "nored":
            h.set_option(rtamd._lib.MOM_OPT_M0_REDUCTION, 0)
        R, T = rtamd.corert.run_scene(h, sc)
        H, up, dw = h.get_hdr()
    helpers.assert_stokes_close(R, Rr, what="R")
    helpers.assert_stokes_close(T, Tr, what="T")
    helpers.assert_stokes_close(H, Hr, what="hdr")
    np.testing.assert_allclose(up, upr, rtol=1e-10, atol=1e-300)
    np.testing.assert_allclose(dw, dwr, rtol=1e-10, atol=1e-300)
    # the surface matters: a Lambertian surface of similar brightness gives a different spectrum
    m.params.brdf = None
    R0, _ = rtamd.rt_run(m)[:2]
    assert np.abs(R0 - R).max() > 1e-6


F32_ERR_RATIO = 2.5      # GPU Float32 error / oracle Float32 error, both against the Float64 oracle (measured 0.9 ... 1.34)
F32_PAIR_ULPS = 80.0     # GPU Float32 vs oracle Float32, in units of eps32 2^nd (measured up to 51 on T at N = 6)


def f32_pair_bound(nd, pairs):
    """Bound on |GPU Float32 - other Float32 result| relative to the view's brightest Float64 intensity: 80 eps32 2^nd, and never
    looser than what the Float32 ORACLE's own error on THIS scene allows: both Float32 results lie within their
    errors of the Float64 oracle, the GPU's being <= F32_ERR_RATIO x the Float32 oracle's (asserted separately), so the two
    differ by at most (1 + F32_ERR_RATIO) x the Float32 oracle's error, measured here in the same continuum-relative norm.
    pairs: iterable of (Float32-oracle spectrum, Float64-oracle spectrum)."""
    o = 0.0
    for Xf, Xr in pairs:
        Imax = np.abs(Xr[:, 0:1, :]).max(axis=2, keepdims=True)
        o = max(o, float(np.max(np.abs(Xf.astype(np.float64) - Xr) / np.maximum(Imax, 1e-300))))
    return min(F32_PAIR_ULPS * 6e-8 * 2.0 ** nd, (1.0 + F32_ERR_RATIO) * o + 1e-5)


@pytest.mark.parametrize("nS,lt,surf,N,N0", [(3, 33, None, 60, 40), (3, 31, None, 57, 38), (3, 25, None, 48, 32), (4, 21, None, 56, 28),
                                             (3, 33, "rpv", 60, 40), (3, 31, "legendre", 57, 38), (3, 27, "rossli", 51, 34)])
def test_float32_m0_reduction_and_padding(rtamd, cref, nS, lt, surf, N, N0):
    """dtype = 1: the (I,Q) reduction of moment 0 (a nested Float32 sub-scene on the nStokes0 = 2 streams: strip image at
    N0 = 40 -> 44, general image at 34 / 38, wave-per-point kernel at 28 / 32) and the padding of the operator edge to the
    Float32 strip images (57 -> 60, 48 -> 52, 51 -> 52) against the same handle with MOM_OPT_M0_REDUCTION = 0 and
    MOM_OPT_STRIP_PAD = 0, against the Float32 oracle's error and the Float64 bound; all three surface kinds (the BRDF's
    reduced m = 0 matrix, the spectral albedo) incl. hdr / BHR."""
    m = rtamd.scenes.make_scene(nS, lt, 5, 24, seed=7 * nS + lt, aerosol_total=0.4, vaz=(10.0, 95.0, 170.0))
    if surf:
        m.params.brdf = _surfaces(rtamd.corert)[surf]
    sc = rtamd.prepare_scene(m)
    assert sc.N == N and 2 * (N // nS) == N0
    p = cref.pack_scene(helpers.oracle_scene(m))
    Rr, Tr, Hr, upr, dwr, info = cref.rt_run_full(p)
    Rf, Tf, info32 = cref.rt_run_f32(p)
    assert info == 0 and info32 == 0
    out, launches = {}, {}
    for red in (1, 0):
        with rtamd.corert.make_handle(m, float_type="Float32") as h:
            h.set_option(rtamd._lib.MOM_OPT_M0_REDUCTION, red)
            h.set_option(rtamd._lib.MOM_OPT_STRIP_PAD, red)
            R, T = rtamd.corert.run_scene(h, sc)
            out[red] = (R, T) + tuple(h.get_hdr())
            launches[red] = h.timers()["layer_launches"]
    assert launches[1] > launches[0]          # moment 0 ran as its own scene
    nd = int(sc.ndoubl.max())
    tol64 = 16 * 6e-8 / (1e-3 * float(m.quad_points.qp_μ.min()))
    pair = f32_pair_bound(nd, ((Rf, Rr), (Tf, Tr)))
    oR = float(np.max(np.abs(Rf - Rr) / np.maximum(np.abs(Rr[:, 0:1, :]), 1e-6 / tol64)))
    oT = float(np.max(np.abs(Tf - Tr) / np.maximum(np.abs(Tr[:, 0:1, :]), 1e-6 / tol64)))
    # the Float64 bound of test_float32_scene_level_path, widened to what the Float32 ORACLE itself loses on scenes whose
    # quadrature puts a view node next to a Gauss node (14 nodes: the Float32 oracle is 0.1 off in T there)
    rt = max(tol64, F32_ERR_RATIO * max(oR, oT))
    errs = {}
    for red in (1, 0):
        R, T, H, up, dw = out[red]
        errs[red] = (helpers.assert_stokes_close(R, Rr, rtol=rt, atol=1e-6, what=f"f32 R red={red}"),
                     helpers.assert_stokes_close(T, Tr, rtol=rt, atol=1e-6, what=f"f32 T red={red}"))
        helpers.assert_stokes_close(H, Hr, rtol=rt, atol=1e-6, what=f"f32 hdr red={red}")
        np.testing.assert_allclose(up[0], upr[0], rtol=rt)
        np.testing.assert_allclose(dw[0], dwr[0], rtol=rt)
    assert errs[1][0] <= F32_ERR_RATIO * oR + 2e-6 and errs[1][1] <= F32_ERR_RATIO * oT + 2e-6, (errs, oR, oT)
    for k, Xr in ((0, Rr), (1, Tr), (2, Hr)):
        Imax = np.abs(Xr[:, 0:1, :]).max(axis=2, keepdims=True)
        d = np.abs(out[1][k] - out[0][k]) / np.maximum(Imax, 1e-300)
        assert np.all(d <= pair), f"reduced vs full, output {k}: {d.max():.3e} > {pair:.3e}"
    if nS > 2:     # Stokes components >= 2 of the BHR get no m = 0 term at all
        assert np.all(out[1][3][2:] == 0) and np.all(out[1][4][2:] == 0)
    assert not np.array_equal(out[1][0], out[0][0])
    print(f"f32 m=0 reduction N={N} (N0={N0}): vs f64 oracle {errs[1][0]:.2e}/{errs[1][1]:.2e} (unreduced {errs[0][0]:.2e}/"
          f"{errs[0][1]:.2e}, f32 oracle {oR:.2e}/{oT:.2e}), launches {launches}")


@pytest.mark.parametrize("nS,lt,kw", [(3, 9, {}), (1, 5, {}), (4, 7, {}), (3, 33, {}), (4, 31, dict(generic=True)), (3, 9, dict(surf="rpv")),
                                      (1, 1, {}), (1, 1, dict(generic=True)), (4, 7, dict(generic=True)), (1, 3, {}), (1, 7, {}), (1, 9, {})])
def test_float32_scene_level_path(rtamd, cref, nS, lt, kw):
    """dtype = 1 (the reference's float_type = Float32): the same scene through the f32 builds -- the lane-per-point kernel
    (N = 4), the wave-per-point kernels (N = 6, 27, 28: v_mfma_f32_16x16x4_f32; N = 5, 7, 8: three / two points packed per wavefront) and the fused workgroup kernels (N = 60, forced
    generic cases) -- against the Float64 oracle.  Tolerance: the elemental layer has
    dtau <= 1e-3 min(mu) (rt_kernel.jl:241), so its transmission along the most vertical stream, t = exp(-dtau/mu_max)
    ~ 1 - 1e-3 min(mu)/max(mu), is stored in Float32 with an absolute error of eps32 = 6e-8: a RELATIVE error of
    eps32 / (1e-3 min(mu)) of that stream's optical depth, which the doublings carry into the layer transmission.
    rtol = 16 eps32 / (1e-3 min(mu)): 2e-2 for 5 Gauss nodes, 0.2 for the 17 of N = 60 (min(mu) = 0.0046); measured
    1e-3 ... 8e-3 and 2.5e-2.  Any Float32 run of the reference's algorithm carries this error."""
    kw = dict(kw)
    generic = kw.pop("generic", False)
    surf = kw.pop("surf", None)
    m = rtamd.scenes.make_scene(nS, lt, 4, 40, seed=23, aerosol_total=0.1)
    if surf:
        m.params.brdf = _surfaces(rtamd.corert)[surf]
    sc = rtamd.prepare_scene(m)
    Rr, Tr, Hr, upr, dwr, info = cref.rt_run_full(cref.pack_scene(helpers.oracle_scene(m)))
    with rtamd.corert.make_handle(m, float_type="Float32") as h:
        if generic:
            h.set_option(rtamd._lib.MOM_OPT_FORCE_GENERIC, 1)
        R, T = rtamd.corert.run_scene(h, sc)
        H, up, dw = h.get_hdr()
        assert h.timers()["layer_launches"] >= 1
    tol = 16 * 6e-8 / (1e-3 * float(m.quad_points.qp_μ.min()))
    eR = helpers.assert_stokes_close(R, Rr, rtol=tol, atol=1e-6, what="f32 R")
    eT = helpers.assert_stokes_close(T, Tr, rtol=tol, atol=1e-6, what="f32 T")
    helpers.assert_stokes_close(H, Hr, rtol=tol, atol=1e-6, what="f32 hdr")
    np.testing.assert_allclose(up[0], upr[0], rtol=tol)
    # The bound above is what ANY Float32 run owes the Float64 result; it cannot tell a bug from rounding.  The checker
    # that can is the Float32 build of the oracle (oracle/momref_f32.c: the same operations rounded to Float32):
    #   (a) the GPU's distance from the Float64 result is at most F32_ERR_RATIO x the Float32 oracle's own distance
    #       (+ the 1e-6 floor) -- the arbiter construction of test_gpu_precision.py, one precision down;
    #   (b) GPU Float32 against oracle Float32 directly, within F32_PAIR_ULPS eps32 2^nd (each doubling squares the direct
    #       transmission, so two Float32 runs that round differently once are 2^nd eps32 apart at the end).  Measured: the
    #       two Float32 runs are as far from each other as each is from the Float64 result (x 1.1 ... 1.4: independent
    #       rounding), i.e. 0.1 ... 0.8 of this bound -- it is (a) that separates "rounds differently" from "wrong".
    Rf, Tf, info32 = cref.rt_run_f32(cref.pack_scene(helpers.oracle_scene(m)))
    assert info32 == 0
    nd = int(sc.ndoubl.max())
    oR = float(np.max(np.abs(Rf - Rr) / np.maximum(np.abs(Rr[:, 0:1, :]), 1e-6 / tol)))
    oT = float(np.max(np.abs(Tf - Tr) / np.maximum(np.abs(Tr[:, 0:1, :]), 1e-6 / tol)))
    pair = f32_pair_bound(nd, ((Rf, Rr), (Tf, Tr)))
    def pair_err(X, Xf, Xref, what):   # |GPU f32 - oracle f32| relative to the view's brightest FLOAT64 intensity of the
        # spectrum (continuum-relative: a dim point's own relative error grows with its optical depth; the dim points are
        # held by (a) and by the bound against the Float64 oracle above)
        Imax = np.abs(Xref[:, 0:1, :]).max(axis=2, keepdims=True)
        d = np.abs(X - Xf.astype(np.float64)) / np.maximum(Imax, 1e-300)
        assert np.all(d <= pair), f"{what}: {d.max():.3e} > {pair:.3e}"
        return float(d.max())
    pR = pair_err(R, Rf, Rr, "f32 R vs f32 oracle")
    pT = pair_err(T, Tf, Tr, "f32 T vs f32 oracle")
    print(f"float32 vs f64 oracle: GPU {eR:.2e} / {eT:.2e}, f32 oracle {oR:.2e} / {oT:.2e}; GPU vs f32 oracle {pR:.2e} / {pT:.2e} "
          f"(bound {pair:.2e}; old bound {tol:.2e}), nd max {nd}")
    assert eR <= F32_ERR_RATIO * oR + 2e-6 and eT <= F32_ERR_RATIO * oT + 2e-6
    assert not np.array_equal(R, Rr)  # it really is a different precision


@pytest.mark.parametrize("nS,lt,N", [(4, 11, 36), (4, 13, 40), (4, 15, 44), (4, 19, 52), (4, 21, 56), (4, 23, 60), (3, 33, 60)])
def test_float32_strip_chains(rtamd, cref, nS, lt, N):
    """dtype = 1 at the edges that have a strip-chained image (N = 36, 40 [4-wave build only], 44, 52, 56, 60; both workgroup
    shapes: MOM_OPT_SMALL_WG = 1 -> 4 waves, two workgroups per CU; 0 -> 8 waves): the Float32 build of mom_strip.hpp's chains
    (accumulator layout row = 4 lq + r: the B operand of k-step (rt, r) supplies k = 16 rt + 4 lq + r and the A fragment follows;
    all 4 NT k-steps run against zero padding rows) against the same image with the chains switched off (MOM_OPT_INVERSE = 2),
    against the Float32 oracle (error ratio) and against the Float64 oracle (the bound every Float32 run owes it)."""
    m = rtamd.scenes.make_scene(nS, lt, 5, 24, seed=7 * nS + lt, aerosol_total=0.4)
    sc = rtamd.prepare_scene(m)
    assert sc.N == N
    p = cref.pack_scene(helpers.oracle_scene(m))
    Rr, Tr, info = cref.rt_run(p)
    Rf, Tf, info32 = cref.rt_run_f32(p)
    assert info == 0 and info32 == 0
    out = {}
    for inv in (0, 2, 8):      # 8: the chains of the 8-wave image
        with rtamd.corert.make_handle(m, float_type="Float32") as h:
            h.set_option(rtamd._lib.MOM_OPT_M0_REDUCTION, 0)     # the image under test runs every moment
            h.set_option(rtamd._lib.MOM_OPT_INVERSE, inv & 2)
            h.set_option(rtamd._lib.MOM_OPT_SMALL_WG, 0 if inv == 8 else 1)
            out[inv] = rtamd.corert.run_scene(h, sc)
    nd = int(sc.ndoubl.max())
    tol64 = 16 * 6e-8 / (1e-3 * float(m.quad_points.qp_μ.min()))
    pair = f32_pair_bound(nd, ((Rf, Rr), (Tf, Tr)))
    for X, Y, Xr in ((out[0][0], out[8][0], Rr), (out[0][1], out[8][1], Tr)):
        Imax = np.abs(Xr[:, 0:1, :]).max(axis=2, keepdims=True)
        assert np.all(np.abs(X - Y) / Imax <= pair), "4-wave vs 8-wave image"
    errs = {}
    for inv in (0, 2, 8):
        R, T = out[inv]
        errs[inv] = (helpers.assert_stokes_close(R, Rr, rtol=tol64, atol=1e-6, what=f"f32 R inv={inv}"),
                     helpers.assert_stokes_close(T, Tr, rtol=tol64, atol=1e-6, what=f"f32 T inv={inv}"))
    oR = float(np.max(np.abs(Rf - Rr) / np.maximum(np.abs(Rr[:, 0:1, :]), 1e-6 / tol64)))
    oT = float(np.max(np.abs(Tf - Tr) / np.maximum(np.abs(Tr[:, 0:1, :]), 1e-6 / tol64)))
    assert errs[0][0] <= F32_ERR_RATIO * oR + 2e-6 and errs[0][1] <= F32_ERR_RATIO * oT + 2e-6, (errs, oR, oT)
    assert errs[8][0] <= 1.3 * errs[2][0] + 2e-6 and errs[8][1] <= 1.3 * errs[2][1] + 2e-6, errs
    # the chains must be as accurate as the general path of the same image, not merely inside the Float32 band: r4 shipped
    # them for an hour with stale elemental tables in the padding rows of P (the Float32 chains run every k-step of the last
    # row tile), which cost a thick layer 3 x the accuracy and still passed the pair bound below
    assert errs[0][0] <= 1.3 * errs[2][0] + 2e-6 and errs[0][1] <= 1.3 * errs[2][1] + 2e-6, errs
    for X, Y, Xr in ((out[0][0], out[2][0], Rr), (out[0][1], out[2][1], Tr)):
        Imax = np.abs(Xr[:, 0:1, :]).max(axis=2, keepdims=True)
        d = np.abs(X - Y) / Imax
        assert np.all(d <= pair), f"strips vs general path: {d.max():.3e} > {pair:.3e}"
    assert not np.array_equal(out[0][0], out[2][0])      # the chains did run
    print(f"f32 strips N={N}: vs f64 oracle {errs[0][0]:.2e}/{errs[0][1]:.2e} (general path {errs[2][0]:.2e}/{errs[2][1]:.2e}, "
          f"f32 oracle {oR:.2e}/{oT:.2e}), nd {nd}")


@pytest.mark.parametrize("nS,lt,Nz,kw", [(3, 9, 6, {}), (1, 5, 5, {}), (4, 7, 5, dict(generic=True)), (3, 33, 6, {}),
                                          (3, 27, 5, dict(brdf="rpv")), (3, 9, 5, dict(brdf="legendre")),
                                          (4, 31, 4, {}), (3, 9, 5, dict(zero=(0, 1)))])
def test_multisensor_sweep(rtamd, cref, nS, lt, Nz, kw):
    """rt_run_test_ms (rt_run_multisensor.jl) -- sensors inside the atmosphere through
    mom_rt_run_multisensor: the slabs above and below each sensor by the fused layer kernels (N = 60 strip chains, N = 51
    padded to 52, N = 76 generic), k_interlayer (interlayer_flux.jl:7-24) and the azimuthal weighting, against the
    oracle's rt_kernel_multisensor restatement; includes the level-0 (TOA/BOA) sensor, the lowest interface, BRDF and
    Legendre surfaces and non-scattering top layers (interface codes 00/01)."""
    m = rtamd.scenes.make_scene(nS, lt, Nz, 10, seed=17 + lt, aerosol_total=0.0 if "zero" in kw else 0.3,
                                vaz=(10.0, 95.0, 170.0))
    if kw.get("brdf"):
        m.params.brdf = _surfaces(rtamd.corert)[kw["brdf"]]
    for z in kw.get("zero", ()):
        m.τ_rayl[:, z] = 0.0
    sc = rtamd.prepare_scene(m)
    levels = [Nz - 1, 0, 1, Nz // 2, 1]   # unsorted, with a repeat: the library orders the top slabs itself
    uwr, dwr, info = cref.rt_run_multisensor(cref.pack_scene(helpers.oracle_scene(m)), levels)
    assert info == 0
    with rtamd.corert.make_handle(m) as h:
        if kw.get("generic"):
            h.set_option(rtamd._lib.MOM_OPT_FORCE_GENERIC, 1)
        rtamd.corert.scene_set(h, sc)
        uw, dw = h.rt_run_multisensor(levels)
        R, T = rtamd.corert.run_scene(h, sc)       # the handle still runs the plain column afterwards
    tol = helpers.stokes_rtol(sc.ndoubl)
    for ims in range(len(levels)):
        helpers.assert_stokes_close(uw[ims], uwr[ims], rtol=tol, what=f"uwJ level {levels[ims]}")
        helpers.assert_stokes_close(dw[ims], dwr[ims], rtol=tol, what=f"dwJ level {levels[ims]}")
    helpers.assert_stokes_close(uw[1], R, rtol=tol, what="level 0 vs mom_rt_run R")
    helpers.assert_stokes_close(dw[1], T, rtol=tol, what="level 0 vs mom_rt_run T")
    assert np.array_equal(uw[2], uw[4]) and np.array_equal(dw[2], dw[4])
    out = rtamd.rt_run_test_ms(levels, m)
    assert len(out) == 4 and len(out[0]) == len(levels) and np.array_equal(out[0][2], uw[2])
    with pytest.raises(rtamd.MomError):
        with rtamd.corert.make_handle(m) as h:
            rtamd.corert.scene_set(h, sc)
            h.rt_run_multisensor([Nz])


def test_multisensor_more_sensors_than_one_target_table(rtamd, cref):
    """Eleven sensors: more than one kernel's target table holds (kMaxTargets = 20: the running top slab + a snapshot and a
    segment per sensor, 9 sensors), so the library sweeps twice; every layer's added operators are built once per sweep and
    feed all composites (rt_kernel_multisensor.jl:51-112)."""
    Nz = 7
    m = rtamd.scenes.make_scene(3, 9, Nz, 9, seed=5, aerosol_total=0.25, vaz=(0.0, 60.0, 140.0))
    sc = rtamd.prepare_scene(m)
    levels = [3, 0, 6, 1, 5, 2, 4, 3, 6, 1, 2]
    uwr, dwr, info = cref.rt_run_multisensor(cref.pack_scene(helpers.oracle_scene(m)), levels)
    assert info == 0
    with rtamd.corert.make_handle(m) as h:
        rtamd.corert.scene_set(h, sc)
        uw, dw = h.rt_run_multisensor(levels)
        uw1, dw1 = h.rt_run_multisensor([4])
    tol = helpers.stokes_rtol(sc.ndoubl)
    for ims in range(len(levels)):
        helpers.assert_stokes_close(uw[ims], uwr[ims], rtol=tol, what=f"uwJ level {levels[ims]}")
        helpers.assert_stokes_close(dw[ims], dwr[ims], rtol=tol, what=f"dwJ level {levels[ims]}")
    # a sensor's result depends on the others only through rounding: the slab below it is its segment joined to the slab
    # below the next sensor (k_combine), not a layer-by-layer sum of its own
    helpers.assert_stokes_close(uw1[0], uw[6], rtol=1e-12, what="sensor alone vs among others (uwJ)")
    helpers.assert_stokes_close(dw1[0], dw[6], rtol=1e-12, what="sensor alone vs among others (dwJ)")
