"""GPU parity of the rotational-Raman path (BASELINE config 5) through the C ABI against oracle/rrsref.py, operator by
operator and scene-level, in both positions of rrs_strict_reference."""
import copy

import numpy as np
import pytest

import helpers
from oracle import momref as mr
from oracle import rrsref as rr

pytestmark = pytest.mark.gpu

A_EL = ("r_pm", "r_mp", "t_mm", "t_pp", "j0p", "j0m")            # which 0..5
C_EL = ("R_mp", "R_pm", "T_pp", "T_mm", "J0p", "J0m")            # which 6..11
A_IE = ("ier_pm", "ier_mp", "iet_mm", "iet_pp", "ieJ0p", "ieJ0m")  # which 18..23
C_IE = ("ieR_mp", "ieR_pm", "ieT_pp", "ieT_mm", "ieJ0p", "ieJ0m")  # which 24..29


def abi(x):
    """oracle layout -> flat ABI (Julia column-major [i,j,n(,dn)] / [i,n(,dn)])."""
    x = np.asarray(x)
    if x.ndim in (3, 4) and x.shape[-1] == x.shape[-2]:
        return np.ascontiguousarray(np.swapaxes(x, -1, -2)).reshape(-1)
    return np.ascontiguousarray(x).reshape(-1)


def from_abi(buf, like):
    like = np.asarray(like)
    x = np.asarray(buf).reshape(like.shape)
    if like.ndim in (3, 4) and like.shape[-1] == like.shape[-2]:
        return np.swapaxes(x, -1, -2).copy()
    return x.copy()


def push(h, added=None, comp=None, surf=None):
    if added is not None:
        for k, nm in enumerate(A_EL):
            h.rrs_upload(k, abi(getattr(added, nm)))
        for k, nm in enumerate(A_IE):
            h.rrs_upload(18 + k, abi(getattr(added, nm)))
    if comp is not None:
        for k, nm in enumerate(C_EL):
            h.rrs_upload(6 + k, abi(getattr(comp, nm)))
        for k, nm in enumerate(C_IE):
            h.rrs_upload(24 + k, abi(getattr(comp, nm)))
    if surf is not None:
        for k, nm in enumerate(A_EL):
            h.rrs_upload(12 + k, abi(getattr(surf, nm)))


def check_added(h, ref, what, rtol=1e-11):
    for k, nm in enumerate(A_EL):
        helpers.assert_op_close(from_abi(h.rrs_download(k), getattr(ref, nm)), getattr(ref, nm), rtol, f"{what} {nm}")
    for k, nm in enumerate(A_IE):
        helpers.assert_op_close(from_abi(h.rrs_download(18 + k), getattr(ref, nm)), getattr(ref, nm), rtol, f"{what} {nm}")


def check_comp(h, ref, what, rtol=1e-11):
    for k, nm in enumerate(C_EL):
        helpers.assert_op_close(from_abi(h.rrs_download(6 + k), getattr(ref, nm)), getattr(ref, nm), rtol, f"{what} {nm}")
    for k, nm in enumerate(C_IE):
        helpers.assert_op_close(from_abi(h.rrs_download(24 + k), getattr(ref, nm)), getattr(ref, nm), rtol, f"{what} {nm}")


def random_layers(N, S, nR, rng):
    scale = 0.4 / N   # row sums of t stay below 1: the doubling recursion does not amplify rounding differences
    a = rr.make_added_layer_rs(N, S, nR)
    c = rr.make_composite_layer_rs(N, S, nR)
    for o, mats, vecs, ie4, ie3 in ((a, ("r_pm", "r_mp", "t_mm", "t_pp"), ("j0p", "j0m"), A_IE[:4], A_IE[4:]),
                                    (c, ("R_mp", "R_pm", "T_pp", "T_mm"), ("J0p", "J0m"), C_IE[:4], C_IE[4:])):
        for nm in mats:
            getattr(o, nm)[:] = scale * rng.random((S, N, N))
        for nm in vecs:
            getattr(o, nm)[:] = rng.random((S, N))
        for nm in ie4:
            getattr(o, nm)[:] = rng.standard_normal((nR, S, N, N))
        for nm in ie3:
            getattr(o, nm)[:] = rng.standard_normal((nR, S, N))
    for o, ns in ((a, ("t_mm", "t_pp")), (c, ("T_mm", "T_pp"))):
        for nm in ns:
            getattr(o, nm)[:] += 0.85 * np.eye(N)
    return a, c


VIEWS = {1: dict(vza=(30.0,), vaz=(20.0,)), 3: {}}   # one view: lt = 5 gives the reference's RRS shape, 5 streams (N = 15 for IQU)


def model_and_handle(rtamd, nS, lt, S, Nz=2, seed=1, nv=3):
    m = rtamd.scenes.make_scene(nS, lt, Nz, S, seed=seed, aerosol_total=0.15, **VIEWS[nv])
    return m, rtamd.corert.make_handle(m)


OFFS = [-3, 1, 0, 5]


@pytest.mark.parametrize("nS,lt,nv", [(1, 7, 3), (3, 5, 1), (4, 5, 1), (3, 11, 3), (1, 43, 3), (4, 9, 3), (3, 21, 3), (4, 21, 3), (3, 33, 3)])   # ... N = 42, 56, 60: 3 x 3 / 4 x 4 tiles
@pytest.mark.parametrize("strict", [True, False])
@pytest.mark.parametrize("nd", [1, 3])
def test_doubling_inelastic(rtamd, nS, lt, nv, strict, nd):
    """doubling_helper!(::RRS) (doubling_inelastic.jl:13-134) on uploaded layers (including the stale iet-- the strict
    position reads, D5) against the restatement: all twelve arrays of the added layer and expk."""
    S = 9
    m, h = model_and_handle(rtamd, nS, lt, S, nv=nv)
    with h:
        N = h.N
        rng = np.random.default_rng(100 * nS + lt + nd)
        a, _ = random_layers(N, S, len(OFFS), rng)
        rrs = rr.RRSInputs(np.array(OFFS), np.ones(len(OFFS)), None, rrs_strict_reference=strict)
        h.rrs_set(OFFS, np.ones(len(OFFS)), strict)
        push(h, added=a)
        expk = 0.9 + 0.1 * rng.random(S)
        ref = copy.deepcopy(a)
        ek = expk.copy()
        pol = mr.pol_from_n(nS)
        rr.doubling_inelastic(pol, rrs, ek, nd, ref, m.params.strict_reference_indexing)
        got = h.rrs_doubling(nd, expk)
        np.testing.assert_allclose(got, ek, rtol=1e-13, atol=1e-300)
        check_added(h, ref, f"doubling nS={nS} N={N} strict={strict} nd={nd}")


@pytest.mark.parametrize("nS,lt,nv", [(1, 7, 3), (3, 5, 1), (4, 5, 1), (3, 11, 3), (3, 21, 3), (4, 21, 3)])
@pytest.mark.parametrize("iface", [3, 0, 1, 2])
@pytest.mark.parametrize("surface", [False, True])
def test_interaction_inelastic(rtamd, nS, lt, nv, iface, surface):
    """interaction_helper!(::RRS, iface) (interaction_inelastic.jl:8-340): the production case 11 in both switch positions,
    00/01/10 in the corrected one (the strict one must report that the reference raises)."""
    S = 8
    m, h = model_and_handle(rtamd, nS, lt, S, nv=nv)
    with h:
        N = h.N
        rng = np.random.default_rng(7 * nS + lt + iface)
        a, c = random_layers(N, S, len(OFFS), rng)
        if surface:  # the surface layer: its ie* arrays are zeros (never written by the reference)
            for nm in A_IE:
                getattr(a, nm)[:] = 0.0
        for strict in (True, False):
            h.rrs_set(OFFS, np.ones(len(OFFS)), strict)
            push(h, added=None if surface else a, comp=c, surf=a if surface else None)
            rrs = rr.RRSInputs(np.array(OFFS), np.ones(len(OFFS)), None, rrs_strict_reference=strict)
            if iface != 3 and strict:
                with pytest.raises(rtamd._lib.MomError) as ei:
                    h.rrs_interaction(iface, surface)
                assert ei.value.code == rtamd._lib.MOM_EUNSUPPORTED
                with pytest.raises(rr.ReferenceRaises):
                    rr.interaction_inelastic(rrs, iface, copy.deepcopy(c), a)
                continue
            ref = copy.deepcopy(c)
            rr.interaction_inelastic(rrs, iface, ref, copy.deepcopy(a))
            h.rrs_interaction(iface, surface)
            check_comp(h, ref, f"interaction nS={nS} N={N} iface={iface} surf={surface} strict={strict}")


@pytest.mark.parametrize("nS,lt", [(1, 7), (3, 5), (4, 7), (3, 21), (4, 21)])
@pytest.mark.parametrize("nd", [0, 2])
@pytest.mark.parametrize("mm", [0, 1])
def test_elemental_stateful(rtamd, nS, lt, nd, mm):
    """rt_kernel!(::RRS) rt_kernel.jl:290-304: elemental_inelastic! + elemental! on the persistent added layer (entries whose
    source index is off the grid keep their previous sources; D is applied to all of ieJ0-)."""
    S = 10
    m, h = model_and_handle(rtamd, nS, lt, S)
    scene = helpers.oracle_scene(m)
    with h:
        N = h.N
        rng = np.random.default_rng(nS + lt + nd)
        a, _ = random_layers(N, S, len(OFFS), rng)
        vp = np.array([0.02, 0.03, 0.01, 0.04])
        h.rrs_set(OFFS, vp, True)
        push(h, added=a)
        rrs = rr.RRSInputs(np.array(OFFS), vp, mr.get_greek_rayleigh(0.2), True)
        layers = mr.construct_core_optical_properties(scene, mm)
        _, tau_sum = mr.extract_effective_props(layers)
        lay = layers[1]
        dtau = lay.tau / 2 ** nd
        fsc = rr.fscatt_rayleigh(scene)[:, 1]
        Zr_pp, Zr_mp = mr.compute_Z_moments(nS, scene.quad.qp_mu, rrs.greek_raman, mm)
        Zpp, Zmp = (x[0] for x in (lay.Zpp_basis, lay.Zmp_basis))   # the Rayleigh basis as THE phase matrix of this call
        ref = copy.deepcopy(a)
        rr.elemental_inelastic(scene.pol, scene.quad, rrs, fsc, tau_sum[:, 1], dtau, lay.varpi, Zr_pp, Zr_mp, mm, nd, ref,
                               scene.strict_reference_indexing)
        mr.elemental(scene.pol, scene.quad, tau_sum[:, 1], dtau, lay.varpi, Zpp, Zmp, mm, nd, ref,
                     scene.strict_reference_indexing)
        h.rrs_elemental(mm, nd, tau_sum[:, 1], dtau, lay.varpi, abi(Zpp[None]), abi(Zmp[None]), fsc, abi(Zr_pp[None]),
                        abi(Zr_mp[None]))
        check_added(h, ref, f"elemental nS={nS} nd={nd} m={mm}", rtol=1e-12)


def _rrs_inputs(rtamd, offsets, strict, amp=0.02, cab=0.96):
    nR = len(offsets)
    vp = amp * (1.0 + 0.1 * np.arange(nR))
    g = rtamd.corert.get_greek_rayleigh(0.2)
    RS = rtamd.corert.RRS(greek_raman=g, ϖ_Cabannes=cab, ϖ_λ1λ0=vp, i_λ1λ0=np.asarray(offsets), rrs_strict_reference=strict)
    ora = rr.RRSInputs(np.asarray(offsets, dtype=np.int64), vp, mr.get_greek_rayleigh(0.2), rrs_strict_reference=strict)
    return RS, ora


@pytest.mark.parametrize("nS,lt,S,Nz,nv", [(1, 3, 24, 3, 3), (3, 5, 20, 3, 1), (4, 5, 16, 2, 1), (3, 11, 12, 2, 3), (3, 5, 40, 5, 1),
                                            (3, 21, 10, 2, 3), (4, 21, 8, 2, 3)])   # the last two: N = 42, 56
@pytest.mark.parametrize("strict", [True, False])
def test_rt_run_rrs_parity(rtamd, nS, lt, S, Nz, nv, strict):
    """rt_run(RS_type::RRS, model, iBand) for seeded scenes (Rayleigh + one aerosol type + gas absorption, Lambertian surface):
    R_SFI, T_SFI, ieR_SFI, ieT_SFI against the restatement, both switch positions.  (3, 5, .., 1 view) is the reference's own
    RRS shape N = 15 (O2Parameters.yaml: IQU, l_trunc 5, one view: 3 Gauss nodes + Sun + view)."""
    m = rtamd.scenes.make_scene(nS, lt, Nz, S, seed=nS + lt + S, aerosol_total=0.1, **VIEWS[nv])
    offs = [-4, -1, 2, 7, 3]
    RS, ora = _rrs_inputs(rtamd, offs, strict)
    R, T, ieR, ieT, hdr, up, dw = rtamd.corert.rt_run_rrs(RS, m)
    scene = helpers.oracle_scene(m)
    scene.varpi_cabannes = RS.ϖ_Cabannes
    Rr, Tr, ieRr, ieTr, hdrr, upr, dwr = rr.rt_run_rrs(scene, ora, full=True)
    _check_rrs_outputs((R, T, ieR, ieT, hdr, up, dw), (Rr, Tr, ieRr, ieTr, hdrr, upr[0], dwr[0]))


def _check_rrs_outputs(got, ref):
    R, T, ieR, ieT, hdr, up, dw = got
    Rr, Tr, ieRr, ieTr, hdrr, upr, dwr = ref
    helpers.assert_stokes_close(R, Rr, what="R")
    helpers.assert_stokes_close(T, Tr, what="T")
    helpers.assert_stokes_close(hdr, hdrr, what="hdr")
    np.testing.assert_allclose(up, upr, rtol=1e-10, atol=helpers.ATOL_STOKES)
    np.testing.assert_allclose(dw, dwr, rtol=1e-10, atol=helpers.ATOL_STOKES)
    assert np.abs(ieRr).max() > 0
    # the inelastic spectra are judged like the elastic ones: relative to the elastic intensity of the same view and point
    scale = np.abs(Rr[:, 0:1, :])
    assert np.all(np.abs(ieR - ieRr) <= 1e-10 * scale + 1e-14), np.abs(ieR - ieRr).max()
    scale = np.abs(Tr[:, 0:1, :])
    assert np.all(np.abs(ieT - ieTr) <= 1e-10 * scale + 1e-14), np.abs(ieT - ieTr).max()
    # ... and tightly against their own size
    helpers.assert_op_close(ieR, ieRr, 1e-9, "ieR_SFI")
    helpers.assert_op_close(ieT, ieTr, 1e-9, "ieT_SFI")


@pytest.mark.parametrize("surf", ["rpv", "rossli", "legendre"])
@pytest.mark.parametrize("nS", [1, 3])
def test_rt_run_rrs_surfaces(rtamd, surf, nS):
    """The RRS run over the other surface types of create_surface_layer! (rpv_surface.jl:20-66, lambertian_surface.jl:77-138):
    the surface layer is elastic (its ie* arrays stay zero), so only its r-+, t, j0+- change; incl. hdr / bhr."""
    rt = rtamd.corert
    m = rtamd.scenes.make_scene(nS, 5, 3, 14, seed=40 + nS, aerosol_total=0.1, **VIEWS[1])
    m.params.brdf = {"rpv": rt.rpvSurfaceScalar(0.12, 0.7, -0.15, 0.9), "rossli": rt.RossLiSurfaceScalar(0.05, 0.02, 0.1),
                     "legendre": rt.LambertianSurfaceLegendre([0.2, 0.05, -0.02])}[surf]
    RS, ora = _rrs_inputs(rtamd, [-3, 2, 6], strict=False)
    got = rt.rt_run_rrs(RS, m)
    scene = helpers.oracle_scene(m)
    scene.varpi_cabannes = RS.ϖ_Cabannes
    ref = rr.rt_run_rrs(scene, ora, full=True)
    _check_rrs_outputs(got, ref[:5] + (ref[5][0], ref[6][0]))


@pytest.mark.parametrize("strict", [True, False])
@pytest.mark.parametrize("world,nS,lt,nv", [(2, 3, 5, 1), (3, 3, 5, 1), (5, 3, 5, 1), (3, 3, 11, 3), (2, 3, 21, 3), (3, 4, 21, 3)])
def test_rrs_windows_reassemble_the_full_run(rtamd, strict, world, nS, lt, nv):
    """Spectral sharding of the RRS path: every rank runs its window = owned slice + halo of max |i_λ₁λ₀|
    (mom_rrs_set_shard) with the global ndoubl; the owned slices, put side by side, are the unsharded run BIT FOR BIT
    (same kernels, same per-point arithmetic, no exchange).  world = 5: slices of 8 points, halo 7.  N = 15 (one wave per pair),
    and -- r5 -- N = 27, 42, 56: the workgroup-per-pair kernels with an owned range that is not the whole axis."""
    rt = rtamd.corert
    m = rtamd.scenes.make_scene(nS, lt, 3, 40, seed=77, aerosol_total=0.1, **VIEWS[nv])
    RS, _ = _rrs_inputs(rtamd, [-4, -1, 2, 7, 3], strict)
    full = rt.rt_run_rrs(RS, m)
    assert np.abs(full[2]).max() > 0
    parts = []
    for rank in range(world):
        lo, hi, wlo, whi = rtamd.sharding.rrs_window(40, world, rank, RS.i_λ1λ0)
        assert wlo == max(0, lo - 7) and whi == min(40, hi + 7)
        parts.append(rt.rt_run_rrs_window(RS, m, lo, hi, (wlo, whi)))
    for k, name in enumerate(("R", "T", "ieR", "ieT", "hdr", "bhr_uw", "bhr_dw")):
        got = np.concatenate([p[k] for p in parts], axis=-1)
        assert np.array_equal(got, full[k]), (name, np.abs(got - full[k]).max())


def test_rrs_shard_halo_too_short_is_refused(rtamd):
    rt = rtamd.corert
    m = rtamd.scenes.make_scene(1, 3, 2, 30, seed=5)
    RS, _ = _rrs_inputs(rtamd, [-6, 3], False)
    with pytest.raises(rtamd._lib.MomError, match="halo"):
        rt.rt_run_rrs_window(RS, m, 10, 20, (7, 26))      # 3 points below, needs 6
    with pytest.raises(rtamd._lib.MomError, match="halo"):
        rt.rt_run_rrs_window(RS, m, 10, 20, (4, 22))      # 2 points above
    got = rt.rt_run_rrs_window(RS, m, 0, 10, (0, 16))      # window starting at the global edge: no lower halo needed
    assert got[0].shape[-1] == 10


@pytest.mark.parametrize("strict", [True, False])
def test_rt_run_rrs_twice_on_one_handle(rtamd, strict):
    """rt_run allocates zeroed layers on every call (rt_run.jl:108-116); the handle's persistent layers must behave the same:
    a second mom_rt_run_rrs on the same handle, and a run after operator-level uploads of arbitrary layer contents, reproduce
    the first run bit for bit.  The strict position is the sensitive one (D5 reads iet-- of the previous layer, k_strict_D and
    the off-grid ieJ0- are read-modify-written); the corrected position's pointer hand-over is covered too."""
    rt = rtamd.corert
    m = rtamd.scenes.make_scene(3, 5, 3, 24, seed=9, aerosol_total=0.1, **VIEWS[1])
    offs = [-4, -1, 2, 7, 3]
    RS, _ = _rrs_inputs(rtamd, offs, strict)
    model = rt._with_cabannes(RS, m)
    sc = rtamd.prepare_scene(model)
    Zr_pp, Zr_mp = rt.raman_z(RS, model)
    fs = rt.fscatt_rayleigh(model)
    fresh = rt.rt_run_rrs(RS, m)

    def spectra(h):
        h.rt_run_rrs()
        return h.get_RT_rrs()[:4] + h.get_hdr_rrs()

    with rt.make_handle(model) as h:
        h.set_option(rtamd._lib.MOM_OPT_STRIP_PAD, 0)
        h.rrs_set(RS.i_λ1λ0, RS.ϖ_λ1λ0, strict)
        rt.scene_set(h, sc)
        h.scene_set_rrs(np.ascontiguousarray(fs.T), rt._abi_mats(Zr_pp), rt._abi_mats(Zr_mp))
        first = spectra(h)
        second = spectra(h)
        # arbitrary layer contents through the operator-level API, then a third run
        rng = np.random.default_rng(3)
        a, c = random_layers(h.N, 24, len(offs), rng)
        push(h, added=a, comp=c, surf=a)
        third = spectra(h)
        # an operator-level read after a scene-level run sees consistent layers (ier+- / iet-- materialised on demand)
        ier_mp = from_abi(h.rrs_download(19), a.ier_mp)
        ier_pm = from_abi(h.rrs_download(18), a.ier_pm)
        assert np.all(np.isfinite(ier_mp)) and np.all(np.isfinite(ier_pm))
    for k in range(4):
        assert np.array_equal(first[k], fresh[k]), k
    for k, (x, y, z) in enumerate(zip(first, second, third)):
        assert np.array_equal(x, y), ("second run", k)
        assert np.array_equal(x, z), ("run after uploads", k)


@pytest.mark.parametrize("nS,lt", [(1, 3), (3, 11), (3, 21)])   # one wave per pair; workgroup kernels at 2 x 2 and 3 x 3 tiles
def test_rrs_empty_owned_range(rtamd, nS, lt):
    """mom_rrs_set_shard with n1_lo == n1_hi (a rank beyond the end of the axis when world > S / per): the run completes
    (no zero-sized launch) and every spectrum is zero (outputs exist for owned points only)."""
    rt = rtamd.corert
    m = rtamd.scenes.make_scene(nS, lt, 2, 20, seed=5)
    RS, _ = _rrs_inputs(rtamd, [-3, 2], False)
    model = rt._with_cabannes(RS, m)
    sc = rtamd.prepare_scene(model)
    Zr_pp, Zr_mp = rt.raman_z(RS, model)
    fs = rt.fscatt_rayleigh(model)
    for strict in (False, True):
        with rt.make_handle(model) as h:
            h.set_option(rtamd._lib.MOM_OPT_STRIP_PAD, 0)
            h.rrs_set(RS.i_λ1λ0, RS.ϖ_λ1λ0, strict)
            h.rrs_set_shard(20, 0, 7, 7)
            rt.scene_set(h, sc)
            h.scene_set_rrs(np.ascontiguousarray(fs.T), rt._abi_mats(Zr_pp), rt._abi_mats(Zr_mp))
            h.rt_run_rrs()
            R, T, ieR, ieT = h.get_RT_rrs()[:4]
        assert not np.any(ieR) and not np.any(ieT) and not np.any(R) and not np.any(T)


@pytest.mark.parametrize("nS,lt,nv", [(3, 5, 1), (4, 5, 1), (3, 11, 3), (4, 9, 3), (3, 21, 3), (4, 21, 3)])   # N = 15, 20 (one view), 27, 32, 42, 56
@pytest.mark.parametrize("strict", [True, False])
def test_rrs_zero_padding_invariant(rtamd, nS, lt, nv, strict):
    """The device blocks of the RRS layers are zero-padded to the MFMA tiling (16 x 16 at N <= 16, 32 x 32 above) and the kernels
    store WHOLE tiles: every stored quantity must keep the padding at exact zeros.  mom_rrs_check_padding counts
    the violations over all layer arrays after a scene-level run (which goes through every kernel of the path) -- and, r5, the
    damaged words of the 16 KB guard bands every layer array is allocated between (mom_rrs.hip dmg): an unmasked store that lands
    before the first or behind the last block of ANY array is a violation too, for one-tile ... 4 x 4-tile images."""
    rt = rtamd.corert
    m = rtamd.scenes.make_scene(nS, lt, 3, 26, seed=2 + nS + lt, aerosol_total=0.1, **VIEWS[nv])
    RS, _ = _rrs_inputs(rtamd, [-4, -1, 2, 7, 3], strict)
    model = rt._with_cabannes(RS, m)
    sc = rtamd.prepare_scene(model)
    Zr_pp, Zr_mp = rt.raman_z(RS, model)
    fs = rt.fscatt_rayleigh(model)
    with rt.make_handle(model) as h:
        h.set_option(rtamd._lib.MOM_OPT_STRIP_PAD, 0)
        h.rrs_set(RS.i_λ1λ0, RS.ϖ_λ1λ0, strict)
        assert h.rrs_check_padding() == 0
        rt.scene_set(h, sc)
        h.scene_set_rrs(np.ascontiguousarray(fs.T), rt._abi_mats(Zr_pp), rt._abi_mats(Zr_mp))
        h.rt_run_rrs()
        assert np.abs(h.get_RT_rrs()[2]).max() > 0
        assert h.rrs_check_padding() == 0
        h.rt_run_rrs()
        assert h.rrs_check_padding() == 0


@pytest.mark.parametrize("nS,lt,S,Nz", [(4, 9, 12, 2), (3, 21, 10, 2), (4, 21, 8, 2), (3, 33, 8, 2)])   # N = 32, 42, 56, 60: 2, 3, 4, 4 waves per pair
@pytest.mark.parametrize("strict", [True, False])
def test_rrs_workgroup_and_wave_kernels_agree(rtamd, nS, lt, S, Nz, strict):
    """Above N = 16 the RRS pair kernels run as one WORKGROUP per pair (mom_rrs_wg.hpp: column strips per wave, left factors from
    LDS); MOM_OPT_RRS_KERNELS without bit 0 selects the wave-per-pair bodies they replace.  Both execute the same products in the
    same order on the same operands, so a scene-level run must agree to rounding of the last place -- asserted at 1e-13 of the
    elastic intensity, three orders below the parity bound against the restatement (which test_rt_run_rrs_parity holds for the
    workgroup form).  (r5 needed a process per switch position: the switches were environment variables read once.)"""
    rt = rtamd.corert
    m = rtamd.scenes.make_scene(nS, lt, Nz, S, seed=nS + lt + S, aerosol_total=0.1)
    offs = np.asarray([-4, -1, 2, 7, 3])
    vp = 0.04 / len(offs) * (1.0 + 0.1 * np.arange(len(offs)))
    RS = rt.RRS(greek_raman=rt.get_greek_rayleigh(0.2), ϖ_Cabannes=0.96, ϖ_λ1λ0=vp, i_λ1λ0=offs, rrs_strict_reference=bool(strict))
    # mask bits: 1 workgroup per pair, 2 ... for 16 < N <= 32 too, 4 workgroup per point, 8 tile-form elemental, 16 / 32 the
    # inelastic elemental layer formed inside the first doubling step always / never.
    # "point": the point kernels as one wave per point; "fuse": elemental inside the first doubling step; "elementwise": r4's kernel
    masks = {"0": 2 | 4 | 8, "1": 15, "point": 1 | 2 | 8, "fuse": 15 | 16, "fuse0": 2 | 4 | 8 | 16, "elementwise": 1 | 2 | 4}
    out = {}
    for wg, mask in masks.items():
        out[wg] = dict(zip(("R", "T", "ieR", "ieT", "hdr", "up", "dw"), rt.rt_run_rrs(RS, m, kernels=mask)))
        out[wg]["N"] = rtamd.prepare_scene(rt._with_cabannes(RS, m)).N
    assert int(out["0"]["N"]) == {(4, 9): 32, (3, 21): 42, (4, 21): 56, (3, 33): 60}[(nS, lt)]
    scale = np.abs(out["0"]["R"][:, 0:1, :]).max()
    assert np.abs(out["0"]["ieR"]).max() > 0
    for other in ("1", "point", "fuse", "fuse0", "elementwise"):
        for k in ("R", "T", "ieR", "ieT", "hdr", "up", "dw"):
            d = np.abs(out["0"][k] - out[other][k]).max()
            # the element-wise elemental kernel evaluates 1 - exp(-(x + y)) where the tile forms evaluate 1 - exp(-x) exp(-y)
            assert d <= (1e-11 if other == "elementwise" else 1e-13) * scale, (other, k, d, scale)
