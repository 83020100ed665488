"""Operator-level parity on the GPU, through the C ABI: each reference operator
(elemental!, doubling!, interaction!, create_surface_layer!, batch_inv!, ⊠) against the C oracle on
the same seeded inputs.  Shapes follow the reference's own GPU-vs-CPU scripts (test/gpu_tests/*.jl)."""
import numpy as np
import pytest

import helpers
from oracle import momref as mr

pytestmark = pytest.mark.gpu


def _streams(rt, nS, Nquad=6, sza=50.0):
    pol = {1: rt.Stokes_I, 3: rt.Stokes_IQU, 4: rt.Stokes_IQUV}[nS]()
    q = rt.rt_set_streams("GaussQuadHemisphere", 2 * (Nquad - 3) - 1 if Nquad > 3 else 1, sza, [0.0, 30.0], pol)
    return pol, q


def _handle(rtamd, pol, q, S, strict=True, generic=False, force_gj=False):
    h = rtamd.Handle(len(q.qp_μN), pol.n, S, 1)
    if generic:
        h.set_option(rtamd._lib.MOM_OPT_FORCE_GENERIC, 1)
    if force_gj:
        h.set_option(rtamd._lib.MOM_OPT_INVERSE, 1)
    h.set_streams(q.qp_μN, q.wt_μN, q.iμ0, q.μ0, pol.I0, pol.D, strict)
    return h


class _P:  # what oracle.cref op wrappers need
    def __init__(self, pol, q, strict=True):
        self.N, self.nS, self.imu0, self.mu0 = len(q.qp_μN), pol.n, q.iμ0, q.μ0
        self.mu, self.wt = np.ascontiguousarray(q.qp_μN), np.ascontiguousarray(q.wt_μN)
        self.I0, self.D, self.strict, self.albedo = np.array(pol.I0), np.array(pol.D), int(strict), 0.3


NAMES = ["r_pm", "r_mp", "t_mm", "t_pp", "j0p", "j0m"]


@pytest.mark.parametrize("nS", [1, 3, 4])
@pytest.mark.parametrize("strict", [True, False])
@pytest.mark.parametrize("nd", [0, 1, 6])
def test_elemental_and_doubling(rtamd, cref, nS, strict, nd):
    """elemental_test.jl (n=40 there) + D_matrix_test.jl (nStokes 1..4) + gpu_cpu_tests.jl doubling."""
    rt = rtamd.corert
    pol, q = _streams(rt, nS, Nquad=8)
    N, S = len(q.qp_μN), 37
    rng = np.random.default_rng(nS * 10 + nd)
    g = rtamd.scenes.hg_like_greek(0.6, 9)
    Zpp, Zmp = rt.compute_Z_moments(pol, q.qp_μ, g, 1)
    zb = 1 if nd == 1 else S  # both the shared and the per-point Z form
    Zp = np.repeat(Zpp[None], zb, 0) * (1 + 0.1 * rng.random((zb, 1, 1)))
    Zm = np.repeat(Zmp[None], zb, 0) * (1 + 0.1 * rng.random((zb, 1, 1)))
    dtau = 10.0 ** rng.uniform(-7, -3, S)
    varpi = rng.uniform(0.0, 1.0, S)
    tau_sum = rng.uniform(0.0, 3.0, S)
    p = _P(pol, q, strict)
    ref = cref.elemental(p, 1, nd, tau_sum, dtau, varpi, mr.to_abi(Zp), mr.to_abi(Zm), zb, S)
    with _handle(rtamd, pol, q, S, strict) as h:
        h.elemental(1, nd, tau_sum, dtau, varpi, mr.to_abi(Zp), mr.to_abi(Zm), zb)
        for k, nm in enumerate(NAMES):
            if nd >= 1 and nm in ("r_pm", "t_mm"):
                continue
            helpers.assert_op_close(h.download(k), ref[k], what=f"elemental {nm}")
        expk = np.exp(-dtau / q.μ0)
        e_ref = expk.copy()
        assert cref.doubling(p, nd, e_ref, ref, S) == 0
        e_gpu = h.doubling(nd, expk)
        np.testing.assert_allclose(e_gpu, e_ref, rtol=1e-15)
        for k, nm in enumerate(NAMES):
            if nd == 0 and False:
                continue
            helpers.assert_op_close(h.download(k), ref[k], rtol=1e-11, what=f"doubling {nm}")


@pytest.mark.parametrize("mode", ["lds", "lds_gj", "generic", "generic_gj"])
def test_doubling_thick_layer_all_inverse_paths(rtamd, cref, mode):
    """Strongly reflecting layer: r large enough that the Neumann bound does not apply -> pivoted
    Gauss-Jordan; thin steps take the series path.  All four code paths must agree with the oracle."""
    rt = rtamd.corert
    pol, q = _streams(rt, 3, Nquad=9)
    N, S, nd = len(q.qp_μN), 19, 14
    rng = np.random.default_rng(4)
    Zpp, Zmp = rt.compute_Z_moments(pol, q.qp_μ, rtamd.scenes.hg_like_greek(0.75, 11), 0)
    dtau = np.full(S, 2e-4) * rng.uniform(0.5, 1.0, S)  # final tau ~ 3: conservative cloud
    varpi = np.full(S, 0.999999)
    p = _P(pol, q)
    ref = cref.elemental(p, 0, nd, np.zeros(S), dtau, varpi, mr.to_abi(Zpp[None]), mr.to_abi(Zmp[None]), 1, S)
    e_ref = np.exp(-dtau / q.μ0)
    assert cref.doubling(p, nd, e_ref, ref, S) == 0
    with _handle(rtamd, pol, q, S, generic="generic" in mode, force_gj="gj" in mode) as h:
        h.elemental(0, nd, np.zeros(S), dtau, varpi, mr.to_abi(Zpp[None]), mr.to_abi(Zmp[None]), 1)
        h.doubling(nd, np.exp(-dtau / q.μ0))
        for k, nm in enumerate(NAMES):
            helpers.assert_op_close(h.download(k), ref[k], rtol=1e-10, what=f"{mode} {nm}")
        assert np.abs(ref[1]).max() > 0.05  # the layer really is reflective


def test_doubling_thick_layer_panel_gemm_series(rtamd, cref):
    """The same strongly reflecting layer at an operator edge above 64 (N = 81: panel GEMM from the slab): where the norm of
    r r leaves the series table (beta > 0.29) the generic mode carries the series on (p from the norm, up to 512 terms by
    repeated squaring) instead of the global-memory Gauss-Jordan; against the oracle's LU and against the forced pivoted path."""
    rt = rtamd.corert
    pol, q = _streams(rt, 3, Nquad=27)
    N, S, nd = len(q.qp_μN), 3, 14
    assert N == 81
    rng = np.random.default_rng(4)
    Zpp, Zmp = rt.compute_Z_moments(pol, q.qp_μ, rtamd.scenes.hg_like_greek(0.75, 11), 0)
    dtau = np.full(S, 2e-4) * rng.uniform(0.5, 1.0, S)
    varpi = np.full(S, 0.999999)
    p = _P(pol, q)
    ref = cref.elemental(p, 0, nd, np.zeros(S), dtau, varpi, mr.to_abi(Zpp[None]), mr.to_abi(Zmp[None]), 1, S)
    assert cref.doubling(p, nd, np.exp(-dtau / q.μ0), ref, S) == 0
    got = {}
    for gj in (False, True):
        with _handle(rtamd, pol, q, S, force_gj=gj) as h:
            h.elemental(0, nd, np.zeros(S), dtau, varpi, mr.to_abi(Zpp[None]), mr.to_abi(Zmp[None]), 1)
            h.doubling(nd, np.exp(-dtau / q.μ0))
            got[gj] = [h.download(k) for k in range(6)]
    for k, nm in enumerate(NAMES):
        helpers.assert_op_close(got[False][k], ref[k], rtol=1e-10, what=f"series {nm}")
        helpers.assert_op_close(got[True][k], ref[k], rtol=1e-10, what=f"gauss-jordan {nm}")
    r = np.asarray(ref[1]).reshape(S, N, N)[0]
    assert np.linalg.norm(r @ r) > 0.3   # beyond the table's beta = 0.29: the extended series really ran


@pytest.mark.parametrize("iface", [0, 1, 2, 3])
@pytest.mark.parametrize("N,S,generic", [(16, 100, False), (16, 100, True), (32, 40, False), (72, 3, False)])
def test_interaction(rtamd, cref, iface, N, S, generic):
    """gpu_batched_interaction2.jl: n=16, nSpec=100, Float64, random operators."""
    rt = rtamd.corert
    rng = np.random.default_rng(iface + N)
    mk = lambda s: rng.random((S, N, N)) * s / N
    added = [mk(0.6), mk(0.6), mk(0.9) + 0.3 * np.eye(N), mk(0.9) + 0.3 * np.eye(N), rng.random((S, N)), rng.random((S, N))]
    comp = [mk(0.6), mk(0.6), mk(0.9), mk(0.9), rng.random((S, N)), rng.random((S, N))]
    c_add = [mr.to_abi(x).copy() for x in added]
    c_comp = [mr.to_abi(x).copy() for x in comp]
    pol = rt.Stokes_I()
    mu = np.linspace(0.1, 1.0, N)
    h = rtamd.Handle(N, 1, S, 1)
    if generic:
        h.set_option(rtamd._lib.MOM_OPT_FORCE_GENERIC, 1)
    h.set_streams(mu, np.full(N, 1.0 / N), 1, mu[0], pol.I0, pol.D, True)
    for k in range(6):
        h.upload(k, c_add[k])
        h.upload(6 + k, c_comp[k])
    h.interaction(iface)
    assert cref.interaction(N, S, iface, c_comp, c_add) == 0
    for k, nm in enumerate(["R_mp", "R_pm", "T_pp", "T_mm", "J0p", "J0m"]):
        helpers.assert_op_close(h.download(6 + k), c_comp[k], rtol=1e-11, what=f"iface {iface} {nm}")
    h.close()


@pytest.mark.parametrize("nS", [1, 3, 4])
def test_surface_lambertian(rtamd, cref, nS):
    rt = rtamd.corert
    pol, q = _streams(rt, nS, Nquad=7)
    S = 11
    p = _P(pol, q)
    tau_tot = np.random.default_rng(1).uniform(0.1, 4.0, S)
    with _handle(rtamd, pol, q, S) as h:
        for m in (0, 1):
            ref = cref.surface_lambertian(p, m, tau_tot, S)
            h.surface_lambertian(m, p.albedo, tau_tot)
            for k, nm in enumerate(NAMES):
                helpers.assert_op_close(h.download(12 + k), ref[k] if (m == 0 or nm != "r_pm") else ref[k] * 0,
                                        rtol=1e-14, what=f"surface m={m} {nm}")


@pytest.mark.parametrize("n,batch", [(32, 1000), (60, 64), (7, 33), (100, 5), (65, 3), (127, 3), (129, 2), (144, 2), (160, 2), (176, 2), (192, 2),
                                     (250, 2), (255, 2)])   # 144 ... 176: 9 ... 11 column tiles, unevenly shared over the wave columns
def test_batch_inv_and_mul(rtamd, cref, n, batch):
    """matrix_inv_test.jl: n=32, batch 10 000 (reduced)."""
    rng = np.random.default_rng(n)
    A = rng.normal(size=(batch, n, n)) + 3 * np.eye(n)
    B = rng.normal(size=(batch, n, n))
    with rtamd.Handle(4, 1, 1, 1) as h:
        X = h.batch_inv(n, batch, mr.to_abi(A))
        Xr, info = cref.batch_inv(n, batch, mr.to_abi(A))
        assert info == 0
        helpers.assert_op_close(X, Xr, rtol=1e-11, what="batch_inv")
        Cg = h.batched_mul(n, batch, mr.to_abi(A), mr.to_abi(B))
        helpers.assert_op_close(Cg, cref.batched_mul(n, batch, mr.to_abi(A), mr.to_abi(B)), rtol=1e-13, what="⊠")


@pytest.mark.parametrize("n,batch,P", [(32, 50, 3), (60, 16, 2), (7, 33, 1), (100, 4, 2), (16, 8, 0)])
def test_dual_batch_inv_and_mul(rtamd, n, batch, P):
    """The ForwardDiff.Dual methods of batch_inv! and batched_mul (gpu_batched.jl:100-150) through
    mom_batch_inv_dual / mom_batched_mul_dual against the numpy restatement (itself pinned by finite differences in
    tests/test_oracle_twin.py)."""
    rng = np.random.default_rng(n + P)
    A = rng.normal(size=(batch, n, n)) + 3 * np.eye(n)
    B = rng.normal(size=(batch, n, n))
    dA, dB = rng.normal(size=(P, batch, n, n)), rng.normal(size=(P, batch, n, n))
    abi = lambda M: mr.to_abi(M.reshape(-1, n, n)) if M.size else np.zeros(0)
    Cr, dCr = mr.batched_mul_dual(A, dA, B, dB)
    Xr, dXr = mr.batch_inv_dual(A, dA)
    with rtamd.Handle(4, 1, 1, 1) as h:
        Cg, dCg = h.batched_mul_dual(n, batch, P, abi(A), abi(dA), abi(B), abi(dB))
        Xg, dXg = h.batch_inv_dual(n, batch, P, abi(A), abi(dA))
    helpers.assert_op_close(Cg, abi(Cr), rtol=1e-13, what="C")
    helpers.assert_op_close(Xg, abi(Xr), rtol=1e-11, what="X")
    if P:
        helpers.assert_op_close(dCg, abi(dCr), rtol=1e-13, what="dC")
        helpers.assert_op_close(dXg, abi(dXr), rtol=1e-10, what="dX")


@pytest.mark.parametrize("nS,strict,nd,m", [(1, True, 0, 0), (3, True, 3, 1), (3, False, 0, 2), (4, True, 1, 0), (4, False, 5, 1)])
def test_elemental_inelastic_rrs(rtamd, nS, strict, nd, m):
    """elemental_inelastic!(::RRS) (elemental_inelastic.jl:23-91) through mom_elemental_inelastic_rrs against the numpy
    restatement: Raman offsets of both signs (entries falling off the grid stay zero), equal and unequal elemental
    optical thicknesses (both branches of the diagonal transmission), ndoubl = 0 (mirror operators by the D rule) and
    >= 1 (row signs, D on ieJ0-)."""
    rt = rtamd.corert
    pol, q = _streams(rt, nS, Nquad=7)
    N, S = len(q.qp_μN), 23
    rng = np.random.default_rng(17 * nS + nd)
    g = rtamd.scenes.hg_like_greek(0.5, 9)
    Zpp, Zmp = rt.compute_Z_moments(pol, q.qp_μ, g, m)
    dtau = 10.0 ** rng.uniform(-6, -2, S)
    dtau[5:9] = dtau[5]                       # equal thicknesses: the |dtau0 - dtau1| <= 1e-6 branch also off n0 == n1
    varpi, fscatt, tau_sum = rng.uniform(0.1, 1.0, S), rng.uniform(0.5, 1.0, S), rng.uniform(0.0, 2.0, S)
    i_l, vp = np.array([-4, -1, 0, 2, 7]), rng.uniform(0.001, 0.05, 5)
    mq = mr.QuadPoints(q.μ0, q.iμ0, pol.n * (q.iμ0 - 1) + 1, np.asarray(q.qp_μ), np.asarray(q.wt_μ), np.asarray(q.qp_μN),
                       np.asarray(q.wt_μN), len(q.qp_μ))
    ref = mr.elemental_inelastic_rrs(mr.pol_from_n(nS), mq, i_l, vp, fscatt, tau_sum, dtau, varpi, Zpp, Zmp, m, nd, strict)
    with _handle(rtamd, pol, q, S, strict) as h:
        got = h.elemental_inelastic_rrs(m, nd, i_l, vp, fscatt, tau_sum, dtau, varpi, mr.to_abi(Zpp[None]), mr.to_abi(Zmp[None]))
    names = ["ier_mp", "iet_pp", "ier_pm", "iet_mm", "ieJ0p", "ieJ0m"]
    for k, nm in enumerate(names):
        if nd >= 1 and nm in ("ier_pm", "iet_mm"):
            assert not np.any(got[k])           # untouched by the reference for ndoubl >= 1
            continue
        want = mr.to_abi(ref[k].reshape(-1, N, N)) if k < 4 else ref[k].reshape(-1)
        assert np.any(want)
        # differences of exponentials over 1 - dtau1/dtau0: one ulp of the device exp vs libm is amplified by the
        # cancellation (same bar as the elastic per-operator tests: 1e-12 of the operator's largest element, here x10)
        helpers.assert_op_close(got[k], want, rtol=1e-11, what=nm)


@pytest.mark.parametrize("n,batch", [(32, 200), (60, 16), (100, 4)])
def test_batch_inv_and_mul_float32(rtamd, n, batch):
    """batch_inv! / batched_mul on Float32 arrays (gpu_batched.jl:45-58, 90-97) through a dtype = 1 handle: f32 MFMA
    products and the pivoted inverse in f32, Float64 host arrays at the ABI."""
    rng = np.random.default_rng(n)
    A = (rng.normal(size=(batch, n, n)) + 4 * np.sqrt(n) * np.eye(n)).astype(np.float32).astype(np.float64)
    B = rng.normal(size=(batch, n, n)).astype(np.float32).astype(np.float64)
    with rtamd.Handle(4, 1, 1, 1, dtype=1) as h:
        X = h.batch_inv(n, batch, mr.to_abi(A))
        Cg = h.batched_mul(n, batch, mr.to_abi(A), mr.to_abi(B))
    helpers.assert_op_close(X, mr.to_abi(np.linalg.inv(A)), rtol=2e-5, what="batch_inv f32")
    helpers.assert_op_close(Cg, mr.to_abi(A @ B), rtol=2e-6, what="batched_mul f32")


@pytest.mark.parametrize("nS,Nquad,nd,generic", [(3, 8, 5, False), (1, 6, 3, False), (4, 8, 7, False), (3, 20, 6, False), (3, 9, 4, True),
                                                 (3, 27, 5, False)])
def test_float32_operator_level(rtamd, cref, nS, Nquad, nd, generic):
    """dtype = 1 through every operator (the reference's float_type = Float32, parameters_from_yaml.jl:160; its GPU tests run
    the batched operators in Float32, test/gpu_tests/gpu_batched_interaction.jl): mom_elemental -> mom_doubling ->
    mom_copy_added_to_composite -> second layer -> mom_interaction (case 11) -> mom_surface_lambertian -> mom_interaction on a
    Float32 handle, each step against the Float32 oracle (oracle/momref_f32.c: the same operations rounded to Float32) and the
    Float64 oracle.  Operator edges 9 ... 81: LDS-resident kernels, forced generic mode, the panel GEMM."""
    rt = rtamd.corert
    pol, q = _streams(rt, nS, Nquad=Nquad)
    N, S = len(q.qp_μN), 9
    rng = np.random.default_rng(nS + Nquad)
    Zpp, Zmp = rt.compute_Z_moments(pol, q.qp_μ, rtamd.scenes.hg_like_greek(0.6, 9), 0)
    Zp, Zm = mr.to_abi(Zpp[None]), mr.to_abi(Zmp[None])
    p = _P(pol, q)
    eps = 6e-8

    def close(got, ref32, ref64, what, amp):
        # against the Float32 oracle: eps32 x the error amplification of the step; against Float64: that + the oracle's own distance
        e32 = helpers.assert_op_close(got, np.asarray(ref32, np.float64), rtol=amp * eps, what=f"{what} vs f32 oracle")
        d = float(np.max(np.abs(np.asarray(ref32, np.float64) - ref64))) / max(float(np.max(np.abs(ref64))), 1e-300)
        helpers.assert_op_close(got, ref64, rtol=amp * eps + 2 * d, what=f"{what} vs f64 oracle")
        return e32

    h = rtamd.Handle(N, pol.n, S, 1, dtype=1)
    if generic:
        h.set_option(rtamd._lib.MOM_OPT_FORCE_GENERIC, 1)
    h.set_streams(q.qp_μN, q.wt_μN, q.iμ0, q.μ0, pol.I0, pol.D, True)
    with h:
        layers32, layers64 = [], []
        for z in range(2):
            dtau = 10.0 ** rng.uniform(-4, -3.3, S)
            varpi = rng.uniform(0.3, 1.0, S)
            tau_sum = rng.uniform(0.0, 1.0, S)
            a64 = cref.elemental(p, 0, nd, tau_sum, dtau, varpi, Zp, Zm, 1, S)
            a32 = cref.elemental_f32(p, 0, nd, tau_sum, dtau, varpi, Zp, Zm, 1, S)
            h.elemental(0, nd, tau_sum, dtau, varpi, Zp, Zm, 1)
            for k, nm in enumerate(NAMES):
                if nm in ("r_pm", "t_mm"):
                    continue                     # not written for nd >= 1 (elemental.jl:255-274)
                # 1 - exp(-x), x = dtau (1/mu_i + 1/mu_j), in Float32: one ulp of the exponential is eps32 / x of the result, and
                # the GPU's expf and the C library's differ by an ulp
                close(h.download(k), a32[k], a64[k], f"layer {z} elemental {nm}", 8.0 / float(dtau.min()))
            expk = np.exp(-dtau / q.μ0)
            e64, e32 = expk.copy(), expk.astype(np.float32)
            assert cref.doubling(p, nd, e64, a64, S) == 0 and cref.doubling_f32(p, nd, e32, a32, S) == 0
            e_gpu = h.doubling(nd, expk)
            np.testing.assert_allclose(e_gpu, e32.astype(np.float64), rtol=4 * eps * 2 ** nd)
            # the elemental layer's relative error stays; each doubling squares the direct transmission (absolute rounding doubles)
            amp = 8.0 / float(dtau.min()) + 64 * 2.0 ** nd
            for k, nm in enumerate(NAMES):
                close(h.download(k), a32[k], a64[k], f"layer {z} doubling {nm}", amp)
            layers32.append(a32); layers64.append(a64)
            if z == 0:
                h.copy_added_to_composite()
                order = [1, 0, 3, 2, 4, 5]       # composite R_mp, R_pm, T_pp, T_mm, J0p, J0m <- added r_mp, r_pm, t_pp, t_mm, j0p, j0m
                c32 = [a32[i].copy() for i in order]
                c64 = [a64[i].copy() for i in order]
        h.interaction(3)
        assert cref.interaction(N, S, 3, c64, layers64[1]) == 0 and cref.interaction_f32(N, S, 3, c32, layers32[1]) == 0
        for k, nm in enumerate(["R_mp", "R_pm", "T_pp", "T_mm", "J0p", "J0m"]):
            close(h.download(6 + k), c32[k], c64[k], f"interaction {nm}", 2 * amp)
        tau_tot = rng.uniform(0.1, 2.0, S)
        s64 = cref.surface_lambertian(p, 0, tau_tot, S)
        h.surface_lambertian(0, p.albedo, tau_tot)
        for k, nm in enumerate(NAMES):
            helpers.assert_op_close(h.download(12 + k), s64[k], rtol=8 * eps, what=f"surface {nm}")
        h.interaction(3, with_surface_layer=True)
        s32 = [np.asarray(x, np.float64).astype(np.float32) for x in s64]
        assert cref.interaction(N, S, 3, c64, s64) == 0 and cref.interaction_f32(N, S, 3, c32, s32) == 0
        for k, nm in enumerate(["R_mp", "R_pm", "T_pp", "T_mm", "J0p", "J0m"]):
            close(h.download(6 + k), c32[k], c64[k], f"surface interaction {nm}", 2 * amp)


def test_singular_operator_is_reported(rtamd):
    """The reference ignores cuBLAS `info` (gpu_batched.jl:65-70); here a zero pivot surfaces as MOM_ESINGULAR."""
    n, batch = 12, 4
    A = np.random.default_rng(0).normal(size=(batch, n, n))
    A[2, :, 5] = 0.0
    with rtamd.Handle(4, 1, 1, 1) as h:
        with pytest.raises(rtamd.MomError) as e:
            h.batch_inv(n, batch, mr.to_abi(A))
        assert e.value.code == rtamd._lib.MOM_ESINGULAR
        h.batch_inv(n, batch, mr.to_abi(A + 5 * np.eye(n)))  # the handle stays usable


def test_call_sequence_errors(rtamd):
    with rtamd.Handle(12, 3, 4, 1) as h:
        with pytest.raises(rtamd.MomError) as e:
            h.interaction(3)
        assert e.value.code == rtamd._lib.MOM_ESTATE
        with pytest.raises(rtamd.MomError):
            h.rt_run()
