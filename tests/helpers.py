"""Shared test helpers: model <-> oracle scene conversion and the parity tolerances."""
import numpy as np

from oracle import momref as mr

# Parity bar of BASELINE.json north_star: Stokes I/Q/U within 1e-10 relative.  "Relative" is
# taken against the intensity |I| of the same view and spectral point (Q and U cross zero), plus an
# absolute floor of 1e-14 of the unit incoming solar flux for points whose reference value is itself
# rounding noise (deep absorption, tau ~ 17+: the reference's own I comes out as -2.6e-16 there).
RTOL_STOKES = 1e-10
ATOL_STOKES = 1e-14
# per-operator bar: 1e-12 of the operator's largest element
RTOL_OP = 1e-12


def stokes_rtol(ndoubl) -> float:
    """Parity tolerance of GPU-vs-Float64-oracle comparisons as a function of the scene's largest doubling number:
    1e-10 (north star) up to ndoubl = 15, then 8 * 2^nd * eps (1.2e-10 at 16, 9.3e-10 at 19, 3.7e-9 at 21, 3.0e-8 at 24).

    Each doubling squares the direct transmission, so a one-ulp difference anywhere early is doubled nd times: every
    Float64 run of the algorithm -- the LU oracle included -- is a multiple of 2^nd eps away from the exact result of the
    same equations.  tests/test_gpu_precision.py measures both against the x87 extended-precision build of the oracle
    (oracle/momref_ext.c) on the thick scenes of this suite: the oracle's own error is 3e-11 at nd = 16, 1.6e-9 at
    nd = 18, 4.3e-9 at nd = 21, 8.5e-9 at nd = 24 (N = 256); the GPU default path's error is 0.7 .. 1.13 x the oracle's
    (asserted <= 4 x), and the two differ from each other by 1e-12 (nd 16), 2e-11 (17..18), 1.5e-10 (21), 5.8e-10 (24) --
    that difference is what this bar bounds.  With MOM_OPT_INVERSE = 1 (pivoted Gauss-Jordan: the oracle's rounding
    pattern) the GPU agrees with the oracle to 1e-11 on all of them."""
    nd = int(np.max(np.asarray(ndoubl))) if np.size(ndoubl) else 0
    return max(RTOL_STOKES, 8.0 * 2.0 ** nd * float(np.finfo(np.float64).eps))


def assert_stokes_close(X, Xref, rtol=RTOL_STOKES, atol=ATOL_STOKES, what=""):
    X, Xref = np.asarray(X), np.asarray(Xref)
    assert X.shape == Xref.shape, (X.shape, Xref.shape)
    assert np.all(np.isfinite(X)), f"{what}: non-finite values"
    scale = np.abs(Xref[:, 0:1, :])
    err = np.abs(X - Xref)
    bad = err > rtol * scale + atol
    if np.any(bad):
        idx = np.unravel_index(np.argmax(err / (rtol * scale + atol)), err.shape)
        raise AssertionError(f"{what}: |Δ|={err[idx]:.3e} at {idx}, ref={Xref[idx]:.6e}, I_ref={scale[idx[0],0,idx[2]]:.3e}")
    return float(np.max(err / np.maximum(scale, atol / rtol)))


def assert_op_close(X, Xref, rtol=RTOL_OP, what=""):
    X, Xref = np.asarray(X).reshape(-1), np.asarray(Xref).reshape(-1)
    assert X.shape == Xref.shape
    assert np.all(np.isfinite(X)), f"{what}: non-finite values"
    scale = max(float(np.max(np.abs(Xref))), 1e-300)
    err = float(np.max(np.abs(X - Xref))) / scale
    assert err <= rtol, f"{what}: max |Δ|/max|ref| = {err:.3e} > {rtol:.1e}"
    return err


def oracle_scene(model) -> mr.Scene:
    """vSmartMOM_Model (product host types) -> numpy-twin Scene (oracle types).  The oracle
    recomputes streams, Z moments and layer optics with ITS OWN code from the same physical inputs."""
    p = model.params
    n = p.polarization_type.n
    quad = mr.rt_set_streams(p.quadrature_type, p.l_trunc, p.sza, p.vza, n)
    aer = [mr.AerosolOptics(mr.GreekCoefs(a.greek_coefs.α, a.greek_coefs.β, a.greek_coefs.γ, a.greek_coefs.δ,
                                          a.greek_coefs.ϵ, a.greek_coefs.ζ), a.ω̃, a.fᵗ) for a in model.aerosol_optics]
    b = getattr(p, "brdf", None)
    brdf = None
    if b is not None:
        nm = type(b).__name__
        if nm == "rpvSurfaceScalar":
            brdf = ("rpv", b.ρ0, b.ρ_c, b.k, b.Θ)
        elif nm == "RossLiSurfaceScalar":
            brdf = ("rossli", b.fvol, b.fgeo, b.fiso)
        elif nm == "LambertianSurfaceLegendre":
            brdf = ("legendre",) + tuple(b.legendre_coeff)
        elif nm == "LambertianSurfaceScalar":
            p = type(p)(**{**p.__dict__, "brdf_albedo": b.albedo})
    return mr.Scene(brdf=brdf, pol=mr.pol_from_n(n), quad=quad, max_m=p.max_m, tau_rayl=model.τ_rayl, tau_abs=model.τ_abs,
                    greek_rayleigh=mr.get_greek_rayleigh(p.depol), tau_aer=model.τ_aer, aerosols=aer,
                    varpi_cabannes=model.ϖ_Cabannes, albedo=p.brdf_albedo, vza=np.asarray(p.vza, float),
                    vaz=np.asarray(p.vaz, float), strict_reference_indexing=p.strict_reference_indexing)


def one_layer_rayleigh(rt, sza, vza, vaz, tau, rho, pol=None, quad="RadauQuad", l_trunc=20, max_m=3):
    """The set-up of test/test_CoreRT.jl: one Rayleigh layer with τ_rayl forced to `tau`
    (test_CoreRT.jl:21,:63), no absorption, Lambertian albedo rho, depol 0."""
    pol = pol or rt.Stokes_IQUV()
    params = rt.vSmartMOM_Parameters(polarization_type=pol, quadrature_type=quad, max_m=max_m, l_trunc=l_trunc,
                                     depol=0.0, sza=sza, vza=np.asarray(vza, float), vaz=np.asarray(vaz, float),
                                     brdf_albedo=rho)
    return rt.model_from_parameters(params, np.full((1, 1), tau), np.zeros((1, 1)))
