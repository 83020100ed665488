"""
oracle/cref.py -- TEST INFRASTRUCTURE ONLY: ctypes binding of oracle/libmomref.so (the
C restatement, momref.c) plus `pack_scene`, which turns a numpy-twin `Scene` into the flat
host-prepared arrays both the C oracle and the HIP library's C-ABI consume.
"""
from __future__ import annotations

import ctypes as C
import os
import subprocess
from pathlib import Path

import numpy as np

from . import momref as mr

_HERE = Path(__file__).resolve().parent
_LIB = None

c_dp = C.POINTER(C.c_double)
c_ip = C.POINTER(C.c_int)


def build(force: bool = False, ext: bool = False, which: str = "") -> Path:
    which = which or ("ext" if ext else "")
    so = _HERE / (f"libmomref_{which}.so" if which else "libmomref.so")
    srcs = [_HERE / "momref.c", _HERE / "Makefile"] + ([_HERE / f"momref_{which}.c"] if which else [])
    if force or (not so.exists()) or so.stat().st_mtime < max(f.stat().st_mtime for f in srcs):
        subprocess.check_call(["make", "-C", str(_HERE), "-B" if force else "-s", so.name])
    return so


def lib():
    global _LIB
    if _LIB is None:
        _LIB = C.CDLL(str(build()))
        _LIB.omp_set_num_threads(effective_cores())  # libgomp's default would be the machine's logical CPU count
        _LIB.ora_w_hw32sd_re.restype = C.c_double
        _LIB.ora_w_hw32sd_re.argtypes = [C.c_double, C.c_double]
    return _LIB


_LIB_EXT = None


def lib_ext():
    """oracle/libmomref_ext.so: momref.c compiled in x87 extended precision (momref_ext.c)."""
    global _LIB_EXT
    if _LIB_EXT is None:
        assert np.finfo(np.longdouble).nmant == 63, "numpy.longdouble is not the x87 80-bit format on this host"
        _LIB_EXT = C.CDLL(str(build(ext=True)))
    return _LIB_EXT


def effective_cores() -> int:
    """CPUs this process may actually use: the affinity mask and the cgroup CPU quota (the GPU boxes show 256 logical
    CPUs but run the container under a 16-CPU quota: 256 OpenMP threads there only fight over 16 CPUs' time)."""
    n = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    try:
        quota, period = Path("/sys/fs/cgroup/cpu.max").read_text().split()[:2]
        if quota != "max":
            n = min(n, max(1, int(int(quota) / int(period))))
    except (OSError, ValueError):
        pass
    return max(1, n)


def dp(a):
    return a.ctypes.data_as(c_dp)


def ip(a):
    return a.ctypes.data_as(c_ip)


class OraScene(C.Structure):
    _fields_ = [("N", C.c_int), ("nS", C.c_int), ("S", C.c_int), ("Nz", C.c_int), ("K", C.c_int), ("M", C.c_int),
                ("imu0", C.c_int), ("strict", C.c_int), ("mu0", C.c_double),
                ("mu", c_dp), ("wt", c_dp), ("I0", c_dp), ("D", c_dp),
                ("tau", c_dp), ("varpi", c_dp), ("zw", c_dp), ("Zpp", c_dp), ("Zmp", c_dp),
                ("nd", c_ip), ("iface", c_ip), ("tau_sum", c_dp), ("albedo", C.c_double),
                ("nVza", C.c_int), ("node", c_ip), ("cos_mphi", c_dp), ("sin_mphi", c_dp),
                ("surf_kind", C.c_int), ("Rsurf", c_dp), ("albedo_spec", c_dp)]


class Packed:
    """Flat, host-prepared inputs of the hot path (what the reference's Julia host code
    hands to rt_kernel!/interaction!/postprocessing_vza!), in ABI memory order."""

    def __init__(self, scene: mr.Scene):
        pol, quad = scene.pol, scene.quad
        self.N, self.nS, self.S, self.Nz, self.M = scene.N, pol.n, scene.S, scene.Nz, scene.max_m
        self.imu0, self.mu0 = quad.imu0, quad.mu0
        self.strict = 1 if scene.strict_reference_indexing else 0
        self.mu = np.ascontiguousarray(quad.qp_muN, dtype=np.float64)
        self.wt = np.ascontiguousarray(quad.wt_muN, dtype=np.float64)
        self.I0 = np.ascontiguousarray(pol.I0, dtype=np.float64)
        self.D = np.ascontiguousarray(pol.D, dtype=np.float64)
        S, Nz, N = self.S, self.Nz, self.N
        Zpp_all, Zmp_all = [], []
        for m in range(self.M):
            layers = mr.construct_core_optical_properties(scene, m)
            Zpp_all.append(layers[0].Zpp_basis)
            Zmp_all.append(layers[0].Zmp_basis)
        self.K = layers[0].Zpp_basis.shape[0]
        K = self.K
        # [N,N,K,M] column-major: element (i,j,k,m) at i + N*(j + N*(k + K*m))
        self.Zpp = np.ascontiguousarray(np.transpose(np.array(Zpp_all), (0, 1, 3, 2))).reshape(-1)  # [m,k,j,i]
        self.Zmp = np.ascontiguousarray(np.transpose(np.array(Zmp_all), (0, 1, 3, 2))).reshape(-1)
        self.tau = np.ascontiguousarray(np.array([l.tau for l in layers])).reshape(-1)  # [z][n]
        self.varpi = np.ascontiguousarray(np.array([l.varpi for l in layers])).reshape(-1)
        # zw[k + K*(n + S*z)] -> array [z][n][k]
        self.zw = np.ascontiguousarray(np.array([l.zweights.T for l in layers])).reshape(-1)
        ifaces, tau_sum = mr.extract_effective_props(layers)
        self.iface = np.array(ifaces, dtype=np.int32)
        self.tau_sum = np.ascontiguousarray(tau_sum.T).reshape(-1)  # [z][n]
        self.nd = np.array([mr.get_dtau_ndoubl(l.tau, l.varpi, quad.qp_mu)[1] for l in layers], dtype=np.int32)
        self.albedo = float(scene.albedo)
        # surface type (Scene.brdf: None = LambertianSurfaceScalar(albedo)); the oracle's own reflectance code
        self.surf_kind, self.Rsurf, self.albedo_spec = mr.surface_inputs(scene)
        if self.Rsurf is not None:  # [M,N,N] -> [N,N,M] column-major
            self.Rsurf = np.ascontiguousarray(np.transpose(self.Rsurf, (0, 2, 1))).reshape(-1)
        self.nVza = len(scene.vza)
        self.node = np.array([mr.nearest_point(quad.qp_mu, float(mr.cosd(v))) + 1 for v in scene.vza], dtype=np.int32)
        self.vaz = np.asarray(scene.vaz, dtype=np.float64)
        cm = np.array([[float(mr.cosd(m * a)) for a in scene.vaz] for m in range(self.M)])
        sm = np.array([[float(mr.sind(m * a)) for a in scene.vaz] for m in range(self.M)])
        self.cos_mphi = np.ascontiguousarray(cm).reshape(-1)  # [m][v]
        self.sin_mphi = np.ascontiguousarray(sm).reshape(-1)

    def c_struct(self) -> OraScene:
        s = OraScene()
        s.N, s.nS, s.S, s.Nz, s.K, s.M = self.N, self.nS, self.S, self.Nz, self.K, self.M
        s.imu0, s.strict, s.mu0 = self.imu0, self.strict, self.mu0
        s.mu, s.wt, s.I0, s.D = dp(self.mu), dp(self.wt), dp(self.I0), dp(self.D)
        s.tau, s.varpi, s.zw = dp(self.tau), dp(self.varpi), dp(self.zw)
        s.Zpp, s.Zmp = dp(self.Zpp), dp(self.Zmp)
        s.nd, s.iface, s.tau_sum = ip(self.nd), ip(self.iface), dp(self.tau_sum)
        s.albedo, s.nVza, s.node = self.albedo, self.nVza, ip(self.node)
        s.cos_mphi, s.sin_mphi = dp(self.cos_mphi), dp(self.sin_mphi)
        s.surf_kind = self.surf_kind
        s.Rsurf = dp(self.Rsurf) if self.Rsurf is not None else None
        s.albedo_spec = dp(self.albedo_spec) if self.albedo_spec is not None else None
        return s


def pack_scene(scene: mr.Scene) -> Packed:
    return Packed(scene)


def rt_run(p: Packed, pts=None, nthreads: int = 0):
    """Full elastic run on the C oracle.  Returns R_SFI, T_SFI as [nVza, nStokes, S]."""
    if nthreads <= 0:
        nthreads = effective_cores()
    R = np.zeros(p.nVza * p.nS * p.S)
    T = np.zeros(p.nVza * p.nS * p.S)
    st = p.c_struct()
    if pts is None:
        info = lib().ora_rt_run(C.byref(st), None, 0, nthreads, dp(R), dp(T))
    else:
        pts = np.ascontiguousarray(pts, dtype=np.int32)
        info = lib().ora_rt_run(C.byref(st), ip(pts), len(pts), nthreads, dp(R), dp(T))
    shp = (p.S, p.nS, p.nVza)
    return np.transpose(R.reshape(shp), (2, 1, 0)).copy(), np.transpose(T.reshape(shp), (2, 1, 0)).copy(), info


def _typed_scene_struct(cfloat):
    """ora_scene of a build whose `double` is another C type (long double: momref_ext.c, float: momref_f32.c)."""
    class S(C.Structure):
        _fields_ = [("N", C.c_int), ("nS", C.c_int), ("S", C.c_int), ("Nz", C.c_int), ("K", C.c_int), ("M", C.c_int),
                    ("imu0", C.c_int), ("strict", C.c_int), ("mu0", cfloat),
                    ("mu", C.c_void_p), ("wt", C.c_void_p), ("I0", C.c_void_p), ("D", C.c_void_p),
                    ("tau", C.c_void_p), ("varpi", C.c_void_p), ("zw", C.c_void_p), ("Zpp", C.c_void_p), ("Zmp", C.c_void_p),
                    ("nd", c_ip), ("iface", c_ip), ("tau_sum", C.c_void_p), ("albedo", cfloat),
                    ("nVza", C.c_int), ("node", c_ip), ("cos_mphi", C.c_void_p), ("sin_mphi", C.c_void_p),
                    ("surf_kind", C.c_int), ("Rsurf", C.c_void_p), ("albedo_spec", C.c_void_p)]
    return S


OraSceneExt = _typed_scene_struct(C.c_longdouble)
OraSceneF32 = _typed_scene_struct(C.c_float)


def _rt_run_typed(L, Struct, npdtype, p: Packed, pts, point_threads: int):
    keep = {}

    def conv(name):
        a = getattr(p, name)
        if a is None:
            return None
        keep[name] = np.ascontiguousarray(a, dtype=npdtype)
        return keep[name].ctypes.data_as(C.c_void_p)

    s = Struct()
    s.N, s.nS, s.S, s.Nz, s.K, s.M = p.N, p.nS, p.S, p.Nz, p.K, p.M
    s.imu0, s.strict, s.mu0 = p.imu0, p.strict, p.mu0
    for nm in ("mu", "wt", "I0", "D", "tau", "varpi", "zw", "Zpp", "Zmp", "tau_sum", "cos_mphi", "sin_mphi", "Rsurf", "albedo_spec"):
        setattr(s, nm, conv(nm))
    s.nd, s.iface, s.node = ip(p.nd), ip(p.iface), ip(p.node)
    s.albedo, s.nVza, s.surf_kind = p.albedo, p.nVza, p.surf_kind
    n = p.nVza * p.nS * p.S
    R, T = np.zeros(n, dtype=npdtype), np.zeros(n, dtype=npdtype)
    pts = np.ascontiguousarray(np.arange(p.S) if pts is None else pts, dtype=np.int32)
    info = L.ora_rt_run(C.byref(s), ip(pts), len(pts), int(point_threads), R.ctypes.data_as(C.c_void_p),
                        T.ctypes.data_as(C.c_void_p))
    shp = (p.S, p.nS, p.nVza)
    return np.transpose(R.reshape(shp), (2, 1, 0)).copy(), np.transpose(T.reshape(shp), (2, 1, 0)).copy(), info


def rt_run_ext(p: Packed, pts, point_threads: int = 1):
    """The elastic run of `rt_run` in x87 EXTENDED precision (oracle/momref_ext.c) for the spectral points `pts`.  The inputs
    are the same Float64 numbers (widened exactly); R_SFI, T_SFI come back as numpy.longdouble [nVza, nStokes, S] (zeros at
    the points not asked for).  point_threads: OpenMP threads over the points (each point runs serially)."""
    return _rt_run_typed(lib_ext(), OraSceneExt, np.longdouble, p, pts, point_threads)


_LIB_F32 = None


def lib_f32():
    """oracle/libmomref_f32.so: momref.c compiled in Float32 (momref_f32.c)."""
    global _LIB_F32
    if _LIB_F32 is None:
        _LIB_F32 = C.CDLL(str(build(which="f32")))
    return _LIB_F32


def rt_run_f32(p: Packed, pts=None, point_threads: int = 0):
    """The elastic run of `rt_run` in FLOAT32 (oracle/momref_f32.c): the inputs are the Float64 numbers of `p` rounded to
    Float32 (what a Float32 model of the reference holds, and what the library's dtype = 1 handle makes of the Float64
    arrays of the ABI); R_SFI, T_SFI come back as numpy.float32 [nVza, nStokes, S]."""
    return _rt_run_typed(lib_f32(), OraSceneF32, np.float32, p, pts, point_threads or effective_cores())


def rt_run_full(p: Packed, pts=None, nthreads: int = 0):
    """Like rt_run plus the RAMI extras: returns R, T, hdr [nVza,nStokes,S], bhr_uw, bhr_dw [nStokes,S], info."""
    if nthreads <= 0:
        nthreads = effective_cores()
    n = p.nVza * p.nS * p.S
    R, T, H = np.zeros(n), np.zeros(n), np.zeros(n)
    up, dw = np.zeros(p.nS * p.S), np.zeros(p.nS * p.S)
    st = p.c_struct()
    if pts is None:
        info = lib().ora_rt_run_full(C.byref(st), None, 0, nthreads, dp(R), dp(T), dp(H), dp(up), dp(dw))
    else:
        pts = np.ascontiguousarray(pts, dtype=np.int32)
        info = lib().ora_rt_run_full(C.byref(st), ip(pts), len(pts), nthreads, dp(R), dp(T), dp(H), dp(up), dp(dw))
    shp = (p.S, p.nS, p.nVza)
    tr = lambda a: np.transpose(a.reshape(shp), (2, 1, 0)).copy()
    return tr(R), tr(T), tr(H), up.reshape(p.S, p.nS).T.copy(), dw.reshape(p.S, p.nS).T.copy(), info


def rt_run_multisensor(p: Packed, sensor_levels, pts=None, nthreads: int = 0):
    """rt_run_test_ms (rt_run_multisensor.jl): returns uwJ, dwJ [nSensors, nVza, nStokes, S], info."""
    if nthreads <= 0:
        nthreads = effective_cores()
    lv = np.ascontiguousarray(sensor_levels, dtype=np.int32)
    n = p.nVza * p.nS * p.S * len(lv)
    uw, dw = np.zeros(n), np.zeros(n)
    st = p.c_struct()
    if pts is None:
        info = lib().ora_rt_run_ms(C.byref(st), None, 0, nthreads, len(lv), ip(lv), dp(uw), dp(dw))
    else:
        pts = np.ascontiguousarray(pts, dtype=np.int32)
        info = lib().ora_rt_run_ms(C.byref(st), ip(pts), len(pts), nthreads, len(lv), ip(lv), dp(uw), dp(dw))
    shp = (len(lv), p.S, p.nS, p.nVza)
    tr = lambda a: np.transpose(a.reshape(shp), (0, 3, 2, 1)).copy()
    return tr(uw), tr(dw), info


# ---- op-level wrappers on ABI-ordered flat arrays -----------------------------------------

def elemental(p: Packed, m, nd, tau_sum, dtau, varpi, Zpp, Zmp, z_batch, S):
    N = p.N
    out = [np.zeros(N * N * S) for _ in range(4)] + [np.zeros(N * S) for _ in range(2)]
    lib().ora_elemental(N, p.nS, S, m, nd, p.imu0, dp(p.mu), dp(p.wt), dp(p.I0), dp(p.D), p.strict,
                        dp(tau_sum), dp(dtau), dp(varpi), dp(Zpp), dp(Zmp), z_batch, *[dp(o) for o in out])
    return out  # r_pm, r_mp, t_mm, t_pp, j0p, j0m


def doubling(p: Packed, nd, expk, added, S):
    return lib().ora_doubling(p.N, p.nS, S, nd, p.strict, dp(expk), *[dp(a) for a in added])


def interaction(N, S, iface, comp, added):
    """comp: [R_mp, R_pm, T_pp, T_mm, J0p, J0m]; added: [r_pm, r_mp, t_mm, t_pp, j0p, j0m]"""
    return lib().ora_interaction(N, S, iface, *[dp(a) for a in comp], *[dp(a) for a in added])


def surface_lambertian(p: Packed, m, tau_tot, S):
    N = p.N
    out = [np.zeros(N * N * S) for _ in range(4)] + [np.zeros(N * S) for _ in range(2)]
    lib().ora_surface_lambertian(N, p.nS, S, m, p.imu0, C.c_double(p.mu0), dp(p.mu), dp(p.wt), dp(p.I0),
                                 C.c_double(p.albedo), dp(tau_tot), *[dp(o) for o in out])
    return out


# ---- the same operators in FLOAT32 (oracle/momref_f32.c): Float64 numbers in, rounded to Float32; Float32 results widened ----

def _f32(a):
    return np.ascontiguousarray(np.asarray(a, dtype=np.float64).astype(np.float32))


def _fp(a):
    return a.ctypes.data_as(C.POINTER(C.c_float))


def elemental_f32(p: Packed, m, nd, tau_sum, dtau, varpi, Zpp, Zmp, z_batch, S):
    N = p.N
    out = [np.zeros(N * N * S, np.float32) for _ in range(4)] + [np.zeros(N * S, np.float32) for _ in range(2)]
    ins = [_f32(x) for x in (p.mu, p.wt, p.I0, p.D, tau_sum, dtau, varpi, Zpp, Zmp)]
    L = lib_f32()
    L.ora_elemental.restype = None
    L.ora_elemental(N, p.nS, S, m, nd, p.imu0, _fp(ins[0]), _fp(ins[1]), _fp(ins[2]), _fp(ins[3]), p.strict,
                    _fp(ins[4]), _fp(ins[5]), _fp(ins[6]), _fp(ins[7]), _fp(ins[8]), z_batch, *[_fp(o) for o in out])
    return out  # float32 arrays: r_pm, r_mp, t_mm, t_pp, j0p, j0m


def doubling_f32(p: Packed, nd, expk32, added32, S):
    """in place on the float32 arrays `expk32`, `added32` (as returned by elemental_f32)"""
    return lib_f32().ora_doubling(p.N, p.nS, S, nd, p.strict, _fp(expk32), *[_fp(a) for a in added32])


def interaction_f32(N, S, iface, comp32, added32):
    return lib_f32().ora_interaction(N, S, iface, *[_fp(a) for a in comp32], *[_fp(a) for a in added32])


def batch_inv(N, S, A):
    X = np.zeros_like(A)
    info = lib().ora_batch_inv(N, S, dp(A), dp(X))
    return X, info


def batched_mul(N, S, A, B):
    Cm = np.zeros_like(A)
    lib().ora_batched_mul(N, S, dp(A), dp(B), dp(Cm))
    return Cm


def voigt_xsec(nu, gamma_d, y, Sline, ind_start, ind_stop, grid):
    sigma = np.zeros(len(grid))
    ind_start = np.ascontiguousarray(ind_start, dtype=np.int32)
    ind_stop = np.ascontiguousarray(ind_stop, dtype=np.int32)
    lib().ora_voigt_xsec(len(nu), dp(nu), dp(gamma_d), dp(y), dp(Sline), ip(ind_start), ip(ind_stop), len(grid),
                         dp(np.ascontiguousarray(grid, dtype=np.float64)), dp(sigma))
    return sigma


# ---- rotational Raman: rt_run(::RRS) on the C oracle (momref.c ora_rt_run_rrs; twin of oracle/rrsref.py) -----------------

class OraRRS(C.Structure):
    _fields_ = [("nR", C.c_int), ("off", c_ip), ("wR", c_dp), ("ZRpp", c_dp), ("ZRmp", c_dp), ("fscatt", c_dp),
                ("rrs_strict", C.c_int), ("own_lo", C.c_int), ("own_hi", C.c_int)]


class ReferenceRaises(RuntimeError):
    """The reference's text raises a Julia exception on this path (see oracle/rrsref.py)."""


def rt_run_rrs(scene: mr.Scene, rrs, nthreads: int = 0, p: Packed = None):
    """rt_run(RS_type::RRS, model, iBand) on the C oracle.  `rrs` is an oracle/rrsref.py RRSInputs (offsets, weights, Raman
    greek coefficients, switch position, optional owned window of n1); `scene` carries the Cabannes albedo
    (scene.varpi_cabannes).  Returns R_SFI, T_SFI, ieR_SFI, ieT_SFI [nVza, nStokes, S] (inelastic spectra: owned points only)."""
    from . import rrsref as rr
    p = p or Packed(scene)
    if nthreads <= 0:
        nthreads = effective_cores()
    N, M, S = p.N, p.M, p.S
    Z = [mr.compute_Z_moments(scene.pol.n, scene.quad.qp_mu, rrs.greek_raman, m) for m in range(M)]
    ZRpp = np.ascontiguousarray(np.transpose(np.array([z[0] for z in Z]), (0, 2, 1))).reshape(-1)   # [m][j][i]
    ZRmp = np.ascontiguousarray(np.transpose(np.array([z[1] for z in Z]), (0, 2, 1))).reshape(-1)
    fs = np.ascontiguousarray(rr.fscatt_rayleigh(scene).T).reshape(-1)                              # [z][n]
    off = np.ascontiguousarray(rrs.i_l1l0, dtype=np.int32)
    wR = np.ascontiguousarray(rrs.varpi_l1l0, dtype=np.float64)
    lo, hi = rrs.owned if rrs.owned is not None else (0, S)
    st = p.c_struct()
    r = OraRRS(len(off), ip(off), dp(wR), dp(ZRpp), dp(ZRmp), dp(fs), 1 if rrs.rrs_strict_reference else 0, int(lo), int(hi))
    out = [np.zeros(p.nVza * p.nS * S) for _ in range(4)]
    L = lib()
    L.ora_rt_run_rrs.restype = C.c_int
    info = L.ora_rt_run_rrs(C.byref(st), C.byref(r), int(nthreads), *[dp(o) for o in out])
    if info == -2:
        raise ReferenceRaises("get_n0_n1: no valid index (BoundsError)")
    if info == -3:
        raise ReferenceRaises("interaction_helper!(::RRS, 00/01/10): MethodError in the reference")
    if info == -1:
        raise MemoryError("ora_rt_run_rrs: allocation failed or bad window")
    shp = (S, p.nS, p.nVza)
    return tuple(np.transpose(o.reshape(shp), (2, 1, 0)).copy() for o in out) + (info,)
