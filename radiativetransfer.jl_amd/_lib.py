"""ctypes binding of libmomcore.so (include/momcore.h).

There is no CPU fallback: if the HIP library is missing or a call fails, a `MomError` is
raised.  The library is built in-tree (radiativetransfer.jl_amd/libmomcore.so) by
`make -C radiativetransfer.jl_amd/csrc` or `__graft_entry__.build()`.
"""
from __future__ import annotations

import ctypes as C
import os
from pathlib import Path

import numpy as np

PKG_DIR = Path(__file__).resolve().parent
# MOM_LIBRARY: another build of the same HIP library (kernel A/B experiments); never a CPU substitute
LIB_PATH = Path(os.environ["MOM_LIBRARY"]) if os.environ.get("MOM_LIBRARY") else PKG_DIR / "libmomcore.so"

c_dp = C.POINTER(C.c_double)
c_ip = C.POINTER(C.c_int)
c_h = C.c_void_p

MOM_OK, MOM_EINVAL, MOM_EHIP, MOM_ESTATE, MOM_ESINGULAR, MOM_EUNSUPPORTED = 0, -1, -2, -3, -4, -5
_CODES = {MOM_EINVAL: "MOM_EINVAL", MOM_EHIP: "MOM_EHIP", MOM_ESTATE: "MOM_ESTATE", MOM_ESINGULAR: "MOM_ESINGULAR",
          MOM_EUNSUPPORTED: "MOM_EUNSUPPORTED"}

# which-codes of mom_upload / mom_download
ADDED = dict(r_pm=0, r_mp=1, t_mm=2, t_pp=3, j0p=4, j0m=5)
COMP = dict(R_mp=6, R_pm=7, T_pp=8, T_mm=9, J0p=10, J0m=11)
SURF = dict(r_pm=12, r_mp=13, t_mm=14, t_pp=15, j0p=16, j0m=17)
# ... of mom_rrs_upload / mom_rrs_download: the codes above for the elastic fields of the RRS layers, these for the 4-D fields
IE_ADDED = dict(ier_pm=18, ier_mp=19, iet_mm=20, iet_pp=21, ieJ0p=22, ieJ0m=23)
IE_COMP = dict(ieR_mp=24, ieR_pm=25, ieT_pp=26, ieT_mm=27, ieJ0p=28, ieJ0m=29)

MOM_OPT_INVERSE, MOM_OPT_FORCE_GENERIC, MOM_OPT_M0_REDUCTION, MOM_OPT_SMALL_WG, MOM_OPT_STAGGER, MOM_OPT_SMALL_N, MOM_OPT_LAYER_SWEEP = 0, 1, 2, 3, 4, 5, 6
MOM_OPT_STRIP_PAD = 7
MOM_OPT_LEAN = 8
MOM_OPT_OVERLAP = 9
MOM_OPT_RRS_KERNELS = 10   # mask: 1 WG pairs, 2 ... for 16 < N <= 32, 4 WG points, 8 tile elemental, 16 / 32 fused elemental always / never
MOM_OPT_DUAL_WORKSPACE_MB = 11   # operator workspace of mom_rt_run_dual (0: 60 % of the free HBM)


class MomError(RuntimeError):
    def __init__(self, code, msg):
        super().__init__(f"{_CODES.get(code, code)}: {msg}")
        self.code = code


# every symbol include/momcore.h declares: name -> (restype, argtypes)
SIGNATURES = {
    "mom_create": (C.c_int, [C.POINTER(c_h), C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int]),
    "mom_destroy": (C.c_int, [c_h]),
    "mom_last_error": (C.c_char_p, [c_h]),
    "mom_last_global_error": (C.c_char_p, []),
    "mom_sync": (C.c_int, [c_h]),
    "mom_check": (C.c_int, [c_h]),
    "mom_set_streams": (C.c_int, [c_h, c_dp, c_dp, C.c_int, C.c_int, C.c_double, c_dp, c_dp, C.c_int]),
    "mom_elemental": (C.c_int, [c_h, C.c_int, C.c_int, c_dp, c_dp, c_dp, c_dp, c_dp, C.c_int]),
    "mom_doubling": (C.c_int, [c_h, C.c_int, c_dp]),
    "mom_interaction": (C.c_int, [c_h, C.c_int, C.c_int]),
    "mom_copy_added_to_composite": (C.c_int, [c_h]),
    "mom_surface_lambertian": (C.c_int, [c_h, C.c_int, C.c_double, c_dp]),
    "mom_elemental_inelastic_rrs": (C.c_int, [c_h, C.c_int, C.c_int, C.c_int, c_ip] + [c_dp] * 13),
    "mom_rrs_set": (C.c_int, [c_h, C.c_int, c_ip, c_dp, C.c_int]),
    "mom_rrs_set_shard": (C.c_int, [c_h, C.c_int, C.c_int, C.c_int, C.c_int]),
    "mom_rrs_elemental": (C.c_int, [c_h, C.c_int, C.c_int] + [c_dp] * 8),
    "mom_rrs_doubling": (C.c_int, [c_h, C.c_int, c_dp]),
    "mom_rrs_interaction": (C.c_int, [c_h, C.c_int, C.c_int]),
    "mom_rrs_copy_added_to_composite": (C.c_int, [c_h]),
    "mom_rrs_surface_lambertian": (C.c_int, [c_h, C.c_int, C.c_double, c_dp]),
    "mom_rrs_upload": (C.c_int, [c_h, C.c_int, c_dp]),
    "mom_rrs_download": (C.c_int, [c_h, C.c_int, c_dp]),
    "mom_scene_set_rrs": (C.c_int, [c_h, c_dp, c_dp, c_dp]),
    "mom_rt_run_rrs": (C.c_int, [c_h]),
    "mom_get_RT_rrs": (C.c_int, [c_h, c_dp, c_dp, c_dp, c_dp, c_dp]),
    "mom_get_hdr_rrs": (C.c_int, [c_h, c_dp, c_dp, c_dp]),
    "mom_rrs_timers": (C.c_int, [c_h, c_dp, c_ip, C.c_int]),
    "mom_batch_inv": (C.c_int, [c_h, C.c_int, C.c_int, c_dp, c_dp]),
    "mom_batched_mul": (C.c_int, [c_h, C.c_int, C.c_int, c_dp, c_dp, c_dp]),
    "mom_batched_mul_dual": (C.c_int, [c_h, C.c_int, C.c_int, C.c_int, c_dp, c_dp, c_dp, c_dp, c_dp, c_dp]),
    "mom_batch_inv_dual": (C.c_int, [c_h, C.c_int, C.c_int, C.c_int, c_dp, c_dp, c_dp, c_dp]),
    "mom_upload": (C.c_int, [c_h, C.c_int, c_dp]),
    "mom_download": (C.c_int, [c_h, C.c_int, c_dp]),
    "mom_scene_set": (C.c_int, [c_h, C.c_int, C.c_int, C.c_int, c_dp, c_dp, c_dp, c_dp, c_dp, c_ip, c_ip, c_dp,
                                C.c_double, C.c_int, c_ip, c_dp, c_dp]),
    "mom_absorption_begin": (C.c_int, [c_h, C.c_int, c_dp]),
    "mom_absorption_set_lines": (C.c_int, [c_h, C.c_int] + [c_dp] * 8 + [c_ip, C.c_int, C.c_int, c_ip, c_dp, c_dp, c_dp]),
    "mom_voigt_tau_abs_layer": (C.c_int, [c_h, C.c_int, C.c_double, C.c_double, C.c_double, C.c_double, C.c_double]),
    "mom_voigt_tau_abs_profile": (C.c_int, [c_h, C.c_int, c_dp, c_dp, C.c_double, C.c_double, c_dp, c_dp]),
    "mom_absorption_get_prefactors": (C.c_int, [c_h, C.c_int, c_dp, c_dp, c_dp, c_dp, c_ip, c_ip]),
    "mom_voigt_tau_abs": (C.c_int, [c_h, C.c_int, C.c_int, c_dp, c_dp, c_dp, c_dp, c_ip, c_ip, C.c_double]),
    "mom_absorption_set": (C.c_int, [c_h, C.c_int, c_dp]),
    "mom_absorption_get": (C.c_int, [c_h, c_dp]),
    "mom_scene_set_optics": (C.c_int, [c_h, C.c_int, C.c_int, C.c_int, c_dp, C.c_double, c_dp, c_dp, c_dp, c_dp, c_dp,
                                       C.c_double, C.c_int, c_ip, c_dp, c_dp]),
    "mom_scene_get_layers": (C.c_int, [c_h, c_ip, c_ip, c_dp, c_dp, c_dp, c_dp]),
    "mom_scene_set_surface": (C.c_int, [c_h, C.c_int, C.c_int, c_dp, c_dp]),
    "mom_rt_run": (C.c_int, [c_h]),
    "mom_scene_set_partials": (C.c_int, [c_h, C.c_int] + [c_dp] * 8),
    "mom_rt_run_dual": (C.c_int, [c_h]),
    "mom_get_RT_partials": (C.c_int, [c_h, c_dp, c_dp]),
    "mom_get_hdr_partials": (C.c_int, [c_h, c_dp, c_dp, c_dp]),
    "mom_rt_run_multisensor": (C.c_int, [c_h, C.c_int, c_ip, c_dp, c_dp]),
    "mom_get_RT": (C.c_int, [c_h, c_dp, c_dp]),
    "mom_get_hdr": (C.c_int, [c_h, c_dp, c_dp, c_dp]),
    "mom_get_RT_device": (C.c_int, [c_h, C.c_void_p, C.c_void_p]),
    "mom_postprocess": (C.c_int, [c_h, C.c_int, C.c_int, c_ip, c_dp, C.c_double, c_dp, c_dp]),
    "mom_comm_unique_id": (C.c_int, [C.c_void_p, C.c_size_t]),
    "mom_comm_init": (C.c_int, [c_h, C.c_int, C.c_int, C.c_void_p]),
    "mom_comm_destroy": (C.c_int, [c_h]),
    "mom_allgather": (C.c_int, [c_h, C.c_void_p, C.c_void_p, C.c_size_t]),
    "mom_rrs_check_padding": (C.c_int, [c_h, C.POINTER(C.c_ulonglong)]),
    "mom_rrs_spectra_count": (C.c_size_t, [c_h, C.c_int]),
    "mom_get_spectra_rrs_device": (C.c_int, [c_h, C.c_int, C.c_void_p]),
    "mom_allgather_rrs_device": (C.c_int, [c_h, C.c_int, C.c_void_p]),
    "mom_allgather_RT_device": (C.c_int, [c_h, C.c_void_p]),
    "mom_allgather_RT": (C.c_int, [c_h, c_dp, c_dp]),
    "mom_timers": (C.c_int, [c_h, c_dp, C.c_int, c_ip]),
    "mom_set_option": (C.c_int, [c_h, C.c_int, C.c_int]),
    "mom_voigt_xsec": (C.c_int, [C.c_int, C.c_int, c_dp, c_dp, c_dp, c_dp, c_ip, c_ip, C.c_int, c_dp, c_dp]),
    "mom_voigt_last_kernel_ms": (C.c_double, []),
}

_lib = None


def load():
    """dlopen libmomcore.so and bind every declared symbol.  Fails loudly."""
    global _lib
    if _lib is not None:
        return _lib
    if not LIB_PATH.exists():
        raise MomError(MOM_EHIP, f"{LIB_PATH} not found - build it with `make -C {PKG_DIR / 'csrc'}` "
                                 "(hipcc --offload-arch=gfx950); there is no CPU fallback")
    lib = C.CDLL(str(LIB_PATH))
    for name, (res, args) in SIGNATURES.items():
        fn = getattr(lib, name)  # AttributeError if the symbol is missing
        fn.restype = res
        fn.argtypes = args
    _lib = lib
    return lib


def f64(a):
    return np.ascontiguousarray(a, dtype=np.float64)


def i32(a):
    return np.ascontiguousarray(a, dtype=np.int32)


def dp(a: np.ndarray):
    assert a.dtype == np.float64 and a.flags["C_CONTIGUOUS"]
    return a.ctypes.data_as(c_dp)


def ip(a: np.ndarray):
    assert a.dtype == np.int32 and a.flags["C_CONTIGUOUS"]
    return a.ctypes.data_as(c_ip)


class Handle:
    """RAII wrapper of mom_t*.  One handle <-> one GPU <-> one stream."""

    def __init__(self, N: int, nStokes: int, S: int, max_m: int = 1, device: int = 0, dtype: int = 0):
        """dtype 0 = Float64 (everything), 1 = Float32 (scene-level path, operator-level API, batched operators)."""
        self.lib = load()
        self.N, self.nS, self.S, self.M = int(N), int(nStokes), int(S), int(max_m)
        self._h = c_h()
        rc = self.lib.mom_create(C.byref(self._h), device, self.N, self.nS, self.S, self.M, int(dtype))
        if rc != MOM_OK:
            msg = self.lib.mom_last_global_error().decode()
            if self._h:
                self.lib.mom_destroy(self._h)
                self._h = c_h()
            raise MomError(rc, msg)

    def check(self, rc):
        if rc != MOM_OK:
            raise MomError(rc, self.lib.mom_last_error(self._h).decode())

    def close(self):
        if getattr(self, "_h", None):
            self.lib.mom_destroy(self._h)
            self._h = c_h()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def __enter__(self):
        return self

    def __exit__(self, *a):
        self.close()

    # -- thin 1:1 wrappers ---------------------------------------------------------------
    def set_option(self, option, value):
        self.check(self.lib.mom_set_option(self._h, option, value))

    def sync(self):
        self.check(self.lib.mom_sync(self._h))

    def check_async(self):
        self.check(self.lib.mom_check(self._h))

    def set_streams(self, qp_muN, wt_muN, imu0_1based, mu0, I0, D, strict=True):
        qp, wt, I0, D = f64(qp_muN), f64(wt_muN), f64(I0), f64(D)
        self.check(self.lib.mom_set_streams(self._h, dp(qp), dp(wt), len(qp), int(imu0_1based), float(mu0), dp(I0),
                                            dp(D), 1 if strict else 0))

    def elemental(self, m, nd, tau_sum, dtau, varpi, Zpp, Zmp, z_batch):
        a = [f64(x) for x in (tau_sum, dtau, varpi, Zpp, Zmp)]
        self.check(self.lib.mom_elemental(self._h, int(m), int(nd), *[dp(x) for x in a], int(z_batch)))

    def doubling(self, nd, expk):
        e = f64(expk).copy()
        self.check(self.lib.mom_doubling(self._h, int(nd), dp(e)))
        return e

    def interaction(self, iface, with_surface_layer=False):
        self.check(self.lib.mom_interaction(self._h, int(iface), 1 if with_surface_layer else 0))

    def copy_added_to_composite(self):
        self.check(self.lib.mom_copy_added_to_composite(self._h))

    def surface_lambertian(self, m, albedo, tau_tot):
        t = f64(tau_tot)
        self.check(self.lib.mom_surface_lambertian(self._h, int(m), float(albedo), dp(t)))

    def upload(self, which, src):
        s = f64(src).reshape(-1)
        self.check(self.lib.mom_upload(self._h, int(which), dp(s)))

    def download(self, which):
        k = which % 6
        out = np.empty((self.N * self.N if k < 4 else self.N) * self.S)
        self.check(self.lib.mom_download(self._h, int(which), dp(out)))
        return out

    def elemental_inelastic_rrs(self, m, ndoubl, i_l1l0, varpi_l1l0, fscatt, tau_sum, dtau, varpi, Zpp, Zmp):
        """elemental_inelastic!(::RRS) (elemental_inelastic.jl:23-91).  Z*: ABI [N,N] flat; returns the six arrays in ABI
        order: ier_mp, iet_pp, ier_pm, iet_mm [N,N,S,nRaman] and ieJ0p, ieJ0m [N,S,nRaman] (flat, column-major)."""
        il = i32(i_l1l0)
        nR = len(il)
        v = [f64(x).reshape(-1) for x in (varpi_l1l0, fscatt, tau_sum, dtau, varpi, Zpp, Zmp)]
        big, vec = self.N * self.N * self.S * nR, self.N * self.S * nR
        out = [np.empty(big) for _ in range(4)] + [np.empty(vec) for _ in range(2)]
        self.check(self.lib.mom_elemental_inelastic_rrs(self._h, int(m), int(ndoubl), nR, ip(il), *[dp(x) for x in v],
                                                        *[dp(x) for x in out]))
        return out

    # -- rotational-Raman path (RS_type::RRS) -------------------------------------------------------
    def rrs_set(self, i_l1l0, varpi_l1l0, rrs_strict_reference=True):
        """The RRS fields the path reads (src/Inelastic/types.jl:13-33); allocates AddedLayerRS / CompositeLayerRS."""
        il, vp = i32(i_l1l0), f64(varpi_l1l0)
        assert il.size == vp.size
        self.nRaman = int(il.size)
        self.check(self.lib.mom_rrs_set(self._h, self.nRaman, ip(il), dp(vp), 1 if rrs_strict_reference else 0))

    def rrs_set_shard(self, nSpec_global, n_glob0, n1_lo, n1_hi):
        """The handle's points are the window [n_glob0, n_glob0 + S) of a global axis; this rank owns [n1_lo, n1_hi)."""
        self.check(self.lib.mom_rrs_set_shard(self._h, int(nSpec_global), int(n_glob0), int(n1_lo), int(n1_hi)))

    def rrs_elemental(self, m, nd, tau_sum, dtau, varpi, Zpp, Zmp, fscatt, Zpp_l1l0, Zmp_l1l0):
        a = [f64(x).reshape(-1) for x in (tau_sum, dtau, varpi, Zpp, Zmp, fscatt, Zpp_l1l0, Zmp_l1l0)]
        self.check(self.lib.mom_rrs_elemental(self._h, int(m), int(nd), *[dp(x) for x in a]))

    def rrs_doubling(self, nd, expk):
        e = f64(expk).copy()
        self.check(self.lib.mom_rrs_doubling(self._h, int(nd), dp(e)))
        return e

    def rrs_interaction(self, iface, with_surface_layer=False):
        self.check(self.lib.mom_rrs_interaction(self._h, int(iface), 1 if with_surface_layer else 0))

    def rrs_copy_added_to_composite(self):
        self.check(self.lib.mom_rrs_copy_added_to_composite(self._h))

    def rrs_surface_lambertian(self, m, albedo, tau_tot):
        t = f64(tau_tot)
        self.check(self.lib.mom_rrs_surface_lambertian(self._h, int(m), float(albedo), dp(t)))

    def rrs_upload(self, which, src):
        s = f64(src).reshape(-1)
        assert s.size == self._rrs_count(which)
        self.check(self.lib.mom_rrs_upload(self._h, int(which), dp(s)))

    def _rrs_count(self, which):
        return (self.N * self.N if which % 6 < 4 else self.N) * self.S * (self.nRaman if which >= 18 else 1)

    def rrs_download(self, which):
        out = np.empty(self._rrs_count(which))
        self.check(self.lib.mom_rrs_download(self._h, int(which), dp(out)))
        return out

    def scene_set_rrs(self, fscatt, Zpp_l1l0, Zmp_l1l0):
        a = [f64(x).reshape(-1) for x in (fscatt, Zpp_l1l0, Zmp_l1l0)]
        self.check(self.lib.mom_scene_set_rrs(self._h, *[dp(x) for x in a]))

    def rt_run_rrs(self):
        self.check(self.lib.mom_rt_run_rrs(self._h))

    def get_RT_rrs(self):
        """(R_SFI, T_SFI, ieR_SFI, ieT_SFI), each [nVza, nStokes, S], and the GPU time of the run in ms."""
        out = [np.empty(self.nVza * self.nS * self.S) for _ in range(4)]
        ms = np.zeros(1)
        self.check(self.lib.mom_get_RT_rrs(self._h, *[dp(x) for x in out], dp(ms)))
        sh = (self.S, self.nS, self.nVza)
        return tuple(np.transpose(x.reshape(sh), (2, 1, 0)).copy() for x in out) + (float(ms[0]),)

    def get_hdr_rrs(self):
        """hdr [nVza, nStokes, S], bhr_uw, bhr_dw [nStokes, S] of the last rt_run_rrs."""
        n = self.nVza * self.nS * self.S
        H, up, dw = np.empty(n), np.empty(self.nS * self.S), np.empty(self.nS * self.S)
        self.check(self.lib.mom_get_hdr_rrs(self._h, dp(H), dp(up), dp(dw)))
        return (np.transpose(H.reshape(self.S, self.nS, self.nVza), (2, 1, 0)).copy(), up.reshape(self.S, self.nS).T.copy(),
                dw.reshape(self.S, self.nS).T.copy())

    def rrs_check_padding(self) -> int:
        """Nonzero entries in the zero padding of the RRS layer arrays (0 = the whole-tile stores kept the invariant)."""
        v = C.c_ulonglong(0)
        self.check(self.lib.mom_rrs_check_padding(self._h, C.byref(v)))
        return int(v.value)

    def rrs_spectra_count(self, per: int) -> int:
        return int(self.lib.mom_rrs_spectra_count(self._h, int(per)))

    def get_spectra_rrs_device(self, per: int, d_local_ptr: int):
        """Owned slices of the seven spectra packed into a device buffer of rrs_spectra_count(per) doubles (asynchronous)."""
        self.check(self.lib.mom_get_spectra_rrs_device(self._h, int(per), C.c_void_p(d_local_ptr)))

    def allgather_rrs_device(self, per: int, d_global_ptr: int):
        """One RCCL all-gather of every rank's packed owned spectra into [nranks][rrs_spectra_count(per)] (asynchronous)."""
        self.check(self.lib.mom_allgather_rrs_device(self._h, int(per), C.c_void_p(d_global_ptr)))

    def rrs_timers(self):
        """{kernel: (ms, launches)} of the last rt_run_rrs (HIP events on the library's stream)."""
        ms, nl = np.zeros(4), np.zeros(4, dtype=np.int32)
        self.check(self.lib.mom_rrs_timers(self._h, dp(ms), ip(nl), 4))
        return {k: (float(ms[i]), int(nl[i])) for i, k in enumerate(("dbl_pair", "int_pair", "ie_elemental", "total"))}

    def batch_inv(self, n, batch, A):
        A = f64(A).reshape(-1)
        X = np.empty_like(A)
        self.check(self.lib.mom_batch_inv(self._h, n, batch, dp(A), dp(X)))
        return X

    def batched_mul(self, n, batch, A, B):
        A, B = f64(A).reshape(-1), f64(B).reshape(-1)
        Cm = np.empty_like(A)
        self.check(self.lib.mom_batched_mul(self._h, n, batch, dp(A), dp(B), dp(Cm)))
        return Cm

    def batch_inv_dual(self, n, batch, P, A, dA):
        """Dual batch_inv! (gpu_batched.jl:129-150): flat ABI arrays, values n*n*batch, partials n*n*batch*P."""
        A, dA = f64(A).reshape(-1), f64(dA).reshape(-1)
        X, dX = np.empty_like(A), np.empty_like(dA)
        self.check(self.lib.mom_batch_inv_dual(self._h, n, batch, P, dp(A), dp(dA), dp(X), dp(dX)))
        return X, dX

    def batched_mul_dual(self, n, batch, P, A, dA, B, dB):
        """Dual batched_mul (gpu_batched.jl:100-110)."""
        A, dA, B, dB = (f64(x).reshape(-1) for x in (A, dA, B, dB))
        Cm, dC = np.empty_like(A), np.empty_like(dA)
        self.check(self.lib.mom_batched_mul_dual(self._h, n, batch, P, dp(A), dp(dA), dp(B), dp(dB), dp(Cm), dp(dC)))
        return Cm, dC

    def scene_set(self, Nz, K, M, tau, varpi, zw, Zpp, Zmp, nd, iface, tau_sum, albedo, node, cos_mphi, sin_mphi):
        d = [f64(x).reshape(-1) for x in (tau, varpi, zw, Zpp, Zmp)]
        nd, iface, node = i32(nd), i32(iface), i32(node)
        ts, cm, sm = f64(tau_sum).reshape(-1), f64(cos_mphi).reshape(-1), f64(sin_mphi).reshape(-1)
        self.nVza = len(node)
        self.check(self.lib.mom_scene_set(self._h, int(Nz), int(K), int(M), *[dp(x) for x in d], ip(nd), ip(iface),
                                          dp(ts), float(albedo), len(node), ip(node), dp(cm), dp(sm)))

    # -- device-side layer optics --------------------------------------------------------------
    def absorption_begin(self, Nz, grid=None):
        g = f64(grid) if grid is not None else None
        assert g is None or g.size == self.S
        self._absNz = int(Nz)
        self.check(self.lib.mom_absorption_begin(self._h, int(Nz), dp(g) if g is not None else None))

    def absorption_set(self, tau_abs):
        """tau_abs: [S, Nz] numpy (reference layout)."""
        t = np.ascontiguousarray(np.asarray(tau_abs, dtype=np.float64).T).reshape(-1)  # column-major [S, Nz]
        self._absNz = int(np.asarray(tau_abs).shape[1])
        self.check(self.lib.mom_absorption_set(self._h, self._absNz, dp(t)))

    def absorption_get(self):
        out = np.empty(self.S * self._absNz)
        self.check(self.lib.mom_absorption_get(self._h, dp(out)))
        return out.reshape(self._absNz, self.S).T.copy()

    def voigt_tau_abs(self, iz_1based, nu, gamma_d, y, S, ind_start, ind_stop, factor):
        a = [f64(x) for x in (nu, gamma_d, y, S)]
        i0, i1 = i32(ind_start), i32(ind_stop)
        self.check(self.lib.mom_voigt_tau_abs(self._h, int(iz_1based), len(a[0]), *[dp(x) for x in a], ip(i0), ip(i1),
                                              float(factor)))

    def absorption_set_lines(self, cols, sqrt_w, iso_index, nT, tips_T, tips_Q, tips_z):
        """cols: the eight per-line arrays nu0, S0, gamma_air, gamma_self, E_lower, n_air, delta_air (+ sqrt_w appended here);
        tips_*: [nIso, nTmax] numpy."""
        a = [f64(x) for x in cols] + [f64(sqrt_w)]
        ii, nT = i32(iso_index), i32(nT)
        tt, tq, tz = (f64(x) for x in (tips_T, tips_Q, tips_z))
        nIso, nTmax = (tt.shape if tt.ndim == 2 else (0, 0))
        self._nLines = len(a[0])
        self.check(self.lib.mom_absorption_set_lines(self._h, self._nLines, *[dp(x) for x in a], ip(ii), int(nIso), int(nTmax),
                                                     ip(nT), dp(tt.reshape(-1)), dp(tq.reshape(-1)), dp(tz.reshape(-1))))

    def voigt_tau_abs_layer(self, iz_1based, p, T, vmr, wing_cutoff, factor):
        self.check(self.lib.mom_voigt_tau_abs_layer(self._h, int(iz_1based), float(p), float(T), float(vmr), float(wing_cutoff),
                                                    float(factor)))

    def voigt_tau_abs_profile(self, p, T, vmr, wing_cutoff, factor) -> float:
        """All layers 1..len(p) in two launches (mom_voigt_tau_abs_profile); returns the GPU time of the two kernels in ms."""
        p, T, f = f64(p), f64(T), f64(factor)
        assert p.size == T.size == f.size
        ms = np.zeros(1)
        self.check(self.lib.mom_voigt_tau_abs_profile(self._h, int(p.size), dp(p), dp(T), float(vmr), float(wing_cutoff), dp(f), dp(ms)))
        return float(ms[0])

    def absorption_get_prefactors(self, n=None):
        n = self._nLines if n is None else int(n)
        d = [np.empty(n) for _ in range(4)]
        i0, i1 = np.empty(n, dtype=np.int32), np.empty(n, dtype=np.int32)
        self.check(self.lib.mom_absorption_get_prefactors(self._h, n, *[dp(x) for x in d], ip(i0), ip(i1)))
        return d + [i0, i1]

    def scene_set_optics(self, Nz, M, tau_rayl, varpi_rayl, tau_aer, omega_aer, ft_aer, Zpp, Zmp, albedo, node, cos_mphi,
                         sin_mphi):
        """tau_rayl [S, Nz], tau_aer [nAer, Nz] numpy (reference layouts); Zpp/Zmp already in ABI order."""
        tr = np.ascontiguousarray(np.asarray(tau_rayl, dtype=np.float64).T).reshape(-1)
        ta = np.asarray(tau_aer, dtype=np.float64)
        nAer = ta.shape[0] if ta.size else 0
        taf = np.ascontiguousarray(ta.T).reshape(-1) if nAer else np.zeros(1)  # [nAer, Nz] column-major
        om, ft = (f64(omega_aer), f64(ft_aer)) if nAer else (np.zeros(1), np.zeros(1))
        zp, zm = f64(Zpp).reshape(-1), f64(Zmp).reshape(-1)
        node = i32(node)
        cm, sm = f64(cos_mphi).reshape(-1), f64(sin_mphi).reshape(-1)
        self.nVza = len(node)
        self._Nz, self._K = int(Nz), 1 + nAer
        self.check(self.lib.mom_scene_set_optics(self._h, int(Nz), nAer, int(M), dp(tr), float(varpi_rayl), dp(taf), dp(om),
                                                 dp(ft), dp(zp), dp(zm), float(albedo), len(node), ip(node), dp(cm), dp(sm)))

    def scene_get_layers(self, Nz, K, arrays=True):
        nd, iface = np.zeros(Nz, dtype=np.int32), np.zeros(Nz, dtype=np.int32)
        S = self.S
        bufs = [np.empty(S * Nz), np.empty(S * Nz), np.empty(K * S * Nz), np.empty(S * (Nz + 1))] if arrays else [None] * 4
        self.check(self.lib.mom_scene_get_layers(self._h, ip(nd), ip(iface), *[dp(b) if b is not None else None for b in bufs]))
        return (nd, iface) + tuple(bufs)

    def scene_set_surface(self, kind, M=0, Rsurf=None, albedo_spec=None):
        """Rsurf: already in ABI order ([N, N, M], i fastest); albedo_spec: [S]."""
        r = f64(Rsurf).reshape(-1) if Rsurf is not None else None
        al = f64(albedo_spec).reshape(-1) if albedo_spec is not None else None
        self.check(self.lib.mom_scene_set_surface(self._h, int(kind), int(M), dp(r) if r is not None else None,
                                                  dp(al) if al is not None else None))

    def rt_run(self):
        self.check(self.lib.mom_rt_run(self._h))

    # -- rt_run on ForwardDiff.Dual numbers (mom_dual.hip) ------------------------------------------
    def scene_set_partials(self, P, dtau=None, dvarpi=None, dzw=None, dZpp=None, dZmp=None, dalbedo=None, dRsurf=None,
                           dalbedo_spec=None):
        """Partials of the scene's inputs, flat buffers in ABI order (partial index slowest); None = no dependence."""
        bufs = [None if x is None else f64(x).reshape(-1) for x in (dtau, dvarpi, dzw, dZpp, dZmp, dalbedo, dRsurf, dalbedo_spec)]
        self.dual_P = int(P)
        self.check(self.lib.mom_scene_set_partials(self._h, int(P), *[None if b is None else dp(b) for b in bufs]))

    def rt_run_dual(self):
        self.check(self.lib.mom_rt_run_dual(self._h))

    def get_RT_partials(self):
        """dR_SFI, dT_SFI as numpy [P, nVza, nStokes, S]."""
        P = self.dual_P
        n = self.nVza * self.nS * self.S * P
        dR, dT = np.empty(n), np.empty(n)
        self.check(self.lib.mom_get_RT_partials(self._h, dp(dR), dp(dT)))
        shp = (P, self.S, self.nS, self.nVza)
        return np.transpose(dR.reshape(shp), (0, 3, 2, 1)).copy(), np.transpose(dT.reshape(shp), (0, 3, 2, 1)).copy()

    def get_hdr_partials(self):
        """dhdr [P, nVza, nStokes, S], dbhr_uw, dbhr_dw [P, nStokes, S]."""
        P = self.dual_P
        n, f = self.nVza * self.nS * self.S * P, self.nS * self.S * P
        dH, up, dw = np.empty(n), np.empty(f), np.empty(f)
        self.check(self.lib.mom_get_hdr_partials(self._h, dp(dH), dp(up), dp(dw)))
        tr = lambda x: np.transpose(x.reshape(P, self.S, self.nS), (0, 2, 1)).copy()
        return np.transpose(dH.reshape(P, self.S, self.nS, self.nVza), (0, 3, 2, 1)).copy(), tr(up), tr(dw)

    def rt_run_multisensor(self, sensor_levels):
        """uwJ, dwJ as numpy [nSensors, nVza, nStokes, S] (the reference's vector of [nVza, nStokes, nSpec] arrays,
        rt_run_multisensor.jl:52-55)."""
        lv = np.ascontiguousarray(sensor_levels, dtype=np.int32)
        n = self.nVza * self.nS * self.S * len(lv)
        uw, dw = np.empty(n), np.empty(n)
        self.check(self.lib.mom_rt_run_multisensor(self._h, len(lv), ip(lv), dp(uw), dp(dw)))
        shp = (len(lv), self.S, self.nS, self.nVza)
        tr = lambda a: np.transpose(a.reshape(shp), (0, 3, 2, 1)).copy()
        return tr(uw), tr(dw)

    def get_RT(self):
        """R_SFI, T_SFI as numpy [nVza, nStokes, S] (reference layout, rt_run.jl:89-90)."""
        n = self.nVza * self.nS * self.S
        R, T = np.empty(n), np.empty(n)
        self.check(self.lib.mom_get_RT(self._h, dp(R), dp(T)))
        shp = (self.S, self.nS, self.nVza)
        return np.transpose(R.reshape(shp), (2, 1, 0)).copy(), np.transpose(T.reshape(shp), (2, 1, 0)).copy()

    def get_hdr(self):
        """hdr [nVza, nStokes, S], bhr_uw, bhr_dw [nStokes, S]."""
        n = self.nVza * self.nS * self.S
        H, up, dw = np.empty(n), np.empty(self.nS * self.S), np.empty(self.nS * self.S)
        self.check(self.lib.mom_get_hdr(self._h, dp(H), dp(up), dp(dw)))
        return (np.transpose(H.reshape(self.S, self.nS, self.nVza), (2, 1, 0)).copy(),
                up.reshape(self.S, self.nS).T.copy(), dw.reshape(self.S, self.nS).T.copy())

    def get_RT_device(self, dR_ptr: int, dT_ptr: int):
        self.check(self.lib.mom_get_RT_device(self._h, C.c_void_p(dR_ptr), C.c_void_p(dT_ptr)))

    def postprocess(self, m, node_1based, vaz_deg, weight, R_SFI, T_SFI):
        """In-place accumulation into R_SFI/T_SFI given as ABI-ordered flat arrays [nVza, nStokes, S] (v fastest)."""
        node, vaz = i32(node_1based), f64(vaz_deg)
        self.check(self.lib.mom_postprocess(self._h, int(m), len(node), ip(node), dp(vaz), float(weight), dp(R_SFI),
                                            dp(T_SFI)))

    # -- multi-GPU (RCCL behind the C ABI) -------------------------------------------------
    def comm_init(self, rank: int, nranks: int, nccl_id: bytes):
        buf = C.create_string_buffer(bytes(nccl_id), COMM_ID_BYTES)
        self.check(self.lib.mom_comm_init(self._h, int(rank), int(nranks), C.cast(buf, C.c_void_p)))
        self.comm_size = int(nranks)

    def comm_destroy(self):
        self.check(self.lib.mom_comm_destroy(self._h))

    def allgather_RT_device(self, d_global_ptr: int):
        self.check(self.lib.mom_allgather_RT_device(self._h, C.c_void_p(d_global_ptr)))

    def allgather_RT(self):
        """R_SFI, T_SFI [nVza, nStokes, nranks*S_loc] on every rank (one RCCL all-gather)."""
        n = self.nVza * self.nS * self.S * self.comm_size
        R, T = np.empty(n), np.empty(n)
        self.check(self.lib.mom_allgather_RT(self._h, dp(R), dp(T)))
        shp = (self.S * self.comm_size, self.nS, self.nVza)
        return np.transpose(R.reshape(shp), (2, 1, 0)).copy(), np.transpose(T.reshape(shp), (2, 1, 0)).copy()

    def timers(self):
        ms = np.zeros(8)
        nl = C.c_int(0)
        self.check(self.lib.mom_timers(self._h, dp(ms), 8, C.byref(nl)))
        return dict(layers_ms=ms[0], surface_ms=ms[1], postprocess_ms=ms[2], total_ms=ms[3], layer_launches=nl.value,
                    full_layers_ms=ms[4], reduced_layers_ms=ms[5], full_launches=int(ms[6]), reduced_launches=int(ms[7]))


COMM_ID_BYTES = 128


def comm_unique_id() -> bytes:
    """RCCL unique id (rank 0 calls this and distributes the bytes)."""
    buf = C.create_string_buffer(COMM_ID_BYTES)
    rc = load().mom_comm_unique_id(C.cast(buf, C.c_void_p), COMM_ID_BYTES)
    if rc != MOM_OK:
        raise MomError(rc, load().mom_last_global_error().decode())
    return buf.raw


def voigt_xsec(nu, gamma_d, y, S, ind_start, ind_stop, grid, device: int = 0):
    lib = load()
    a = [f64(x) for x in (nu, gamma_d, y, S)]
    i0, i1, g = i32(ind_start), i32(ind_stop), f64(grid)
    sigma = np.empty(len(g))
    rc = lib.mom_voigt_xsec(device, len(a[0]), *[dp(x) for x in a], ip(i0), ip(i1), len(g), dp(g), dp(sigma))
    if rc != MOM_OK:
        raise MomError(rc, lib.mom_last_global_error().decode())
    return sigma


def voigt_last_kernel_ms() -> float:
    return float(load().mom_voigt_last_kernel_ms())
