"""Phase breakdown of the workgroup-per-pair doubling kernel of the rotational-Raman path (momr_big::k_dbl_pair_wg2/3/4,
mom_rrs_wg.hpp) on the scenes of tools/bench_rrs_nt2.py.  MOM_LIBRARY must point at a library whose RRS objects were built
with -DMOMR_DIAG_STAMPS (tools/build_variant_fast.sh rrsdiag -DMOMR_DIAG_STAMPS mom_rrs.o mom_rrs_big.o).  Prints the s_memtime
deltas per code section of wave 0 of the middle workgroup, summed over all doubling launches of one run; the stamps do not
wait for outstanding loads, so a section's share includes the latency of the loads it CONSUMES.
usage: python tools/phase_stamps_rrs_wg.py [N ...]      (N in 24 30 32 42 56 60; default 32 42 60)"""
import ctypes as C
import os
import sys

ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")
sys.path.insert(0, ROOT)
os.environ["MOM_LIBRARY"] = os.environ.get("MOM_LIBRARY", os.path.join(ROOT, "scratch", "ab", "lib_rrsdiag.so"))
import numpy as np  # noqa: E402
import rtamd  # noqa: E402

rt = rtamd.corert
SCENES = {24: (4, 7, (0.0,), 2000), 30: (3, 13, (0.0, 30.0), 2000), 32: (4, 9, (0.0, 30.0), 2000), 42: (3, 21, (0.0, 30.0, 60.0), 500),
          56: (4, 21, (0.0, 30.0, 60.0), 500), 60: (3, 33, (0.0, 30.0, 60.0), 500)}
NAMES = {20: "loop head, off-grid pairs", 21: "first loads issued (+ fused elemental), wait for ieJ0+-", 23: "publish a, b, r0 (waits for the pair's blocks)",
         24: "source chain: 4 mat-vecs, 3 exchanges", 25: "6 mat-vecs, X: 2 products", 26: "barriers (13 per pair)",
         27: "publications of X, Gt0, Y, bn, Gr0, t0", 28: "Y: 1 product", 29: "iet: 2 products", 30: "Q: 2 products",
         31: "ier: 2 products, read-back of a, bn", 32: "D signs, operator stores issued"}
lib = rtamd._lib.load()
rd = lib.momr_big_diag_read
rd.argtypes = [C.POINTER(C.c_ulonglong), C.c_int]
buf = (C.c_ulonglong * 64)()
for N in [int(x) for x in sys.argv[1:]] or [32, 42, 60]:
    nS, lt, vza, S = SCENES[N]
    nR = 40
    m = rtamd.scenes.make_scene(nS, lt, 5, S, seed=3, aerosol_total=0.1, vza=vza, vaz=tuple(20.0 * i for i in range(len(vza))))
    rng = np.random.default_rng(1)
    offs = np.unique(rng.integers(-min(600, S // 2), min(600, S // 2), 2 * nR))[:nR]
    offs = offs[offs != 0]
    RS = rt.RRS(greek_raman=rt.get_greek_rayleigh(0.2), ϖ_Cabannes=0.96, ϖ_λ1λ0=np.full(len(offs), 0.04 / len(offs)), i_λ1λ0=offs,
                rrs_strict_reference=False)
    model = rt._with_cabannes(RS, m)
    sc = rtamd.prepare_scene(model)
    assert sc.N == N, (sc.N, N)
    Zr_pp, Zr_mp = rt.raman_z(RS, model)
    with rt.make_handle(model) as h:
        h.set_option(rtamd._lib.MOM_OPT_STRIP_PAD, 0)
        h.rrs_set(RS.i_λ1λ0, RS.ϖ_λ1λ0, False)
        rt.scene_set(h, sc)
        h.scene_set_rrs(np.ascontiguousarray(rt.fscatt_rayleigh(model).T), rt._abi_mats(Zr_pp), rt._abi_mats(Zr_mp))
        h.rt_run_rrs(); h.sync()
        rd(buf, 1)
        h.rt_run_rrs(); h.sync()
        rd(buf, 0)
        tk = h.rrs_timers()
    a = np.array(buf[:], dtype=np.float64)
    tot = sum(a[k] for k in NAMES)
    nt = 2 if N <= 32 else (3 if N <= 48 else 4)
    mfma = 9 * nt * nt * 4 * 64  # cycles of the matrix pipe per pair and wave: 9 products x NT^2 tiles x 4 k-steps x 64 cycles
    print(f"N = {N} ({nt} x {nt} tiles, {nt} waves per pair), S = {S}, {len(offs)} offsets; k_dbl_pair of the instrumented run: {tk['dbl_pair'][0]:.1f} ms "
          f"in {tk['dbl_pair'][1]} launches; matrix-pipe time of a pair: {mfma} cycles per wave")
    for k in sorted(NAMES, key=lambda k: -a[k]):
        if a[k] > 0:
            print(f"  {k:2d} {NAMES[k]:70s} {100 * a[k] / tot:6.2f} %   {a[k] / 1e6:9.2f} M ticks")
