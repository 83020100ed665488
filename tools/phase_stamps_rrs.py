"""Phase breakdown of the doubling pair kernel of the rotational-Raman path (momr::k_dbl_pair1, C5: N = 15, 178 Raman lines):
MOM_LIBRARY must point at a library whose mom_rrs.o was built with -DMOMR_DIAG_STAMPS
(tools/build_variant_fast.sh rrsdiag -DMOMR_DIAG_STAMPS mom_rrs.o mom_rrs_big.o).  Prints the s_memtime deltas per code section of
wave 0 of the middle workgroup, summed over all doubling launches of one run.  Every stamp waits for the wave's outstanding
memory operations first (s_waitcnt vmcnt(0)), so a section's share includes the latency of the loads it issued or consumes --
the instrumented kernel serialises what the shipped one overlaps, and runs slower."""
import sys, os, ctypes as C
ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")
sys.path.insert(0, ROOT)
os.environ["MOM_LIBRARY"] = os.environ.get("MOM_LIBRARY", os.path.join(ROOT, "scratch", "ab", "lib_rrsdiag.so"))
import numpy as np
import rtamd
rt = rtamd.corert
S = int(sys.argv[1]) if len(sys.argv) > 1 else 6837
m, RS = rtamd.scenes.scene_C5(S=S)
m = rt._with_cabannes(RS, m)
sc = rtamd.prepare_scene(m)
Zr_pp, Zr_mp = rt.raman_z(RS, m)
lib = rtamd._lib.load()
rd = lib.momr_diag_read; rd.argtypes = [C.POINTER(C.c_ulonglong), C.c_int]
buf = (C.c_ulonglong * 64)()
with rt.make_handle(m, S=S) as h:
    h.set_option(rtamd._lib.MOM_OPT_STRIP_PAD, 0)
    h.rrs_set(RS.i_λ1λ0, RS.ϖ_λ1λ0, RS.rrs_strict_reference)
    h.rrs_set_shard(S, 0, 0, S)
    rt.scene_set(h, sc)
    h.scene_set_rrs(np.ascontiguousarray(rt.fscatt_rayleigh(m).T), rt._abi_mats(Zr_pp), rt._abi_mats(Zr_mp))
    h.rt_run_rrs(); h.sync()
    rd(buf, 1)
    h.rt_run_rrs(); h.sync()
    rd(buf, 0)
    tk = h.rrs_timers()
a = np.array(buf[:], dtype=np.float64)
names = {0: "loop overhead / off-grid pairs", 1: "first operand loads (8 tiles / vectors; + fused elemental)", 2: "transpose ier (LDS)",
         3: "X = ier r0 + r1 ier (2 products)", 4: "source vectors (4 loads, 10 mat-vecs, 4 c2r, 2 stores)", 5: "iet: 2 transposes + 3 products",
         6: "three late operand loads", 7: "ier: 4 products", 8: "D signs + operator stores",
         # dbl_pair_body1 (r5, LDS prefetch; stamps without a memory wait)
         10: "r5: loop head, off-grid pairs, item set-up", 11: "r5: wait for the prefetch (vmcnt 0)", 12: "r5: LDS reads, late loads / prefetch / stores issued",
         13: "r5: (fused elemental), transpose ier, X", 14: "r5: source vectors", 15: "r5: iet: 2 transposes, 3 products", 16: "r5: ier: 4 products (+ late operands)"}
tot = sum(a[k] for k in names)
print(f"C5, S = {S}, {RS.n_Raman} lines; timers of the instrumented run: {tk}")
print("share of the doubling pair kernels' run time per code section (wave 0 of the middle workgroup, all doubling launches of a run):")
for k in sorted(names, key=lambda k: -a[k]):
    if a[k] > 0: print(f"{k:2d} {names[k]:62s} {100 * a[k] / tot:6.2f} %   {a[k] / 1e6:10.2f} M ticks")
