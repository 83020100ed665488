"""Phase breakdown of the GENERAL layer kernel (k_layer<false, 3, 0>: N > 64, operands in the per-workgroup global slab) on a
C4 run: MOM_LIBRARY must point at a library built with `make -C radiativetransfer.jl_amd/csrc EXTRA=-DMOM_DIAG_STAMPS`; prints
the s_memtime deltas per code section of the middle workgroup (ids: MOM_STAMP in csrc/mom_kernels.hpp, mom_entry.hpp)."""
import sys, os, ctypes as C
ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")
sys.path.insert(0, ROOT)
os.environ["MOM_LIBRARY"] = os.environ.get("MOM_LIBRARY", os.path.join(ROOT, "scratch", "ab", "lib_diag.so"))
import numpy as np
import rtamd
# usage: phase_stamps_gen.py [S]                 -> the C4 scene (N = 256)
#        phase_stamps_gen.py S nStokes l_trunc   -> a make_scene of the size sweep (20 layers; e.g. 1024 3 49 -> N = 84)
S = int(sys.argv[1]) if len(sys.argv) > 1 else 256
if len(sys.argv) > 3:
    model = rtamd.scenes.make_scene(int(sys.argv[2]), int(sys.argv[3]), 20, S, vza=(0.0, 30.0), vaz=(0.0, 20.0))
else:
    model = rtamd.scenes.scene_C4(S=S)
sc = rtamd.prepare_scene(model)
lib = rtamd._lib.load()
rd = lib.mom_diag_read_gen; rd.argtypes = [C.POINTER(C.c_ulonglong), C.c_int]
buf = (C.c_ulonglong * 128)()
with rtamd.corert.make_handle(model) as h:
    h.set_option(rtamd._lib.MOM_OPT_M0_REDUCTION, 0)   # every moment on the full problem: one kernel image
    rtamd.corert.run_scene(h, sc)
    rd(buf, 1)
    rtamd.corert.run_scene(h, sc)
    rd(buf, 0)
a = np.array(buf[:], dtype=np.float64)
names = {40: "prologue", 43: "loop top (scalars)", 46: "elem: tables+barrier", 49: "elem: Z loads", 57: "elem: element math", 47: "elem: main loop+barrier",
         48: "elem: J vectors", 41: "elem: rest", 0: "dbl: before r r", 1: "dbl: P = r r (gemm + norm)", 70: "dbl: beta2", 6: "times_inv: entry/series choice",
         2: "dbl: times_inv (series / Gauss-Jordan)", 3: "dbl: rider vectors", 4: "dbl: P = Q r (gemm)", 5: "dbl: r += P t, t = Q t (gemm2)", 30: "doubling: rest (apply D)",
         10: "int: start", 11: "int: copy R+- -> P", 12: "int: Q = r R+- (gemm + norm)", 13: "int: times_inv T01", 14: "int: J0- (no-ride)", 15: "int: T-- and T01 r (2 gemm)",
         16: "int: copy T++", 17: "int: R-+ += (T01 r) T++ (gemm)", 18: "int: copy R+-", 19: "int: P = R+- r (gemm + norm)", 20: "int: times_inv T21", 21: "int: J0+ (no-ride)",
         22: "int: copy R+- again", 23: "int: P = T21 R+- (gemm)", 24: "int: R+- = r+- + P t-- (gemm)", 25: "int: copy T++", 26: "int: T++ = T21 T++ (gemm)", 45: "interaction: rest", 42: "store first"}
tot = sum(a[k] for k in names)
print(f"N = {sc.N}, S = {S}, sum(ndoubl) = {int(sc.ndoubl.sum())}: share of k_layer<false,3,0>'s run time per code section (wave 0 of the middle workgroup)")
for k in sorted(names, key=lambda k: -a[k]):
    if a[k] > 0:
        print(f"{k:3d} {names[k]:44s} {100 * a[k] / tot:6.2f} %")
grp = {"doubling": (0, 1, 70, 6, 2, 3, 4, 5, 30), "interaction": tuple(range(10, 27)) + (45,), "elemental": (41, 46, 47, 48, 49, 57), "other": (40, 42, 43)}
for g, ids in grp.items():
    print(f"{g:12s} {100 * sum(a[k] for k in ids) / tot:6.2f} %")
