"""Phase breakdown of the quad-block image of the headline's m = 0 launch (momq::k_layer_q4<10>: the (I,Q) sub-problem of C2,
N0 = 40, one wavefront per unit, four per CU): MOM_LIBRARY must point at a library whose momcore_q4s10.o was built with
-DMOM_DIAG_STAMPS (tools/build_variant_fast.sh diagq4 -DMOM_DIAG_STAMPS momcore_q4s10.o); prints the s_memtime deltas per code
section of the middle workgroup's wave.  One wave per SIMD: the sections are that wave's own time, nothing else runs on its SIMD."""
import sys, os, ctypes as C
ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")
sys.path.insert(0, ROOT)
os.environ["MOM_LIBRARY"] = os.environ.get("MOM_LIBRARY", os.path.join(ROOT, "scratch", "ab", "lib_diagq4.so"))
import numpy as np
import rtamd
S = int(sys.argv[1]) if len(sys.argv) > 1 else 10000
model = rtamd.scenes.scene_C2(S=S)
sc = rtamd.prepare_scene(model)
lib = rtamd._lib.load()
rd = lib.momq_q4_diag_read10; rd.argtypes = [C.POINTER(C.c_ulonglong), C.c_int]
buf = (C.c_ulonglong * 128)()
with rtamd.corert.make_handle(model) as h:
    h.set_option(rtamd._lib.MOM_OPT_LEAN, 3)
    h.set_option(rtamd._lib.MOM_OPT_OVERLAP, 0)
    rtamd.corert.run_scene(h, sc)
    rd(buf, 1)
    rtamd.corert.run_scene(h, sc)
    rd(buf, 0)
    print("stages:", {k: round(v, 2) for k, v in h.timers().items() if isinstance(v, float)})
a = np.array(buf[:], dtype=np.float64)
names = {43: "loop top: layer scalars", 49: "elem: Z loads done", 57: "elem: element math", 46: "elem: tables", 47: "elem: main loop rest", 48: "elem: J vectors",
         41: "elem: rest", 42: "store first", 70: "dbl: P = r r + store", 71: "dbl: norm, w1/w2, Horner series", 72: "dbl: A r, r + t(Ar), t A, write-back",
         73: "dbl: D signs", 50: "int: R+- -> P", 52: "int: B, W0 products, B store, norm", 54: "int: Horner X", 55: "int: T-- load + Horner T01",
         56: "int: T++ -> P, J0+", 58: "int: chain 1", 59: "int: chain 2",
         74: "horner: before a product", 80: "dbl: norm reduction, series length", 81: "dbl: w1 / w2 to LDS + fence", 82: "horner: initial copy (+ what precedes it in the interaction)", 75: "horner: the product, first round of a series", 77: "horner: the product, later rounds", 76: "horner: copy of the running value"}
tot = sum(a[k] for k in names)
print(f"share of momq::k_layer_q4<10>'s time per code section (the middle workgroup, sweep mode, C2 m = 0, S = {S}); 100 MHz ticks total {tot:.0f}")
for k in sorted(names, key=lambda k: -a[k]):
    if a[k] > 0:
        print(f"{k:3d} {names[k]:44s} {100 * a[k] / tot:6.2f} %")
grp = {"doubling + interaction (horner parts counted once, in 74..76)": (70, 71, 72, 73, 50, 52, 54, 55, 56, 58, 59, 74, 75, 76, 77, 80, 81, 82), "elemental": (41, 46, 47, 48, 49, 57), "other": (42, 43)}
for g, ids in grp.items():
    print(f"{g:12s} {100 * sum(a[k] for k in ids) / tot:6.2f} %")
print(f"ticks per Horner product (s_memtime, ~2.39 GHz): first round of a series {a[75] / max(a[78], 1):.0f} ({a[78]:.0f} products), later rounds {a[77] / max(a[79], 1):.0f} ({a[79]:.0f}); 300 MFMAs x 16 cycles = 4800")
