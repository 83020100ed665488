"""Phase breakdown of the 4-wave strip-chained image of the headline's m = 0 launch (mom4::k_layer<true,3,10>: the (I,Q)
sub-problem of C2, N0 = 40, two workgroups per CU): MOM_LIBRARY must point at a library whose momcore_s10.o was built with
-DMOM_DIAG_STAMPS (tools/build_variant_fast.sh diag10 -DMOM_DIAG_STAMPS momcore_s10.o); prints the s_memtime deltas per code
section of wave 0 of the middle workgroup.  With two workgroups per CU these are ELAPSED shares: a section's time includes
waiting for the MFMA pipe / LDS the co-resident workgroup is using."""
import sys, os, ctypes as C
ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")
sys.path.insert(0, ROOT)
os.environ["MOM_LIBRARY"] = os.environ.get("MOM_LIBRARY", os.path.join(ROOT, "scratch", "ab", "lib_diag10.so"))
import numpy as np
import rtamd
S = int(sys.argv[1]) if len(sys.argv) > 1 else 10000
model = rtamd.scenes.scene_C2(S=S)
sc = rtamd.prepare_scene(model)
lib = rtamd._lib.load()
rd = lib.mom_strip_diag_read10; rd.argtypes = [C.POINTER(C.c_ulonglong), C.c_int]
buf = (C.c_ulonglong * 128)()
with rtamd.corert.make_handle(model) as h:
    rtamd.corert.run_scene(h, sc)
    rd(buf, 1)
    rtamd.corert.run_scene(h, sc)
    rd(buf, 0)
    print("stages:", {k: round(v, 2) for k, v in h.timers().items() if isinstance(v, float)})
a = np.array(buf[:], dtype=np.float64)
names = {49: "elem: Z loads done", 57: "elem: element math", 43: "loop top (scalars)", 46: "elem: tables+barrier", 47: "elem: main loop+barrier",
         48: "elem: J vectors+barrier", 40: "prologue", 41: "elem: rest", 30: "doubling: rest", 42: "store first", 45: "interaction: rest (wave0)",
         70: "dbl: rr strip product + barrier", 71: "dbl: chain (wave0)", 72: "dbl: barrier after chain", 73: "dbl: write-back + barrier",
         50: "int: composite loads issued+stored (wave0)", 51: "int: barrier A", 52: "int: B + W0 strip products (wave0)", 53: "int: B store, norm, 2 barriers",
         54: "int: chain1 Horner", 55: "int: chain1 tail", 58: "int: chain2 Horner", 59: "int: chain2 tail", 56: "int: final barrier wait (wave0)"}
tot = sum(a[k] for k in names)
print(f"share of mom4::k_layer<true,3,10>'s run time per code section (wave 0 of the middle workgroup, sweep mode, C2 m = 0, S = {S}):")
for k in sorted(names, key=lambda k: -a[k]):
    if a[k] > 0:
        print(f"{k:3d} {names[k]:44s} {100 * a[k] / tot:6.2f} %")
grp = {"doubling": (30, 70, 71, 72, 73), "interaction": (45, 50, 51, 52, 53, 54, 55, 56, 58, 59), "elemental": (41, 46, 47, 48, 49, 57), "other": (40, 42, 43)}
for g, ids in grp.items():
    print(f"{g:12s} {100 * sum(a[k] for k in ids) / tot:6.2f} %")
print(f"MFMA chains (70 + 71 + 52 + 54 + 55 + 58 + 59): {100 * sum(a[k] for k in (70, 71, 52, 54, 55, 58, 59)) / tot:6.2f} %")
