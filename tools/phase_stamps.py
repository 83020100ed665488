"""Phase breakdown of the strip-chained layer kernel: run on a GPU box against a library built with
`make -C radiativetransfer.jl_amd/csrc EXTRA=-DMOM_DIAG_STAMPS` (MOM_LIBRARY selects it); prints the s_memtime deltas
per code section of the middle workgroup (ids: MOM_STAMP in csrc/)."""
import sys, os, ctypes as C
ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")
sys.path.insert(0, ROOT)
os.environ["MOM_LIBRARY"] = os.environ.get("MOM_LIBRARY", os.path.join(ROOT, "scratch", "ab", "lib_diag15.so"))
import numpy as np
import rtamd
model = rtamd.scenes.scene_C2(S=10000)
sc = rtamd.prepare_scene(model)
lib = rtamd._lib.load()
rd = lib.mom_strip_diag_read15; rd.argtypes = [C.POINTER(C.c_ulonglong), C.c_int]
buf = (C.c_ulonglong * 128)()
with rtamd.corert.make_handle(model) as h:
    rtamd.corert.run_scene(h, sc)
    rd(buf, 1)
    rtamd.corert.run_scene(h, sc)
    rd(buf, 0)
a = np.array(buf[:], dtype=np.float64)
names = {49: "elem: Z loads done", 57: "elem: element math", 43: "loop top (scalars)", 46: "elem: tables+barrier", 47: "elem: main loop+barrier", 48: "elem: J vectors+barrier", 40: "prologue", 41: "elem: rest", 30: "doubling: rest", 42: "store first", 45: "interaction: rest (wave0)",
         70: "dbl: rr product + barrier", 71: "dbl: chain (wave0)", 72: "dbl: barrier after chain", 73: "dbl: write-back + barrier",
         50: "int: copies issued (wave0)", 51: "int: barrier A", 52: "int: B gemm+store (wave0)", 53: "int: barrier B",
         54: "int: chain1 Horner", 55: "int: chain1 tail", 56: "int: final barrier wait (wave0)"}
# every stamp accumulates the time since the previous stamp of the same wave: the wave-0 ids partition the kernel's run time
tot = sum(a[k] for k in names)
print("share of the dominant kernel's run time per code section (wave 0 of the middle workgroup, sweep mode, C2):")
for k in sorted(names, key=lambda k: -a[k]):
    print(f"{k:3d} {names[k]:34s} {100 * a[k] / tot:6.2f} %")
grp = {"doubling": (30, 70, 71, 72, 73), "interaction": (45, 50, 51, 52, 53, 54, 55, 56), "elemental": (41, 46, 47, 48, 49, 57), "other": (40, 42, 43)}
for g, ids in grp.items():
    print(f"{g:12s} {100 * sum(a[k] for k in ids) / tot:6.2f} %")
print(f"MFMA chains (71 + 54 + 55): {100 * (a[71] + a[54] + a[55]) / tot:6.2f} %")
