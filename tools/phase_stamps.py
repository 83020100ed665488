"""Phase breakdown of the strip-chained layer kernel: run on a GPU box against a library built with
`make -C radiativetransfer.jl_amd/csrc EXTRA=-DMOM_DIAG_STAMPS` (MOM_LIBRARY selects it); prints the s_memtime deltas
per code section of the middle workgroup (ids: MOM_STAMP in csrc/)."""
import sys, os, ctypes as C
ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")
sys.path.insert(0, ROOT)
os.environ["MOM_LIBRARY"] = os.environ.get("MOM_LIBRARY", os.path.join(ROOT, "scratch", "lib_diag.so"))
import numpy as np
import rtamd
model = rtamd.scenes.scene_C2(S=10000)
sc = rtamd.prepare_scene(model)
lib = rtamd._lib.load()
rd = lib.mom_diag_read_strip15; rd.argtypes = [C.POINTER(C.c_ulonglong), C.c_int]
buf = (C.c_ulonglong * 128)()
with rtamd.corert.make_handle(model) as h:
    rtamd.corert.run_scene(h, sc)
    rd(buf, 1)
    rtamd.corert.run_scene(h, sc)
    rd(buf, 0)
a = np.array(buf[:], dtype=np.float64) * 0.01   # 100 MHz ticks -> us
units = 78.0 * 40  # units of the middle workgroup over one run (persistent grid 256: 20000/256 per layer)
names = {49: "elem: Z loads done", 57: "elem: element math", 43: "loop top (scalars)", 46: "elem: tables+barrier", 47: "elem: main loop+barrier", 48: "elem: J vectors+barrier", 40: "prologue", 41: "elem: rest", 30: "doubling total", 42: "store first", 45: "interaction total (wave0)",
         70: "dbl: rr product + barrier", 71: "dbl: chain (wave0)", 72: "dbl: barrier after chain", 73: "dbl: write-back + barrier",
         50: "int: copies issued (wave0)", 51: "int: barrier A", 52: "int: B gemm+store (wave0)", 53: "int: barrier B",
         54: "int: chain1 Horner", 55: "int: chain1 tail", 56: "int: final barrier wait (wave0)",
         91: "int4: phase A (wave4)", 92: "int4: W0", 93: "int4: gemm+barriers", 94: "int4: Horner", 95: "int4: chain2 tail", 96: "int4: final barrier"}
for k in sorted(names):
    print(f"{k:3d} {names[k]:34s} {a[k]:12.1f} us total  {a[k]/units:8.2f} us/unit")
print("sum of ids 41+30+45+42:", (a[41] + a[30] + a[45] + a[42]) / units, "us/unit;  steps:", 185 * 78)
print("per doubling step: rr %.2f chain %.2f bar %.2f wb %.2f" % tuple(a[k] / (185 * 78.0) for k in (70, 71, 72, 73)))
print("per interaction (39 layers): " + ", ".join(f"{k}:{a[k]/(39*78.0):.2f}" for k in (50, 51, 52, 53, 54, 55, 56, 91, 92, 93, 94, 95, 96)))
print("elemental per unit %.2f" % (a[41] / units))
