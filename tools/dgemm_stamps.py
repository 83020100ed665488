"""Phase stamps of momd::k_dgemm (diagnostic build -DMOMD_STAMPS, scratch/ab/lib_stamps.so): s_memtime ticks (100 MHz) per section of a
(term, K chunk) phase, wave 0 of one workgroup in the middle of every launch, summed over the Dual run.
usage: MOM_LIBRARY=scratch/ab/lib_stamps.so python tools/dgemm_stamps.py"""
import ctypes as C
import sys
from pathlib import Path

import numpy as np

sys.path.insert(0, str(Path(__file__).resolve().parents[1]))
import rtamd  # noqa: E402

S, P = 2000, 3
m = rtamd.scenes.make_scene(3, 33, 4, S)
sc = rtamd.prepare_scene(m)
L = rtamd.corert.construct_layer_inputs(m)
rng = np.random.default_rng(0)
parts = [rtamd.ScenePartial(dτ=L.τ * rng.uniform(-1, 1, L.τ.shape), dϖ=0.1 * L.ϖ, dzw=L.zw.copy(), dalbedo=1.0) for _ in range(P)]
lib = rtamd._lib.load()
rd = lib.momd_stamps_read
rd.argtypes = [C.POINTER(C.c_ulonglong), C.c_int]
buf = (C.c_ulonglong * 8)()
with rtamd.corert.make_handle(m) as h:
    rtamd.corert.scene_set(h, sc)
    rtamd.corert.scene_set_partials(h, sc, parts)
    h.rt_run_dual(); h.sync()
    rd(buf, 1)
    h.rt_run_dual(); h.sync()
    rd(buf, 0)
v = np.array(list(buf), dtype=float)
names = ["prologue", "top barrier", "wait loads + LDS stores", "barrier after stores", "issue next loads", "k-steps (LDS reads + MFMA)", "drain", ""]
tot = v[:7].sum()
for n, x in zip(names[:7], v[:7]):
    print(f"{n:32s} {x:12.0f} ticks  {100 * x / tot:5.1f} %")
