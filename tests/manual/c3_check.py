import sys, os, time
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import numpy as np
import rtamd, helpers
from oracle import cref
m = rtamd.scenes.scene_C3()
sc = rtamd.prepare_scene(m)
print("C3", sc.N, sc.Nz, sc.S, sc.M, "sum nd", int(np.sum(sc.ndoubl)))
with rtamd.corert.make_handle(m) as h:
    R, T = rtamd.corert.run_scene(h, sc)
    t0 = time.time(); h.rt_run(); h.sync(); dt = time.time() - t0
    tm = h.timers()
print("C3 points/s %.0f  step %.1f ms" % (sc.S / dt, dt * 1e3), {k: round(v, 1) for k, v in tm.items() if k.endswith("ms")})
pts = np.random.default_rng(1).choice(sc.S, 8, replace=False).astype(np.int32)
p = cref.pack_scene(helpers.oracle_scene(m))
Rr, Tr, info = cref.rt_run(p, pts=pts)
helpers.assert_stokes_close(R[:, :, pts], Rr[:, :, pts], what="C3 R")
helpers.assert_stokes_close(T[:, :, pts], Tr[:, :, pts], what="C3 T")
print("C3 parity ok on", len(pts), "sampled points; finite:", bool(np.isfinite(R).all()))
