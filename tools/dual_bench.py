"""Time mom_rt_run_dual on a C2-shaped scene (N = 60, Nz = 40, M = 3) next to the value run: tools/dual_bench.py S P [reps [nStokes l_trunc [Nz]]]
(nStokes, l_trunc: another operator edge, 40 layers)."""
import sys
import time
from pathlib import Path

import numpy as np

ROOT = Path(__file__).resolve().parents[1]
sys.path.insert(0, str(ROOT))
import rtamd  # noqa: E402

S = int(sys.argv[1]) if len(sys.argv) > 1 else 2000
P = int(sys.argv[2]) if len(sys.argv) > 2 else 3
reps = int(sys.argv[3]) if len(sys.argv) > 3 else 2
m = rtamd.scenes.scene_C2(S=S) if len(sys.argv) <= 5 else rtamd.scenes.make_scene(int(sys.argv[4]), int(sys.argv[5]), int(sys.argv[6]) if len(sys.argv) > 6 else 40, S)
sc = rtamd.prepare_scene(m)
rng = np.random.default_rng(0)
L = rtamd.corert.construct_layer_inputs(m)
parts = [rtamd.ScenePartial(dτ=L.τ * rng.uniform(-1, 1, L.τ.shape), dϖ=0.1 * L.ϖ * rng.uniform(-1, 1, L.ϖ.shape),
                            dzw=L.zw * rng.uniform(-1, 1, L.zw.shape), dalbedo=1.0) for _ in range(P)]
with rtamd.corert.make_handle(m) as h:
    rtamd.corert.scene_set(h, sc)
    h.rt_run(); h.sync()
    t0 = time.perf_counter(); h.rt_run(); h.sync(); tv = time.perf_counter() - t0
    rtamd.corert.scene_set_partials(h, sc, parts)
    h.rt_run_dual(); h.sync()
    ts = []
    for _ in range(reps):
        t0 = time.perf_counter(); h.rt_run_dual(); h.sync(); ts.append(time.perf_counter() - t0)
    R, T = h.get_RT()
    dR, dT = h.get_RT_partials() if P else (None, None)
td = min(ts)
print(f"N={sc.N} S={S} P={P}: value run {tv*1e3:.1f} ms, Dual run {td*1e3:.1f} ms = {td/tv:.1f} x (ideal 1+2P = {1+2*P}); "
      f"{S/td:.0f} points/s; finite: {bool(np.isfinite(R).all() and (dR is None or np.isfinite(dR).all()))}")
