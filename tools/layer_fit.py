"""Fit per-layer kernel time = a + b * ndoubl from a rocprofv3 kernel trace of `bench.py` (C2 workload).
usage: python tools/layer_fit.py <..._kernel_trace.csv>"""
import csv
import sys
from pathlib import Path

import numpy as np

sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
import rtamd  # noqa: E402

rows = list(csv.DictReader(open(sys.argv[1])))
dur = lambda r: (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e6
full = [dur(r) for r in rows if r["Kernel_Name"].startswith("void mom::k_layer")][-40:]
red = [dur(r) for r in rows if r["Kernel_Name"].startswith("void mom4::k_layer")][-40:]
sc = rtamd.prepare_scene(rtamd.scenes.scene_C2(S=10000))
nd = np.asarray(sc.ndoubl)
for name, t, units in (("full (2 moments x 1e4 points / 256 CUs)", np.array(full), 20000 / 256),
                       ("reduced m=0 (1e4 points / 512 slots)", np.array(red), 10000 / 512)):
    if len(t) != 40:
        continue
    A = np.stack([np.ones(39), nd[1:]], 1)
    c, *_ = np.linalg.lstsq(A, t[1:], rcond=None)
    print(f"{name}: first layer {t[0]:.2f} ms = {t[0] / units * 1e3:.1f} us/unit; fit a = {c[0]:.3f} ms, b = {c[1]:.3f} ms"
          f" -> per unit a = {c[0] / units * 1e3:.1f} us, b = {c[1] / units * 1e3:.1f} us per doubling; sum {t.sum():.1f} ms")
    print("   layers:", np.round(t, 2).tolist())
