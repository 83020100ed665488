import sys, os
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "tests"))
import numpy as np
import rtamd
os.makedirs("gpurun_out", exist_ok=True)
out = {}
for tot in (2.0, 8.0):
    m = rtamd.scenes.make_scene(3, 33, 4, 8, seed=77, aerosol_total=tot, aerosol_p0=600.0, aerosol_σp=200.0, absorption=False)
    sc = rtamd.prepare_scene(m)
    for inv in (0, 2, 1):
        with rtamd.corert.make_handle(m) as h:
            h.set_option(rtamd._lib.MOM_OPT_INVERSE, inv)
            R, T = rtamd.corert.run_scene(h, sc)
        out[f"R_{tot}_{inv}"] = R; out[f"T_{tot}_{inv}"] = T
np.savez("gpurun_out/thick.npz", **out)
