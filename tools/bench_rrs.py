#!/usr/bin/env python3
"""C5 (rotational-Raman) timing: mom_rt_run_rrs on the resident scene_C5.  `python tools/bench_rrs.py [S] [nRaman] [steps]`."""
import sys
import time
from pathlib import Path

import numpy as np

sys.path.insert(0, str(Path(__file__).resolve().parents[1]))
import rtamd  # noqa: E402


def main():
    S = int(sys.argv[1]) if len(sys.argv) > 1 else 6837
    nR = int(sys.argv[2]) if len(sys.argv) > 2 else 0
    steps = int(sys.argv[3]) if len(sys.argv) > 3 else 3
    rt = rtamd.corert
    m, RS = rtamd.scenes.scene_C5(S=S, nRaman=nR or None)
    m = rt._with_cabannes(RS, m)
    sc = rtamd.prepare_scene(m)
    Zr_pp, Zr_mp = rt.raman_z(RS, m)
    with rt.make_handle(m) as h:
        h.set_option(rtamd._lib.MOM_OPT_STRIP_PAD, 0)
        h.rrs_set(RS.i_λ1λ0, RS.ϖ_λ1λ0, RS.rrs_strict_reference)
        rt.scene_set(h, sc)
        h.scene_set_rrs(np.ascontiguousarray(rt.fscatt_rayleigh(m).T), rt._abi_mats(Zr_pp), rt._abi_mats(Zr_mp))
        for k in range(steps + 1):
            t0 = time.perf_counter()
            h.rt_run_rrs()
            out = h.get_RT_rrs()
            dt = time.perf_counter() - t0
            print(f"run {k}: wall {dt * 1e3:.1f} ms, gpu {out[4]:.1f} ms, {S / dt:.0f} points/s, nRaman {RS.n_Raman}, "
                  f"sum nd {int(sc.ndoubl.sum())}, |ieR|max {np.abs(out[2]).max():.3e}", flush=True)


if __name__ == "__main__":
    main()
