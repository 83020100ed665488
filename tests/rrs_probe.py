"""Child process of tests/test_gpu_rrs.py::test_rrs_workgroup_and_wave_kernels_agree: one seeded rt_run(::RRS) through the C ABI,
outputs to an .npz.  The choice between the workgroup-per-pair and the wave-per-pair RRS kernels (mom_rrs.hip wg_nt) is read
from the environment once per process, so the two runs of the comparison need a process each.
usage: python tests/rrs_probe.py nS lt S Nz strict out.npz"""
import sys
from pathlib import Path

import numpy as np

sys.path.insert(0, str(Path(__file__).resolve().parents[1]))
import rtamd  # noqa: E402

nS, lt, S, Nz, strict = (int(x) for x in sys.argv[1:6])
rt = rtamd.corert
m = rtamd.scenes.make_scene(nS, lt, Nz, S, seed=nS + lt + S, aerosol_total=0.1)
offs = np.asarray([-4, -1, 2, 7, 3])
vp = 0.04 / len(offs) * (1.0 + 0.1 * np.arange(len(offs)))
RS = rt.RRS(greek_raman=rt.get_greek_rayleigh(0.2), ϖ_Cabannes=0.96, ϖ_λ1λ0=vp, i_λ1λ0=offs, rrs_strict_reference=bool(strict))
R, T, ieR, ieT, hdr, up, dw = rt.rt_run_rrs(RS, m)
np.savez(sys.argv[6], R=R, T=T, ieR=ieR, ieT=ieT, hdr=hdr, up=up, dw=dw, N=rtamd.prepare_scene(rt._with_cabannes(RS, m)).N)
