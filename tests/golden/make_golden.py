#!/usr/bin/env python3
"""Generates tests/golden/*.npz with the numpy twin of the oracle (oracle/momref.py), in this
container.  The twin is pinned on the reference's own tables by
tests/test_oracle_reference_tables.py; these vectors freeze its per-operator outputs so that
the C oracle and the HIP library can be checked against stored numbers too.

  small_iqu.npz   IQU, 4 streams (N=12), S=4, Nz=3 incl. one aerosol basis: inputs + snapshots
                  after elemental / doublings 1,2,nd / interaction of every (m, layer) + final R,T
  natraj.npz      the Natraj scene (test_CoreRT.jl:40-83) inputs after host prep + R per azimuth
  cef.npz         Re/Im w(z) of the HW32SD approximation on a 64x64 (x, y) grid
  voigt_co2.npz   Voigt spectrum of the 16 lines of the reference's test file testCO2.data
                  (copied as data to tests/golden/) at two (p, T)
"""
import sys
from pathlib import Path

import numpy as np

ROOT = Path(__file__).resolve().parents[2]
sys.path.insert(0, str(ROOT))
sys.path.insert(0, str(ROOT / "tests"))
from oracle import momref as mr, cref  # noqa: E402
import rtamd  # noqa: E402  (host-side scene generator only; no GPU needed)
import helpers  # noqa: E402

OUT = ROOT / "tests" / "golden"


def small_iqu():
    model = rtamd.scenes.make_scene(3, 3, 3, 4, vza=(0.0,), vaz=(35.0,), seed=7, aerosol_total=0.3, aerosol_p0=500.0,
                                    aerosol_σp=300.0)
    sc = helpers.oracle_scene(model)
    snaps = {}

    def hook(stage, m, iz, added, comp):
        key = f"{stage}_m{m}_z{iz}"
        if stage in ("elemental", "doubling"):
            for nm in ("r_mp", "t_pp", "r_pm", "t_mm", "j0p", "j0m"):
                snaps[f"{key}_{nm}"] = getattr(added, nm).copy()
        else:
            for nm in ("R_mp", "R_pm", "T_pp", "T_mm", "J0p", "J0m"):
                snaps[f"{key}_{nm}"] = getattr(comp, nm).copy()

    R, T = mr.rt_run(sc, hook=hook)
    # doubling iterations 1, 2 of the thickest layer, m = 0
    layers = mr.construct_core_optical_properties(sc, 0)
    _, tau_sum = mr.extract_effective_props(layers)
    iz = int(np.argmax([mr.get_dtau_ndoubl(l.tau, l.varpi, sc.quad.qp_mu)[1] for l in layers]))
    lay = layers[iz]
    dtau, nd = mr.get_dtau_ndoubl(lay.tau, lay.varpi, sc.quad.qp_mu)
    added = mr.make_added_layer(sc.N, sc.S)
    Zpp, Zmp = lay.Zfull()
    mr.elemental(sc.pol, sc.quad, tau_sum[:, iz], dtau, lay.varpi, Zpp, Zmp, 0, nd, added)
    it = []
    mr.doubling(sc.pol, np.exp(-dtau / sc.quad.mu0), nd, added, snapshots=it)
    for k in (0, 1):
        for nm, arr in zip(("r_mp", "t_pp", "j0p", "j0m"), it[k]):
            snaps[f"dbl_iter{k + 1}_z{iz + 1}_{nm}"] = arr
    p = cref.pack_scene(sc)
    np.savez_compressed(OUT / "small_iqu.npz", R=R, T=T, tau_rayl=model.τ_rayl, tau_abs=model.τ_abs, tau_aer=model.τ_aer,
                        nd=p.nd, iface=p.iface, Zpp=p.Zpp, Zmp=p.Zmp, zw=p.zw, tau=p.tau, varpi=p.varpi,
                        tau_sum=p.tau_sum, mu=p.mu, wt=p.wt, dbl_layer=iz + 1, **snaps)
    print("small_iqu: N", sc.N, "nd", p.nd, "keys", len(snaps))


def natraj():
    sys.path.insert(0, str(ROOT / "tests"))
    import test_oracle_reference_tables as t
    sc = t.natraj_scene()
    p = cref.pack_scene(sc)
    R, T, info = cref.rt_run(p)  # C oracle (fast); cross-checked against the twin on one azimuth below
    assert info == 0
    np.savez_compressed(OUT / "natraj.npz", R=R, T=T, mu=p.mu, wt=p.wt, imu0=p.imu0, mu0=p.mu0, nd=p.nd, iface=p.iface,
                        Zpp=p.Zpp, Zmp=p.Zmp, tau=p.tau, varpi=p.varpi, zw=p.zw, tau_sum=p.tau_sum, node=p.node,
                        cos_mphi=p.cos_mphi, sin_mphi=p.sin_mphi, vza=sc.vza, vaz=sc.vaz)
    print("natraj: N", sc.N, "errors", t.natraj_errors(R))


def cef():
    x = np.linspace(-40, 40, 64)
    y = np.logspace(-4, np.log10(9.0), 64)
    X, Y = np.meshgrid(x, y, indexing="ij")
    w = mr.w_hw32sd(X + 1j * Y)
    np.savez_compressed(OUT / "cef.npz", x=x, y=y, w_re=w.real, w_im=w.imag)


def voigt_co2():
    """Inputs: the reference's 16-line HITRAN fixture (test/test_profiles/testCO2.data).  Expected values come from
    the ORACLE side only (oracle/absref.py host restatement + oracle/momref.py line shape): per-line parameters with
    the TIPS-2017 correction, and the Voigt cross section on a 0.01 cm^-1 grid at two (p, T)."""
    from oracle import absref
    ht = rtamd.absorption.read_hitran(OUT / "testCO2.data")
    grid = np.arange(5990.0, 6400.0, 0.01)
    out = {"grid": grid}
    for tag, (p, T) in {"a": (1013.25, 296.0), "b": (250.0, 220.0)}.items():
        nu, gd, y, S, i0, i1 = absref.line_parameters(ht, grid, p, T, 0.0, 40.0)
        out[f"sigma_{tag}"] = mr.voigt_xsec(nu, gd, y, S, i0, i1, grid)
        for k, v in (("nu", nu), ("gamma_d", gd), ("y", y), ("S", S), ("ind_start", i0), ("ind_stop", i1)):
            out[f"{k}_{tag}"] = v
        out[f"pT_{tag}"] = np.array([p, T])
    out["qoft_cases"] = np.array([(M, I, T, absref.qoft(M, I, T)) for (M, I, T) in
                                  [(2, 1, 220.0), (2, 2, 250.0), (7, 1, 250.0), (7, 1, 1000.0), (1, 1, 310.5), (6, 1, 180.0), (5, 1, 77.0)]])
    np.savez_compressed(OUT / "voigt_co2.npz", **out)
    print("voigt_co2: lines", len(out["nu_a"]), "max sigma", out["sigma_a"].max())


if __name__ == "__main__":
    small_iqu()
    natraj()
    cef()
    voigt_co2()
    for f in sorted(OUT.glob("*")):
        print(f.name, f.stat().st_size)
