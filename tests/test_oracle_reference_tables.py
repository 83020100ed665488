"""Pins the oracle on the reference's own known-answer tests (test/test_CoreRT.jl:3-83):
the 6SV1 and Natraj tables, committed as data in tests/golden/reference_tables.json
(extracted by tools/make_reference_tables.py).  Thresholds are the reference's."""
import json
from pathlib import Path

import numpy as np
import pytest

import helpers
from oracle import momref as mr

G = json.loads((Path(__file__).parent / "golden" / "reference_tables.json").read_text())


def _scene(sza, vza, vaz, tau, rho):
    pol = mr.Stokes_IQUV()
    quad = mr.rt_set_streams("RadauQuad", 20, sza, vza, pol.n)
    return mr.Scene(pol=pol, quad=quad, max_m=3, tau_rayl=np.full((1, 1), tau), tau_abs=np.zeros((1, 1)),
                    greek_rayleigh=mr.get_greek_rayleigh(0.0), albedo=rho, vza=np.asarray(vza, float),
                    vaz=np.asarray(vaz, float))


def natraj_scene():
    mu = np.array(G["natraj_mu"])
    vza1 = np.degrees(np.arccos(mu))
    phis = G["natraj_phi"]
    vza = np.tile(vza1, len(phis))
    vaz = np.repeat(phis, len(mu))
    return _scene(float(np.degrees(np.arccos(G["natraj_mu0"]))), vza, vaz, G["natraj_tau"], 0.0)


def natraj_errors(R):
    """R: [7*16, nStokes, 1] -> max relative errors (I, Q, U) with the reference's masks."""
    It, Qt, Ut = (np.array(G["natraj"][k]).T for k in ("I_trues", "Q_trues", "U_trues"))  # [7,16]
    Im, Qm, Um = (R[:, k, 0].reshape(7, 16) for k in range(3))
    with np.errstate(all="ignore"):
        dI = np.abs(It - Im) / It
        dQ = np.abs(Qt - Qm) / Qt
        dU = np.abs(Ut - Um) / Ut
    return float(dI.max()), float(dQ[Qm >= 0.01].max()), float(np.nanmax(dU[Um >= 0.01]))


def test_natraj_c_oracle(cref):
    """test_CoreRT.jl:40-83: I < 0.002, Q and U < 0.008 where the modelled value >= 0.01.
    Also pins the fingerprints recorded in SURVEY.md section 8c / BASELINE.md."""
    sc = natraj_scene()
    assert sc.N == 136 and sc.quad.imu0 == 10
    p = cref.pack_scene(sc)
    assert list(p.nd) == [18]
    R, _, info = cref.rt_run(p)
    assert info == 0
    eI, eQ, eU = natraj_errors(R)
    assert eI < 0.002 and eQ < 0.008 and eU < 0.008
    assert abs(eI - 1.3668e-3) < 2e-7 and abs(eQ - 7.7745e-3) < 2e-7 and abs(eU - 3.7466e-3) < 2e-7


@pytest.mark.parametrize("case", range(6))
def test_6sv1_c_oracle(cref, case):
    """test_CoreRT.jl:3-38: reflectance / μ₀ within ε = 0.006 of the 6SV1 tables, 3 SZA × 3 azimuths × 16 VZA."""
    c = G["sixsv_cases"][case]
    Rt = np.array(G["sixsv_R"][case])  # [sza][az][vza]
    vza1 = np.array(G["sixsv_vza"])
    worst = 0.0
    for si, sza in enumerate(c["sza"]):
        vza = np.tile(vza1, 3)
        vaz = np.repeat(np.array(c["az"], float), 16)
        sc = _scene(sza, vza, vaz, c["tau"], c["rho"])
        R, _, info = cref.rt_run(cref.pack_scene(sc))
        assert info == 0
        Rm = (R[:, 0, 0] / sc.quad.mu0).reshape(3, 16)
        worst = max(worst, float(np.max(np.abs(Rt[si] - Rm) / Rt[si])))
    assert worst < 0.006


def test_numpy_twin_natraj_subset():
    """The numpy twin on one azimuth (it is slow): same thresholds."""
    mu = np.array(G["natraj_mu"])
    sc = _scene(float(np.degrees(np.arccos(0.2))), np.degrees(np.arccos(mu)), [90.0] * 16, 0.5, 0.0)
    R, _ = mr.rt_run(sc)
    It = np.array(G["natraj"]["I_trues"])[:, 3]
    assert np.max(np.abs(It - R[:, 0, 0]) / It) < 0.002
