"""Voigt line-by-line kernel (csrc/voigt.hip) against the oracle and the golden spectrum."""
from pathlib import Path

import numpy as np
import pytest

pytestmark = pytest.mark.gpu
GOLD = Path(__file__).parent / "golden"


def test_golden_co2_spectrum(rtamd):
    g = np.load(GOLD / "voigt_co2.npz")
    for tag in ("a", "b"):
        args = [g[f"{k}_{tag}"] for k in ("nu", "gamma_d", "y", "S", "ind_start", "ind_stop")]
        sig = rtamd.voigt_xsec(*args, g["grid"])
        ref = g[f"sigma_{tag}"]
        assert np.max(np.abs(sig - ref)) <= 1e-13 * ref.max()  # same rational approximation: SURVEY 8c


def test_synthetic_o2a_against_oracle(rtamd, cref):
    ab = rtamd.absorption
    tab = ab.synthetic_o2a_lines(300)
    grid = np.linspace(12903.0, 13245.0, 10_000)
    for p, T in ((1013.25, 296.0), (100.0, 210.0)):
        pf = ab.line_prefactors(tab, grid, p, T, vmr=0.21, wing_cutoff=40.0)
        ref = cref.voigt_xsec(pf.ν, pf.γ_d, pf.y, pf.S, pf.ind_start, pf.ind_stop, grid)
        sig = ab.compute_absorption_cross_section(tab, grid, p, T, vmr=0.21, wing_cutoff=40.0)
        assert np.max(np.abs(sig - ref)) <= 1e-13 * ref.max()
        assert np.all(sig >= 0)


def test_edge_cases(rtamd, cref):
    grid = np.linspace(100.0, 101.0, 777)  # not a multiple of the block size
    one = lambda v: np.array([v])
    # no lines at all
    assert np.array_equal(rtamd.voigt_xsec(np.zeros(0), np.zeros(0), np.zeros(0), np.zeros(0), [], [], grid), np.zeros(777))
    # a single-point window, an empty window (start > stop), a full window; far wings hit the humlicek2 branch
    nu = np.array([100.5, 100.2, 100.9]); gd = np.array([1e-3, 2e-3, 5e-4]); y = np.array([0.5, 2.0, 1e-3])
    S = np.array([1e-20, 2e-20, 3e-20]); i0 = [300, 500, 1]; i1 = [300, 499, 777]
    sig = rtamd.voigt_xsec(nu, gd, y, S, i0, i1, grid)
    ref = cref.voigt_xsec(nu, gd, y, S, i0, i1, grid)
    assert np.max(np.abs(sig - ref)) <= 1e-13 * ref.max()
    # one-point grid (compute_absorption_cross_section.jl:111-114)
    assert rtamd.voigt_xsec(one(5.0), one(1e-2), one(1.0), one(1e-20), [1], [1], one(5.0))[0] > 0
    with pytest.raises(rtamd.MomError):
        rtamd.voigt_xsec(nu, gd, y, S, [0, 1, 1], [1, 1, 1], grid)  # window outside the grid
