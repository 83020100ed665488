"""CPU tests of the line-by-line absorption HOST path: the product's prefactor code (absorption.py: TIPS-2017 qoft,
mol_weight, windows) against the oracle-side restatement (oracle/absref.py, no shared code) and against published
TIPS-2017 partition sums; the committed golden spectrum against the oracle."""
from pathlib import Path

import numpy as np
import pytest

from oracle import absref
from oracle import momref as mr

GOLD = Path(__file__).parent / "golden"

# Q(296 K) of the principal isotopologues as published with TIPS-2017 (Gamache et al., JQSRT 203 (2017); HITRAN
# "Q(296 K)" column of the isotopologue metadata)
PUBLISHED_Q296 = {(1, 1): 174.58, (2, 1): 286.09, (2, 2): 576.64, (5, 1): 107.42, (6, 1): 590.48, (7, 1): 215.73}


def test_tips_tables_match_published_q296():
    t = absref.tables()
    for (M, I), q in PUBLISHED_Q296.items():
        TT, TQ = t[f"T_{M}_{I}"], t[f"Q_{M}_{I}"]
        z, h = absref.spline_second_derivatives(TQ, TT)
        assert abs(absref.spline_eval(TQ, TT, z, h, 296.0) / q - 1) < 5e-4, (M, I)


def test_qoft_product_vs_oracle(rtamd):
    ab = rtamd.absorption
    g = np.load(GOLD / "voigt_co2.npz")
    for M, I, T, q in g["qoft_cases"]:
        a, b = ab.qoft(int(M), int(I), float(T)), absref.qoft(int(M), int(I), float(T))
        assert b == q                       # the committed value is the oracle's
        assert abs(a - b) <= 1e-9 * b       # Float32 spline set-up: LAPACK dense solve vs the product's recurrence
    assert ab.qoft(2, 2, 296.0) == 1.0
    # the reference's stand-in-free behaviour: CO2 at 220 K differs from a rigid rotor by > 5 %
    assert abs(ab.qoft(2, 1, 220.0) / ab.linear_rotor_qratio(220.0) - 1) > 0.05
    with pytest.raises(AssertionError):
        ab.qoft(2, 1, 0.5)                  # TIPS2017: T must be between Tmin and Tmax (:204)
    with pytest.raises(KeyError):
        ab.mol_weight(7, 4)                 # unfilled (mol, iso) pair: check_exists (mol_weights.jl:19)
    assert ab.mol_weight(2, 1) == np.float32(43.98983) and ab.mol_weight(2, 1).dtype == np.float32


@pytest.mark.parametrize("p,T,vmr", [(1013.25, 296.0, 0.0), (250.0, 220.0, 0.0), (500.0, 260.0, 0.3)])
def test_line_prefactors_product_vs_oracle(rtamd, p, T, vmr):
    """compute_absorption_cross_section.jl:73-116 on the reference's HITRAN fixture: every per-line quantity of the
    product's vectorised host code against the oracle's line-by-line loop."""
    ab = rtamd.absorption
    ht = ab.read_hitran(GOLD / "testCO2.data")
    grid = np.arange(5990.0, 6400.0, 0.01)
    pf = ab.line_prefactors(ab.hitran_table(ht), grid, p, T, vmr=vmr, wing_cutoff=40.0)
    nu, gd, y, S, i0, i1 = absref.line_parameters(ht, grid, p, T, vmr, 40.0)
    assert len(pf.ν) == len(nu) == 6
    np.testing.assert_array_equal(pf.ν, nu)
    np.testing.assert_allclose(pf.γ_d, gd, rtol=1e-15)
    np.testing.assert_allclose(pf.y, y, rtol=1e-14)
    np.testing.assert_allclose(pf.S, S, rtol=1e-9)
    np.testing.assert_array_equal(pf.ind_start, i0)
    np.testing.assert_array_equal(pf.ind_stop, i1)


def test_window_fill_values_outside_the_grid(rtamd):
    """compute_absorption_cross_section.jl:60-61, :104-105: grid_idx_interp_low / _high are LinearInterpolations with the
    CONSTANT extrapolation values 1 and length(grid) on both sides.  A line inside the padded grid whose pressure-shifted
    centre puts nu - wing beyond the last grid point starts at index 1 (not n), one whose nu + wing falls before the first
    grid point stops at n (not 1): both act on the whole grid.  Product host code and oracle, hand-derived expectation."""
    ab = rtamd.absorption
    tab = ab.synthetic_o2a_lines(4, seed=1)
    grid = np.linspace(12990.0, 13010.0, 900)
    tab.νᵢ[:] = [12988.0005, 12995.0, 13005.0, 13011.9995]
    tab.δ_air[:] = [-0.02, -0.02, 0.02, 0.02]
    hit = {"mol": tab.mol, "iso": tab.iso, "νᵢ": tab.νᵢ, "Sᵢ": tab.Sᵢ, "γ_air": tab.γ_air, "γ_self": tab.γ_self,
            "E_lower": tab.E_lower, "n_air": tab.n_air, "δ_air": tab.δ_air}
    pf = ab.line_prefactors(tab, grid, 930.0, 288.0, vmr=0.21, wing_cutoff=2.0)
    _, _, _, _, i0, i1 = absref.line_parameters(hit, grid, 930.0, 288.0, 0.21, 2.0)
    assert list(i0) == list(pf.ind_start) and list(i1) == list(pf.ind_stop)
    assert (i0[0], i1[0]) == (1, 900) and (i0[3], i1[3]) == (1, 900)          # whole grid
    assert 1 < i0[1] < i1[1] < 900 and 1 < i0[2] < i1[2] < 900                  # interior lines: ordinary windows
    # at p = 5 hPa the shift is 1e-4 cm-1: the same outer lines keep one-point windows at the grid's ends
    pf5 = ab.line_prefactors(tab, grid, 5.0, 215.0, vmr=0.21, wing_cutoff=2.0)
    assert (pf5.ind_start[0], pf5.ind_stop[0]) == (1, 1) and (pf5.ind_start[3], pf5.ind_stop[3]) == (900, 900)


def test_golden_spectrum_is_the_oracles():
    g = np.load(GOLD / "voigt_co2.npz")
    for tag in ("a", "b"):
        args = [g[f"{k}_{tag}"] for k in ("nu", "gamma_d", "y", "S", "ind_start", "ind_stop")]
        sig = mr.voigt_xsec(*args, g["grid"])
        assert np.array_equal(sig, g[f"sigma_{tag}"])
        assert sig.max() > 1e-27 and np.all(sig >= 0)


def test_spline_reproduces_nodes_and_is_smooth(rtamd):
    ab = rtamd.absorption
    TT, TQ = ab.get_TT(7, 1), ab.get_TQ(7, 1)
    sp = ab.CubicSpline(TQ, TT)
    for i in (0, 5, 100, len(TT) - 1):
        assert abs(sp(float(TT[i])) - float(TQ[i])) <= 2e-6 * float(TQ[i])
    xs = np.linspace(150.0, 350.0, 41)
    q = np.array([sp(x) for x in xs])
    assert np.all(np.diff(q) > 0)  # partition sums grow with temperature


def test_tips_tables_cover_every_molecule_of_the_reference(rtamd):
    """The bundled TIPS-2017 / isotopologue tables hold every HITRAN molecule of the reference's NetCDF files (ids 1-49, 157
    (molecule, isotopologue) pairs): qoft! and mol_weight work beyond the first seven molecules (CH3Cl = 24, HCN = 23, NH3 = 11)."""
    ab = rtamd.absorption
    tab = absref.tables()
    assert list(tab["molecules"]) == list(range(1, 50)) and len(tab["pairs"]) == 157
    for M, I in ((11, 1), (23, 1), (24, 2), (26, 1)):
        q_prod, q_ora = ab.qoft(M, I, 250.0, 296.0), absref.qoft(M, I, 250.0, 296.0)
        assert 1.0 < q_prod < 5.0 and abs(q_prod - q_ora) <= 1e-9 * q_ora
        assert float(ab.mol_weight(M, I)) == float(absref.mol_weight(M, I)) > 10.0
