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


def test_line_order_sorted_and_unsorted(rtamd, cref):
    """Sorted line lists (window starts and stops non-decreasing: a block finds its lines by bisection) and a shuffled
    list (full scan) give the oracle's sums -- each in its own line order, which is the accumulation order."""
    ab = rtamd.absorption
    tab = ab.synthetic_o2a_lines(2000)
    grid = np.linspace(12903.0, 13245.0, 30_000)
    pf = ab.line_prefactors(tab, grid, 500.0, 250.0, vmr=0.21, wing_cutoff=3.0)
    assert np.all(np.diff(pf.ind_start) >= 0) and np.all(np.diff(pf.ind_stop) >= 0)
    args = [pf.ν, pf.γ_d, pf.y, pf.S, pf.ind_start, pf.ind_stop]
    ref = cref.voigt_xsec(*args, grid)
    sig = rtamd.voigt_xsec(*args, grid)
    assert np.max(np.abs(sig - ref)) <= 1e-13 * ref.max()
    perm = np.random.default_rng(0).permutation(len(pf.ν))
    args_p = [np.ascontiguousarray(a[perm]) for a in args]
    sig_p = rtamd.voigt_xsec(*args_p, grid)
    assert np.max(np.abs(sig_p - cref.voigt_xsec(*args_p, grid))) <= 1e-13 * ref.max()
    assert np.max(np.abs(sig_p - sig)) <= 1e-12 * ref.max()


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


def test_product_host_path_end_to_end_vs_golden(rtamd):
    """compute_absorption_cross_section (product host code: read_hitran -> TIPS qoft -> prefactors -> GPU kernel) on
    the reference's HITRAN fixture against the ORACLE-side golden spectrum (tests/golden/make_golden.py: oracle/absref.py
    + oracle/momref.py): nothing is shared between the two routes but the input file and the TIPS data table."""
    ab = rtamd.absorption
    g = np.load(GOLD / "voigt_co2.npz")
    tab = ab.hitran_table(ab.read_hitran(GOLD / "testCO2.data"))
    for tag in ("a", "b"):
        p, T = g[f"pT_{tag}"]
        sig = ab.compute_absorption_cross_section(tab, g["grid"], float(p), float(T), vmr=0.0, wing_cutoff=40.0)
        ref = g[f"sigma_{tag}"]
        assert np.max(np.abs(sig - ref)) <= 1e-9 * ref.max()  # Float32 spline set-up differs by ~3e-11 between the routes


def test_error_text_reaches_the_caller(rtamd):
    grid = np.linspace(100.0, 101.0, 50)
    one = lambda v: np.array([v])
    with pytest.raises(rtamd.MomError) as e:
        rtamd.voigt_xsec(one(100.5), one(1e-3), one(0.5), one(1e-20), [0], [10], grid)
    assert "window [0, 10] outside the grid 1..50" in str(e.value)
    with pytest.raises(rtamd.MomError) as e:
        rtamd.voigt_xsec(one(100.5), one(1e-3), one(0.5), one(1e-20), [1], [10], grid, device=99)
    assert "device index out of range" in str(e.value)


def test_voigt_tau_abs_accumulates_into_resident_table(rtamd):
    """compute_absorption_profile! (atmo_prof.jl:427-449) on the handle: tau_abs[:, iz] += sigma * vcd_dry[iz] * vmr with
    sigma never leaving the GPU, two absorbers into the same table; bitwise the host accumulation of mom_voigt_xsec
    spectra (same kernel, separately rounded product and sum)."""
    ab = rtamd.absorption
    S, Nz = 3000, 4
    grid = np.linspace(12950.0, 13150.0, S)
    o2, co2ish = ab.synthetic_o2a_lines(120, seed=5), ab.synthetic_o2a_lines(40, seed=6)
    p_full = np.array([50.0, 300.0, 700.0, 980.0]); T = np.array([220.0, 235.0, 270.0, 290.0])
    vcd = np.array([2.0e23, 1.1e24, 2.5e24, 1.9e24])
    expect = np.zeros((S, Nz))
    for tab, vmr, mv in ((o2, 0.21, 0.21), (co2ish, np.array([4e-4, 4.1e-4, 4.2e-4, 4.3e-4]), 0.0)):
        for iz in range(Nz):
            v = vmr if np.ndim(vmr) == 0 else vmr[iz]
            sig = ab.compute_absorption_cross_section(tab, grid, p_full[iz], T[iz], vmr=mv)
            expect[:, iz] += sig * (vcd[iz] * v)
    with rtamd.Handle(4, 1, S, 1) as h:
        ab.compute_absorption_profile(h, o2, grid, p_full, T, vcd, 0.21, model_vmr=0.21)
        ab.compute_absorption_profile(h, co2ish, grid, p_full, T, vcd, np.array([4e-4, 4.1e-4, 4.2e-4, 4.3e-4]), begin=False)
        got = h.absorption_get()
        with pytest.raises(rtamd.MomError) as e:
            h.voigt_tau_abs(5, [1.0], [1.0], [1.0], [1.0], [1], [1], 1.0)   # layer out of range
        assert e.value.code == rtamd._lib.MOM_EINVAL
    assert expect.max() > 1e-3 and np.array_equal(got, expect)


def _hit(tab):
    """product HitranTable -> the read_hitran-style columns oracle/absref.py takes"""
    return {"mol": tab.mol, "iso": tab.iso, "νᵢ": tab.νᵢ, "Sᵢ": tab.Sᵢ, "γ_air": tab.γ_air, "γ_self": tab.γ_self,
            "E_lower": tab.E_lower, "n_air": tab.n_air, "δ_air": tab.δ_air}


@pytest.mark.parametrize("lines", ["o2a", "co2_file", "shuffled", "edge_shift"])
def test_device_side_line_prefactors(rtamd, cref, lines):
    """SURVEY 8f-1, last clause: the per-line prefactors of compute_absorption_cross_section.jl:73-107 (pressure shift, Lorentz
    and Doppler widths, y, TIPS-2017 spline ratio qoft!, Boltzmann / stimulated-emission factors, grid windows) formed on the
    device from ONE resident line table (mom_absorption_set_lines); per layer only (p, T, vmr, vcd) are passed.  Asserted
    DIRECTLY against the oracle (oracle/absref.line_parameters: the host loop of the reference restated line by line, no code
    shared with the product): windows identical, prefactors to rounding (Float32 spline set-up differs by ~3e-11 between
    the routes, device exp / pow by a few ulp), tau_abs of every layer against absref + the oracle's Voigt sum; and, as
    before, against the product's host route.  "edge_shift": lines inside the padded grid whose pressure-shifted centre puts
    nu - wing beyond the last grid point or nu + wing before the first one -- the reference's two interpolators return their
    constant (1 / n) on BOTH sides of the grid (compute_absorption_cross_section.jl:60-61), so those lines act on the whole
    grid."""
    from oracle import absref
    ab = rtamd.absorption
    if lines == "co2_file":
        tab = ab.hitran_table(ab.read_hitran(GOLD / "testCO2.data"))
        grid = np.linspace(float(tab.νᵢ.min()) - 2.0, float(tab.νᵢ.max()) + 2.0, 3000)
        model_vmr, wing = 4e-4, 5.0
    elif lines == "edge_shift":
        tab = ab.synthetic_o2a_lines(60, seed=9)
        grid = np.linspace(12990.0, 13010.0, 900)
        model_vmr, wing = 0.21, 2.0
        tab.νᵢ[:] = np.sort(np.concatenate([np.linspace(12988.001, 13011.999, 56), [12988.0005, 12988.002, 13011.998, 13011.9995]]))
        tab.δ_air[:] = np.where(tab.νᵢ < 13000.0, -0.02, 0.02)     # shifts the outermost lines past the padded edge at p ~ 1 atm
    else:
        tab = ab.synthetic_o2a_lines(400, seed=5)
        grid = np.linspace(12920.0, 13230.0, 4000)      # some lines fall outside the padded grid and are dropped
        model_vmr, wing = 0.21, 8.0
        if lines == "shuffled":                           # windows not monotone in the line index: the strided search path
            rng = np.random.default_rng(3)
            perm = rng.permutation(tab.νᵢ.size)
            import dataclasses
            tab = ab.HitranTable(**{f.name: getattr(tab, f.name)[perm] for f in dataclasses.fields(tab)})
            tab.E_lower[::7] = -1.0                       # "no temperature correction" rows (:96)
    p_full = np.array([5.0, 120.0, 480.0, 930.0])
    T = np.array([215.0, 231.5, 262.25, 288.0])
    vcd = np.array([1.1e23, 2.4e24, 7.7e24, 1.3e25])
    S = grid.size
    m = rtamd.scenes.make_scene(1, 3, 4, S)
    with rtamd.corert.make_handle(m) as h_dev, rtamd.corert.make_handle(m) as h_host:
        ab.compute_absorption_profile(h_host, tab, grid, p_full, T, vcd, 0.3, wing_cutoff=wing, model_vmr=model_vmr)
        ab.compute_absorption_profile(h_dev, tab, grid, p_full, T, vcd, 0.3, wing_cutoff=wing, model_vmr=model_vmr,
                                      device_prefactors=True)
        # the prefactors of the last layer, device vs host
        pf = ab.line_prefactors(tab, grid, p_full[-1], T[-1], vmr=model_vmr, wing_cutoff=wing)
        nu, gd, y, Sl, i0, i1 = h_dev.absorption_get_prefactors()
        assert nu.size == pf.ν.size
        assert np.array_equal(i0, pf.ind_start) and np.array_equal(i1, pf.ind_stop)
        assert np.array_equal(nu, pf.ν)
        np.testing.assert_allclose(gd, pf.γ_d, rtol=4e-16)
        np.testing.assert_allclose(y, pf.y, rtol=2e-15)
        np.testing.assert_allclose(Sl, pf.S, rtol=2e-14)
        # ... and the oracle's, directly
        onu, ogd, oy, oS, oi0, oi1 = absref.line_parameters(_hit(tab), grid, p_full[-1], T[-1], model_vmr, wing)
        assert onu.size == nu.size and np.array_equal(i0, oi0) and np.array_equal(i1, oi1)
        assert np.array_equal(nu, onu)
        np.testing.assert_allclose(gd, ogd, rtol=1e-15)
        np.testing.assert_allclose(y, oy, rtol=4e-15)
        np.testing.assert_allclose(Sl, oS, rtol=1e-9)
        if lines == "edge_shift":
            assert np.any((i0 == 1) & (i1 == S) & (nu - wing > grid[-1])) and np.any((i0 == 1) & (i1 == S) & (nu + wing < grid[0]))
        a, b = h_dev.absorption_get(), h_host.absorption_get()
        # the profile entry (all layers in two launches) against the layer-by-layer entry: the same arithmetic, bit for bit
        ab.compute_absorption_profile(h_host, tab, grid, p_full, T, vcd, 0.3, wing_cutoff=wing, model_vmr=model_vmr,
                                      device_prefactors=True, layer_by_layer=True)
        assert np.array_equal(a, h_host.absorption_get())
    assert b.max() > 0
    np.testing.assert_allclose(a, b, rtol=1e-12, atol=1e-300)
    for iz in range(4):     # tau_abs[:, iz] = sigma(p, T) * (vcd * vmr) with sigma from the oracle alone (atmo_prof.jl:446)
        prm = absref.line_parameters(_hit(tab), grid, p_full[iz], T[iz], model_vmr, wing)
        ref = cref.voigt_xsec(*prm, grid) * (vcd[iz] * 0.3)
        assert np.max(np.abs(a[:, iz] - ref)) <= 1e-9 * ref.max()
    # a temperature outside the TIPS tables is refused like qoft! does (:204)
    with rtamd.corert.make_handle(m) as h:
        h.absorption_begin(1, grid)
        ab.resident_line_table(h, tab, grid, wing)
        with pytest.raises(rtamd._lib.MomError) as e:
            h.voigt_tau_abs_layer(1, 500.0, 0.5, 0.0, wing, 1.0)
        assert "TIPS2017" in str(e.value)
