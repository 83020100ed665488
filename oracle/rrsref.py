"""
oracle/rrsref.py -- TEST INFRASTRUCTURE ONLY (numpy restatement of the reference's rotational-Raman path).

Restates, in plain numpy, the inelastic (RRS) branch of vSmartMOM.jl's CoreRT layer loop (BASELINE config 5, SURVEY
section 8f-3): rt_run(::RRS) -> rt_kernel!(::RRS) -> elemental_inelastic! + elemental! -> doubling_helper!(::RRS) ->
interaction_helper!(::RRS, iface) -> surface interaction -> postprocessing_vza!(::RRS).  Every function cites the
reference file:line it follows (paths relative to the reference root, src/CoreRT unless stated).  Nothing in the
product path may import this file (tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg only).

PARITY UNPINNED: the reference holds no known-answer test for its Raman path (test/benchmarks/prototype_inelastic*.jl
are scripts without recorded outputs) and Julia is absent, so this restatement is pinned only through (a) its elastic
limit, (b) first-order perturbation identities of the adding equations (tests/test_oracle_rrs.py) and (c) a reading of
the source text.

Layout: elastic operators M[n, i, j], sources J[n, i] (as oracle/momref.py); inelastic operators ie[dn, n1, i, j] and
sources ieJ[dn, n1, i] (the reference's [i, j, n1, dn] with the axes reversed).  The Raman offset table `i_l1l0[dn]` is
n0 - n1 in grid points: the operator ie[dn, n1] takes radiation at spectral index n0 = n1 + i_l1l0[dn] to index n1.

The reference's RRS text has defects; like quirk Q1 they are handled by a switch, `rrs_strict_reference`:
  True  = the text as written, with the semantics of a single-threaded Julia run, wherever it is executable; where
          it is not executable (a MethodError at run time) ReferenceRaises is raised;
  False = the corrections D1..D5 below -- and nothing else: every other line is as written in both positions.
D1  CoreKernel/doubling_inelastic.jl:90-95   the ELASTIC source update j0-+ and `expk .= expk.^2` sit inside the
    `for dn = 1:nRaman` loop, i.e. run nRaman times per doubling step (and the ieJ0- update of line :78-87 reads the
    j0+ of the previous pass).  Corrected: once per doubling step, after the loop.
D2  doubling_inelastic.jl:291-311 apply_D_IE_RRS! indexes the RAMAN axis with n0 = n + i_l1l0[dn] and tests
    1 <= n0 <= size(.,4) = nRaman.  Corrected: element [.., n, dn], every (n, dn).
D3  doubling_inelastic.jl:345-357 apply_D_SFI_IE_RRS! writes ieJ0-[i,1,n,n0] = -ieJ0-[i,1,n,dn] under the same test
    (a read-write overlap across work items: single-thread column-major order is taken, dn slowest).  Corrected:
    in-place sign flip of element [i,1,n,dn].
D4  interaction_inelastic.jl:8-22 (interface 00), :28-76 (01), :139-180 (10): the methods are typed on
    CompositeLayer{FT}/AddedLayer{FT}, which CompositeLayerRS/AddedLayerRS (types.jl:145-205) are not, and 01/10
    iterate `for n1 in eachindex ieJ1+[1,1,:,1]`, which parses as a loop over the function `eachindex` followed by a
    reference to the undefined `ieJ1+`: a MethodError when called.  Corrected: loops over n1 = 1:nSpec, dn = 1:nRaman
    with n0 = n1 + i_l1l0[dn] restricted to the grid, bodies as written.
D5  doubling_inelastic.jl:78-87 reads added_layer.iet-- inside the doubling loop; for ndoubl >= 1 nothing has written
    it for the current layer (apply_D_elemental_RRS!, elemental_inelastic.jl:384-402, fills it only for ndoubl < 1):
    it holds the previous layer's post-doubling value.  Corrected: iet++ of the current step (in the doubling's
    D-transformed basis t-- == t++, doubling.jl:43 comment).
Kept as written in both positions (noted, not "corrected"): the ieJ0- update reads the ieJ0+ it has just updated
(:67-87); the ier-+ update reads the iet++ it has just updated (:104-124); get_elem_rt_SFI_RRS! does not reset
off-grid entries (elemental_inelastic.jl:345) and applies D for ndoubl >= 1 to every entry (:378-380);
apply_D_matrix_elemental_SFI! changes nothing (:478-490); the surface layer's ie* arrays are never written (zeros).
"""
from __future__ import annotations

from dataclasses import dataclass, field
from typing import List, Optional

import numpy as np

from . import momref as mr


class ReferenceRaises(RuntimeError):
    """The reference's text raises a Julia exception on this path (strict position of the switch)."""


@dataclass
class RRSInputs:
    """The fields of InelasticScattering.RRS (src/Inelastic/types.jl:13-33) the hot path reads."""
    i_l1l0: np.ndarray          # [nRaman] int grid offsets n0 - n1
    varpi_l1l0: np.ndarray      # [nRaman] Raman single-scattering albedo per offset
    greek_raman: mr.GreekCoefs
    rrs_strict_reference: bool = True
    # test-side window (not a reference field): restrict the inelastic pairs (n1, dn) to n1 in [lo, hi).  The pair (n1, dn)
    # reads the elastic operators at n1 and n0 = n1 + i_l1l0[dn] and the inelastic ones at (n1, dn) only, so the owned
    # entries are those of the unrestricted run (the GPU's spectral shards rest on the same fact);
    # corrected position only (the strict position's defects D2/D3 index across the Raman axis).
    owned: tuple = None

    @property
    def nRaman(self):
        return len(self.i_l1l0)


@dataclass
class AddedLayerRS(mr.AddedLayer):
    """types.jl:175-205"""
    ier_pm: np.ndarray = None
    ier_mp: np.ndarray = None
    iet_mm: np.ndarray = None
    iet_pp: np.ndarray = None
    ieJ0p: np.ndarray = None
    ieJ0m: np.ndarray = None


@dataclass
class CompositeLayerRS(mr.CompositeLayer):
    """types.jl:145-173"""
    ieR_mp: np.ndarray = None
    ieR_pm: np.ndarray = None
    ieT_pp: np.ndarray = None
    ieT_mm: np.ndarray = None
    ieJ0p: np.ndarray = None
    ieJ0m: np.ndarray = None


def make_added_layer_rs(N, S, nR) -> AddedLayerRS:
    """make_added_layer(::RRS, ...) tools/rt_helper_functions.jl:127-141 (zeros)."""
    a = mr.make_added_layer(N, S)
    z4 = lambda: np.zeros((nR, S, N, N))
    return AddedLayerRS(a.r_pm, a.r_mp, a.t_mm, a.t_pp, a.j0p, a.j0m, z4(), z4(), z4(), z4(), np.zeros((nR, S, N)),
                        np.zeros((nR, S, N)))


def make_composite_layer_rs(N, S, nR) -> CompositeLayerRS:
    c = mr.make_composite_layer(N, S)
    z4 = lambda: np.zeros((nR, S, N, N))
    return CompositeLayerRS(c.R_mp, c.R_pm, c.T_pp, c.T_mm, c.J0p, c.J0m, z4(), z4(), z4(), z4(), np.zeros((nR, S, N)),
                            np.zeros((nR, S, N)))


def get_n0_n1(S: int, delta: int, owned=None):
    """get_n0_n1 (src/Inelastic/inelastic_helper.jl:13-21): 0-based slices (n0, n1) of the valid index pairs
    (owned = (lo, hi): only those with lo <= n1 < hi, possibly none)."""
    if abs(delta) >= S:
        raise ReferenceRaises("get_n0_n1: no valid index (BoundsError on sub[1])")
    a, b = max(0, -delta), min(S, S - delta)
    if owned is not None:
        a, b = max(a, owned[0]), min(b, owned[1])
        b = max(a, b)
    return slice(a + delta, b + delta), slice(a, b)


def _mv(A, x):
    return np.einsum("sij,sj->si", A, x)


def elemental_inelastic(pol, quad, rrs: RRSInputs, fscatt, tau_sum, dtau, varpi, Zpp, Zmp, m, nd, added: AddedLayerRS,
                        strict_idx: bool):
    """elemental_inelastic!(::RRS) CoreKernel/elemental_inelastic.jl:23-91 on the PERSISTENT added layer: get_elem_rt_RRS!
    (:93-160) writes every (n1, dn) of ier-+/iet++ (zeros off the grid); get_elem_rt_SFI_RRS! (:320-382) writes only
    on-grid entries of ieJ0+-, then multiplies every entry of ieJ0- by D when ndoubl >= 1; apply_D_elemental_RRS! (:384-402)."""
    n, N = pol.n, len(quad.qp_muN)
    S, nR = len(dtau), rrs.nRaman
    ier, iet, _, _, jp, jm = mr.elemental_inelastic_rrs(pol, quad, rrs.i_l1l0, rrs.varpi_l1l0, fscatt, tau_sum, dtau, varpi,
                                                        Zpp, Zmp, m, 0, strict_idx, rrs.owned)  # nd = 0: no D applied in there
    added.ier_mp[:] = ier
    added.iet_pp[:] = iet
    for dn in range(nR):
        n0, n1 = get_n0_n1(S, int(rrs.i_l1l0[dn]), rrs.owned)
        added.ieJ0p[dn, n1] = jp[dn, n1]
        added.ieJ0m[dn, n1] = jm[dn, n1]
    if nd >= 1:
        added.ieJ0m *= np.tile(np.asarray(pol.D, dtype=np.float64), N // n)[None, None, :]
    comp = mr.stokes_comp(np.arange(N), n, strict_idx)
    if nd < 1:
        same = ((comp[:, None] <= 2) & (comp[None, :] <= 2)) | ((comp[:, None] > 2) & (comp[None, :] > 2))
        sgn = np.where(same, 1.0, -1.0)
        added.ier_pm[:] = sgn * added.ier_mp
        added.iet_mm[:] = sgn * added.iet_pp
    else:
        neg = comp > 2
        added.ier_mp[:, :, neg, :] = -added.ier_mp[:, :, neg, :]


def apply_D_IE(pol, rrs: RRSInputs, added: AddedLayerRS, strict_idx: bool):
    """apply_D_matrix_IE!(::RRS) doubling_inelastic.jl:410-425 with apply_D_IE_RRS! (:291-311)  [D2]."""
    n = pol.n
    if n == 1:
        added.ier_pm[:] = added.ier_mp
        added.iet_mm[:] = added.iet_pp
        return
    nR, S, N, _ = added.ier_mp.shape
    comp = mr.stokes_comp(np.arange(N), n, strict_idx)
    neg = comp > 2
    same = ((comp[:, None] <= 2) & (comp[None, :] <= 2)) | ((comp[:, None] > 2) & (comp[None, :] > 2))
    sgn = np.where(same, 1.0, -1.0)
    if not rrs.rrs_strict_reference:
        added.ier_mp[:, :, neg, :] = -added.ier_mp[:, :, neg, :]
        added.ier_pm[:] = sgn * added.ier_mp
        added.iet_mm[:] = sgn * added.iet_pp
        return
    for dn in range(nR):  # work item (n, dn) touches element [.., n, n0], n0 = n + i[dn], if 1 <= n0 <= nRaman
        off = int(rrs.i_l1l0[dn])
        nn = np.arange(S)
        kk = nn + off            # 0-based 4th index: n0 - 1 = (n + 1 + off) - 1
        ok = (kk >= 0) & (kk < nR)
        nn, kk = nn[ok], kk[ok]
        if nn.size == 0:
            continue
        blk = added.ier_mp[kk, nn]
        blk[:, neg, :] = -blk[:, neg, :]
        added.ier_mp[kk, nn] = blk
        added.ier_pm[kk, nn] = sgn * blk
        added.iet_mm[kk, nn] = sgn * added.iet_pp[kk, nn]


def apply_D_SFI_IE(pol, rrs: RRSInputs, added: AddedLayerRS, strict_idx: bool):
    """apply_D_matrix_SFI_IE!(::RRS) doubling_inelastic.jl:441-451 with apply_D_SFI_IE_RRS! (:345-357)  [D3]."""
    n = pol.n
    if n == 1:
        return
    nR, S, N = added.ieJ0m.shape
    neg = mr.stokes_comp(np.arange(N), n, strict_idx) > 2
    if not rrs.rrs_strict_reference:
        added.ieJ0m[:, :, neg] = -added.ieJ0m[:, :, neg]
        return
    for dn in range(nR):  # single-thread column-major order: dn is the slowest index
        off = int(rrs.i_l1l0[dn])
        nn = np.arange(S)
        kk = nn + off
        ok = (kk >= 0) & (kk < nR)
        nn, kk = nn[ok], kk[ok]
        if nn.size == 0:
            continue
        src = added.ieJ0m[dn, nn][:, neg].copy()
        tgt = added.ieJ0m[kk, nn]
        tgt[:, neg] = -src
        added.ieJ0m[kk, nn] = tgt


def doubling_inelastic(pol, rrs: RRSInputs, expk, nd, added: AddedLayerRS, strict_idx: bool):
    """doubling_helper!(::RRS) doubling_inelastic.jl:13-134.  expk is updated in place."""
    if nd == 0:
        return
    strict = rrs.rrs_strict_reference
    r, t, jp, jm = added.r_mp, added.t_pp, added.j0p, added.j0m
    ier, iet, ieJp, ieJm = added.ier_mp, added.iet_pp, added.ieJ0p, added.ieJ0m
    S, N = r.shape[0], r.shape[1]
    nR = rrs.nRaman
    I = np.eye(N)[None]
    for _ in range(nd):
        gp = mr.batch_inv(I - r @ r)                                           # :47
        ttgp = t @ gp                                                          # :48
        j1p = jp * expk[:, None]                                               # :51
        ieJ1p = ieJp * expk[None, :, None]                                     # :52
        j1m = jm * expk[:, None]                                               # :55
        ieJ1m = ieJm * expk[None, :, None]                                     # :56
        tmp1 = _mv(gp, jp + _mv(r, j1m))                                       # :58
        tmp2 = _mv(gp, j1m + _mv(r, jp))                                       # :59
        for dn in range(nR):                                                   # :61-96
            n0, n1 = get_n0_n1(S, int(rrs.i_l1l0[dn]), rrs.owned)
            X = r[n1] @ ier[dn, n1] + ier[dn, n1] @ r[n0]
            ieJp[dn, n1] = ieJ1p[dn, n1] + _mv(ttgp[n1], ieJp[dn, n1] + _mv(r[n1], ieJ1m[dn, n1]) +
                                               _mv(ier[dn, n1], j1m[n0]) + _mv(X, tmp1[n0])) + _mv(iet[dn, n1], tmp1[n0])
            X2 = ier[dn, n1] @ r[n0] + r[n1] @ ier[dn, n1]
            itm = added.iet_mm[dn, n1] if strict else iet[dn, n1]              # D5
            ieJm[dn, n1] = ieJm[dn, n1] + _mv(ttgp[n1], ieJ1m[dn, n1] + _mv(ier[dn, n1], jp[n0]) +
                                              _mv(r[n1], ieJp[dn, n1]) + _mv(X2, tmp2[n0])) + _mv(itm, tmp2[n0])
            if strict:                                                         # D1: :90-95 inside the loop
                jm_new = jm + _mv(ttgp, j1m + _mv(r, jp))
                jp_new = j1p + _mv(ttgp, jp + _mv(r, j1m))
                jm[:] = jm_new
                jp[:] = jp_new
                expk[:] = expk ** 2
        if not strict:
            jm_new = jm + _mv(ttgp, j1m + _mv(r, jp))
            jp_new = j1p + _mv(ttgp, jp + _mv(r, j1m))
            jm[:] = jm_new
            jp[:] = jp_new
            expk[:] = expk ** 2
        for dn in range(nR):                                                   # :98-125
            n0, n1 = get_n0_n1(S, int(rrs.i_l1l0[dn]), rrs.owned)
            X = ier[dn, n1] @ r[n0] + r[n1] @ ier[dn, n1]
            tg1 = t[n1] @ gp[n1]
            Y = (X @ gp[n0]) @ t[n0]
            iet[dn, n1] = tg1 @ (iet[dn, n1] + Y) + (iet[dn, n1] @ gp[n0]) @ t[n0]
            ier[dn, n1] = ier[dn, n1] + (tg1 @ r[n1]) @ (iet[dn, n1] + Y) + \
                ((iet[dn, n1] @ gp[n0]) @ r[n0] + tg1 @ ier[dn, n1]) @ t[n0]
        r_new = r + (ttgp @ r) @ t                                             # :128
        t_new = ttgp @ t                                                       # :131
        r[:] = r_new
        t[:] = t_new
    # apply_D_matrix! (doubling.jl:120-134 / :93-110), apply_D_matrix_IE!, apply_D_matrix_SFI!, apply_D_matrix_SFI_IE!
    n = pol.n
    if n == 1:
        added.r_pm[:] = r
        added.t_mm[:] = t
    else:
        comp = mr.stokes_comp(np.arange(N), n, strict_idx)
        neg = comp > 2
        r[:, neg, :] = -r[:, neg, :]
        same = ((comp[:, None] <= 2) & (comp[None, :] <= 2)) | ((comp[:, None] > 2) & (comp[None, :] > 2))
        sgn = np.where(same, 1.0, -1.0)[None]
        added.r_pm[:] = sgn * r
        added.t_mm[:] = sgn * t
    apply_D_IE(pol, rrs, added, strict_idx)
    if n > 1:
        neg = mr.stokes_comp(np.arange(N), n, strict_idx) > 2
        jm[:, neg] = -jm[:, neg]
    apply_D_SFI_IE(pol, rrs, added, strict_idx)


def interaction_inelastic(rrs: RRSInputs, iface: int, comp: CompositeLayerRS, added: AddedLayerRS):
    """interaction_helper!(::RRS, iface, SFI = true, ...) interaction_inelastic.jl: 00 :8-22, 01 :28-76, 10 :139-180,
    11 :230-340.  iface: 0 = '00', 1 = '01', 2 = '10', 3 = '11'."""
    S, N = comp.R_mp.shape[0], comp.R_mp.shape[1]
    nR = rrs.nRaman
    a, c = added, comp
    if iface != 3 and rrs.rrs_strict_reference:                                # D4
        raise ReferenceRaises("interaction_helper!(::RRS, ::ScatteringInterface_%s): MethodError in the reference "
                              "(interaction_inelastic.jl:%s)" % (("00", "01", "10")[iface], ("8-12", "28-37", "139-148")[iface]))
    if iface == 0:
        c.ieJ0p[:] = 0.0
        c.ieJ0m[:] = 0.0
        J0p = a.j0p + _mv(a.t_pp, c.J0p)
        J0m = c.J0m + _mv(c.T_mm, a.j0m)
        c.J0p[:] = J0p
        c.J0m[:] = J0m
        c.T_mm[:] = a.t_mm @ c.T_mm
        c.T_pp[:] = a.t_pp @ c.T_pp
        return
    if iface == 1:
        for dn in range(nR):
            n0, n1 = get_n0_n1(S, int(rrs.i_l1l0[dn]), rrs.owned)
            c.ieJ0m[dn, n1] = _mv(c.T_mm[n1], _mv(a.ier_mp[dn, n1], c.J0p[n0]) + a.ieJ0m[dn, n1])
            c.ieJ0p[dn, n1] = a.ieJ0p[dn, n1] + _mv(a.iet_pp[dn, n1], c.J0p[n0])
        J0m = c.J0m + _mv(c.T_mm, _mv(a.r_mp, c.J0p) + a.j0m)
        J0p = a.j0p + _mv(a.t_pp, c.J0p)
        c.J0m[:] = J0m
        c.J0p[:] = J0p
        for dn in range(nR):
            n0, n1 = get_n0_n1(S, int(rrs.i_l1l0[dn]), rrs.owned)
            c.ieR_mp[dn, n1] = (c.T_mm[n1] @ a.ier_mp[dn, n1]) @ c.T_pp[n0]
            c.ieR_pm[dn, n1] = a.ier_pm[dn, n1]
            c.ieT_pp[dn, n1] = a.iet_pp[dn, n1] @ c.T_pp[n0]
            c.ieT_mm[dn, n1] = c.T_mm[n1] @ a.iet_mm[dn, n1]
        c.R_mp[:] = (c.T_mm @ a.r_mp) @ c.T_pp
        c.R_pm[:] = a.r_pm
        c.T_pp[:] = a.t_pp @ c.T_pp
        c.T_mm[:] = c.T_mm @ a.t_mm
        return
    if iface == 2:
        for dn in range(nR):
            n0, n1 = get_n0_n1(S, int(rrs.i_l1l0[dn]), rrs.owned)
            c.ieJ0p[dn, n1] = _mv(a.t_pp[n1], c.ieJ0p[dn, n1] + _mv(c.ieR_pm[dn, n1], a.j0m[n0]))
            c.ieJ0m[dn, n1] = c.ieJ0m[dn, n1] + _mv(c.ieT_mm[dn, n1], a.j0m[n0])
        J0p = a.j0p + _mv(a.t_pp, c.J0p + _mv(c.R_pm, a.j0m))
        J0m = c.J0m + _mv(c.T_mm, a.j0m)
        c.J0p[:] = J0p
        c.J0m[:] = J0m
        for dn in range(nR):
            n0, n1 = get_n0_n1(S, int(rrs.i_l1l0[dn]), rrs.owned)
            c.ieT_pp[dn, n1] = a.t_pp[n1] @ c.ieT_pp[dn, n1]
            c.ieT_mm[dn, n1] = c.ieT_mm[dn, n1] @ a.t_mm[n0]
            c.ieR_pm[dn, n1] = (a.t_pp[n1] @ c.ieR_pm[dn, n1]) @ a.t_mm[n0]
        c.T_pp[:] = a.t_pp @ c.T_pp
        c.T_mm[:] = c.T_mm @ a.t_mm
        c.R_pm[:] = (a.t_pp @ c.R_pm) @ a.t_mm
        return
    # ---- ScatteringInterface_11 (:230-340)
    I = np.eye(N)[None]
    r, t_pp, t_mm, r_pm = a.r_mp, a.t_pp, a.t_mm, a.r_pm
    tmp_inv = mr.batch_inv(I - r @ c.R_pm)                                     # :244
    T01 = c.T_mm @ tmp_inv                                                     # :247
    for dn in range(nR):                                                       # :249-265
        n0, n1 = get_n0_n1(S, int(rrs.i_l1l0[dn]), rrs.owned)
        A = T01[n1] @ (a.ier_mp[dn, n1] @ c.R_pm[n0] + r[n1] @ c.ieR_pm[dn, n1]) + c.ieT_mm[dn, n1]
        c.ieJ0m[dn, n1] = c.ieJ0m[dn, n1] + \
            _mv(T01[n1], _mv(a.ier_mp[dn, n1], c.J0p[n0]) + _mv(r[n1], c.ieJ0p[dn, n1]) + a.ieJ0m[dn, n1]) + \
            _mv(A @ tmp_inv[n0], a.j0m[n0] + _mv(r[n0], c.J0p[n0]))
    J0m = c.J0m + _mv(T01, _mv(r, c.J0p) + a.j0m)                              # :267
    c.J0m[:] = J0m
    for dn in range(nR):                                                       # :269-285
        n0, n1 = get_n0_n1(S, int(rrs.i_l1l0[dn]), rrs.owned)
        A = T01[n1] @ (a.ier_mp[dn, n1] @ c.R_pm[n0] + r[n1] @ c.ieR_pm[dn, n1]) + c.ieT_mm[dn, n1]
        c.ieR_mp[dn, n1] = c.ieR_mp[dn, n1] + T01[n1] @ (a.ier_mp[dn, n1] @ c.T_pp[n0] + r[n1] @ c.ieT_pp[dn, n1]) + \
            ((A @ tmp_inv[n0]) @ r[n0]) @ c.T_pp[n0]
        c.ieT_mm[dn, n1] = T01[n1] @ a.iet_mm[dn, n1] + (A @ tmp_inv[n0]) @ t_mm[n0]
    c.R_mp[:] = c.R_mp + (T01 @ r) @ c.T_pp                                    # :288
    c.T_mm[:] = T01 @ t_mm                                                     # :290
    tmp_inv = mr.batch_inv(I - c.R_pm @ r)                                     # :295
    T21 = t_pp @ tmp_inv                                                       # :297
    for dn in range(nR):                                                       # :299-313
        n0, n1 = get_n0_n1(S, int(rrs.i_l1l0[dn]), rrs.owned)
        B = T21[n1] @ (c.ieR_pm[dn, n1] @ r[n0] + c.R_pm[n1] @ a.ier_mp[dn, n1]) + a.iet_pp[dn, n1]
        c.ieJ0p[dn, n1] = a.ieJ0p[dn, n1] + \
            _mv(T21[n1], c.ieJ0p[dn, n1] + _mv(c.ieR_pm[dn, n1], a.j0m[n0]) + _mv(c.R_pm[n1], a.ieJ0m[dn, n1])) + \
            _mv(B @ tmp_inv[n0], c.J0p[n0] + _mv(c.R_pm[n0], a.j0m[n0]))
    J0p = a.j0p + _mv(T21, c.J0p + _mv(c.R_pm, a.j0m))                         # :315
    c.J0p[:] = J0p
    for dn in range(nR):                                                       # :317-335
        n0, n1 = get_n0_n1(S, int(rrs.i_l1l0[dn]), rrs.owned)
        B = T21[n1] @ (c.ieR_pm[dn, n1] @ r[n0] + c.R_pm[n1] @ a.ier_mp[dn, n1]) + a.iet_pp[dn, n1]
        c.ieT_pp[dn, n1] = T21[n1] @ c.ieT_pp[dn, n1] + (B @ tmp_inv[n0]) @ c.T_pp[n0]
        c.ieR_pm[dn, n1] = a.ier_pm[dn, n1] + T21[n1] @ (c.ieR_pm[dn, n1] @ t_mm[n0] + c.R_pm[n1] @ a.iet_mm[dn, n1]) + \
            ((B @ tmp_inv[n0]) @ c.R_pm[n0]) @ t_mm[n0]
    c.T_pp[:] = T21 @ c.T_pp                                                   # :338
    c.R_pm[:] = r_pm + (T21 @ c.R_pm) @ t_mm                                   # :340


def fscatt_rayleigh(scene: mr.Scene) -> np.ndarray:
    """fScattRayleigh of constructCoreOpticalProperties (LayerOpticalProperties/compEffectiveLayerProperties.jl:58):
    rayl.tau ./ combo.tau with combo = Rayleigh + aerosols (before the gas absorption is merged).  [S, Nz]."""
    tau = scene.tau_rayl.astype(np.float64).copy()
    combo = tau.copy()
    for ia, aer in enumerate(scene.aerosols):
        combo = combo + ((1 - aer.ft * aer.omega) * scene.tau_aer[ia])[None, :]
    return tau / combo


def postprocessing_vza_rrs(pol, comp: CompositeLayerRS, vza, qp_mu, m, vaz, weight, R_SFI, T_SFI, ieR_SFI, ieT_SFI):
    """postprocessing_vza!(::RRS) tools/postprocessing_vza.jl:95-147 (SFI branch): the elastic sums plus the sum over
    EVERY Raman index t of ieJ0-+."""
    mr.postprocessing_vza(pol, comp, vza, qp_mu, m, vaz, weight, R_SFI, T_SFI)
    n = pol.n
    for i in range(len(vza)):
        imu = mr.nearest_point(qp_mu, float(mr.cosd(vza[i])))
        istart = imu * n
        cs = np.array([float(mr.cosd(m * vaz[i])), float(mr.cosd(m * vaz[i])), float(mr.sind(m * vaz[i])),
                       float(mr.sind(m * vaz[i]))])[:n]
        bigCS = weight * cs
        for t in range(comp.ieJ0m.shape[0]):
            ieR_SFI[i] += (bigCS[None, :] * comp.ieJ0m[t][:, istart:istart + n]).T
            ieT_SFI[i] += (bigCS[None, :] * comp.ieJ0p[t][:, istart:istart + n]).T


def rt_kernel_rrs(pol, quad, rrs: RRSInputs, Zr_pp, Zr_mp, fscatt, added: AddedLayerRS, comp: CompositeLayerRS,
                  lay: mr.LayerOptics, iface, tau_sum, m, iz, strict_idx=True, hook=None):
    """rt_kernel!(::RRS, ..., ::CoreScatteringOpticalProperties, ...) CoreKernel/rt_kernel.jl:277-340 (scatter = true)."""
    dtau, nd = mr.get_dtau_ndoubl(lay.tau, lay.varpi, quad.qp_mu)
    expk = np.exp(-dtau / quad.mu0)
    elemental_inelastic(pol, quad, rrs, fscatt, tau_sum, dtau, lay.varpi, Zr_pp, Zr_mp, m, nd, added, strict_idx)
    Zpp, Zmp = lay.Zfull()
    mr.elemental(pol, quad, tau_sum, dtau, lay.varpi, Zpp, Zmp, m, nd, added, strict_idx)
    if hook:
        hook("elemental", m, iz, added, comp)
    doubling_inelastic(pol, rrs, expk, nd, added, strict_idx)
    if hook:
        hook("doubling", m, iz, added, comp)
    if iz == 1:                                                                # :326-333
        comp.T_pp[:], comp.T_mm[:] = added.t_pp, added.t_mm
        comp.R_mp[:], comp.R_pm[:] = added.r_mp, added.r_pm
        comp.J0p[:], comp.J0m[:] = added.j0p, added.j0m
        comp.ieT_pp[:], comp.ieT_mm[:] = added.iet_pp, added.iet_mm
        comp.ieR_mp[:], comp.ieR_pm[:] = added.ier_mp, added.ier_pm
        comp.ieJ0p[:], comp.ieJ0m[:] = added.ieJ0p, added.ieJ0m
    else:
        interaction_inelastic(rrs, iface, comp, added)
    if hook:
        hook("interaction", m, iz, added, comp)
    return nd


def rt_run_rrs(scene: mr.Scene, rrs: RRSInputs, hook=None, full: bool = False):
    """rt_run(RS_type::RRS, model, iBand) rt_run.jl:41-230, SFI = true.  Returns (R_SFI, T_SFI, ieR_SFI, ieT_SFI), each
    [nVza, nStokes, S]; full=True appends the elastic RAMI extras of the same return tuple (rt_run.jl:187-213, 226): hdr
    [nVza, nStokes, S], bhr_uw, bhr_dw [nStokes, S] (interaction_hdrf! + postprocessing_vza_hdrf!).  The added / composite /
    surface layers are allocated once and persist over layers and Fourier moments like the reference's (rt_run.jl:108-116)."""
    pol, quad = scene.pol, scene.quad
    S, Nz, N = scene.S, scene.Nz, scene.N
    nR = rrs.nRaman
    nV = len(scene.vza)
    out = [np.zeros((nV, pol.n, S)) for _ in range(4)]
    hdr = np.zeros((nV, pol.n, S))
    bhr_uw, bhr_dw = np.zeros((pol.n, S)), np.zeros((pol.n, S))
    added = make_added_layer_rs(N, S, nR)
    surf = make_added_layer_rs(N, S, nR)
    comp = make_composite_layer_rs(N, S, nR)
    strict_idx = scene.strict_reference_indexing
    sinp = mr.surface_inputs(scene)
    fsc = fscatt_rayleigh(scene)
    for m in range(scene.max_m):
        weight = 0.5 if m == 0 else 1.0
        Zr_pp, Zr_mp = mr.compute_Z_moments(pol.n, quad.qp_mu, rrs.greek_raman, m)   # computeRamanZλ! inelastic_helper.jl:457-464
        layers = mr.construct_core_optical_properties(scene, m)
        ifaces, tau_sum_all = mr.extract_effective_props(layers)
        for iz in range(Nz):
            rt_kernel_rrs(pol, quad, rrs, Zr_pp, Zr_mp, fsc[:, iz], added, comp, layers[iz], ifaces[iz],
                          tau_sum_all[:, iz], m, iz + 1, strict_idx, hook)
        mr.create_surface_layer(scene, sinp, surf, m, tau_sum_all[:, -1])
        interaction_inelastic(rrs, ifaces[-1], comp, surf)
        if hook:
            hook("surface", m, Nz + 1, surf, comp)
        hdr_J0m = mr.interaction_hdrf(surf, comp, m, pol, quad, bhr_uw, bhr_dw)
        postprocessing_vza_rrs(pol, comp, scene.vza, quad.qp_mu, m, scene.vaz, weight, *out)
        dummy = mr.CompositeLayer(None, None, None, None, np.zeros_like(hdr_J0m), hdr_J0m)
        mr.postprocessing_vza(pol, dummy, scene.vza, quad.qp_mu, m, scene.vaz, weight, hdr, np.zeros_like(hdr))
    return tuple(out) + ((hdr, bhr_uw, bhr_dw) if full else ())
