"""Spectral sharding over the GPUs of one node.

Every spectral point is independent in the elastic path, so the concatenated spectral axis is cut
into contiguous slices, one per rank (one process per GPU).  The only quantities that couple points
-- ndoubl and the scattering-interface codes, which the reference derives from maxima over the
WHOLE axis (rt_kernel.jl:241-242, compEffectiveLayerProperties.jl:104) -- are computed by
prepare_scene() on the global axis before slicing, so an N-way run reproduces the 1-way run bit
for bit.  The single collective is an all-gather of the R/T spectra (RCCL over xGMI when the
process group backend is "nccl"; gloo in the CPU tests).

The rotational-Raman path couples spectral points at the offsets i_λ₁λ₀ (inelastic_helper.jl:13-21): element [.., n₁, Δn]
of the inelastic operators takes radiation from n₀ = n₁ + i_λ₁λ₀[Δn].  A shard therefore carries a HALO of
H = max |i_λ₁λ₀| points on each interior edge: the elastic layers are computed on the window [lo - H, hi + H) (they are
the operands at n₀), the inelastic pairs and the spectra only for the owned [lo, hi) (mom_rrs_set_shard).  No exchange
is needed during the run -- the halo is recomputed, not communicated (elastic work is 1 / nRaman of the pair work) --
and the collective stays the one all-gather of the spectra.
"""
from __future__ import annotations

from typing import Callable, Tuple

import numpy as np


def shard_bounds(S: int, world: int, rank: int) -> Tuple[int, int]:
    """Contiguous slice [lo, hi) of rank `rank`: ceil(S/world) points each, the tail may be short or empty."""
    per = -(-S // world)
    lo = min(S, rank * per)
    return lo, min(S, lo + per)


def rrs_window(S: int, world: int, rank: int, offsets) -> Tuple[int, int, int, int]:
    """(lo, hi, wlo, whi): the owned slice [lo, hi) of rank `rank` and its window [wlo, whi) = the owned slice widened by
    H = max |offsets| and clipped to the axis."""
    lo, hi = shard_bounds(S, world, rank)
    H = int(np.max(np.abs(np.asarray(offsets)))) if len(offsets) else 0
    if hi <= lo:
        return lo, hi, lo, hi
    return lo, hi, max(0, lo - H), min(S, hi + H)


def gather_spectra(local, S: int, dist, device=None):
    """All-gather per-rank slices of spectra: `local` = list of arrays whose LAST axis is this rank's owned points (possibly
    empty); returns the list of full arrays (last axis S) on every rank.  One collective for all of them."""
    import torch

    world, rank = dist.get_world_size(), dist.get_rank()
    per = -(-S // world)
    lo, hi = shard_bounds(S, world, rank)
    rows = [int(np.prod(a.shape[:-1])) for a in local]
    buf = torch.zeros((sum(rows), per), dtype=torch.float64, device=device)
    if hi > lo:
        flat = np.concatenate([np.asarray(a, dtype=np.float64).reshape(r, hi - lo) for a, r in zip(local, rows)], axis=0)
        buf[:, : hi - lo] = torch.from_numpy(np.ascontiguousarray(flat)).to(buf.device)
    out = torch.empty(world * buf.numel(), dtype=torch.float64, device=device)
    dist.all_gather_into_tensor(out, buf.reshape(-1))
    out = out.cpu().numpy().reshape(world, sum(rows), per).transpose(1, 0, 2).reshape(sum(rows), world * per)[:, :S]
    res, r0 = [], 0
    for a, r in zip(local, rows):
        res.append(out[r0:r0 + r].reshape(a.shape[:-1] + (S,)).copy())
        r0 += r
    return res


def unpack_rrs_spectra(G, nVza: int, nStokes: int, per: int, S: int = None):
    """The receive buffer of mom_allgather_rrs_device, G [world, (5 nVza nStokes + 2 nStokes) per] (per rank: [R | T | ieR |
    ieT | hdr][per, nStokes, nVza] then [bhr_uw | bhr_dw][per, nStokes], the ABI's memory order), as the 7-tuple of
    rt_run(::RRS) on the whole axis: five arrays [nVza, nStokes, S] and two [nStokes, S] (S defaults to world * per; a
    ragged tail is cut off)."""
    G = np.asarray(G)
    world = G.shape[0]
    S = world * per if S is None else S
    a, b = nVza * nStokes * per, nStokes * per
    out = []
    for k in range(5):
        x = G[:, k * a:(k + 1) * a].reshape(world * per, nStokes, nVza)
        out.append(np.transpose(x, (2, 1, 0))[..., :S].copy())
    for k in range(2):
        x = G[:, 5 * a + k * b:5 * a + (k + 1) * b].reshape(world * per, nStokes)
        out.append(x.T[..., :S].copy())
    return tuple(out)


def rt_run_sharded(scene, run_local: Callable, dist, device=None):
    """Run `run_local(shard) -> (R, T)` ([nVza, nStokes, S_loc] numpy) on this rank's slice and
    all-gather the spectra.  `dist` is torch.distributed (initialised).  Returns the full
    (R, T) [nVza, nStokes, S] on every rank."""
    import torch

    world, rank = dist.get_world_size(), dist.get_rank()
    per = -(-scene.S // world)
    lo, hi = shard_bounds(scene.S, world, rank)
    nV, nS = len(scene.node), scene.nStokes
    buf = torch.zeros((2, per, nS, nV), dtype=torch.float64, device=device)  # padded to `per` points
    if hi > lo:
        R, T = run_local(scene.spectral_slice(lo, hi))
        buf[0, : hi - lo] = torch.from_numpy(np.ascontiguousarray(R.transpose(2, 1, 0))).to(buf.device)
        buf[1, : hi - lo] = torch.from_numpy(np.ascontiguousarray(T.transpose(2, 1, 0))).to(buf.device)
    out = torch.empty(world * buf.numel(), dtype=torch.float64, device=device)
    dist.all_gather_into_tensor(out, buf.reshape(-1))  # one collective: R and T together
    out = out.cpu().numpy().reshape((world,) + tuple(buf.shape))  # [world, 2, per, nS, nV]
    full = out.transpose(1, 0, 2, 3, 4).reshape(2, world * per, nS, nV)[:, : scene.S]
    return full[0].transpose(2, 1, 0).copy(), full[1].transpose(2, 1, 0).copy()


def slice_partials(partials, lo: int, hi: int):
    """The spectral slice [lo, hi) of a list of corert.ScenePartial (per-point arrays cut along the spectral axis; the partials of
    the phase-matrix bases, of the scalar albedo and of the BRDF matrices are the same for every point)."""
    from dataclasses import replace
    cut = lambda a, ax: None if a is None else np.ascontiguousarray(np.take(np.asarray(a), range(lo, hi), axis=ax))
    return [replace(p, dτ=cut(p.dτ, 0), dϖ=cut(p.dϖ, 0), dzw=cut(p.dzw, 1), dalbedo_spec=cut(p.dalbedo_spec, 0)) for p in partials]


def rt_run_dual_sharded(scene, partials, run_local: Callable, dist, device=None):
    """rt_run on Dual numbers over the ranks of a node: every spectral point and each of its partials is independent, so rank r
    runs `run_local(shard, partials_of_the_shard) -> (R, T, dR, dT)` on its slice ([nVza, nStokes, S_loc] and [P, nVza, nStokes,
    S_loc]; ndoubl / interface codes stay the global ones of `scene`) and ONE all-gather (gather_spectra) assembles the four
    arrays on every rank."""
    world, rank = dist.get_world_size(), dist.get_rank()
    lo, hi = shard_bounds(scene.S, world, rank)
    nV, nS, P = len(scene.node), scene.nStokes, len(partials)
    if hi > lo:
        loc = list(run_local(scene.spectral_slice(lo, hi), slice_partials(partials, lo, hi)))
    else:
        loc = [np.zeros((nV, nS, 0)), np.zeros((nV, nS, 0)), np.zeros((P, nV, nS, 0)), np.zeros((P, nV, nS, 0))]
    return tuple(gather_spectra(loc, scene.S, dist, device))
