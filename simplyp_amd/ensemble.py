"""Ensemble sharding across the GPUs of one node.

Members are independent (no reference state couples two parameter sets; the only coupling is
reach -> reach inside a member, model.py:508-544), so the ensemble axis shards with no
data-path collective: rank r integrates the contiguous member block
``[r*E/G, (r+1)*E/G)`` on its own GPU, with forcing and topology replicated.  The single
exchange step is the final gather of small per-member summaries to rank 0 -- over RCCL/xGMI
when the process group's backend is ``nccl``, over gloo in the CPU tests.
"""

import numpy as np


def shard_bounds(n_members, world_size, rank):
    """Contiguous block of members owned by ``rank``: sizes differ by at most one."""
    if world_size < 1 or not (0 <= rank < world_size):
        raise ValueError("bad rank %d / world size %d" % (rank, world_size))
    base, extra = divmod(int(n_members), int(world_size))
    lo = rank * base + min(rank, extra)
    hi = lo + base + (1 if rank < extra else 0)
    return lo, hi


def shard_arrays(member_params, reach_params, forcing_of_member, world_size, rank):
    """Slice the ensemble-major arrays (member axis last) for one rank."""
    E = member_params.shape[-1]
    lo, hi = shard_bounds(E, world_size, rank)
    fom = None if forcing_of_member is None else forcing_of_member[lo:hi]
    return member_params[..., lo:hi], reach_params[..., lo:hi], fom, (lo, hi)


def member_summaries(out, step_axis=1):
    """Per-member sums over the day axis of every requested column: [n_cols, n_reaches, E].
    (For the flux columns this is the total mass/volume exported over the run.)"""
    return out.sum(dim=step_axis)


def gather_to_root(local, n_members, group=None, dst=0):
    """Gather per-member data (member axis last) from every rank to ``dst``.

    ``local`` is this rank's block (torch tensor, any device the backend supports); blocks may be
    ragged by one member, so they are padded to the widest block for the collective.  Returns the
    assembled ``[..., n_members]`` tensor on ``dst`` and ``None`` elsewhere.  With world size 1 (or
    torch.distributed not initialised) it returns ``local`` unchanged.
    """
    import torch
    import torch.distributed as dist
    if not (dist.is_available() and dist.is_initialized()) or dist.get_world_size(group) == 1:
        return local
    world, rank = dist.get_world_size(group), dist.get_rank(group)
    widths = [b - a for a, b in (shard_bounds(n_members, world, r) for r in range(world))]
    wmax = max(widths)
    if local.shape[-1] != widths[rank]:
        raise ValueError("rank %d holds %d members, expected %d" % (rank, local.shape[-1], widths[rank]))
    pad = torch.zeros(local.shape[:-1] + (wmax,), dtype=local.dtype, device=local.device)
    pad[..., :widths[rank]] = local
    pad = pad.contiguous()
    bufs = [torch.empty_like(pad) for _ in range(world)] if rank == dst else None
    dist.gather(pad, gather_list=bufs, dst=dst, group=group)
    if rank != dst:
        return None
    return torch.cat([b[..., :w] for b, w in zip(bufs, widths)], dim=-1)


def run_sharded(run_fn, forcing, doy, member_params, reach_params, up_ptr, up_idx, opts,
                forcing_of_member=None, out_reaches=None, group=None, gather=True):
    """Run this rank's member block through ``run_fn`` (normally ``Engine.run``) and gather the
    per-member summaries on rank 0.

    Returns ``dict(bounds, out, status, stats, summaries, all_status)``; the last two are the
    gathered tensors on rank 0 (``None`` on other ranks or when ``gather`` is False).
    """
    import torch.distributed as dist
    if dist.is_available() and dist.is_initialized():
        world, rank = dist.get_world_size(group), dist.get_rank(group)
    else:
        world, rank = 1, 0
    E = member_params.shape[-1]
    mp, rp, fom, bounds = shard_arrays(member_params, reach_params, forcing_of_member, world, rank)
    mp = mp.contiguous() if hasattr(mp, 'contiguous') else np.ascontiguousarray(mp)
    rp = rp.contiguous() if hasattr(rp, 'contiguous') else np.ascontiguousarray(rp)
    out, status, stats = run_fn(forcing, doy, mp, rp, up_ptr, up_idx, opts, forcing_of_member=fom,
                                out_reaches=out_reaches)
    res = dict(bounds=bounds, out=out, status=status, stats=stats, summaries=None, all_status=None)
    if gather:
        res['summaries'] = gather_to_root(member_summaries(out), E, group)
        res['all_status'] = gather_to_root(status, E, group)
    return res
