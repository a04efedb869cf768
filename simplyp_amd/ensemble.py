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


def _group_is_nccl(group=None):
    import torch.distributed as dist
    try:
        return 'nccl' in str(dist.get_backend(group))
    except Exception:
        return False


def gather_to_root(local, n_members, group=None, dst=0, widths=None):
    """Gather per-member data (member axis last) from every rank to ``dst``.

    ``local`` is this rank's block (torch tensor); blocks may be ragged, so they are padded to the widest block for
    the collective.  ``widths``: members per rank (default: the contiguous split of ``n_members`` by
    ``shard_bounds``).  With an ``nccl`` group (RCCL over xGMI) the blocks travel device to device; with ``gloo`` they
    are staged through host memory.  Returns the assembled ``[..., n_members]`` tensor on ``dst`` (on the device
    ``local`` lives on) and ``None`` elsewhere.  With world size 1 (or torch.distributed not initialised) it returns
    ``local`` unchanged.
    """
    import torch
    import torch.distributed as dist
    if not (dist.is_available() and dist.is_initialized()) or dist.get_world_size(group) == 1:
        return local
    world, rank = dist.get_world_size(group), dist.get_rank(group)
    if widths is None:
        widths = [b - a for a, b in (shard_bounds(n_members, world, r) for r in range(world))]
    if len(widths) != world or sum(widths) != n_members:
        raise ValueError("widths %s do not add up to %d members over %d ranks" % (widths, n_members, world))
    wmax = max(widths)
    if local.shape[-1] != widths[rank]:
        raise ValueError("rank %d holds %d members, expected %d" % (rank, local.shape[-1], widths[rank]))
    home = local.device
    wire = local if (_group_is_nccl(group) or home.type == 'cpu') else local.cpu()
    pad = torch.zeros(wire.shape[:-1] + (wmax,), dtype=wire.dtype, device=wire.device)
    pad[..., :widths[rank]] = wire
    pad = pad.contiguous()
    bufs = [torch.empty_like(pad) for _ in range(world)] if rank == dst else None
    dist.gather(pad, gather_list=bufs, dst=dst, group=group)
    if rank != dst:
        return None
    return torch.cat([b[..., :w] for b, w in zip(bufs, widths)], dim=-1).to(home)


def run_sharded(run_fn, forcing, doy, member_params, reach_params, up_ptr, up_idx, opts,
                forcing_of_member=None, out_reaches=None, group=None, gather=True, sharded_inputs=False,
                total_members=None, member_counts=None, **run_kwargs):
    """Run this rank's member block through ``run_fn`` (normally ``Engine.run``) and gather the
    per-member summaries on rank 0.

    Default: ``member_params`` / ``reach_params`` / ``forcing_of_member`` hold the WHOLE ensemble on every rank and
    each rank takes its contiguous block (strong scaling: a fixed ensemble split over the GPUs).
    ``sharded_inputs=True``: the arrays already are this rank's own block (weak scaling: every rank brings its own
    members); the global ensemble is the concatenation of the blocks in rank order, ``total_members`` its size
    (default: the sum over the ranks, found with one all_gather of the block sizes; ``member_counts`` -- members per rank,
    the same list on every rank -- saves that exchange when the caller knows it, e.g. inside a timed loop).
    Extra keyword arguments (``out=``, ``host_out=``, ``member_of_slot=`` ...) go to ``run_fn``.

    Returns ``dict(bounds, out, status, stats, summaries, all_status)``; ``summaries`` ([n_cols, n_reaches, E_total],
    the day-sums of every requested column, in MEMBER order whatever the order of the columns of ``out``) and
    ``all_status`` are the gathered tensors on rank 0 (``None`` on other ranks or when ``gather`` is False).
    """
    import torch
    import torch.distributed as dist
    if dist.is_available() and dist.is_initialized():
        world, rank = dist.get_world_size(group), dist.get_rank(group)
    else:
        world, rank = 1, 0
    widths = None
    if sharded_inputs:
        mp, rp, fom = member_params, reach_params, forcing_of_member
        e_local = int(mp.shape[-1])
        if member_counts is not None:
            widths = [int(x) for x in member_counts]
            if len(widths) != world or widths[rank] != e_local:
                raise ValueError("member_counts %s do not describe rank %d of %d holding %d members" % (widths, rank, world, e_local))
        elif world > 1:
            sizes = [None] * world
            dist.all_gather_object(sizes, e_local, group=group)
            widths = [int(x) for x in sizes]
        else:
            widths = [e_local]
        E = sum(widths)
        if total_members is not None and int(total_members) != E:
            raise ValueError("total_members=%d but the ranks hold %d members" % (total_members, E))
        lo = sum(widths[:rank])
        bounds = (lo, lo + e_local)
    else:
        E = member_params.shape[-1]
        mp, rp, fom, bounds = shard_arrays(member_params, reach_params, forcing_of_member, world, rank)
        mp = mp.contiguous() if hasattr(mp, 'contiguous') else np.ascontiguousarray(mp)
        rp = rp.contiguous() if hasattr(rp, 'contiguous') else np.ascontiguousarray(rp)
    out, status, stats = run_fn(forcing, doy, mp, rp, up_ptr, up_idx, opts, forcing_of_member=fom,
                                out_reaches=out_reaches, **run_kwargs)
    res = dict(bounds=bounds, out=out, status=status, stats=stats, summaries=None, all_status=None)
    if gather:
        summ = member_summaries(out)
        mos = stats.get('member_of_slot') if isinstance(stats, dict) else None
        if mos is not None and getattr(opts, 'out_slot_order', 0):
            by_member = torch.empty_like(summ)            # columns of `out` are lane slots: back to member order
            by_member[..., mos.long()] = summ
            summ = by_member
        fin = stats.pop('finish', None) if isinstance(stats, dict) else None
        if fin is not None:
            # deferred run (Engine.run(defer_sync=True)): the day-sums above were enqueued behind the kernel and run beside the
            # tail of the streamed copies; now wait for the run (and the last byte of the host table) and take its statistics
            stats.update(fin())
        res['summaries'] = gather_to_root(summ, E, group, widths=widths)
        res['all_status'] = gather_to_root(status, E, group, widths=widths)
    else:
        fin = stats.pop('finish', None) if isinstance(stats, dict) else None
        if fin is not None:
            stats.update(fin())
    return res
