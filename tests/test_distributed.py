"""N > 1 path on CPU: two gloo ranks shard an ensemble, run their blocks, rank 0 gathers the per-member
summaries.  The compute function here is the CPU oracle (there is no GPU in this test); on the GPU box the
same run_sharded() drives Engine.run and the gather goes over RCCL."""

import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

import helpers
from simplyp_amd import ensemble, marshal


def test_shard_bounds_cover_and_balance():
    for E in (1, 7, 64, 100000, 1000003):
        for G in (1, 2, 3, 4, 8):
            blocks = [ensemble.shard_bounds(E, G, r) for r in range(G)]
            assert blocks[0][0] == 0 and blocks[-1][1] == E
            assert all(blocks[i][1] == blocks[i + 1][0] for i in range(G - 1))
            widths = [b - a for a, b in blocks]
            assert max(widths) - min(widths) <= 1
    with pytest.raises(ValueError):
        ensemble.shard_bounds(10, 2, 2)


def _oracle_run_fn(forcing, doy, mp_, rp_, up_ptr, up_idx, opts, forcing_of_member=None, out_reaches=None):
    from oracle import oracle
    out, status, stats = oracle.run(forcing, doy, np.asarray(mp_), np.asarray(rp_), up_ptr, up_idx, opts,
                                    forcing_of_member=forcing_of_member, out_reaches=out_reaches)
    return torch.from_numpy(out), torch.from_numpy(status), stats


def _worker(rank, world, port, E, ret):
    os.environ['MASTER_ADDR'] = '127.0.0.1'
    os.environ['MASTER_PORT'] = str(port)
    dist.init_process_group('gloo', rank=rank, world_size=world)
    try:
        m = helpers.marshal_scenario('confluence3_nc_2004', E=E, out_mask=marshal.MASK_REACH5,
                                     solver=dict(integrator='rk4', substeps=32))
        rng = np.random.default_rng(11)                       # same draw on every rank
        m['member_params'][marshal.PM_NAMES.index('fc')] *= rng.uniform(0.9, 1.1, E)
        m['member_params'][marshal.PM_NAMES.index('T_g')] *= rng.uniform(0.7, 1.3, E)
        res = ensemble.run_sharded(_oracle_run_fn, m['forcing'], m['doy'], m['member_params'], m['reach_params'],
                                   m['up_ptr'], m['up_idx'], m['opts'])
        lo, hi = res['bounds']
        full, status, _ = _oracle_run_fn(m['forcing'], m['doy'], m['member_params'], m['reach_params'],
                                         m['up_ptr'], m['up_idx'], m['opts'])
        # members are independent: this rank's block is bit-identical to the same members of the unsharded run
        ok = res['out'].shape[-1] == hi - lo and bool(torch.equal(res['out'], full[..., lo:hi]))
        # weak scaling: every rank brings its own block; block sizes exchanged by run_sharded, or handed in by the caller
        counts = [b - a for a, b in (ensemble.shard_bounds(E, world, r) for r in range(world))]
        for mc in (None, counts):
            weak = ensemble.run_sharded(_oracle_run_fn, m['forcing'], m['doy'], np.ascontiguousarray(m['member_params'][:, lo:hi]),
                                        np.ascontiguousarray(m['reach_params'][:, :, lo:hi]), m['up_ptr'], m['up_idx'], m['opts'],
                                        sharded_inputs=True, total_members=E, member_counts=mc)
            ok = ok and weak['bounds'] == (lo, hi) and bool(torch.equal(weak['out'], res['out']))
            ok = ok and (rank != 0 or (bool(torch.equal(weak['summaries'], res['summaries'])) and bool(torch.equal(weak['all_status'], res['all_status']))))
        try:
            ensemble.run_sharded(_oracle_run_fn, m['forcing'], m['doy'], np.ascontiguousarray(m['member_params'][:, lo:hi]),
                                 np.ascontiguousarray(m['reach_params'][:, :, lo:hi]), m['up_ptr'], m['up_idx'], m['opts'],
                                 sharded_inputs=True, member_counts=[c + 1 for c in counts])
            ok = False
        except ValueError:
            pass
        if rank == 0:
            want = ensemble.member_summaries(full)        # (a day-sum: reduction order may differ in the last bits)
            ok = ok and bool(torch.allclose(res['summaries'], want, rtol=1e-13, atol=0.0)) \
                and bool(torch.equal(res['all_status'], status))
            ret.put((rank, ok, tuple(res['summaries'].shape)))
        else:
            ret.put((rank, ok and res['summaries'] is None, None))
        dist.barrier()
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize('E', [5, 8])
def test_two_rank_sharded_run_equals_unsharded(E, oracle_lib):
    """Ragged (5 = 3 + 2) and even (8) splits: every rank's block is bit-identical to the unsharded run and
    rank 0's gathered summaries match it."""
    with socket.socket() as s:
        s.bind(('127.0.0.1', 0))
        port = s.getsockname()[1]
    ctx = mp.get_context('spawn')
    ret = ctx.Queue()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, E, ret)) for r in range(2)]
    for p in procs:
        p.start()
    results = [ret.get(timeout=120) for _ in procs]
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    by_rank = dict((r[0], r) for r in results)
    assert by_rank[0][1] is True and by_rank[0][2] == (5, 3, E)
    assert by_rank[1][1] is True


def test_gather_without_process_group_is_identity():
    t = torch.arange(12.0).reshape(3, 4)
    assert ensemble.gather_to_root(t, 4) is t
