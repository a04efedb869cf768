"""The N > 1 path on real devices: one process per GPU, torch.distributed backend 'nccl' (= RCCL over xGMI), every rank runs
its member block through Engine.run via ensemble.run_sharded and rank 0 receives the per-member summaries through ONE
dist.gather.  World sizes 1, 2, 4, 8 -- each skipped when the box has fewer devices (the build's GPU box has one: world size
1 still goes through init_process_group('nccl', device_id=...) and run_sharded; the 2-rank case is rehearsed over gloo in
tests/test_distributed.py on CPU and in test_gpu_stream.py::test_bench_starts_its_own_ranks on one GPU)."""

import os
import socket

import numpy as np
import pytest
import torch

import helpers
from simplyp_amd import marshal

pytestmark = pytest.mark.gpu


def _worker(rank, world, port, E, ret):
    import torch.distributed as dist
    from simplyp_amd import engine, ensemble
    os.environ['MASTER_ADDR'] = '127.0.0.1'
    os.environ['MASTER_PORT'] = str(port)
    os.environ.setdefault('HSA_ENABLE_IPC_MODE_LEGACY', '0')
    torch.cuda.set_device(rank)
    dist.init_process_group('nccl', rank=rank, world_size=world, device_id=torch.device('cuda', rank))
    try:
        m = helpers.marshal_scenario('tarland_2004_dynamic', E=E, out_mask=marshal.MASK_REACH5, solver=dict(out_slot_order=1, balance=1))
        rng = np.random.default_rng(11)                       # the same ensemble on every rank
        m['member_params'][marshal.PM_NAMES.index('fc')] *= rng.uniform(0.9, 1.1, E)
        m['member_params'][marshal.PM_NAMES.index('T_g')] *= rng.uniform(0.7, 1.3, E)
        eng = engine.get_engine(rank)
        # strong: every rank holds the whole ensemble and takes its block; weak: every rank passes only its own block
        res = ensemble.run_sharded(eng.run, m['forcing'], m['doy'], m['member_params'], m['reach_params'], m['up_ptr'], m['up_idx'], m['opts'])
        lo, hi = res['bounds']
        weak = ensemble.run_sharded(eng.run, m['forcing'], m['doy'], np.ascontiguousarray(m['member_params'][:, lo:hi]),
                                    np.ascontiguousarray(m['reach_params'][:, :, lo:hi]), m['up_ptr'], m['up_idx'], m['opts'],
                                    sharded_inputs=True, total_members=E)
        ok = weak['bounds'] == (lo, hi)
        if rank == 0:
            m['opts'].out_slot_order = 0
            full, status, _ = eng.run(m['forcing'], m['doy'], m['member_params'], m['reach_params'], m['up_ptr'], m['up_idx'], m['opts'])
            want = ensemble.member_summaries(full)            # member order
            for r in (res, weak):
                ok = ok and tuple(r['summaries'].shape) == (5, 1, E) and bool(torch.allclose(r['summaries'], want, rtol=1e-13, atol=0.0))
                ok = ok and bool(torch.equal(r['all_status'], status))
        else:
            ok = ok and res['summaries'] is None and weak['summaries'] is None
        ret.put((rank, bool(ok)))
        dist.barrier()
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize('world', [1, 2, 4, 8])
def test_nccl_sharded_run_and_gather(world):
    if torch.cuda.device_count() < world:
        pytest.skip("needs %d GPUs, this box has %d" % (world, torch.cuda.device_count()))
    import torch.multiprocessing as mp
    with socket.socket() as s:
        s.bind(('127.0.0.1', 0))
        port = s.getsockname()[1]
    ctx = mp.get_context('spawn')
    ret = ctx.Queue()
    E = 64 * world + 37                                       # ragged blocks
    procs = [ctx.Process(target=_worker, args=(r, world, port, E, ret)) for r in range(world)]
    for p in procs:
        p.start()
    results = [ret.get(timeout=600) for _ in procs]
    for p in procs:
        p.join(timeout=120)
        assert p.exitcode == 0
    assert all(ok for _, ok in results), results
