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


# ---- several GPUs from ONE process: run_simply_p_ensemble(devices=[...]) (SURVEY.md section 8b's batched entry) ----------------

def _chain4_call(**kw):
    import simplyp_amd as sp
    met, p_struc, p_SU, p_LU, p_SC, p, dyn = helpers.scenario_inputs('chain4_val_2004')
    E = 203                                                   # ragged blocks
    rng = np.random.default_rng(5)
    over = {'fc': 290 * rng.uniform(0.8, 1.2, E), 'T_g': 65 * rng.uniform(0.7, 1.3, E),
            'L_reach': np.array([[3000.], [5000.], [7000.], [10000.]]) * rng.uniform(0.7, 1.3, (1, E)),
            'f_TDP': rng.uniform(0.5, 0.9, E)}
    return sp.run_simply_p_ensemble(met, p_struc, p_SU, p_LU, p_SC, p, dyn, overrides=over, **kw)


@pytest.mark.parametrize('devices', [[0, 0], [0, 0, 0]])
def test_devices_list_in_one_process_equals_one_device(devices):
    """Two (three) engine contexts driven by host threads of one process -- on this box both on GPU 0; on an 8-GPU node
    devices=[0..7] -- against the plain device=0 call: daily table, status, goodness of fit and waterbody bit-identical."""
    if torch.cuda.device_count() <= max(devices):
        pytest.skip("needs GPU %d" % max(devices))
    met = helpers.scenario_inputs('chain4_val_2004')[0]
    rng = np.random.default_rng(9)
    obs = {4: __import__('pandas').DataFrame({'Q': rng.uniform(0.1, 2.0, len(met)), 'TDP': rng.uniform(0.01, 0.1, len(met))},
                                             index=met.index)}
    kw = dict(out_reaches=[2, 4], obs_dict=obs, waterbody=[2, 4])
    one = _chain4_call(**kw)
    many = _chain4_call(devices=devices, **kw)
    assert many['stats']['bounds'][0][0] == 0 and many['stats']['bounds'][-1][1] == 203 and len(many['stats']['per_device']) == len(devices)
    assert many['data'].shape == one['data'].shape and np.array_equal(many['data'], one['data'], equal_nan=True)
    assert np.array_equal(many['status'], one['status'])
    assert np.array_equal(many['gof']['data'], one['gof']['data'], equal_nan=True)
    assert np.array_equal(many['waterbody']['data'], one['waterbody']['data'], equal_nan=True)
    assert many['stats']['rhs_evals'] == one['stats']['rhs_evals']


def test_devices_list_on_every_gpu_of_the_box():
    """devices = every GPU this box has (1 here, 8 on the driver's node): same table as device=0."""
    n = torch.cuda.device_count()
    one = _chain4_call(out_reaches=[4])
    many = _chain4_call(out_reaches=[4], devices=list(range(n)))
    assert np.array_equal(many['data'], one['data'], equal_nan=True) and len(many['stats']['bounds']) == n


def test_devices_list_device_resident_blocks():
    res = _chain4_call(out_reaches=[4], devices=[0, 0], to_host=False)
    assert isinstance(res['data'], list) and [t.shape[-1] for t in res['data']] == [102, 101]
    one = _chain4_call(out_reaches=[4], to_host=False)
    assert bool(torch.equal(torch.cat(res['data'], dim=-1), one['data']))
