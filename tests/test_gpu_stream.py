"""Streamed output (simplyp_stream_out) and the BASELINE configurations at their stated sizes.

The reference produces its 25 values per catchment-day in host memory (model.py:644, :721-724); the engine delivers an
ensemble's table there chunk by chunk while later chunks compute.  Whatever the path (time-chunk task queue with the
copy beside the kernel; chain kernel / RK4 / time-reduced rows with the copy behind the last launch), the host table
must equal the device table bit for bit, and the device table must be what a run without streaming writes.
"""

import ctypes as C
import os
import subprocess
import sys

import numpy as np
import pytest

import helpers
from simplyp_amd import abi, engine, marshal, synthetic

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
REACH_COLS = ['Vr', 'Qr_EndOfDay', 'Qr', 'Msus_EndOfDay', 'Msus_kg/day', 'TDPr_EndOfDay', 'TDP_kg/day',
              'PPr_EndOfDay', 'PP_kg/day']


def run(eng, m, **kw):
    return eng.run(m['forcing'], m['doy'], m['member_params'], m['reach_params'], m['up_ptr'], m['up_idx'], m['opts'], **kw)


def perturbed(name, E, seed=3, **kw):
    m = helpers.marshal_scenario(name, E=E, **kw)
    rng = np.random.default_rng(seed)
    m['member_params'][marshal.PM_NAMES.index('fc')] *= rng.uniform(0.85, 1.15, E)
    m['member_params'][marshal.PM_NAMES.index('T_g')] *= rng.uniform(0.6, 1.5, E)
    m['member_params'][marshal.PM_NAMES.index('a_Q')] *= rng.uniform(0.6, 1.6, E)
    return m


def raw_run(eng, m, out=None):
    """simplyp_run straight through the C ABI (device buffers via torch), WITHOUT Engine.run's own simplyp_stream_out call:
    what a C caller does.  Returns (rc, out tensor)."""
    import torch
    L = engine.lib()
    o = m['opts']
    dev = [eng.to_device(m['forcing'], torch.float64), eng.to_device(m['doy'], torch.int32),
           eng.to_device(m['member_params'], torch.float64), eng.to_device(m['reach_params'], torch.float64)]
    n_sets, _, D = dev[0].shape
    _, S, E = dev[3].shape
    if out is None:
        out = torch.empty((bin(o.out_mask).count('1'), D, S, E), dtype=torch.float64, device=eng.tdev)
    status = torch.empty((E,), dtype=torch.int32, device=eng.tdev)
    up_ptr = np.ascontiguousarray(m['up_ptr'], dtype=np.int32); up_idx = np.ascontiguousarray(m['up_idx'], dtype=np.int32)
    ip = lambda a: None if a.size == 0 else a.ctypes.data_as(C.POINTER(C.c_int32))
    torch.cuda.synchronize()
    dims = abi.Dims(E, S, D, n_sets)
    stats = abi.Stats()
    assert L.simplyp_ctx_set_stream(eng._h, None) == 0
    rc = L.simplyp_run(eng._h, C.byref(dims), C.byref(o), dev[0].data_ptr(), dev[1].data_ptr(), None, None, dev[2].data_ptr(),
                       dev[3].data_ptr(), ip(up_ptr), ip(up_idx), None, S, out.data_ptr(), status.data_ptr(), None, None,
                       C.byref(stats))
    return rc, out


@pytest.mark.parametrize('name,E,solver,expect_queue', [
    ('tarland_1981_2010_dynamic', 700, None, True),                             # 172 chunks of 64 days, copies beside the kernel
    ('tarland_1981_2010_dynamic', 130, dict(time_chunk_days=512), True),        # longer chunks
    ('tarland_2004_dynamic', 200, None, True),                                  # 366 days: 6 chunks, the last one ragged
    ('tarland_2004_dynamic', 200, dict(time_chunk_days=-1), False),             # chain kernel: whole table after the launch
    ('tarland_2004_dynamic', 70, dict(integrator='rk4', substeps=16), False),   # RK4 never runs through the queue
    ('chain4_val_2004', 150, None, True),                                       # reach network, pipelined queue, 4 reaches out
    ('confluence3_nc_2004', 96, dict(balance=1), True),                         # cost-ordered slots, member-order table
])
def test_host_table_equals_device_table(engine0, name, E, solver, expect_queue):
    m = perturbed(name, E, out_mask=marshal.MASK_REACH5, solver=solver)
    ref, ref_status, _ = run(engine0, m)                                        # no streaming
    host = engine.pinned_empty(tuple(ref.shape))
    host[...] = -7.0
    out, status, st = run(engine0, m, host_out=host)
    assert st['queued'] == (1 if expect_queue else 0)
    assert np.array_equal(host, out.cpu().numpy(), equal_nan=True)
    import torch
    assert bool(torch.equal(out, ref)) and bool(torch.equal(status, ref_status))       # streaming does not change results
    if expect_queue:
        n_chunks = -(-m['forcing'].shape[2] // 64)          # auto: 64-day chunks for a single reach, 256 for networks / as given
        assert 0 <= st['streamed_chunks'] <= n_chunks
    else:
        assert st['streamed_chunks'] == 0
    assert st['wall_ms'] > 0 and st['d2h_tail_ms'] >= 0
    # the arm is one-shot: the next run leaves the host buffer alone
    host[...] = -7.0
    run(engine0, m)
    assert float(host.max()) == -7.0 and float(host.min()) == -7.0


def test_copies_overlap_the_kernel_on_a_long_run(engine0):
    """30 years x 4096 members: all but the last few of the 172 64-day chunks must have gone out while the kernel was still
    running, and what the copy adds after the last launch is a small part of the run."""
    m = perturbed('tarland_1981_2010_dynamic', 4096, out_mask=marshal.MASK_REACH5)
    shape = (5, m['forcing'].shape[2], 1, 4096)
    host = engine.pinned_empty(shape)
    run(engine0, m, host_out=host)                                              # warm-up (allocations)
    out, status, st = run(engine0, m, host_out=host)
    assert st['queued'] == 1 and st['streamed_chunks'] >= 150, st
    assert st['d2h_tail_ms'] < 0.25 * st['kernel_ms'], st
    assert np.array_equal(host, out.cpu().numpy(), equal_nan=True)


def test_time_reduced_rows_and_slot_order_are_streamed_too(engine0):
    m = perturbed('tarland_1981_2010_dynamic', 300, out_mask=marshal.mask_of_columns(['Qr', 'TDP_kg/day']),
                  solver=dict(out_slot_order=1, balance=1))
    periods, pod = np.unique(m['met'].index.year.values, return_inverse=True)
    m['opts'].n_periods = len(periods)
    host = engine.pinned_empty((2, len(periods), 1, 300))
    out, status, st = run(engine0, m, period_of_day=np.ascontiguousarray(pod, dtype=np.int32), host_out=host)
    assert st['streamed_chunks'] == 0                      # sums are complete only when the run is
    assert np.array_equal(host, out.cpu().numpy())
    assert sorted(st['member_of_slot'].cpu().numpy().tolist()) == list(range(300))


def test_pageable_host_buffer_and_argument_errors(engine0):
    m = perturbed('tarland_2004_dynamic', 64, out_mask=marshal.MASK_REACH5)
    host = np.full((5, 366, 1, 64), -1.0)                  # not pinned: slower, still correct
    out, _, _ = run(engine0, m, host_out=host)
    assert np.array_equal(host, out.cpu().numpy())
    with pytest.raises(ValueError):
        run(engine0, m, host_out=np.zeros((5, 366, 1, 63)))
    with pytest.raises(ValueError):
        run(engine0, m, host_out=np.zeros((5, 366, 1, 64), dtype=np.float32))
    # straight through the C ABI: a host buffer smaller than the table is refused at the run, and the arm is spent
    L = engine.lib()
    small = engine.pinned_empty((100,))
    assert L.simplyp_stream_out(engine0._h, C.c_void_p(small.ctypes.data), C.c_int64(small.nbytes)) == 0
    rc, _ = raw_run(engine0, m)
    assert rc == -1 and b'smaller than the output table' in L.simplyp_last_error(engine0._h)
    rc, out2 = raw_run(engine0, m)
    import torch
    assert rc == 0 and bool(torch.equal(out, out2))


def test_a_refused_run_spends_the_arm(engine0):
    """simplyp_stream_out is one-shot whatever becomes of the next run.  A run refused for its arguments (rtol <= 0:
    check_args) must not leave the arm behind -- the caller drops the buffer, and the next plain run on the cached context
    would stream its whole table into freed memory.  Straight through the C ABI (Engine.run disarms by itself)."""
    import torch
    L = engine.lib()
    m = perturbed('tarland_2004_dynamic', 64, out_mask=marshal.MASK_REACH5)
    host = engine.pinned_empty((5, 366, 1, 64))
    host[...] = -3.0
    assert L.simplyp_stream_out(engine0._h, C.c_void_p(host.ctypes.data), C.c_int64(host.nbytes)) == 0
    good_rtol = m['opts'].rtol
    m['opts'].rtol = -1.0
    rc, _ = raw_run(engine0, m)
    assert rc == -1 and b'rtol' in L.simplyp_last_error(engine0._h)
    m['opts'].rtol = good_rtol
    rc, out = raw_run(engine0, m)                              # a plain run: nothing armed any more
    assert rc == 0 and bool(torch.isfinite(out).all())
    assert float(host.max()) == -3.0 and float(host.min()) == -3.0
    # and the Engine wrapper disarms whatever another user of the handle left armed
    assert L.simplyp_stream_out(engine0._h, C.c_void_p(host.ctypes.data), C.c_int64(host.nbytes)) == 0
    out2, _, _ = run(engine0, m)
    assert bool(torch.equal(out, out2)) and float(host.max()) == -3.0 and float(host.min()) == -3.0


def test_queue_waits_are_bounded_by_lack_of_progress_not_by_time(engine0, monkeypatch):
    """The task queue's dependency waits fail only when NO task of the run completes for SIMPLYP_QUEUE_MAX_POLLS polls -- not
    when a wait is merely long.  One member group, 172 chunks of 64 days, streamed: the 172 tasks are taken by 172 waves at
    once and form a serial chain -- the wave that holds the last chunk waits for 171 predecessors (~0.2 s, ~100 000 polls of
    ~2 us), while a task completes every ~1 ms (~500 polls).  With the bound at 3 000 polls (several tasks, a thirtieth of the
    longest wait) the run must complete and equal the default run bit for bit; its statistics show a wait far longer than the
    bound (which the poll-counting rule of rounds 1-2 would have failed with "timed out") and no stall anywhere near it.  A
    bound below one task's duration is a genuine stall: the error path (every worker drains, simplyp_sync reports it, the
    context stays usable)."""
    import torch
    m = perturbed('tarland_1981_2010_dynamic', 8, out_mask=marshal.MASK_REACH5, solver=dict(time_chunk_days=64, lanes_per_wave=8))
    host = engine.pinned_empty((5, m['forcing'].shape[2], 1, 8))
    ref, ref_status, st0 = run(engine0, m, host_out=host)
    assert st0['queued'] == 1 and st0['queue_waits'] >= 100 and int(ref_status.max()) == 0, st0
    monkeypatch.setenv('SIMPLYP_QUEUE_MAX_POLLS', '3000')
    host[...] = -1.0
    out, status, st = run(engine0, m, host_out=host)
    assert st['queued'] == 1 and bool(torch.equal(out, ref)) and bool(torch.equal(status, ref_status))
    assert np.array_equal(host, ref.cpu().numpy(), equal_nan=True)
    assert st['queue_longest_wait_polls'] > 10 * 3000, st      # a wait that outlasted the bound many times over ...
    assert st['queue_longest_stall_polls'] <= 3000, st         # ... while tasks kept completing
    monkeypatch.setenv('SIMPLYP_QUEUE_MAX_POLLS', '5')
    with pytest.raises(engine.EngineError, match='no task of the run completed'):
        run(engine0, m)
    monkeypatch.delenv('SIMPLYP_QUEUE_MAX_POLLS')
    out2, status2, st2 = run(engine0, m, host_out=host)        # the context is usable afterwards
    assert bool(torch.equal(out2, ref)) and st2['queue_longest_stall_polls'] > 5, st2


def test_inputs_still_in_flight_on_torchs_default_stream(engine0):
    """torch's default stream has handle 0, which the library reads as "private stream"; a cast launched on
    the default stream just before the run must be finished before the kernels read its result."""
    import torch
    m = perturbed('tarland_2004_dynamic', 4096, out_mask=marshal.MASK_REACH5)
    want, _, _ = run(engine0, m)
    big = torch.from_numpy(m['member_params']).to('cuda')
    for _ in range(3):
        # a long-running producer on the default stream: the parameters are the LAST thing it writes
        junk = torch.randn(8192, 8192, device='cuda')
        junk = junk @ junk
        mp_dev = (big.to(torch.float32).to(torch.float64) * 0 + big) + 0 * junk[0, 0].double()
        got, _, _ = engine0.run(m['forcing'], m['doy'], mp_dev, m['reach_params'], m['up_ptr'], m['up_idx'], m['opts'])
        assert bool(torch.equal(got, want))
    s = torch.cuda.Stream()
    with torch.cuda.stream(s):                             # a real (non-default) torch stream is used as is
        junk = torch.randn(8192, 8192, device='cuda')
        junk = junk @ junk
        mp_dev = big + 0 * junk[0, 0].double()
        got, _, _ = engine0.run(m['forcing'], m['doy'], mp_dev, m['reach_params'], m['up_ptr'], m['up_idx'], m['opts'])
    s.synchronize()
    assert bool(torch.equal(got, want))


# ---------------------------------------------------------------------------------------------------
# BASELINE configurations at their stated sizes

def test_config_c2_exactly(engine0):
    """C2 as SURVEY.md section 8(d) states it: Tarland 1981-2010 (10 957 days), 1024 identical copies of the base
    parameters, all 25 columns (2.24 GB), delivered to host memory: every member equals the E = 1 run bit for bit,
    and the base member meets the <= 1e-6 bar on the 9 reach outputs against the reference's odeint(rtol=atol=1e-12)
    tables (tests/golden/tarland_1981_2010_dynamic.npz)."""
    import torch
    name = 'tarland_1981_2010_dynamic'
    m = helpers.marshal_scenario(name, E=1024)
    D = m['forcing'].shape[2]
    assert D == 10957 and bin(m['opts'].out_mask).count('1') == 25
    host = engine.pinned_empty((25, D, 1, 1024))
    out, status, st = run(engine0, m, host_out=host)
    assert int(status.max()) == 0 and out.numel() * 8 == 25 * 10957 * 1024 * 8
    one = helpers.marshal_scenario(name, E=1)
    o1, _, _ = run(engine0, one)
    assert bool(torch.equal(out, o1.expand(-1, -1, -1, 1024)))
    assert np.array_equal(host, out.cpu().numpy(), equal_nan=True)
    gold = helpers.golden_tables(name, 'tight')['R'][1]
    got = o1[..., 0, 0].cpu().numpy()
    for c in REACH_COLS:
        rel = np.abs(got[marshal.OUT_COLUMNS.index(c)] - gold[c].values) / np.abs(gold[c].values)
        assert rel.max() < 1e-6, (c, rel.max())


def test_config_c3_at_100k_members_through_the_benched_path(engine0, oracle_lib):
    """C3 at its stated size: 100 000 Monte-Carlo members x 10 957 days, REACH-5, slot-order table, auto load balance +
    task queue (the configuration bench.py times), table streamed to the host.  No member flagged; every member in
    exactly one slot; seeded members pulled from the big table agree with the CPU oracle to 10 x rtol; a 4096-member
    slice run on its own (chain kernel, no balancing) equals the big run bit for bit; host table == device table on a
    row sample."""
    import torch
    E = 100000
    pr = synthetic.c3_problem(E, solver=dict(out_slot_order=1))
    D = pr['forcing'].shape[2]
    host = engine.pinned_empty((5, D, 1, E))
    out, status, st = run(engine0, pr, host_out=host)
    assert st['queued'] == 1 and st['balanced'] == 1 and st['streamed_chunks'] >= 120, st
    assert int((status != 0).sum()) == 0 and bool(torch.isfinite(out).all())
    mos = st['member_of_slot']
    assert int(torch.unique(mos).numel()) == E
    slot_of = torch.empty(E, dtype=torch.long, device=out.device)
    slot_of[mos.long()] = torch.arange(E, device=out.device)
    rng = np.random.default_rng(2024)
    pick = np.sort(rng.choice(E, 8, replace=False))
    sub = dict(pr, member_params=pr['member_params'][:, pick], reach_params=pr['reach_params'][:, :, pick])
    ref, _, _ = oracle_lib.run(sub['forcing'], sub['doy'], sub['member_params'], sub['reach_params'], sub['up_ptr'],
                               sub['up_idx'], sub['opts'], n_threads=8)
    got = out[..., slot_of[torch.as_tensor(pick, device=out.device)]].cpu().numpy()
    assert helpers.max_rel_err(got, ref, floor=1e-12) < helpers.TOL_WORKING
    sl = np.arange(4096) * 24
    alone = dict(pr, member_params=pr['member_params'][:, sl], reach_params=pr['reach_params'][:, :, sl])
    alone['opts'] = abi.make_opts(dynamic_epc0=True, out_mask=marshal.MASK_REACH5)
    a = run(engine0, alone)[0]
    assert bool(torch.equal(a, out[..., slot_of[torch.as_tensor(sl, device=out.device)]]))
    rows = np.arange(0, D, 97)
    assert bool(torch.equal(torch.from_numpy(np.ascontiguousarray(host[:, rows])).to(out.device), out[:, rows]))


def test_bench_starts_its_own_ranks(tmp_path):
    """`python bench.py --gpus 2` from a plain shell: the parent must spawn its ranks before touching the GPU (gloo
    rehearsal: two ranks share this box's one GPU) and relay rank 0's ONE JSON line.  Default scaling = strong (BASELINE C3 is
    one ensemble split over the GPUs): members_total stays what was asked for; the line carries cpu_baseline at world size 2,
    per-rank kernel times and the collective's rank count (0 here: gloo rehearsal, no RCCL).  The tables are streamed to
    pinned host memory by both ranks at once -- two PROCESSES whose persistent task-queue kernels share one GPU: slow (the
    waves of both kernels compete for the same SIMDs), which the queue's progress-based wait bound tolerates (round 2 had a
    poll-count bound and this rehearsal once failed with a wait "timeout")."""
    import json
    env = dict(os.environ, SIMPLYP_BENCH_BACKEND='gloo')
    env.pop('WORLD_SIZE', None); env.pop('RANK', None); env.pop('LOCAL_RANK', None)
    for scaling, per_gpu, total in ((None, 1500, 3000), ('weak', 1500, 3000)):
        members = per_gpu if scaling == 'weak' else total
        cmd = [sys.executable, os.path.join(ROOT, 'bench.py'), '--gpus', '2', '--steps', '1', '--warmup', '0', '--members', str(members)]
        cmd += ['--scaling', scaling, '--no-cpu-baseline'] if scaling else ['--secondary-scale', '0.002']
        r = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=900)
        assert r.returncode == 0, r.stderr[-2000:]
        lines = [l for l in r.stdout.splitlines() if l.startswith('{')]
        assert len(lines) == 1
        j = json.loads(lines[0])
        assert j['n_gpus'] == 2 and j['scaling'] == (scaling or 'strong')
        assert j['members_total'] == j['config']['members_total'] == total and j['config']['members_per_gpu'] == per_gpu
        assert j['value'] > 0 and j['parity']['timed_run_sample']['max_rel_err_vs_oracle'] < helpers.TOL_WORKING
        assert j['parity']['timed_run_sample']['host_table_equals_device_table'] is True and j['value_device_resident'] > 0
        assert len(j['per_rank']['kernel_ms']) == 2 and min(j['per_rank']['kernel_ms']) > 0
        assert j['rccl_ranks'] == 0 and 'gloo' in j['collective_backend']
        if scaling is None:
            assert j['cpu_baseline']['value'] > 0 and j['cpu_baseline']['cores'] >= 1
            assert j['parity']['golden']['knee_members']['worst_member_max_rel_err'] < 1e-6
            assert j['parity']['golden']['heldout_members']['worst_member_max_rel_err'] < 5e-7
            assert j['parity']['golden']['wide_members']['worst_member_max_rel_err'] < 5e-7
            assert j['parity']['golden']['dry_members']['worst_member_max_rel_err'] < j['parity']['golden']['dry_members']['bar']
            assert j['cpu_baseline']['cores'] == j['cpu_baseline']['usable_cores'] and j['cpu_baseline']['reference_python']['value'][0] > 100
            # the secondary legs at rehearsal size (--secondary-scale): the leg that can scale strongly is ONE ensemble split over both ranks
            s1 = j['secondary']['strong_1m']
            assert s1['scaling'] == 'strong' and s1['members_total'] == 2000 and s1['members_per_gpu'] == 1000 and len(s1['per_rank']['kernel_ms']) == 2
            assert s1['parity']['within_bar'] is True and s1['value'] > 0
            c4 = j['secondary']['c4']
            assert c4['reaches'] == 256 and c4['days'] == 18262 and c4['members_per_gpu'] == 20 and c4['steps'] == 1 and c4['warmup'] == 0
            assert c4['parity']['within_bar'] is True and c4['parity']['rows'].startswith('the first 730') and c4['members_flagged'] == 0
            assert j['secondary']['c5']['parity']['within_bar'] is True
            assert j['value_weak']['members_per_gpu'] == 200


def test_default_bench_line_carries_the_secondary_legs(tmp_path):
    """The driver's `python bench.py` (N = 1, no flags): the headline C3 line plus, after the timed region, BASELINE C2, C5, C4 AT ITS
    STATED SIZE (one pass of 10 000 x 256 x 18 262) and the 1M-member strong-scaling leg, each with a sample of its benchmarked table
    against the CPU oracle, the goodness-of-fit-only pass, cpu_baseline on every usable core, value_weak (= value at N = 1).
    Run here with 1 step."""
    import json
    env = dict(os.environ)
    env.pop('WORLD_SIZE', None); env.pop('RANK', None); env.pop('LOCAL_RANK', None)
    r = subprocess.run([sys.executable, os.path.join(ROOT, 'bench.py'), '--steps', '1', '--warmup', '1'],
                       env=env, capture_output=True, text=True, timeout=1500)
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [l for l in r.stdout.splitlines() if l.startswith('{')]
    assert len(lines) == 1
    j = json.loads(lines[0])
    assert j['n_gpus'] == 1 and j['scaling'] == 'strong' and j['members_total'] == 100000 and j['rccl_ranks'] == 1
    assert j['roofline']['frac'] > 0.01 and j['cpu_baseline']['kind'] == 'port' and j['members_flagged'] == 0
    sec = j['secondary']
    assert sec['c2']['replicas_bit_identical'] is True and sec['c2']['host_rows_equal'] is True and sec['c2']['lanes_per_member'] >= 4
    assert sec['c5']['parity_grade'] is False and sec['c5']['value'] > j['value']
    assert sec['c5']['parity']['within_bar'] is True and sec['c5']['members_per_gpu'] == 125000
    c4 = sec['c4']
    assert (c4['members_per_gpu'], c4['reaches'], c4['days'], c4['steps']) == (10000, 256, 18262, 1) and c4['members_flagged'] == 0
    assert c4['parity']['within_bar'] is True and c4['roofline']['frac'] > 0 and c4['rhs_evals_per_catchment_day'] > 50
    assert c4['table'] == 'pinned host memory' and c4['streamed_chunks'] > 0
    s1 = sec['strong_1m']
    assert s1['members_total'] == 1000000 and s1['scaling'] == 'strong' and s1['parity']['within_bar'] is True and s1['members_flagged'] == 0
    assert j['cpu_baseline']['cores'] == j['cpu_baseline']['usable_cores'] >= 1
    assert j['parity']['golden']['dry_members']['worst_member_max_rel_err'] < j['parity']['golden']['dry_members']['bar']
    assert sec['c3_gof_only']['value'] > j['value'] and sec['c3_gof_only']['gof_ms'] > 0 and sec['c3_gof_only']['best_member_nse_q'] > 0.0
    assert j['value_weak']['same_run_as_value'] is True and j['value_weak']['value'] == j['value']


@pytest.mark.parametrize('lanes', [1, 13, 32])
def test_thin_waves_are_bitwise_identical(engine0, lanes):
    """opts.lanes_per_wave: a wave that carries fewer than 64 members (what a small ensemble gets automatically, so that
    its members spread over the chip's SIMDs) computes the same bits: chain kernel and time-chunk queue, ragged last group."""
    import torch
    for name, E, solver in (('tarland_2004_dynamic', 150, dict(time_chunk_days=-1)), ('tarland_2004_dynamic', 150, dict(time_chunk_days=256)),
                            ('chain4_val_2004', 70, None)):
        m = perturbed(name, E, out_mask=marshal.MASK_REACH5, solver=dict(solver or {}, lanes_per_wave=64, lanes_per_member=1))
        ref, rs, rst = run(engine0, m)
        assert rst['lanes_per_wave'] == 64
        m['opts'].lanes_per_wave = lanes
        got, gs, gst = run(engine0, m)
        assert gst['lanes_per_wave'] == lanes and gst['rhs_evals'] == rst['rhs_evals']
        assert bool(torch.equal(got, ref)) and bool(torch.equal(gs, rs))
    # auto: a single-reach ensemble that cannot fill the chip is spread; a reach network keeps full waves
    m = perturbed('tarland_2004_dynamic', 3000, out_mask=marshal.MASK_REACH5, solver=dict(lanes_per_member=1))
    assert run(engine0, m)[2]['lanes_per_wave'] == 3
    m = perturbed('chain4_val_2004', 300, out_mask=marshal.MASK_REACH5, solver=dict(lanes_per_member=1))
    assert run(engine0, m)[2]['lanes_per_wave'] == 64


def test_deferred_sync_lets_the_callers_work_run_beside_the_copy_tail(engine0):
    """Engine.run(defer_sync=True) on a non-default torch stream: the call returns once the launches are enqueued, work the
    caller enqueues on that stream is ordered behind the kernel (not behind the copies), finish() waits for the run and the
    host table.  Results equal the synchronous run's; on the default stream the mode is refused."""
    import torch
    m = perturbed('tarland_1981_2010_dynamic', 2048, out_mask=marshal.MASK_REACH5)
    ref, ref_status, ref_stats = run(engine0, m)
    want = ref.sum(dim=1)
    host = engine.pinned_empty(tuple(ref.shape))
    s = torch.cuda.Stream()
    with torch.cuda.stream(s):
        out, status, st = run(engine0, m, host_out=host, defer_sync=True)
        assert 'finish' in st and 'kernel_ms' not in st
        sums = out.sum(dim=1)                     # behind the kernel, beside the copies
        st.update(st.pop('finish')())
    s.synchronize()
    assert st['queued'] == 1 and st['rhs_evals'] == ref_stats['rhs_evals'] and st['streamed_chunks'] > 0
    assert bool(torch.equal(out, ref)) and bool(torch.equal(sums, want)) and bool(torch.equal(status, ref_status))
    assert np.array_equal(host, ref.cpu().numpy(), equal_nan=True)
    with pytest.raises(engine.EngineError, match='non-default torch stream'):
        run(engine0, m, defer_sync=True)


def test_ensemble_entry_falls_back_to_a_pageable_host_table(engine0, monkeypatch):
    """run_simply_p_ensemble(to_host=True) page-locks the whole output table; on a host that cannot lock that much
    (hipHostMalloc fails, or the table exceeds 60 % of the available memory) it must fall back to an ordinary array -- which
    simplyp_stream_out accepts -- and return the same table."""
    import simplyp_amd as sp
    args = lambda: tuple(x.copy() for x in helpers.scenario_inputs('tarland_2004_dynamic'))
    fc = np.array([280.0, 290.0, 300.0, 310.0])
    want = sp.run_simply_p_ensemble(*args(), overrides={'fc': fc})
    calls = {'n': 0}
    real = engine.pinned_empty

    def refuse_big(shape, dtype=np.float64):
        n = int(np.prod(shape))
        if n > 1000:                      # the output table; the small parameter tables still get pinned memory
            calls['n'] += 1
            raise engine.EngineError("simplyp_host_alloc(%d bytes) failed (pinned host memory)" % (n * 8))
        return real(shape, dtype)
    monkeypatch.setattr(engine, 'pinned_empty', refuse_big)
    got = sp.run_simply_p_ensemble(*args(), overrides={'fc': fc})
    assert calls['n'] >= 1
    assert isinstance(got['data'], np.ndarray) and np.array_equal(got['data'], want['data']) and np.array_equal(got['status'], want['status'])
