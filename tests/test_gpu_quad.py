"""One member over four lanes (opts.lanes_per_member = 4, `ck_day_quad` in simplyp_kernels.hip.h).

The quad kernel spreads a member's Cash-Karp attempt over a DPP quad -- soil box A, soil box S, groundwater, reach -- with
per-lane coefficient tables that make every value come out of the same IEEE operations in the same order as in the one-lane
kernel.  So the test is exact equality: whole output tables, status words, right-hand-side counts per member, against the
one-lane kernel (which the other GPU tests pin to the oracle and to the reference), whatever the kernel (chain / task
queue), wave shape, network, snow prologue or member order.
"""

import numpy as np
import pytest

import helpers
from simplyp_amd import abi, engine, marshal, synthetic

pytestmark = pytest.mark.gpu


def run(eng, m, **kw):
    return eng.run(m['forcing'], m['doy'], m['member_params'], m['reach_params'], m['up_ptr'], m['up_idx'], m['opts'], **kw)


def perturbed(name, E, seed=5, **kw):
    m = helpers.marshal_scenario(name, E=E, **kw)
    rng = np.random.default_rng(seed)
    for par, lo, hi in (('fc', 0.8, 1.2), ('T_g', 0.6, 1.5), ('a_Q', 0.6, 1.6), ('b_Q', 0.8, 1.15), ('k_M', 0.8, 1.2),
                        ('T_s_A', 0.5, 2.0), ('T_s_S', 0.5, 2.0), ('beta', 0.8, 1.2), ('Qg_min', 0.0, 1.5)):
        m['member_params'][marshal.PM_NAMES.index(par)] *= rng.uniform(lo, hi, E)
    return m


def both(eng, m, **kw):
    """The same run with one lane and with four lanes per member."""
    import torch
    rhs1 = torch.zeros(m['member_params'].shape[1], dtype=torch.int32, device='cuda')
    rhs4 = torch.zeros_like(rhs1)
    m['opts'].lanes_per_member = 1
    ref, rs, rst = run(eng, m, member_rhs=rhs1, **kw)
    m['opts'].lanes_per_member = 4
    got, gs, gst = run(eng, m, member_rhs=rhs4, **kw)
    assert rst['lanes_per_member'] == 1 and gst['lanes_per_member'] == 4
    return (ref, rs, rst, rhs1), (got, gs, gst, rhs4)


CASES = [
    ('tarland_2004_static', 1, None),                                   # one member, one quad
    ('tarland_2004_static', 37, dict(time_chunk_days=-1)),              # chain kernel, ragged wave
    ('tarland_2004_dynamic', 150, dict(time_chunk_days=-1)),            # dynamic EPC0 + erodibility, 10 waves of 16 quads
    ('tarland_2004_dynamic', 150, dict(time_chunk_days=256)),           # time-chunk task queue: state handed over through HBM
    ('tarland_2004_dynamic', 150, dict(time_chunk_days=64, lanes_per_wave=5)),   # thin waves: 5 quads per wave
    ('tarland_2004_dynamic', 99, dict(time_chunk_days=-1, lanes_per_wave=1)),    # one quad per wave
    ('tarland_1981_2010_dynamic', 40, None),                            # 30 years
    ('confluence3_nc_2004', 70, None),                                  # reach network with newly-converted land, pipelined queue
    ('confluence3_nc_2004', 70, dict(time_chunk_days=-1)),              # same through the chain kernel (three launches)
    ('chain4_val_2004', 45, dict(balance=1)),                           # validation mode, Qg_min = 0 (plain step gate), cost-ordered slots
    ('tarland_2004_dynamic', 64, dict(rtol=1e-5, atol=1e-9)),           # loose tolerance: many rejected steps
    ('tarland_2004_dynamic', 64, dict(rtol=1e-11, atol=1e-14)),         # tight
]


@pytest.mark.parametrize('name,E,solver', CASES)
def test_four_lanes_per_member_are_bitwise_identical(engine0, name, E, solver):
    import torch
    m = perturbed(name, E, solver=solver)                               # all 25 columns
    (ref, rs, rst, rhs1), (got, gs, gst, rhs4) = both(engine0, m)
    assert int(rs.max()) == 0
    assert bool(torch.equal(got, ref)), 'max abs diff %g' % float((got - ref).abs().max())
    assert bool(torch.equal(gs, rs)) and bool(torch.equal(rhs4, rhs1))
    assert (gst['rhs_evals'], gst['steps'], gst['rejected']) == (rst['rhs_evals'], rst['steps'], rst['rejected'])
    assert gst['queued'] == rst['queued'] and gst['lanes_per_wave'] <= 16


def test_quad_with_snow_prologue_and_forcing_sets(engine0):
    import torch
    m = helpers.marshal_scenario('tarland_2004_dynamic', E=50, snow=True, out_mask=marshal.MASK_REACH5)
    rng = np.random.default_rng(2)
    m['member_params'][marshal.PM_NAMES.index('f_DDSM')] = rng.uniform(0.5, 6.0, 50)
    m['member_params'][marshal.PM_NAMES.index('D_snow_0')] = rng.uniform(0.0, 40.0, 50)
    for chunk in (-1, 128):
        m['opts'].time_chunk_days = chunk
        (ref, rs, rst, rhs1), (got, gs, gst, rhs4) = both(engine0, m)
        assert bool(torch.equal(got, ref)) and bool(torch.equal(gs, rs)) and bool(torch.equal(rhs4, rhs1))
    # per-member forcing sets (no LDS staging)
    m = perturbed('tarland_2004_dynamic', 9, out_mask=marshal.MASK_REACH5)
    f2 = np.concatenate([m['forcing'], m['forcing'] * 1.1], axis=0)
    m['forcing'] = np.ascontiguousarray(f2)
    fom = np.array([0, 1, 1, 0, 1, 0, 0, 1, 1], dtype=np.int32)
    (ref, rs, _, _), (got, gs, _, _) = both(engine0, m, forcing_of_member=fom)
    assert bool(torch.equal(got, ref)) and not bool(torch.equal(got[..., 0], got[..., 1]))


def test_quad_status_words(engine0):
    """A member that is non-finite from the start is flagged the same way (its numbers are garbage either way: the quad
    kernel promises equality for members whose status is 0); the step cap raises the same flag and gives the same numbers."""
    m = helpers.marshal_scenario('tarland_2004_static', E=6)
    m['member_params'][marshal.PM_NAMES.index('T_s_A'), 2] = np.nan
    (ref, rs, _, _), (got, gs, _, _) = both(engine0, m)
    rs, gs = rs.cpu().numpy(), gs.cpu().numpy()
    assert np.array_equal(rs, gs) and rs[2] & abi.STATUS_NONFINITE and rs[[0, 1, 3, 4, 5]].max() == 0
    keep = [0, 1, 3, 4, 5]
    assert np.array_equal(got.cpu().numpy()[..., keep], ref.cpu().numpy()[..., keep])
    m = perturbed('tarland_2004_static', 8, solver=dict(max_steps=12))
    (ref, rs, _, _), (got, gs, _, _) = both(engine0, m)
    assert (rs.cpu().numpy() & abi.STATUS_STEPCAP).all() and np.array_equal(rs.cpu().numpy(), gs.cpu().numpy())
    ok = (rs.cpu().numpy() & abi.STATUS_NONFINITE) == 0
    assert np.array_equal(got.cpu().numpy()[..., ok], ref.cpu().numpy()[..., ok], equal_nan=True)


def test_quad_is_chosen_for_small_ensembles_only(engine0):
    n_simd = 1024
    m = perturbed('tarland_2004_static', 300, out_mask=marshal.MASK_REACH5)
    st = run(engine0, m)[2]
    assert st['lanes_per_member'] == 4 and st['lanes_per_wave'] == 1          # 300 quads on 300 waves
    m = perturbed('tarland_2004_static', 16 * n_simd, out_mask=marshal.MASK_REACH5)
    st = run(engine0, m)[2]
    assert st['lanes_per_member'] == 4 and st['lanes_per_wave'] == 16         # the largest ensemble whose quads all find a resident wave
    m = perturbed('tarland_2004_static', 28 * n_simd, out_mask=marshal.MASK_REACH5)
    st = run(engine0, m)[2]
    assert st['lanes_per_member'] == 4 and st['queued'] == 1                  # up to 1.75 rounds of quads, through the task queue
    m = perturbed('tarland_2004_static', 28 * n_simd + 1, out_mask=marshal.MASK_REACH5)
    assert run(engine0, m)[2]['lanes_per_member'] == 1
    m = perturbed('chain4_val_2004', 40, out_mask=marshal.MASK_REACH5)
    assert run(engine0, m)[2]['lanes_per_member'] == 4                        # reach networks: groups x reaches <= SIMDs
    m = perturbed('chain4_val_2004', 4097, out_mask=marshal.MASK_REACH5)
    assert run(engine0, m)[2]['lanes_per_member'] == 1                        # 257 groups x 4 reaches > 1024
    m = perturbed('tarland_2004_static', 40, out_mask=marshal.MASK_REACH5, solver=dict(integrator='cashkarp'))
    assert run(engine0, m)[2]['lanes_per_member'] == 1                        # scheme 2 only
    m['opts'].lanes_per_member = 4
    with pytest.raises(engine.EngineError, match='integrator 2'):
        run(engine0, m)
    m = perturbed('tarland_2004_static', 40, out_mask=marshal.MASK_REACH5, solver=dict(lanes_per_member=3))
    with pytest.raises(engine.EngineError, match='lanes_per_member'):
        run(engine0, m)
    # the second pair (opts.stiff_pair): integrator 2 only when asked for explicitly; auto leaves the other schemes alone
    m = perturbed('chain4_val_2004', 40, out_mask=marshal.MASK_REACH5, solver=dict(integrator='cashkarp', stiff_pair=1))
    with pytest.raises(engine.EngineError, match='stiff_pair'):
        run(engine0, m)
    m['opts'].stiff_pair = 0
    assert run(engine0, m)[2]['stiff_pair'] == 0
    m = perturbed('chain4_val_2004', 40, out_mask=marshal.MASK_REACH5)
    assert run(engine0, m)[2]['stiff_pair'] == 1                              # a network under the default solver: auto = on
    m = perturbed('tarland_2004_static', 40, out_mask=marshal.MASK_REACH5)
    assert run(engine0, m)[2]['stiff_pair'] == 0                              # a single reach: auto = off


def test_quad_on_the_monte_carlo_distribution_against_the_oracle(engine0, oracle_lib):
    """Members of the bench's C3 distribution through the quad kernel at the bench's solver settings against the CPU oracle
    (10 x rtol, the bar of the timed-run sample in bench.py), and bit for bit against the one-lane kernel."""
    import torch
    pr = synthetic.c3_problem(96, solver=dict(out_slot_order=0))
    m = dict(forcing=pr['forcing'], doy=pr['doy'], member_params=pr['member_params'], reach_params=pr['reach_params'],
             up_ptr=pr['up_ptr'], up_idx=pr['up_idx'], opts=pr['opts'])
    (ref, rs, rst, _), (got, gs, gst, _) = both(engine0, m)
    assert bool(torch.equal(got, ref)) and int(gs.max()) == 0
    want, ws, _ = oracle_lib.run(m['forcing'], m['doy'], m['member_params'][:, :8].copy(), m['reach_params'][:, :, :8].copy(),
                                 m['up_ptr'], m['up_idx'], m['opts'])
    assert helpers.max_rel_err(got.cpu().numpy()[..., :8], want, floor=1e-12) < 10 * m['opts'].rtol


def test_quad_with_reduced_rows_slot_order_and_selected_reaches(engine0):
    """Time-reduced rows (running sums updated by the quad's writer lane only), slot-ordered columns after cost ordering, a
    subset of output reaches: the same bits as with one lane per member."""
    import torch
    m = perturbed('confluence3_nc_2004', 90, out_mask=marshal.MASK_REACH5, solver=dict(balance=1, balance_pilot_days=40, out_slot_order=1))
    D = m['forcing'].shape[2]
    pod = (np.arange(D) // 31).astype(np.int32)
    m['opts'].n_periods = int(pod.max()) + 1
    res = {}
    for team in (1, 4):
        m['opts'].lanes_per_member = team
        out, status, st = run(engine0, m, period_of_day=pod, out_reaches=[2, 0])
        assert st['lanes_per_member'] == team and st['balanced'] == 1 and int(status.max()) == 0
        res[team] = (out, st['member_of_slot'])
    assert res[4][0].shape == (5, int(pod.max()) + 1, 2, 90)
    assert bool(torch.equal(res[4][1], res[1][1])) and bool(torch.equal(res[4][0], res[1][0]))
