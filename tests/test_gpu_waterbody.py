"""simplyp_waterbody: the reference's sum_to_waterbody (model.py:851-900) as a device epilogue over an ensemble's table,
called through the C ABI, against its CPU restatement (oracle/waterbody.py: bit for bit -- same operations in the same
order, IEEE division included) and against the tables the unmodified reference returned (waterbody_golden.npz: <= 1e-6,
the solver's parity bar, since the reach series underneath come from the kernel)."""

import os

import numpy as np
import pytest

import helpers
import simplyp_amd as sp
from simplyp_amd import abi, engine, marshal
from oracle import gof as gofo
from oracle import waterbody as wbo

pytestmark = pytest.mark.gpu
NAME = 'confluence3_nc_2004'
FLUX_COLS = ['Qr', 'Msus_kg/day', 'TDP_kg/day', 'PP_kg/day']


def ensemble(E, solver=None, cols=None, seed=5):
    m = helpers.marshal_scenario(NAME, E=E, out_mask=marshal.mask_of_columns(cols or marshal.REACH5_COLUMNS), solver=solver)
    rng = np.random.default_rng(seed)
    m['member_params'][marshal.PM_NAMES.index('fc')] *= rng.uniform(0.85, 1.15, E)
    m['member_params'][marshal.PM_NAMES.index('T_g')] *= rng.uniform(0.6, 1.5, E)
    m['reach_params'][marshal.PR_NAMES.index('A_catch')] *= rng.uniform(0.8, 1.25, (3, E))
    m['f_tdp'] = rng.uniform(0.4, 0.9, E)
    return m


def oracle_wb(out, cols, pos, reaches, m, member_of_col=None):
    """oracle table for the device table `out` [ncols, D, R, E] (numpy)."""
    c = lambda name: out[cols.index(name)][:, pos, :]
    A = m['reach_params'][marshal.PR_NAMES.index('A_catch')][reaches]
    f = m['f_tdp']
    if member_of_col is not None:
        A, f = A[:, member_of_col], f[member_of_col]
    return wbo.sum_to_waterbody(c('Qr'), c('Msus_kg/day'), c('TDP_kg/day'), c('PP_kg/day'), A, f)


@pytest.mark.parametrize('E', [129, 130])                    # one / two member slots per lane
@pytest.mark.parametrize('reaches', [[0, 2], [1, 2], [0, 1, 2]])
def test_device_sum_is_bit_identical_to_the_oracle(engine0, E, reaches):
    m = ensemble(E)
    out, status, st = engine0.run(m['forcing'], m['doy'], m['member_params'], m['reach_params'], m['up_ptr'], m['up_idx'], m['opts'])
    wb, info = engine0.waterbody(out, m['opts'].out_mask, reaches, m['f_tdp'], m['reach_params'])
    want = oracle_wb(out.cpu().numpy(), marshal.REACH5_COLUMNS, reaches, reaches, m)
    assert info['columns'] == abi.WB_COLUMNS == wbo.COLUMNS
    assert np.array_equal(wb.cpu().numpy(), want, equal_nan=True)
    assert info['bytes_moved'] == E * 366 * (32 * len(reaches) + 8 * 11) and info['kernel_ms'] > 0


def test_selected_output_reaches_slot_order_and_column_subset(engine0):
    """Table holding reaches [2, 0] only (in that order), cost-ordered lane slots: the waterbody series keeps the slot
    order of the table, f_TDP and A_catch are looked up by member; a column subset is a slice of the full result."""
    m = ensemble(200, solver=dict(balance=1, out_slot_order=1), cols=FLUX_COLS)
    out, status, st = engine0.run(m['forcing'], m['doy'], m['member_params'], m['reach_params'], m['up_ptr'], m['up_idx'], m['opts'],
                                  out_reaches=[2, 0])
    mos = st['member_of_slot']
    assert st['balanced'] == 1
    wb, info = engine0.waterbody(out, m['opts'].out_mask, [0, 2], m['f_tdp'], m['reach_params'], out_reaches=[2, 0],
                                 member_of_slot=mos)
    want = oracle_wb(out.cpu().numpy(), FLUX_COLS, [1, 0], [0, 2], m, member_of_col=mos.cpu().numpy())
    assert np.array_equal(wb.cpu().numpy(), want, equal_nan=True)
    sub, info2 = engine0.waterbody(out, m['opts'].out_mask, [0, 2], m['f_tdp'], m['reach_params'], out_reaches=[2, 0],
                                   member_of_slot=mos, columns=['SRP_mgl', 'Q_cumecs', 'TP_kg/day'])
    assert info2['columns'] == ['Q_cumecs', 'TP_kg/day', 'SRP_mgl']
    import torch
    assert bool(torch.equal(sub, wb[[0, 8, 9]]))


@pytest.mark.parametrize('key', ['r13', 'r23', 'r123'])
def test_base_member_meets_the_parity_bar_against_the_reference_tables(engine0, key):
    gold = np.load(os.path.join(helpers.GOLDEN, 'waterbody_golden.npz'), allow_pickle=False)
    reaches = [i for i, f in enumerate(gold['flags/' + key]) if f == 1]
    m = helpers.marshal_scenario(NAME, E=64, out_mask=marshal.mask_of_columns(FLUX_COLS))
    out, _, _ = engine0.run(m['forcing'], m['doy'], m['member_params'], m['reach_params'], m['up_ptr'], m['up_idx'], m['opts'])
    wb, _ = engine0.waterbody(out, m['opts'].out_mask, reaches, float(gold['f_TDP']), m['reach_params'])
    got = wb[:, :, 0].cpu().numpy()
    want = gold['tight/' + key].T
    assert helpers.max_rel_err(got, want) < 1e-6


def test_argument_errors(engine0):
    m = ensemble(64)
    out, _, _ = engine0.run(m['forcing'], m['doy'], m['member_params'], m['reach_params'], m['up_ptr'], m['up_idx'], m['opts'],
                            out_reaches=[0, 1])
    with pytest.raises(engine.EngineError, match='not among'):
        engine0.waterbody(out, m['opts'].out_mask, [0, 2], 0.7, m['reach_params'], out_reaches=[0, 1])
    with pytest.raises(engine.EngineError, match='ascending'):
        engine0.waterbody(out, m['opts'].out_mask, [1, 0], 0.7, m['reach_params'], out_reaches=[0, 1])
    with pytest.raises(engine.EngineError, match='must contain'):
        engine0.waterbody(out[:4].contiguous(), marshal.mask_of_columns(['Vr', 'Qr', 'Msus_kg/day', 'TDP_kg/day']), [0, 1], 0.7,
                          m['reach_params'], out_reaches=[0, 1])


def test_goodness_of_fit_of_the_summed_series(engine0):
    """simplyp_gof_waterbody vs the numpy oracle of goodness_of_fit_stats applied to the oracle's waterbody series,
    synthetic observations: 1e-9 on every statistic."""
    E = 96
    m = ensemble(E)
    out, _, _ = engine0.run(m['forcing'], m['doy'], m['member_params'], m['reach_params'], m['up_ptr'], m['up_idx'], m['opts'])
    wb, info = engine0.waterbody(out, m['opts'].out_mask, [0, 1, 2], m['f_tdp'], m['reach_params'])
    series = wb.cpu().numpy()
    rng = np.random.default_rng(9)
    D = series.shape[1]
    names = {'Q': 'Q_cumecs', 'SS': 'SS_mgl', 'TDP': 'TDP_mgl', 'PP': 'PP_mgl', 'TP': 'TP_mgl', 'SRP': 'SRP_mgl'}
    obs = np.full((6, D), np.nan)
    for vi, v in enumerate(abi.GOF_VARS):
        days = np.arange(D) if v == 'Q' else np.sort(rng.choice(D, 40 if v != 'SRP' else 8, replace=False))
        base = series[abi.WB_COLUMNS.index(names[v]), days, 0]
        obs[vi, days] = base * rng.uniform(0.7, 1.4, len(days))
    gof, ginfo = engine0.gof_waterbody(wb, info['columns'], obs, m['f_tdp'])
    got = gof.cpu().numpy()[:, :, 0, :]
    for e in range(0, E, 7):
        for vi, v in enumerate(abi.GOF_VARS):
            want = gofo.stats_of_pair(obs[vi], series[abi.WB_COLUMNS.index(names[v]), :, e])
            assert helpers.max_rel_err(got[:, vi, e], want, floor=1e-12) < 1e-9, (e, v)
    assert np.isnan(got[1:, abi.GOF_VARS.index('SRP')]).all()                 # 8 observations: dropped (:430)


def test_run_simply_p_ensemble_returns_the_waterbody(engine0, capsys):
    met, p_struc, p_SU, p_LU, p_SC, p, dyn = helpers.scenario_inputs(NAME)
    p_struc['In_final_flux?'] = [1, 0, 1]
    E = 70
    rng = np.random.default_rng(2)
    over = dict(fc=290 * rng.uniform(0.9, 1.1, E), f_TDP=rng.uniform(0.5, 0.9, E))
    res = sp.run_simply_p_ensemble(met, p_struc, p_SU, p_LU, p_SC, p, dyn, overrides=over, waterbody=True)
    assert 'Sub-catchments flowing directly into receiving waterbody: [1 3]' in capsys.readouterr().out
    wb = res['waterbody']
    assert wb['reaches'] == [1, 3] and wb['columns'] == abi.WB_COLUMNS and wb['data'].shape == (11, 366, E)
    cols = res['columns']
    pick = lambda c: res['data'][cols.index(c)][:, [0, 2], :]
    A = np.array([[float(p_SC.loc['A_catch', r])] * E for r in (1, 3)])
    want = wbo.sum_to_waterbody(pick('Qr'), pick('Msus_kg/day'), pick('TDP_kg/day'), pick('PP_kg/day'), A, over['f_TDP'])
    assert np.array_equal(wb['data'], want, equal_nan=True)
    # one flagged reach: None and the reference's message
    met, p_struc, p_SU, p_LU, p_SC, p, dyn = helpers.scenario_inputs(NAME)
    res = sp.run_simply_p_ensemble(met, p_struc, p_SU, p_LU, p_SC, p, dyn, n_members=4, waterbody=True)
    assert res['waterbody'] is None and 'One or fewer reaches' in capsys.readouterr().out
