"""Device goodness-of-fit reduction (simplyp_gof) through the C ABI against the CPU oracle
(oracle/gof.py, pinned to the reference function by tests/test_gof.py) and against the reference's own tables.

Tolerances: the device accumulates shifted one-pass sums in day-list order, the oracle uses numpy's two-pass formulas:
1e-9 relative on every statistic (NSE-type ratios lose a few digits to the subtraction 1 - a/b: compared as
|got-ref| <= 1e-9 * max(1, |ref|)).  Device statistics of a *simulated* member vs the reference's table of its own
(tight) run: 2e-5 (the daily series agree to 1e-6; the statistics are ratios of sums of their differences)."""

import numpy as np
import pandas as pd
import pytest

import helpers
import simplyp_amd as sp
from simplyp_amd import abi, engine, marshal, visualise_results as vr
from oracle import gof as ogof

pytestmark = pytest.mark.gpu

FLUX = ['Qr', 'Msus_kg/day', 'TDP_kg/day', 'PP_kg/day']


def close(got, ref, tol):
    got, ref = np.asarray(got, float), np.asarray(ref, float)
    both_nan = np.isnan(got) & np.isnan(ref)
    same_inf = np.isinf(ref) & (got == ref)
    ok = both_nan | same_inf | (np.abs(got - ref) <= tol * np.maximum(1.0, np.abs(ref)))
    return bool(ok.all()), float(np.nanmax(np.where(both_nan | same_inf, 0.0, np.abs(got - ref) / np.maximum(1.0, np.abs(ref)))))


def perturbed_run(engine0, name, E, mask, seed=3, solver=None, out_reaches=None):
    m = helpers.marshal_scenario(name, E=E, out_mask=mask, solver=solver)
    rng = np.random.default_rng(seed)
    for pname, lo, hi in (('a_Q', 0.6, 1.6), ('T_g', 0.7, 1.4), ('E_M', 0.5, 2.0), ('fc', 0.85, 1.15)):
        m['member_params'][marshal.PM_NAMES.index(pname)] *= rng.uniform(lo, hi, E)
    out, status, stats = engine0.run(m['forcing'], m['doy'], m['member_params'], m['reach_params'], m['up_ptr'], m['up_idx'],
                                     m['opts'], out_reaches=out_reaches)
    return m, out, status, stats


@pytest.mark.parametrize('E', [1, 130])
def test_device_reduction_matches_oracle(engine0, E):
    """Ragged ensemble, full 25-column table (the flux columns sit at slots 5, 7, 9, 11), Tarland observations."""
    name = 'tarland_1981_2010_dynamic'
    m, out, status, _ = perturbed_run(engine0, name, E, marshal.MASK_ALL, solver=dict(rtol=1e-6, atol=1e-8))
    info = helpers.meta()[name]['inputs']
    obs = vr.observation_array(helpers.observations(info['p_SU']['st_dt'], info['p_SU']['end_dt']), [1], m['met'].index)
    f_tdp = np.linspace(0.5, 0.9, E)
    gof, ginfo = engine0.gof(out, marshal.MASK_ALL, obs, f_tdp, m['reach_params'])
    assert gof.shape == (len(abi.GOF_STATS), 6, 1, E)
    assert ginfo['n_q_days'] == int((~np.isnan(obs[0, 0])).sum()) and ginfo['n_chem_days'] == int((~np.isnan(obs[0, 1:])).any(axis=0).sum())
    assert ginfo['bytes_read'] == (8 * ginfo['n_q_days'] + 32 * ginfo['n_chem_days']) * E
    o = out.cpu().numpy()
    out4 = np.stack([o[marshal.OUT_COLUMNS.index(c), :, 0, :] for c in FLUX])
    ref = ogof.ensemble_stats(out4, m['reach_params'][marshal.PR_NAMES.index('A_catch'), 0], f_tdp, obs[0])
    ok, worst = close(gof.cpu().numpy()[:, :, 0, :], ref, 1e-9)
    assert ok, worst


def test_device_statistics_reproduce_the_reference_table(engine0):
    """Base member, default (parity-grade) solver: the device statistics equal the table the reference function made
    from the reference's own tight run (gof_golden.json), to 2e-5."""
    name = 'tarland_1981_2010_dynamic'
    m = helpers.marshal_scenario(name, E=2, out_mask=marshal.MASK_REACH5)
    out, status, _ = engine0.run(m['forcing'], m['doy'], m['member_params'], m['reach_params'], m['up_ptr'], m['up_idx'], m['opts'])
    info = helpers.meta()[name]['inputs']
    obs = vr.observation_array(helpers.observations(info['p_SU']['st_dt'], info['p_SU']['end_dt']), [1], m['met'].index)
    gof, _ = engine0.gof(out, marshal.MASK_REACH5, obs, info['p']['f_TDP'], m['reach_params'])
    g = gof.cpu().numpy()[:, :, 0, 0]
    t = helpers.gof_golden()[name + '/tight/base']
    gold = pd.DataFrame(t['values'], index=t['index'], columns=t['columns'])
    for vi, v in enumerate(abi.GOF_VARS):
        want = gold.loc[v, ['N obs', 'NSE', 'log NSE', 'r2', 'Bias (%)', 'nRMSD (%)']].values.astype(float)
        ok, worst = close(g[:6, vi], want, 2e-5)
        assert ok, (v, worst, g[:6, vi], want)
    assert np.array_equal(gof.cpu().numpy()[..., 0], gof.cpu().numpy()[..., 1])


def test_slot_order_table_and_selected_reaches(engine0):
    """Reach network, outputs of two selected reaches in slot order (cost-sorted members): statistics come back in
    member order and equal the member-order run's; observations of reach 3 are those of Tarland shifted by a week."""
    import torch
    name = 'confluence3_nc_2004'
    E = 200
    mask = marshal.mask_of_columns(FLUX)
    m, out, status, st0 = perturbed_run(engine0, name, E, mask, solver=dict(balance=0), out_reaches=[2, 0])
    base = helpers.observations('2004-01-01', '2004-12-31')[1]
    shifted = base.copy(); shifted.index = shifted.index + pd.Timedelta(days=7)
    obs = vr.observation_array({3: base, 1: shifted[shifted.index <= '2004-12-31']}, [3, 1], m['met'].index)
    f = 0.7
    ref, _ = engine0.gof(out, mask, obs, f, m['reach_params'], out_reaches=[2, 0])
    m['opts'].balance, m['opts'].balance_pilot_days, m['opts'].out_slot_order = 1, 60, 1
    out2, _, st = engine0.run(m['forcing'], m['doy'], m['member_params'], m['reach_params'], m['up_ptr'], m['up_idx'],
                              m['opts'], out_reaches=[2, 0])
    assert st['balanced'] == 1
    got, _ = engine0.gof(out2, mask, obs, f, m['reach_params'], out_reaches=[2, 0], member_of_slot=st['member_of_slot'])
    assert bool(torch.equal(got.nan_to_num(nan=-7.0), ref.nan_to_num(nan=-7.0)))
    o = out.cpu().numpy()
    for slot, reach in enumerate([2, 0]):
        want = ogof.ensemble_stats(o[:, :, slot, :], m['reach_params'][marshal.PR_NAMES.index('A_catch'), reach], np.full(E, f), obs[slot])
        ok, worst = close(ref.cpu().numpy()[:, :, slot, :], want, 1e-9)
        assert ok, (reach, worst)


def test_dropped_variables_and_poisoned_members(engine0):
    name = 'tarland_2004_dynamic'
    E = 70
    m = helpers.marshal_scenario(name, E=E, out_mask=marshal.MASK_REACH5)
    m['member_params'][marshal.PM_NAMES.index('T_g'), 9] = np.nan              # member 9 goes non-finite
    out, status, _ = engine0.run(m['forcing'], m['doy'], m['member_params'], m['reach_params'], m['up_ptr'], m['up_idx'], m['opts'])
    assert int(status[9]) & abi.STATUS_NONFINITE
    obs = vr.observation_array(helpers.observations('2004-01-01', '2004-12-31'), [1], m['met'].index)
    tp = abi.GOF_VARS.index('TP')
    obs[0, tp, np.flatnonzero(~np.isnan(obs[0, tp]))[10:]] = np.nan              # exactly 10 observations: dropped (> 10 needed)
    obs[0, abi.GOF_VARS.index('SRP')] = np.nan                                   # none at all
    gof, info = engine0.gof(out, marshal.MASK_REACH5, obs, 0.7, m['reach_params'])
    g = gof.cpu().numpy()[:, :, 0, :]
    assert (g[0, tp] == 10).all() and np.isnan(g[1:, tp]).all()
    assert (g[0, 5] == 0).all() and np.isnan(g[1:, 5]).all()
    assert np.isnan(g[1:6, :4, 9]).all() and np.isfinite(g[1:6, :4, 0]).all()   # every pair of member 9 is dropped (NaN sim)
    o = out.cpu().numpy()
    out4 = np.stack([o[marshal.columns_of_mask(marshal.MASK_REACH5).index(c), :, 0, :] for c in FLUX])
    ref = ogof.ensemble_stats(out4, m['reach_params'][0, 0], np.full(E, 0.7), obs[0])
    ok, worst = close(g, ref, 1e-9)
    assert ok, worst


def test_argument_errors(engine0):
    m = helpers.marshal_scenario('tarland_2004_static', E=2, out_mask=marshal.mask_of_columns(['Qr', 'Vr']))
    out, _, _ = engine0.run(m['forcing'], m['doy'], m['member_params'], m['reach_params'], m['up_ptr'], m['up_idx'], m['opts'])
    obs = np.full((1, 6, out.shape[1]), np.nan)
    with pytest.raises(engine.EngineError, match='out_mask must contain'):
        engine0.gof(out, marshal.mask_of_columns(['Qr', 'Vr']), obs, 0.7, m['reach_params'])
    with pytest.raises(ValueError, match='obs must have shape'):
        engine0.gof(out, marshal.mask_of_columns(['Qr', 'Vr']), obs[:, :, :5], 0.7, m['reach_params'])


def test_run_simply_p_ensemble_with_observations(engine0):
    """End to end: ensemble call with obs_dict vs the host mirror of goodness_of_fit_stats on the tables of the
    single-member drop-in run (same engine, same solver): the statistics agree to 1e-9."""
    name = 'tarland_2004_dynamic'
    met, p_struc, p_SU, p_LU, p_SC, p, dyn = helpers.scenario_inputs(name)
    obs_dict = helpers.observations(p_SU['st_dt'], p_SU['end_dt'])
    E = 4
    res = sp.run_simply_p_ensemble(met.copy(), p_struc.copy(), p_SU.copy(), p_LU.copy(), p_SC.copy(), p.copy(), dyn.copy(),
                                   overrides={'fc': np.array([290.0, 250.0, 300.0, 290.0]), 'f_TDP': np.array([0.7, 0.7, 0.7, 0.5])},
                                   obs_dict=obs_dict, keep_daily=False)
    assert res['data'] is None and res['gof']['data'].shape == (8, 6, 1, E)
    assert res['gof']['stats'] == abi.GOF_STATS and res['gof']['variables'] == abi.GOF_VARS
    p_SU2 = p_SU.copy(); p_SU2['save_stats_csv'] = 'n'
    _, df_R, _, _ = sp.run_simply_p(met.copy(), p_struc.copy(), p_SU2, p_LU.copy(), p_SC.copy(), p.copy(), dyn.copy())
    tab = sp.goodness_of_fit_stats(p_SU2, df_R, obs_dict)
    g = res['gof']['data'][:, :, 0, :]
    for v in tab.index:
        want = tab.loc[v, ['N obs', 'NSE', 'log NSE', 'r2', 'Bias (%)', 'nRMSD (%)']].values.astype(float)
        ok, worst = close(g[:6, abi.GOF_VARS.index(v), 0], want, 1e-9)
        assert ok, (v, worst)
    # member 3 differs from member 0 only in f_TDP: only SRP changes
    srp = abi.GOF_VARS.index('SRP')
    assert np.array_equal(np.delete(g[:, :, 3], srp, axis=1), np.delete(g[:, :, 0], srp, axis=1))
    assert not np.allclose(g[1:, srp, 3], g[1:, srp, 0])
    assert not np.allclose(g[1, 0, 1], g[1, 0, 0])            # another fc: another NSE of Q


def test_spearman_matches_the_oracle_and_the_reference_table(engine0):
    """simplyp_gof_spearman (ranks counted on the device) against scipy's rankdata-based restatement of
    DataFrame.corr(method='spearman') (oracle/gof.py::spearman_of_pair) for a ragged perturbed ensemble: 1e-9; the base
    member against the 'Spearmans r' column of the table the reference function made from its own tight run: 2e-5
    (a rank statistic moves only when two simulated values change order)."""
    name = 'tarland_1981_2010_dynamic'
    E = 70
    m, out, status, _ = perturbed_run(engine0, name, E, marshal.MASK_REACH5, solver=dict(rtol=1e-6, atol=1e-8))
    info = helpers.meta()[name]['inputs']
    obs = vr.observation_array(helpers.observations(info['p_SU']['st_dt'], info['p_SU']['end_dt']), [1], m['met'].index)
    f_tdp = np.linspace(0.5, 0.9, E)
    gof, ginfo = engine0.gof(out, marshal.MASK_REACH5, obs, f_tdp, m['reach_params'], spearman=True)
    rho = ginfo['spearman'].cpu().numpy()
    assert rho.shape == (6, 1, E) and ginfo['spearman_ms'] > 0
    o = out.cpu().numpy()
    cols = marshal.REACH5_COLUMNS
    A = m['reach_params'][marshal.PR_NAMES.index('A_catch'), 0]
    for e in (0, 1, 33, 69):
        sim = ogof.simulated_series(*[o[cols.index(c), :, 0, e] for c in FLUX], A[e], f_tdp[e])
        for vi, v in enumerate(abi.GOF_VARS):
            want = ogof.spearman_of_pair(obs[0, vi], sim[v])
            assert abs(rho[vi, 0, e] - want) < 1e-9, (e, v, rho[vi, 0, e], want)
    # base member, parity-grade solver, against the reference's own table
    b = helpers.marshal_scenario(name, E=2, out_mask=marshal.MASK_REACH5)
    bo, _, _ = engine0.run(b['forcing'], b['doy'], b['member_params'], b['reach_params'], b['up_ptr'], b['up_idx'], b['opts'])
    _, binfo = engine0.gof(bo, marshal.MASK_REACH5, obs, info['p']['f_TDP'], b['reach_params'], spearman=True)
    t = helpers.gof_golden()[name + '/tight/base']
    gold = pd.DataFrame(t['values'], index=t['index'], columns=t['columns'])
    got = binfo['spearman'].cpu().numpy()[:, 0, 0]
    for vi, v in enumerate(abi.GOF_VARS):
        assert abs(got[vi] - float(gold.loc[v, 'Spearmans r'])) < 2e-5, (v, got[vi], gold.loc[v, 'Spearmans r'])


def test_spearman_ties_dropped_variables_slot_order(engine0):
    """Ties in the simulated series (replicated days) and in the observations get average ranks; a variable with <= 10
    observations stays NaN; a slot-ordered table comes back in member order."""
    import torch
    name = 'tarland_2004_dynamic'
    E = 130
    m, out, status, _ = perturbed_run(engine0, name, E, marshal.mask_of_columns(FLUX), solver=dict(balance=1, out_slot_order=1))
    o = out.clone()
    o[:, 40:60] = o[:, 39:40]                     # 21 equal days in every simulated series
    rng = np.random.default_rng(4)
    D = o.shape[1]
    obs = np.full((1, 6, D), np.nan)
    obs[0, 0, 10:200] = np.round(rng.uniform(0.2, 3.0, 190), 1)      # many tied observations
    obs[0, 2, ::9] = rng.uniform(0.01, 0.1, len(range(0, D, 9)))
    obs[0, 5, :8] = 0.02                                             # 8 observations: dropped
    st = engine0.run(m['forcing'], m['doy'], m['member_params'], m['reach_params'], m['up_ptr'], m['up_idx'], m['opts'])[2]
    mos = st['member_of_slot']
    _, ginfo = engine0.gof(o, marshal.mask_of_columns(FLUX), obs, 0.7, m['reach_params'], member_of_slot=mos, spearman=True)
    rho = ginfo['spearman'].cpu().numpy()[:, 0, :]
    assert np.isnan(rho[[1, 3, 4, 5]]).all()
    on = o.cpu().numpy()
    slot_of = np.empty(E, dtype=np.int64); slot_of[mos.cpu().numpy()] = np.arange(E)
    A = m['reach_params'][marshal.PR_NAMES.index('A_catch'), 0]
    for e in (0, 7, 129):
        sim = ogof.simulated_series(*[on[i, :, 0, slot_of[e]] for i in range(4)], A[e], 0.7)
        for vi in (0, 2):
            want = ogof.spearman_of_pair(obs[0, vi], sim[abi.GOF_VARS[vi]])
            assert abs(rho[vi, e] - want) < 1e-9, (e, vi, rho[vi, e], want)
