"""Goodness of fit on the CPU: the oracle restatement and the host mirror of the
reference's goodness_of_fit_stats against tables produced by the unmodified reference function
(tests/golden/gof_golden.json, made by tests/golden/make_gof_golden.py), and the host helpers around the device
reduction."""

import numpy as np
import pandas as pd
import pytest

import helpers
import simplyp_amd as sp
from simplyp_amd import abi, visualise_results as vr
from oracle import gof as ogof

GOLD = helpers.gof_golden()


def case_inputs(key):
    name, label, case = key.split('/')
    info = helpers.meta()[name]['inputs']['p_SU']
    obs = helpers.observations(info['st_dt'], info['end_dt'])
    sim = helpers.golden_tables(name, label)['R'][1].copy()
    f = helpers.gof_case_factors(sim.index)[case]
    sim['Q_cumecs'] = sim['Q_cumecs'] * f['Q']
    for c in ('SS_mgl', 'PP_mgl', 'TP_mgl', 'TDP_mgl', 'SRP_mgl'):
        sim[c] = sim[c] * f['C']
    return sim, obs


@pytest.mark.parametrize('key', sorted(GOLD))
def test_oracle_matches_reference_table(key):
    """numpy restatement vs the reference function's table: 1e-10 relative on every statistic incl. Spearman."""
    sim, obs = case_inputs(key)
    tab = ogof.table({v: sim[c].values for v, c in zip(vr.GOF_VARS, vr.GOF_SIM_COLUMNS)},
                     {v: obs[1][v].reindex(sim.index).values for v in obs[1].columns})
    gold = pd.DataFrame(GOLD[key]['values'], index=GOLD[key]['index'], columns=GOLD[key]['columns'])
    assert list(tab) == list(gold.index)                        # same variables kept, same order
    for v, row in tab.items():
        np.testing.assert_allclose(row, gold.loc[v].values[:7], rtol=1e-10, err_msg=v)


@pytest.mark.parametrize('key', sorted(GOLD))
def test_host_mirror_matches_reference_table(key):
    sim, obs = case_inputs(key)
    p_SU = pd.Series(dict(run_mode='cal', save_stats_csv='n'), dtype=object)
    mine = sp.goodness_of_fit_stats(p_SU, {1: sim}, obs)
    assert list(mine.index) == GOLD[key]['index'] and list(mine.columns) == GOLD[key]['columns']
    np.testing.assert_allclose(mine.to_numpy(dtype=float), np.array(GOLD[key]['values']), rtol=1e-12)


def test_host_mirror_side_effects(tmp_path, capsys):
    sim, obs = case_inputs('tarland_2004_dynamic/tight/base')
    p_SU = pd.Series(dict(run_mode='cal', save_stats_csv='y', output_fpath=str(tmp_path)), dtype=object)
    tab = sp.goodness_of_fit_stats(p_SU, {1: sim}, obs)
    saved = pd.read_csv(tmp_path / 'GoF_stats.csv', index_col=0)                     # visualise_results.py:467-469
    np.testing.assert_allclose(saved.to_numpy(dtype=float), tab.to_numpy(dtype=float), rtol=1e-12)
    # scenario mode or no observations: message, returns None (:472-474)
    p_SU['run_mode'] = 'scenario'
    assert sp.goodness_of_fit_stats(p_SU, {1: sim}, obs) is None
    assert 'cannot calculate model performance statistics' in capsys.readouterr().out
    # fewer than 11 observations of a variable: its row is dropped (:430, :453)
    few = {1: obs[1].copy()}
    few[1].loc[few[1]['TP'].dropna().index[10:], 'TP'] = np.nan
    p_SU['run_mode'], p_SU['save_stats_csv'] = 'cal', 'n'
    assert 'TP' not in sp.goodness_of_fit_stats(p_SU, {1: sim}, few).index


def test_ensemble_oracle_equals_table_rows():
    """ensemble_stats (what the device returns) = the table's rows, member by member, and NaN rows for dropped variables."""
    name = 'tarland_2004_dynamic'
    R = helpers.golden_tables(name, 'tight')['R'][1]
    info = helpers.meta()[name]['inputs']
    obs_d = helpers.observations(info['p_SU']['st_dt'], info['p_SU']['end_dt'])
    obs = vr.observation_array(obs_d, [1], R.index)
    pp = abi.GOF_VARS.index('PP')
    obs[0, pp, np.flatnonzero(~np.isnan(obs[0, pp]))[5:]] = np.nan       # PP: 5 observations left -> dropped
    E = 3
    scale = np.array([1.0, 0.8, 1.3])
    out4 = np.stack([R[c].values[:, None] * scale for c in ('Qr', 'Msus_kg/day', 'TDP_kg/day', 'PP_kg/day')])
    A, f = np.full(E, info['p_SC']['1']['A_catch']), np.full(E, info['p']['f_TDP'])
    st = ogof.ensemble_stats(out4, A, f, obs[0])
    assert st.shape == (len(abi.GOF_STATS), 6, E) and ogof.GOF_STATS == abi.GOF_STATS and ogof.GOF_VARS == abi.GOF_VARS
    assert np.isnan(st[1:, pp]).all() and (st[0, pp] == 5).all()
    gold = pd.DataFrame(GOLD[name + '/tight/base']['values'], index=GOLD[name + '/tight/base']['index'],
                        columns=GOLD[name + '/tight/base']['columns'])
    for v in ('Q', 'SS', 'TDP', 'SRP'):                            # member 0 = the golden run itself
        got = st[:6, abi.GOF_VARS.index(v), 0]
        np.testing.assert_allclose(got, gold.loc[v, ['N obs', 'NSE', 'log NSE', 'r2', 'Bias (%)', 'nRMSD (%)']].values, rtol=1e-9)


def test_loglik_from_device_sums_is_the_reference_likelihood():
    """-n/2 ln(2 pi) - n ln m - sum ln sim - sum (obs/sim-1)^2 / (2 m^2) == sum norm(sim, m*sim).logpdf(obs)
    (Development/2016/MCMC.ipynb cell 6)."""
    from scipy.stats import norm
    rng = np.random.default_rng(5)
    sim = rng.uniform(0.5, 3.0, 400)
    obs = sim * rng.normal(1.0, 0.2, 400)
    obs[::7] = np.nan
    row = ogof.stats_of_pair(obs, sim)
    ok = ~np.isnan(obs)
    for m in (0.1, 0.25):
        ref = np.sum(norm(sim[ok], m * sim[ok]).logpdf(obs[ok]))
        assert ogof.loglik(row, m) == pytest.approx(ref, rel=1e-12)
        assert float(vr.loglik(row[:, None], m)[0]) == pytest.approx(ref, rel=1e-12)


def test_observation_array_layout():
    idx = pd.date_range('2004-01-01', periods=10)
    obs = {3: pd.DataFrame({'Q': [1.0, 2.0], 'TP': [np.nan, 0.5], 'other': [9, 9]},
                           index=pd.to_datetime(['2004-01-02', '2004-01-05']))}
    arr = vr.observation_array(obs, [1, 3], idx)
    assert arr.shape == (2, 6, 10) and np.isnan(arr[0]).all()
    assert arr[1, 0, 1] == 1.0 and arr[1, 0, 4] == 2.0 and arr[1, 4, 4] == 0.5 and np.isnan(arr[1, 4, 1])
    assert np.isnan(arr[1]).sum() == 60 - 3


def test_plotting_names_exist_and_refuse():
    for name in ('plot_snow', 'plot_terrestrial', 'plot_in_stream'):
        with pytest.raises(NotImplementedError, match='outside the scope'):
            getattr(sp, name)()


def test_waterbody_csv_is_written_by_plot_instream_summed(tmp_path, capsys):
    """visualise_results.py:382-384: the third on-disk result format comes out of the plotting function."""
    idx = pd.date_range('2004-01-01', periods=3)
    df = pd.DataFrame({'Q_cumecs': [1.0, 2.0, 3.0], 'TP_mgl': [0.1, 0.2, 0.3]}, index=idx)
    p_SU = pd.Series(dict(save_output_csvs='y', output_fpath=str(tmp_path), plot_R='n'), dtype=object)
    sp.plot_instream_summed(p_SU, df, 'paper')
    assert 'Results saved to csv' in capsys.readouterr().out
    back = pd.read_csv(tmp_path / 'Instream_results_receiving_waterbody.csv', index_col=0, parse_dates=True)
    assert list(back.columns) == ['Q_cumecs', 'TP_mgl'] and np.allclose(back.to_numpy(), df.to_numpy())
    p_SU['plot_R'] = 'y'
    with pytest.raises(NotImplementedError, match='outside the scope'):
        sp.plot_instream_summed(p_SU, df, 'paper')
    p_SU['save_output_csvs'], p_SU['plot_R'] = 'n', 'n'
    assert sp.plot_instream_summed(p_SU, df, 'paper') is None
