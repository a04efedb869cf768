"""Goodness-of-fit statistics (host mirror of the reference's ``goodness_of_fit_stats``,
``Current_Release/v0-2A/simplyP/visualise_results.py:387-474``) and the per-member device reduction that
ensembles use instead (``gof_ensemble`` -> ``simplyp_gof`` of the C ABI).

The reference's plotting functions (``plot_snow``, ``plot_terrestrial``, ``plot_in_stream``,
``plot_instream_summed``; matplotlib/seaborn figures) are outside the scope of this engine: the names exist and raise
``NotImplementedError`` so that a notebook fails at the plotting cell with a clear message, not at import
(``plot_instream_summed`` first writes the receiving-waterbody CSV, the reference function's data side effect).
"""

import os

import numpy as np
import pandas as pd

GOF_VARS = ['Q', 'SS', 'TDP', 'PP', 'TP', 'SRP']                                  # ref :400
GOF_SIM_COLUMNS = ['Q_cumecs', 'SS_mgl', 'TDP_mgl', 'PP_mgl', 'TP_mgl', 'SRP_mgl']  # ref :413-414
GOF_TABLE_COLUMNS = ['N obs', 'NSE', 'log NSE', 'Spearmans r', 'r2', 'Bias (%)', 'nRMSD (%)']   # ref :460-461
GOF_DEVICE_STATS = ['N obs', 'NSE', 'log NSE', 'r2', 'Bias (%)', 'nRMSD (%)', 'sum_log_sim', 'sum_relsq']


def goodness_of_fit_stats(p_SU, df_R_dict, obs_dict):
    """Table of goodness-of-fit statistics per reach and variable, optionally saved to ``GoF_stats.csv``.

    Same arguments, return value (DataFrame: one row per variable with more than 10 observations, columns
    ``N obs, NSE, log NSE, Spearmans r, r2, Bias (%), nRMSD (%), Reach``), CSV side effect and printed message as the
    reference function (visualise_results.py:387-474)."""
    if p_SU.run_mode != 'scenario' and len(obs_dict) > 0:
        stats_df_li = []
        for SC in df_R_dict.keys():
            if SC not in obs_dict.keys():
                continue
            sim_df = df_R_dict[SC][GOF_SIM_COLUMNS]
            obs_df = obs_dict[SC]
            rows, names = [], []
            for var, col in zip(GOF_VARS, GOF_SIM_COLUMNS):
                if var not in obs_df.columns:
                    continue
                obs = obs_df[var]
                n_obs = int(obs.notnull().sum())
                if n_obs <= 10:
                    continue
                tdf = pd.concat([obs, sim_df[col]], axis=1).dropna(how='any')
                tdf.columns = ['obs', 'sim']
                with np.errstate(divide='ignore', invalid='ignore'):
                    tldf = np.log(tdf)
                o, s = tdf['obs'], tdf['sim']
                lo, ls = tldf['obs'], tldf['sim']
                NSE = 1 - (np.sum((o - s) ** 2) / np.sum((o - np.mean(o)) ** 2))
                log_NSE = 1 - (np.sum((lo - ls) ** 2) / np.sum((lo - np.mean(lo)) ** 2))
                spearmans_r = tdf.corr(method='spearman').iloc[0, 1]
                r2 = (tdf.corr(method='pearson') ** 2).iloc[0, 1]
                pbias = 100 * np.sum(s - o) / np.sum(o)
                RMSD_norm = 100 * np.mean(np.abs(s - o)) / np.std(o.to_numpy())
                rows.append([n_obs, NSE, log_NSE, spearmans_r, r2, pbias, RMSD_norm])
                names.append(var)
            stats_df = pd.DataFrame(data=rows, columns=GOF_TABLE_COLUMNS, index=names)
            stats_df['Reach'] = SC
            stats_df_li.append(stats_df)
        stats_df_allSC = pd.concat(stats_df_li)
        if p_SU.save_stats_csv == 'y':
            stats_df_allSC.to_csv(os.path.join(p_SU.output_fpath, "GoF_stats.csv"))
        return stats_df_allSC
    print('No observations read in, therefore cannot calculate model performance statistics')


def observation_array(obs_dict, reaches, index):
    """``[n_reaches][6][D]`` float64, NaN = no observation: the layout ``simplyp_gof`` takes.  ``reaches``: sub-catchment
    ids in output order; ``index``: the run's DatetimeIndex.  Observations outside the run period are ignored (the
    reference truncates them when it reads the files, inputs.py:133-134)."""
    arr = np.full((len(reaches), len(GOF_VARS), len(index)), np.nan)
    for ri, SC in enumerate(reaches):
        if SC not in obs_dict:
            continue
        df = obs_dict[SC]
        df = df[~df.index.duplicated(keep='first')].reindex(index)
        for vi, var in enumerate(GOF_VARS):
            if var in df.columns:
                arr[ri, vi] = df[var].to_numpy(dtype=float)
    return arr


def loglik(gof_data, m, stat_names=GOF_DEVICE_STATS):
    """Gaussian log-likelihood with sigma = m * sim (the reference's calibration notebooks,
    Development/2016/MCMC.ipynb cell 6) from the device statistics ``gof_data[n_stats, ...]``; ``m`` broadcasts."""
    n = gof_data[stat_names.index('N obs')]
    return (-0.5 * n * np.log(2 * np.pi) - n * np.log(m) - gof_data[stat_names.index('sum_log_sim')]
            - gof_data[stat_names.index('sum_relsq')] / (2 * m * m))


def _no_plots(name):
    def f(*args, **kwargs):
        raise NotImplementedError("%s: plotting is outside the scope of simplyp_amd (DESIGN.md section 6); the result "
                                  "tables are the reference's, pass them to the reference's own plotting code" % name)
    f.__name__ = name
    return f


plot_snow = _no_plots('plot_snow')
plot_terrestrial = _no_plots('plot_terrestrial')
plot_in_stream = _no_plots('plot_in_stream')


def plot_instream_summed(p_SU, df_summed, fig_display_type=None):
    """The reference function draws the summed series and -- its only data side effect -- writes
    ``Instream_results_receiving_waterbody.csv`` when ``p_SU.save_output_csvs == 'y'`` (visualise_results.py:382-384; one of
    the three on-disk result formats).  This stand-in writes that file (same name, same
    ``to_csv`` call, same printed line); the figure itself is outside the scope of simplyp_amd, so asking for it
    (``p_SU.plot_R == 'y'``) raises ``NotImplementedError`` after the file is written."""
    if p_SU.save_output_csvs == 'y':
        df_summed.to_csv(os.path.join(p_SU.output_fpath, "Instream_results_receiving_waterbody.csv"))
        print('Results saved to csv')
    if p_SU.plot_R == 'y':
        _no_plots('plot_instream_summed')()
