"""CPU oracle of the goodness-of-fit reduction.  TEST INFRASTRUCTURE ONLY: imported by
tests/ and tools that check the device reduction, never by the product package.

Restates, in numpy, ``goodness_of_fit_stats`` of the reference
(Current_Release/v0-2A/simplyP/visualise_results.py:387-474): for each variable of ``stats_var_li`` (:400) with more
than 10 observations (:430), on the days where both series are non-null (:435-437):

    NSE      = 1 - sum((obs-sim)^2) / sum((obs-mean(obs))^2)                      (:441)
    log NSE  = the same on np.log of both series                                  (:442-443)
    r2       = Pearson correlation squared                                        (:446-447)
    Bias (%) = 100 * sum(sim-obs) / sum(obs)                                      (:448)
    nRMSD(%) = 100 * mean(|sim-obs|) / std(obs),  std with ddof = 0 (np.std)      (:449)
    Spearman's r (:444-445): rank correlation -- host only, not part of the device reduction.

and the simulated series themselves from the four daily reach outputs (model.py:784-793, :831-847):

    Q = Qr*A_catch*1000/86400;  SS,TDP,PP = (flux/Qr)/A_catch;  TP = TDP+PP;  SRP = f_TDP*TDP.

Pinned against tests/golden/gof_golden.json (tables produced by the unmodified reference function, see
tests/golden/make_gof_golden.py).  Two extra per-member sums that the reference's MCMC likelihood needs
(Development/2016/MCMC.ipynb cell 6: Gaussian errors with sigma = m*sim) ride along:
``sum_log_sim = sum(ln sim)`` and ``sum_relsq = sum((obs/sim - 1)^2)`` so that
``loglik(m) = -n/2 ln(2 pi) - n ln(m) - sum_log_sim - sum_relsq/(2 m^2)``.
"""

import numpy as np

GOF_VARS = ['Q', 'SS', 'TDP', 'PP', 'TP', 'SRP']                    # visualise_results.py:400
GOF_STATS = ['N obs', 'NSE', 'log NSE', 'r2', 'Bias (%)', 'nRMSD (%)', 'sum_log_sim', 'sum_relsq']
MIN_OBS = 10                                                        # "> 10 observations", :430


def simulated_series(Qr, Msus_f, TDP_f, PP_f, A_catch, f_TDP):
    """The six simulated series (model.py:784-793, :831-847) from the four daily reach outputs; any array shape."""
    Q = Qr * A_catch * 1000 / 86400
    SS = (Msus_f / Qr) / A_catch
    TDP = (TDP_f / Qr) / A_catch
    PP = (PP_f / Qr) / A_catch
    return dict(Q=Q, SS=SS, TDP=TDP, PP=PP, TP=TDP + PP, SRP=TDP * f_TDP)


def stats_of_pair(obs, sim):
    """One row of the reference's table (without Spearman) + the two likelihood sums, or NaNs when the variable is
    dropped (n_obs <= 10).  obs, sim: 1-D arrays over days, NaN = missing."""
    obs = np.asarray(obs, dtype=float); sim = np.asarray(sim, dtype=float)
    n_obs = int(np.sum(~np.isnan(obs)))                                             # :428
    row = np.full(len(GOF_STATS), np.nan)
    row[0] = n_obs
    if n_obs <= MIN_OBS:
        return row
    ok = ~np.isnan(obs) & ~np.isnan(sim)                                            # dropna(how='any'), :436
    o, s = obs[ok], sim[ok]
    with np.errstate(divide='ignore', invalid='ignore'):
        lo, ls = np.log(o), np.log(s)
        row[1] = 1 - np.sum((o - s) ** 2) / np.sum((o - np.mean(o)) ** 2)
        # np.sum / np.mean of a pandas Series skip NaN (:442-443): a day with log(sim < 0) = NaN drops out of the sums
        row[2] = 1 - np.nansum((lo - ls) ** 2) / np.nansum((lo - np.nanmean(lo)) ** 2)
        do, ds = o - np.mean(o), s - np.mean(s)
        row[3] = np.sum(do * ds) ** 2 / (np.sum(do * do) * np.sum(ds * ds))
        row[4] = 100 * np.sum(s - o) / np.sum(o)
        row[5] = 100 * np.mean(np.abs(s - o)) / np.std(o)
        row[6] = np.sum(ls)
        row[7] = np.sum((o / s - 1) ** 2)
    return row


def spearman_of_pair(obs, sim):
    """Spearman's r as pandas' ``corr(method='spearman')`` computes it (:444): Pearson on average ranks."""
    from scipy.stats import rankdata
    ok = ~np.isnan(obs) & ~np.isnan(sim)
    ro, rs = rankdata(obs[ok]), rankdata(sim[ok])
    return float(np.corrcoef(ro, rs)[0, 1])


def table(sim_by_var, obs_by_var):
    """The reference's table for one reach as {var: [N obs, NSE, log NSE, Spearmans r, r2, Bias, nRMSD]}; variables
    without (enough) observations are absent, like the rows the reference removes (:453-457)."""
    out = {}
    for v in GOF_VARS:
        if v not in obs_by_var:
            continue
        row = stats_of_pair(obs_by_var[v], sim_by_var[v])
        if row[0] <= MIN_OBS:
            continue
        out[v] = [row[0], row[1], row[2], spearman_of_pair(np.asarray(obs_by_var[v], float), np.asarray(sim_by_var[v], float)),
                  row[3], row[4], row[5]]
    return out


def ensemble_stats(out4, A_catch, f_TDP, obs):
    """What the device reduction returns: ``[n_stats][6][E]`` for one reach.
    out4: [4][D][E] = Qr, Msus_kg/day, TDP_kg/day, PP_kg/day; A_catch, f_TDP: [E]; obs: [6][D] with NaN = missing."""
    E = out4.shape[2]
    res = np.full((len(GOF_STATS), len(GOF_VARS), E), np.nan)
    for e in range(E):
        sim = simulated_series(out4[0, :, e], out4[1, :, e], out4[2, :, e], out4[3, :, e], A_catch[e], f_TDP[e])
        for vi, v in enumerate(GOF_VARS):
            res[:, vi, e] = stats_of_pair(obs[vi], sim[v])
    return res


def loglik(stats_row, m):
    """Gaussian log-likelihood with sigma = m*sim (MCMC.ipynb cell 6) from one variable's device statistics.
    Uses the number of paired days = N obs when no simulated value is missing."""
    n = stats_row[0]
    return -0.5 * n * np.log(2 * np.pi) - n * np.log(m) - stats_row[6] - stats_row[7] / (2 * m * m)
