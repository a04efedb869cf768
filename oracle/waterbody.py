"""CPU restatement of the reference's sum_to_waterbody (Current_Release/v0-2A/simplyP/model.py:851-900) for whole
ensembles.  TEST INFRASTRUCTURE ONLY (the checker of simplyp_waterbody); never imported by the product package.
Pinned against tables the unmodified reference function returned (tests/golden/waterbody_golden.npz, written by
tests/golden/make_waterbody_golden.py) in tests/test_waterbody.py.
"""

import numpy as np

# column order of the reference's df_summed: vars_to_sum (:866), the three concentrations (:886-888), then
# derived_P_species (:842-845)
COLUMNS = ['Q_cumecs', 'Msus_kg/day', 'TDP_kg/day', 'PP_kg/day', 'SS_mgl', 'TDP_mgl', 'PP_mgl',
           'TP_mgl', 'TP_kg/day', 'SRP_mgl', 'SRP_kg/day']


def q_cumecs(qr, a_catch):
    """df_R['Q_cumecs'] = df_R['Qr']*A_catch*1000/86400 (model.py:784), in that order of operations."""
    return qr * a_catch * 1000 / 86400


def sum_to_waterbody(qr, msus, tdp, pp, a_catch, f_tdp):
    """qr, msus, tdp, pp: [D, R, E] daily series of the flagged reaches in ascending reach order (Qr in mm/d, fluxes in
    kg/day); a_catch [R, E]; f_tdp scalar or [E].  Returns [11, D, E] in COLUMNS order.

    model.py:875-881: for each variable the flagged reaches' columns are added by DataFrame.sum(axis=1), i.e. left to
    right in reach order (pandas' nansum: a NaN counts as 0)."""
    def rowsum(x):
        acc = np.zeros(x.shape[:1] + x.shape[2:])
        for r in range(x.shape[1]):
            acc = acc + np.where(np.isnan(x[:, r]), 0.0, x[:, r])
        return acc
    Q = rowsum(q_cumecs(qr, a_catch[None]))
    M, T, P = rowsum(msus), rowsum(tdp), rowsum(pp)
    with np.errstate(divide='ignore', invalid='ignore'):
        ss = (M / Q) * (1000. / 86400.)                      # :886
        tdc = (T / Q) * (1000. / 86400.)                     # :887
        ppc = (P / Q) * (1000. / 86400.)                     # :888
    f = np.asarray(f_tdp, dtype=float)
    return np.stack([Q, M, T, P, ss, tdc, ppc,
                     tdc + ppc,                              # TP_mgl      :842
                     T + P,                                  # TP_kg/day   :843
                     tdc * f,                                # SRP_mgl     :844
                     T * f])                                 # SRP_kg/day  :845
