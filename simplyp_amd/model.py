"""SimplyP model entry points on the MI355X engine.

Drop-in for the reference's ``Current_Release/v0-2A/simplyP/model.py``:
``run_simply_p`` keeps the signature, return types, column names, in-place
edits of ``p_LU``/``p_SC``, printed lines and error cases of model.py:193-827,
but the sub-catchment x day loop nest (model.py:365-724) runs as one batched HIP
launch sequence behind the C ABI (``engine.py``).  ``run_simply_p_ensemble`` is
the batched entry the reference does not have.

The three scalar helpers ``f_x``, ``discretized_soilP`` and ``ode_f`` are kept
as plain Python for API compatibility (``__init__.py`` of the reference exports
them); they are not used by ``run_simply_p``.
"""

import os

import numpy as np
import pandas as pd

from . import abi, marshal
from . import helper_functions as hf


# ------------------------------------------------------------------------------------------
# scalar helpers kept for API compatibility (reference model.py:23-187)

def f_x(x, threshold, reld):
    """Smooth 0..1 gate above ``threshold`` (reference model.py:23-37)."""
    d = threshold * reld
    if x < threshold:
        return 0
    if x > threshold + d:
        return 1
    s = (x - threshold) / d
    return -2 * s**3 + 3 * s**2


def discretized_soilP(P_netInput, catchment_area, SC, Kf, Msoil, EPC0, Qs, Qq, Vs, TDPs, Plab):
    """One-day closed-form update of soil-water TDP and labile soil P (reference model.py:39-56)."""
    a = P_netInput * catchment_area * 100 / 365. + Kf * Msoil * EPC0
    b = (Kf * Msoil + Qs + Qq) / Vs
    TDPs = a / b + (TDPs - a / b) * np.exp(-b)
    b0 = b * Vs
    if Vs > 0:
        sorp = Kf * Msoil * (a / b0 - EPC0 + (1 / b) * (TDPs / Vs - a / b0) * (1 - np.exp(-b)))
    else:
        sorp = 0.
    Plab = Plab + sorp
    return (TDPs, Plab)


def ode_f(y, t, ode_params):
    """Right-hand side of the 12-variable daily system, same argument tuple as the reference
    (model.py:58-187).  Python convenience only; the engine evaluates this on the GPU."""
    (P, E, mu, Qq_i, Qr_US_i, Esus_i, Msus_US_i, TDPr_US_i, PPr_US_i,
     f_A, f_Ar, f_IG, f_S, f_NC_A, f_NC_Ar, f_NC_IG, f_NC_S, NC_type,
     f_quick, alpha, beta, T_s, T_g, fc, L_reach, A_catch,
     a_Q, b_Q, E_M, k_M, conc_TDPs_A, conc_TDPs_NC, PlabA_i, PlabNC_i,
     Msoil, TDPeff, TDPg, E_PP, P_inactive, dynamic_EPC0, Qg_min) = ode_params
    VsA, VsS, Vg, Vr, Qr = y[0], y[1], y[2], y[3], y[4]
    Msus, TDPr, PPr = y[6], y[8], y[10]
    QsA = (VsA - fc) * f_x(VsA, fc, 0.01) / T_s['A']
    dVsA = P * (1 - f_quick) - alpha * E * (1 - np.exp(-mu * VsA)) - QsA
    QsS = (VsS - fc) * f_x(VsS, fc, 0.01) / T_s['S']
    dVsS = P * (1 - f_quick) - alpha * E * (1 - np.exp(-mu * VsS)) - QsS
    QsNC = QsA if NC_type == 'A' else QsS
    f_Qg = f_x(Vg / T_g, Qg_min, 0.01)
    Qg = (1 - f_Qg) * Qg_min + f_Qg * (Vg / T_g)
    Qsum = f_A * QsA + f_S * QsS
    dVg = beta * Qsum - Qg
    inflow = Qq_i + (1 - beta) * Qsum + Qg + Qr_US_i - Qr
    dQr = inflow * a_Q * (Qr**b_Q) * 86400. / ((1 - b_Q) * L_reach)
    QrkM = Qr**k_M
    MA, MS, MIG = Esus_i['A'] * QrkM, Esus_i['S'] * QrkM, Esus_i['IG'] * QrkM
    out_M, out_T, out_P = (Msus / Vr) * Qr, Qr * (TDPr / Vr), Qr * PPr / Vr
    dMsus = f_Ar * MA + f_IG * MIG + f_S * MS + Msus_US_i - out_M
    dTDPr = ((1 - beta) * (f_A * (1 - f_NC_A) * QsA * conc_TDPs_A + f_A * f_NC_A * QsNC * conc_TDPs_NC
                           + f_S * f_NC_S * QsNC * conc_TDPs_NC)
             + f_A * (1 - f_NC_A) * Qq_i * conc_TDPs_A + f_A * f_NC_A * Qq_i * conc_TDPs_NC
             + f_S * f_NC_S * Qq_i * conc_TDPs_NC
             + Qg * hf.UC_Cinv(TDPg, A_catch) + TDPeff + TDPr_US_i - out_T)
    dPPr = (E_PP * (f_Ar * (1 - f_NC_Ar) * MA * (PlabA_i + P_inactive) / Msoil
                    + f_IG * (1 - f_NC_IG) * MIG * (PlabA_i + P_inactive) / Msoil
                    + f_S * (1 - f_NC_S) * MS * P_inactive / Msoil
                    + f_Ar * f_NC_Ar * MA * (PlabNC_i + P_inactive) / Msoil
                    + f_IG * f_NC_IG * MIG * (PlabNC_i + P_inactive) / Msoil
                    + f_S * f_NC_S * MS * (PlabNC_i + P_inactive) / Msoil)
            + PPr_US_i - out_P)
    return np.array([dVsA, dVsS, dVg, inflow, dQr, Qr, dMsus, out_M, dTDPr, out_T, dPPr, out_P])


# ------------------------------------------------------------------------------------------
# post-processing (reference model.py:831-900)

def derived_P_species(df_R, f_TDP):
    """TP = TDP + PP and SRP = f_TDP * TDP, for fluxes and concentrations (reference model.py:831-847)."""
    df_R['TP_mgl'] = df_R['TDP_mgl'] + df_R['PP_mgl']
    df_R['TP_kg/day'] = df_R['TDP_kg/day'] + df_R['PP_kg/day']
    df_R['SRP_mgl'] = df_R['TDP_mgl'] * f_TDP
    df_R['SRP_kg/day'] = df_R['TDP_kg/day'] * f_TDP
    return df_R


def sum_to_waterbody(p_struc, n_SC, df_R_dict, f_TDP):
    """Sum the reaches flagged 'In_final_flux?' into one series (reference model.py:851-900)."""
    vars_to_sum = ['Q_cumecs', 'Msus_kg/day', 'TDP_kg/day', 'PP_kg/day']
    reaches_in_final_flux = p_struc['In_final_flux?'][p_struc['In_final_flux?'] == 1].index.values
    if len(reaches_in_final_flux) > n_SC:
        raise ValueError("Mismatch between the number of subcatchments in the 'Setup' parameter sheet \n(parameter 'n_SC') and in the 'Reach_structure' parameter sheet")
    print('Sub-catchments flowing directly into receiving waterbody: %s' % reaches_in_final_flux)
    if len(reaches_in_final_flux) > 1:
        index = df_R_dict[reaches_in_final_flux[0]].index
        df_summed = pd.DataFrame(
            {var: np.sum([df_R_dict[r][var].to_numpy(dtype=float) for r in reaches_in_final_flux], axis=0)
             for var in vars_to_sum}, index=index, columns=vars_to_sum)
        df_summed['SS_mgl'] = (df_summed['Msus_kg/day'] / df_summed['Q_cumecs']) * (1000. / 86400.)
        df_summed['TDP_mgl'] = (df_summed['TDP_kg/day'] / df_summed['Q_cumecs']) * (1000. / 86400.)
        df_summed['PP_mgl'] = (df_summed['PP_kg/day'] / df_summed['Q_cumecs']) * (1000. / 86400.)
        df_summed = derived_P_species(df_summed, f_TDP)
        return df_summed
    else:
        print('One or fewer reaches were selected to be included in the sum, check your reach structure parameters')
        return None


# ------------------------------------------------------------------------------------------
# the hot path

def _engine_opts(p_SU, p, dynamic_options, step_len, solver, out_mask, n_periods=0, snow=False):
    scs = marshal.sc_list(p)
    return abi.make_opts(solver,
                         dynamic_epc0=(dynamic_options['Dynamic_EPC0'] == 'y'),
                         dynamic_erod=(dynamic_options['Dynamic_erodibility'] == 'y'),
                         run_mode_cal=(p_SU.run_mode == 'cal'),
                         sc_qr0=scs.index(int(p['SC_Qr0'])),
                         out_mask=out_mask, step_len=step_len, n_periods=n_periods, snow=snow)


def _kf_last(p_SU, p_LU, p_SC, p):
    """Kf as the reference returns it: the value of the last sub-catchment (model.py:449-453, :827)."""
    SC = marshal.sc_list(p)[-1]
    if p_SU.run_mode == 'cal':
        return 10**-6 * (p_LU['A']['SoilPconc'] - p_LU['S']['SoilPconc']) / \
            hf.UC_Cinv(p_LU['A']['EPC0_init_mgl'], p_SC.loc['A_catch', SC])
    return p['Kf']



def _output_dict(stats, status, solver):
    """The 4th return value.  The reference hands back LSODA's infodict of its LAST odeint call (model.py:640, :827: last
    sub-catchment, last day) -- arrays of one element under 'nst', 'nfe', 'nje', 'hu', ..., and 'message'.  LSODA is not what
    integrates here, so the engine's own statistics are returned, and the keys a caller of the reference could have read are
    kept as aliases with the same types: 'nfe' (right-hand-side evaluations), 'nst' (accepted steps), 'nje' (Jacobian
    evaluations: none, the pair is explicit) as one-element int32 arrays -- totals over the whole run, not the last day's
    call; saturating at 2**31 - 1 (a large network exceeds int32: the exact totals are under 'rhs_evals' / 'steps') --,
    'mused' (1 = non-stiff method) and 'message' ('Integration successful.' unless the member was flagged)."""
    msg = 'Integration successful.'
    if status & abi.STATUS_NONFINITE:
        msg = 'Non-finite state met (member status %d); results from that day on are NaN.' % status
    elif status & abi.STATUS_STEPCAP:
        msg = 'Excess work done on a day (max_steps attempts; member status %d).' % status
    d = dict(stats, member_status=status, solver=dict(abi.DEFAULT_SOLVER, **(solver or {})), engine='MI355X batched engine',
             message=msg)
    i32max = np.iinfo(np.int32).max
    d['nfe'] = np.array([min(int(stats['rhs_evals']), i32max)], dtype=np.int32)
    d['nst'] = np.array([min(int(stats['steps']), i32max)], dtype=np.int32)
    d['nje'] = np.array([0], dtype=np.int32)
    d['mused'] = np.array([1], dtype=np.int32)
    return d

def run_simply_p(met_df, p_struc, p_SU, p_LU, p_SC, p, dynamic_options, step_len=1., solver=None, device=0):
    """Simple hydrology, sediment and phosphorus model (reference model.py:193-827).

    Same arguments and 4-tuple return ``(df_TC_dict, df_R_dict, Kf, output_dict)`` as the
    reference.  Extra keyword arguments: ``solver`` (dict overriding ``abi.DEFAULT_SOLVER``:
    integrator 'cashkarp'|'rk4', rtol, atol, substeps, max_steps, project_vr) and ``device``.
    ``output_dict`` holds the engine's solver statistics, with LSODA's infodict keys ``nfe``, ``nst``, ``nje``, ``mused``,
    ``message`` kept as aliases (``_output_dict``).
    """
    from . import engine

    # derived parameters, validation, in-place edits of p_LU / p_SC (model.py:311-361)
    marshal.prologue(p_SU, p_LU, p_SC, p)
    scs = marshal.sc_list(p)
    up_ptr, up_idx, up_lists = marshal.topology(p_struc, p)
    mp = marshal.member_params(p, p_LU, 1)
    rp = marshal.reach_params(p_SC, p, 1)
    forcing, doy = marshal.forcing_arrays(met_df)
    opts = _engine_opts(p_SU, p, dynamic_options, step_len, solver, marshal.MASK_ALL)

    eng = engine.get_engine(device)          # raises when the HIP library / device is missing
    out_d, status_d, stats = eng.run(forcing, doy, mp, rp, up_ptr, up_idx, opts)
    out = out_d.cpu().numpy()                # [25, D, S, 1]
    status = int(status_d.cpu().numpy()[0])
    marshal.epilogue_mutations(p_SU, p_LU, p_SC, p)

    df_TC_dict, df_R_dict = {}, {}
    for j, SC in enumerate(scs):
        print('Starting model run for sub-catchment: %s' % SC)                               # :367
        if len(up_lists[SC]) > 0:
            print('Reaches directly upstream of this reach: %s' % up_lists[SC])              # :512
        else:
            print('No reaches directly upstream of this reach')                              # :542
        A_catch = p_SC.loc['A_catch', SC]
        df_ODE = pd.DataFrame(out[:12, :, j, 0].T, columns=marshal.ODE_COLUMNS, index=met_df.index)       # :736-740
        df_nonODE = pd.DataFrame(out[12:, :, j, 0].T, columns=marshal.NONODE_COLUMNS, index=met_df.index)  # :742-746

        df_TC = pd.concat([df_ODE[['VsA', 'VsS', 'Vg']], df_nonODE], axis=1)                  # :755
        df_TC['TDPs_A_mgl'] = hf.UC_C(df_TC['conc_TDPs_A_kgmm'], A_catch)                     # :758
        df_TC['EPC0_A_mgl'] = hf.UC_C(df_TC['EPC0_A_kgmm'], A_catch)                          # :759
        df_TC['Plabile_A_mgkg'] = (10**6 * df_TC['P_labile_A_kg'] / (p['Msoil_m2'] * 10**6 * A_catch))   # :760-761
        if p_SC.loc['NC_type', SC] != 'None':                                                # :764-773
            if p_SC.loc['NC_type', SC] == 'A':
                df_TC['VsNC'] = df_TC['VsA']
                df_TC['QsNC'] = df_TC['QsA']
            else:
                df_TC['VsNC'] = df_TC['VsS']
                df_TC['QsNC'] = df_TC['QsS']
            df_TC['TDPs_NC_mgl'] = hf.UC_C(df_TC['conc_TDPs_NC_kgmm'], A_catch)
            df_TC['Plabile_NC_mgkg'] = (10**6 * df_TC['P_labile_NC_kg'] / (p['Msoil_m2'] * 10**6 * A_catch))
        if p_SU.inc_snowmelt == 'y':                                                         # :775-776
            df_TC['D_snow'] = met_df['D_snow_end']

        df_R = df_ODE.drop(['VsA', 'VsS', 'Vg'], axis=1)                                      # :779
        df_R['Q_cumecs'] = df_R['Qr'] * A_catch * 1000 / 86400                                # :784
        df_R['SS_mgl'] = hf.UC_C(df_R['Msus_kg/day'] / df_R['Qr'], A_catch)                   # :788
        df_R['TDP_mgl'] = hf.UC_C(df_R['TDP_kg/day'] / df_R['Qr'], A_catch)                   # :789
        df_R['PP_mgl'] = hf.UC_C(df_R['PP_kg/day'] / df_R['Qr'], A_catch)                     # :790
        df_R = derived_P_species(df_R, p['f_TDP'])                                            # :793

        df_TC_dict[SC] = df_TC.sort_index(axis=1)                                             # :797-800
        df_R_dict[SC] = df_R.sort_index(axis=1)
        print('Finished!\n')                                                                  # :802

    Kf = _kf_last(p_SU, p_LU, p_SC, p)
    if p_SU.run_mode == 'cal':                                                               # :809-812
        print("Running in calibration mode; the soil P sorption coefficient has been estimated as %s mm/kg\n" % Kf)
    else:
        print("Running in validation or scenario mode, so the soil P sorption coefficient has been read from the parameter file")

    if p_SU.save_output_csvs == 'y':                                                         # :815-825
        out_dir = p_SU.output_fpath.replace('\\', os.sep) if isinstance(p_SU.output_fpath, str) else p_SU.output_fpath
        for SC in df_R_dict.keys():
            df_TC_dict[SC].to_csv(os.path.join(out_dir, "Results_TC_SC%s.csv" % SC))
            df_R_toSave = df_R_dict[SC].drop(['Msus_EndOfDay', 'PPr_EndOfDay', 'Qr',
                                              'Qr_EndOfDay', 'TDPr_EndOfDay', 'Vr'], axis=1)
            df_R_toSave.to_csv(os.path.join(out_dir, "Instream_results_Reach%s.csv" % SC))
        print('Results saved to csv\n')

    output_dict = _output_dict(stats, status, solver)
    return (df_TC_dict, df_R_dict, Kf, output_dict)                                          # :827



def _host_table(shape, pin):
    """Host array that receives an ensemble's output table: page-locked (the streamed copies then run at PCIe speed beside the
    kernel) when the host can lock that much -- not more than 60 % of what it has available, the guard bench.py applies --
    else an ordinary pageable array, which ``simplyp_stream_out`` accepts too (slower copies, same table)."""
    from . import engine
    nbytes = int(np.prod(shape, dtype=np.int64)) * 8
    avail = None
    try:
        with open('/proc/meminfo') as fh:
            for line in fh:
                if line.startswith('MemAvailable:'):
                    avail = int(line.split()[1]) * 1024
    except OSError:
        pass
    if avail is None or nbytes <= 0.6 * avail:
        try:
            return pin(shape)
        except engine.EngineError:
            pass
    return np.empty(shape, dtype=np.float64)

def run_simply_p_ensemble(met_df, p_struc, p_SU, p_LU, p_SC, p, dynamic_options, overrides=None, n_members=None,
                          outputs=None, out_reaches=None, step_len=1., solver=None, device=0, to_host=True,
                          reduce=None, obs_dict=None, keep_daily=True, snow_in_kernel=None, forcing_of_member=None,
                          waterbody=None, waterbody_obs=None, spearman=False, devices=None):
    """Run an ensemble of parameter sets through the engine in one call.

    ``overrides``: dict name -> array[E] (member parameters, see ``marshal.PM_NAMES``) or
    array broadcastable to [S, E] (reach parameters, ``marshal.PR_NAMES``); parameters not
    listed take the workbook value for every member.  ``outputs``: list of reference column
    names (default: the five documented reach outputs, model.py:272-277); with the in-kernel snow module also ``'D_snow'``,
    every member's snow depth at the end of the day (what the reference returns as ``df_TC['D_snow']``, model.py:775-776).  ``out_reaches``:
    sub-catchment ids to return (default all).  ``reduce``: None for daily rows, ``'annual'`` for one row
    per calendar year holding the sum of that year's daily values (e.g. annual fluxes), or an int array
    [D] of period indices; the periods are returned under ``'periods'``.
    ``met_df`` may also be a list of met dataframes over the same dates (forcing scenarios: climate members, bias-corrected
    series ...) with ``forcing_of_member`` [E] giving each member's scenario; all sets sit in HBM once and members read theirs.
    ``snow_in_kernel``: run the degree-day snow module (reference ``snow_hydrol_inputs``, inputs.py:159-210) per member
    inside the kernel from ``met_df['Precipitation']`` / ``['T_air']`` instead of taking ``met_df['P']``; default: on when
    ``overrides`` perturbs ``f_DDSM`` or ``D_snow_0`` (with the workbook values the result is bit-identical either way).
    ``obs_dict``: observations as returned by ``read_input_data`` (dict SC -> DataFrame with columns among
    Q, SS, TDP, PP, TP, SRP): every member's goodness-of-fit table (the reference's ``goodness_of_fit_stats``,
    visualise_results.py:387-474, minus Spearman's r, plus the two sums of its Gaussian likelihood) is reduced on the
    device from the daily series and returned under ``'gof'`` = dict(stats, variables, data[n_stats, 6, n_reaches, E],
    info); needs daily rows (``reduce=None``); the four flux columns are added to ``outputs`` when missing.  ``SRP``
    uses ``p['f_TDP']`` or ``overrides['f_TDP']`` (array[E]).  ``spearman=True`` adds the table's rank statistic under
    ``['gof']['spearman']`` [6, n_reaches, E] (counted on the device, ~0.1-0.2 s for 100 000 members).  ``keep_daily=False`` drops the daily table once the
    statistics exist (``data`` is None): the 44 GB of a 100 000-member run never leave the device.

    ``waterbody``: the reference's ``sum_to_waterbody`` (model.py:851-900) for every member, on the device: ``True`` sums
    the sub-catchments flagged ``p_struc['In_final_flux?'] == 1``, a list sums the given sub-catchment ids.  Like the
    reference, fewer than two reaches give ``None`` (:872, :895).  The result comes back under ``'waterbody'`` =
    dict(columns (the reference's 11: ``abi.WB_COLUMNS``), reaches, data[11, D, E], info); needs daily rows and adds the
    four flux columns / the summed reaches to the outputs when missing.  ``waterbody_obs``: DataFrame of observations
    at the waterbody's inflow (columns among Q, SS, TDP, PP, TP, SRP): the members' goodness-of-fit table of the summed
    series under ``['waterbody']['gof']``.

    ``devices``: a list of GPU ids, e.g. ``[0, 1, ..., 7]`` -- the ensemble is split into contiguous member blocks
    (``ensemble.shard_bounds``), one per entry, and the blocks run at the same time from this one process: one engine context and
    one host thread per entry (ctypes releases the GIL), every block's table streamed over its own GPU's PCIe link, the
    goodness-of-fit / waterbody reductions done where the block's table lies.  No process group, no ``torchrun``: a notebook call
    (SURVEY.md section 8b's ``devices=`` argument; what the reference's only ensemble caller did with an ``IPython.parallel`` pool,
    Development/2016/MCMC.ipynb:30-31, :379-385).  Members are independent, so every table is bit-identical to the one-device
    run's.  An id may repeat (``[0, 0]``: two contexts on one GPU).  ``device`` is ignored when ``devices`` is given.
    ``stats`` then holds sums / maxima over the blocks and the per-block dicts under ``'per_device'``; with ``to_host=False``
    ``data`` is the list of the blocks' device tensors (member axis split as ``stats['bounds']``).

    Returns ``dict(columns, reaches, data[n_cols, D or n_periods, n_reaches, E], status[E], stats)``; ``data``
    and ``status`` are numpy arrays, or device tensors when ``to_host`` is False.
    The caller's ``p_LU``/``p_SC`` are edited in place exactly as by ``run_simply_p``.
    """
    from . import engine

    marshal.prologue(p_SU, p_LU, p_SC, p)
    scs = marshal.sc_list(p)
    up_ptr, up_idx, _ = marshal.topology(p_struc, p)
    overrides = dict(overrides or {})
    f_tdp = overrides.pop('f_TDP', None)
    m_over, r_over = marshal.split_member_reach_overrides(overrides)
    sizes = {np.asarray(v).shape[-1] for v in list(m_over.values()) + list(r_over.values()) + [f_tdp] if v is not None and np.ndim(v) > 0}
    if n_members is None:
        if len(sizes) != 1:
            raise ValueError("cannot infer the ensemble size: give n_members or override arrays of one length")
        n_members = sizes.pop()
    E = int(n_members)
    # the SoA arrays are marshalled straight into page-locked host memory: the uploads are asynchronous DMA transfers
    pin = engine.pinned_empty
    mp = marshal.member_params(p, p_LU, E, m_over, alloc=pin)
    rp = marshal.reach_params(p_SC, p, E, r_over, alloc=pin)
    if m_over or r_over:
        marshal.validate_ensemble(mp, rp, scs)       # the reference's checks, for every member (model.py:321-335, :355-357)
    if snow_in_kernel is None:
        snow_in_kernel = 'f_DDSM' in m_over or 'D_snow_0' in m_over
    met_sets = list(met_df) if isinstance(met_df, (list, tuple)) else [met_df]
    met_df = met_sets[0]
    for m in met_sets:
        if snow_in_kernel and not {'Precipitation', 'T_air'} <= set(m.columns):
            raise ValueError("the in-kernel snow module needs met_df columns 'Precipitation' and 'T_air'")
        if not m.index.equals(met_df.index):
            raise ValueError("all forcing sets must cover the same dates")
    if len(met_sets) > 1:
        if forcing_of_member is None:
            raise ValueError("several forcing sets need forcing_of_member (one set index per member)")
        forcing_of_member = np.ascontiguousarray(forcing_of_member, dtype=np.int32)
        if forcing_of_member.shape != (E,) or forcing_of_member.min() < 0 or forcing_of_member.max() >= len(met_sets):
            raise ValueError("forcing_of_member needs one index in [0, %d) per member" % len(met_sets))
    elif forcing_of_member is not None:
        raise ValueError("forcing_of_member given but met_df is a single forcing set")
    parts = [marshal.forcing_arrays(m, snow=snow_in_kernel) for m in met_sets]
    forcing, doy = np.ascontiguousarray(np.concatenate([f for f, _ in parts], axis=0)), parts[0][1]
    cols = list(outputs) if outputs is not None else list(marshal.REACH5_COLUMNS)
    wb_reaches = None
    if waterbody is not None and waterbody is not False:
        if reduce is not None:
            raise ValueError("the waterbody sum needs the daily series: waterbody cannot be combined with reduce")
        if waterbody is True:
            flags = p_struc['In_final_flux?']
            wb_reaches = [int(r) for r in flags[flags == 1].index.values]                    # model.py:867
            if len(wb_reaches) > len(scs):                                                       # :871-872
                raise ValueError("Mismatch between the number of subcatchments in the 'Setup' parameter sheet \n(parameter 'n_SC') and in the 'Reach_structure' parameter sheet")
        else:
            wb_reaches = sorted(int(r) for r in waterbody)
        print('Sub-catchments flowing directly into receiving waterbody: %s' % np.asarray(wb_reaches))   # :874
        if out_reaches is not None:
            out_reaches = list(out_reaches) + [r for r in wb_reaches if r not in out_reaches]
    if obs_dict is not None or wb_reaches is not None:
        if reduce is not None:
            raise ValueError("goodness of fit needs the daily series: obs_dict cannot be combined with reduce")
        cols += [c for c in ('Qr', 'Msus_kg/day', 'TDP_kg/day', 'PP_kg/day') if c not in cols]
    mask = marshal.mask_of_columns(cols)
    if mask & marshal.MASK_D_SNOW and not snow_in_kernel:
        raise ValueError("output 'D_snow' is the per-member snow depth of the in-kernel snow module: needs snow_in_kernel=True "
                         "(without it every member shares met_df['D_snow_end'])")
    period_of_day, periods = None, None
    if reduce is not None:
        if isinstance(reduce, str):
            if reduce != 'annual':
                raise ValueError("reduce must be None, 'annual' or an array of period indices")
            years = np.asarray(met_df.index.year)
            periods, period_of_day = np.unique(years, return_inverse=True)
        else:
            period_of_day = np.asarray(reduce)
            if period_of_day.shape != (len(met_df),) or period_of_day.min() < 0:
                raise ValueError("reduce array needs one non-negative period index per day")
            periods = np.arange(int(period_of_day.max()) + 1)
        period_of_day = np.ascontiguousarray(period_of_day, dtype=np.int32)
    opts = _engine_opts(p_SU, p, dynamic_options, step_len, solver, mask,
                        n_periods=0 if periods is None else len(periods), snow=snow_in_kernel)
    oreach = None if out_reaches is None else [scs.index(int(r)) for r in out_reaches]

    reaches = scs if out_reaches is None else list(out_reaches)
    n_or_ = len(scs) if oreach is None else len(oreach)
    rows_ = len(met_df) if periods is None else len(periods)
    ncols_ = bin(mask).count('1')
    want_host_table = to_host and (keep_daily or obs_dict is None)
    obs = wobs = None
    if obs_dict is not None or (wb_reaches is not None and len(wb_reaches) > 1 and waterbody_obs is not None):
        from . import visualise_results as vr
        if obs_dict is not None:
            obs = vr.observation_array(obs_dict, reaches, met_df.index)
        if wb_reaches is not None and len(wb_reaches) > 1 and waterbody_obs is not None:
            wobs = vr.observation_array({0: waterbody_obs}, [0], met_df.index)[0]
    ft_all = p['f_TDP'] if f_tdp is None else f_tdp

    def block_pass(eng, lo, hi):
        """Members [lo, hi) on one engine context: the run, then the reductions that want the table where it lies."""
        whole = (lo == 0 and hi == E)
        mp_b = mp if whole else np.ascontiguousarray(mp[:, lo:hi])
        rp_b = rp if whole else np.ascontiguousarray(rp[:, :, lo:hi])
        fom_b = None if forcing_of_member is None else np.ascontiguousarray(forcing_of_member[lo:hi])
        ft = ft_all if np.ndim(ft_all) == 0 else np.ascontiguousarray(np.asarray(ft_all, dtype=np.float64)[lo:hi])
        rp_d = eng.to_device(rp_b)
        # to_host: the table is delivered into page-locked host memory while the kernel runs (simplyp_stream_out) -- unless the
        # caller only wants the statistics (keep_daily=False)
        host_out = _host_table((ncols_, rows_, n_or_, hi - lo), pin) if want_host_table else None
        out_d, status_d, stats = eng.run(forcing, doy, mp_b, rp_d, up_ptr, up_idx, opts, out_reaches=oreach,
                                         period_of_day=period_of_day, forcing_of_member=fom_b, host_out=host_out)
        part = dict(stats=stats, status=status_d.cpu().numpy() if to_host else status_d, host_out=host_out)
        mos = stats.get('member_of_slot') if opts.out_slot_order else None
        if obs is not None:
            gof_d, info = eng.gof(out_d, mask, obs, ft, rp_d, out_reaches=oreach, member_of_slot=mos, spearman=spearman)
            rho = info.pop('spearman', None)
            part['gof'] = (gof_d.cpu().numpy() if to_host else gof_d, info,
                           None if rho is None else (rho.cpu().numpy() if to_host else rho))
        if wb_reaches is not None and len(wb_reaches) > 1:
            wb_d, winfo = eng.waterbody(out_d, mask, [scs.index(r) for r in wb_reaches], ft, rp_d, out_reaches=oreach,
                                        member_of_slot=mos)
            part['wb'] = (wb_d.cpu().numpy() if to_host else wb_d, winfo)
            if wobs is not None:
                g_d, ginfo = eng.gof_waterbody(wb_d, winfo['columns'], wobs, ft, member_of_slot=mos)
                part['wb_gof'] = (g_d.cpu().numpy() if to_host else g_d, ginfo)
        part['out_d'] = None if (obs_dict is not None and not keep_daily) else out_d
        return part

    if devices is None:
        parts, bounds = [block_pass(engine.get_engine(device), 0, E)], [(0, E)]
    else:
        from . import ensemble
        devs = [int(d) for d in devices]
        if not devs:
            raise ValueError("devices must name at least one GPU")
        if opts.out_slot_order:
            raise ValueError("devices=[...] returns tables in member order: solver['out_slot_order'] must stay 0")
        bounds = [ensemble.shard_bounds(E, len(devs), r) for r in range(len(devs))]
        # one context per list entry (a repeated id gets a context of its own: contexts are not re-entrant)
        engs = [engine.get_engine(d, replica=devs[:r].count(d)) for r, d in enumerate(devs)]
        live = [(eng_, lo, hi) for eng_, (lo, hi) in zip(engs, bounds) if hi > lo]
        bounds = [(lo, hi) for _, lo, hi in live]
        if len(live) == 1:
            parts = [block_pass(*live[0])]
        else:
            from concurrent.futures import ThreadPoolExecutor
            with ThreadPoolExecutor(max_workers=len(live)) as pool:
                parts = list(pool.map(lambda a_: block_pass(*a_), live))
    marshal.epilogue_mutations(p_SU, p_LU, p_SC, p)

    cat = (lambda xs: np.concatenate(xs, axis=-1)) if to_host else (lambda xs: xs[0] if len(xs) == 1 else _torch_cat(xs))
    if len(parts) == 1:
        stats = parts[0]['stats']
    else:
        per = [pt['stats'] for pt in parts]
        stats = {k: sum(d_[k] for d_ in per) for k in ('rhs_evals', 'steps', 'rejected', 'n_launches', 'streamed_chunks',
                                                         'queue_waits')}
        stats.update({k: max(d_[k] for d_ in per) for k in ('kernel_ms', 'pilot_ms', 'wall_ms', 'd2h_tail_ms',
                                                            'queue_longest_wait_polls', 'queue_longest_stall_polls')})
        for k in ('balanced', 'queued', 'lanes_per_wave', 'lanes_per_member'):
            stats[k] = per[0][k]
        stats['simt_efficiency'] = float(np.mean([d_['simt_efficiency'] for d_ in per]))
        stats['per_device'] = [dict(d_, device=int(dv), members=[int(lo), int(hi)])
                               for d_, dv, (lo, hi) in zip(per, [e_.device for e_, _, _ in live], bounds)]
    stats['bounds'] = [[int(lo), int(hi)] for lo, hi in bounds]
    res = dict(columns=marshal.columns_of_mask(mask), reaches=reaches, periods=periods, stats=stats,
               status=cat([pt['status'] for pt in parts]))
    if obs is not None:
        res['gof'] = dict(stats=list(abi.GOF_STATS), variables=list(abi.GOF_VARS), info=parts[0]['gof'][1],
                          data=cat([pt['gof'][0] for pt in parts]))
        if parts[0]['gof'][2] is not None:       # Spearman's r [6, n_reaches, E]: the rank statistic of the reference's table (:444-445)
            res['gof']['spearman'] = cat([pt['gof'][2] for pt in parts])
    if wb_reaches is not None:
        if len(wb_reaches) > 1:
            winfo = parts[0]['wb'][1]
            res['waterbody'] = dict(columns=winfo['columns'], reaches=wb_reaches, info=winfo, data=cat([pt['wb'][0] for pt in parts]))
            if wobs is not None:
                res['waterbody']['gof'] = dict(stats=list(abi.GOF_STATS), variables=list(abi.GOF_VARS), info=parts[0]['wb_gof'][1],
                                               data=cat([pt['wb_gof'][0] for pt in parts]))
        else:
            print('One or fewer reaches were selected to be included in the sum, check your reach structure parameters')   # :896
            res['waterbody'] = None
    if obs_dict is not None and not keep_daily:
        res['data'] = None
    elif not to_host:
        res['data'] = parts[0]['out_d'] if len(parts) == 1 else [pt['out_d'] for pt in parts]
    elif len(parts) == 1:
        res['data'] = parts[0]['host_out']
    else:
        # one table in member order: every block's page-locked table is copied into its member range (one host thread per
        # block; numpy releases the GIL in the copy), block by block freed
        data = np.empty((ncols_, rows_, n_or_, E), dtype=np.float64)

        def place(k):
            lo, hi = bounds[k]
            data[..., lo:hi] = parts[k]['host_out']
            parts[k]['host_out'] = None
        from concurrent.futures import ThreadPoolExecutor
        with ThreadPoolExecutor(max_workers=len(parts)) as pool:
            list(pool.map(place, range(len(parts))))
        res['data'] = data
    return res


def _torch_cat(xs):
    import torch
    return torch.cat([x.to(xs[0].device) for x in xs], dim=-1)
