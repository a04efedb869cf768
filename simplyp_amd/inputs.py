"""Input layer: parameter workbook, met data, snow module.

Same public names and return shapes as the reference's
``Current_Release/v0-2A/simplyP/inputs.py`` (``read_input_data`` :19-155,
``snow_hydrol_inputs`` :159-210, ``daily_PET`` :232), written for the current
pandas (no ``.ix``, no Excel engine: the workbook is read by ``xlsx.py``).
"""

import os

import numpy as np
import pandas as pd

from . import xlsx


def _resolve_path(path, workbook_path):
    """Find a data file named in the Setup sheet.

    The shipped workbook holds Windows-style relative paths
    (``..\\..\\Example_Data\\...``, Setup!C2-C5) meant to be resolved from the
    notebook's working directory.  Try the string as given, then with the
    separators normalised, relative to the cwd and to the workbook's folder.
    """
    if not isinstance(path, str):
        return path
    norm = path.replace('\\', os.sep)
    for cand in (path, norm, os.path.join(os.path.dirname(os.path.abspath(workbook_path)), norm)):
        if os.path.exists(cand):
            return cand
    return norm


def read_input_data(params_fpath, setup_overrides=None):
    """Read the SimplyP setup workbook, met data and (optional) observations.

    Returns the 8-tuple ``(p_SU, dynamic_options, p, p_LU, p_SC, p_struc,
    met_df, obs_dict)`` of reference inputs.py:155, with the same indices,
    column labels and error cases (:80-83).  ``setup_overrides`` (extension)
    replaces entries of the 'Setup' sheet before they are used, e.g.
    ``{'st_dt': '1981-01-01', 'metdata_fpath': '/data/met.csv'}``; a value of
    None blanks the entry (NaN), as an empty cell would.
    """
    wb = xlsx.Workbook(params_fpath)

    # Setup parameters (ref :44-50)
    p_SU = xlsx.read_excel(wb, 'Setup', index_col=0, usecols="A,C")['Value']
    for k, v in (setup_overrides or {}).items():
        p_SU[k] = np.nan if v is None else v
    dynamic_options = p_SU[['Dynamic_EPC0', 'Dynamic_effluent_inputs',
                            'Dynamic_terrestrialP_inputs', 'Dynamic_erodibility']]

    # Constants (ref :57-58)
    p = xlsx.read_excel(wb, 'Constant', index_col=0, usecols="B,E")['Value']

    # Land use parameters, columns A, S, IG, NC (ref :61)
    p_LU = xlsx.read_excel(wb, 'LU', index_col=0, usecols="B,E,F,G,H")
    p_LU = p_LU.astype(float)

    # Sub-catchment / reach parameters, one column per sub-catchment (ref :65-71)
    n_SC = int(p_SU.n_SC)
    p = p.astype(object)
    p['SC_list'] = np.arange(1, n_SC + 1)
    last_col = chr(ord('E') + n_SC - 1)
    usecols_str = "B,E" if n_SC == 1 else "B,E:%s" % last_col
    p_SC = xlsx.read_excel(wb, 'SC_reach', index_col=0, usecols=usecols_str)
    p_SC = p_SC.astype(float)

    # Reach structure (ref :76-77)
    p_struc = xlsx.read_excel(wb, 'Reach_structure', index_col=0, usecols="A,B,C")
    p_struc.columns = ['Upstream_SCs', 'In_final_flux?']

    if n_SC != len(p_struc['Upstream_SCs']):
        raise ValueError("The number of sub-catchments specified in your 'Setup' parameter sheet doesn't \nmatch the number of rows in your 'Reach_structure' sheet")
    if n_SC != len(p_SC.columns):
        raise ValueError("The number of columns in your 'SC_reach' sheet should match the number of sub-catchments specified in your 'Setup' parameter sheet")

    print('Parameter values successfully read in')

    # Met data (ref :91-106)
    met_path = _resolve_path(p_SU.metdata_fpath, params_fpath)
    met_df = pd.read_csv(met_path, parse_dates=True, dayfirst=True, index_col=0)
    met_df = met_df.truncate(before=p_SU.st_dt, after=p_SU.end_dt)
    print('Input meteorological data read in')

    if p_SU.inc_snowmelt == 'y':
        met_df = snow_hydrol_inputs(p['D_snow_0'], p['f_DDSM'], met_df)
        print('Snow accumulation and melt module run to estimate snowmelt inputs to the soil')
    else:
        met_df.rename(columns={'Precipitation': 'P'}, inplace=True)

    if 'PET' not in met_df.columns:
        met_df = daily_PET(latitude=p['latitude'], met_df=met_df)
        print('PET estimated using the Thornthwaite method')

    # Observations (ref :118-152)
    obs_dict = {}
    q_path = _resolve_path(p_SU.get('Qobsdata_fpath'), params_fpath)
    c_path = _resolve_path(p_SU.get('chemObsData_fpath'), params_fpath)
    q_wb = c_wb = None
    sc_q, sc_c = [], []
    if isinstance(q_path, str):
        q_wb = xlsx.Workbook(q_path)
        sc_q = [int(x) for x in q_wb.sheet_names]
        print('Observed discharge data read in')
    if isinstance(c_path, str):
        c_wb = xlsx.Workbook(c_path)
        sc_c = [int(x) for x in c_wb.sheet_names]
        print('Observed water chemistry data read in')
    for SC in p['SC_list']:
        df_li = []
        for wbk, present in ((q_wb, sc_q), (c_wb, sc_c)):
            if SC in present:
                df = xlsx.read_excel(wbk, str(SC), index_col=0)
                df.index = pd.to_datetime(df.index)
                df = df.sort_index().truncate(before=p_SU.st_dt, after=p_SU.end_dt)
                df_li.append(df)
        if df_li:
            obs_dict[SC] = pd.concat(df_li, axis=1)

    return (p_SU, dynamic_options, p, p_LU, p_SC, p_struc, met_df, obs_dict)


def snow_hydrol_inputs(D_snow_0, f_DDSM, met_df):
    """Degree-day snow accumulation and melt (reference inputs.py:159-210).

    Adds columns ``P_snow, P_rain, P_melt, D_snow_start, D_snow_end, P`` to
    ``met_df`` (in place, and returns it); ``P`` = rain + melt is the
    hydrological input the time-stepping engine consumes.  The snow-pack
    recurrence (:197-205) is a sequential scan over days; it runs here as a
    plain numpy loop over a handful of float arrays (O(D), not on the hot path).
    """
    precip = met_df['Precipitation'].to_numpy(dtype=float)
    t_air = met_df['T_air'].to_numpy(dtype=float)
    n = len(met_df)

    p_snow = np.where(t_air < 0, precip, 0.0)          # :183-184
    p_rain = precip - p_snow                           # :187
    p_melt = f_DDSM * (t_air - 0)                      # :190
    p_melt[p_melt < 0] = 0.0                           # :191

    d_start = np.full(n, np.nan)
    d_end = np.full(n, np.nan)
    depth = float(D_snow_0)
    for i in range(n):                                 # :197-205
        d_start[i] = depth
        p_melt[i] = min(p_melt[i], depth)
        depth = depth + p_snow[i] - p_melt[i]
        d_end[i] = depth

    met_df['P_snow'] = p_snow
    met_df['P_rain'] = p_rain
    met_df['P_melt'] = p_melt
    met_df['D_snow_start'] = d_start
    met_df['D_snow_end'] = d_end
    met_df['P'] = p_rain + p_melt                      # :208
    return met_df


def daily_PET(latitude, met_df):
    """Thornthwaite PET from air temperature (reference inputs.py:232-312).

    Out of scope for the time-stepping hot path:
    supply a ``PET`` column in the met data, as the Tarland example does.
    """
    raise NotImplementedError(
        "daily_PET (Thornthwaite) is not part of the MI355X engine; "
        "provide a 'PET' column in the meteorological input file")
