"""Synthetic ensembles on the Tarland example (the BASELINE.json configurations).

Everything starts from the reference's shipped parameter workbook and 30-year met file (data copies
under data/ at the repository root, read through the package's own read_input_data), so the bench and
the full-size tests go through the same host path a user's notebook would.
"""

import os

import numpy as np

from . import inputs, marshal

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
REF_DATA = os.path.join(REPO, 'data')
WORKBOOK = os.path.join(REF_DATA, 'Parameters_v0-2A_Tarland.xlsx')
MET_CSV = os.path.join(REF_DATA, 'Tarland_MetData_1981-2010.csv')

Q_OBS = os.path.join(REF_DATA, 'Coull_DailyMeanQ.xlsx')
CHEM_OBS = os.path.join(REF_DATA, 'Coull_ChemObs.xlsx')

C3_SEED = 20240601


def tarland_inputs(st_dt='1981-01-01', end_dt='2010-12-31', dynamic_epc0='y', dynamic_erod='n', quiet=True):
    """The reference-shaped inputs of the Tarland example for a period: the 7 arguments of run_simply_p."""
    import contextlib
    import io
    over = dict(metdata_fpath=MET_CSV, Qobsdata_fpath=None, chemObsData_fpath=None, st_dt=st_dt, end_dt=end_dt,
                Dynamic_EPC0=dynamic_epc0, Dynamic_erodibility=dynamic_erod)
    with (contextlib.redirect_stdout(io.StringIO()) if quiet else contextlib.nullcontext()):
        p_SU, dyn, p, p_LU, p_SC, p_struc, met_df, _ = inputs.read_input_data(WORKBOOK, setup_overrides=over)
    return met_df, p_struc, p_SU, p_LU, p_SC, p, dyn


def tarland_observations(st_dt='1981-01-01', end_dt='2010-12-31'):
    """obs_dict (sub-catchment -> DataFrame of observed Q / SS / TDP / PP / TP / SRP) of the Tarland example for a period,
    as the package's read_input_data returns it from the shipped observation workbooks (reference inputs.py:118-152)."""
    import contextlib
    import io
    over = dict(metdata_fpath=MET_CSV, Qobsdata_fpath=Q_OBS, chemObsData_fpath=CHEM_OBS, st_dt=st_dt, end_dt=end_dt)
    with contextlib.redirect_stdout(io.StringIO()):
        obs_dict = inputs.read_input_data(WORKBOOK, setup_overrides=over)[7]
    return obs_dict


def monte_carlo_overrides(p, p_LU, n_members, seed=C3_SEED):
    """Config C3's parameter distribution: independent draws per member,
    log-uniform x/÷2 around the base value for T_s[A], T_s[S], T_g, a_Q, E_M, E_PP, TDPg, EPC0_init_mgl[A];
    uniform +-25 % for fc, beta, f_quick; k_M in [1.5, 2.5], b_Q in [0.3, 0.5], Qg_min in [0, 0.6].
    Land-use fractions stay fixed (keeps f_A + f_S == 1 exactly)."""
    rng = np.random.default_rng(seed)
    E = int(n_members)
    base = marshal.member_params(p, p_LU, 1)[:, 0]
    val = lambda name: base[marshal.PM_NAMES.index(name)]
    over = {}
    for name in ('T_s_A', 'T_s_S', 'T_g', 'a_Q', 'E_M', 'E_PP', 'TDPg', 'EPC0_init_mgl_A'):
        over[name] = val(name) * np.exp(rng.uniform(-np.log(2.0), np.log(2.0), E))
    for name in ('fc', 'beta', 'f_quick'):
        over[name] = val(name) * rng.uniform(0.75, 1.25, E)
    over['k_M'] = rng.uniform(1.5, 2.5, E)
    over['b_Q'] = rng.uniform(0.3, 0.5, E)
    over['Qg_min'] = rng.uniform(0.0, 0.6, E)
    return over


WIDE_NAMES = ('T_s_A', 'T_s_S', 'T_g', 'a_Q', 'E_M', 'f_quick')


def widen_overrides(over, p, p_LU, n_members, seed, wide):
    """A wider parameter distribution than BASELINE C3's: the soil, groundwater and reach time scales / rates of `over`
    (``monte_carlo_overrides``; parameters it does not draw start from the workbook value) times a further log-uniform factor in
    [1/wide, wide], drawn with ``default_rng(seed + 2000)`` in the order of ``WIDE_NAMES`` -- the draw of
    tools/probe_tolerance.py's SIMPLYP_PROBE_WIDE and of the dry-reach fixture (tests/golden/make_golden.py --only dry)."""
    rng = np.random.default_rng(seed + 2000)
    base = marshal.member_params(p, p_LU, 1)[:, 0]
    out = dict(over)
    for name in WIDE_NAMES:
        cur = np.asarray(out[name], dtype=float) if name in out else np.full(int(n_members), base[marshal.PM_NAMES.index(name)])
        out[name] = cur * np.exp(rng.uniform(-np.log(wide), np.log(wide), int(n_members)))
    return out


def c3_problem(n_members, st_dt='1981-01-01', end_dt='2010-12-31', seed=C3_SEED, solver=None,
               out_mask=marshal.MASK_REACH5, replicated=False):
    """Arrays + options of the Tarland Monte-Carlo ensemble (BASELINE config C3; `replicated=True` gives
    config C2: every member = the base parameters).  Returns a dict for Engine.run / oracle.run."""
    from . import abi
    met_df, p_struc, p_SU, p_LU, p_SC, p, dyn = tarland_inputs(st_dt, end_dt)
    marshal.prologue(p_SU, p_LU, p_SC, p)
    up_ptr, up_idx, _ = marshal.topology(p_struc, p)
    over = {} if replicated else monte_carlo_overrides(p, p_LU, n_members, seed)
    mp = marshal.member_params(p, p_LU, n_members, over)
    rp = marshal.reach_params(p_SC, p, n_members)
    forcing, doy = marshal.forcing_arrays(met_df)
    scs = marshal.sc_list(p)
    opts = abi.make_opts(solver, dynamic_epc0=dyn['Dynamic_EPC0'] == 'y', dynamic_erod=dyn['Dynamic_erodibility'] == 'y',
                         run_mode_cal=p_SU.run_mode == 'cal', sc_qr0=scs.index(int(p['SC_Qr0'])), out_mask=out_mask)
    return dict(forcing=forcing, doy=doy, member_params=mp, reach_params=rp, up_ptr=up_ptr, up_idx=up_idx,
                opts=opts, met=met_df)


C4_SEED = 20240602


def c4_inputs(n_reaches=256, seed=C4_SEED, st_dt='1981-01-01', end_dt='2010-12-31'):
    """The reference-shaped inputs (the 7 arguments of run_simply_p) of BASELINE config C4's synthetic reach chain: the Tarland
    workbook with `n_reaches` sub-catchments in a line (reach i drains reach i - 1), both dynamic options on.  Per-reach
    parameters are drawn reach by reach, so a shorter chain is the upper end of the 256-reach one."""
    import pandas as pd
    met_df, p_struc, p_SU, p_LU, p_SC, p, dyn = tarland_inputs(st_dt, end_dt, dynamic_epc0='y', dynamic_erod='y')
    rng = np.random.default_rng(seed)
    S = int(n_reaches)
    frac = np.round(rng.dirichlet([2.0, 3.0, 5.0], S) * 1024.0)
    frac[:, 2] = 1024.0 - frac[:, 0] - frac[:, 1]
    frac /= 1024.0
    cols = {}
    base = p_SC[1]
    for s in range(S):
        col = base.copy()
        col['A_catch'] = float(np.exp(rng.uniform(np.log(5.0), np.log(50.0))))
        col['L_reach'] = float(rng.uniform(2000.0, 15000.0))
        col['S_Ar'], col['S_IG'], col['S_SN'] = (float(x) for x in rng.uniform(1.0, 12.0, 3))
        col['S_reach'] = float(rng.uniform(0.3, 2.5))
        col['f_Ar'], col['f_IG'], col['f_S'] = (float(x) for x in frac[s])
        col['f_NC_Ar'] = 0.1 if s % 4 == 3 else 0.0
        col['f_NC_IG'] = 0.0
        col['f_NC_S'] = 0.0
        col['TDPeff'] = float(rng.uniform(0.0, 0.3))
        cols[s + 1] = col
    p_SC = pd.DataFrame(cols)
    p = p.copy()
    p['SC_list'] = np.arange(1, S + 1)
    p['SC_Qr0'] = float(S)
    p_SU = p_SU.copy()
    p_SU['n_SC'] = S
    p_struc = pd.DataFrame({'Upstream_SCs': pd.Series([np.nan] + list(range(1, S)), index=range(1, S + 1), dtype=object),
                            'In_final_flux?': pd.Series([0] * (S - 1) + [1], index=range(1, S + 1))})
    p_struc.index.name = 'Reach'
    return met_df, p_struc, p_SU, p_LU, p_SC, p, dyn


def c4_problem(n_members, n_reaches=256, n_days=18262, seed=C4_SEED, solver=None, out_mask=marshal.MASK_REACH5,
               out_reaches='last'):
    """BASELINE config C4: a synthetic linear chain of `n_reaches` sub-catchments (reach i drains reach i-1),
    4 land-use classes active (IG erodibility, newly-converted arable land on every 4th reach), Tarland forcing
    tiled to `n_days` (1981-2010 then again from 1981), members drawn from the C3 distribution.

    Per-reach parameters: A_catch log-uniform 5-50 km2, L_reach 2-15 km, slopes 1-12
    degrees, land-use fractions Dirichlet(2,3,5) rounded to multiples of 1/1024 so that f_Ar + f_IG + f_S == 1
    exactly in floating point, f_NC_Ar = 0.1 on every 4th reach."""
    from . import abi
    met_df, p_struc, p_SU, p_LU, p_SC, p, dyn = c4_inputs(n_reaches, seed)
    S = int(n_reaches)
    marshal.prologue(p_SU, p_LU, p_SC, p)
    up_ptr, up_idx, _ = marshal.topology(p_struc, p)
    over = monte_carlo_overrides(p, p_LU, n_members, seed)
    mp = marshal.member_params(p, p_LU, n_members, over)
    rp = marshal.reach_params(p_SC, p, n_members)
    forcing, doy = marshal.forcing_arrays(met_df)
    reps = -(-int(n_days) // forcing.shape[2])
    forcing = np.ascontiguousarray(np.tile(forcing, (1, 1, reps))[:, :, :n_days])
    doy = np.ascontiguousarray(np.tile(doy, reps)[:n_days])
    opts = abi.make_opts(solver, dynamic_epc0=True, dynamic_erod=True, run_mode_cal=True, sc_qr0=S - 1,
                         out_mask=out_mask)
    oreach = [S - 1] if out_reaches == 'last' else out_reaches
    return dict(forcing=forcing, doy=doy, member_params=mp, reach_params=rp, up_ptr=up_ptr, up_idx=up_idx,
                opts=opts, out_reaches=oreach)
