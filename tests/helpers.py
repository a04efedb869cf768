"""Shared test helpers: rebuild the reference-shaped pandas inputs of a golden scenario,
and compare result tables."""

import json
import os

import numpy as np
import pandas as pd

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), 'golden')
DATA = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), 'data')      # the reference's shipped data files (workbook, met series, observations, example outputs)
_meta = None


def meta():
    global _meta
    if _meta is None:
        with open(os.path.join(GOLDEN, 'series_meta.json')) as fh:
            _meta = json.load(fh)
    return _meta


def _nan(v):
    return np.nan if v is None else v


def scenario_inputs(name):
    """(met_df, p_struc, p_SU, p_LU, p_SC, p, dynamic_options) exactly as the golden run got them."""
    info = meta()[name]['inputs']
    z = np.load(os.path.join(GOLDEN, name + '.npz'), allow_pickle=False)
    met = pd.DataFrame(z['met/values'], columns=[str(c) for c in z['met/columns']],
                       index=pd.to_datetime([str(d) for d in z['met/index']]))
    met.index.name = 'Date'
    p_SU = pd.Series({k: _nan(v) for k, v in info['p_SU'].items()}, dtype=object)
    pd_ = {k: _nan(v) for k, v in info['p'].items()}
    pd_['SC_list'] = np.asarray(pd_['SC_list'])
    p = pd.Series(pd_, dtype=object)
    p_LU = pd.DataFrame({c: {k: _nan(v) for k, v in col.items()} for c, col in info['p_LU'].items()}, dtype=float)
    p_LU = p_LU[['A', 'S', 'IG', 'NC']]
    p_SC = pd.DataFrame({int(c): {k: _nan(v) for k, v in col.items()} for c, col in info['p_SC'].items()}, dtype=float)
    rows = {int(i): {k: _nan(v) for k, v in row.items()} for i, row in info['p_struc'].items()}
    p_struc = pd.DataFrame.from_dict(rows, orient='index')
    p_struc['Upstream_SCs'] = p_struc['Upstream_SCs'].astype(object)
    p_struc.index.name = 'Reach'
    dyn = pd.Series(dict(info['dyn'], Dynamic_effluent_inputs='n', Dynamic_terrestrialP_inputs='n'))
    return met, p_struc, p_SU, p_LU, p_SC, p, dyn


def golden_tables(name, label):
    """{'R': {SC: DataFrame}, 'TC': {SC: DataFrame}} of the golden run `label` ('tight' | 'shipped')."""
    z = np.load(os.path.join(GOLDEN, name + '.npz'), allow_pickle=False)
    idx = pd.to_datetime([str(d) for d in z['met/index']])
    out = {'R': {}, 'TC': {}}
    for key in z.files:
        parts = key.split('/')
        if parts[0] != label or len(parts) != 2:
            continue
        kind = 'R' if parts[1].startswith('R') else 'TC'
        sc = int(parts[1][len(kind):])
        out[kind][sc] = pd.DataFrame(z[key], columns=[str(c) for c in z[key + '/columns']], index=idx)
    return out


def max_rel_err(a, b, floor=0.0):
    """max over elements of |a-b| / max(|b|, floor); NaN-in-both counts as equal."""
    a = np.asarray(a, dtype=float); b = np.asarray(b, dtype=float)
    both_nan = np.isnan(a) & np.isnan(b)
    denom = np.maximum(np.abs(b), floor)
    with np.errstate(divide='ignore', invalid='ignore'):
        r = np.abs(a - b) / denom
    r = np.where((a == b) | both_nan, 0.0, r)
    return float(np.nanmax(r)) if r.size else 0.0


def marshal_scenario(name, E=1, out_mask=None, solver=None, snow=False):
    """Arrays + opts for engine/oracle `run` from a golden scenario (base parameters replicated E times)."""
    from simplyp_amd import marshal, abi
    met, p_struc, p_SU, p_LU, p_SC, p, dyn = scenario_inputs(name)
    marshal.prologue(p_SU, p_LU, p_SC, p)
    up_ptr, up_idx, _ = marshal.topology(p_struc, p)
    mp = marshal.member_params(p, p_LU, E)
    rp = marshal.reach_params(p_SC, p, E)
    forcing, doy = marshal.forcing_arrays(met, snow=snow)
    scs = marshal.sc_list(p)
    opts = abi.make_opts(solver, dynamic_epc0=dyn['Dynamic_EPC0'] == 'y', dynamic_erod=dyn['Dynamic_erodibility'] == 'y',
                         run_mode_cal=p_SU.run_mode == 'cal', sc_qr0=scs.index(int(p['SC_Qr0'])),
                         out_mask=marshal.MASK_ALL if out_mask is None else out_mask, snow=snow)
    return dict(forcing=forcing, doy=doy, member_params=mp, reach_params=rp, up_ptr=up_ptr, up_idx=up_idx, opts=opts,
                scs=scs, met=met)


def observations(st_dt, end_dt):
    """obs_dict of the shipped Tarland observation workbooks (data/ at the repository root), truncated to the run
    period the way the reference's read_input_data does (inputs.py:118-152)."""
    from simplyp_amd import xlsx
    df_li = []
    for f in ('Coull_DailyMeanQ.xlsx', 'Coull_ChemObs.xlsx'):
        wb = xlsx.Workbook(os.path.join(DATA, f))
        df = xlsx.read_excel(wb, '1', index_col=0)
        df.index = pd.to_datetime(df.index)
        df_li.append(df.sort_index().truncate(before=st_dt, after=end_dt))
    return {1: pd.concat(df_li, axis=1)}


def gof_golden():
    with open(os.path.join(GOLDEN, 'gof_golden.json')) as fh:
        return json.load(fh)


def gof_case_factors(index):
    """The deterministic multiplicative perturbations of tests/golden/make_gof_golden.py (case -> Q and concentration factors)."""
    t = np.arange(len(index)) / 365.25
    return {'base': dict(Q=1.0, C=1.0),
            'wet': dict(Q=1.25 + 0.1 * np.sin(2 * np.pi * t), C=0.8),
            'dry': dict(Q=0.7, C=1.4 + 0.3 * np.cos(2 * np.pi * t / 3.0))}


# Two valid integrations at the default solver's working tolerance (kernel and oracle mirror the same step rule, but an
# accept/reject decision may flip on rounding) agree to about ten times that tolerance.
def _working_tolerance():
    from simplyp_amd import abi
    return 10.0 * abi.DEFAULT_SOLVER['rtol']


TOL_WORKING = _working_tolerance()


def member_fixture_problem(fname, solver=None):
    """Arrays + opts for the members of a per-member reference fixture (tests/golden/knee_members.npz, heldout_members.npz, wide_members.npz:
    members of a C3-distribution draw run one by one through the unmodified reference at rtol=atol=1e-12 by
    tests/golden/make_golden.py) and their reference tables: (problem dict, [table[D, 9 reach columns] per member]).
    The parameter values are regenerated from the recorded seed and checked against what the fixture recorded."""
    from simplyp_amd import synthetic, marshal, abi
    z = np.load(os.path.join(GOLDEN, fname), allow_pickle=False)
    members = [int(m) for m in z['members']]
    years = [str(y) for y in z['years']]
    met_df, p_struc, p_SU, p_LU, p_SC, p, dyn = synthetic.tarland_inputs(years[0], years[1], dynamic_epc0='y', dynamic_erod='n')
    over_all = synthetic.monte_carlo_overrides(p, p_LU, int(z['n_draw']), seed=int(z['seed']))
    if 'wide' in z.files:          # wide_members.npz: the draw's time constants and rates widened (synthetic.widen_overrides)
        over_all = synthetic.widen_overrides(over_all, p, p_LU, int(z['n_draw']), int(z['seed']), float(z['wide']))
    over = {k: v[members] for k, v in over_all.items()}
    for k, nm in enumerate(str(n) for n in z['names']):                # the generator still draws what the fixture recorded
        np.testing.assert_array_equal(over[nm], z['values'][k])
    marshal.prologue(p_SU, p_LU, p_SC, p)
    up_ptr, up_idx, _ = marshal.topology(p_struc, p)
    n = len(members)
    mp = marshal.member_params(p, p_LU, n, over)
    rp = marshal.reach_params(p_SC, p, n)
    forcing, doy = marshal.forcing_arrays(met_df)
    opts = abi.make_opts(solver, dynamic_epc0=True, run_mode_cal=True)
    assert [str(c) for c in z['columns']] == ['Vr', 'Qr_EndOfDay', 'Qr', 'Msus_EndOfDay', 'Msus_kg/day', 'TDPr_EndOfDay',
                                              'TDP_kg/day', 'PPr_EndOfDay', 'PP_kg/day']
    tables = [z['R/%d' % m] for m in members]
    return dict(forcing=forcing, doy=doy, member_params=mp, reach_params=rp, up_ptr=up_ptr, up_idx=up_idx, opts=opts,
                members=members, met=met_df), tables


def dry_fixture_problem(solver=None):
    """Arrays + opts for the ten members of tests/golden/dry_members.npz -- the dry-reach regime: members of two 100 000-member draws
    (one with the time constants widened x/÷ 2) with Qg_min ~ 0 on a climate with 0.6 x Tarland's precipitation and 1/0.6 x its PET,
    each run through the unmodified reference at rtol=atol=1e-12 from 1981-01-01 to a year past its worst day
    (tests/golden/make_golden.py --only dry) -- and what to compare: (problem dict over the days [0, max window end),
    [(lo, hi, table[hi - lo, 9 reach columns]) per member]).  Parameter values are regenerated from the recorded seeds and checked."""
    from simplyp_amd import synthetic, marshal, abi
    z = np.load(os.path.join(GOLDEN, 'dry_members.npz'), allow_pickle=False)
    members = [int(m) for m in z['members']]
    windows = [(int(lo), int(hi)) for lo, hi in z['window']]
    n_days = max(hi for _, hi in windows)
    days = pd.date_range('1981-01-01', periods=n_days)
    met_df, p_struc, p_SU, p_LU, p_SC, p, dyn = synthetic.tarland_inputs('1981-01-01', days[-1].strftime('%Y-%m-%d'), dynamic_epc0='y',
                                                                         dynamic_erod='n')
    names = [str(n) for n in z['names']]
    over = {nm: np.empty(len(members)) for nm in names}
    draws = {}
    for k, (m, dseed, wide) in enumerate(zip(members, z['seed_offset'], z['wide'])):
        key = (int(dseed), float(wide))
        if key not in draws:
            seed = synthetic.C3_SEED + int(dseed)
            o = synthetic.monte_carlo_overrides(p, p_LU, int(z['n_draw']), seed=seed)
            draws[key] = o if wide == 1.0 else synthetic.widen_overrides(o, p, p_LU, int(z['n_draw']), seed, float(wide))
        for nm in names:
            over[nm][k] = draws[key][nm][m]
    for k, nm in enumerate(names):                                     # the generators still draw what the fixture recorded
        np.testing.assert_array_equal(over[nm], z['values'][k])
    marshal.prologue(p_SU, p_LU, p_SC, p)
    up_ptr, up_idx, _ = marshal.topology(p_struc, p)
    n = len(members)
    mp = marshal.member_params(p, p_LU, n, over)
    rp = marshal.reach_params(p_SC, p, n)
    forcing, doy = marshal.forcing_arrays(met_df)
    forcing = forcing.copy()
    forcing[:, 0] *= float(z['pscale'])                                # the hydrological input x 0.6, PET / 0.6 (make_golden.run_reference)
    forcing[:, 1] /= float(z['pscale'])
    opts = abi.make_opts(solver, dynamic_epc0=True, run_mode_cal=True)
    assert [str(c) for c in z['columns']] == ['Vr', 'Qr_EndOfDay', 'Qr', 'Msus_EndOfDay', 'Msus_kg/day', 'TDPr_EndOfDay',
                                              'TDP_kg/day', 'PPr_EndOfDay', 'PP_kg/day']
    tables = [(lo, hi, z['R/%d' % k]) for k, (lo, hi) in enumerate(windows)]
    return dict(forcing=forcing, doy=doy, member_params=mp, reach_params=rp, up_ptr=up_ptr, up_idx=up_idx, opts=opts,
                members=members, met=met_df), tables


def dry_worst_per_member(got, tables, columns):
    """max relative error of every fixture member over its window and the 9 reach columns; got [n_cols, D, 1, E] with `columns`."""
    cols = ['Vr', 'Qr_EndOfDay', 'Qr', 'Msus_EndOfDay', 'Msus_kg/day', 'TDPr_EndOfDay', 'TDP_kg/day', 'PPr_EndOfDay', 'PP_kg/day']
    return [max(max_rel_err(got[columns.index(c), lo:hi, 0, k], tab[:, j], floor=1e-300) for j, c in enumerate(cols))
            for k, (lo, hi, tab) in enumerate(tables)]


def c4_members_problem(solver=None, fname='c4_members.npz'):
    """Arrays + opts for tests/golden/c4_members.npz -- 4 members of BASELINE config C4's own distribution on the upper 16 reaches of
    its chain, 1981, run through the unmodified reference at rtol=atol=1e-12 (tests/golden/make_golden.py --only c4mc) -- and the
    reference tables: (problem dict with all 25 columns of the kept reaches, {(member, position among the kept reaches): table[366, 9]}).
    fname='c4_deep.npz': 2 members on the whole 256-reach chain, reaches 32, 64, 128, 192 and the outlet kept (--only c4deep)."""
    from simplyp_amd import synthetic, marshal
    z = np.load(os.path.join(GOLDEN, fname), allow_pickle=False)
    S, E = int(z['n_reaches']), int(z['n_members'])
    keep = [int(r) for r in z['reaches']]
    pr = synthetic.c4_problem(E, n_reaches=S, n_days=365, solver=solver, out_mask=marshal.MASK_ALL, out_reaches=[r - 1 for r in keep])
    for k, nm in enumerate(str(n) for n in z['names']):                 # the generator still draws what the fixture recorded
        np.testing.assert_array_equal(pr['member_params'][marshal.PM_NAMES.index(nm)], z['values'][k])
    tables = {(m, j): z['R/%d/%d' % (m, r)] for m in range(E) for j, r in enumerate(keep)}
    assert all(t.shape == (365, 9) for t in tables.values())
    return pr, tables


def c4_members_worst(got, tables):
    """max relative error over the 9 reach columns, per (member, kept reach); got [25, 365, n_kept, E]"""
    from simplyp_amd import marshal
    cols = ['Vr', 'Qr_EndOfDay', 'Qr', 'Msus_EndOfDay', 'Msus_kg/day', 'TDPr_EndOfDay', 'TDP_kg/day', 'PPr_EndOfDay', 'PP_kg/day']
    return {key: max(max_rel_err(got[marshal.OUT_COLUMNS.index(c), :, key[1], key[0]], tab[:, j], floor=1e-300) for j, c in enumerate(cols))
            for key, tab in tables.items()}
