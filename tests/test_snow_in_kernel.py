"""the degree-day snow module (reference inputs.py:159-210) as a per-member prologue of the
day loop (opts.snow = 1), so that ensembles can perturb f_DDSM / D_snow_0 without one forcing set per member.

The recurrence is a handful of IEEE operations in a fixed order, so the in-kernel P is bit-identical to the host
function's (which is bit-identical to the reference's own columns, tests/test_host.py): every comparison here is
exact equality of whole output tables, against runs that were fed host-computed P series."""

import numpy as np
import pytest

import helpers
import simplyp_amd as sp
from simplyp_amd import abi, marshal


def snow_members(E, seed=11):
    rng = np.random.default_rng(seed)
    f_ddsm = rng.uniform(0.5, 6.0, E)
    d0 = rng.uniform(0.0, 40.0, E)
    f_ddsm[0], d0[0] = 2.74, 0.0                     # member 0 = the workbook
    return f_ddsm, d0


def per_member_forcing(met, f_ddsm, d0):
    """[E][2][D]: P of every member from the host snow function, PET shared."""
    rows = []
    for f, d in zip(f_ddsm, d0):
        m = sp.snow_hydrol_inputs(d, f, met[['T_air', 'PET', 'Precipitation']].copy())
        rows.append(np.stack([m['P'].to_numpy(), m['PET'].to_numpy()]))
    return np.ascontiguousarray(np.stack(rows))


def setup(name, E, solver=None, out_mask=None):
    m1 = helpers.marshal_scenario(name, E=E, solver=solver, out_mask=out_mask, snow=True)
    m0 = helpers.marshal_scenario(name, E=E, solver=solver, out_mask=out_mask, snow=False)
    f_ddsm, d0 = snow_members(E)
    m1['member_params'][marshal.PM_NAMES.index('f_DDSM')] = f_ddsm
    m1['member_params'][marshal.PM_NAMES.index('D_snow_0')] = d0
    m0['forcing'] = per_member_forcing(m0['met'], f_ddsm, d0)
    m0['fom'] = np.arange(E, dtype=np.int32)
    return m1, m0


def test_forcing_rows_and_parameters():
    m1 = helpers.marshal_scenario('tarland_2004_static', E=2, snow=True)
    met = m1['met']
    assert m1['forcing'].shape == (1, 3, 366) and m1['opts'].snow == 1
    assert np.array_equal(m1['forcing'][0, 0], met['Precipitation'].values) and np.array_equal(m1['forcing'][0, 2], met['T_air'].values)
    assert marshal.PM_NAMES[-2:] == ['f_DDSM', 'D_snow_0'] and marshal.NP_M == 33
    assert (m1['member_params'][-2] == 2.74).all() and (m1['member_params'][-1] == 0.0).all()


@pytest.mark.parametrize('name', ['tarland_2004_dynamic', 'confluence3_nc_2004'])
def test_oracle_snow_prologue_equals_host_snow_function(oracle_lib, name):
    m1, m0 = setup(name, 5, solver=dict(integrator='rk4', substeps=16))
    a, sa, _ = oracle_lib.run(m1['forcing'], m1['doy'], m1['member_params'], m1['reach_params'], m1['up_ptr'], m1['up_idx'], m1['opts'])
    b, sb, _ = oracle_lib.run(m0['forcing'], m0['doy'], m0['member_params'], m0['reach_params'], m0['up_ptr'], m0['up_idx'], m0['opts'],
                              forcing_of_member=m0['fom'])
    assert np.array_equal(a, b, equal_nan=True) and sa.max() == 0
    assert not np.array_equal(a[..., 0], a[..., 1])          # the snow parameters matter


@pytest.mark.gpu
@pytest.mark.parametrize('name,solver', [('tarland_2004_dynamic', None), ('confluence3_nc_2004', dict(integrator='rk4', substeps=16)),
                                         ('tarland_1981_2010_dynamic', dict(rtol=1e-6, atol=1e-8))])
def test_kernel_snow_prologue_equals_host_snow_function(engine0, oracle_lib, name, solver):
    E = 130
    m1, m0 = setup(name, E, solver=solver, out_mask=marshal.MASK_REACH5 | marshal.mask_of_columns(['Qq', 'VsA']))
    a, sa, _ = engine0.run(m1['forcing'], m1['doy'], m1['member_params'], m1['reach_params'], m1['up_ptr'], m1['up_idx'], m1['opts'])
    b, sb, _ = engine0.run(m0['forcing'], m0['doy'], m0['member_params'], m0['reach_params'], m0['up_ptr'], m0['up_idx'], m0['opts'],
                           forcing_of_member=m0['fom'])
    import torch
    assert bool(torch.equal(a, b)) and int(sa.max()) == 0 and bool(torch.isfinite(a).all())
    if solver and solver.get('integrator') == 'rk4':            # and the oracle agrees (fixed step: rounding level)
        ref, _, _ = oracle_lib.run(m1['forcing'], m1['doy'], m1['member_params'][:, :3], m1['reach_params'][:, :, :3],
                                   m1['up_ptr'], m1['up_idx'], m1['opts'])
        assert helpers.max_rel_err(a.cpu().numpy()[..., :3], ref, floor=1e-12) < 1e-10


@pytest.mark.gpu
def test_snow_state_crosses_time_chunks(engine0):
    """Task-queue kernel: the snow depth is part of the state handed over between 256-day chunks."""
    import torch
    E = 200
    m1, _ = setup('tarland_1981_2010_dynamic', E, solver=dict(rtol=1e-6, atol=1e-8, balance=0, time_chunk_days=-1), out_mask=marshal.MASK_REACH5)
    D = 1500
    m1['forcing'] = np.ascontiguousarray(m1['forcing'][:, :, :D]); m1['doy'] = np.ascontiguousarray(m1['doy'][:D])
    ref, _, st0 = engine0.run(m1['forcing'], m1['doy'], m1['member_params'], m1['reach_params'], m1['up_ptr'], m1['up_idx'], m1['opts'])
    m1['opts'].time_chunk_days, m1['opts'].balance, m1['opts'].balance_pilot_days = 256, 1, 100
    got, _, st = engine0.run(m1['forcing'], m1['doy'], m1['member_params'], m1['reach_params'], m1['up_ptr'], m1['up_idx'], m1['opts'])
    assert st0['queued'] == 0 and st['queued'] == 1 and st['balanced'] == 1
    assert bool(torch.equal(got, ref))


@pytest.mark.gpu
def test_run_simply_p_ensemble_perturbs_snow_parameters(engine0):
    met, p_struc, p_SU, p_LU, p_SC, p, dyn = helpers.scenario_inputs('tarland_2004_dynamic')
    f = np.array([2.74, 1.0, 5.0])
    args = lambda: (met.copy(), p_struc.copy(), p_SU.copy(), p_LU.copy(), p_SC.copy(), p.copy(), dyn.copy())
    res = sp.run_simply_p_ensemble(*args(), overrides={'f_DDSM': f})
    assert res['stats']['rhs_evals'] > 0
    base = sp.run_simply_p_ensemble(*args(), n_members=1)                       # met_df['P'] as the reference feeds it
    assert np.array_equal(res['data'][..., 0], base['data'][..., 0])
    for k in (1, 2):                                                            # = a run on that member's own host-made P
        mk = sp.snow_hydrol_inputs(p['D_snow_0'], f[k], met[['T_air', 'PET', 'Precipitation']].copy())
        a = args()
        one = sp.run_simply_p_ensemble(mk, *a[1:], n_members=1)
        assert np.array_equal(res['data'][..., k], one['data'][..., 0])
    with pytest.raises(ValueError, match='Precipitation'):
        sp.run_simply_p_ensemble(met[['P', 'PET']].copy(), *args()[1:], overrides={'f_DDSM': f})


# ---- the 26th output column: every member's snow depth at the end of the day (SIMPLYP_OUT_D_SNOW) -------------------------
# The reference returns met_df['D_snow_end'] (inputs.py:197-207) as df_TC['D_snow'] (model.py:775-776); with the snow module
# run per member inside the kernel the depth is a per-member series, so an ensemble that perturbs f_DDSM gets it back.

def host_snow_depth(met, f_ddsm, d0):
    """[D][E] D_snow_end of every member from the host snow function (bit-equal to the reference's column, tests/test_host.py)."""
    return np.stack([sp.snow_hydrol_inputs(d, f, met[['T_air', 'PET', 'Precipitation']].copy())['D_snow_end'].to_numpy()
                     for f, d in zip(f_ddsm, d0)], axis=1)


def test_oracle_snow_depth_column_equals_host_snow_function(oracle_lib):
    E = 5
    m1, _ = setup('tarland_2004_dynamic', E, solver=dict(integrator='rk4', substeps=8), out_mask=marshal.mask_of_columns(['Qr', 'D_snow']))
    out, st, _ = oracle_lib.run(m1['forcing'], m1['doy'], m1['member_params'], m1['reach_params'], m1['up_ptr'], m1['up_idx'], m1['opts'])
    f_ddsm, d0 = snow_members(E)
    assert out.shape[0] == 2 and np.array_equal(out[1, :, 0, :], host_snow_depth(m1['met'], f_ddsm, d0))
    assert out[1].max() > 1.0                                   # there was snow
    m1['opts'].snow = 0
    with pytest.raises(ValueError, match='D_snow'):
        oracle_lib.run(m1['forcing'][:, :2], m1['doy'], m1['member_params'], m1['reach_params'], m1['up_ptr'], m1['up_idx'], m1['opts'])


@pytest.mark.gpu
@pytest.mark.parametrize('name,E,solver', [('tarland_2004_dynamic', 130, None),                       # quad kernel (auto), chain
                                           ('tarland_2004_dynamic', 130, dict(lanes_per_member=1)),
                                           ('tarland_1981_2010_dynamic', 300, dict(time_chunk_days=256, balance=1, out_slot_order=0)),   # queue: depth crosses chunks
                                           ('confluence3_nc_2004', 70, None)])                         # every reach reports its member's depth
def test_kernel_snow_depth_column(engine0, name, E, solver):
    import torch
    m1, _ = setup(name, E, solver=solver, out_mask=marshal.MASK_REACH5 | marshal.MASK_D_SNOW)
    out, st, stats = engine0.run(m1['forcing'], m1['doy'], m1['member_params'], m1['reach_params'], m1['up_ptr'], m1['up_idx'], m1['opts'])
    f_ddsm, d0 = snow_members(E)
    want = host_snow_depth(m1['met'], f_ddsm, d0)
    got = out[5].cpu().numpy()                                  # [D][n_reaches][E]
    assert int(st.max()) == 0
    for r in range(got.shape[1]):
        assert np.array_equal(got[:, r, :], want)
    # the other columns are what a run without the extra column writes
    m1['opts'].out_mask = marshal.MASK_REACH5
    ref, _, _ = engine0.run(m1['forcing'], m1['doy'], m1['member_params'], m1['reach_params'], m1['up_ptr'], m1['up_idx'], m1['opts'])
    assert bool(torch.equal(out[:5], ref))


@pytest.mark.gpu
def test_snow_depth_column_needs_the_snow_module(engine0):
    from simplyp_amd import engine
    m0 = helpers.marshal_scenario('tarland_2004_dynamic', E=4, out_mask=marshal.MASK_REACH5 | marshal.MASK_D_SNOW)
    with pytest.raises(engine.EngineError, match='opts.snow'):
        engine0.run(m0['forcing'], m0['doy'], m0['member_params'], m0['reach_params'], m0['up_ptr'], m0['up_idx'], m0['opts'])
    met, p_struc, p_SU, p_LU, p_SC, p, dyn = helpers.scenario_inputs('tarland_2004_dynamic')
    f = np.array([2.74, 1.0, 5.0])
    res = sp.run_simply_p_ensemble(met.copy(), p_struc, p_SU, p_LU, p_SC, p, dyn, overrides={'f_DDSM': f}, outputs=['Qr', 'D_snow'])
    assert res['columns'] == ['Qr', 'D_snow']
    assert np.array_equal(res['data'][1, :, 0, :], host_snow_depth(met, f, np.full(3, float(p['D_snow_0']))))
    assert np.array_equal(res['data'][1, :, 0, 0], met['D_snow_end'].to_numpy())       # member 0 = the workbook = the reference's column
    with pytest.raises(ValueError, match='snow_in_kernel'):
        sp.run_simply_p_ensemble(met.copy(), p_struc, p_SU, p_LU, p_SC, p, dyn, n_members=2, outputs=['Qr', 'D_snow'])


@pytest.mark.gpu
def test_snow_depth_column_together_with_goodness_of_fit_and_waterbody(engine0):
    """the 26th column's bit is outside SIMPLYP_MASK_ALL; simplyp_gof / _spearman / _waterbody used to refuse a table
    that carries it (after the whole ensemble had run).  It is the highest bit, so the offsets of Qr and the fluxes do not move:
    statistics and sums equal those of a run without the column."""
    import pandas as pd
    met, p_struc, p_SU, p_LU, p_SC, p, dyn = helpers.scenario_inputs('confluence3_nc_2004')
    f = np.array([2.0, 2.74, 3.5, 1.2])
    rng = np.random.default_rng(2)
    obs = {3: pd.DataFrame({'Q': rng.uniform(0.1, 2.0, len(met)), 'PP': rng.uniform(0.01, 0.1, len(met))}, index=met.index)}
    kw = dict(overrides={'f_DDSM': f}, obs_dict=obs, waterbody=[1, 2], spearman=True)
    with_snow = sp.run_simply_p_ensemble(met.copy(), p_struc, p_SU, p_LU.copy(), p_SC.copy(), p, dyn, outputs=['Qr', 'D_snow'], **kw)
    without = sp.run_simply_p_ensemble(met.copy(), p_struc, p_SU, p_LU.copy(), p_SC.copy(), p, dyn, outputs=['Qr'], **kw)
    assert with_snow['columns'][-1] == 'D_snow' and 'D_snow' not in without['columns']
    assert np.array_equal(with_snow['gof']['data'], without['gof']['data'], equal_nan=True)
    assert np.array_equal(with_snow['gof']['spearman'], without['gof']['spearman'], equal_nan=True)
    assert np.array_equal(with_snow['waterbody']['data'], without['waterbody']['data'], equal_nan=True)
