"""Host-side logic (no GPU): stdlib xlsx reader, read_input_data, snow module, prologue side effects,
topology parsing, SoA marshalling, routing schedule, unit helpers."""

import os
import shutil

import numpy as np
import pandas as pd
import pytest

import helpers
import simplyp_amd as sp
from simplyp_amd import marshal, xlsx, engine, abi

REF_DATA = helpers.DATA
WORKBOOK = os.path.join(REF_DATA, 'Parameters_v0-2A_Tarland.xlsx')


@pytest.fixture()
def tarland_tree(tmp_path):
    """The reference's directory layout, so the workbook's relative Windows paths resolve."""
    rel = tmp_path / 'Current_Release' / 'v0-2A'
    dat = tmp_path / 'Example_Data' / 'Tarland_Scotland'
    obs = dat / 'Observations'
    for d in (rel, obs):
        d.mkdir(parents=True)
    shutil.copy(WORKBOOK, rel / 'Parameters_v0-2A_Tarland.xlsx')
    shutil.copy(os.path.join(REF_DATA, 'Tarland_MetData_1981-2010.csv'), dat)
    for f in ('Coull_DailyMeanQ.xlsx', 'Coull_ChemObs.xlsx'):
        shutil.copy(os.path.join(REF_DATA, f), obs)
    return str(rel / 'Parameters_v0-2A_Tarland.xlsx')


def test_xlsx_reader_sheets_and_values():
    wb = xlsx.Workbook(WORKBOOK)
    assert wb.sheet_names == ['Readme', 'Setup', 'Reach_structure', 'LU', 'SC_reach', 'Constant', 'Preprocessing']
    p = xlsx.read_excel(wb, 'Constant', index_col=0, usecols="B,E")['Value']
    assert p['fc'] == 290 and p['T_g'] == 65 and p['Qg_min'] == 0.4 and p['b_Q'] == 0.42
    assert p['Kf'] == pytest.approx(1.131528046e-4, rel=1e-12)
    lu = xlsx.read_excel(wb, 'LU', index_col=0, usecols="B,E,F,G,H")
    assert list(lu.columns) == ['A', 'S', 'IG', 'NC']
    assert lu.loc['T_s', 'A'] == 2 and lu.loc['T_s', 'S'] == 10 and np.isnan(lu.loc['T_s', 'IG'])
    assert lu.loc['C_cover', 'IG'] == 0.09 and lu.loc['P_netInput', 'NC'] == 10
    sc = xlsx.read_excel(wb, 'SC_reach', index_col=0, usecols="B,E")
    assert list(sc.columns) == [1] and sc.loc['A_catch', 1] == 51.7 and sc.loc['L_reach', 1] == 10000
    su = xlsx.read_excel(wb, 'Setup', index_col=0, usecols="A,C")['Value']
    assert su['run_mode'] == 'cal' and su['n_SC'] == 1 and su['st_dt'] == '2004-01-01'


def test_read_input_data_matches_reference_shapes(tarland_tree, capsys):
    p_SU, dyn, p, p_LU, p_SC, p_struc, met_df, obs = sp.read_input_data(tarland_tree)
    printed = capsys.readouterr().out
    for line in ('Parameter values successfully read in', 'Input meteorological data read in',
                 'Snow accumulation and melt module run to estimate snowmelt inputs to the soil',
                 'Observed discharge data read in', 'Observed water chemistry data read in'):
        assert line in printed
    assert list(dyn.index) == ['Dynamic_EPC0', 'Dynamic_effluent_inputs', 'Dynamic_terrestrialP_inputs',
                               'Dynamic_erodibility']
    assert list(p['SC_list']) == [1]
    assert len(met_df) == 366 and met_df.index[0] == pd.Timestamp('2004-01-01')
    assert list(met_df.columns) == ['T_air', 'PET', 'Precipitation', 'P_snow', 'P_rain', 'P_melt',
                                    'D_snow_start', 'D_snow_end', 'P']
    assert p_struc.shape == (1, 2) and np.isnan(p_struc.loc[1, 'Upstream_SCs'])
    assert set(obs.keys()) == {1} and 'Q' in obs[1].columns and 'TP' in obs[1].columns
    assert obs[1].index.min() >= pd.Timestamp('2004-01-01') and obs[1].index.max() <= pd.Timestamp('2004-12-31')
    # same inputs as the golden scenario (which the unmodified reference ran on)
    met_g, p_struc_g, p_SU_g, p_LU_g, p_SC_g, p_g, _ = helpers.scenario_inputs('tarland_2004_static')
    # (the fixture's JSON keeps the rows in sorted order)
    pd.testing.assert_frame_equal(p_LU.sort_index(), p_LU_g.sort_index(), check_names=False)
    pd.testing.assert_frame_equal(p_SC.sort_index(), p_SC_g.sort_index(), check_names=False)
    assert list(p_LU.index) == ['T_s', 'SoilPconc', 'P_netInput', 'EPC0_init_mgl', 'C_cover', 'C_measures']
    for k in p_g.index:
        if k != 'SC_list':
            assert p[k] == p_g[k], k


def test_read_input_data_reference_errors(tmp_path, tarland_tree):
    with pytest.raises(FileNotFoundError):
        sp.read_input_data(str(tmp_path / 'nope.xlsx'))


@pytest.mark.parametrize('name', ['tarland_2004_static', 'tarland_1981_2010_dynamic'])
def test_snow_module_matches_reference(name):
    """snow_hydrol_inputs vs the columns the reference's own snow module produced (golden 'met')."""
    met, *_ = helpers.scenario_inputs(name)
    mine = sp.snow_hydrol_inputs(0.0, 2.74, met[['T_air', 'PET', 'Precipitation']].copy())
    assert list(mine.columns) == list(met.columns)
    for c in ['P_snow', 'P_rain', 'P_melt', 'D_snow_start', 'D_snow_end', 'P']:
        np.testing.assert_allclose(mine[c].values, met[c].values, rtol=0, atol=0, err_msg=c)


def test_snow_initial_depth_and_melt_limit():
    idx = pd.date_range('2000-01-01', periods=4)
    met = pd.DataFrame({'T_air': [-1.0, 5.0, 5.0, 5.0], 'PET': 0.0, 'Precipitation': [4.0, 1.0, 0.0, 2.0]}, index=idx)
    out = sp.snow_hydrol_inputs(3.0, 1.0, met)
    np.testing.assert_allclose(out['D_snow_end'].values, [7.0, 2.0, 0.0, 0.0])
    np.testing.assert_allclose(out['P_melt'].values, [0.0, 5.0, 2.0, 0.0])
    np.testing.assert_allclose(out['P'].values, [0.0, 6.0, 2.0, 2.0])


def test_daily_pet_is_out_of_scope():
    with pytest.raises(NotImplementedError):
        sp.daily_PET(57.0, pd.DataFrame({'T_air': [1.0]}, index=pd.date_range('2000-01-01', periods=1)))


def test_prologue_side_effects_and_validation():
    """model.py:311-361: rows added in place, NC types, ValueError / AssertionError cases."""
    met, p_struc, p_SU, p_LU, p_SC, p, dyn = helpers.scenario_inputs('confluence3_nc_2004')
    nc = marshal.prologue(p_SU, p_LU, p_SC, p)
    assert nc == {1: 'None', 2: 'A', 3: 'S'}
    assert list(p_SC.loc['NC_type']) == ['None', 'A', 'S']
    for row in ('EPC0_0', 'Plab0', 'TDPs0'):
        assert row in p_LU.index and p_LU.loc[row].isna().all()
    assert p_SC.loc['f_A', 2] == 0.375 + 0.125
    assert p_SC.loc['f_NC_A', 2] == 0.125 * 0.2
    marshal.epilogue_mutations(p_SU, p_LU, p_SC, p)
    after = helpers.meta()['confluence3_nc_2004']
    for col, rows in after['p_LU_after'].items():
        for k, v in rows.items():
            got = p_LU.loc[k, col]
            assert (np.isnan(got) if v is None else got == pytest.approx(v, rel=1e-15)), (k, col)
    for col, rows in after['p_SC_after'].items():
        for k, v in rows.items():
            got = p_SC.loc[k, int(col)]
            if isinstance(v, str):
                assert got == v
            else:
                assert got == pytest.approx(v, rel=1e-15), (k, col)

    met, p_struc, p_SU, p_LU, p_SC, p, dyn = helpers.scenario_inputs('tarland_2004_static')
    p_SC.loc['f_S', 1] = 0.4
    with pytest.raises(ValueError, match='Land use proportions do not add to 1 in SC 1'):
        marshal.prologue(p_SU, p_LU, p_SC, p)
    met, p_struc, p_SU, p_LU, p_SC, p, dyn = helpers.scenario_inputs('tarland_2004_static')
    p_SC.loc['f_NC_Ar', 1] = 0.1
    p_SC.loc['f_NC_S', 1] = 0.1
    with pytest.raises(ValueError, match='2 kinds of newly-converted land'):
        marshal.prologue(p_SU, p_LU, p_SC, p)
    met, p_struc, p_SU, p_LU, p_SC, p, dyn = helpers.scenario_inputs('tarland_2004_static')
    p['d_maxE_spr'] = 20
    with pytest.raises(AssertionError, match="'d_maxE_spr' must be between 30 and 335"):
        marshal.prologue(p_SU, p_LU, p_SC, p)


def test_topology_parsing_rules():
    """Upstream cell may be blank, an int, or a string list (model.py:480-487)."""
    p = pd.Series({'SC_list': np.arange(1, 5)}, dtype=object)
    struc = pd.DataFrame({'Upstream_SCs': pd.Series([np.nan, 1, '1, 2', ' 3 '], index=[1, 2, 3, 4], dtype=object)})
    up_ptr, up_idx, lists = marshal.topology(struc, p)
    assert list(up_ptr) == [0, 0, 1, 3, 4] and list(up_idx) == [0, 0, 1, 2]
    assert lists == {1: [], 2: [1], 3: [1, 2], 4: [3]}
    struc.loc[2, 'Upstream_SCs'] = 3          # upstream id not yet simulated: KeyError in the reference
    with pytest.raises(KeyError):
        marshal.topology(struc, p)


def test_marshal_shapes_and_overrides():
    met, p_struc, p_SU, p_LU, p_SC, p, dyn = helpers.scenario_inputs('chain4_val_2004')
    marshal.prologue(p_SU, p_LU, p_SC, p)
    mp = marshal.member_params(p, p_LU, 5, {'fc': np.arange(5) + 280.0, 'T_s_A': 3.0})
    assert mp.shape == (marshal.NP_M, 5) and abi.Opts  # noqa
    assert list(mp[marshal.PM_NAMES.index('fc')]) == [280, 281, 282, 283, 284]
    assert (mp[marshal.PM_NAMES.index('T_s_A')] == 3.0).all()
    assert (mp[marshal.PM_NAMES.index('k_M')] == 1.7).all()
    rp = marshal.reach_params(p_SC, p, 5, {'L_reach': np.array([[1.], [2.], [3.], [4.]]) * 1000})
    assert rp.shape == (marshal.NP_R, 4, 5)
    assert (rp[marshal.PR_NAMES.index('L_reach'), 2] == 3000).all()
    assert (rp[marshal.PR_NAMES.index('A_catch'), :, 0] == [8.5, 12.25, 20.0, 51.7]).all()
    f, doy = marshal.forcing_arrays(met)
    assert f.shape == (1, 2, 366) and doy[0] == 1 and doy[-1] == 366 and doy.dtype == np.int32
    with pytest.raises(KeyError):
        marshal.split_member_reach_overrides({'not_a_param': 1.0})
    assert marshal.columns_of_mask(marshal.MASK_REACH5) == marshal.REACH5_COLUMNS
    assert marshal.mask_of_columns(marshal.OUT_COLUMNS) == marshal.MASK_ALL


def _check_plan(up_ptr, up_idx):
    """Schedule invariants: upstream before downstream; a routing slot is never rewritten while a reader
    that is not the writer's own chain successor may still need it."""
    S = len(up_ptr) - 1
    pl = engine.plan(up_ptr, up_idx)
    order = sorted(range(S), key=lambda s: (pl['launch'][s], pl['chain'][s], pl['pos'][s]))
    key = {s: (pl['launch'][s], pl['chain'][s], pl['pos'][s]) for s in range(S)}
    down = {s: [] for s in range(S)}
    for s in range(S):
        for u in up_idx[up_ptr[s]:up_ptr[s + 1]]:
            down[int(u)].append(s)
            lu, ls = pl['launch'][u], pl['launch'][s]
            assert lu < ls or (lu == ls and pl['chain'][u] == pl['chain'][s] and pl['pos'][u] == pl['pos'][s] - 1), \
                "reach %d must be finished (or be the chain predecessor) before %d" % (u, s)
    for s in range(S):
        assert (pl['route_slot'][s] >= 0) == bool(down[s])
    # slot lifetimes: writer w holds slot until its last reader; any other writer of the same slot whose
    # (launch) lies within that window must be in the same chain and after w's reader
    for w in range(S):
        slot = pl['route_slot'][w]
        if slot < 0:
            continue
        last_reader = max(down[w], key=lambda s: key[s])
        for x in range(S):
            if x == w or pl['route_slot'][x] != slot:
                continue
            if key[x] < key[w]:
                continue
            # x writes the slot after w: allowed only once every reader of w is done
            if pl['launch'][x] > pl['launch'][last_reader]:
                continue
            assert pl['launch'][x] == pl['launch'][last_reader] and pl['chain'][x] == pl['chain'][last_reader] \
                and pl['pos'][x] > pl['pos'][last_reader] and len(down[w]) == 1, (w, x, pl)
    return pl


def test_plan_single_chain_confluence_and_random_dags():
    pl = _check_plan(np.array([0, 0]), np.array([], dtype=np.int32))
    assert pl['n_launches'] == 1 and pl['n_slots'] == 0
    S = 256                                   # BASELINE config C4: one in-kernel chain, two alternating slots
    pl = _check_plan(np.r_[0, np.arange(0, S)], np.arange(S - 1))
    assert pl['n_launches'] == 1 and pl['n_slots'] == 2 and list(pl['pos']) == list(range(S))
    pl = _check_plan(np.array([0, 0, 0, 2]), np.array([0, 1]))
    assert pl['n_launches'] == 2 and list(pl['launch']) == [0, 0, 1]
    rng = np.random.default_rng(5)
    for trial in range(40):
        S = int(rng.integers(2, 30))
        ups = [[]]
        for s in range(1, S):
            k = int(rng.integers(0, 3))
            ups.append(sorted(set(int(u) for u in rng.integers(0, s, size=k))))
        up_ptr = np.r_[0, np.cumsum([len(u) for u in ups])]
        up_idx = np.array([u for us in ups for u in us], dtype=np.int32)
        _check_plan(up_ptr, up_idx)


def test_plan_rejects_bad_topology():
    with pytest.raises(engine.EngineError):
        engine.plan(np.array([0, 1, 1]), np.array([1]))      # reach 0 lists reach 1 as upstream


def test_unit_conversion_helpers():
    assert sp.UC_Q(2.0, 10.0) == 20000.0
    assert sp.UC_Qinv(1.0, 51.7) == 1.0 * 86400 / (1000 * 51.7)
    assert sp.UC_C(5.0, 2.0) == 2.5 and sp.UC_Cinv(2.5, 2.0) == 5.0
    assert sp.UC_V(1.0, 2.0, 'm3') == 2000.0 and sp.UC_V(1.0, 2.0, 'l') == 2000000.0
    assert sp.lin_interp(45, 30.0, 60.0, 0.2, 1.0) == 0.2 + 0.8 * 15 / 30.0


def test_derived_species_and_sum_to_waterbody(capsys):
    idx = pd.date_range('2004-01-01', periods=3)
    def reach(k):
        df = pd.DataFrame({'Q_cumecs': [1.0, 2.0, 3.0], 'Msus_kg/day': [10.0, 20.0, 30.0],
                           'TDP_kg/day': [1.0, 1.0, 1.0], 'PP_kg/day': [2.0, 2.0, 2.0]}, index=idx) * k
        df['TDP_mgl'] = 0.1 * k
        df['PP_mgl'] = 0.2 * k
        return sp.derived_P_species(df, 0.7)
    R = {1: reach(1), 2: reach(2), 3: reach(3)}
    np.testing.assert_allclose(R[1]['TP_mgl'].values, 0.3, rtol=1e-15)
    np.testing.assert_allclose(R[2]['SRP_kg/day'].values, 1.4, rtol=1e-15)
    struc = pd.DataFrame({'Upstream_SCs': [np.nan] * 3, 'In_final_flux?': [1, 0, 1]}, index=[1, 2, 3])
    tot = sp.sum_to_waterbody(struc, 3, R, 0.7)
    assert list(tot['Q_cumecs']) == [4.0, 8.0, 12.0]
    assert tot['SS_mgl'].iloc[0] == pytest.approx((40.0 / 4.0) * 1000. / 86400.)
    assert list(tot['TP_kg/day']) == [12.0, 12.0, 12.0]
    struc['In_final_flux?'] = [1, 0, 0]
    assert sp.sum_to_waterbody(struc, 3, R, 0.7) is None
    assert 'One or fewer reaches were selected' in capsys.readouterr().out
    struc['In_final_flux?'] = [1, 1, 1]
    with pytest.raises(ValueError, match="Mismatch between the number of subcatchments"):
        sp.sum_to_waterbody(struc, 2, R, 0.7)


def test_ensemble_overrides_go_through_the_reference_input_checks():
    """Per-member overrides must not slip past the reference's validation (model.py:321-335, :355-357): land-use
    fractions that do not add to 1 exactly, both kinds of newly-converted land, erosion-window days outside (30, 335)."""
    from simplyp_amd import marshal
    import helpers
    met, p_struc, p_SU, p_LU, p_SC, p, dyn = helpers.scenario_inputs('confluence3_nc_2004')
    marshal.prologue(p_SU, p_LU, p_SC, p)
    scs = marshal.sc_list(p)
    E = 5
    mp = marshal.member_params(p, p_LU, E)
    rp = marshal.reach_params(p_SC, p, E)
    marshal.validate_ensemble(mp, rp, scs)                      # the workbook values pass
    bad = rp.copy(); bad[marshal.PR_NAMES.index('f_S'), 1, 3] += 0.01
    with pytest.raises(ValueError, match=r'do not add to 1 in SC 2 \(ensemble member 3\)'):
        marshal.validate_ensemble(mp, bad, scs)
    bad = rp.copy(); bad[marshal.PR_NAMES.index('f_NC_S'), 1, 4] = 0.1       # SC 2 already has f_NC_Ar > 0
    with pytest.raises(ValueError, match='2 kinds of newly-converted land'):
        marshal.validate_ensemble(mp, bad, scs)
    badm = mp.copy(); badm[marshal.PM_NAMES.index('d_maxE_aut'), 2] = 340.0
    with pytest.raises(AssertionError, match="'d_maxE_aut' must be between 30 and 335"):
        marshal.validate_ensemble(badm, rp, scs)


def test_bench_parent_launches_its_own_ranks_without_touching_the_gpu(monkeypatch):
    """`python bench.py --gpus N` from a plain shell: the parent builds a torch.distributed.run command for N
    ranks on 127.0.0.1 and relays its exit code; nothing GPU-related is imported before that.  Also: the config table is
    consistent and the kernel-source hash is stable."""
    import importlib
    import subprocess
    import sys
    bench = importlib.import_module('bench')
    seen = {}

    def fake_call(cmd, env=None):
        seen['cmd'], seen['env'] = cmd, env
        return 7
    monkeypatch.setattr(subprocess, 'call', fake_call)
    monkeypatch.setattr(sys, 'argv', ['bench.py', '--gpus', '4', '--steps', '2', '--scaling', 'strong'])
    monkeypatch.delenv('WORLD_SIZE', raising=False)
    with pytest.raises(SystemExit) as ex:
        bench.main()
    assert ex.value.code == 7
    cmd = seen['cmd']
    assert cmd[1:3] == ['-m', 'torch.distributed.run'] and '--nproc-per-node' in cmd and cmd[cmd.index('--nproc-per-node') + 1] == '4'
    assert cmd[cmd.index('--master-addr') + 1] == '127.0.0.1' and cmd[-6:] == ['--gpus', '4', '--steps', '2', '--scaling', 'strong']
    assert seen['env'].get('HSA_ENABLE_IPC_MODE_LEGACY') == '0'
    assert set(bench.CONFIGS) == {'c2', 'c3', 'c4', 'c5', 'strong_1m'} and bench.CONFIGS['strong_1m']['scaling'] == 'strong' and bench.CONFIGS['c2']['bytes_per_cd'] == 216.0
    assert bench.CONFIGS['c3']['bytes_per_cd'] == 56.0 and not bench.CONFIGS['c5']['parity_grade']
    assert bench.kernel_source_hash() == bench.kernel_source_hash() and len(bench.kernel_source_hash()) == 16


def test_second_pair_coefficients_are_what_the_derivation_gives():
    """include/simplyp_controller.h SIMPLYP_STIFF_*: the stability-optimised 4(3) pair.  The header's numbers satisfy the 8 + 4 order
    conditions to rounding, are stable on the real axis beyond the cap the controller uses, keep every internal stage polynomial
    bounded there -- and are exactly what tools/derive_stiff_pair.py prints for the recorded search (deterministic, ~5 s)."""
    import re
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    with open(os.path.join(root, 'include', 'simplyp_controller.h')) as fh:
        txt = fh.read()
    macros = {k: float(v) for k, v in re.findall(r'#define SIMPLYP_STIFF_([A-Z0-9_]+) \(?(-?[0-9.e+-]+)\)?', txt)}
    sys.path.insert(0, os.path.join(root, 'tools'))
    import derive_stiff_pair as d
    x = np.zeros(27)
    for k, nm in enumerate(d.NAMES_A):
        x[k] = macros[nm]
    b = np.zeros(6); e = np.zeros(6)
    for k in (0, 2, 3, 5):
        b[k] = macros['B%d' % (k + 1)]
    for k in (0, 2, 3, 4, 5):
        e[k] = macros['E%d' % (k + 1)]
    x[15:21], x[21:27] = b, b - e
    A, bb, bh = d.unpack(x)
    assert np.abs(d.order4(A, bb)).max() < 1e-15 and np.abs(d.order4(A, bh)[:4]).max() < 1e-15
    pr = d.properties(x)
    assert pr['beta'] > 9.0 > macros['CAP'] > macros['Z_ON'] > 3.0 and pr['max_stage_poly'] < 1.51 and pr['max_Rhat'] <= 1.0 + 1e-9
    assert macros['Z_ON'] < 3.73                       # Cash-Karp's own real stability interval: the switch happens inside it
    out = subprocess.run([sys.executable, os.path.join(root, 'tools', 'derive_stiff_pair.py'), '--emit'], capture_output=True, text=True, timeout=300)
    assert out.returncode == 0, out.stderr[-500:]
    derived = {k: float(v) for k, v in re.findall(r'#define SIMPLYP_STIFF_([A-Z0-9_]+) (-?[0-9.e+-]+)', out.stdout)}
    assert set(derived) == set(macros) - {'Z_ON', 'CAP', 'ERR_EXP', 'Z_START'}
    assert 0.0 < macros['Z_START'] < 1.0
    for k, v in derived.items():
        assert abs(v - macros[k]) <= 1e-15 * max(1.0, abs(v)), k
