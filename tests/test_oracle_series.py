"""The CPU oracle's full driver against the reference's own results (tests/golden/*.npz, produced by
make_golden.py from the unmodified reference).

Primary gate: reference run_simply_p with odeint(rtol=atol=1e-12) -- the converged solution of the
reference's equations ("tight").  Informational bound: the reference as shipped (LSODA rtol=0.01), which
is itself only ~1e-3..1e-1 accurate (hard part 1) and the two example CSVs the
reference ships.
"""

import os

import numpy as np
import pandas as pd
import pytest

import helpers
from simplyp_amd import marshal

SCENARIOS = ['tarland_2004_static', 'tarland_2004_dynamic', 'confluence3_nc_2004', 'chain4_val_2004', 'stiff_chain12_2004']
REACH_COLS = ['Vr', 'Qr_EndOfDay', 'Qr', 'Msus_EndOfDay', 'Msus_kg/day', 'TDPr_EndOfDay', 'TDP_kg/day',
              'PPr_EndOfDay', 'PP_kg/day']


def column_errors(out, scs, gold, member=0):
    errs = {}
    for j, sc in enumerate(scs):
        R, TC = gold['R'][sc], gold['TC'][sc]
        for ci, c in enumerate(marshal.OUT_COLUMNS):
            ref = R[c].values if c in R.columns else TC[c].values
            errs[c] = max(errs.get(c, 0.0), helpers.max_rel_err(out[ci, :, j, member], ref, floor=1e-300))
    return errs


@pytest.mark.parametrize('name', SCENARIOS)
@pytest.mark.parametrize('integrator', ['cashkarp', 'cashkarp_aug'])
def test_oracle_converged_matches_reference_tight(oracle_lib, name, integrator):
    """Cash-Karp at rtol=1e-9: every one of the 25 raw columns within 1e-6 of the tight reference, the 9
    reach columns within 1e-7."""
    m = helpers.marshal_scenario(name, solver=dict(integrator=integrator, rtol=1e-9, atol=1e-11))
    out, status, stats = oracle_lib.run(m['forcing'], m['doy'], m['member_params'], m['reach_params'],
                                        m['up_ptr'], m['up_idx'], m['opts'])
    assert status.max() == 0
    errs = column_errors(out, m['scs'], helpers.golden_tables(name, 'tight'))
    assert max(errs[c] for c in REACH_COLS) < 1e-7, errs
    assert max(errs.values()) < 1e-6, errs


@pytest.mark.parametrize('name', SCENARIOS)
def test_oracle_default_solver_meets_parity_bar(oracle_lib, name):
    """The default solver settings (abi.DEFAULT_SOLVER) give <= 1e-6 on all reach outputs."""
    m = helpers.marshal_scenario(name)
    out, status, _ = oracle_lib.run(m['forcing'], m['doy'], m['member_params'], m['reach_params'],
                                    m['up_ptr'], m['up_idx'], m['opts'])
    errs = column_errors(out, m['scs'], helpers.golden_tables(name, 'tight'))
    assert max(errs[c] for c in REACH_COLS) < 1e-6, errs


def test_oracle_rk4_literal_matches_reference_tight(oracle_lib):
    """Fixed-step RK4, 256 substeps, Vr integrated literally (no drift control)."""
    name = 'tarland_2004_dynamic'
    m = helpers.marshal_scenario(name, solver=dict(integrator='rk4', substeps=256, project_vr=0))
    out, _, _ = oracle_lib.run(m['forcing'], m['doy'], m['member_params'], m['reach_params'],
                               m['up_ptr'], m['up_idx'], m['opts'])
    errs = column_errors(out, m['scs'], helpers.golden_tables(name, 'tight'))
    assert max(errs.values()) < 1e-7, errs


def test_oracle_30yr_default_solver(oracle_lib):
    """BASELINE config C2/C3's series (Tarland 1981-2010, 10 957 days): default solver <= 1e-6 on all reach
    outputs against the tight reference; without the Vr drift control the same tolerance drifts past it."""
    name = 'tarland_1981_2010_dynamic'
    gold = helpers.golden_tables(name, 'tight')
    m = helpers.marshal_scenario(name)
    out, status, stats = oracle_lib.run(m['forcing'], m['doy'], m['member_params'], m['reach_params'],
                                        m['up_ptr'], m['up_idx'], m['opts'])
    errs = column_errors(out, m['scs'], gold)
    assert status.max() == 0
    assert max(errs[c] for c in REACH_COLS) < 2e-7, errs
    # the literal 12-variable system with the same pair (its controller does not know about the knees of the gates, so it
    # needs rtol 1e-8 where the default scheme works at 1e-7): same accuracy class
    m = helpers.marshal_scenario(name, solver=dict(integrator='cashkarp', rtol=1e-8))
    out, status, stats = oracle_lib.run(m['forcing'], m['doy'], m['member_params'], m['reach_params'],
                                        m['up_ptr'], m['up_idx'], m['opts'])
    assert max(column_errors(out, m['scs'], gold)[c] for c in REACH_COLS) < 2e-7
    m = helpers.marshal_scenario(name, solver=dict(integrator='cashkarp', rtol=1e-7, project_vr=0))
    out, _, _ = oracle_lib.run(m['forcing'], m['doy'], m['member_params'], m['reach_params'],
                               m['up_ptr'], m['up_idx'], m['opts'])
    errs0 = column_errors(out, m['scs'], gold)
    assert errs0['Vr'] > 1e-6


@pytest.mark.parametrize('name', SCENARIOS)
def test_oracle_vs_reference_as_shipped(oracle_lib, name):
    """Informational: distance to the reference at its own rtol=0.01.  Daily mean flow and the three daily
    fluxes of the as-shipped run are themselves only ~5e-3 accurate on a 1-year run."""
    m = helpers.marshal_scenario(name)
    out, _, _ = oracle_lib.run(m['forcing'], m['doy'], m['member_params'], m['reach_params'],
                               m['up_ptr'], m['up_idx'], m['opts'])
    errs = column_errors(out, m['scs'], helpers.golden_tables(name, 'shipped'))
    bar = 6e-2 if name == 'stiff_chain12_2004' else 3e-2      # (LSODA's rtol = 0.01 errors add up along twelve reaches: 3.6e-2 there)
    for c in ['Qr', 'Msus_kg/day', 'TDP_kg/day', 'PP_kg/day']:
        assert errs[c] < bar, (c, errs[c])


def test_oracle_vs_shipped_example_output_csv(oracle_lib):
    """The two result files the reference ships (Example_Data/Example_Output): reproduced to the ~3e-3 that
    LSODA at rtol=0.01 allows (the unmodified reference itself re-run here only gets 1e-3..3.4e-3)."""
    ref_dir = helpers.DATA
    csv_R = pd.read_csv(os.path.join(ref_dir, 'Instream_results_Reach1.csv'), index_col=0)
    csv_TC = pd.read_csv(os.path.join(ref_dir, 'Results_TC_SC1.csv'), index_col=0)
    m = helpers.marshal_scenario('tarland_2004_dynamic')
    out, _, _ = oracle_lib.run(m['forcing'], m['doy'], m['member_params'], m['reach_params'],
                               m['up_ptr'], m['up_idx'], m['opts'])
    col = {c: out[i, :, 0, 0] for i, c in enumerate(marshal.OUT_COLUMNS)}
    assert len(csv_R) == out.shape[1] == 366
    for c in ['Msus_kg/day', 'PP_kg/day', 'TDP_kg/day']:
        assert helpers.max_rel_err(col[c], csv_R[c].values) < 6e-3, c
    q = col['Qr'] * 51.7 * 1000 / 86400
    assert helpers.max_rel_err(q, csv_R['Q_cumecs'].values) < 3e-3
    for c, tol in [('VsA', 3e-4), ('VsS', 3e-4), ('Vg', 5e-3), ('P_labile_A_kg', 1e-6), ('TDPs_A_kg', 1e-3),
                   ('C_cover_A', 1e-12), ('Qq', 1e-12), ('EPC0_A_kgmm', 1e-6)]:
        assert helpers.max_rel_err(col[c], csv_TC[c].values) < tol, c


def test_oracle_members_are_independent_and_threads_agree(oracle_lib):
    """E = 5 replicated members, OpenMP on: every member bit-identical to the E = 1 result."""
    name = 'confluence3_nc_2004'
    m1 = helpers.marshal_scenario(name, E=1)
    m5 = helpers.marshal_scenario(name, E=5)
    o1, _, s1 = oracle_lib.run(m1['forcing'], m1['doy'], m1['member_params'], m1['reach_params'],
                               m1['up_ptr'], m1['up_idx'], m1['opts'])
    o5, _, s5 = oracle_lib.run(m5['forcing'], m5['doy'], m5['member_params'], m5['reach_params'],
                               m5['up_ptr'], m5['up_idx'], m5['opts'], n_threads=3)
    for k in range(5):
        assert np.array_equal(o5[..., k], o1[..., 0], equal_nan=True)
    assert s5['rhs_evals'] == 5 * s1['rhs_evals']


def test_oracle_flags_poisoned_member(oracle_lib):
    m = helpers.marshal_scenario('tarland_2004_static', E=3)
    m['member_params'][marshal.PM_NAMES.index('T_s_A'), 1] = np.nan
    out, status, _ = oracle_lib.run(m['forcing'], m['doy'], m['member_params'], m['reach_params'],
                                    m['up_ptr'], m['up_idx'], m['opts'])
    assert status[1] & 1 and status[0] == 0 and status[2] == 0
    assert np.isnan(out[marshal.OUT_COLUMNS.index('Qr'), -1, 0, 1])
    assert np.array_equal(out[..., 0], out[..., 2])


def test_oracle_default_solver_across_parameter_distribution(oracle_lib):
    """8 members of the bench's Monte-Carlo distribution (BASELINE config C3), 3 years each, against the
    unmodified reference at rtol=atol=1e-12 (tests/golden/monte_carlo_members.npz): <= 1e-6 on all reach outputs
    with the default solver -- the parity bar holds across the distribution, not only at the workbook's values."""
    import os
    from simplyp_amd import synthetic, abi
    z = np.load(os.path.join(helpers.GOLDEN, 'monte_carlo_members.npz'), allow_pickle=False)
    years = [str(y) for y in z['years']]
    names = [str(n) for n in z['names']]
    vals = z['values']
    n = vals.shape[1]
    met_df, p_struc, p_SU, p_LU, p_SC, p, dyn = synthetic.tarland_inputs(years[0], years[1], dynamic_epc0='y', dynamic_erod='n')
    over = synthetic.monte_carlo_overrides(p, p_LU, n)
    for k, nm in enumerate(names):                     # the generator still draws what the fixture recorded
        np.testing.assert_array_equal(over[nm], vals[k])
    marshal.prologue(p_SU, p_LU, p_SC, p)
    up_ptr, up_idx, _ = marshal.topology(p_struc, p)
    mp = marshal.member_params(p, p_LU, n, over)
    rp = marshal.reach_params(p_SC, p, n)
    forcing, doy = marshal.forcing_arrays(met_df)
    opts = abi.make_opts(dynamic_epc0=True, run_mode_cal=True)
    out, status, _ = oracle_lib.run(forcing, doy, mp, rp, up_ptr, up_idx, opts)
    assert status.max() == 0
    rcols = [str(c) for c in z['R/columns']]
    worst = 0.0
    for e in range(n):
        R = z['R/%d' % e]
        for c in REACH_COLS:
            worst = max(worst, helpers.max_rel_err(out[marshal.OUT_COLUMNS.index(c), :, 0, e], R[:, rcols.index(c)], floor=1e-300))
    assert worst < 5e-7, worst


@pytest.mark.parametrize('name', ['tarland_2004_dynamic', 'confluence3_nc_2004', 'chain4_val_2004'])
def test_fp32_stage_mirror_is_a_valid_integration_at_its_tolerance(oracle_lib, name):
    """oracle integrator 3 (the same-arithmetic mirror of the kernel's fp32-stage mode, BASELINE config C5) against the
    reference's converged tables: 5e-4 on the 9 reach columns at rtol 1e-5 -- the bar the GPU test holds the kernel to --
    and about as many right-hand-side evaluations (10 %) as the fp64 scheme at that tolerance.  (The fp32 mode keeps the
    plain step controller: the knee logic of the fp64 scheme holds the soil boxes to rtol * |Vs - fc| -- at rtol 1e-5 one
    float ulp of a 300 mm store, which fp32 stages cannot deliver -- and costs it a few percent more evaluations here.)"""
    gold = helpers.golden_tables(name, 'tight')
    res = {}
    for integ in ('cashkarp_aug_f32', 'cashkarp_aug'):
        m = helpers.marshal_scenario(name, E=1, solver=dict(integrator=integ, rtol=1e-5, atol=1e-7))
        out, status, stats = oracle_lib.run(m['forcing'], m['doy'], m['member_params'], m['reach_params'], m['up_ptr'], m['up_idx'], m['opts'])
        assert status.max() == 0
        res[integ] = (max(helpers.max_rel_err(out[marshal.OUT_COLUMNS.index(c), :, j, 0], gold['R'][sc][c].values, floor=1e-300)
                          for j, sc in enumerate(m['scs']) for c in REACH_COLS), stats['rhs_evals'])
    assert res['cashkarp_aug_f32'][0] < 5e-4 and res['cashkarp_aug'][0] < 5e-4, res
    assert abs(res['cashkarp_aug_f32'][1] - res['cashkarp_aug'][1]) < 0.10 * res['cashkarp_aug'][1], res


# Members of the bench's 100 000-member ensemble whose worst day, under a plain relative-tolerance controller, was a step
# across a knee of one of the reference's smooth-step gates (f_x, model.py:23-37: C1 only).  Found with tools/probe_tolerance.py
# and tools/probe_member.py; profiles/r02_experiments.md.  53752, 60773: the groundwater gate on a recession (5.3e-7 at rtol
# 1e-8, 1.9e-6 at 2e-8); 37627: a soil box starts to drain while Vg sits at its floor; 41834: a nearly dry reach is wetted
# (Qr**k_M grows 200-fold within the day); 71711, 27305: groundwater gate / sediment flux at low flow.
KNEE_MEMBERS = [53752, 60773, 37627, 41834, 71711, 27305]


def test_knee_aware_controller_on_the_members_that_needed_it(oracle_lib):
    """The default solver (knee-aware step controller, rtol 1e-7) on those members, 30 years, REACH-5 columns, against the same
    scheme at rtol 1e-11 (which test_oracle_converged_matches_reference_tight pins to the reference's tight tables): every one
    of them below 5e-7 -- and at a cost in right-hand sides that stays where the bench line reports it."""
    from simplyp_amd import synthetic
    pr = synthetic.c3_problem(100000)
    mp = np.ascontiguousarray(pr['member_params'][:, KNEE_MEMBERS])
    rp = np.ascontiguousarray(pr['reach_params'][:, :, KNEE_MEMBERS])
    D = pr['forcing'].shape[2]

    def run(rtol, atol):
        pr['opts'].rtol, pr['opts'].atol = rtol, atol
        out, status, stats = oracle_lib.run(pr['forcing'], pr['doy'], mp, rp, pr['up_ptr'], pr['up_idx'], pr['opts'], n_threads=6)
        assert status.max() == 0
        return out, stats

    default_rtol, default_atol = pr['opts'].rtol, pr['opts'].atol
    assert default_rtol == 1e-7
    truth, _ = run(1e-11, 1e-13)
    out, stats = run(default_rtol, default_atol)
    rel = np.abs(out - truth) / np.maximum(np.abs(truth), 1e-300)
    worst = rel.max(axis=(0, 1, 2))
    assert worst.max() < 5e-7, dict(zip(KNEE_MEMBERS, worst))
    # (these are low-flow members: fewer steps than the ensemble's 87 right-hand sides per catchment-day)
    assert 55 < stats['rhs_evals'] / (len(KNEE_MEMBERS) * D) < 95


@pytest.mark.parametrize('fname,bar', [('knee_members.npz', 5e-7), ('heldout_members.npz', 5e-7), ('wide_members.npz', 5e-7)])
def test_oracle_default_solver_against_reference_tables_of_single_members(oracle_lib, fname, bar):
    """The CPU mirror of the kernel's default solver against reference-made tables of single members (the unmodified reference
    at rtol=atol=1e-12, tests/golden/make_golden.py --only knee | heldout | wide): the six members the knee-aware controller was tuned
    on (30 years), 16 members of a held-out draw (3 years) and 24 held-out members of a draw with the time constants and rates widened
    x/÷ 2 (3 years; round 4: measured 1.5e-7); north_star's bar on all 9 reach outputs.  The same fixtures are run through the HIP kernel
    in tests/test_gpu_parity.py."""
    m, tables = helpers.member_fixture_problem(fname)
    out, status, _ = oracle_lib.run(m['forcing'], m['doy'], m['member_params'], m['reach_params'], m['up_ptr'], m['up_idx'], m['opts'],
                                    n_threads=8)
    assert status.max() == 0
    worst = [max(helpers.max_rel_err(out[marshal.OUT_COLUMNS.index(c), :, 0, k], tables[k][:, j], floor=1e-300)
                 for j, c in enumerate(REACH_COLS)) for k in range(len(tables))]
    assert max(worst) < bar, dict(zip(m['members'], worst))


def test_oracle_expansive_reach_on_a_drier_climate(oracle_lib):
    """The CPU mirror of the controller's expansive-reach rule (include/simplyp_controller.h: SIMPLYP_CTRL_EXPAND): the 24 members
    of a 100 000-member draw with the smallest Qg_min on a climate with 0.6 x Tarland's precipitation -- reaches that nearly dry up
    and are wetted again -- stay below 1e-6 against the converged solution (before the rule: up to 8.6e-6)."""
    from simplyp_amd import synthetic
    pr = synthetic.c3_problem(100000, seed=synthetic.C3_SEED + 12)
    pr['forcing'] = pr['forcing'].copy()
    pr['forcing'][:, 0] *= 0.6
    pr['forcing'][:, 1] /= 0.6
    sel = np.argsort(pr['member_params'][marshal.PM_NAMES.index('Qg_min')])[:24]
    mp = np.ascontiguousarray(pr['member_params'][:, sel]); rp = np.ascontiguousarray(pr['reach_params'][:, :, sel])

    def run(rtol, atol):
        pr['opts'].rtol, pr['opts'].atol = rtol, atol
        out, status, _ = oracle_lib.run(pr['forcing'], pr['doy'], mp, rp, pr['up_ptr'], pr['up_idx'], pr['opts'], n_threads=8)
        assert status.max() == 0
        return out
    rtol, atol = pr['opts'].rtol, pr['opts'].atol
    truth = run(1e-11, 1e-13)
    out = run(rtol, atol)
    worst = (np.abs(out - truth) / np.maximum(np.abs(truth), 1e-300)).max(axis=(0, 1, 2))
    assert worst.max() < 1e-6, dict(zip(sel.tolist(), worst))
    assert truth[1].min() < 1e-3


def test_oracle_default_solver_in_the_dry_reach_regime_against_the_reference(oracle_lib):
    """The regime the controller's expansive-reach rule (include/simplyp_controller.h: SIMPLYP_CTRL_EXPAND) was written for, pinned to the
    REFERENCE (round 3 had it kernel-vs-kernel only): ten members whose reach nearly dries up (Qg_min ~ 0, min Qr 1e-5 ... 1e-3 mm/d)
    and is wetted again, on a climate with 0.6 x Tarland's precipitation -- six of draw C3_SEED + 12, four of a draw with the time
    constants widened x/÷ 2 (the 7.2e-7 case of profiles/r03_tolerance) -- against tables the unmodified reference made with odeint at
    rtol=atol=1e-12 (tests/golden/dry_members.npz), two years around each member's worst day: north_star's bar on all 9 reach outputs."""
    m, tables = helpers.dry_fixture_problem()
    out, status, _ = oracle_lib.run(m['forcing'], m['doy'], m['member_params'], m['reach_params'], m['up_ptr'], m['up_idx'], m['opts'],
                                    n_threads=8)
    assert status.max() == 0
    worst = helpers.dry_worst_per_member(out, tables, marshal.OUT_COLUMNS)
    assert max(worst) < 1e-6, dict(zip(m['members'], worst))
    qr = out[marshal.OUT_COLUMNS.index('Qr')]
    assert min(float(qr[lo:hi, 0, k].min()) for k, (lo, hi, _) in enumerate(tables)) < 1e-3      # the reaches do nearly dry up


def test_oracle_second_pair_on_a_stiff_reach_chain(oracle_lib):
    """opts.stiff_pair (round 4): far down a reach network Cash-Karp's steps are bound by its real stability interval (|h x rate| <= 3.73,
    rate = cQ Qr**b_Q), not by accuracy; those attempts go, lane by lane, to a stability-optimised explicit 4(3) pair of the same six
    stages (include/simplyp_controller.h SIMPLYP_STIFF_*).  On config C4's chain (2 members, 240 days): a fifth fewer right-hand sides at
    no loss of accuracy against the converged solution; auto = on for a network, off for a single reach."""
    from simplyp_amd import synthetic
    pr = synthetic.c4_problem(2, n_reaches=256, n_days=240)

    def run(stiff, rtol=None, atol=None):
        o = pr['opts']
        r0, a0 = o.rtol, o.atol
        if rtol:
            o.rtol, o.atol = rtol, atol
        o.stiff_pair = stiff
        try:
            out, status, stats = oracle_lib.run(pr['forcing'], pr['doy'], pr['member_params'], pr['reach_params'], pr['up_ptr'], pr['up_idx'], o,
                                                out_reaches=pr['out_reaches'], n_threads=4)
        finally:
            o.rtol, o.atol, o.stiff_pair = r0, a0, 0
        assert status.max() == 0
        return out, stats
    truth, _ = run(-1, 1e-11, 1e-13)
    off, s_off = run(-1)
    on, s_on = run(1)
    auto, s_auto = run(0)
    assert np.array_equal(auto, on) and s_auto['rhs_evals'] == s_on['rhs_evals']          # a network: auto = on
    err = lambda a: float((np.abs(a - truth) / np.maximum(np.abs(truth), 1e-300)).max())
    assert err(off) < 5e-7 and err(on) < 5e-7 and err(on) < 1.5 * err(off), (err(off), err(on))
    assert s_on['rhs_evals'] < 0.75 * s_off['rhs_evals'], (s_on['rhs_evals'], s_off['rhs_evals'])      # (second pair + damping-aware weights)
    # a single reach: auto = off (bit-identical to the round-3 solver), and on changes next to nothing there
    m = helpers.marshal_scenario('tarland_2004_dynamic', E=1, out_mask=marshal.MASK_REACH5)
    outs = {}
    for sp in (0, -1, 1):
        m['opts'].stiff_pair = sp
        outs[sp] = oracle_lib.run(m['forcing'], m['doy'], m['member_params'], m['reach_params'], m['up_ptr'], m['up_idx'], m['opts'])[0]
    assert np.array_equal(outs[0], outs[-1])
    assert helpers.max_rel_err(outs[1], outs[-1], floor=1e-300) < 1e-6


def test_oracle_network_scheme_on_the_whole_c4_chain_against_the_reference(oracle_lib):
    """tests/golden/c4_deep.npz: 2 members of config C4's distribution on ALL 256 reaches of its chain, one year, through the
    unmodified reference at rtol=atol=1e-12 (make_golden.py --only c4deep, ~40 minutes of the reference per member); reaches 32, 64, 128,
    192 and the outlet.  The scheme at rtol 1e-10 meets those tables at 3e-10 (oracle and fixture are both converged at depth); the default
    solver -- second pair and damping-aware weights on -- at < 5e-7 (measured 1.9e-7), no worse than Cash-Karp alone (2.7e-7) at 60 % of
    its right-hand sides."""
    res = {}
    for key, solver, stiff in (('default', None, 0), ('ck', None, -1), ('tight', dict(rtol=1e-10, atol=1e-13), -1)):
        pr, tables = helpers.c4_members_problem(solver=solver, fname='c4_deep.npz')
        pr['opts'].stiff_pair = stiff
        out, status, stats = oracle_lib.run(pr['forcing'], pr['doy'], pr['member_params'], pr['reach_params'], pr['up_ptr'], pr['up_idx'], pr['opts'],
                                            out_reaches=pr['out_reaches'], n_threads=2)
        assert status.max() == 0
        res[key] = (max(helpers.c4_members_worst(out, tables).values()), stats['rhs_evals'])
    assert res['tight'][0] < 2e-9, res
    assert res['default'][0] < 5e-7 and res['ck'][0] < 5e-7 and res['default'][0] < 1.2 * res['ck'][0], res
    assert res['default'][1] < 0.7 * res['ck'][1], res


def test_oracle_default_solver_on_members_of_the_c4_distribution_against_the_reference(oracle_lib):
    """tests/golden/c4_members.npz: 4 members of config C4's own parameter distribution on the upper 16 reaches of its chain (routing,
    newly-converted land on every 4th reach, both dynamic options), one year, through the unmodified reference at rtol=atol=1e-12.
    Default solver (second pair on: a network) and Cash-Karp alone: north_star's bar on all 9 reach columns of the kept reaches."""
    pr, tables = helpers.c4_members_problem()
    for stiff in (0, -1):
        pr['opts'].stiff_pair = stiff
        out, status, _ = oracle_lib.run(pr['forcing'], pr['doy'], pr['member_params'], pr['reach_params'], pr['up_ptr'], pr['up_idx'], pr['opts'],
                                        out_reaches=pr['out_reaches'], n_threads=4)
        assert status.max() == 0
        worst = helpers.c4_members_worst(out, tables)
        assert max(worst.values()) < 1e-6, (stiff, worst)
    pr['opts'].stiff_pair = 0
