"""Oracle-only prototype of the next step for reach networks: the slow stores (two soil boxes, groundwater)
integrated apart from the reach, which reads the land-phase inflow and its TDP load off a piecewise Hermite record (oracle integrator
14, ORACLE_INTEG_SPLIT_AUG in oracle/simplyp_oracle.c).  Prints, beside the shipped scheme: the worst error against the converged
solution on config C4's chain and against the reference's own tables on the committed fixtures, attempts of the reach pass per
reach-day and of the slow pass per reach-day (a kernel that integrates the slow stores once per member pays the latter once per
member-day), for cubic and quintic records and a range of slow-pass tolerances.  Log: profiles/r04_c4/split_prototype.log.
Usage: python tools/probe_split.py [members reaches days]"""
import ctypes as C, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, 'tests'))
import numpy as np
import helpers
import test_oracle_series as series
from simplyp_amd import synthetic
from oracle import oracle

E = int(sys.argv[1]) if len(sys.argv) > 1 else 4
S = int(sys.argv[2]) if len(sys.argv) > 2 else 256
D = int(sys.argv[3]) if len(sys.argv) > 3 else 200
pr = synthetic.c4_problem(E, n_reaches=S, n_days=D)
L = oracle.lib()
L.simplyp_oracle_split_config.argtypes = [C.c_double, C.c_int, C.c_int, C.c_int, C.c_double]
L.simplyp_oracle_split_counts.argtypes = [C.POINTER(C.c_uint64)]
L.simplyp_oracle_split_order.argtypes = [C.c_int]
oreach = sorted(set(list(range(0, S, 8)) + [S - 1]))
SPLIT = 14


def configure(order=5, slow_tol=1e-4, ni=6, carry=1, cut_keeps=1, h0=0.05):
    L.simplyp_oracle_split_config(slow_tol, ni, carry, cut_keeps, h0)
    L.simplyp_oracle_split_order(order)


def counts():
    c = (C.c_uint64 * 4)()
    L.simplyp_oracle_split_counts(c)
    return list(c)


def chain(integ, rtol=None, atol=None, stiff=1, **kw):
    o = pr['opts']
    keep = (o.integrator, o.rtol, o.atol, o.stiff_pair)
    o.integrator, o.stiff_pair = integ, stiff
    if rtol:
        o.rtol, o.atol = rtol, atol
    configure(**kw)
    try:
        out, st, stats = oracle.run(pr['forcing'], pr['doy'], pr['member_params'], pr['reach_params'], pr['up_ptr'], pr['up_idx'], o,
                                    out_reaches=oreach, n_threads=1 if integ == SPLIT else 8)     # (the probe's counters are not thread-safe)
    finally:
        o.integrator, o.rtol, o.atol, o.stiff_pair = keep
    return out, stats, counts()


def fixtures(integ, **kw):
    """worst error on the 9 reach columns against the reference's tables (odeint at rtol = atol = 1e-12)"""
    res = []
    for name in series.SCENARIOS[1:]:
        m = helpers.marshal_scenario(name)
        m['opts'].integrator = integ
        configure(**kw)
        out, status, stats = oracle.run(m['forcing'], m['doy'], m['member_params'], m['reach_params'], m['up_ptr'], m['up_idx'], m['opts'])
        errs = series.column_errors(out, m['scs'], helpers.golden_tables(name, 'tight'))
        res.append('%s %.1e' % (name.split('_')[0], max(errs[c] for c in series.REACH_COLS)))
    p4, tables = helpers.c4_members_problem()
    p4['opts'].integrator = integ
    configure(**kw)
    out, status, stats = oracle.run(p4['forcing'], p4['doy'], p4['member_params'], p4['reach_params'], p4['up_ptr'], p4['up_idx'], p4['opts'],
                                    out_reaches=p4['out_reaches'], n_threads=1)
    res.append('c4_members %.1e' % max(helpers.c4_members_worst(out, tables).values()))
    return ', '.join(res)


truth, _, _ = chain(2, 1e-11, 1e-13, stiff=-1)
cd = E * D * S
print('config C4 chain: %d members x %d reaches x %d days; fixtures: %s + c4_members' % (E, S, D, ' '.join(n.split('_')[0] for n in series.SCENARIOS[1:])))
out, stats, _ = chain(2)
print('%-44s chain %.2e, %.2f attempts per reach-day | %s' % ('shipped (11 states per reach, second pair)', (np.abs(out - truth) / np.maximum(np.abs(truth), 1e-300)).max(),
                                                                (stats['steps'] + stats['rejected']) / cd, fixtures(2)), flush=True)
for order in (3, 5):
    for slow_tol in (1e-3, 1e-4, 1e-5):
        out, stats, c = chain(SPLIT, order=order, slow_tol=slow_tol)
        rel = np.abs(out - truth) / np.maximum(np.abs(truth), 1e-300)
        print('%-44s chain %.2e, reach pass %.2f + slow pass %.2f attempts per reach-day (worst day %d), %.2f segments of 6 | %s'
              % ('split, %s record, slow tolerance x %g' % ('cubic' if order == 3 else 'quintic', slow_tol), rel.max(), c[1] / cd, c[0] / cd, c[3],
                 c[2] / cd, fixtures(SPLIT, order=order, slow_tol=slow_tol)), flush=True)
out, stats, c = chain(SPLIT, rtol=1e-11, atol=1e-13, stiff=-1)
print('%-44s chain %.2e' % ('split at rtol 1e-11 (quintic, x 1e-4)', (np.abs(out - truth) / np.maximum(np.abs(truth), 1e-300)).max()))
