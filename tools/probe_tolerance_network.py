"""Ensemble-wide error of the default solver on a reach network (config C4's synthetic chain, reduced): every member's max
relative error over the outlet's REACH-5 daily columns against the same kernel at rtol 1e-11 / atol 1e-13.
Usage: python tools/probe_tolerance_network.py [members reaches days]
Environment: SIMPLYP_PROBE_PSCALE=x runs it on a climate with x times the precipitation and 1/x times the PET, SIMPLYP_PROBE_QGMIN=f
scales every member's Qg_min by f (towards reaches that nearly dry up: where the flow equation amplifies errors)."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np, torch
from simplyp_amd import engine, synthetic

E = int(sys.argv[1]) if len(sys.argv) > 1 else 2048
S = int(sys.argv[2]) if len(sys.argv) > 2 else 32
D = int(sys.argv[3]) if len(sys.argv) > 3 else 2000
eng = engine.get_engine(0)


PSCALE = float(os.environ.get('SIMPLYP_PROBE_PSCALE', '1'))
QGMIN = float(os.environ.get('SIMPLYP_PROBE_QGMIN', '1'))


def run(solver):
    pr = synthetic.c4_problem(E, n_reaches=S, n_days=D, solver=solver)
    if PSCALE != 1.0:
        f = pr['forcing'].copy()
        f[:, 0] *= PSCALE
        f[:, 1] /= PSCALE
        pr['forcing'] = f
    if QGMIN != 1.0:
        from simplyp_amd import marshal
        mp = pr['member_params'].copy()
        mp[[n for n, _ in marshal.PM_SPEC].index('Qg_min')] *= QGMIN
        pr['member_params'] = mp
    return eng.run(pr['forcing'], pr['doy'], pr['member_params'], pr['reach_params'], pr['up_ptr'], pr['up_idx'], pr['opts'],
                   out_reaches=pr['out_reaches'])


truth, st, s0 = run(dict(rtol=1e-11, atol=1e-13, stiff_pair=-1))       # (Cash-Karp alone: no second pair, no damping-aware weights)
out, st1, s1 = run(None)
rel = (out - truth).abs() / truth.abs().clamp_min(1e-300)
pm = rel.amax(dim=(0, 1, 2))
print('E=%d S=%d D=%d%s: default rhs/cd %.1f (truth %.1f) flagged %d/%d | member max-rel-err median %.2e p99 %.2e max %.2e | > 1e-6: %d | per column %s'
      % (E, S, D, '' if PSCALE == 1.0 and QGMIN == 1.0 else ' (precipitation x %g, Qg_min x %g)' % (PSCALE, QGMIN), s1['rhs_evals'] / (E * S * D), s0['rhs_evals'] / (E * S * D), int((st1 != 0).sum()), int((st != 0).sum()),
         pm.median(), torch.quantile(pm.float(), 0.99), pm.max(), int((pm > 1e-6).sum()), ['%.1e' % v for v in rel.amax(dim=(1, 2, 3)).tolist()]))
