"""Ensemble-wide error of the default solver on a reach network (config C4's synthetic chain, reduced): every member's max
relative error over the outlet's REACH-5 daily columns against the same kernel at rtol 1e-11 / atol 1e-13.
Usage: python tools/probe_tolerance_network.py [members reaches days]"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np, torch
from simplyp_amd import engine, synthetic

E = int(sys.argv[1]) if len(sys.argv) > 1 else 2048
S = int(sys.argv[2]) if len(sys.argv) > 2 else 32
D = int(sys.argv[3]) if len(sys.argv) > 3 else 2000
eng = engine.get_engine(0)


def run(solver):
    pr = synthetic.c4_problem(E, n_reaches=S, n_days=D, solver=solver)
    return eng.run(pr['forcing'], pr['doy'], pr['member_params'], pr['reach_params'], pr['up_ptr'], pr['up_idx'], pr['opts'],
                   out_reaches=pr['out_reaches'])


truth, st, s0 = run(dict(rtol=1e-11, atol=1e-13, stiff_pair=-1))       # (Cash-Karp alone: no second pair, no damping-aware weights)
out, st1, s1 = run(None)
rel = (out - truth).abs() / truth.abs().clamp_min(1e-300)
pm = rel.amax(dim=(0, 1, 2))
print('E=%d S=%d D=%d: default rhs/cd %.1f (truth %.1f) flagged %d/%d | member max-rel-err median %.2e p99 %.2e max %.2e | > 1e-6: %d | per column %s'
      % (E, S, D, s1['rhs_evals'] / (E * S * D), s0['rhs_evals'] / (E * S * D), int((st1 != 0).sum()), int((st != 0).sum()),
         pm.median(), torch.quantile(pm.float(), 0.99), pm.max(), int((pm > 1e-6).sum()), ['%.1e' % v for v in rel.amax(dim=(1, 2, 3)).tolist()]))
