"""CPU: would another embedded 5(4) pair need fewer right-hand sides than Cash-Karp under the same knee-aware controller?
Tsitouras 5(4) (7 stages, first-same-as-last: 6 new right-hand sides per attempt, like Cash-Karp) against Cash-Karp on members of
the C3 bench ensemble, 30 years, REACH-5: right-hand sides per catchment-day against the WORST member's error (truth: Cash-Karp
at rtol 1e-11).  Also Dormand-Prince 8(5,3) (12 new right-hand sides per attempt; would suit the four-lane kernel, whose lanes hold
three components each).  The oracle carries both tableaux for this probe only (ORACLE_INTEG_TSIT5_AUG = 12, _DOP853_AUG = 13);
SIMPLYP_PROBE_PAIRS=dop853,cash-karp selects.
Usage: python tools/probe_pair.py [members [threads]]"""
import os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from simplyp_amd import synthetic
from oracle import oracle

E = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
thr = int(sys.argv[2]) if len(sys.argv) > 2 else (os.cpu_count() or 1)
pr = synthetic.c3_problem(100000)
sel = np.arange(E) * (100000 // E)
mp = np.ascontiguousarray(pr['member_params'][:, sel]); rp = np.ascontiguousarray(pr['reach_params'][:, :, sel])
D = pr['forcing'].shape[2]

def run(integ, rtol, atol):
    o = pr['opts']
    o.integrator, o.rtol, o.atol = integ, rtol, atol
    t0 = time.time()
    out, status, st = oracle.run(pr['forcing'], pr['doy'], mp, rp, pr['up_ptr'], pr['up_idx'], o, n_threads=thr)
    assert status.max() == 0
    return out, st, time.time() - t0

truth, st, dt = run(2, 1e-11, 1e-13)
print('truth: Cash-Karp rtol 1e-11: %.1f rhs/cd, %.0f s' % (st['rhs_evals'] / (E * D), dt), flush=True)
PAIRS = (('cash-karp', 2, (1e-7, 2e-7, 5e-8)), ('tsit5', 12, (1e-7, 2e-7, 4e-7, 5e-8)), ('dop853', 13, (1e-6, 3e-7, 1e-7, 3e-8, 1e-8, 1e-9)))
only = os.environ.get('SIMPLYP_PROBE_PAIRS')
for name, integ, rtols in [p for p in PAIRS if not only or p[0] in only.split(',')]:
    for rtol in rtols:
        out, st, dt = run(integ, rtol, 1e-12)
        rel = np.abs(out - truth) / np.maximum(np.abs(truth), 1e-300)
        w = rel.max(axis=(0, 1, 2))
        print('%-9s rtol %.0e: %6.1f rhs/cd  %4.1f %% rejected  worst member %.2e  p99.9 %.2e  median %.2e  members > 1e-6: %d  (%.0f s)'
              % (name, rtol, st['rhs_evals'] / (E * D), 100.0 * st['rejected'] / max(st['steps'], 1), w.max(), np.percentile(w, 99.9),
                 np.median(w), int((w > 1e-6).sum()), dt), flush=True)
