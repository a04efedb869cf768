"""CPU oracle (= the kernel's step rule): where do the attempts of a reach network go?  For config C4's synthetic chain, per reach:
attempts per reach-day, share rejected, mean relaxation rate cQ Qr**b_Q (1/day), and the accepted steps binned by h x rate (the step in
units of the reach's relaxation time; Cash-Karp's 5th-order solution is stable on the real axis down to -3.73).  Then the same run with
the step capped at h x rate <= cap (a controller that never probes the stability limit): attempts, rejections and error against the
converged solution.  Result (profiles/r04_c4/steps_by_reach.log): every reach spends ~17 attempts a day on the
transient that follows midnight's jump in the forcing (h x rate < 2: accuracy-bound) and the rest -- 0 for a headwater, 90 for the
stiffest reach, 20 on average -- at h x rate 3-4: bound by the explicit pair's stability, not by accuracy.
Usage: python tools/probe_c4_steps.py [members reaches days [cap ...]]"""
import ctypes as C, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
from simplyp_amd import synthetic
from oracle import oracle

E = int(sys.argv[1]) if len(sys.argv) > 1 else 8
S = int(sys.argv[2]) if len(sys.argv) > 2 else 256
D = int(sys.argv[3]) if len(sys.argv) > 3 else 400
caps = [float(x) for x in sys.argv[4:]] or [3.0, 3.4, 3.6]
pr = synthetic.c4_problem(E, n_reaches=S, n_days=D)
L = oracle.lib()
L.simplyp_oracle_set_stab_cap.argtypes = [C.c_double]
oreach = sorted(set(list(range(0, S, 8)) + [S - 1]))


def run(cap, rtol=None, atol=None, stiff=-1):
    """stiff = -1: Cash-Karp alone (what this probe takes apart); 1: with the stability-optimised second pair (opts.stiff_pair)"""
    pr['opts'].stiff_pair = stiff
    buf = np.zeros((S, 12))
    L.simplyp_oracle_set_reach_trace(buf.ctypes.data_as(C.POINTER(C.c_double)))
    L.simplyp_oracle_set_stab_cap(cap)
    r0, a0 = pr['opts'].rtol, pr['opts'].atol
    if rtol:
        pr['opts'].rtol, pr['opts'].atol = rtol, atol
    try:
        out, st, stats = oracle.run(pr['forcing'], pr['doy'], pr['member_params'], pr['reach_params'], pr['up_ptr'], pr['up_idx'], pr['opts'],
                                    out_reaches=oreach, n_threads=8)
    finally:
        pr['opts'].rtol, pr['opts'].atol = r0, a0
        pr['opts'].stiff_pair = 0
        L.simplyp_oracle_set_reach_trace(None)
        L.simplyp_oracle_set_stab_cap(0.0)
    return out, buf, stats


truth, _, _ = run(0.0, 1e-11, 1e-13)
np.set_printoptions(linewidth=200, precision=2, suppress=True)
for cap in [0.0] + caps + ['second pair']:
    if cap == 'second pair':
        out, buf, stats = run(0.0, stiff=1)
    else:
        out, buf, stats = run(cap)
    rel = np.abs(out - truth) / np.maximum(np.abs(truth), 1e-300)
    cd = E * D
    print('cap %s: %d members x %d reaches x %d days: attempts per reach-day %.2f, rejected %.2f %%, rhs per catchment-day %.1f, worst error vs the converged '
          'solution %.2e | accepted steps per reach-day with h x rate < 0.5, < 1, < 2, < 3, < 4, >= 4: %s'
          % ('off' if cap == 0 else (cap if isinstance(cap, str) else '%.1f' % cap), E, S, D, buf[:, 0].sum() / (cd * S), 100 * buf[:, 1].sum() / buf[:, 0].sum(), stats['rhs_evals'] / (cd * S),
             rel.max(), np.round(buf[:, 2:8].sum(0) / (cd * S), 2)), flush=True)
    if cap == 0.0 or cap == 'second pair':
        print('  reach  attempts/day  rejected %  rate 1/day   accepted per day by h x rate [<.5 <1 <2 <3 <4 >=4]   rejections at h x rate >= 2 (%)')
        for s in sorted(set(list(range(0, S, max(1, S // 32))) + [S - 1])):
            b = buf[s]
            print('  %5d  %12.1f  %10.1f  %10.0f   %s   %3.0f' % (s, b[0] / cd, 100 * b[1] / b[0], b[8] / b[0], b[2:8] / cd, 100 * b[9] / max(b[1], 1)))
