"""CPU cost model: a small ensemble that runs in ONE round of waves (fewer member groups than SIMDs -- a 100 000-member
ensemble split over 8 GPUs: 12 500 members on 1 024 SIMDs) takes as long as its SLOWEST wave, and a wave's day costs the
attempts of its slowest member.  Would handing the expensive members to thin waves (few members each, nearly their own pace)
and packing the cheap ones densely shorten the pass?

a[member, day] = attempted Cash-Karp steps from the CPU oracle (it mirrors the kernel's step rule), first `members` of the bench's
C3 ensemble.  Wave time model: sum_d (max_m a[m, d] * ATT + BND) issue slots (four-lane kernel: ATT ~ 570, BND ~ 1000).
Strategies: (a) as built -- consecutive members, equal widths; (b) members ranked by total cost, widths grown along the rank so
that every wave's predicted time is about equal.
Usage: python tools/probe_widths.py [members] [threads]"""
import ctypes as C, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
from simplyp_amd import synthetic
from oracle import oracle

E = int(sys.argv[1]) if len(sys.argv) > 1 else 12500
threads = int(sys.argv[2]) if len(sys.argv) > 2 else 8
ATT, BND, SLOTS, WMAX = 570.0, 1000.0, 1024, 16
pr = synthetic.c3_problem(E)
D = pr['forcing'].shape[2]
trace = np.zeros((E, D), dtype=np.uint16)
L = oracle.lib()
L.simplyp_oracle_set_trace.restype = None
L.simplyp_oracle_set_trace(trace.ctypes.data_as(C.c_void_p))
t0 = time.time()
oracle.run(pr['forcing'], pr['doy'], pr['member_params'], pr['reach_params'], pr['up_ptr'], pr['up_idx'], pr['opts'], n_threads=threads)
L.simplyp_oracle_set_trace(None)
tot = trace.sum(axis=1, dtype=np.int64)
print('oracle: %d members x %d days in %.0f s; attempts/day mean %.2f, min member %.2f, median %.2f, p99 %.2f, max %.2f'
      % (E, D, time.time() - t0, trace.mean(), tot.min() / D, np.median(tot) / D, np.percentile(tot, 99) / D, tot.max() / D), flush=True)


def wave_time(members):
    return float(trace[members].max(axis=0).astype(np.int64).sum()) * ATT + BND * D


def report(name, waves):
    t = np.array([wave_time(w) for w in waves])
    print('%-58s %4d waves, widths %d..%d: slowest wave %.3e slots (= %.0f ms at 2.1 ns), mean wave %.3e, slowest lone member %.3e'
          % (name, len(waves), min(len(w) for w in waves), max(len(w) for w in waves), t.max(), t.max() * 2.1e-6, t.mean(),
             tot.max() * ATT + BND * D), flush=True)
    return t.max()


w0 = (E + SLOTS - 1) // SLOTS
base = report('(a) as built: consecutive members, %d per wave' % w0, [np.arange(i, min(E, i + w0)) for i in range(0, E, w0)])
order = np.argsort(-tot, kind='stable')
report('(a2) cost-ranked, equal widths', [order[i:i + w0] for i in range(0, E, w0)])
# (b) greedy: walk down the cost rank; close a wave when adding the next member would push its predicted time above the target.
# The target is found by bisection so that the waves just fit the SIMD slots.
def pack(target):
    waves, cur = [], []
    cur_max = None
    for m in order:
        cand = trace[m] if cur_max is None else np.maximum(cur_max, trace[m])
        t = float(cand.astype(np.int64).sum()) * ATT + BND * D
        if cur and (t > target or len(cur) >= WMAX):
            waves.append(np.array(cur)); cur = [m]; cur_max = trace[m].copy()
        else:
            cur.append(m); cur_max = cand
    if cur: waves.append(np.array(cur))
    return waves
lo, hi = tot.max() * ATT + BND * D, base
for _ in range(12):
    mid = 0.5 * (lo + hi)
    if len(pack(mid)) <= SLOTS: hi = mid
    else: lo = mid
best = report('(b) cost-ranked, widths grown so that wave times are equal', pack(hi))
print('model gain of (b) over (a): %.3f x' % (base / best))
