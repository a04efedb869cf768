"""Cost model of a lane-asynchronous day loop, fed with REAL step counts.

The kernel's lanes attempt in lockstep and meet at every day boundary, so a wave's day costs its slowest lane's attempts:
trips = sum_d max_lane a[lane, d].  A lane-asynchronous loop would let every lane run through its own days and pay the
day-boundary code (outputs, soil-P update, next day's constants: ~800 issue slots against ~780 for one attempt, i.e. B ~ 1
attempt) under an exec mask whenever some lanes cross a boundary.  Because SIMT code costs the same for 1 or 64 active lanes,
the boundary block is paid per EXECUTION, not per lane: the model below executes it when at least K lanes wait at a boundary
(or nobody can integrate), K = 1 .. 64, and counts wave-trips.

a[member, day] = attempted Cash-Karp steps, from the CPU oracle (it mirrors the kernel's step rule: same counts), for the first
`members` of the bench's C3 ensemble, grouped into waves of 64 by total cost as the load balancer does.
Usage: python tools/probe_async.py [members] [threads]   (CPU only; ~0.06 core-seconds per member)"""
import ctypes as C, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
from simplyp_amd import synthetic
from oracle import oracle

E = int(sys.argv[1]) if len(sys.argv) > 1 else 2048
threads = int(sys.argv[2]) if len(sys.argv) > 2 else 8
B = 800.0 / 777.0                    # day-boundary code in attempt-equivalents (issue slots, tools/isa_stats.py)
pr = synthetic.c3_problem(E)
D = pr['forcing'].shape[2]
trace = np.zeros((E, D), dtype=np.uint16)
L = oracle.lib()
L.simplyp_oracle_set_trace.restype = None
L.simplyp_oracle_set_trace(trace.ctypes.data_as(C.c_void_p))
t0 = time.time()
oracle.run(pr['forcing'], pr['doy'], pr['member_params'], pr['reach_params'], pr['up_ptr'], pr['up_idx'], pr['opts'], n_threads=threads)
L.simplyp_oracle_set_trace(None)
print('oracle: %d members x %d days in %.1f s; mean attempts/day %.2f' % (E, D, time.time() - t0, trace.mean()), flush=True)
order = np.argsort(-trace.sum(axis=1, dtype=np.int64), kind='stable')
a = trace[order].reshape(E // 64, 64, D).astype(np.int64)
useful = a.sum()
lock = a.max(axis=1).sum()                                  # wave-trips of the lockstep loop
lock_total = lock + B * a.shape[0] * D
print('lockstep (as built): SIMT efficiency %.3f; trips + boundaries = %.3e attempt-equivalents' % (useful / (64.0 * lock), lock_total))


def simulate(aw, K):
    """one wave: aw[64, D]; returns (attempt trips, boundary executions)"""
    day = np.zeros(64, dtype=np.int64)
    left = aw[:, 0].copy()
    waiting = np.zeros(64, dtype=bool)
    done = np.zeros(64, dtype=bool)
    trips = bounds = 0
    Dn = aw.shape[1]
    while not done.all():
        active = ~waiting & ~done
        if active.any() and waiting.sum() < K:
            # jump ahead to the next event: the smallest remaining count among the integrating lanes
            step = int(left[active].min())
            trips += step
            left[active] -= step
            waiting |= active & (left == 0)
        else:
            bounds += 1
            w = waiting.copy()
            day[w] += 1
            fin = w & (day >= Dn)
            done |= fin
            cont = w & ~fin
            left[cont] = aw[cont, day[cont]]
            waiting[:] = False
            # a day with zero attempts cannot happen (every day takes >= 1 step)
    return trips, bounds


n_w = min(a.shape[0], 8)
for K in (1, 2, 4, 8, 16, 32, 64):
    tt = bb = 0
    for w in range(n_w):
        t, b = simulate(a[w], K)
        tt += t; bb += b
    lk = a[:n_w].max(axis=1).sum() + B * n_w * D
    print('async, boundary block when >= %2d lanes wait: attempt trips %.3e (lockstep %.3e), boundary executions per wave-day %.2f, '
          'total %.3e = %.3f x lockstep' % (K, tt, a[:n_w].max(axis=1).sum(), bb / (n_w * D), tt + B * bb, (tt + B * bb) / lk), flush=True)
