"""Cost model: would re-forming the 64-member groups at every time-chunk boundary raise SIMT efficiency?

The task-queue kernel hands a member's state through HBM at every chunk boundary anyway, so the groups could be re-formed
there (outputs would become 8-byte scattered stores; HBM has 99 % headroom).  A wave's day costs its slowest lane's attempts:
trips = sum over days of max over lanes of a[lane, day]; efficiency = sum a / (64 trips).

a[member, day] = attempted Cash-Karp steps from the CPU oracle (it mirrors the kernel's step rule: same counts) for the first
`members` of the bench's C3 ensemble.  Strategies (groups of 64 consecutive members after sorting by a key):
  static      one order for the whole run: by total cost (what opts.balance did first: 0.78 on the GPU at 100 000 members)
  foresight   per chunk, by the member's attempts in THAT chunk (upper bound of anything that sorts on chunk totals)
  previous    per chunk, by the member's attempts in the PREVIOUS chunk (what a kernel could actually do)
  foresight2  per chunk, by total, then inside blocks of 8 groups by the chunk's storm share (max-day attempts / total)
Usage: python tools/probe_regroup.py [members] [threads] [chunk_days]   (CPU only; ~0.06 core-seconds per member)"""
import ctypes as C, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
from simplyp_amd import synthetic
from oracle import oracle

E = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
threads = int(sys.argv[2]) if len(sys.argv) > 2 else 8
chunk = int(sys.argv[3]) if len(sys.argv) > 3 else 64
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
a = trace.astype(np.int32)
useful = float(a.sum())


def trips(block, order):
    """wave-trips of the day range `block` ([E, days]) when members are grouped 64 at a time in `order`"""
    b = block[order]
    n = (len(order) // 64) * 64
    t = b[:n].reshape(-1, 64, b.shape[1]).max(axis=1).sum()
    if n < len(order): t += b[n:].max(axis=0).sum()
    return float(t)


static = np.argsort(-a.sum(axis=1), kind='stable')
res = {'static': trips(a, static), 'foresight': 0.0, 'previous': 0.0, 'foresight2': 0.0}
prev_key = a[:, :chunk].sum(axis=1)
for c0 in range(0, D, chunk):
    blk = a[:, c0:c0 + chunk]
    tot = blk.sum(axis=1)
    res['foresight'] += trips(blk, np.argsort(-tot, kind='stable'))
    res['previous'] += trips(blk, static if c0 == 0 else np.argsort(-prev_key, kind='stable'))
    o = np.argsort(-tot, kind='stable')
    share = blk.max(axis=1) / np.maximum(tot, 1)
    o2 = []
    for b0 in range(0, E, 512):
        sub = o[b0:b0 + 512]
        o2.append(sub[np.argsort(share[sub], kind='stable')])
    res['foresight2'] += trips(blk, np.concatenate(o2))
    prev_key = tot
for k, v in res.items():
    print('%-11s SIMT efficiency %.3f  (wave-trips %.4e)' % (k, useful / (64.0 * v), v))
