"""How much of the Monte-Carlo slowdown is lane divergence, and how much does sorting members recover?"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'tests'))
import numpy as np, torch
from simplyp_amd import engine, marshal, synthetic

eng = engine.get_engine(0)
E = int(sys.argv[1]) if len(sys.argv) > 1 else 65536
pr = synthetic.c3_problem(E)
f, doy = eng.to_device(pr['forcing']), eng.to_device(pr['doy'])
mp, rp = eng.to_device(pr['member_params']), eng.to_device(pr['reach_params'])
D = f.shape[2]
out = torch.empty((5, D, 1, E), dtype=torch.float64, device='cuda')

def run(mp_, rp_, f_=f, doy_=doy, out_=out, tag=''):
    cnt = torch.zeros(E, dtype=torch.int32, device='cuda')
    for rep in range(2):
        o, st, stats = eng.run(f_, doy_, mp_, rp_, pr['up_ptr'], pr['up_idx'], pr['opts'], out=out_, member_rhs=cnt)
    n = E * f_.shape[2]
    print('%-34s kernel %8.1f ms  rhs/cd %.1f  -> %.3e cd/s' % (tag, stats['kernel_ms'], stats['rhs_evals'] / n, n / stats['kernel_ms'] * 1e3), flush=True)
    return cnt.clone(), stats

cnt, _ = run(mp, rp, tag='unsorted (as drawn)')
c = cnt.cpu().numpy().astype(np.int64)
print('per-member rhs/day: min %.1f p5 %.1f median %.1f p95 %.1f max %.1f' % tuple(np.percentile(c / D, [0, 5, 50, 95, 100])))
w = c.reshape(-1, 64) if E % 64 == 0 else None
if w is not None:
    print('mean over waves of (max lane / mean lane) of TOTAL counts: %.3f' % np.mean(w.max(1) / w.mean(1)))

for name, order in [('sorted by full-run count, ascending', np.argsort(c, kind='stable')),
                    ('sorted by full-run count, descending', np.argsort(-c, kind='stable'))]:
    idx = torch.as_tensor(order, device='cuda')
    run(mp[:, idx].contiguous(), rp[:, :, idx].contiguous(), tag=name)

# pilot: first 256 days only
Dp = 256
outp = torch.empty((5, Dp, 1, E), dtype=torch.float64, device='cuda')
cp, sp = run(mp, rp, f[:, :, :Dp].contiguous(), doy[:Dp].contiguous(), outp, tag='pilot run (256 days)')
order = np.argsort(-cp.cpu().numpy().astype(np.int64), kind='stable')
idx = torch.as_tensor(order, device='cuda')
run(mp[:, idx].contiguous(), rp[:, :, idx].contiguous(), tag='sorted by 256-day pilot, descending')
# analytic key: reach rate constant a_Q/(1-b_Q)/L
aQ = pr['member_params'][marshal.PM_NAMES.index('a_Q')]; bQ = pr['member_params'][marshal.PM_NAMES.index('b_Q')]
key = aQ / (1 - bQ)
idx = torch.as_tensor(np.argsort(-key, kind='stable'), device='cuda')
run(mp[:, idx].contiguous(), rp[:, :, idx].contiguous(), tag='sorted by a_Q/(1-b_Q), descending')
