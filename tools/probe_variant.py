"""One variant library (SIMPLYP_HIP_LIB): ensemble-wide error at the default tolerance + time of the bench-shaped run."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np, torch
from simplyp_amd import engine, synthetic
eng = engine.get_engine(0)
E = 8192
def run(solver):
    pr = synthetic.c3_problem(E, solver=dict(solver, balance=0, time_chunk_days=-1))
    return eng.run(pr['forcing'], pr['doy'], pr['member_params'], pr['reach_params'], pr['up_ptr'], pr['up_idx'], pr['opts'])
truth, st, _ = run(dict(rtol=1e-11, atol=1e-13))
for rtol in (1e-8, 2e-8):
    out, st, stats = run(dict(rtol=rtol, atol=1e-12))
    pm = ((out - truth).abs() / truth.abs().clamp_min(1e-300)).amax(dim=(0, 1, 2))
    print('%s rtol %.0e: rhs/day %.1f max %.2e p99.9 %.2e >1e-6: %d' % (os.path.basename(engine.LIB_PATH), rtol, stats['rhs_evals'] / (E * out.shape[1]),
          pm.max(), torch.quantile(pm.float(), 0.999), int((pm > 1e-6).sum())), flush=True)
    del out
del truth
pr = synthetic.c3_problem(100000, solver=dict(out_slot_order=1, atol=1e-12))
dev = [eng.to_device(pr[k]) for k in ('forcing', 'doy', 'member_params', 'reach_params')]
o = None
for rep in range(2):
    o, st, stats = eng.run(dev[0], dev[1], dev[2], dev[3], pr['up_ptr'], pr['up_idx'], pr['opts'], out=o)
print('   E=100000 kernel %.1f ms pilot %.1f rhs/cd %.1f simt %.3f' % (stats['kernel_ms'], stats['pilot_ms'], stats['rhs_evals'] / (1e5 * o.shape[1]), stats['simt_efficiency']), flush=True)
