"""Time the bench-shaped run with whatever library SIMPLYP_HIP_LIB points to (compiler-flag / kernel variants)."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np, torch
from simplyp_amd import engine, synthetic
eng = engine.get_engine(0)
pr = synthetic.c3_problem(100000, solver=dict(out_slot_order=1))
dev = [eng.to_device(pr[k]) for k in ('forcing', 'doy', 'member_params', 'reach_params')]
o = None
for rep in range(3):
    o, st, stats = eng.run(dev[0], dev[1], dev[2], dev[3], pr['up_ptr'], pr['up_idx'], pr['opts'], out=o)
print('%s: kernel %.1f ms pilot %.1f rhs/cd %.1f simt %.3f checksum %.12e' % (os.path.basename(engine.LIB_PATH), stats['kernel_ms'], stats['pilot_ms'],
      stats['rhs_evals'] / (1e5 * o.shape[1]), stats['simt_efficiency'], float(o[1].sum())), flush=True)
