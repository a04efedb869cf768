"""Streamed C3 pass (table delivered to pinned host memory) against the length of the load balancer's pilot: the streamed pass is
bound by the copies, so a shorter pilot (less exact member order, slower kernel, but an earlier first copy) may shorten it."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np, torch
from simplyp_amd import engine, synthetic
eng = engine.get_engine(0)
E = 100000
pr = synthetic.c3_problem(E, solver=dict(out_slot_order=1))
dev = [eng.to_device(pr[k]) for k in ('forcing', 'doy', 'member_params', 'reach_params')]
D = pr['forcing'].shape[2]
host = engine.pinned_empty((5, D, 1, E))
o = None
for days in (64, 8, 16, 32, 64, 0):
    pr['opts'].balance_pilot_days = days if days else 64
    pr['opts'].balance = 2 if days else 0
    w = []
    for rep in range(4):
        o, st, stats = eng.run(dev[0], dev[1], dev[2], dev[3], pr['up_ptr'], pr['up_idx'], pr['opts'], out=o, host_out=host)
        if rep: w.append(stats['wall_ms'])
    print('pilot %3d days: pilot %.1f ms kernel %.1f ms simt %.3f | streamed pass wall %.1f ms (min %.1f) tail %.1f chunks %d' % (
        days, stats['pilot_ms'], stats['kernel_ms'], stats['simt_efficiency'], np.mean(w), min(w), stats['d2h_tail_ms'], stats['streamed_chunks']), flush=True)
