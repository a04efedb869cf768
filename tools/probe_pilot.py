"""Pilot length of the load balancer vs total time on the bench workload."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np, torch
from simplyp_amd import engine, synthetic
eng = engine.get_engine(0)
E = 100000
pr = synthetic.c3_problem(E, solver=dict(out_slot_order=1))
dev = [eng.to_device(pr[k]) for k in ('forcing', 'doy', 'member_params', 'reach_params')]
o = None
for days in (32, 48, 64, 96, 128, 160, 224):
    pr['opts'].balance_pilot_days = days
    for rep in range(2):
        o, st, stats = eng.run(dev[0], dev[1], dev[2], dev[3], pr['up_ptr'], pr['up_idx'], pr['opts'], out=o)
    print('pilot %3d days: pilot %.1f ms kernel %.1f ms total %.1f simt %.3f' % (days, stats['pilot_ms'], stats['kernel_ms'], stats['pilot_ms'] + stats['kernel_ms'], stats['simt_efficiency']), flush=True)
