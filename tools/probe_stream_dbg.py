import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'tests'))
import numpy as np, torch
from simplyp_amd import engine, marshal, synthetic
print({k: v for k, v in os.environ.items() if k.startswith(('HSA', 'HIP', 'ROC', 'AMD'))})
E = int(sys.argv[1]) if len(sys.argv) > 1 else 20000
eng = engine.get_engine(0)
pr = synthetic.c3_problem(E, solver=dict(out_slot_order=1))
D = pr['forcing'].shape[2]
host = engine.pinned_empty((5, D, 1, E))
for i in range(2):
    t = time.perf_counter()
    o, s, st = eng.run(pr['forcing'], pr['doy'], pr['member_params'], pr['reach_params'], pr['up_ptr'], pr['up_idx'], pr['opts'], host_out=host)
    print('wall %.1f' % ((time.perf_counter() - t) * 1e3), {k: st[k] for k in ('kernel_ms', 'pilot_ms', 'd2h_tail_ms', 'streamed_chunks', 'wall_ms', 'queued')}, flush=True)
