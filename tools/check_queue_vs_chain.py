import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'tests'))
import numpy as np, torch
import helpers
from simplyp_amd import engine
eng = engine.get_engine(0)
E = 130
m = helpers.marshal_scenario('tarland_2004_dynamic', E=E, solver=dict(time_chunk_days=-1, balance=0))
m['forcing'] = np.ascontiguousarray(np.tile(m['forcing'], (1, 1, 3))); m['doy'] = np.ascontiguousarray(np.tile(m['doy'], 3))
a, sa, _ = eng.run(m['forcing'], m['doy'], m['member_params'], m['reach_params'], m['up_ptr'], m['up_idx'], m['opts'])
m['opts'].time_chunk_days = 256
for trial in range(3):
    b, sb, st = eng.run(m['forcing'], m['doy'], m['member_params'], m['reach_params'], m['up_ptr'], m['up_idx'], m['opts'])
    a_, b_ = a.cpu().numpy(), b.cpu().numpy()
    bad = ~((a_ == b_) | (np.isnan(a_) & np.isnan(b_)))
    mem = np.unique(np.nonzero(bad)[3]); days = np.nonzero(bad)[1]
    print('trial', trial, 'members differing', mem[:20], 'n', len(mem), 'first day', days.min() if len(days) else None, 'status', sb.cpu().numpy()[mem[:10]] if len(mem) else None, flush=True)
    if len(mem):
        e = mem[0]; d0 = np.nonzero(bad[:, :, 0, e])[1].min()
        print('  member', e, 'first bad day', d0, 'cols', np.nonzero(bad[:, d0, 0, e])[0][:8], a_[4, d0-1:d0+2, 0, e], b_[4, d0-1:d0+2, 0, e])
