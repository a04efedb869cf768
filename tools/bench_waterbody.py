"""simplyp_waterbody at ensemble scale: the 3-reach confluence scenario (2004, 366 days) x `members` Monte-Carlo members,
two and three reaches summed, all 11 columns; prints kernel ms and achieved GB/s (algorithmic bytes: 32 B read per member,
day and summed reach + 8 B written per column).  Usage: python tools/bench_waterbody.py [members]"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'tests'))
import numpy as np, torch
import helpers
from simplyp_amd import engine, marshal

E = int(sys.argv[1]) if len(sys.argv) > 1 else 100000
m = helpers.marshal_scenario('confluence3_nc_2004', E=E, out_mask=marshal.mask_of_columns(['Qr', 'Msus_kg/day', 'TDP_kg/day', 'PP_kg/day']),
                             solver=dict(rtol=1e-6, atol=1e-9))
rng = np.random.default_rng(1)
m['member_params'][marshal.PM_NAMES.index('fc')] *= rng.uniform(0.85, 1.15, E)
eng = engine.get_engine(0)
out, status, st = eng.run(m['forcing'], m['doy'], m['member_params'], m['reach_params'], m['up_ptr'], m['up_idx'], m['opts'])
print('table %s (%.2f GB) in %.1f ms' % (tuple(out.shape), out.numel() * 8 / 1e9, st['kernel_ms']), flush=True)
rp = eng.to_device(m['reach_params'])
for reaches in ([0, 2], [0, 1, 2]):
    for rep in range(3):
        wb, info = eng.waterbody(out, m['opts'].out_mask, reaches, 0.7, rp)
    print('sum of reaches %s: %.3f ms, %.2f GB moved -> %.0f GB/s = %.1f %% of 8 TB/s' %
          (reaches, info['kernel_ms'], info['bytes_moved'] / 1e9, info['bytes_moved'] / info['kernel_ms'] / 1e6,
           info['bytes_moved'] / info['kernel_ms'] / 1e6 / 80.0), flush=True)
