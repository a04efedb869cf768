"""Goodness-of-fit reduction on the bench workload's daily table (C3: 100 000 members x 30 years, REACH-5 columns in
slot order) against the shipped Tarland observations: time, algorithmic bytes, fraction of the HBM roofline.
Usage: python tools/bench_gof.py [members] [reps]"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'tests'))
import numpy as np, torch
import helpers
from simplyp_amd import engine, marshal, synthetic, visualise_results as vr

E = int(sys.argv[1]) if len(sys.argv) > 1 else 100000
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 5
eng = engine.get_engine(0)
pr = synthetic.c3_problem(E, solver=dict(out_slot_order=1, rtol=1e-6, atol=1e-8))     # the table's content does not matter here
rp = eng.to_device(pr['reach_params'])
out, status, st = eng.run(pr['forcing'], pr['doy'], pr['member_params'], rp, pr['up_ptr'], pr['up_idx'], pr['opts'])
obs = vr.observation_array(helpers.observations('1981-01-01', '2010-12-31'), [1], pr['met'].index)
ms = []
for _ in range(reps):
    gof, info = eng.gof(out, marshal.MASK_REACH5, obs, 0.7, rp, member_of_slot=st['member_of_slot'])
    ms.append(info['kernel_ms'])
t = float(np.median(ms[1:])) if reps > 1 else ms[0]
print('E=%d: %d discharge days + %d chemistry days, %d/%d slices; %.1f MB algorithmic; kernels %.3f ms (median of %d) -> %.0f GB/s = %.1f %% of 8 TB/s'
      % (E, info['n_q_days'], info['n_chem_days'], info['n_chunks_q'], info['n_chunks_chem'], info['bytes_read'] / 1e6, t, reps - 1,
         info['bytes_read'] / t / 1e6, info['bytes_read'] / t / 1e6 / 80.0), flush=True)
g = gof.cpu().numpy()
print('NSE(Q): median %.3f best %.3f; finite %d of %d members' % (np.nanmedian(g[1, 0, 0]), np.nanmax(g[1, 0, 0]), int(np.isfinite(g[1, 0, 0]).sum()), E))
