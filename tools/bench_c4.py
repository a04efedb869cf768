"""BASELINE config C4 (synthetic 256-reach chain x 4 land-use classes, 50-yr daily, 10k-member ensemble) on one GPU.
Usage: python tools/bench_c4.py [members] [days] [reaches]"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np, torch
from simplyp_amd import engine, marshal, synthetic

E = int(sys.argv[1]) if len(sys.argv) > 1 else 10000
D = int(sys.argv[2]) if len(sys.argv) > 2 else 18262
S = int(sys.argv[3]) if len(sys.argv) > 3 else 256
t0 = time.time()
bal = int(sys.argv[4]) if len(sys.argv) > 4 else 2
pr = synthetic.c4_problem(E, n_reaches=S, n_days=D, solver=dict(balance=bal))
eng = engine.get_engine(0)
dev = [eng.to_device(pr[k]) for k in ('forcing', 'doy', 'member_params', 'reach_params')]
print('problem built in %.1f s: E=%d S=%d D=%d' % (time.time() - t0, E, S, D), flush=True)
t0 = time.time()
out, status, st = eng.run(dev[0], dev[1], dev[2], dev[3], pr['up_ptr'], pr['up_idx'], pr['opts'], out_reaches=pr['out_reaches'])
torch.cuda.synchronize()
wall = time.time() - t0
cd = float(E) * S * D
print('balance %d pilot %.0f ms' % (st['balanced'], st['pilot_ms']), end=' ')
print('queued %d launches %d kernel %.1f ms (wall %.1f s) rhs/cd %.1f simt_eff %.3f flagged %d -> %.3e catchment-days/s' %
      (st['queued'], st['n_launches'], st['kernel_ms'], wall, st['rhs_evals'] / cd, st['simt_efficiency'],
       int((status != 0).sum()), cd / (st['kernel_ms'] * 1e-3)), flush=True)
print('outlet mean Qr %.3f mm/d, finite %s' % (float(out[1].mean()), bool(torch.isfinite(out).all())), flush=True)
