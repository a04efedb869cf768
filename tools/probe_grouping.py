"""Experiment: does grouping members by cost *pattern* (not only total pilot cost) raise SIMT efficiency?
Members are pre-permuted on the host and run with balance=0 (slot order = given order)."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np, torch
from simplyp_amd import engine, synthetic
eng = engine.get_engine(0)
E = 100000
pr = synthetic.c3_problem(E, solver=dict(out_slot_order=1, balance=0))
mp, rp = pr['member_params'], pr['reach_params']
dmp, drp = eng.to_device(mp), eng.to_device(rp)


def window_cost(start, length):
    cost = torch.zeros(E, dtype=torch.int32, device='cuda')
    f = np.ascontiguousarray(pr['forcing'][:, :, start:start + length]); d = np.ascontiguousarray(pr['doy'][start:start + length])
    pr['opts'].time_chunk_days = -1
    eng.run(f, d, dmp, drp, pr['up_ptr'], pr['up_idx'], pr['opts'], member_rhs=cost)
    pr['opts'].time_chunk_days = 0
    return cost.cpu().numpy().astype(float)


def pcs(feat):
    lf = np.log(feat + 1.0)
    z = (lf - lf.mean(1, keepdims=True)) / (lf.std(1, keepdims=True) + 1e-12)
    u, s, vt = np.linalg.svd(z, full_matrices=False)
    return u.T @ z


o = None
def run(name, order):
    global o
    dm, dr = dmp[:, torch.as_tensor(order, device='cuda')].contiguous(), drp[:, :, torch.as_tensor(order, device='cuda')].contiguous()
    for rep in range(2):
        o, st, stats = eng.run(pr['forcing'], pr['doy'], dm, dr, pr['up_ptr'], pr['up_idx'], pr['opts'], out=o)
    print('%-40s kernel %.1f ms simt %.4f' % (name, stats['kernel_ms'], stats['simt_efficiency']), flush=True)


def hier(total, pc, nbs, final_key):
    """nested blocks: nbs[0] by total, nbs[1] by pc[1], nbs[2] by pc[2], ...; leaves sorted by final_key"""
    order = np.argsort(-total, kind='stable')
    groups = [order]
    for lvl, nb in enumerate(nbs):
        new = []
        for gi, g in enumerate(groups):
            if lvl > 0:
                k = pc[lvl][g] * (1.0 if gi % 2 == 0 else -1.0)
                g = g[np.argsort(k, kind='stable')]
            new += [g[len(g) * i // nb: len(g) * (i + 1) // nb] for i in range(nb)]
        groups = new
    out = []
    for gi, g in enumerate(groups):
        k = final_key[g] * (1.0 if gi % 2 == 0 else -1.0)
        out.append(g[np.argsort(k, kind='stable')])
    return np.concatenate(out)


feat = np.stack([window_cost(80 * k, 8) for k in range(8)])
total = feat.sum(0)
pc = pcs(feat)
run('total', np.argsort(-total, kind='stable'))
run('24/6/pc3 (library)', hier(total, pc, [24, 6], pc[2]))
run('24/4/4/pc4', hier(total, pc, [24, 4, 4], pc[3]))
run('48/4/pc3', hier(total, pc, [48, 4], pc[2]))
run('12/8/pc3', hier(total, pc, [12, 8], pc[2]))
run('24/6/3/total', hier(total, pc, [24, 6, 3], -total))
run('24/6/total', hier(total, pc, [24, 6], -total))
run('96/pc2', hier(total, pc, [96], pc[1]))
run('24/12/pc3', hier(total, pc, [24, 12], pc[2]))
run('8/8/8/pc4', hier(total, pc, [8, 8, 8], pc[3]))
