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


def features(kind):
    if kind == 'contig8x20':
        pref = np.stack([window_cost(0, 20 * (k + 1)) for k in range(8)])
        return np.diff(np.concatenate([np.zeros((1, E)), pref]), axis=0)
    if kind == 'spread8x20':
        return np.stack([window_cost(80 * k, 20) for k in range(8)])
    if kind == 'spread16x10':
        return np.stack([window_cost(45 * k, 10) for k in range(16)])
    if kind == 'spread8x20_4y':
        return np.stack([window_cost(170 * k, 20) for k in range(8)])


def pcs(feat):
    lf = np.log(feat + 1.0)
    z = (lf - lf.mean(1, keepdims=True)) / (lf.std(1, keepdims=True) + 1e-12)
    u, s, vt = np.linalg.svd(z, full_matrices=False)
    return u.T @ z


def two_level(total, key2, nb1):
    b1 = np.floor(np.argsort(np.argsort(-total)) / (E / nb1)).astype(int)
    return np.lexsort((key2 * np.where(b1 % 2 == 0, 1.0, -1.0), b1)), b1


def three_level(total, pc, nb1, nb2):
    order2, b1 = two_level(total, pc[1], nb1)
    b2 = np.zeros(E, dtype=int)
    for b in range(nb1):
        idx = np.flatnonzero(b1 == b)
        r = np.argsort(np.argsort(pc[1][idx]))
        b2[idx] = np.floor(r / (len(idx) / nb2)).astype(int)
    snake = np.where((b1 * nb2 + b2) % 2 == 0, 1.0, -1.0)
    return np.lexsort((pc[2] * snake, b2, b1))


o = None
def run(name, order):
    global o
    dm, dr = dmp[:, torch.as_tensor(order, device='cuda')].contiguous(), drp[:, :, torch.as_tensor(order, device='cuda')].contiguous()
    for rep in range(2):
        o, st, stats = eng.run(pr['forcing'], pr['doy'], dm, dr, pr['up_ptr'], pr['up_idx'], pr['opts'], out=o)
    print('%-40s kernel %.1f ms simt %.4f' % (name, stats['kernel_ms'], stats['simt_efficiency']), flush=True)


for kind in ('contig8x20', 'spread8x20', 'spread16x10', 'spread8x20_4y'):
    feat = features(kind)
    total = feat.sum(0)
    pc = pcs(feat)
    run(kind + ' total', np.argsort(-total, kind='stable'))
    for nb1 in (12, 48, 192):
        run(kind + ' total%d/pc2' % nb1, two_level(total, pc[1], nb1)[0])
    run(kind + ' total24/pc2x6/pc3', three_level(total, pc, 24, 6))
    run(kind + ' total12/pc2x12/pc3', three_level(total, pc, 12, 12))
