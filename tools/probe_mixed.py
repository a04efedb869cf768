"""Would running the most expensive members of a mid-size single-reach ensemble four lanes each BESIDE one-lane waves of the cheap ones
shorten the pass?  (A 50 000-member shard = one rank of a 2-GPU strong-scaling run: one round of one-lane waves, as slow as its most
expensive member's one-lane chain.)  Emulation with the existing kernels: per-member cost from a full run, then the K most expensive
members alone with four lanes each and the rest alone with one lane each -- on the chip the two kernels would run side by side on
disjoint SIMDs, so the pass would take about the longer of the two.   Usage: python tools/probe_mixed.py [members [K ...]]"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np, torch
from simplyp_amd import engine, synthetic

E = int(sys.argv[1]) if len(sys.argv) > 1 else 50000
eng = engine.get_engine(0)
pr = synthetic.c3_problem(E, solver=dict(out_slot_order=1))
rhs = torch.zeros(E, dtype=torch.int32, device='cuda')
out, st, s0 = eng.run(pr['forcing'], pr['doy'], pr['member_params'], pr['reach_params'], pr['up_ptr'], pr['up_idx'], pr['opts'], member_rhs=rhs)
out, st, s0 = eng.run(pr['forcing'], pr['doy'], pr['member_params'], pr['reach_params'], pr['up_ptr'], pr['up_idx'], pr['opts'], member_rhs=rhs)
print('as built: kernel %.1f ms, lanes/member %d, members/wave %d' % (s0['kernel_ms'], s0['lanes_per_member'], s0['lanes_per_wave']), flush=True)
cost = rhs.cpu().numpy().astype(np.int64)
order = np.argsort(-cost, kind='stable')
slots = 1024
for K in [int(k) for k in sys.argv[2:]] or [2000, 4000, 5184]:
    top, rest = order[:K], order[K:]
    res = []
    for sel, team, lanes in ((top, 4, 16), (rest, 1, min(64, -(-len(rest) // max(1, slots - -(-K // 16)))))):
        m = dict(pr, member_params=np.ascontiguousarray(pr['member_params'][:, sel]), reach_params=np.ascontiguousarray(pr['reach_params'][:, :, sel]))
        o = m['opts']
        o.lanes_per_member, o.lanes_per_wave, o.balance, o.time_chunk_days = team, lanes, 0, -1
        for _ in range(2):
            _, _, s = eng.run(m['forcing'], m['doy'], m['member_params'], m['reach_params'], m['up_ptr'], m['up_idx'], o)
        res.append((s['kernel_ms'], -(-len(sel) // lanes), lanes))
    print('K = %5d four-lane members: %.1f ms on %d waves | %d one-lane members, %d per wave: %.1f ms on %d waves | side by side ~ %.1f ms (%d waves)'
          % (K, res[0][0], res[0][1], len(rest), res[1][2], res[1][0], res[1][1], max(res[0][0], res[1][0]), res[0][1] + res[1][1]), flush=True)
