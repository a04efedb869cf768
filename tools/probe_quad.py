"""One member over four lanes (opts.lanes_per_member = 4) against one lane per member: kernel time of small single-reach
ensembles of the C3 distribution, and whether the tables are equal bit for bit.  Usage: python tools/probe_quad.py [E ...]"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from simplyp_amd import engine, synthetic
eng = engine.get_engine(0)
for E in [int(x) for x in sys.argv[1:]] or [1024, 4096, 12500, 16384, 25000]:
    ref = None
    for team in (1, 4):
        pr = synthetic.c3_problem(E, solver=dict(out_slot_order=1, lanes_per_member=team), replicated=(E == 1024))
        for rep in range(2):
            out, status, st = eng.run(pr['forcing'], pr['doy'], pr['member_params'], pr['reach_params'], pr['up_ptr'], pr['up_idx'], pr['opts'])
        # slot-ordered tables: bring both into member order before comparing
        mos = st['member_of_slot'].long()
        inv = torch.empty_like(mos); inv[mos] = torch.arange(len(mos), device=mos.device)
        tab = out[:, ::97].index_select(-1, inv)          # every 97th day, member order
        same = None if ref is None else bool(torch.equal(tab, ref))
        ref = tab if ref is None else ref
        print("E=%6d lanes/member=%d members/wave=%2d kernel %.1f ms pilot %.1f queued=%d simt_eff=%.3f rhs/cd=%.1f bitwise_equal_to_one_lane=%s"
              % (E, st['lanes_per_member'], st['lanes_per_wave'], st['kernel_ms'], st['pilot_ms'], st['queued'], st['simt_efficiency'],
                 st['rhs_evals'] / (E * pr['forcing'].shape[2]), same), flush=True)
    del ref, out
