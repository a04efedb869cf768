"""Time the C3 workload (sorted / unsorted) with whatever library SIMPLYP_HIP_LIB points to."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'tests'))
import numpy as np, torch
from simplyp_amd import engine, marshal, synthetic
eng = engine.get_engine(0)
print('lib:', engine.LIB_PATH)
for E in [int(x) for x in sys.argv[1:]] or [65536, 100000]:
    for replicated in (True, False):
        pr = synthetic.c3_problem(E, replicated=replicated)
        f, doy = eng.to_device(pr['forcing']), eng.to_device(pr['doy'])
        mp, rp = eng.to_device(pr['member_params']), eng.to_device(pr['reach_params'])
        D = f.shape[2]
        out = torch.empty((5, D, 1, E), dtype=torch.float64, device='cuda')
        def run(mp_, rp_, tag):
            for rep in range(2):
                o, st, stats = eng.run(f, doy, mp_, rp_, pr['up_ptr'], pr['up_idx'], pr['opts'], out=out)
            n = E * D
            print('E=%d %-12s %-20s kernel %8.1f ms pilot %5.1f rhs/cd %.1f simt_eff %.3f queued %d -> %.3e cd/s' % (E, 'replicated' if replicated else 'monte-carlo', tag, stats['kernel_ms'], stats['pilot_ms'], stats['rhs_evals'] / n, stats['simt_efficiency'], stats['queued'], n / stats['kernel_ms'] * 1e3), flush=True)
        run(mp, rp, 'as drawn')
        if not replicated:
            Dp = 256
            cnt = torch.zeros(E, dtype=torch.int32, device='cuda')
            outp = torch.empty((5, Dp, 1, E), dtype=torch.float64, device='cuda')
            eng.run(f[:, :, :Dp].contiguous(), doy[:Dp].contiguous(), mp, rp, pr['up_ptr'], pr['up_idx'], pr['opts'], out=outp, member_rhs=cnt)
            idx = torch.argsort(cnt, descending=True, stable=True)
            run(mp[:, idx].contiguous(), rp[:, :, idx].contiguous(), 'pilot-sorted desc')
        del out
