"""Quick GPU probe: parity vs oracle on the golden scenarios + a throughput number."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'tests'))
import numpy as np
import helpers
from simplyp_amd import engine, marshal
from oracle import oracle

eng = engine.get_engine(0)
for name in ['tarland_2004_static', 'tarland_2004_dynamic', 'confluence3_nc_2004', 'chain4_val_2004']:
    for solver in [dict(integrator='rk4', substeps=16), dict(integrator='cashkarp', rtol=1e-8, atol=1e-10)]:
        m = helpers.marshal_scenario(name, E=70, solver=solver)
        out, status, stats = eng.run(m['forcing'], m['doy'], m['member_params'], m['reach_params'], m['up_ptr'], m['up_idx'], m['opts'])
        got = out.cpu().numpy()
        ref, rstat, rstats = oracle.run(m['forcing'], m['doy'], m['member_params'][:, :1], m['reach_params'][:, :, :1], m['up_ptr'], m['up_idx'], m['opts'])
        errs = [helpers.max_rel_err(got[c, :, :, 0], ref[c, :, :, 0], floor=1e-12) for c in range(25)]
        same = all(np.array_equal(got[..., 0], got[..., k], equal_nan=True) for k in range(1, got.shape[3]))
        n = got.shape[1] * got.shape[2] * got.shape[3]
        print('%-24s %-9s max rel err vs oracle %.2e (col %s) members identical %s | rhs/day gpu %.2f oracle %.2f launches %d status %d'
              % (name, solver['integrator'], max(errs), marshal.OUT_COLUMNS[int(np.argmax(errs))], same,
                 stats['rhs_evals'] / n, rstats['rhs_evals'] / (n / got.shape[3]), stats['n_launches'], int(status.max())), flush=True)

# throughput: Tarland 30 yr
import torch
for E in [1024, 65536, 100000, 131072, 262144]:
    for solver in [dict(integrator='cashkarp', rtol=1e-8, atol=1e-10), dict(integrator='rk4', substeps=16)]:
        m = helpers.marshal_scenario('tarland_1981_2010_dynamic', E=E, solver=solver, out_mask=marshal.MASK_REACH5)
        args = [eng.to_device(m[k]) for k in ('forcing', 'doy', 'member_params', 'reach_params')]
        out = None
        for rep in range(2):
            t0 = time.time()
            out, status, stats = eng.run(args[0], args[1], args[2], args[3], m['up_ptr'], m['up_idx'], m['opts'], out=out)
            torch.cuda.synchronize(); dt = time.time() - t0
        n = E * out.shape[1]
        print('E=%d %s: kernel %.1f ms wall %.1f ms -> %.3e cd/s, rhs/day %.1f, %.1f GB out' %
              (E, solver['integrator'], stats['kernel_ms'], dt * 1e3, n / (stats['kernel_ms'] * 1e-3), stats['rhs_evals'] / n, out.numel() * 8 / 1e9), flush=True)
        del out
