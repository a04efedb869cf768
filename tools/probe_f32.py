"""fp32-stage integrator (config C5 shape): accuracy against the fp64 default on the same GPU, and throughput."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'tests'))
import numpy as np, torch
import helpers
from simplyp_amd import engine, marshal, synthetic, abi
eng = engine.get_engine(0)
print('lib', engine.LIB_PATH)
name = 'tarland_1981_2010_dynamic'
gold = helpers.golden_tables(name, 'tight')['R'][1]
cols = ['Vr', 'Qr_EndOfDay', 'Qr', 'Msus_EndOfDay', 'Msus_kg/day', 'TDPr_EndOfDay', 'TDP_kg/day', 'PPr_EndOfDay', 'PP_kg/day']
for solver in [dict(integrator='cashkarp_aug_f32', rtol=1e-4, atol=1e-6), dict(integrator='cashkarp_aug_f32', rtol=1e-5, atol=1e-7),
               dict(integrator='cashkarp_aug_f32', rtol=3e-6, atol=1e-7), dict(integrator='cashkarp_aug', rtol=1e-5, atol=1e-7)]:
    m = helpers.marshal_scenario(name, E=64, solver=solver)
    out, st, stats = eng.run(m['forcing'], m['doy'], m['member_params'], m['reach_params'], m['up_ptr'], m['up_idx'], m['opts'])
    got = out[..., 0].cpu().numpy()
    errs = {c: float(np.max(np.abs(got[marshal.OUT_COLUMNS.index(c), :, 0] - gold[c].values) / np.abs(gold[c].values))) for c in cols}
    fl = {c: errs[c] for c in ('Qr', 'Msus_kg/day', 'TDP_kg/day', 'PP_kg/day')}
    print(solver, 'rhs/day %.1f status %d max err reach %.2e (daily flows/fluxes %.2e) worst %s' % (stats['rhs_evals'] / (64 * got.shape[1]), int(st.max()), max(errs.values()), max(fl.values()), max(errs, key=errs.get)), flush=True)
for E in (65536, 131072, 1000000 // 8):
    for solver in [dict(integrator='cashkarp_aug_f32', rtol=1e-5, atol=1e-7), dict(integrator='cashkarp_aug', rtol=1e-5, atol=1e-7), None]:
        pr = synthetic.c3_problem(E, solver=dict(solver or {}, out_slot_order=1))
        dev = [eng.to_device(pr[k]) for k in ('forcing', 'doy', 'member_params', 'reach_params')]
        o = None
        for rep in range(2):
            o, st, stats = eng.run(dev[0], dev[1], dev[2], dev[3], pr['up_ptr'], pr['up_idx'], pr['opts'], out=o)
        n = E * o.shape[1]
        print('E=%d %s kernel %.1f ms pilot %.1f queued %d rhs/cd %.1f simt %.3f flagged %d -> %.3e cd/s' % (E, (solver or {'integrator': 'default fp64 rtol 1e-8'})['integrator'], stats['kernel_ms'], stats['pilot_ms'], stats['queued'], stats['rhs_evals'] / n, stats['simt_efficiency'], int((st != 0).sum()), n / stats['kernel_ms'] * 1e3), flush=True)
        del o
