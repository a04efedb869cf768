"""One GPU's share of BASELINE config C5 (1M-member Tarland ensemble over 8 GPUs = 125 000 members per GPU, seed
20240603): fp32 stages + fp64 daily integrals / soil P (integrator 'cashkarp_aug_f32'), output = per-member annual
sums of the four reach fluxes + status, no daily series.  Also checks the annual sums of the first members against
the parity-grade fp64 solver on the same GPU.
Usage: python tools/bench_c5.py [members] [rtol]"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np, torch
from simplyp_amd import engine, marshal, synthetic

E = int(sys.argv[1]) if len(sys.argv) > 1 else 125000
rtol = float(sys.argv[2]) if len(sys.argv) > 2 else 1e-5
FLUXES = ['Qr', 'Msus_kg/day', 'TDP_kg/day', 'PP_kg/day']
mask = marshal.mask_of_columns(FLUXES)
eng = engine.get_engine(0)


def run(n, solver, reps):
    pr = synthetic.c3_problem(n, seed=20240603, solver=solver, out_mask=mask)
    years = pr['met'].index.year.values
    periods, pod = np.unique(years, return_inverse=True)
    pr['opts'].n_periods = len(periods)
    dev = [eng.to_device(pr[k]) for k in ('forcing', 'doy', 'member_params', 'reach_params')]
    pod = eng.to_device(np.ascontiguousarray(pod, dtype=np.int32))
    mos = torch.empty(n, dtype=torch.int32, device='cuda')
    out = None
    for _ in range(reps):
        out, status, st = eng.run(dev[0], dev[1], dev[2], dev[3], pr['up_ptr'], pr['up_idx'], pr['opts'], out=out,
                                  period_of_day=pod, member_of_slot=mos)
    by_member = torch.empty_like(out)
    by_member[..., mos.long()] = out                     # out_slot_order=1: columns are lane slots
    return by_member, status, st, len(pr['doy'])


f32 = dict(integrator='cashkarp_aug_f32', rtol=rtol, atol=rtol * 1e-2, out_slot_order=1)
out, status, st, D = run(E, f32, 2)
cd = float(E) * D
print('C5 shard: E=%d D=%d annual sums %s (%.1f MB out) kernel %.1f ms + pilot %.1f ms, rhs/cd %.1f, simt_eff %.3f, flagged %d '
      '-> %.3e catchment-days/s' % (E, D, tuple(out.shape), out.numel() * 8 / 1e6, st['kernel_ms'], st['pilot_ms'],
                                    st['rhs_evals'] / cd, st['simt_efficiency'], int((status != 0).sum()),
                                    cd / ((st['kernel_ms'] + st['pilot_ms']) * 1e-3)), flush=True)
n = min(E, 4096)
a, sa, _, _ = run(n, f32, 1)
b, sb, _, _ = run(n, dict(out_slot_order=1), 1)
ok = (sa == 0) & (sb == 0)
rel = ((a - b).abs() / b.abs().clamp_min(1e-300))[..., ok]
print('annual sums, fp32 stages (rtol %g) vs fp64 default (rtol 1e-8), %d members: max rel %.2e, p99 %.2e' %
      (rtol, int(ok.sum()), float(rel.max()), float(torch.quantile(rel.flatten()[:4000000].float(), 0.99))), flush=True)
