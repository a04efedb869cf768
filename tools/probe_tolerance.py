"""Error of the working tolerances across the bench's Monte-Carlo ensemble: each member's max relative error over the
REACH-5 daily columns (30 years) against the same kernel at rtol 1e-11 / atol 1e-13 (which the goldens pin to the
reference's tight solution to 1e-9).  SIMPLYP_PROBE_SEED=<int> draws the ensemble with another seed (held-out draws: the bench's
ranks > 0 run C3_SEED + rank); SIMPLYP_PROBE_SNOW=1 perturbs f_DDSM / D_snow_0 too and runs the in-kernel snow module; SIMPLYP_PROBE_PSCALE=s multiplies the
precipitation by s and divides PET by s (a wetter / drier climate than Tarland's); SIMPLYP_PROBE_WIDE=w widens the draws of the time
constants and rates (T_s, T_g, a_Q, E_M, f_quick) by a further log-uniform factor in [1/w, w].
Usage: python tools/probe_tolerance.py [members [rtol ...]]"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np, torch
from simplyp_amd import engine, synthetic

E = int(sys.argv[1]) if len(sys.argv) > 1 else 8192
SEED = int(os.environ.get('SIMPLYP_PROBE_SEED', synthetic.C3_SEED))
SNOW = os.environ.get('SIMPLYP_PROBE_SNOW') == '1'
PSCALE = float(os.environ.get('SIMPLYP_PROBE_PSCALE', '1'))          # precipitation x this, PET / this: another climate
TSHIFT = float(os.environ.get('SIMPLYP_PROBE_TSHIFT', '0'))         # with SIMPLYP_PROBE_SNOW: air temperature + this many degrees (a colder / warmer winter)
WIDE = float(os.environ.get('SIMPLYP_PROBE_WIDE', '1'))              # every drawn time constant / rate x a further log-uniform factor in [1/WIDE, WIDE]
eng = engine.get_engine(0)
print('seed', SEED, ('snow, T_air %+g' % TSHIFT) if SNOW else '', 'wide x/%g' % WIDE if WIDE != 1 else '', 'P x %g, PET / %g' % (PSCALE, PSCALE) if PSCALE != 1 else '', flush=True)


def run(solver):
    pr = synthetic.c3_problem(E, seed=SEED, solver=dict(solver, balance=0, time_chunk_days=-1))
    if WIDE != 1:      # a wider parameter distribution than BASELINE C3's: soil, groundwater and reach time scales (synthetic.widen_overrides:
        from simplyp_amd import marshal      # the same draw the dry-reach fixture of tests/golden/make_golden.py uses)
        rng = np.random.default_rng(SEED + 2000)
        for name in synthetic.WIDE_NAMES:
            pr['member_params'][marshal.PM_NAMES.index(name)] *= np.exp(rng.uniform(-np.log(WIDE), np.log(WIDE), E))
    if PSCALE != 1 and not SNOW:
        pr['forcing'] = pr['forcing'].copy()
        pr['forcing'][:, 0] *= PSCALE
        pr['forcing'][:, 1] /= PSCALE
    if SNOW:      # the snow module per member inside the kernel, snow parameters drawn per member
        from simplyp_amd import marshal
        met = pr['met']
        pr['forcing'], pr['doy'] = marshal.forcing_arrays(met, snow=True)
        pr['forcing'] = pr['forcing'].copy()
        pr['forcing'][:, 2] += TSHIFT
        if PSCALE != 1:
            pr['forcing'][:, 0] *= PSCALE
            pr['forcing'][:, 1] /= PSCALE
        pr['opts'].snow = 1
        rng = np.random.default_rng(SEED + 1000)
        pr['member_params'][marshal.PM_NAMES.index('f_DDSM')] = rng.uniform(1.0, 5.0, E)
        pr['member_params'][marshal.PM_NAMES.index('D_snow_0')] = rng.uniform(0.0, 30.0, E)
    out, st, stats = eng.run(pr['forcing'], pr['doy'], pr['member_params'], pr['reach_params'], pr['up_ptr'], pr['up_idx'], pr['opts'])
    return out, st, stats


truth, st, _ = run(dict(rtol=1e-11, atol=1e-13))
print('truth flagged', int((st != 0).sum()), flush=True)
print('lib', engine.LIB_PATH)
rtols = [float(x) for x in sys.argv[2:]] or [1e-7]
for rtol, atol in [(r, 1e-12) for r in rtols]:
    out, st, stats = run(dict(rtol=rtol, atol=atol))
    rel = (out - truth).abs_().div_(truth.abs().clamp_min(1e-300))
    per_member = rel.amax(dim=(0, 1, 2))                      # [E]
    per_col = rel.amax(dim=(1, 2, 3))
    q = torch.quantile(per_member.float(), torch.tensor([0.5, 0.99, 0.999], device=per_member.device))
    print('rtol %.1e atol %.0e: rhs/day %.1f kernel %.1f ms | member max-rel-err: median %.2e p99 %.2e p99.9 %.2e max %.2e | members > 5e-7: %d > 1e-6: %d | per column %s'
          % (rtol, atol, stats['rhs_evals'] / (E * out.shape[1]), stats['kernel_ms'], q[0], q[1], q[2], per_member.max(), int((per_member > 5e-7).sum()), int((per_member > 1e-6).sum()),
             ['%.1e' % v for v in per_col.tolist()]), flush=True)
    worst = int(per_member.argmax())
    flat = rel[..., worst].flatten().argmax()
    c, d = int(flat // rel.shape[1]), int(flat % rel.shape[1])
    print('   worst member %d: column %d day %d value %.3e truth %.3e flagged %d' % (worst, c, d, float(out[c, d, 0, worst]), float(truth[c, d, 0, worst]), int(st[worst])), flush=True)
    del out, rel
