"""BUILD CONTAINER ONLY (needs /root/reference): what does the reference's own integrator -- SciPy odeint = LSODA, variable-order
Adams / BDF with automatic switching -- need in right-hand sides per day for a given accuracy on the reference's own system?
Tarland 2004, Dynamic_* = 'y', the unmodified reference driven as in tests/golden/make_golden.py, truth = rtol=atol=1e-12.
Round 3 result: 76 / 94 / 113 / 135 / 158 evaluations per day for 5.0e-6 / 1.0e-6 / 1.4e-7 / 3.1e-8 / 2.8e-9 -- the engine's knee-aware
Cash-Karp takes 84 for 8.0e-8 on the same member: neither a multistep nor an implicit method is the way to fewer evaluations."""
import sys, io, contextlib, numpy as np, scipy.integrate
import os
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'tests', 'golden'))
import make_golden as mg
mods = mg.load_reference()
class Sw(object):
    def __init__(self): self.rtol=None; self.atol=None; self.nfe=0; self.calls=0
    def __call__(self, func, y0, t, args=(), full_output=0, rtol=None, mxstep=0, **kw):
        out = scipy.integrate.odeint(func, y0, t, args=args, full_output=full_output, rtol=self.rtol, atol=self.atol, mxstep=100000, **kw)
        if full_output: self.nfe += int(out[1]['nfe'][-1]); self.calls += 1
        return out
sw = Sw(); mods['model'].odeint = sw
sc = mg.scenarios(False)['tarland_2004_dynamic']
cols = mg.REACH_COLS
def run(rtol, atol):
    sw.rtol, sw.atol, sw.nfe, sw.calls = rtol, atol, 0, 0
    # reuse run_reference's body with our switch
    import copy, pandas as pd, warnings
    model, inputs = mods['model'], mods['inputs']
    p_SU, p, p_LU, p_SC, p_struc = (copy.deepcopy(sc[k]) for k in ('p_SU', 'p', 'p_LU', 'p_SC', 'p_struc'))
    met = sc['met'].copy()
    with contextlib.redirect_stdout(io.StringIO()), warnings.catch_warnings():
        warnings.simplefilter('ignore')
        met = inputs.snow_hydrol_inputs(p['D_snow_0'], p['f_DDSM'], met)
        dyn = pd.Series(dict(sc['dyn'], Dynamic_effluent_inputs='n', Dynamic_terrestrialP_inputs='n'))
        with np.errstate(all='ignore'):
            df_TC, df_R, Kf, od = model.run_simply_p(met, p_struc, p_SU, p_LU, p_SC, p, dyn)
    return df_R[1][cols].to_numpy(dtype=float), sw.nfe / max(sw.calls, 1)
truth, n0 = run(1e-12, 1e-12)
print('truth nfe/day %.1f' % n0, flush=True)
for tol in (1e-6, 1e-7, 1e-8, 1e-9, 1e-10):
    out, n = run(tol, tol * 1e-2)
    rel = np.abs(out - truth) / np.abs(truth)
    print('LSODA rtol %.0e atol %.0e: nfe/day %.1f  max rel err %.2e  p99 %.2e' % (tol, tol * 1e-2, n, rel.max(), np.percentile(rel, 99)), flush=True)
