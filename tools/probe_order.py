"""Would a higher-order explicit pair need fewer right-hand-side evaluations than the 5(4) pair at parity-grade accuracy?
Single Tarland reach (literal 6-state form + the daily flow / sediment integrals), day by day with restarts like the
reference; scipy's RK45 (Dormand-Prince 5(4), 6 new evaluations per step) against DOP853 (8(5,3), 12 per step) over a
tolerance sweep; error = max relative error of the daily means of Qr and the daily sediment flux against DOP853 at
rtol = atol-scaled 1e-13.  The right-hand side is only C1 at the smooth-step gates.  CPU only (scipy).
Usage: python tools/probe_order.py [days]"""
import numpy as np, sys, time, os
from scipy.integrate import solve_ivp
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'tests'))
import helpers
nd = int(sys.argv[1]) if len(sys.argv) > 1 else 366
met, *_ = helpers.scenario_inputs('tarland_2004_dynamic')
P = met['P'].values[:nd]; PET = met['PET'].values[:nd]
fc=290.; f_quick=.02; alpha=1.; beta=.7; T_g=65.; Qg_min=.4; a_Q=.5; b_Q=.42; k_M=2.; T_sA=2.; T_sS=10.; fA=.5; fS=.5
L=10000.; E_sum=50.; mu=-np.log(.01)/fc
def fx(x,th,reld=.01):
    d=th*reld
    if x<th: return 0.
    if x>th+d: return 1.
    s=(x-th)/d; return 3*s*s-2*s**3
def make_rhs(Pd,Ed):
    def f(t,y):
        VsA,VsS,Vg,Vr,Qr,Ms,_,_=y
        QsA=(VsA-fc)*fx(VsA,fc)/T_sA; QsS=(VsS-fc)*fx(VsS,fc)/T_sS
        dVsA=Pd*(1-f_quick)-alpha*Ed*(1-np.exp(-mu*VsA))-QsA
        dVsS=Pd*(1-f_quick)-alpha*Ed*(1-np.exp(-mu*VsS))-QsS
        Qg=(1-fx(Vg/T_g,Qg_min))*Qg_min+fx(Vg/T_g,Qg_min)*Vg/T_g
        Qs=fA*QsA+fS*QsS
        inflow=f_quick*Pd+(1-beta)*Qs+Qg
        out=Ms*Qr/Vr
        return [dVsA,dVsS,beta*Qs-Qg,inflow-Qr,(inflow-Qr)*a_Q*Qr**b_Q*86400/((1-b_Q)*L),E_sum*Qr**k_M-out,Qr,out]
    return f
def run(method, rtol, atol=1e-12):
    Qr0=1.*86400/(1000*51.7)
    y=np.array([fc,fc,beta*Qr0*T_g,L/(a_Q*Qr0**b_Q*86400)*Qr0,Qr0,0.,0.,0.])
    nfev=0; daily=[]; h0=None
    for d in range(nd):
        y[6]=0.; y[7]=0.
        s=solve_ivp(make_rhs(P[d],PET[d]),(0,1),y,method=method,rtol=rtol,atol=atol,first_step=h0)
        y=s.y[:,-1].copy(); nfev+=s.nfev; daily.append(y[[6,7,4,5]].copy())
        h0=min(1.0, s.t[-1]-s.t[-2]) if len(s.t)>1 else None
    return nfev/nd, np.array(daily)
t0=time.time()
_, ref = run('DOP853', 1e-13, 1e-16)
print('reference: DOP853 rtol 1e-13 (%.0f s)' % (time.time()-t0), flush=True)
for method, tols in (('RK45', (1e-6, 1e-7, 1e-8, 1e-9)), ('DOP853', (1e-5, 1e-6, 1e-7, 1e-8, 1e-9))):
    for rtol in tols:
        nf, tr = run(method, rtol)
        e = np.abs(tr - ref) / np.abs(ref).clip(1e-300)
        print('%-7s rtol %.0e: %6.1f rhs/day   max rel err: daily Qr %.1e, daily sediment flux %.1e, end-of-day Qr %.1e Msus %.1e'
              % (method, rtol, nf, e[:,0].max(), e[1:,1].max(), e[:,2].max(), e[1:,3].max()), flush=True)
