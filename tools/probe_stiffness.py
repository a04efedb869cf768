"""How many RHS evaluations per day do implicit one-step / multistep methods need on a stiff downstream reach?
Single reach of the SimplyP system (literal 8-state form, Tarland parameters), upstream inflow = k x its own local runoff
(k = 0 headwater ... 300 big river), daily-constant forcing, solved day by day (restart each day like the reference).  CPU only (scipy).

Result (this container): at parity-grade tolerances Radau / BDF / LSODA need as many right-hand-side evaluations as the
explicit pairs even on the big-river reach (factor 300: RK45 324, LSODA 250, BDF 250, Radau 311-534 per day) -- the reach
dynamics have to be *resolved* to 1e-8, they are not merely a stability limit.  An implicit scheme is not the way to fewer
evaluations on reach networks."""
import numpy as np, sys, time
from scipy.integrate import solve_ivp
import os
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'tests'))
import helpers
met, *_ = helpers.scenario_inputs('tarland_2004_dynamic')
P = met['P'].values[:120]; PET = met['PET'].values[:120]
fc=290.; f_quick=.02; alpha=1.; beta=.7; T_g=65.; Qg_min=.4; a_Q=.5; b_Q=.42; k_M=2.; T_sA=2.; T_sS=10.; fA=.5; fS=.5
L=10000.; E_sum=50.; mu=-np.log(.01)/fc
def fx(x,th,reld=.01):
    d=th*reld
    if x<th: return 0.
    if x>th+d: return 1.
    s=(x-th)/d; return 3*s*s-2*s**3
def make_rhs(Pd,Ed,QrUS,MUS):
    def f(t,y):
        VsA,VsS,Vg,Vr,Qr,Ms=y
        QsA=(VsA-fc)*fx(VsA,fc)/T_sA; QsS=(VsS-fc)*fx(VsS,fc)/T_sS
        dVsA=Pd*(1-f_quick)-alpha*Ed*(1-np.exp(-mu*VsA))-QsA
        dVsS=Pd*(1-f_quick)-alpha*Ed*(1-np.exp(-mu*VsS))-QsS
        Qg=(1-fx(Vg/T_g,Qg_min))*Qg_min+fx(Vg/T_g,Qg_min)*Vg/T_g
        Qs=fA*QsA+fS*QsS
        dVg=beta*Qs-Qg
        inflow=f_quick*Pd+QrUS+(1-beta)*Qs+Qg
        dVr=inflow-Qr
        dQr=(inflow-Qr)*a_Q*Qr**b_Q*86400/((1-b_Q)*L)
        dMs=E_sum*Qr**k_M+MUS-Ms*Qr/Vr
        return [dVsA,dVsS,dVg,dVr,dQr,dMs]
    return f
for kup in (0., 10., 100., 300.):
    res={}
    for method,rtol in (('DOP853',1e-9),('RK45',1e-8),('Radau',1e-7),('Radau',1e-8),('BDF',1e-8),('LSODA',1e-8)):
        Qr0=1.*86400/(1000*51.7)*(1+kup)
        y=np.array([fc,fc,beta*1.67*T_g,L/(a_Q*Qr0**b_Q*86400)*Qr0,Qr0,0.])
        nfev=njev=0; traj=[]
        t0=time.time()
        for d in range(len(P)):
            QrUS=kup*(0.6+0.02*P[d]); MUS=kup*20.*(0.6+0.02*P[d])**2
            s=solve_ivp(make_rhs(P[d],PET[d],QrUS,MUS),(0,1),y,method=method,rtol=rtol,atol=1e-12)
            y=s.y[:,-1]; nfev+=s.nfev; njev+=getattr(s,'njev',0) or 0; traj.append(y.copy())
        res[(method,rtol)]=(nfev/len(P),njev/len(P),np.array(traj),time.time()-t0)
    ref=res[('DOP853',1e-9)][2]
    print('upstream factor %g: outlet Qr ~ %.1f mm/d'%(kup, ref[:,4].mean()))
    for k,(nf,nj,tr,tm) in res.items():
        err=np.max(np.abs(tr[:,[4,5]]-ref[:,[4,5]])/np.abs(ref[:,[4,5]]).clip(1e-300))
        print('   %-7s rtol %.0e: %.0f rhs/day, %.1f jac/day, max rel err (Qr,Msus) vs DOP853 %.1e  (%.1fs)'%(k[0],k[1],nf,nj,err,tm))
