import sys, time, io, contextlib
import os
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'tests'))
import helpers
import simplyp_amd as sp
for name in ('tarland_1981_2010_dynamic', 'tarland_2004_dynamic', 'confluence3_nc_2004'):
    for rep in range(3):
        args = helpers.scenario_inputs(name)
        t0 = time.perf_counter()
        with contextlib.redirect_stdout(io.StringIO()):
            df_TC, df_R, Kf, info = sp.run_simply_p(*args)
        dt = time.perf_counter() - t0
    print('%s: run_simply_p wall %.3f s (3rd call); kernel %.1f ms; lanes/member %s; %d days x %d reaches' % (name, dt, info['kernel_ms'], info.get('lanes_per_member'), len(args[0]), len(df_R)), flush=True)
