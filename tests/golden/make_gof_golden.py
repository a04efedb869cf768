#!/usr/bin/env python3
"""Golden goodness-of-fit tables from the *unmodified* reference function
``goodness_of_fit_stats`` (Current_Release/v0-2A/simplyP/visualise_results.py:387-474).

Runs only in the build container (needs /root/reference).  visualise_results.py imports seaborn at module level
(not installed), so the one function is taken out of the module's syntax tree at run time and compiled on its own
with numpy / pandas / os in scope; nothing of its text is written anywhere.  ``.ix`` is shimmed as in make_golden.py.

Inputs: the golden reach tables (df_R of the reference at rtol=atol=1e-12 and as shipped) already in this directory,
and the observations the reference ships (data/Coull_*.xlsx, read by simplyp_amd.xlsx), truncated to the
run period exactly as the reference's read_input_data does (inputs.py:118-152).

Also records, for a few perturbed simulated series (sim * fixed smooth factors; no reference model run needed: the
statistic is a pure function of the two series), the same table -- so the restatement is pinned away from one point.

Output: gof_golden.json.   Usage: python tests/golden/make_gof_golden.py
"""
import ast
import json
import os
import sys

import numpy as np
import pandas as pd

HERE = os.path.dirname(os.path.abspath(__file__))
REPO = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, REPO)
sys.path.insert(0, os.path.join(REPO, 'tests'))
sys.path.insert(0, HERE)
REF_VIS = '/root/reference/Current_Release/v0-2A/simplyP/visualise_results.py'

import helpers                      # noqa: E402
import make_golden                  # noqa: E402  (the .ix shim)


class _Ix(make_golden._Ix):
    """``.ix[i, j]`` with two integers on string-labelled axes (the 2x2 correlation tables, visualise_results.py:445,
    :447) was positional on both axes in the pandas the reference was written for."""

    def _split(self, key):
        obj = self.obj
        if (isinstance(obj, pd.DataFrame) and isinstance(key, tuple) and all(isinstance(k, (int, np.integer)) for k in key)
                and not pd.api.types.is_integer_dtype(obj.index) and not pd.api.types.is_integer_dtype(obj.columns)):
            return ('iloc', key)
        return make_golden._Ix._split(self, key)


def reference_function():
    if not hasattr(np, 'NaN'):
        np.NaN = np.nan
    pd.DataFrame.ix = property(lambda self: _Ix(self))
    pd.Series.ix = property(lambda self: _Ix(self))
    tree = ast.parse(open(REF_VIS).read(), REF_VIS)
    fn = [n for n in tree.body if isinstance(n, ast.FunctionDef) and n.name == 'goodness_of_fit_stats']
    assert len(fn) == 1
    ns = dict(np=np, pd=pd, os=os)
    exec(compile(ast.Module(body=fn, type_ignores=[]), REF_VIS, 'exec'), ns)
    return ns['goodness_of_fit_stats']


def main():
    gof = reference_function()
    p_SU = pd.Series(dict(run_mode='cal', save_stats_csv='n', output_fpath=HERE), dtype=object)
    out = {}
    for name, labels in (('tarland_1981_2010_dynamic', ('tight', 'shipped')), ('tarland_2004_dynamic', ('tight',))):
        info = helpers.meta()[name]['inputs']['p_SU']
        obs = helpers.observations(info['st_dt'], info['end_dt'])
        for label in labels:
            df_R = helpers.golden_tables(name, label)['R']
            for case, f in helpers.gof_case_factors(df_R[1].index).items():
                sim = {1: df_R[1].copy()}
                sim[1]['Q_cumecs'] = sim[1]['Q_cumecs'] * f['Q']
                for c in ('SS_mgl', 'PP_mgl', 'TP_mgl', 'TDP_mgl', 'SRP_mgl'):
                    sim[1][c] = sim[1][c] * f['C']
                tab = gof(p_SU, sim, obs)
                out['%s/%s/%s' % (name, label, case)] = dict(
                    index=list(tab.index), columns=list(tab.columns),
                    values=[[float(v) for v in row] for row in tab.to_numpy(dtype=float)])
                print(name, label, case); print(tab.to_string())
    json.dump(out, open(os.path.join(HERE, 'gof_golden.json'), 'w'), indent=1, sort_keys=True)


if __name__ == '__main__':
    main()
