#!/usr/bin/env python3
"""Golden vectors for sum_to_waterbody (reference model.py:851-900) -> waterbody_golden.npz.

Runs only in the build container (needs /root/reference).  The UNMODIFIED reference function (loaded the way
make_golden.py loads model.py) is applied to the reach tables the reference itself produced for the 3-reach confluence
scenario (confluence3_nc_2004.npz, recorded by make_golden.py; both the as-shipped and the rtol=atol=1e-12 run), with
three settings of the 'In_final_flux?' column: reaches {1, 3}, {2, 3} and {1, 2, 3}.  The fixture holds the flag
vectors, f_TDP and the returned tables; inputs are the committed reach tables.

Usage:  python tests/golden/make_waterbody_golden.py
"""

import contextlib
import io
import os
import sys

import numpy as np
import pandas as pd

HERE = os.path.dirname(os.path.abspath(__file__))
REPO = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, REPO)
sys.path.insert(0, os.path.join(REPO, 'tests'))
sys.path.insert(0, HERE)

import helpers            # noqa: E402
import make_golden        # noqa: E402


def main():
    mods = make_golden.load_reference()
    model = mods['model']
    name = 'confluence3_nc_2004'
    met, p_struc, p_SU, p_LU, p_SC, p, dyn = helpers.scenario_inputs(name)
    f_TDP = float(p['f_TDP'])
    arrays = {'f_TDP': np.array(f_TDP), 'scenario': np.array(name)}
    flag_sets = {'r13': [1, 0, 1], 'r23': [0, 1, 1], 'r123': [1, 1, 1]}
    for label in ('shipped', 'tight'):
        df_R = helpers.golden_tables(name, label)['R']
        for key, flags in flag_sets.items():
            ps = p_struc.copy()
            ps['In_final_flux?'] = flags
            with contextlib.redirect_stdout(io.StringIO()):
                df = model.sum_to_waterbody(ps, int(p_SU['n_SC']), df_R, f_TDP)
            arrays['%s/%s' % (label, key)] = df.to_numpy(dtype=float)
            arrays['%s/%s/columns' % (label, key)] = np.array(list(df.columns))
            arrays['flags/%s' % key] = np.array(flags)
            print(label, key, df.shape, list(df.columns))
        # one flagged reach: the reference returns None (model.py:895-897)
        ps = p_struc.copy()
        ps['In_final_flux?'] = [0, 0, 1]
        with contextlib.redirect_stdout(io.StringIO()) as buf:
            none = model.sum_to_waterbody(ps, int(p_SU['n_SC']), df_R, f_TDP)
        assert none is None
        arrays['single_reach_stdout'] = np.array(buf.getvalue())
    np.savez_compressed(os.path.join(HERE, 'waterbody_golden.npz'), **arrays)
    print('waterbody_golden.npz written')


if __name__ == '__main__':
    main()
