"""sum_to_waterbody (reference model.py:851-900): the oracle restatement and the host drop-in against tables the unmodified
reference function returned (tests/golden/waterbody_golden.npz, written by tests/golden/make_waterbody_golden.py from the
reference's own reach tables of the 3-reach confluence scenario)."""

import os

import numpy as np
import pandas as pd
import pytest

import helpers
import simplyp_amd as sp
from oracle import waterbody as wbo

NAME = 'confluence3_nc_2004'
SETS = ['r13', 'r23', 'r123']


@pytest.fixture(scope='module')
def gold():
    return np.load(os.path.join(helpers.GOLDEN, 'waterbody_golden.npz'), allow_pickle=False)


@pytest.mark.parametrize('label', ['shipped', 'tight'])
@pytest.mark.parametrize('key', SETS)
def test_oracle_reproduces_the_reference_tables_bit_for_bit(gold, label, key):
    met, p_struc, p_SU, p_LU, p_SC, p, dyn = helpers.scenario_inputs(NAME)
    df_R = helpers.golden_tables(NAME, label)['R']
    reaches = [i + 1 for i, f in enumerate(gold['flags/' + key]) if f == 1]
    stack = lambda c: np.stack([df_R[r][c].to_numpy(dtype=float) for r in reaches], axis=1)[:, :, None]
    A = np.array([[float(p_SC.loc['A_catch', r])] for r in reaches])
    got = wbo.sum_to_waterbody(stack('Qr'), stack('Msus_kg/day'), stack('TDP_kg/day'), stack('PP_kg/day'), A, float(gold['f_TDP']))
    assert [str(c) for c in gold['%s/%s/columns' % (label, key)]] == wbo.COLUMNS
    want = gold['%s/%s' % (label, key)].T                                # [11, D]
    # Q_cumecs of a reach is recomputed here from Qr (model.py:784), exactly as the reference's df_R column was
    assert np.array_equal(got[:, :, 0], want)


@pytest.mark.parametrize('key', SETS)
def test_host_drop_in_equals_the_reference(gold, key, capsys):
    met, p_struc, p_SU, p_LU, p_SC, p, dyn = helpers.scenario_inputs(NAME)
    df_R = helpers.golden_tables(NAME, 'tight')['R']
    p_struc = p_struc.copy()
    p_struc['In_final_flux?'] = gold['flags/' + key]
    df = sp.sum_to_waterbody(p_struc, 3, df_R, float(gold['f_TDP']))
    assert list(df.columns) == [str(c) for c in gold['tight/%s/columns' % key]]
    assert np.array_equal(df.to_numpy(dtype=float), gold['tight/' + key])
    assert df.index.equals(df_R[1].index)


def test_one_flagged_reach_returns_none_with_the_reference_message(gold, capsys):
    met, p_struc, p_SU, p_LU, p_SC, p, dyn = helpers.scenario_inputs(NAME)
    df_R = helpers.golden_tables(NAME, 'tight')['R']
    p_struc = p_struc.copy()
    p_struc['In_final_flux?'] = [0, 0, 1]
    assert sp.sum_to_waterbody(p_struc, 3, df_R, 0.7) is None
    assert capsys.readouterr().out == str(gold['single_reach_stdout'])
    p_struc['In_final_flux?'] = [1, 1, 1]
    with pytest.raises(ValueError):
        sp.sum_to_waterbody(p_struc, 2, df_R, 0.7)


def test_nan_counts_as_zero_in_the_sum():
    """DataFrame.sum(axis=1) skips NaN (model.py:880)."""
    q = np.array([[[1.0], [np.nan]], [[2.0], [3.0]]])          # [D=2, R=2, E=1]
    A = np.array([[10.0], [20.0]])
    out = wbo.sum_to_waterbody(q, q * 2, q * 3, q * 4, A, 0.5)
    assert out[0, 0, 0] == 1.0 * 10.0 * 1000 / 86400
    assert out[1, 0, 0] == 2.0 and out[1, 1, 0] == 10.0
