"""The DEVICE restatements of the reference's scalar functions against in/out vectors recorded from the unmodified reference
(tests/golden/unit_vectors.npz, made by make_golden.py), through the C ABI's simplyp_eval_units: f_x (model.py:23-37) as the
kernels evaluate it -- `gate()` for the end-of-day flows, the fused clamp form inside the right-hand side -- and
discretized_soilP (:39-56) with the clamps of :696-699 and the soil-water concentration of :702-703, including Vs == 0."""

import os

import numpy as np
import pytest

import helpers

pytestmark = pytest.mark.gpu
V = np.load(os.path.join(helpers.GOLDEN, 'unit_vectors.npz'), allow_pickle=False)


def test_device_fx_matches_reference(engine0):
    got = engine0.eval_units('f_x', np.stack([V['fx_x'], V['fx_th']], axis=1))
    for k in range(2):
        # the fused form computes s = fma(x, 1/d, -th/d): one rounding of x/d against (x - th)/d -- 1e-13 of a gate value that
        # is itself O(1); both forms are 0 / 1 exactly outside the zone
        np.testing.assert_allclose(got[:, k], V['fx_y'], rtol=0.0, atol=[1e-14, 2e-12][k])
    outside = (V['fx_y'] == 0.0) | (V['fx_y'] == 1.0)
    assert np.array_equal(got[outside, 0], V['fx_y'][outside])


def test_device_fx_threshold_zero_is_a_step(engine0):
    # the reference divides 0/0 at x == threshold == 0; oracle and kernels define a plain step
    got = engine0.eval_units('f_x', np.array([[0.0, 0.0], [1e-300, 0.0], [-1e-300, 0.0], [3.0, 0.0]]))
    assert got[:, 0].tolist() == [0.0, 1.0, 0.0, 1.0]


def test_device_soilp_matches_reference(engine0):
    """Rows 0-2 have Vs == 0: the reference gets b = inf, TDPs = 0, sorp = 0 (model.py:50), Plab unchanged, and 0/0 = NaN for the
    concentration; the kernel's reciprocals of 0 are NaN, so the case is handled explicitly (soil_p_update) -- ADVICE r3."""
    rows, want = V['sp_in'], V['sp_out']
    got = engine0.eval_units('soilp', rows)
    clamp = lambda a: np.where(0.0 > a, 0.0, a)          # Python's max(a, 0.): a NaN stays NaN (model.py:696-699)
    np.testing.assert_allclose(got[:, 0], clamp(want[:, 0]), rtol=1e-12, atol=0.0, equal_nan=True)
    np.testing.assert_allclose(got[:, 1], clamp(want[:, 1]), rtol=1e-12, atol=0.0, equal_nan=True)
    Vs = rows[:, 7]
    with np.errstate(all='ignore'):
        conc = clamp(want[:, 0]) / Vs                                                           # :702-703
    np.testing.assert_allclose(got[:, 2], conc, rtol=1e-12, atol=0.0, equal_nan=True)
    dry = Vs == 0.0
    assert dry.sum() == 3 and np.array_equal(got[dry, 0], np.zeros(3)) and np.array_equal(got[dry, 1], rows[dry, 9])
    assert np.isnan(got[dry, 2]).all() and np.isfinite(got[~dry]).all()
