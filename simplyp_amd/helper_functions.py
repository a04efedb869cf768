"""Unit conversions and small helpers.

Mirrors the public names of the reference's
``Current_Release/v0-2A/simplyP/helper_functions.py`` (UC_Q :6, UC_Qinv :19,
UC_C :32, UC_Cinv :46, UC_V :59, lin_interp :77).  All functions work on
scalars, numpy arrays and pandas objects alike.
"""


def UC_Q(Q_mmd, A_catch):
    """mm/day -> m3/day for a catchment of ``A_catch`` km2 (ref hf.py:6-17)."""
    return Q_mmd * 1000 * A_catch


def UC_Qinv(Q_m3s, A_catch):
    """m3/s -> mm/day (ref hf.py:19-30)."""
    return Q_m3s * 86400 / (1000 * A_catch)


def UC_C(C_kgmm, A_catch):
    """kg/mm -> mg/l (ref hf.py:32-44)."""
    return C_kgmm / A_catch


def UC_Cinv(C_mgl, A_catch):
    """mg/l -> kg/mm (ref hf.py:46-57)."""
    return C_mgl * A_catch


def UC_V(V_mm, A_catch, outUnits):
    """mm -> 'm3' or 'l' (ref hf.py:59-72)."""
    factor = {'m3': 10**3, 'l': 10**6}[outUnits]
    return V_mm * factor * A_catch


def lin_interp(x, x0, x1, y0, y1):
    """Linear interpolation between (x0, y0) and (x1, y1) (ref hf.py:77-92)."""
    return y0 + (y1 - y0) * (x - x0) / (x1 - x0)
