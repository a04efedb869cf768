"""pandas -> SoA marshalling for the device engine.

Turns the reference's six pandas inputs (``run_simply_p`` arguments,
``Current_Release/v0-2A/simplyP/model.py:193``) into the ensemble-major
arrays ``include/simplyp.h`` describes, and reproduces the host-side prologue
of the reference (model.py:311-361: derived rows, validation, in-place edits).
"""

import numpy as np
import pandas as pd

# Row order of member_params == enum SIMPLYP_PM_* in include/simplyp.h.
# (flat name, source): source is ('p', key) or ('LU', row, column).
PM_SPEC = [
    ('f_quick', ('p', 'f_quick')), ('alpha', ('p', 'alpha')), ('fc', ('p', 'fc')), ('beta', ('p', 'beta')),
    ('T_g', ('p', 'T_g')), ('Qg_min', ('p', 'Qg_min')), ('a_Q', ('p', 'a_Q')), ('b_Q', ('p', 'b_Q')),
    ('Qr0_init', ('p', 'Qr0_init')), ('Msoil_m2', ('p', 'Msoil_m2')), ('Kf', ('p', 'Kf')),
    ('TDPg', ('p', 'TDPg')), ('E_PP', ('p', 'E_PP')), ('E_M', ('p', 'E_M')), ('k_M', ('p', 'k_M')),
    ('d_maxE_spr', ('p', 'd_maxE_spr')), ('d_maxE_aut', ('p', 'd_maxE_aut')),
    ('T_s_A', ('LU', 'T_s', 'A')), ('T_s_S', ('LU', 'T_s', 'S')),
    ('SoilPconc_A', ('LU', 'SoilPconc', 'A')), ('SoilPconc_S', ('LU', 'SoilPconc', 'S')),
    ('P_netInput_A', ('LU', 'P_netInput', 'A')), ('P_netInput_NC', ('LU', 'P_netInput', 'NC')),
    ('EPC0_init_mgl_A', ('LU', 'EPC0_init_mgl', 'A')), ('EPC0_init_mgl_S', ('LU', 'EPC0_init_mgl', 'S')),
    ('C_cover_A', ('LU', 'C_cover', 'A')), ('C_cover_S', ('LU', 'C_cover', 'S')), ('C_cover_IG', ('LU', 'C_cover', 'IG')),
    ('C_measures_A', ('LU', 'C_measures', 'A')), ('C_measures_S', ('LU', 'C_measures', 'S')),
    ('C_measures_IG', ('LU', 'C_measures', 'IG')),
    ('f_DDSM', ('p', 'f_DDSM')), ('D_snow_0', ('p', 'D_snow_0')),      # snow module, read when opts.snow = 1
]
PM_NAMES = [n for n, _ in PM_SPEC]
NP_M = len(PM_NAMES)

# Row order of reach_params == enum SIMPLYP_PR_*.
PR_NAMES = ['A_catch', 'f_Ar', 'f_IG', 'f_S', 'f_NC_Ar', 'f_NC_IG', 'f_NC_S', 'f_spr',
            'S_Ar', 'S_IG', 'S_SN', 'L_reach', 'S_reach', 'TDPeff']
NP_R = len(PR_NAMES)

# Output columns == enum SIMPLYP_OUT_*: reference names, model.py:737-739 and :743-745.
ODE_COLUMNS = ['VsA', 'VsS', 'Vg', 'Vr', 'Qr_EndOfDay', 'Qr', 'Msus_EndOfDay', 'Msus_kg/day',
               'TDPr_EndOfDay', 'TDP_kg/day', 'PPr_EndOfDay', 'PP_kg/day']
NONODE_COLUMNS = ['Qq', 'QsA', 'QsS', 'Qg', 'C_cover_A', 'EPC0_A_kgmm', 'EPC0_NC_kgmm',
                  'TDPs_A_kg', 'P_labile_A_kg', 'conc_TDPs_A_kgmm',
                  'TDPs_NC_kgmm', 'P_labile_NC_kg', 'conc_TDPs_NC_kgmm']
OUT_COLUMNS = ODE_COLUMNS + NONODE_COLUMNS
N_OUT = len(OUT_COLUMNS)            # the reference's 25 columns (SIMPLYP_N_OUT_REF)
MASK_ALL = (1 << N_OUT) - 1
# 26th column (SIMPLYP_OUT_D_SNOW), only when the snow module runs inside the kernel (opts.snow): the member's snow depth at
# the end of the day = met_df['D_snow_end'] (inputs.py:197-207), which the reference returns as df_TC['D_snow'] (model.py:775-776)
SNOW_COLUMN = 'D_snow'
ALL_COLUMNS = OUT_COLUMNS + [SNOW_COLUMN]
MASK_D_SNOW = 1 << ALL_COLUMNS.index(SNOW_COLUMN)
REACH5_COLUMNS = ['Vr', 'Qr', 'Msus_kg/day', 'TDP_kg/day', 'PP_kg/day']     # model.py:272-277
MASK_REACH5 = sum(1 << OUT_COLUMNS.index(c) for c in REACH5_COLUMNS)


def columns_of_mask(mask):
    return [c for i, c in enumerate(ALL_COLUMNS) if (mask >> i) & 1]


def mask_of_columns(cols):
    return sum(1 << ALL_COLUMNS.index(c) for c in cols)


def sc_list(p):
    return [int(x) for x in np.asarray(p['SC_list']).ravel()]


def prologue(p_SU, p_LU, p_SC, p):
    """Host prologue of run_simply_p (model.py:311-361), side effects included.

    Adds rows EPC0_0 / Plab0 / TDPs0 to ``p_LU`` and f_A / f_NC_A / NC_type to
    ``p_SC`` in place, raises the reference's ``ValueError``/``AssertionError``
    cases, and returns the per-sub-catchment NC types.
    """
    p_LU.loc['EPC0_0', :] = 4 * [np.nan]                       # :311-313
    p_LU.loc['Plab0', :] = 4 * [np.nan]
    p_LU.loc['TDPs0', :] = 4 * [np.nan]

    p_SC.loc['f_A'] = p_SC.loc['f_IG'] + p_SC.loc['f_Ar']       # :318
    p_SC.loc['f_NC_A'] = (p_SC.loc['f_Ar'] * p_SC.loc['f_NC_Ar']) + (p_SC.loc['f_NC_IG'] * p_SC.loc['f_IG'])  # :319
    nc_types = {}
    for SC in sc_list(p):                                      # :321-335
        if (p_SC.loc['f_A', SC] + p_SC.loc['f_S', SC]) != 1:
            raise ValueError('Land use proportions do not add to 1 in SC %s' % SC)
        if p_SC.loc['f_NC_A', SC] > 0:
            if p_SC.loc['f_NC_S', SC] > 0:
                raise ValueError("Sub-catchment %s has 2 kinds of newly-converted land;\n\
                only one permitted (Semi-natural or agricultural, agricultural can be both arable & IG)" % SC)
            else:
                NC_type = 'A'
        elif p_SC.loc['f_NC_S', SC] > 0:
            NC_type = 'S'
        else:
            NC_type = 'None'
        nc_types[SC] = NC_type
    # a string row in a float frame: make the columns object-typed first (the reference relies on
    # pandas' silent upcast, which newer pandas deprecates)
    for col in p_SC.columns:
        if p_SC[col].dtype != object:
            p_SC[col] = p_SC[col].astype(object)
    for SC, t in nc_types.items():
        p_SC.loc['NC_type', SC] = t

    for season in ['spr', 'aut']:                              # :355-357
        assert (30 < p['d_maxE_%s' % season] < 335), "'d_maxE_%s' must be between 30 and 335" % season
    return nc_types


def epilogue_mutations(p_SU, p_LU, p_SC, p):
    """The in-place edits the reference's SC loop leaves behind (model.py:409-422, :462-463):
    p_LU rows EPC0_0 / Plab0 / TDPs0 hold the *last* sub-catchment's values, blank TDPeff -> 0."""
    from . import helper_functions as hf
    for SC in sc_list(p):
        A_catch = p_SC.loc['A_catch', SC]
        Msoil = p['Msoil_m2'] * 10**6 * A_catch
        for LU in ['A', 'S']:
            p_LU.loc['EPC0_0', LU] = hf.UC_Cinv(p_LU[LU]['EPC0_init_mgl'], A_catch)
            p_LU.loc['Plab0', LU] = 10**-6 * (p_LU[LU]['SoilPconc'] - p_LU['S']['SoilPconc']) * Msoil
            p_LU.loc['TDPs0', LU] = p_LU[LU]['EPC0_0'] * p['fc'] if LU == 'A' else 0
        if pd.isna(p_SC.loc['TDPeff', SC]):
            p_SC.loc['TDPeff', SC] = 0.


def topology(p_struc, p):
    """CSR of directly-upstream reaches, zero-based (parsing rules of model.py:480-487)."""
    scs = sc_list(p)
    pos = {sc: i for i, sc in enumerate(scs)}
    up_ptr, up_idx, up_lists = [0], [], {}
    for SC in scs:
        cell = p_struc.loc[SC, 'Upstream_SCs']
        if isinstance(cell, str):
            ups = [int(x.strip()) for x in cell.split(',')]
        elif isinstance(cell, (int, np.integer)) or not pd.isna(cell):
            ups = [int(cell)]
        else:
            ups = []
        up_lists[SC] = ups
        for u in ups:
            if u not in pos or pos[u] >= pos[SC]:
                # the reference fails here with KeyError on df_R_dict[upstream_SC] (model.py:524)
                raise KeyError(u)
            up_idx.append(pos[u])
        up_ptr.append(len(up_idx))
    return (np.asarray(up_ptr, dtype=np.int32), np.asarray(up_idx, dtype=np.int32), up_lists)


def member_params(p, p_LU, E=1, overrides=None, alloc=np.empty):
    """[NP_M, E] fp64.  ``overrides``: flat name (see PM_NAMES) -> scalar or array[E].  ``alloc(shape, dtype)``: where
    the array lives (``engine.pinned_empty`` = page-locked host memory, so the upload is an asynchronous DMA)."""
    overrides = overrides or {}
    out = alloc((NP_M, E), np.float64)
    for i, (name, src) in enumerate(PM_SPEC):
        if name in overrides:
            out[i, :] = np.asarray(overrides[name], dtype=np.float64)
            continue
        if src[0] == 'p':
            v = p[src[1]] if src[1] in p.index else np.nan
        else:
            v = p_LU.loc[src[1], src[2]]
        out[i, :] = np.nan if v is None else float(v)
    return out


def reach_params(p_SC, p, E=1, overrides=None, alloc=np.empty):
    """[NP_R, S, E] fp64.  ``overrides``: name (see PR_NAMES) -> array broadcastable to [S, E].  ``alloc``: see
    ``member_params``."""
    overrides = overrides or {}
    scs = sc_list(p)
    S = len(scs)
    out = alloc((NP_R, S, E), np.float64)
    for i, name in enumerate(PR_NAMES):
        if name in overrides:
            out[i] = np.broadcast_to(np.asarray(overrides[name], dtype=np.float64), (S, E))
            continue
        for j, SC in enumerate(scs):
            out[i, j, :] = float(p_SC.loc[name, SC])
    return out


def validate_ensemble(mp, rp, scs):
    """The reference's input checks (model.py:321-335, :355-357) for every member of an ensemble: per-member overrides of
    the land-use fractions or the erosion-window days skip ``prologue`` (which sees the workbook values only), and the
    kernel would integrate such members where the reference raises.  Vectorised over [S, E]; the message names the first
    offending member."""
    r = lambda name: rp[PR_NAMES.index(name)]
    f_A = r('f_IG') + r('f_Ar')                                                                  # :318
    f_NC_A = (r('f_Ar') * r('f_NC_Ar')) + (r('f_NC_IG') * r('f_IG'))                             # :319
    bad = (f_A + r('f_S')) != 1                                                                  # :322 (exact, like the reference)
    if bad.any():
        s, e = np.argwhere(bad)[0]
        raise ValueError('Land use proportions do not add to 1 in SC %s (ensemble member %d)' % (scs[s], e))
    bad = (f_NC_A > 0) & (r('f_NC_S') > 0)                                                       # :325-329
    if bad.any():
        s, e = np.argwhere(bad)[0]
        raise ValueError("Sub-catchment %s has 2 kinds of newly-converted land (ensemble member %d);\n\
                only one permitted (Semi-natural or agricultural, agricultural can be both arable & IG)" % (scs[s], e))
    for season in ['spr', 'aut']:                                                                # :355-357
        d = mp[PM_NAMES.index('d_maxE_%s' % season)]
        assert bool(((30 < d) & (d < 335)).all()), \
            "'d_maxE_%s' must be between 30 and 335 (ensemble member %d)" % (season, int(np.argmax(~((30 < d) & (d < 335)))))


def forcing_arrays(met_df, snow=False, alloc=None):
    """([1, 2, D] fp64 with rows P, PET ; doy[D] int32) from the met dataframe (model.py:497-498, :550).
    ``snow=True``: [1, 3, D] with rows Precipitation, PET, T_air -- the raw met columns the in-kernel snow module
    (opts.snow = 1; reference inputs.py:159-210) turns into each member's own P."""
    rows = ['Precipitation', 'PET', 'T_air'] if snow else ['P', 'PET']
    f = np.stack([met_df[c].to_numpy(dtype=np.float64) for c in rows])[None]
    doy = np.asarray(met_df.index.dayofyear, dtype=np.int32)
    if alloc is not None:
        fa, da = alloc(f.shape, np.float64), alloc(doy.shape, np.int32)
        fa[...] = f
        da[...] = doy
        return fa, da
    return np.ascontiguousarray(f), np.ascontiguousarray(doy)


def split_member_reach_overrides(overrides):
    m, r = {}, {}
    for k, v in (overrides or {}).items():
        if k in PM_NAMES:
            m[k] = v
        elif k in PR_NAMES:
            r[k] = v
        else:
            raise KeyError("unknown parameter %r (member: %s; reach: %s)" % (k, PM_NAMES, PR_NAMES))
    return m, r
