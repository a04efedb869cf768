"""ctypes mirror of include/simplyp.h (structs, enums, default options)."""

import ctypes as C

ABI_VERSION = 16

INTEG_RK4 = 0
INTEG_CASHKARP = 1
INTEG_CASHKARP_AUG = 2
INTEG_CASHKARP_AUG_F32 = 3
INTEGRATORS = {'rk4': INTEG_RK4, 'cashkarp': INTEG_CASHKARP, 'ck45': INTEG_CASHKARP,
               'cashkarp_aug': INTEG_CASHKARP_AUG, 'cashkarp_aug_f32': INTEG_CASHKARP_AUG_F32}

STATUS_NONFINITE = 1
STATUS_STEPCAP = 2


class Dims(C.Structure):
    _fields_ = [('E', C.c_int32), ('S', C.c_int32), ('D', C.c_int32), ('n_forcing_sets', C.c_int32)]


class Opts(C.Structure):
    _fields_ = [('integrator', C.c_int32), ('substeps', C.c_int32),
                ('rtol', C.c_double), ('atol', C.c_double),
                ('max_steps', C.c_int32), ('dynamic_epc0', C.c_int32), ('dynamic_erod', C.c_int32),
                ('run_mode_cal', C.c_int32), ('sc_qr0', C.c_int32), ('out_mask', C.c_uint32),
                ('step_len', C.c_double), ('project_vr', C.c_int32), ('balance', C.c_int32),
                ('balance_pilot_days', C.c_int32), ('out_slot_order', C.c_int32),
                ('time_chunk_days', C.c_int32), ('n_periods', C.c_int32), ('snow', C.c_int32), ('lanes_per_wave', C.c_int32),
                ('lanes_per_member', C.c_int32), ('stiff_pair', C.c_int32)]


class Stats(C.Structure):
    _fields_ = [('rhs_evals', C.c_uint64), ('steps', C.c_uint64), ('rejected', C.c_uint64),
                ('kernel_ms', C.c_double), ('pilot_ms', C.c_double), ('simt_efficiency', C.c_double), ('n_launches', C.c_int32), ('balanced', C.c_int32),
                ('queued', C.c_int32), ('lanes_per_wave', C.c_int32), ('lanes_per_member', C.c_int32), ('streamed_chunks', C.c_int32),
                ('d2h_tail_ms', C.c_double), ('wall_ms', C.c_double), ('stream_gbs', C.c_double),
                ('queue_waits', C.c_uint64), ('queue_longest_wait_polls', C.c_uint64), ('queue_longest_stall_polls', C.c_uint64),
                ('stiff_pair', C.c_int32), ('reserved0', C.c_int32)]

    def as_dict(self):
        return {k: getattr(self, k) for k, _ in self._fields_ if not k.startswith('reserved')}


class GofInfo(C.Structure):
    """simplyp_gof_info of include/simplyp.h."""
    _fields_ = [('kernel_ms', C.c_double), ('bytes_read', C.c_int64), ('n_q_days', C.c_int32), ('n_chem_days', C.c_int32),
                ('n_chunks_q', C.c_int32), ('n_chunks_chem', C.c_int32)]

    def as_dict(self):
        return {k: getattr(self, k) for k, _ in self._fields_}


class WbInfo(C.Structure):
    """simplyp_wb_info of include/simplyp.h."""
    _fields_ = [('kernel_ms', C.c_double), ('bytes_moved', C.c_int64)]

    def as_dict(self):
        return {k: getattr(self, k) for k, _ in self._fields_}


# SIMPLYP_WB_*: the reference's df_summed columns in its order (model.py:866, :886-888, :842-845)
WB_COLUMNS = ['Q_cumecs', 'Msus_kg/day', 'TDP_kg/day', 'PP_kg/day', 'SS_mgl', 'TDP_mgl', 'PP_mgl',
              'TP_mgl', 'TP_kg/day', 'SRP_mgl', 'SRP_kg/day']
WB_MASK_ALL = (1 << len(WB_COLUMNS)) - 1

GOF_VARS = ['Q', 'SS', 'TDP', 'PP', 'TP', 'SRP']                                      # SIMPLYP_GOF_*
GOF_STATS = ['N obs', 'NSE', 'log NSE', 'r2', 'Bias (%)', 'nRMSD (%)', 'sum_log_sim', 'sum_relsq']   # SIMPLYP_GOFSTAT_*


# Solver settings used when the caller does not choose: Cash-Karp 5(4) with per-thread step
# control on the augmented (transcendental-free) form of the system, at the tolerance that meets
# the <= 1e-6 parity bar against odeint(rtol=atol=1e-12) on every member of a 100 000-member Monte-Carlo
# ensemble with a margin (1e-8 before the step controller learned about the knees
# of the gates, round 2).
DEFAULT_SOLVER = dict(integrator='cashkarp_aug', substeps=8, rtol=1e-7, atol=1e-12, max_steps=4000, project_vr=1,
                      balance=2, balance_pilot_days=0, out_slot_order=0, time_chunk_days=0, lanes_per_wave=0, lanes_per_member=0,
                      stiff_pair=0)


def make_opts(solver=None, dynamic_epc0=False, dynamic_erod=False, run_mode_cal=True, sc_qr0=0,
              out_mask=(1 << 25) - 1, step_len=1.0, n_periods=0, snow=False):
    s = dict(DEFAULT_SOLVER)
    s.update(solver or {})
    integ = s['integrator']
    if isinstance(integ, str):
        integ = INTEGRATORS[integ.lower()]
    o = Opts()
    o.integrator = int(integ)
    o.substeps = int(s['substeps'])
    o.rtol = float(s['rtol'])
    o.atol = float(s['atol'])
    o.max_steps = int(s['max_steps'])
    o.project_vr = int(s['project_vr'])
    o.balance = int(s['balance'])
    o.balance_pilot_days = int(s['balance_pilot_days'])
    o.out_slot_order = int(s['out_slot_order'])
    o.time_chunk_days = int(s['time_chunk_days'])
    o.lanes_per_wave = int(s['lanes_per_wave'])
    o.lanes_per_member = int(s['lanes_per_member'])
    o.stiff_pair = int(s['stiff_pair'])
    o.dynamic_epc0 = int(bool(dynamic_epc0))
    o.dynamic_erod = int(bool(dynamic_erod))
    o.run_mode_cal = int(bool(run_mode_cal))
    o.sc_qr0 = int(sc_qr0)
    o.out_mask = int(out_mask)
    o.step_len = float(step_len)
    o.n_periods = int(n_periods)
    o.snow = 1 if snow else 0
    return o
