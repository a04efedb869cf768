/*
 * simplyp.h -- C ABI of the MI355X SimplyP time-stepping engine (libsimplyp_hip.so).
 *
 * The reference (JoeyYHT/SimplyP, pure Python) has no FFI.  Its seam for this path is the
 * double loop inside run_simply_p() -- `for SC in p['SC_list']` (model.py:365) x
 * `for idx in range(len(met_df))` (model.py:491) around `odeint(ode_f, ...)` (model.py:640).
 * Every entry point below replaces a piece of that loop nest; the Python host in
 * simplyp_amd/ binds them with ctypes (see INTEGRATION.md for the stub).
 *
 * Conventions
 *   - plain C, no torch / HIP types in signatures; all pointers marked "device" are
 *     device-resident (hipMalloc'd, e.g. torch CUDA tensors' data_ptr()), "host" are host.
 *   - ensemble-major SoA: the member index e is always the fastest-varying one.
 *   - every function returns 0 (SIMPLYP_OK) or a negative simplyp_status; no exceptions
 *     cross the boundary; simplyp_last_error() gives the message for the last failure.
 *   - the caller owns every buffer; the library keeps no pointer after a call returns.
 */
#ifndef SIMPLYP_H
#define SIMPLYP_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define SIMPLYP_ABI_VERSION 16

typedef enum {
    SIMPLYP_OK = 0,
    SIMPLYP_ERR_ARG = -1,        /* bad dimension / option / NULL pointer            */
    SIMPLYP_ERR_TOPOLOGY = -2,   /* upstream id >= own id, out of range, ...          */
    SIMPLYP_ERR_DEVICE = -3,     /* HIP runtime error (message in simplyp_last_error) */
    SIMPLYP_ERR_NOMEM = -4
} simplyp_status;

/* ---- per-member parameters: rows of member_params[SIMPLYP_NP_M][E] --------------------
 * Raw reference parameters (sheet 'Constant' -> series p, sheet 'LU' -> frame p_LU); every
 * derived quantity (mu model.py:349, initial conditions :377-459, Kf :449-453) is computed
 * from these inside the kernel prologue, so an ensemble may perturb any of them.          */
enum {
    SIMPLYP_PM_F_QUICK = 0, SIMPLYP_PM_ALPHA, SIMPLYP_PM_FC, SIMPLYP_PM_BETA, SIMPLYP_PM_T_G,
    SIMPLYP_PM_QG_MIN, SIMPLYP_PM_A_Q, SIMPLYP_PM_B_Q, SIMPLYP_PM_QR0_INIT, SIMPLYP_PM_MSOIL_M2,
    SIMPLYP_PM_KF, SIMPLYP_PM_TDPG, SIMPLYP_PM_E_PP, SIMPLYP_PM_E_M, SIMPLYP_PM_K_M,
    SIMPLYP_PM_D_MAXE_SPR, SIMPLYP_PM_D_MAXE_AUT,
    SIMPLYP_PM_T_S_A, SIMPLYP_PM_T_S_S, SIMPLYP_PM_SOILPCONC_A, SIMPLYP_PM_SOILPCONC_S,
    SIMPLYP_PM_P_NETINPUT_A, SIMPLYP_PM_P_NETINPUT_NC, SIMPLYP_PM_EPC0_INIT_A, SIMPLYP_PM_EPC0_INIT_S,
    SIMPLYP_PM_C_COVER_A, SIMPLYP_PM_C_COVER_S, SIMPLYP_PM_C_COVER_IG,
    SIMPLYP_PM_C_MEAS_A, SIMPLYP_PM_C_MEAS_S, SIMPLYP_PM_C_MEAS_IG,
    SIMPLYP_PM_F_DDSM, SIMPLYP_PM_D_SNOW_0,   /* snow module (inputs.py:159-210); read only when opts.snow = 1 */
    SIMPLYP_NP_M
};

/* ---- per-reach parameters: rows of reach_params[SIMPLYP_NP_R][S][E] (sheet 'SC_reach') - */
enum {
    SIMPLYP_PR_A_CATCH = 0, SIMPLYP_PR_F_AR, SIMPLYP_PR_F_IG, SIMPLYP_PR_F_S,
    SIMPLYP_PR_F_NC_AR, SIMPLYP_PR_F_NC_IG, SIMPLYP_PR_F_NC_S, SIMPLYP_PR_F_SPR,
    SIMPLYP_PR_S_AR, SIMPLYP_PR_S_IG, SIMPLYP_PR_S_SN, SIMPLYP_PR_L_REACH, SIMPLYP_PR_S_REACH,
    SIMPLYP_PR_TDPEFF,
    SIMPLYP_NP_R
};

/* ---- output columns, in the reference's own order: 12 ODE results (model.py:737-739)
 * followed by 13 non-ODE results (model.py:721-723, names :743-745).                     */
enum {
    SIMPLYP_OUT_VSA = 0, SIMPLYP_OUT_VSS, SIMPLYP_OUT_VG, SIMPLYP_OUT_VR, SIMPLYP_OUT_QR_END,
    SIMPLYP_OUT_QR, SIMPLYP_OUT_MSUS_END, SIMPLYP_OUT_MSUS_FLUX, SIMPLYP_OUT_TDPR_END,
    SIMPLYP_OUT_TDP_FLUX, SIMPLYP_OUT_PPR_END, SIMPLYP_OUT_PP_FLUX,
    SIMPLYP_OUT_QQ, SIMPLYP_OUT_QSA, SIMPLYP_OUT_QSS, SIMPLYP_OUT_QG, SIMPLYP_OUT_C_COVER_A,
    SIMPLYP_OUT_EPC0_A, SIMPLYP_OUT_EPC0_NC, SIMPLYP_OUT_TDPS_A, SIMPLYP_OUT_PLAB_A,
    SIMPLYP_OUT_CONC_TDPS_A, SIMPLYP_OUT_TDPS_NC, SIMPLYP_OUT_PLAB_NC, SIMPLYP_OUT_CONC_TDPS_NC,
    SIMPLYP_N_OUT_REF,           /* = 25: the columns the reference's loop produces (model.py:644, :721-724)            */
    /* 26th column, only with opts.snow = 1: the member's snow depth at the end of the day -- met_df['D_snow_end']
     * (inputs.py:197-207), which the reference returns as df_TC['D_snow'] (model.py:775-776).  With the snow module run
     * per member inside the kernel it is a per-member series; same arithmetic and order as the host function.           */
    SIMPLYP_OUT_D_SNOW = SIMPLYP_N_OUT_REF,
    SIMPLYP_N_OUT
};
#define SIMPLYP_MASK_ALL    ((uint32_t)((1u << SIMPLYP_N_OUT_REF) - 1u))   /* the reference's 25 columns                */
#define SIMPLYP_MASK_D_SNOW ((uint32_t)(1u << SIMPLYP_OUT_D_SNOW))        /* accepted only together with opts.snow = 1 */
/* the five documented model outputs of a reach (model.py:272-277): Vr, Qr, and the three
 * daily fluxes */
#define SIMPLYP_MASK_REACH5 ((1u << SIMPLYP_OUT_VR) | (1u << SIMPLYP_OUT_QR) | (1u << SIMPLYP_OUT_MSUS_FLUX) | \
                             (1u << SIMPLYP_OUT_TDP_FLUX) | (1u << SIMPLYP_OUT_PP_FLUX))

/* per-member status bits written to member_status[E] */
#define SIMPLYP_STATUS_NONFINITE 1   /* a state became NaN/Inf                               */
#define SIMPLYP_STATUS_STEPCAP   2   /* adaptive solver hit max_steps in some day            */

typedef enum {
    SIMPLYP_INTEG_RK4 = 0,       /* classical RK4, `substeps` equal steps per day            */
    SIMPLYP_INTEG_CASHKARP = 1,  /* Cash-Karp 5(4) embedded pair, per-thread step control, on the reference's
                                    12-variable system as written (ode_f, model.py:58-187)                       */
    SIMPLYP_INTEG_CASHKARP_AUG = 2, /* same pair and step rule on the augmented form of that system: exp(-mu Vs),
                                    Qr**b_Q, Qr**k_M carried as extra states through their own exact ODEs and
                                    re-evaluated every day, Vr taken from its invariant -- no transcendental in
                                    the right-hand side.  Default.  Its step controller knows
                                    the knees of the reference's smooth-step gates (f_x, model.py:23-37: C1 only):
                                    steps are aimed at them, and the error estimate of a step that crosses one
                                    unannounced is not trusted -- parity-grade (<= 1e-6 against odeint at
                                    rtol=atol=1e-12) at rtol 1e-7.                                               */
    SIMPLYP_INTEG_CASHKARP_AUG_F32 = 3 /* scheme 2 with the stage arithmetic in fp32 (BASELINE config C5): 11 float states per
                                    member inside the day's integration; the four daily integrals, the carried state,
                                    labile soil P / soil-water TDP and the day constants stay fp64.  Meant for
                                    rtol ~ 1e-5 (rtol < 1e-6 is refused); not a parity-grade mode.                 */
} simplyp_integrator;

typedef struct {
    int32_t E;               /* ensemble members                                           */
    int32_t S;               /* sub-catchments / reaches (ids 0..S-1 = reference SC 1..S)  */
    int32_t D;               /* days                                                       */
    int32_t n_forcing_sets;  /* >= 1                                                       */
} simplyp_dims;

typedef struct {
    int32_t  integrator;     /* simplyp_integrator                                         */
    int32_t  substeps;       /* RK4: steps per day.  Cash-Karp: first trial step = step_len/substeps */
    double   rtol;           /* Cash-Karp: err_i <= atol + rtol*max(|y_i|,|y_i + h k1_i|)  (schemes 2, 3: the soil boxes
                                measured from field capacity)          */
    double   atol;
    int32_t  max_steps;      /* Cash-Karp: attempted steps per day before SIMPLYP_STATUS_STEPCAP */
    int32_t  dynamic_epc0;   /* dynamic_options['Dynamic_EPC0'] == 'y'  (model.py:600,684) */
    int32_t  dynamic_erod;   /* dynamic_options['Dynamic_erodibility'] == 'y' (model.py:555) */
    int32_t  run_mode_cal;   /* p_SU.run_mode == 'cal' -> Kf calibrated (model.py:449-453) */
    int32_t  sc_qr0;         /* zero-based reach that Qr0_init refers to (p['SC_Qr0']-1, model.py:386) */
    uint32_t out_mask;       /* bit c set -> column c (SIMPLYP_OUT_*) is written; SIMPLYP_MASK_D_SNOW needs opts.snow */
    double   step_len;       /* integration span per day, model.py:345 (default 1.0)       */
    int32_t  project_vr;     /* 1: at each day end reset Vr to the invariant of the reference's own equations,
                                L_reach*Qr^(1-b_Q)/(a_Q*86400) (drift control; 0 = integrate Vr literally) */
    int32_t  balance;        /* member load balancing: 0 off, 1 on, 2 auto (on when the ensemble needs more waves
                                than the chip holds at once, or a reach network that runs through the task queue).  Short pilot
                                runs measure each member's cost in 8 windows of the forcing; lane slots then take members in
                                blocks of decreasing total cost, ordered inside a block by cost pattern.
                                Results are unchanged bit for bit (members are independent).              */
    int32_t  balance_pilot_days;   /* total days of the pilot, split into 8 windows spread over the first ~1.6 years;
                                      0 = default (64)                                                          */
    int32_t  out_slot_order; /* 0: `out` is written in member order (when members were reordered for balance this is
                                a scatter of 8-byte words: correct, but HBM sees ~4x the output bytes as partial-sector
                                writes).  1: `out` is written in lane-slot order, fully coalesced, and the caller gets
                                the member id of every slot in `member_of_slot` (identity when no reordering happened) */
    int32_t  time_chunk_days;/* single-reach ensembles, adaptive integrators: run as (time chunk x 64-member group) tasks
                                pulled by one persistent wave per SIMD, so every SIMD stays busy whatever the members'
                                relative costs.  0 = auto (chunks of 256 days when the ensemble needs more waves than
                                the chip holds; 64 days for a single-reach run whose output is streamed to the host,
                                simplyp_stream_out), > 0 = always, with this chunk length (rounded up to a multiple of 64),
                                < 0 = never.
                                Results are unchanged bit for bit.                                              */
    int32_t  n_periods;      /* time-reduced output: 0 = one output row per day; > 0 = `out` has n_periods rows per column,
                                row p = sum over the days d with period_of_day[d] == p (e.g. calendar years) */
    int32_t  snow;           /* 0: forcing row 0 is the hydrological input P as the reference's snow_hydrol_inputs left it in
                                met_df['P'] (2 rows per set: P, PET).  1: the snow module runs inside the kernel, per member:
                                forcing has 3 rows per set -- Precipitation, PET, T_air (the met file's columns) -- and every
                                member accumulates / melts its own snow pack with its SIMPLYP_PM_F_DDSM and
                                SIMPLYP_PM_D_SNOW_0 (inputs.py:183-208), so an ensemble can perturb the snow parameters
                                without one forcing set per member.  Same arithmetic, same order: P is bit-identical
                                to the host function's.                                                          */
    int32_t  lanes_per_wave; /* member slots per 64-lane wavefront: 0 = auto (64, except that a single-reach ensemble too small
                                to fill the chip's SIMDs with full waves under an adaptive integrator is spread over more,
                                thinner waves: a wave's day costs the attempts of its slowest lane), 1..64 = as given.
                                Results are unchanged bit for bit.                                               */
    int32_t  lanes_per_member;/* lanes of a wavefront that work on one member: 1, or 4 (integrator 2 only) = a member's Cash-Karp
                                attempt spread over a DPP quad -- one lane each for the two soil boxes, the groundwater and the
                                reach; stage sums, error norm and state update three components per lane instead of eleven --
                                which makes an attempt ~1.4 x shorter.  0 = auto: 4 for an ensemble so small
                                (ceil(E / 16) x S <= SIMDs, i.e. E <= 16 384 single-reach members on MI355X) that the run is
                                bound by one member's serial chain of attempts rather than by throughput -- and for
                                single-reach ensembles up to 1.75 x that size (28 672 members), whose quads then share all
                                SIMDs through the task queue where one-lane waves would occupy a third of them --, else 1.
                                lanes_per_wave then counts member slots of 4 lanes (at most 16).  Results are unchanged bit
                                for bit for every member whose status is 0.                                      */
    int32_t  stiff_pair;     /* integrator 2: attempts whose step is bound by Cash-Karp's stability interval (a reach far down a
                                network relaxes at several hundred per day; |h x rate| <= 3.73) are taken by a second, stability-
                                optimised explicit 4(3) pair of the same 6 stages (include/simplyp_controller.h SIMPLYP_STIFF_*),
                                chosen lane by lane and attempt by attempt from the lane's own state.  0 = auto: on for reach
                                networks (S > 1), off for a single reach; > 0 on; < 0 off.  The same switch turns on the
                                damping-aware error weights (SIMPLYP_DAMP_*: the estimate of what a fast reach forgets within a
                                fraction of the day -- its flow, its three masses -- is discounted accordingly).  Same <= 1e-6
                                parity bar; 40 % fewer right-hand sides on BASELINE config C4.                    */
} simplyp_opts;

typedef struct {
    uint64_t rhs_evals;      /* right-hand-side evaluations, all members/reaches/days      */
    uint64_t steps;          /* accepted steps                                             */
    uint64_t rejected;       /* rejected steps (Cash-Karp)                                 */
    double   kernel_ms;      /* device time of the main launches of this run (HIP events on the run's stream) */
    double   pilot_ms;       /* load balancing: pilot launches + host sort of the cost keys (0 when off)      */
    double   simt_efficiency;/* adaptive integrators: lanes that needed the attempt / lanes that executed it (1 = no
                                divergence between the members of a wavefront)                                  */
    int32_t  n_launches;     /* kernel launches issued (one per routing stage)             */
    int32_t  balanced;       /* 1 when the cost-sorted member order was used                */
    int32_t  queued;         /* 1 when the time-chunk task queue kernel ran                  */
    int32_t  lanes_per_wave; /* member slots per wavefront the run used (opts.lanes_per_wave)                     */
    int32_t  lanes_per_member;/* lanes per member the run used (opts.lanes_per_member): 1 or 4                          */
    int32_t  streamed_chunks;/* simplyp_stream_out: time chunks whose device-to-host copy started while the kernel was still
                                running (0 = the table was copied after the last launch)                           */
    double   d2h_tail_ms;    /* simplyp_stream_out: device time between the end of the last launch and the last output
                                byte reaching the host buffer (what the copy added to the run; 0 when not armed)   */
    double   wall_ms;        /* host wall clock from the entry of simplyp_run / simplyp_run_async to the end of simplyp_sync */
    double   stream_gbs;     /* simplyp_stream_out, chunked runs: table bytes / device time from the start of the main launch to
                                the last output byte in the host buffer, GB/s (0 otherwise) -- a diagnostic of the PCIe link */
    uint64_t queue_waits;    /* task-queue kernel: dependency waits (own previous chunk, upstream reach, ring reader) that
                                found their flag not yet raised and had to poll                                        */
    uint64_t queue_longest_wait_polls;  /* the longest of them, in polls (~2 us each)                                   */
    uint64_t queue_longest_stall_polls; /* the longest stretch of polls, inside any such wait, during which NO task of the run
                                completed: what the wait's bound counts (simplyp_sync)                                 */
    int32_t  stiff_pair;     /* 1 when the run used the stability-optimised second pair (opts.stiff_pair resolved to on)  */
    int32_t  reserved0;
} simplyp_stats;

typedef struct simplyp_ctx simplyp_ctx;

int  simplyp_abi_version(void);
int  simplyp_device_count(void);

/* A context is bound to one HIP device and owns a stream, two events and grow-only device
 * scratch (routing series between reaches, launch schedule).  Not re-entrant; distinct
 * contexts may be driven from distinct host threads. */
int  simplyp_ctx_create(int device, simplyp_ctx** out);
void simplyp_ctx_destroy(simplyp_ctx* ctx);
const char* simplyp_last_error(const simplyp_ctx* ctx);   /* ctx may be NULL: create errors */
/* Run on the caller's stream (a hipStream_t passed as void*, e.g. torch.cuda.current_stream()
 * .cuda_stream) so the launches order with the caller's own copies; NULL = a private stream. */
int  simplyp_ctx_set_stream(simplyp_ctx* ctx, void* hip_stream);

/* Bytes of `out` that simplyp_run will write: popcount(out_mask) * (n_periods ? n_periods : D) * n_out_reaches * E * 8. */
int64_t simplyp_out_bytes(const simplyp_dims* dims, const simplyp_opts* opts, int32_t n_out_reaches);

/*
 * simplyp_run -- integrate every (member, reach) through all D days.
 * Replaces model.py:365-724 for the whole ensemble in one call.
 *
 *   forcing           device  [n_forcing_sets][2][D]   row 0 = P (met_df['P'], model.py:497),
 *                                                      row 1 = PET (model.py:498);
 *                             with opts.snow = 1: [n_forcing_sets][3][D], rows Precipitation, PET, T_air
 *   doy               device  [D]        day of year 1..366 (met_df.index[idx].dayofyear, :550)
 *   period_of_day     device  [D] int32 in [0, opts.n_periods), or NULL when opts.n_periods == 0
 *   forcing_of_member device  [E] or NULL (all members use set 0)
 *   member_params     device  [SIMPLYP_NP_M][E]
 *   reach_params      device  [SIMPLYP_NP_R][S][E]
 *   up_ptr, up_idx    host    CSR of directly-upstream reaches (p_struc 'Upstream_SCs',
 *                             model.py:480-487), zero-based, up_idx[k] < own id, shared by all
 *                             members; up_ptr has S+1 entries
 *   out_reaches       host    [n_out_reaches] reaches whose columns are written, or NULL = all S
 *   out               device  [n_cols][D or n_periods][n_out_reaches][E] fp64, n_cols = popcount(out_mask),
 *                             columns in ascending SIMPLYP_OUT_* order
 *   member_status     device  [E] int32, OR of SIMPLYP_STATUS_* bits (zeroed by the call)
 *   member_of_slot    device  [E] int32 or NULL: with opts.out_slot_order = 1, column j of `out` belongs to member
 *                             member_of_slot[j] (required in that mode)
 *   member_rhs_evals  device  [E] uint32 or NULL: right-hand-side evaluations spent on each member, summed
 *                             over its reaches and days (what LSODA's infodict['nfe'] was to the reference's
 *                             caller; also the key the host sorts members by, see simplyp_amd/engine.py)
 *   stats             host    may be NULL
 *
 * The call is synchronous: it returns after the last kernel has finished.
 */
int simplyp_run(simplyp_ctx* ctx, const simplyp_dims* dims, const simplyp_opts* opts,
                const double* forcing, const int32_t* doy, const int32_t* period_of_day,
                const int32_t* forcing_of_member,
                const double* member_params, const double* reach_params,
                const int32_t* up_ptr, const int32_t* up_idx,
                const int32_t* out_reaches, int32_t n_out_reaches,
                double* out, int32_t* member_status, int32_t* member_of_slot, uint32_t* member_rhs_evals,
                simplyp_stats* stats);

/*
 * simplyp_plan -- the routing schedule simplyp_run will use for a reach graph, without touching a
 * device (host-only; also what the CPU tests check).  Reaches are grouped into launches; inside a
 * launch each chain is walked by one thread per member, upstream to downstream.
 *   launch_of_reach, chain_of_reach, pos_in_chain, route_slot : host [S] outputs (any may be NULL);
 *   route_slot[s] = slot of the routing buffer that carries reach s's daily series downstream, -1
 *   when no reach reads it.  n_launches / n_slots: totals.
 */
int simplyp_plan(int32_t S, const int32_t* up_ptr, const int32_t* up_idx,
                 int32_t* n_launches, int32_t* n_slots,
                 int32_t* launch_of_reach, int32_t* chain_of_reach, int32_t* pos_in_chain, int32_t* route_slot);

/* Same as simplyp_run but returns once the main launch is enqueued on the context's stream (for overlap with the
 * caller's own copies); simplyp_sync() waits and fills `stats`.  With load balancing active (opts.balance) the call
 * first BLOCKS for the pilot: pilot launches, a device-to-host copy of the cost table, the member ordering on host
 * threads and the upload of the permutation (stats.pilot_ms, ~12 ms for 100 000 members) happen before it returns. */
int simplyp_run_async(simplyp_ctx* ctx, const simplyp_dims* dims, const simplyp_opts* opts,
                      const double* forcing, const int32_t* doy, const int32_t* period_of_day,
                      const int32_t* forcing_of_member,
                      const double* member_params, const double* reach_params,
                      const int32_t* up_ptr, const int32_t* up_idx,
                      const int32_t* out_reaches, int32_t n_out_reaches,
                      double* out, int32_t* member_status, int32_t* member_of_slot, uint32_t* member_rhs_evals);
/* simplyp_sync fails with SIMPLYP_ERR_DEVICE ("no task completed for ... polls") when a wave of the task-queue kernel waited
 * for a dependency while NO task of the whole run completed for `max_polls` polls (2e7 x ~2 us; environment variable
 * SIMPLYP_QUEUE_MAX_POLLS) -- a run that is merely slow (a shared GPU, a profiler) keeps completing tasks and never fails;
 * results are incomplete after such an error.  On every exit, error or not, the streamed-output machinery is quiesced: the
 * copier thread joined, both copy streams idle, nothing of the library still writes to the host buffer. */
int simplyp_sync(simplyp_ctx* ctx, simplyp_stats* stats);

/*
 * simplyp_stream_out -- deliver the NEXT run's output table to host memory as well (one-shot; NULL disarms).
 * One-shot means: the next simplyp_run / simplyp_run_async consumes the arm whatever its outcome -- also when it is refused
 * for its arguments -- so a buffer the caller has since freed is never written by a later run.
 * The reference produces its 25 values per catchment-day in host memory (model.py:644, :721-724); an ensemble's table is
 * tens of GB, so the copy is overlapped with the computation: adaptive integrators then run through the time-chunk task
 * queue (opts.time_chunk_days = 0 then means 64-day chunks for a single reach, 256 for a network), the wave that finishes the
 * last task of a time chunk raises a flag in pinned host memory, and a host thread of the library enqueues that chunk's rows (one contiguous block per column) on a
 * second HIP stream while later chunks compute.  simplyp_sync / simplyp_run return once the last byte is in `host_out`;
 * stats.d2h_tail_ms is what the copy added after the last launch.  Runs without time chunks (RK4, opts.time_chunk_days < 0,
 * time-reduced rows, runs no longer than one chunk) copy the whole table after the last launch.
 *   host_out    host  same layout and size as `out` (simplyp_out_bytes); pinned memory (simplyp_host_alloc) for the copy
 *                     engine to run at PCIe speed beside the kernel -- pageable memory works, slowly
 *   host_bytes  capacity of host_out, checked against the table size at the run
 * The device table `out` is written as always (it feeds simplyp_gof / simplyp_waterbody).
 */
int simplyp_stream_out(simplyp_ctx* ctx, double* host_out, int64_t host_bytes);

/* Host-pinned staging buffers for callers that do not use torch (hipHostMalloc/hipHostFree). */
void* simplyp_host_alloc(int64_t bytes);
void  simplyp_host_free(void* p);

/* Plain device buffers + copies for callers without a device allocator of their own. */
void* simplyp_device_alloc(simplyp_ctx* ctx, int64_t bytes);
void  simplyp_device_free(simplyp_ctx* ctx, void* p);
int   simplyp_memcpy_h2d(simplyp_ctx* ctx, void* dst, const void* src, int64_t bytes);
int   simplyp_memcpy_d2h(simplyp_ctx* ctx, void* dst, const void* src, int64_t bytes);

/* ---- goodness of fit per member (the reference's goodness_of_fit_stats, visualise_results.py:387-474, for a whole
 * ensemble on the device) ------------------------------------------------------------ */
enum {  /* variables, in the order of stats_var_li (visualise_results.py:400); simulated series = df_R columns
           Q_cumecs, SS_mgl, TDP_mgl, PP_mgl, TP_mgl, SRP_mgl (:413-414; model.py:784-793, :831-847) */
    SIMPLYP_GOF_Q = 0, SIMPLYP_GOF_SS, SIMPLYP_GOF_TDP, SIMPLYP_GOF_PP, SIMPLYP_GOF_TP, SIMPLYP_GOF_SRP,
    SIMPLYP_N_GOF_VARS
};
enum {  /* rows of `gof`: the reference's table columns (:460-461) except Spearman's r (a rank statistic: host only),
           plus the two sums the reference's Gaussian likelihood with sigma = m*sim needs
           (Development/2016/MCMC.ipynb cell 6):
           loglik(m) = -n/2 ln(2 pi) - n ln m - SUM_LOG_SIM - SUM_RELSQ / (2 m^2)                               */
    SIMPLYP_GOFSTAT_N_OBS = 0,    /* non-null observations of the variable in the run period (:428)             */
    SIMPLYP_GOFSTAT_NSE,          /* 1 - sum (obs-sim)^2 / sum (obs-mean obs)^2                         (:441)  */
    SIMPLYP_GOFSTAT_LOG_NSE,      /* the same on natural logs                                            (:442)  */
    SIMPLYP_GOFSTAT_R2,           /* squared Pearson correlation                                         (:446)  */
    SIMPLYP_GOFSTAT_PBIAS,        /* 100 sum (sim-obs) / sum obs                                         (:448)  */
    SIMPLYP_GOFSTAT_NRMSD,        /* 100 mean |sim-obs| / std(obs), ddof 0                               (:449)  */
    SIMPLYP_GOFSTAT_SUM_LOG_SIM,  /* sum ln sim over the paired days                                             */
    SIMPLYP_GOFSTAT_SUM_RELSQ,    /* sum (obs/sim - 1)^2 over the paired days                                    */
    SIMPLYP_N_GOF_STATS
};

typedef struct {
    double  kernel_ms;        /* both kernels, HIP events on the context's stream                                */
    int64_t bytes_read;       /* algorithmic bytes: 8 per member and discharge day + 32 per member and chemistry day */
    int32_t n_q_days;         /* discharge-observation days summed over the output reaches                       */
    int32_t n_chem_days;      /* days with any chemistry observation, summed over the output reaches             */
    int32_t n_chunks_q;       /* slices the discharge-day lists were cut into                                    */
    int32_t n_chunks_chem;    /* slices the chemistry-day lists were cut into                                    */
} simplyp_gof_info;

/*
 * simplyp_gof -- statistics of every member's simulated series against shared observations, from the daily table a
 * previous simplyp_run left on the device.  Variables with 10 or fewer observations get NaN rows (the reference drops
 * them, :430, :453); a day is used when the observation and the simulated value are both non-NaN (:436).
 *
 *   dims            E, S, D as in the run (n_forcing_sets ignored)
 *   out_mask, out_reaches, n_out_reaches   as passed to simplyp_run; the mask must contain Qr, Msus_kg/day,
 *                   TDP_kg/day and PP_kg/day, and the run must have written daily rows (opts.n_periods == 0)
 *   out             device  [popcount(out_mask)][D][n_out_reaches][E]
 *   member_of_slot  device  [E] or NULL (columns of `out` are in member order)
 *   f_tdp           device  [E]: p['f_TDP'] of each member (SRP = f_TDP * TDP, model.py:844); not a run parameter
 *   reach_params    device  as in the run (row SIMPLYP_PR_A_CATCH is read)
 *   obs             HOST    [n_out_reaches][SIMPLYP_N_GOF_VARS][D], NaN = no observation that day
 *   gof             device  [SIMPLYP_N_GOF_STATS][SIMPLYP_N_GOF_VARS][n_out_reaches][E], member order
 *   info            host    may be NULL
 * Synchronous.
 */
int simplyp_gof(simplyp_ctx* ctx, const simplyp_dims* dims, uint32_t out_mask,
                const int32_t* out_reaches, int32_t n_out_reaches,
                const double* out, const int32_t* member_of_slot,
                const double* f_tdp, const double* reach_params,
                const double* obs, double* gof, simplyp_gof_info* info);

/*
 * simplyp_gof_spearman -- Spearman's r (visualise_results.py:444-445: DataFrame.corr(method='spearman'), the Pearson
 * correlation of the average ranks of the paired observed and simulated values) of every member, the one column of the
 * reference's table that simplyp_gof does not produce.  A rank statistic: each member's simulated values on the
 * observation days are ranked among themselves by counting (n^2 compares per member and variable, n <= a few thousand),
 * so this pass costs ~0.1-0.2 s for a 100 000-member table where simplyp_gof costs milliseconds -- call it when wanted.
 * Arguments as for simplyp_gof;
 *   rho   device  [SIMPLYP_N_GOF_VARS][n_out_reaches][E], member order; NaN for variables with 10 or fewer observations and
 *                 for members with a NaN simulated value on an observation day (the reference would rank the remaining
 *                 pairs; such members carry SIMPLYP_STATUS_NONFINITE anyway)
 *   info  kernel_ms = all passes; bytes_read = rows of the compact value table streamed in the counting pass
 */
int simplyp_gof_spearman(simplyp_ctx* ctx, const simplyp_dims* dims, uint32_t out_mask,
                         const int32_t* out_reaches, int32_t n_out_reaches,
                         const double* out, const int32_t* member_of_slot,
                         const double* f_tdp, const double* reach_params,
                         const double* obs, double* rho, simplyp_gof_info* info);

/* ---- receiving waterbody: the reference's sum_to_waterbody (model.py:851-900) for a whole ensemble -------------------- */
enum {  /* columns of the reference's df_summed, in its order: the four summed series (vars_to_sum, :866), the three
           volume-weighted concentrations (:886-888), derived_P_species (:842-845) */
    SIMPLYP_WB_Q_CUMECS = 0, SIMPLYP_WB_MSUS_FLUX, SIMPLYP_WB_TDP_FLUX, SIMPLYP_WB_PP_FLUX,
    SIMPLYP_WB_SS_MGL, SIMPLYP_WB_TDP_MGL, SIMPLYP_WB_PP_MGL,
    SIMPLYP_WB_TP_MGL, SIMPLYP_WB_TP_FLUX, SIMPLYP_WB_SRP_MGL, SIMPLYP_WB_SRP_FLUX,
    SIMPLYP_N_WB
};
#define SIMPLYP_WB_MASK_ALL ((uint32_t)((1u << SIMPLYP_N_WB) - 1u))

typedef struct {
    double  kernel_ms;        /* HIP events on the context's stream                                              */
    int64_t bytes_moved;      /* algorithmic bytes: 32 read per member, day and summed reach + 8 written per member, day
                                 and requested column                                                            */
} simplyp_wb_info;

/*
 * simplyp_waterbody -- sum the daily series of the reaches that flow into the receiving waterbody
 * (p_struc['In_final_flux?'] == 1, model.py:867) into one series per member, from the table a previous simplyp_run left
 * on the device: Q_cumecs (= Qr * A_catch * 1000 / 86400 per reach, :784) and the three daily fluxes added in ascending
 * reach order (DataFrame.sum: NaN counts as 0), concentrations = (flux / Q_cumecs) * 1000/86400, TP and SRP as
 * derived_P_species.  Same operations in the same order as the reference: bit-identical to its restatement in
 * oracle/waterbody.py.  The reference returns nothing for fewer than two flagged reaches (:872, :895); the host wrapper
 * keeps that rule, this entry sums whatever it is given (n_sum >= 1).
 *
 *   dims, out_mask, out_reaches, n_out_reaches, out, member_of_slot   as for simplyp_gof (daily rows; the mask must
 *                   contain Qr and the three fluxes)
 *   f_tdp           device  [E], member order (p['f_TDP'])
 *   reach_params    device  as in the run (row SIMPLYP_PR_A_CATCH is read)
 *   sum_reaches     host    [n_sum] zero-based reach ids, ascending, each one of the table's output reaches; n_sum <= 16
 *   wb_mask         bit c set -> column c (SIMPLYP_WB_*) is written
 *   wb              device  [popcount(wb_mask)][D][E]; member axis in the order of `out`'s (slots when the run wrote
 *                           slot order)
 *   info            host    may be NULL
 * Synchronous.
 */
int simplyp_waterbody(simplyp_ctx* ctx, const simplyp_dims* dims, uint32_t out_mask,
                      const int32_t* out_reaches, int32_t n_out_reaches,
                      const double* out, const int32_t* member_of_slot,
                      const double* f_tdp, const double* reach_params,
                      const int32_t* sum_reaches, int32_t n_sum,
                      uint32_t wb_mask, double* wb, simplyp_wb_info* info);

/*
 * simplyp_gof_waterbody -- simplyp_gof for the summed series: statistics of every member's waterbody series (table
 * written by simplyp_waterbody) against observations taken at the waterbody's inflow.
 *   dims            E, D (S, n_forcing_sets ignored)
 *   wb_mask, wb     as written by simplyp_waterbody; the mask must contain Q_cumecs and the three summed fluxes
 *   obs             HOST  [SIMPLYP_N_GOF_VARS][D], NaN = no observation
 *   gof             device [SIMPLYP_N_GOF_STATS][SIMPLYP_N_GOF_VARS][1][E], member order
 */
int simplyp_gof_waterbody(simplyp_ctx* ctx, const simplyp_dims* dims, uint32_t wb_mask, const double* wb,
                          const int32_t* member_of_slot, const double* f_tdp,
                          const double* obs, double* gof, simplyp_gof_info* info);

/*
 * simplyp_eval_units -- the path's scalar device functions on caller-given arguments, one thread per row: how the tests pin the
 * DEVICE restatements of f_x (model.py:23-37) and discretized_soilP (model.py:39-56 followed by the >= 0 clamps of :696-699 and
 * conc_TDPs = TDPs / Vs of :702-703) to vectors the unmodified reference functions produced (tests/golden/unit_vectors.npz).
 *   which = 0   in  device [n][2]  = x, threshold (reld = 0.01)
 *               out device [n][2]  = f_x as the end-of-day flows evaluate it, f_x as the right-hand side's fused form does
 *   which = 1   in  device [n][10] = P_netInput, A_catch, Kf, Msoil, EPC0, Qs, Qq, Vs, TDPs, Plab
 *               out device [n][3]  = TDPs, Plab (clamped at 0 like :696-697), conc_TDPs
 * Synchronous.  Not on the hot path.
 */
int simplyp_eval_units(simplyp_ctx* ctx, int32_t which, int32_t n, const double* in, double* out);

#ifdef __cplusplus
}
#endif
#endif /* SIMPLYP_H */
