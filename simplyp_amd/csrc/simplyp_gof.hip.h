// simplyp_gof.hip.h -- per-member goodness-of-fit reduction over the daily reach outputs (gfx950).
//
// What it replaces: goodness_of_fit_stats of the reference (visualise_results.py:387-474) applied to every member of an
// ensemble -- the one scalar table per member that the reference's calibration notebooks consume -- without the
// daily series ever leaving HBM.  Simulated series per day from the four reach outputs (model.py:784-793, :831-847):
//   Q = Qr*A*1000/86400;  SS,TDP,PP = (flux/Qr)/A;  TP = TDP+PP;  SRP = f_TDP*TDP.
//
// Layout: lane = member slot (the `out` table's fastest axis), so every wave load is one 512-byte row segment.
// Observations are shared by all members: the host compacts them into two day lists per reach (days with a discharge
// observation -> only the Qr row is read; days with any chemistry observation -> Qr and the three flux rows), which the
// wave reads through the scalar cache.  HBM-bound: 8 or 32 bytes per member and observation day against ~15 / ~130
// fp64 operations.  The day lists are cut into `n_chunks` slices (blockIdx.y) so that a 100 000-member ensemble puts
// several thousand waves in flight; partial sums go through a [chunk][reach][78][E] scratch table and a second kernel
// adds them in chunk order (deterministic, no atomics) and finishes the statistics.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "../../include/simplyp.h"
#include "simplyp_kernels.hip.h"      // sp_log, sp_rcp

namespace simplyp {

constexpr int GOF_NACC = 13;           // running sums per variable
constexpr int GOF_NV = SIMPLYP_N_GOF_VARS;
constexpr int GOF_BATCH_Q = 8;         // discharge days whose Qr rows are loaded together
constexpr int GOF_BATCH_C = 4;         // chemistry days (4 rows each) loaded together

struct GofArgs {
    int E, R, D;                       // members, output reaches, days
    const double* out;                 // [n_cols][D][R][E]
    long long col_stride;              // D*R*E
    int col[4];                        // column slots of Qr, Msus_kg/day, TDP_kg/day, PP_kg/day in `out`
    const int32_t* member_of_slot;     // [E] or nullptr
    const double* f_tdp;               // member_params row [E]
    const double* a_catch;             // reach_params row [S][E]; nullptr = the table is a waterbody series (simplyp_waterbody):
                                       // column 0 holds Q_cumecs, so Q = column 0 and conc = (flux / Q) * 1000/86400
    const int32_t* reach_of;           // [R] reach index of output reach r
    const int32_t* q_ptr;              // [R+1] offsets into q_day / q_obs
    const int32_t* q_day;              // days with a discharge observation
    const double* q_obs;               // [Kq][2]: obs, ln obs
    const int32_t* c_ptr;              // [R+1] offsets into c_day / c_obs
    const int32_t* c_day;              // days with any chemistry observation
    const double* c_obs;               // [Kc][10]: obs of SS,TDP,PP,TP,SRP (NaN = none), then their logs
    const double* shift;               // [R][12]: per variable mean obs, mean ln obs (conditioning shifts)
    const double* n_obs;               // [R][6] observation counts (0 when the variable is dropped)
    int n_chunks_q, n_chunks_c;        // slices of the discharge / chemistry day lists
    double* partial;                   // [max(n_chunks_q, n_chunks_c)][R][6*13][E]
    double* gof;                       // [SIMPLYP_N_GOF_STATS][6][R][E], member order
};

struct GofAcc {
    double a[GOF_NACC];
};

// One (obs, sim) pair of one variable.  o, lo: observation and its log (wave-uniform); s: simulated value (per lane);
// co, clo: shifts.  A NaN simulated value drops the pair (pandas dropna, visualise_results.py:436); +-inf stays.
__device__ __forceinline__ void gof_add(GofAcc& A, double o, double lo, double s, double co, double clo)
{
    if (s != s) return;
    double ls, q;
    if (s > 1e-290 && s < 1e290) {          // every ordinary value: own log / reciprocal (~1 ulp, a third of libm's cost)
        ls = sp_log(s);
        q = o * sp_rcp(s) - 1.0;
    } else {                                // 0, negative, infinite: IEEE results as numpy gives them (-inf, nan, ...)
        ls = log(s);
        q = o / s - 1.0;
    }
    const double op = o - co, sp = s - co, lop = lo - clo;
    const double d = o - s, dl = lo - ls;
    A.a[0] += 1.0;
    A.a[1] += op;
    A.a[2] = __builtin_fma(op, op, A.a[2]);       // explicit: the library is built with -ffp-contract=off
    A.a[3] += sp;
    A.a[4] = __builtin_fma(sp, sp, A.a[4]);
    A.a[5] = __builtin_fma(op, sp, A.a[5]);
    A.a[6] += fabs(d);
    A.a[7] = __builtin_fma(d, d, A.a[7]);
    A.a[8] += lop;
    A.a[9] = __builtin_fma(lop, lop, A.a[9]);
    if (dl == dl) A.a[10] = __builtin_fma(dl, dl, A.a[10]);      // np.sum of a pandas Series skips NaN: log(sim < 0) drops out of :442
    A.a[11] += ls;
    A.a[12] = __builtin_fma(q, q, A.a[12]);
}

// One chemistry day: the five simulated concentrations and their pairs.  ob: 5 observations then their logs (wave-uniform,
// NaN = none); sh: the reach's 12 shifts.
__device__ __forceinline__ void gof_chem_day(GofAcc (&acc)[GOF_NV], const double* ob, const double* sh, double qr, double ms,
                                             double td, double pp, double A, double f_tdp)
{
    double SS, TDP, PP;
    if (qr > 1e-290 && qr < 1e290) {          // (flux/Qr)/A with one reciprocal (2 ulp); literal divisions otherwise
        const double inv = sp_rcp(qr * A);
        SS = ms * inv; TDP = td * inv; PP = pp * inv;
    } else {
        SS = (ms / qr) / A; TDP = (td / qr) / A; PP = (pp / qr) / A;
    }
    const double TP = TDP + PP, SRP = TDP * f_tdp;
    if (ob[0] == ob[0]) gof_add(acc[SIMPLYP_GOF_SS], ob[0], ob[5], SS, sh[1], sh[7]);
    if (ob[1] == ob[1]) gof_add(acc[SIMPLYP_GOF_TDP], ob[1], ob[6], TDP, sh[2], sh[8]);
    if (ob[2] == ob[2]) gof_add(acc[SIMPLYP_GOF_PP], ob[2], ob[7], PP, sh[3], sh[9]);
    if (ob[3] == ob[3]) gof_add(acc[SIMPLYP_GOF_TP], ob[3], ob[8], TP, sh[4], sh[10]);
    if (ob[4] == ob[4]) gof_add(acc[SIMPLYP_GOF_SRP], ob[4], ob[9], SRP, sh[5], sh[11]);
}

// PART 0: discharge days (reads the Qr row only, 13 running sums); PART 1: chemistry days (Qr + three flux rows, 65
// running sums).  Two instantiations so that the discharge pass -- most of the bytes -- is not held to the register
// budget of the chemistry pass.
template <int PART>
__global__ __launch_bounds__(64) void simplyp_gof_partial_kernel(const GofArgs g)
{
    const int slot = blockIdx.x * 64 + threadIdx.x;
    const int chunk = blockIdx.y, r = blockIdx.z;
    if (slot >= g.E) return;
    const int member = g.member_of_slot ? g.member_of_slot[slot] : slot;
    // waterbody series: (flux / Q_cumecs) * (1000/86400) = flux / (Q_cumecs * 86.4), i.e. the reach formulas with A = 86.4
    // for the concentrations and a unit discharge scale
    const double A = g.a_catch ? g.a_catch[(size_t)g.reach_of[r] * g.E + member] : 86.4;
    const size_t day_stride = (size_t)g.R * g.E;
    const double* qr_col = g.out + (size_t)g.col[0] * g.col_stride + (size_t)r * g.E + slot;
    const double* sh = g.shift + (size_t)r * 12;
    double* p = g.partial + ((size_t)chunk * g.R + r) * (GOF_NV * GOF_NACC) * g.E + slot;

    if constexpr (PART == 0) {
        GofAcc acc;
#pragma unroll
        for (int j = 0; j < GOF_NACC; ++j) acc.a[j] = 0.0;
        const long long k0 = g.q_ptr[r], n = g.q_ptr[r + 1] - k0;
        const int kb = (int)(k0 + n * chunk / g.n_chunks_q), ke = (int)(k0 + n * (chunk + 1) / g.n_chunks_q);
        const double co = sh[0], clo = sh[6];
        const double q_scale = g.a_catch ? A * 1000 / 86400 : 1.0;      // Q = Qr*A*1000/86400 (model.py:784) with the constants folded: 2 ulp
        // batches of GOF_BATCH_Q days: all row loads are issued before the first is consumed (one wave keeps 8 x 512 B
        // in flight; a load-use pair per iteration leaves HBM waiting on latency)
        int k = kb;
        for (; k + GOF_BATCH_Q <= ke; k += GOF_BATCH_Q) {
            double qv[GOF_BATCH_Q];
#pragma unroll
            for (int j = 0; j < GOF_BATCH_Q; ++j) qv[j] = qr_col[(size_t)g.q_day[k + j] * day_stride];
#pragma unroll
            for (int j = 0; j < GOF_BATCH_Q; ++j)
                gof_add(acc, g.q_obs[2 * (k + j)], g.q_obs[2 * (k + j) + 1], qv[j] * q_scale, co, clo);
        }
        for (; k < ke; ++k)
            gof_add(acc, g.q_obs[2 * k], g.q_obs[2 * k + 1], qr_col[(size_t)g.q_day[k] * day_stride] * q_scale, co, clo);
#pragma unroll
        for (int j = 0; j < GOF_NACC; ++j) p[(size_t)(SIMPLYP_GOF_Q * GOF_NACC + j) * g.E] = acc.a[j];
    } else {
        const double f_tdp = g.f_tdp[member];
        const double* ms_col = g.out + (size_t)g.col[1] * g.col_stride + (size_t)r * g.E + slot;
        const double* td_col = g.out + (size_t)g.col[2] * g.col_stride + (size_t)r * g.E + slot;
        const double* pp_col = g.out + (size_t)g.col[3] * g.col_stride + (size_t)r * g.E + slot;
        GofAcc acc[GOF_NV];
#pragma unroll
        for (int v = 1; v < GOF_NV; ++v)
#pragma unroll
            for (int j = 0; j < GOF_NACC; ++j) acc[v].a[j] = 0.0;
        const long long k0 = g.c_ptr[r], n = g.c_ptr[r + 1] - k0;
        const int kb = (int)(k0 + n * chunk / g.n_chunks_c), ke = (int)(k0 + n * (chunk + 1) / g.n_chunks_c);
        int k = kb;
        for (; k + GOF_BATCH_C <= ke; k += GOF_BATCH_C) {
            double qv[GOF_BATCH_C], mv[GOF_BATCH_C], tv[GOF_BATCH_C], pv[GOF_BATCH_C];
#pragma unroll
            for (int j = 0; j < GOF_BATCH_C; ++j) {
                const size_t off = (size_t)g.c_day[k + j] * day_stride;
                qv[j] = qr_col[off]; mv[j] = ms_col[off]; tv[j] = td_col[off]; pv[j] = pp_col[off];
            }
#pragma unroll
            for (int j = 0; j < GOF_BATCH_C; ++j)
                gof_chem_day(acc, g.c_obs + (size_t)(k + j) * 10, sh, qv[j], mv[j], tv[j], pv[j], A, f_tdp);
        }
        for (; k < ke; ++k) {
            const size_t off = (size_t)g.c_day[k] * day_stride;
            gof_chem_day(acc, g.c_obs + (size_t)k * 10, sh, qr_col[off], ms_col[off], td_col[off], pp_col[off], A, f_tdp);
        }
#pragma unroll
        for (int v = 1; v < GOF_NV; ++v)
#pragma unroll
            for (int j = 0; j < GOF_NACC; ++j) p[(size_t)(v * GOF_NACC + j) * g.E] = acc[v].a[j];
    }
}

__global__ __launch_bounds__(64) void simplyp_gof_finish_kernel(const GofArgs g)
{
    const int slot = blockIdx.x * 64 + threadIdx.x;
    const int r = blockIdx.y, v = blockIdx.z;
    if (slot >= g.E) return;
    const int member = g.member_of_slot ? g.member_of_slot[slot] : slot;
    double a[GOF_NACC];
#pragma unroll
    for (int j = 0; j < GOF_NACC; ++j) a[j] = 0.0;
    const int n_chunks = v == SIMPLYP_GOF_Q ? g.n_chunks_q : g.n_chunks_c;
    for (int c = 0; c < n_chunks; ++c) {
        const double* p = g.partial + (((size_t)c * g.R + r) * (GOF_NV * GOF_NACC) + (size_t)v * GOF_NACC) * g.E + slot;
#pragma unroll
        for (int j = 0; j < GOF_NACC; ++j) a[j] += p[(size_t)j * g.E];
    }
    const double n_obs = g.n_obs[r * GOF_NV + v];
    const double nan = __longlong_as_double(0x7ff8000000000000LL);
    double st[SIMPLYP_N_GOF_STATS];
    st[SIMPLYP_GOFSTAT_N_OBS] = n_obs;
    if (n_obs > 10.0) {                                          // visualise_results.py:430
        const double n = a[0], co = g.shift[(size_t)r * 12 + v];
        const double var_o = a[2] - a[1] * a[1] / n;             // sum (o - mean o)^2 over the paired days
        const double var_s = a[4] - a[3] * a[3] / n;
        const double cov = a[5] - a[1] * a[3] / n;
        const double var_lo = a[9] - a[8] * a[8] / n;
        st[SIMPLYP_GOFSTAT_NSE] = 1.0 - a[7] / var_o;                                    // :441
        st[SIMPLYP_GOFSTAT_LOG_NSE] = 1.0 - a[10] / var_lo;                              // :442-443
        st[SIMPLYP_GOFSTAT_R2] = cov * cov / (var_o * var_s);                            // :446-447
        st[SIMPLYP_GOFSTAT_PBIAS] = 100.0 * (a[3] - a[1]) / (a[1] + n * co);             // :448
        st[SIMPLYP_GOFSTAT_NRMSD] = 100.0 * (a[6] / n) / sqrt(var_o / n);                // :449 (ddof 0)
        st[SIMPLYP_GOFSTAT_SUM_LOG_SIM] = a[11];
        st[SIMPLYP_GOFSTAT_SUM_RELSQ] = a[12];
    } else {
#pragma unroll
        for (int j = 1; j < SIMPLYP_N_GOF_STATS; ++j) st[j] = nan;
    }
#pragma unroll
    for (int j = 0; j < SIMPLYP_N_GOF_STATS; ++j)
        g.gof[(((size_t)j * GOF_NV + v) * g.R + r) * g.E + member] = st[j];
}


// ---------------------------------------------------------------------------------------------------------------------
// Spearman's r per member (visualise_results.py:444-445: DataFrame.corr(method='spearman') = Pearson correlation of the
// average ranks of the paired observed and simulated values).  The observed ranks are shared by all members (host); the
// simulated ranks need, per member, the position of each of its n values among its own n values.  n <= a few thousand
// (4 303 discharge days on the Tarland record, a few hundred chemistry days), so the ranks are COUNTED, not sorted:
//     rank_i = (#{j : s_j < s_i} + #{j : s_j <= s_i} + 1) / 2            (ties get their average rank, as pandas gives them)
// -- n^2 compare-and-count per member, independent lanes, no scratch per lane, coalesced: pass 1 writes the member's
// simulated values at the observation days into a compact [n][E] table; pass 2 holds SP_TI values in registers per lane,
// streams all n rows past them, and leaves three partial sums per (block of SP_TI days, member); pass 3 adds the partials in
// block order and forms r.  Cost for the 100 000-member bench table: 1.9e12 compares ~ 0.15 s -- an option, not a default.
constexpr int SP_TI = 32;

struct SpearmanArgs {
    int E, R, D, n;                    // members, output reaches of the table, days, paired days of this (reach, variable)
    int var;                           // SIMPLYP_GOF_*
    int r;                             // output reach (position in the table)
    const double* out;                 // [n_cols][D][R][E]
    long long col_stride;
    int col[4];
    const int32_t* member_of_slot;
    const double* f_tdp;
    const double* a_catch;             // row of the reach [E] (member order), or nullptr for a waterbody table
    const int32_t* day;                // [n] device
    const double* rank_obs;            // [n] device, average ranks of the observations (1-based)
    double sum_ro, sum_ro2;            // sums of rank_obs and rank_obs^2
    double* vals;                      // [n][E] scratch
    double* partial;                   // [n_blocks][3][E]
    int n_blocks;
    double* rho;                       // [6][R][E] member order
};

__device__ __forceinline__ double spearman_sim_value(const SpearmanArgs& g, int d, int slot, double A, double f_tdp)
{
    const size_t off = ((size_t)d * g.R + g.r) * g.E + slot;
    const double qr = g.out[(size_t)g.col[0] * g.col_stride + off];
    // The reference's expressions operation for operation (model.py:784, :788-790), not the reciprocal forms of gof_chem_day: a
    // rank statistic depends on which near-equal values tie, and on a long low-flow recession hundreds of daily values differ
    // in their last bits only -- a differently rounded unit conversion collapses other pairs (seen: 7e-8 in r for one member).
    if (g.var == SIMPLYP_GOF_Q) return g.a_catch ? qr * A * 1000 / 86400 : qr;
    const double ms = g.out[(size_t)g.col[1] * g.col_stride + off];
    const double td = g.out[(size_t)g.col[2] * g.col_stride + off];
    const double pp = g.out[(size_t)g.col[3] * g.col_stride + off];
    const double SS = (ms / qr) / A, TDP = (td / qr) / A, PP = (pp / qr) / A;
    switch (g.var) {
        case SIMPLYP_GOF_SS: return SS;
        case SIMPLYP_GOF_TDP: return TDP;
        case SIMPLYP_GOF_PP: return PP;
        case SIMPLYP_GOF_TP: return TDP + PP;
        default: return TDP * f_tdp;
    }
}

__global__ __launch_bounds__(64) void simplyp_spearman_fill_kernel(const SpearmanArgs g)
{
    const int slot = blockIdx.x * 64 + threadIdx.x;
    if (slot >= g.E) return;
    const int member = g.member_of_slot ? g.member_of_slot[slot] : slot;
    const double A = g.a_catch ? g.a_catch[member] : 86.4;
    const double f = g.f_tdp[member];
    for (int k = blockIdx.y; k < g.n; k += gridDim.y)
        g.vals[(size_t)k * g.E + slot] = spearman_sim_value(g, g.day[k], slot, A, f);
}

__global__ __launch_bounds__(64) void simplyp_spearman_count_kernel(const SpearmanArgs g)
{
    const int slot = blockIdx.x * 64 + threadIdx.x;
    const int i0 = blockIdx.y * SP_TI;
    if (slot >= g.E) return;
    double si[SP_TI];
    unsigned less[SP_TI], le[SP_TI];
    const double* col = g.vals + slot;
#pragma unroll
    for (int t = 0; t < SP_TI; ++t) {
        si[t] = (i0 + t < g.n) ? col[(size_t)(i0 + t) * g.E] : 0.0;
        less[t] = 0u; le[t] = 0u;
    }
    unsigned nan_seen = 0u;
    for (int j = 0; j < g.n; ++j) {
        const double sj = col[(size_t)j * g.E];
        nan_seen |= (sj != sj) ? 1u : 0u;
#pragma unroll
        for (int t = 0; t < SP_TI; ++t) {
            less[t] += (sj < si[t]) ? 1u : 0u;
            le[t] += (sj <= si[t]) ? 1u : 0u;
        }
    }
    double s_r = 0.0, s_rr = 0.0, s_ro = 0.0;
#pragma unroll
    for (int t = 0; t < SP_TI; ++t) {
        if (i0 + t < g.n) {
            const double rk = 0.5 * ((double)less[t] + (double)le[t] + 1.0);
            s_r += rk;
            s_rr = __builtin_fma(rk, rk, s_rr);
            s_ro = __builtin_fma(rk, g.rank_obs[i0 + t], s_ro);
        }
    }
    if (nan_seen) s_r = __longlong_as_double(0x7ff8000000000000LL);       // a NaN simulated value: no ranks for this member
    double* p = g.partial + (size_t)blockIdx.y * 3 * g.E + slot;
    p[0] = s_r; p[(size_t)g.E] = s_rr; p[(size_t)2 * g.E] = s_ro;
}

__global__ __launch_bounds__(64) void simplyp_spearman_finish_kernel(const SpearmanArgs g)
{
    const int slot = blockIdx.x * 64 + threadIdx.x;
    if (slot >= g.E) return;
    const int member = g.member_of_slot ? g.member_of_slot[slot] : slot;
    double s_r = 0.0, s_rr = 0.0, s_ro = 0.0;
    for (int b = 0; b < g.n_blocks; ++b) {
        const double* p = g.partial + (size_t)b * 3 * g.E + slot;
        s_r += p[0]; s_rr += p[(size_t)g.E]; s_ro += p[(size_t)2 * g.E];
    }
    const double n = (double)g.n;
    const double cov = s_ro - s_r * g.sum_ro / n;
    const double var_s = s_rr - s_r * s_r / n;
    const double var_o = g.sum_ro2 - g.sum_ro * g.sum_ro / n;
    g.rho[((size_t)g.var * g.R + g.r) * g.E + member] = cov / sqrt(var_s * var_o);
}

}  // namespace simplyp
