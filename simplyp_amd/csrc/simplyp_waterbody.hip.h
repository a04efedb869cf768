// simplyp_waterbody.hip.h -- the reference's sum_to_waterbody (model.py:851-900) for a whole ensemble (gfx950).
//
// For every member and day: the flagged reaches' Q_cumecs (= Qr*A_catch*1000/86400, model.py:784) and the three daily
// fluxes are added in ascending reach order (DataFrame.sum(axis=1): left to right, NaN counts as 0), then the three
// volume-weighted concentrations (:886-888) and derived_P_species (:842-845).  Same operations in the same order as the
// reference, IEEE division included, so the table is bit-identical to the oracle's (oracle/waterbody.py).
//
// HBM-bound elementwise pass over the table the run left on the device: 32 bytes read per member, day and flagged reach,
// 8 bytes written per requested column.  Lane = member slot (the table's fastest axis), two slots per lane when the
// ensemble size is even (16-byte accesses), one (blockIdx.y) day per workgroup row.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "../../include/simplyp.h"

namespace simplyp {

constexpr int WB_MAX_REACHES = 16;     // flagged reaches handled per launch (a receiving waterbody has a handful of inflows)

struct WaterbodyArgs {
    int E, R, D;                       // members, output reaches of the table, days
    const double* out;                 // [n_cols][D][R][E]
    long long col_stride;              // D*R*E
    int col[4];                        // column slots of Qr, Msus_kg/day, TDP_kg/day, PP_kg/day in `out`
    int n_sum;                         // flagged reaches
    int pos[WB_MAX_REACHES];           // their positions among the table's output reaches, ascending reach id
    int reach[WB_MAX_REACHES];         // their reach ids (rows of a_catch)
    const int32_t* member_of_slot;     // [E] or nullptr
    const double* f_tdp;               // [E], member order
    const double* a_catch;             // reach_params row [S][E], member order
    uint32_t wb_mask;
    double* wb;                        // [popcount(wb_mask)][D][E]
};

template <int W>
__global__ __launch_bounds__(256) void simplyp_waterbody_kernel(const WaterbodyArgs g)
{
    const int i0 = (blockIdx.x * 256 + threadIdx.x) * W;
    const int d = blockIdx.y;
    if (i0 >= g.E) return;
    double Q[W], M[W], T[W], P[W], f[W];
#pragma unroll
    for (int w = 0; w < W; ++w) { Q[w] = 0.0; M[w] = 0.0; T[w] = 0.0; P[w] = 0.0; }
    int member[W];
#pragma unroll
    for (int w = 0; w < W; ++w) {
        member[w] = g.member_of_slot ? g.member_of_slot[i0 + w] : i0 + w;
        f[w] = g.f_tdp[member[w]];
    }
    for (int k = 0; k < g.n_sum; ++k) {
        const size_t row = ((size_t)d * g.R + g.pos[k]) * g.E + i0;
        double v[4][W];
#pragma unroll
        for (int c = 0; c < 4; ++c) {
            const double* p = g.out + (size_t)g.col[c] * g.col_stride + row;
            if (W == 2) { const double2 t = *reinterpret_cast<const double2*>(p); v[c][0] = t.x; v[c][W - 1] = t.y; }
            else v[c][0] = *p;
        }
#pragma unroll
        for (int w = 0; w < W; ++w) {
            const double A = g.a_catch[(size_t)g.reach[k] * g.E + member[w]];
            const double q = v[0][w] * A * 1000 / 86400;                      // model.py:784
            Q[w] += (q != q) ? 0.0 : q;                                       // DataFrame.sum skips NaN (:880)
            M[w] += (v[1][w] != v[1][w]) ? 0.0 : v[1][w];
            T[w] += (v[2][w] != v[2][w]) ? 0.0 : v[2][w];
            P[w] += (v[3][w] != v[3][w]) ? 0.0 : v[3][w];
        }
    }
    double res[SIMPLYP_N_WB][W];
#pragma unroll
    for (int w = 0; w < W; ++w) {
        const double ss = (M[w] / Q[w]) * (1000. / 86400.);                   // :886
        const double td = (T[w] / Q[w]) * (1000. / 86400.);                   // :887
        const double pp = (P[w] / Q[w]) * (1000. / 86400.);                   // :888
        res[SIMPLYP_WB_Q_CUMECS][w] = Q[w]; res[SIMPLYP_WB_MSUS_FLUX][w] = M[w];
        res[SIMPLYP_WB_TDP_FLUX][w] = T[w]; res[SIMPLYP_WB_PP_FLUX][w] = P[w];
        res[SIMPLYP_WB_SS_MGL][w] = ss; res[SIMPLYP_WB_TDP_MGL][w] = td; res[SIMPLYP_WB_PP_MGL][w] = pp;
        res[SIMPLYP_WB_TP_MGL][w] = td + pp;                                  // :842
        res[SIMPLYP_WB_TP_FLUX][w] = T[w] + P[w];                             // :843
        res[SIMPLYP_WB_SRP_MGL][w] = td * f[w];                               // :844
        res[SIMPLYP_WB_SRP_FLUX][w] = T[w] * f[w];                            // :845
    }
    double* o = g.wb + (size_t)d * g.E + i0;
    const size_t wb_stride = (size_t)g.D * g.E;
#pragma unroll
    for (int c = 0; c < SIMPLYP_N_WB; ++c) {
        if (g.wb_mask & (1u << c)) {
            if (W == 2) *reinterpret_cast<double2*>(o) = make_double2(res[c][0], res[c][W - 1]);
            else *o = res[c][0];
            o += wb_stride;
        }
    }
}

}  // namespace simplyp
