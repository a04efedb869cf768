// simplyp_kernels.hip.h -- device code of the SimplyP time-stepping engine for gfx950 (MI355X).
//
// One thread integrates one (ensemble member, reach) 12-variable system through the whole
// daily series: the work of the reference's `for SC` x `for idx` loop nest around
// `odeint(ode_f, ...)` (Current_Release/v0-2A/simplyP/model.py:365, :491, :640).
//
//   * thread -> member e = blockIdx.x*64 + lane, routing chain = blockIdx.y.  A chain is a
//     list of reaches the same thread walks in order (reach-outer, day-inner, like the
//     reference), reading the upstream reaches' daily series it or an earlier launch wrote.
//   * one wavefront per workgroup: a member needs no cooperation, so 64-thread groups give
//     the dispatcher the finest grain to spread waves over the 1024 SIMDs.
//   * met forcing (P, PET, day-of-year) is staged per wavefront through LDS in tiles of
//     TILE_D days; every lane reads the same LDS word (broadcast, conflict-free).
//   * all per-member data is ensemble-major: lane e touches base[e], so every global load and
//     store of a wave is one contiguous 512-byte segment (when the host has reordered members for
//     load balance, parameter loads and output stores become 8-byte gathers/scatters instead; the
//     kernel is arithmetic-bound by three orders of magnitude, so that costs nothing measurable).
//   * the right-hand side is the reference's ode_f (model.py:58-187) with everything that is
//     constant within a day hoisted into `DayConst`; Qr**b_Q and Qr**k_M share one log.
//   * no MFMA: the system is 12 scalar fluxes, there is no contraction to put on matrix cores.
//   * small ensembles (fewer member groups than SIMDs) spread one member over the four lanes of a DPP quad instead
//     (TEAM = 4, `ck_day_quad`): same operations, bit-identical results, ~1.4 x shorter attempts.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include "../../include/simplyp.h"
#include "../../include/simplyp_controller.h"     // the step controller's constants, shared with the CPU oracle

namespace simplyp {

constexpr int WAVE = 64;
constexpr int TILE_D = 256;     // days of forcing staged in LDS at a time (256*(8+8+4) B = 5 KB)

struct KernelArgs {
    int E, S, D, n_sets;
    int lanes;                      // member slots per wavefront (1..64 / team): lanes [l*team, (l+1)*team) of group g own slot g*lanes + l, the rest idle
    int D_stride;                   // row stride of `forcing` in days (D of the full run)
    const int* perm;                // [E] member handled by each lane slot, or nullptr = identity
    int out_by_slot;                // 1: `out` columns are lane slots (coalesced), 0: member ids
    int params_by_slot;             // 1: mp / rp / forcing_of_member were gathered into slot order by the host side
    const double* forcing;          // [n_sets][2][D]
    const int* doy;                 // [D]
    const int* period_of_day;       // [D] output period of each day (time-reduced output), or nullptr
    int n_periods;                  // 0: one output row per day; > 0: rows are sums over periods
    const int* forcing_of_member;   // [E] or nullptr
    const double* mp;               // [NP_M][E]
    const double* rp;               // [NP_R][S][E]
    double* out;                    // [ncols][D][n_out_reaches][E]
    int* status;                    // [E]
    unsigned* member_rhs;           // [E] or nullptr: rhs evaluations per member
    unsigned long long* counters;   // rhs, steps, rejected, wave-level attempts | queue: waits, longest wait, longest stall (polls)
    double* route;                  // [n_slots][4][route_days][E] daily series handed downstream
    int route_days;                 // rows per series buffer: D (chain kernel) or a ring of a few time chunks (queue)
    const int* chain_ptr;           // [n_chains+1]   (this launch; chain kernel)
    const int* chain_reach;         // reach ids in processing order
    const int* up_ptr;              // [S+1]
    const int* up_idx;              // CSR of directly-upstream reaches
    const int* route_slot;          // [S] slot a reach writes its series to, or -1
    const int* out_slot;            // [S] position among the output reaches, or -1
    int n_out_reaches;
    unsigned out_mask;
    int integrator, substeps, max_steps;
    int win_stride;                 // pilot of the load balancer: blockIdx.z runs the window of the forcing that starts
                                    // win_stride * blockIdx.z days in, with its own cost counters and routing scratch (0 = off)
    long long win_route_stride;     // doubles of routing scratch per window
    int dynamic_epc0, dynamic_erod, run_mode_cal, sc_qr0, project_vr;
    double rtol, atol, step_len;
    int team_shift;                 // log2(lanes per member): 0, or 2 when a member is spread over a quad (opts.lanes_per_member = 4)
                                    // (last: the argument block of the one-lane kernels keeps the layout it was tuned with)
};

// ---------------------------------------------------------------------------------------
// fp64 elementary functions on the VALU, trimmed to what this kernel needs: finite,
// in-range arguments (no inf/nan/denormal branches), ~1 ulp.

// exp(x) for |x| < 700, N independent arguments evaluated in lockstep.  n = rint(x/ln2), r = x - n ln2 (two-word ln2),
// degree-13 Taylor on |r| <= 0.3466 (remainder < 4e-18), scale by 2^n.  Every result is produced by the same operations in
// the same order whatever N is and whichever slot it sits in, so exp values do not depend on how call sites group them (the
// one-lane and the four-lane kernels group them differently and stay bit-identical).  Why group at all: the 14 polynomial
// literals are 64-bit, i.e. two s_mov_b32 each, per call site and day -- and a lone wave pays a full issue slot for every
// scalar instruction (tools/micro/valu_rates.hip).  The day boundary evaluates up to seven exponentials (exp(-mu VsA),
// exp(-mu VsS), Qr**b_Q, Qr**k_M, Qr**(1-b_Q) and the two of discretized_soilP): one call site, one set of literals.
template <int N>
__device__ __forceinline__ void sp_expn(const double (&x)[N], double (&r)[N])
{
    double n[N], a[N], p[N];
#pragma unroll
    for (int i = 0; i < N; ++i) {
        n[i] = __builtin_rint(x[i] * 1.4426950408889634);
        a[i] = __builtin_fma(n[i], -6.93147180369123816490e-01, x[i]);
        a[i] = __builtin_fma(n[i], -1.90821492927058770002e-10, a[i]);
        p[i] = 1.6059043836821613e-10;                          // 1/13!
    }
#define SP_STEP(c) _Pragma("unroll") for (int i = 0; i < N; ++i) p[i] = __builtin_fma(p[i], a[i], c);
    SP_STEP(2.08767569878681e-09) SP_STEP(2.505210838544172e-08) SP_STEP(2.755731922398589e-07)       // 1/12! 1/11! 1/10!
    SP_STEP(2.7557319223985893e-06) SP_STEP(2.48015873015873e-05) SP_STEP(1.984126984126984e-04)      // 1/9!  1/8!  1/7!
    SP_STEP(1.388888888888889e-03) SP_STEP(8.333333333333333e-03) SP_STEP(4.1666666666666664e-02)     // 1/6!  1/5!  1/4!
    SP_STEP(1.6666666666666666e-01) SP_STEP(0.5) SP_STEP(1.0) SP_STEP(1.0)                            // 1/3!  1/2!  1  1
#undef SP_STEP
#pragma unroll
    for (int i = 0; i < N; ++i) r[i] = __builtin_amdgcn_ldexp(p[i], (int)n[i]);
}

__device__ __forceinline__ double sp_exp(double x)
{
    const double xs[1] = {x};
    double rs[1];
    sp_expn<1>(xs, rs);
    return rs[0];
}

__device__ __forceinline__ void sp_exp2(double x0, double x1, double& r0, double& r1)
{
    const double xs[2] = {x0, x1};
    double rs[2];
    sp_expn<2>(xs, rs);
    r0 = rs[0]; r1 = rs[1];
}

// 1/x: hardware seed + two Newton steps (full double precision for normal x).
__device__ __forceinline__ double sp_rcp(double x)
{
    double r = __builtin_amdgcn_rcp(x);
    r = __builtin_fma(__builtin_fma(-x, r, 1.0), r, r);
    r = __builtin_fma(__builtin_fma(-x, r, 1.0), r, r);
    return r;
}

// log(x) for normal x > 0.  x = 2^k (1+f), sqrt(1/2) < 1+f <= sqrt(2); s = f/(2+f);
// log(1+f) = f - hfsq + s (hfsq + R(s^2)), R = the classic degree-7 even minimax polynomial
// (fdlibm e_log.c coefficients Lg1..Lg7).
// One Newton step on v_rcp_f64 (raw: 4.6e-8 relative, one step: 2e-15, two: correctly rounded -- tools/micro/rcp_accuracy.hip).
// Enough for the derivatives of the auxiliary states, which are re-evaluated exactly every few steps anyway.
__device__ __forceinline__ double sp_rcp1(double x)
{
    double r = __builtin_amdgcn_rcp(x);
    return __builtin_fma(__builtin_fma(-x, r, 1.0), r, r);
}

__device__ __forceinline__ double sp_log(double x)
{
    double m = __builtin_amdgcn_frexp_mant(x);            // [0.5, 1)
    int k = __builtin_amdgcn_frexp_exp(x);
    const bool lo = m < 0.70710678118654752440;
    m = lo ? m + m : m;
    k = lo ? k - 1 : k;
    const double f = m - 1.0;
    const double s = f * sp_rcp(2.0 + f);
    const double z = s * s;
    double R = 1.479819860511658591e-01;
    R = __builtin_fma(R, z, 1.531383769920937332e-01);
    R = __builtin_fma(R, z, 1.818357216161805012e-01);
    R = __builtin_fma(R, z, 2.222219843214978396e-01);
    R = __builtin_fma(R, z, 2.857142874366239149e-01);
    R = __builtin_fma(R, z, 3.999999999940941908e-01);
    R = __builtin_fma(R, z, 6.666666666666735130e-01);
    R = R * z;
    const double hfsq = 0.5 * f * f;
    const double dk = (double)k;
    // k ln2_hi + (f - (hfsq - (s (hfsq + R) + k ln2_lo)))
    const double t = __builtin_fma(s, hfsq + R, dk * 1.90821492927058770002e-10);
    return __builtin_fma(dk, 6.93147180369123816490e-01, f - (hfsq - t));
}

// ---------------------------------------------------------------------------------------
// Right-hand side.  State y[8] = VsA, VsS, Vg, Vr, Qr, Msus, TDPr, PPr (the reference's slots
// 0,1,2,3,4,6,8,10); the four daily integrators (slots 5,7,9,11) never feed back, so only their
// integrands q[4] = Qr, Msus Qr/Vr, TDPr Qr/Vr, PPr Qr/Vr are returned.

struct DayConst {
    // soil boxes (model.py:105-110)
    double c0;          // P (1 - f_quick)
    double aE;          // alpha * PET
    double mu, fc, inv_d;       // inv_d = 1 / (0.01 fc)
    double invTsA, invTsS;
    // groundwater (:121-124)
    double invTg, Qgmin, inv_dg, beta, fA, fS;
    // reach hydrology (:127-132)
    double qin;         // Qq + Qr_US
    double omb;         // 1 - beta
    double cQ;          // a_Q 86400 / ((1 - b_Q) L_reach)
    double bQ, kM;
    // sediment (:138-147)
    double Esum;        // f_Ar Esus_A + f_IG Esus_IG + f_S Esus_S
    double MsusUS;
    // TDP (:154-168)
    double tA, tS;      // coefficients of QsA, QsS
    double tg;          // TDPg * A_catch
    double tconst;      // quick-flow terms + TDPeff + TDPr_US
    // PP (:171-180)
    double cPP, PPrUS;
    // augmented form only
    double invKv;       // a_Q 86400 / L_reach = (Qr/Vr) / Qr**b_Q
    // fused forms used by SysAug::f (one FMA where the literal form has a subtraction and a multiplication)
    double wA0, wS0;    // -fc / T_s:   (Vs - fc) / T_s = fma(Vs, invTs, w0)
    double s0;          // -fc inv_d:   (Vs - fc) inv_d = fma(Vs, inv_d, s0)
    double c0m;         // c0 - aE:     c0 + aE (E - 1) = fma(aE, E, c0m)
    double invKvc;      // invKv / cQ:  the state carried for Qr**b_Q is cQ Qr**b_Q, so dQr = inflow * state and Qr/Vr = state * invKvc
    double dgate;       // 0.01 fc: width of the soil-water gate (model.py:35), floor of the soil boxes' error scale (SysAug)
    double dgq;         // 0.01 Qg_min: width of the groundwater gate (:121); 0 when Qg_min is 0 (a plain step at 0)
};

// f_x(x, threshold, 0.01) with u = x - threshold and inv_d = 1/(0.01 threshold) (model.py:23-37):
// 0 below the threshold, 1 above threshold + d, 3s^2 - 2s^3 in between -- as one clamped cubic.
__device__ __forceinline__ double gate(double u, double inv_d)
{
    double s = u * inv_d;
    s = __builtin_fmin(__builtin_fmax(s, 0.0), 1.0);
    return s * s * __builtin_fma(-2.0, s, 3.0);
}

__device__ __forceinline__ void rhs(const double (&y)[8], const DayConst& c, double (&dy)[8], double (&q)[4])
{
    const double uA = y[0] - c.fc, uS = y[1] - c.fc;
    const double QsA = uA * gate(uA, c.inv_d) * c.invTsA;                         // :105
    const double QsS = uS * gate(uS, c.inv_d) * c.invTsS;                         // :109
    double eA, eS;
    sp_exp2(-c.mu * y[0], -c.mu * y[1], eA, eS);
    dy[0] = __builtin_fma(c.aE, eA - 1.0, c.c0) - QsA;                            // :106
    dy[1] = __builtin_fma(c.aE, eS - 1.0, c.c0) - QsS;                            // :110
    const double Qsum = __builtin_fma(c.fA, QsA, c.fS * QsS);
    const double ug = __builtin_fma(y[2], c.invTg, -c.Qgmin);
    const double Qg = __builtin_fma(gate(ug, c.inv_dg), ug, c.Qgmin);             // :121-122
    dy[2] = __builtin_fma(c.beta, Qsum, -Qg);                                     // :124
    const double Qr = y[4];
    const double inflow = __builtin_fma(c.omb, Qsum, c.qin) + Qg - Qr;            // :127-129
    const double lq = sp_log(Qr);
    double pb, pk;
    sp_exp2(c.bQ * lq, c.kM * lq, pb, pk);                                        // Qr**b_Q, Qr**k_M
    dy[4] = inflow * c.cQ * pb;                                                   // :130
    dy[3] = inflow;                                                               // :131
    const double qv = Qr * sp_rcp(y[3]);
    const double oM = y[5] * qv, oT = y[6] * qv, oP = y[7] * qv;
    dy[5] = __builtin_fma(c.Esum, pk, c.MsusUS) - oM;                             // :141-145
    dy[6] = __builtin_fma(c.tA, QsA, __builtin_fma(c.tS, QsS, __builtin_fma(c.tg, Qg, c.tconst))) - oT;   // :154-166
    dy[7] = __builtin_fma(c.cPP, pk, c.PPrUS) - oP;                               // :171-178
    q[0] = Qr; q[1] = oM; q[2] = oT; q[3] = oP;                                   // :132,:147,:168,:180
}

// ---------------------------------------------------------------------------------------
// Integrators over one day [0, T].  yq[4] (the daily integrators) starts at 0 (model.py:618).

__device__ __forceinline__ void rk4_day(double (&y)[8], double (&yq)[4], const DayConst& c, double T, int n)
{
    const double h = T / (double)n;
    for (int st = 0; st < n; ++st) {
        double k[8], kq[4], yt[8], acc[8], accq[4];
        rhs(y, c, k, kq);
#pragma unroll
        for (int i = 0; i < 8; ++i) { acc[i] = k[i]; yt[i] = __builtin_fma(0.5 * h, k[i], y[i]); }
#pragma unroll
        for (int i = 0; i < 4; ++i) accq[i] = kq[i];
        rhs(yt, c, k, kq);
#pragma unroll
        for (int i = 0; i < 8; ++i) { acc[i] = __builtin_fma(2.0, k[i], acc[i]); yt[i] = __builtin_fma(0.5 * h, k[i], y[i]); }
#pragma unroll
        for (int i = 0; i < 4; ++i) accq[i] = __builtin_fma(2.0, kq[i], accq[i]);
        rhs(yt, c, k, kq);
#pragma unroll
        for (int i = 0; i < 8; ++i) { acc[i] = __builtin_fma(2.0, k[i], acc[i]); yt[i] = __builtin_fma(h, k[i], y[i]); }
#pragma unroll
        for (int i = 0; i < 4; ++i) accq[i] = __builtin_fma(2.0, kq[i], accq[i]);
        rhs(yt, c, k, kq);
        const double h6 = h * (1.0 / 6.0);
#pragma unroll
        for (int i = 0; i < 8; ++i) y[i] = __builtin_fma(h6, acc[i] + k[i], y[i]);
#pragma unroll
        for (int i = 0; i < 4; ++i) yq[i] = __builtin_fma(h6, accq[i] + kq[i], yq[i]);
    }
}

// ---- scalar helpers overloaded on the working precision of an integrator -------------------------------
__device__ __forceinline__ double sp_fma(double a, double b, double c) { return __builtin_fma(a, b, c); }
__device__ __forceinline__ float sp_fma(float a, float b, float c) { return __builtin_fmaf(a, b, c); }
__device__ __forceinline__ double sp_max(double a, double b) { return __builtin_fmax(a, b); }
__device__ __forceinline__ float sp_max(float a, float b) { return __builtin_fmaxf(a, b); }
// max(|a|, |b|) in ONE instruction.  Written in C the compiler first quiets a possible signalling NaN in an operand it did not
// produce itself (v_max_f64 x, |a|, |a|), once per component and attempt; the instruction does that by itself.
__device__ __forceinline__ double sp_absmax(double a, double b)
{
    double r;
    asm("v_max_f64 %0, |%1|, |%2|" : "=v"(r) : "v"(a), "v"(b));
    return r;
}
__device__ __forceinline__ float sp_absmax(float a, float b) { return __builtin_fmaxf(__builtin_fabsf(a), __builtin_fabsf(b)); }
__device__ __forceinline__ double sp_min(double a, double b) { return __builtin_fmin(a, b); }
__device__ __forceinline__ float sp_min(float a, float b) { return __builtin_fminf(a, b); }
__device__ __forceinline__ double sp_abs(double a) { return __builtin_fabs(a); }
__device__ __forceinline__ float sp_abs(float a) { return __builtin_fabsf(a); }
__device__ __forceinline__ double sp_rcp_fast(double a) { return __builtin_amdgcn_rcp(a); }     // error-norm scale only
__device__ __forceinline__ float sp_rcp_fast(float a) { return __builtin_amdgcn_rcpf(a); }
// sign bits of a and b differ <=> the result is negative (one 32-bit xor on the high words)
__device__ __forceinline__ int sp_sign_xor(double a, double b) { return __double2hiint(a) ^ __double2hiint(b); }
__device__ __forceinline__ int sp_sign_xor(float a, float b) { return __float_as_int(a) ^ __float_as_int(b); }
// t if t > lo, else something huge (knee targeting: a candidate time that does not count).  For doubles only the high word is
// selected -- the low word of t under a huge exponent is still huge -- which saves a v_cndmask per candidate.
__device__ __forceinline__ double sp_if_above(double t, double lo)
{
    const int hi = (t > lo) ? __double2hiint(t) : 0x7e37e43c;       // high word of 1.0e300
    return __hiloint2double(hi, __double2loint(t));
}
__device__ __forceinline__ float sp_if_above(float t, float lo) { return (t > lo) ? t : 1.0e30f; }
// min of two finite or huge values as ONE instruction (the builtin first canonicalises operands it cannot prove quiet -- the
// bit-assembled results of sp_if_above -- with a v_max_f64 x, x each)
__device__ __forceinline__ double sp_min_raw(double a, double b)
{
    double r;
    asm("v_min_f64 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b));
    return r;
}
__device__ __forceinline__ float sp_min_raw(float a, float b) { return __builtin_fminf(a, b); }
template <typename R> __device__ __forceinline__ R sp_huge();
template <> __device__ __forceinline__ double sp_huge<double>() { return 1.0e300; }
template <> __device__ __forceinline__ float sp_huge<float>() { return 1.0e30f; }

// The augmented form of the same system (SIMPLYP_INTEG_CASHKARP_AUG; derivation and pinning:
// oracle/simplyp_oracle.c `ode_aug`): exp(-mu Vs), Qr**b_Q, Qr**k_M are carried as extra states through
// their exact ODEs, Vr comes from its invariant, so the right-hand side is ~70 multiply-adds and one
// reciprocal.  State z[11] = VsA VsS Vg Qr Msus TDPr PPr EA ES pb pk.
struct SysLiteral {
    typedef double real;
    typedef DayConst dayconst;
    static constexpr int NS = 8;
    static constexpr int RESYNC_EVERY = 0;
    static constexpr int N_ERR = 8;             // all 8 states and the 4 daily integrals enter the error norm
    static constexpr bool QUAD_IN_NORM = true;
    static constexpr bool SOIL_REL = false;
    static constexpr bool KINK_AWARE = false;
    static __device__ __forceinline__ void resync(double (&)[8], const DayConst&) {}
    static __device__ __forceinline__ void f(const double (&y)[8], const DayConst& c, double (&dy)[8], double (&q)[4])
    {
        rhs(y, c, dy, q);
    }
};

struct SysAug {
    typedef double real;
    typedef DayConst dayconst;
    static constexpr int NS = 11;
    // pb, pk are only neutrally stable about Qr**b_Q, Qr**k_M and are not in the error norm: re-evaluate them after every
    // 8th attempt of the day (a storm day can take 100+), as the oracle does.  With every 16 accepted steps, 3 of 8192
    // Monte-Carlo members had a day above 1e-6 in the sediment / PP fluxes (tools/probe_tolerance.py).
    static constexpr int RESYNC_EVERY = SIMPLYP_CTRL_RESYNC_EVERY;
    // error norm over the 7 physical states (the auxiliary states are functions of them, the daily integrals
    // quadratures of them; see oracle/simplyp_oracle.c)
    static constexpr int N_ERR = SIMPLYP_CTRL_N_ERR;
    static constexpr bool QUAD_IN_NORM = false;
    // Error scale of the two soil boxes: rtol * max(|Vs - fc|, 0.01 fc) instead of rtol * |Vs|.  Everything downstream sees a
    // soil box through Vs - fc (the gate argument and the flow (Vs - fc)/T_s, model.py:105-110), a difference of ~1 mm between
    // numbers of ~300 mm: an error that is 1e-8 of Vs is 3e-6 of the flow.  Mostly the soil boxes do not limit the step and
    // their error stays far below their tolerance -- but a step that straddles a knee of the gate (C1 only) uses the tolerance
    // up, and the flow error then shows in every reach output (worst member-day of 12 288 Monte-Carlo members at rtol 2e-8:
    // 9.0e-7 in the TDP flux before, 2.7e-7 after; profiles/r02_experiments.md).  Costs no steps: see there.
    static constexpr bool SOIL_REL = true;
    // Steps across a knee of a gate.  f_x (model.py:23-37) is C1 only: where a soil box crosses fc or 1.01 fc, or Vg / T_g crosses
    // Qg_min or 1.01 Qg_min, the second derivative of the right-hand side jumps, a 5(4) pair drops to third order on the step
    // that straddles the knee and its embedded estimate no longer bounds the error (measured: the worst member-days of the
    // 100 000-member bench ensemble were all such steps, 50 x above their tolerance).  A step that, along its first slope, has a
    // knee within KINK_REACH x its length without having been aimed at it (below) therefore has its error estimate multiplied
    // by KINK_SOIL / KINK_GW: it is accepted only if it is short.  With this, and Qr**k_M in the norm at AUX_WEIGHT x the
    // tolerance (on a day when a nearly dry reach is wetted Qr**k_M grows 200-fold and its own truncation error showed in the
    // sediment flux), the error at a given rtol drops 7-fold: rtol 1e-7 now gives what 1e-8 gave -- 84 instead of 126
    // right-hand sides per catchment-day (profiles/r02_experiments.md).
    // (KINK_SOIL for the knees of the two soil-water gates, crossed on most wet days; KINK_GW for the groundwater gate, whose
    // zone is 1 % of Qg_min wide and is crossed a few times a year by members with a low Qg_min: with 10 there, 7 members of
    // the 100 000 still had a day at 5-9e-7)
    static constexpr bool KINK_AWARE = true;
    static constexpr double KINK_SOIL = SIMPLYP_CTRL_KINK_SOIL, KINK_GW = SIMPLYP_CTRL_KINK_GW;      // (values: include/simplyp_controller.h)
    static constexpr double KINK_REACH = SIMPLYP_CTRL_KINK_REACH;
    // Better than paying for a crossing with a rejected attempt: aim at the knee.  After the first stage the time to the
    // nearest knee along the first slope is known; if it lies inside the step (between KNEE_LO and KNEE_HI of it) the step is
    // cut to end KNEE_OVER x that far, i.e. just past the knee -- the right-hand side is then smooth over all but the last
    // few percent of the step, the pair keeps its order, no inflation is applied, and the next step starts on the far side.
    // The step size carried on is not reduced by such a cut.  Wave-level attempts per day on the bench ensemble 22.2 -> 18.7
    // (fewer rejections, and lanes that cross a knee no longer hold their wave for 2-3 extra attempts); inflation stays for
    // knees in the step's last tenth or just beyond its end; knees within its first KNEE_LO are left alone (benign).
    // DAY_START: a day starts with this share of the step size carried over midnight (see ck_day).
    // (A step aimed at a knee of the groundwater gate keeps a factor KNEE_GW: where the first slope misjudges the crossing
    // time the knee sits well inside the step, and that gate does not forgive it -- one member-day of the 100 000 at 1.5e-6.)
    static constexpr double KNEE_LO = SIMPLYP_CTRL_KNEE_LO, KNEE_HI = SIMPLYP_CTRL_KNEE_HI, KNEE_OVER = SIMPLYP_CTRL_KNEE_OVER,
                            KNEE_GW = SIMPLYP_CTRL_KNEE_GW;
    static constexpr double DAY_START = SIMPLYP_CTRL_DAY_START;
    static constexpr double AUX_WEIGHT = SIMPLYP_CTRL_AUX_WEIGHT;
    // Expansive reach (include/simplyp_controller.h): where the flow equation amplifies errors instead of damping them -- a nearly dry
    // reach being wetted, b_Q x (net inflow) > Qr -- a step's estimate is multiplied by EXPAND.  With z[9] = cQ Qr**b_Q the first
    // slope of Qr is k1[3] = (net inflow) z[9], so the test is b_Q k1[3] > Qr z[9]: two multiplications and a compare.
    static constexpr double EXPAND = SIMPLYP_CTRL_EXPAND;
    // z[9] carries cQ * Qr**b_Q (the factor the flow equation multiplies it with, folded into the state: one multiplication
    // less per right-hand side; its ODE is linear in it, so the scaling changes nothing else)
    static __device__ __forceinline__ void resync(double (&z)[11], const DayConst& c)
    {
        const double lq = sp_log(z[3]);
        double pb;
        sp_exp2(c.bQ * lq, c.kM * lq, pb, z[10]);
        z[9] = c.cQ * pb;
    }
    // 61 fp64 operations (the literal grouping of the same expressions takes 66): every "(x - a) * b" is one FMA with the
    // per-day / per-member constant -a*b, and cQ rides inside the auxiliary state.
    static __device__ __forceinline__ void f(const double (&z)[11], const DayConst& c, double (&dz)[11], double (&q)[4])
    {
        // soil boxes: Qs = (Vs - fc)/T_s * f_x(Vs, fc) with f_x = s^2 (3 - 2 s), s = clamp((Vs - fc) inv_d, 0, 1)   (:105, :109)
        const double wA = __builtin_fma(z[0], c.invTsA, c.wA0), wS = __builtin_fma(z[1], c.invTsS, c.wS0);
        double sA = __builtin_fma(z[0], c.inv_d, c.s0), sS = __builtin_fma(z[1], c.inv_d, c.s0);
        sA = __builtin_fmin(__builtin_fmax(sA, 0.0), 1.0);
        sS = __builtin_fmin(__builtin_fmax(sS, 0.0), 1.0);
        const double QsA = wA * ((sA * sA) * __builtin_fma(-2.0, sA, 3.0));
        const double QsS = wS * ((sS * sS) * __builtin_fma(-2.0, sS, 3.0));
        dz[0] = __builtin_fma(c.aE, z[7], c.c0m) - QsA;                               // :106 with EA = exp(-mu VsA)
        dz[1] = __builtin_fma(c.aE, z[8], c.c0m) - QsS;                               // :110
        const double Qsum = __builtin_fma(c.fA, QsA, c.fS * QsS);
        const double ug = __builtin_fma(z[2], c.invTg, -c.Qgmin);
        const double Qg = __builtin_fma(gate(ug, c.inv_dg), ug, c.Qgmin);             // :121-122
        dz[2] = __builtin_fma(c.beta, Qsum, -Qg);                                     // :124
        const double Qr = z[3], pbc = z[9], pk = z[10];
        const double inflow = __builtin_fma(c.omb, Qsum, c.qin) + Qg - Qr;            // :127-129
        const double dQr = inflow * pbc;                                              // :130
        dz[3] = dQr;
        const double kap = pbc * c.invKvc;                                            // Qr / Vr on the invariant
        const double oM = z[4] * kap, oT = z[5] * kap, oP = z[6] * kap;
        dz[4] = __builtin_fma(c.Esum, pk, c.MsusUS) - oM;                             // :141-145
        dz[5] = __builtin_fma(c.tA, QsA, __builtin_fma(c.tS, QsS, __builtin_fma(c.tg, Qg, c.tconst))) - oT;   // :154-166
        dz[6] = __builtin_fma(c.cPP, pk, c.PPrUS) - oP;                               // :171-178
        dz[7] = -c.mu * z[7] * dz[0];                                                 // d exp(-mu VsA)
        dz[8] = -c.mu * z[8] * dz[1];
        const double r = dQr * sp_rcp1(Qr);
        dz[9] = c.bQ * pbc * r;                                                       // d (cQ Qr**b_Q)
        dz[10] = c.kM * pk * r;                                                       // d Qr**k_M
        q[0] = Qr; q[1] = oM; q[2] = oT; q[3] = oP;                                   // :132,:147,:168,:180
    }
};


// fp32 working precision for the augmented system (SIMPLYP_INTEG_CASHKARP_AUG_F32, BASELINE config C5): the 11 stage
// states and all stage arithmetic in float (a lone wave issues v_fma_f32 twice as fast as v_fma_f64: 1.29 against 2.6 ns);
// the four daily integrals are accumulated in fp64, and everything outside the day's integration -- the carried
// state, labile soil P and soil-water TDP (whose daily increments are ~1e-4 of their size), the day constants --
// stays fp64.
struct DayConstF {
    float c0, aE, mu, fc, inv_d, invTsA, invTsS, invTg, Qgmin, inv_dg, beta, fA, fS, qin, omb, cQ, bQ, kM,
          Esum, MsusUS, tA, tS, tg, tconst, cPP, PPrUS, invKv;
    __device__ __forceinline__ explicit DayConstF(const DayConst& c)
        : c0((float)c.c0), aE((float)c.aE), mu((float)c.mu), fc((float)c.fc), inv_d((float)c.inv_d),
          invTsA((float)c.invTsA), invTsS((float)c.invTsS), invTg((float)c.invTg), Qgmin((float)c.Qgmin),
          inv_dg((float)__builtin_fmin(c.inv_dg, 1.0e30)), beta((float)c.beta), fA((float)c.fA), fS((float)c.fS),
          qin((float)c.qin), omb((float)c.omb), cQ((float)c.cQ), bQ((float)c.bQ), kM((float)c.kM),
          Esum((float)c.Esum), MsusUS((float)c.MsusUS), tA((float)c.tA), tS((float)c.tS), tg((float)c.tg),
          tconst((float)c.tconst), cPP((float)c.cPP), PPrUS((float)c.PPrUS), invKv((float)c.invKv) {}
};

__device__ __forceinline__ float gate(float u, float inv_d)
{
    float s = u * inv_d;
    s = __builtin_fminf(__builtin_fmaxf(s, 0.0f), 1.0f);
    return s * s * __builtin_fmaf(-2.0f, s, 3.0f);
}

struct SysAugF {
    typedef float real;
    typedef DayConstF dayconst;
    static constexpr int NS = 11;
    static constexpr int RESYNC_EVERY = SIMPLYP_CTRL_RESYNC_EVERY;
    static constexpr int N_ERR = SIMPLYP_CTRL_N_ERR;
    static constexpr bool QUAD_IN_NORM = false;
    // The plain controller (relative tolerance on every state, no knee logic): at this mode's tolerances (rtol ~ 1e-5) the
    // float stages, not the knees, limit the accuracy -- rtol * |Vs - fc| would be one float ulp of a 300 mm store -- and the
    // knee logic costs a fifth more instructions per attempt (measured on C5: 306 -> 373 ms per pass with it).
    static constexpr bool SOIL_REL = false;
    static constexpr bool KINK_AWARE = false;
    static __device__ __forceinline__ void resync(float (&z)[11], const DayConstF& c)
    {
        const float lq = __logf(z[3]);
        z[9] = __expf(c.bQ * lq);
        z[10] = __expf(c.kM * lq);
    }
    static __device__ __forceinline__ void f(const float (&z)[11], const DayConstF& c, float (&dz)[11], float (&q)[4])
    {
        const float uA = z[0] - c.fc, uS = z[1] - c.fc;
        const float QsA = uA * gate(uA, c.inv_d) * c.invTsA;
        const float QsS = uS * gate(uS, c.inv_d) * c.invTsS;
        dz[0] = __builtin_fmaf(c.aE, z[7] - 1.0f, c.c0) - QsA;
        dz[1] = __builtin_fmaf(c.aE, z[8] - 1.0f, c.c0) - QsS;
        const float Qsum = __builtin_fmaf(c.fA, QsA, c.fS * QsS);
        const float ug = __builtin_fmaf(z[2], c.invTg, -c.Qgmin);
        const float Qg = __builtin_fmaf(gate(ug, c.inv_dg), ug, c.Qgmin);
        dz[2] = __builtin_fmaf(c.beta, Qsum, -Qg);
        const float Qr = z[3], pb = z[9], pk = z[10];
        const float inflow = __builtin_fmaf(c.omb, Qsum, c.qin) + Qg - Qr;
        const float dQr = inflow * c.cQ * pb;
        dz[3] = dQr;
        const float kap = pb * c.invKv;
        const float oM = z[4] * kap, oT = z[5] * kap, oP = z[6] * kap;
        dz[4] = __builtin_fmaf(c.Esum, pk, c.MsusUS) - oM;
        dz[5] = __builtin_fmaf(c.tA, QsA, __builtin_fmaf(c.tS, QsS, __builtin_fmaf(c.tg, Qg, c.tconst))) - oT;
        dz[6] = __builtin_fmaf(c.cPP, pk, c.PPrUS) - oP;
        dz[7] = -c.mu * z[7] * dz[0];
        dz[8] = -c.mu * z[8] * dz[1];
        float rq = __builtin_amdgcn_rcpf(Qr);
        rq = __builtin_fmaf(__builtin_fmaf(-Qr, rq, 1.0f), rq, rq);
        const float r = dQr * rq;
        dz[9] = c.bQ * pb * r;
        dz[10] = c.kM * pk * r;
        q[0] = Qr; q[1] = oM; q[2] = oT; q[3] = oP;
    }
};

// Cash-Karp 5(4), per-lane step control; the rule is documented (and mirrored for the parity
// tests) in oracle/simplyp_oracle.c `cashkarp_day` / `cashkarp_aug_day`.  Lanes that have reached T idle
// with a zero step until the slowest lane of the wavefront is done.
struct CkCounters { unsigned rhs, steps, rejected, wave_trips; bool capped, poisoned; };

// The tableau(s) in LDS, for the instantiations that switch pairs lane by lane (STIFF, opts.stiff_pair): row 0 Cash-Karp, row 1 the
// stability-optimised 4(3) pair of include/simplyp_controller.h (same sparsity: 15 a, 4 b, 5 e).  A lane reads its coefficients
// through a per-lane row offset; rows are TAB_STRIDE doubles apart so that the two rows of a coefficient sit in different banks.
constexpr int TAB_STRIDE = 26;
enum { TA21 = 0, TA31, TA32, TA41, TA42, TA43, TA51, TA52, TA53, TA54, TA61, TA62, TA63, TA64, TA65,
       TB1, TB3, TB4, TB6, TE1, TE3, TE4, TE5, TE6, TAB_N };
__device__ __forceinline__ void tableau_to_lds(double* s_tab, int lane)
{
    const double ck[TAB_N] = {1.0 / 5, 3.0 / 40, 9.0 / 40, 3.0 / 10, -9.0 / 10, 6.0 / 5, -11.0 / 54, 5.0 / 2, -70.0 / 27, 35.0 / 27,
                              1631.0 / 55296, 175.0 / 512, 575.0 / 13824, 44275.0 / 110592, 253.0 / 4096,
                              37.0 / 378, 250.0 / 621, 125.0 / 594, 512.0 / 1771,
                              37.0 / 378 - 2825.0 / 27648, 250.0 / 621 - 18575.0 / 48384, 125.0 / 594 - 13525.0 / 55296, -277.0 / 14336,
                              512.0 / 1771 - 1.0 / 4};
    const double st[TAB_N] = {SIMPLYP_STIFF_A21, SIMPLYP_STIFF_A31, SIMPLYP_STIFF_A32, SIMPLYP_STIFF_A41, SIMPLYP_STIFF_A42, SIMPLYP_STIFF_A43,
                              SIMPLYP_STIFF_A51, SIMPLYP_STIFF_A52, SIMPLYP_STIFF_A53, SIMPLYP_STIFF_A54,
                              SIMPLYP_STIFF_A61, SIMPLYP_STIFF_A62, SIMPLYP_STIFF_A63, SIMPLYP_STIFF_A64, SIMPLYP_STIFF_A65,
                              SIMPLYP_STIFF_B1, SIMPLYP_STIFF_B3, SIMPLYP_STIFF_B4, SIMPLYP_STIFF_B6,
                              SIMPLYP_STIFF_E1, SIMPLYP_STIFF_E3, SIMPLYP_STIFF_E4, SIMPLYP_STIFF_E5, SIMPLYP_STIFF_E6};
    if (lane < TAB_N) { s_tab[lane] = ck[lane]; s_tab[TAB_STRIDE + lane] = st[lane]; }
    __syncthreads();
}

template <class SYS, bool STIFF = false>
__device__ __forceinline__ void ck_day(typename SYS::real (&y)[SYS::NS], double (&yq)[4], const typename SYS::dayconst& c,
                                       const double T_, const double rtol_, const double atol_, int max_steps, double& h_carry,
                                       CkCounters& cnt, const bool lane_active, const double* s_tab = nullptr)
{
    static_assert(!STIFF || SYS::KINK_AWARE, "the second pair exists for the knee-aware fp64 scheme only");
    typedef typename SYS::real R;       // working precision of the stages (the daily integrals yq stay fp64)
    constexpr int NS = SYS::NS;
    const R T = (R)T_, rtol = (R)rtol_, atol = (R)atol_;
    R rtol_aux = rtol, atol_aux = atol;
    if constexpr (SYS::KINK_AWARE) { rtol_aux = (R)SYS::AUX_WEIGHT * rtol; atol_aux = (R)SYS::AUX_WEIGHT * atol; }
    R a21 = 1.0 / 5;
    R a31 = 3.0 / 40, a32 = 9.0 / 40;
    R a41 = 3.0 / 10, a42 = -9.0 / 10, a43 = 6.0 / 5;
    R a51 = -11.0 / 54, a52 = 5.0 / 2, a53 = -70.0 / 27, a54 = 35.0 / 27;
    R a61 = 1631.0 / 55296, a62 = 175.0 / 512, a63 = 575.0 / 13824, a64 = 44275.0 / 110592, a65 = 253.0 / 4096;
    R b1 = 37.0 / 378, b3 = 250.0 / 621, b4 = 125.0 / 594, b6 = 512.0 / 1771;
    R e1 = 37.0 / 378 - 2825.0 / 27648, e3 = 250.0 / 621 - 18575.0 / 48384,
      e4 = 125.0 / 594 - 13525.0 / 55296, e5 = -277.0 / 14336, e6 = 512.0 / 1771 - 1.0 / 4;
    // Keep the tableau in scalar registers across the attempt loop.  Left alone, the compiler re-materialises each 64-bit
    // literal with two s_mov_b32 right before its use -- free when other waves fill the issue slots, but this kernel runs
    // one wave per SIMD, where every scalar instruction costs a full slot (tools/micro/valu_rates.hip: 2.6 ns, the price of
    // an fp64 FMA).  Passing the values through an empty asm with an "s" operand makes them opaque: -1.4 % kernel time.
#define SP_KEEP_SCALAR(x) asm volatile("" : "+s"(x))
    if constexpr (!STIFF) {
    SP_KEEP_SCALAR(a21); SP_KEEP_SCALAR(a31); SP_KEEP_SCALAR(a32); SP_KEEP_SCALAR(a41); SP_KEEP_SCALAR(a42); SP_KEEP_SCALAR(a43);
    SP_KEEP_SCALAR(a51); SP_KEEP_SCALAR(a52); SP_KEEP_SCALAR(a53); SP_KEEP_SCALAR(a54);
    SP_KEEP_SCALAR(a61); SP_KEEP_SCALAR(a62); SP_KEEP_SCALAR(a63); SP_KEEP_SCALAR(a64); SP_KEEP_SCALAR(a65);
    SP_KEEP_SCALAR(b1); SP_KEEP_SCALAR(b3); SP_KEEP_SCALAR(b4); SP_KEEP_SCALAR(b6);
    SP_KEEP_SCALAR(e1); SP_KEEP_SCALAR(e3); SP_KEEP_SCALAR(e4); SP_KEEP_SCALAR(e5); SP_KEEP_SCALAR(e6);
    }
    R huge = sp_huge<R>(), c11 = (R)1.1;
    SP_KEEP_SCALAR(huge); SP_KEEP_SCALAR(c11);
#undef SP_KEEP_SCALAR
    // STIFF: a coefficient is read from this lane's row of the LDS table (row offset toff, set once the attempt's pair is known); else the
    // SGPR constant
    // (an integer row offset into the __shared__ table, not a pointer: a selected / laundered pointer loses its address space and
    // the reads become flat loads with a full wait behind each)
#define TC(name, idx) (STIFF ? (R)s_tab[toff + (idx)] : name)
    int toff = 0;
    R t = 0, h = (R)h_carry;
    // (the step size carried over the day boundary belongs to the smooth end of a day; the forcing jumps at midnight, and the
    // first attempt of the new day with it was rejected on 85 % of the member-days: SYS::DAY_START of it is the better guess)
    if constexpr (SYS::KINK_AWARE) h *= (R)SYS::DAY_START;
    if (!(h > (R)0) || h > T) h = T;
    if constexpr (STIFF) {
        // (opts.stiff_pair) ... and at most SIMPLYP_STIFF_Z_START relaxation times of the reach: the day opens with its transient
        const R h0 = (R)SIMPLYP_STIFF_Z_START * sp_rcp1(y[9]);
        h = (h > h0) ? h0 : h;
    }
    // attempts made today by every lane that is still alive: lanes attempt in lockstep (one attempt per trip of the loop
    // below for everyone who has not finished), so one wave-uniform counter serves them all -- and the step cap and the
    // resync schedule derived from it are scalar branches
    int trip = 0;
    unsigned n_alive = 0, n_acc = 0;       // this lane's attempts and accepted steps today (the counters are settled after the loop)
    bool gave_up_today = false;
    bool alive = true;
    // a member whose state is already non-finite is not integrated further
#pragma unroll
    for (int i = 0; i < NS; ++i) alive = alive && (sp_abs(y[i]) < sp_huge<R>());
    if (!alive) {
        cnt.poisoned = true;
#pragma unroll
        for (int i = 0; i < 4; ++i) yq[i] = __builtin_nan("");
    }
    // a lane without a member of its own (ragged last group, or a wave that carries fewer than 64 members on purpose) shadows a
    // valid slot for its addresses but must not keep the wave in the attempt loop
    alive = alive && lane_active;

    // Loop nest: the inner loop is one attempt per trip and has a single back edge; what happens only every RESYNC_EVERY-th
    // attempt (re-evaluating the auxiliary states) sits in the outer loop.  Same sequence of operations as one flat loop with
    // the resync inside -- but with the rare block as a second latch of the attempt loop the register allocator split the
    // loop-carried state across the back edge and paid ~40 register copies per attempt.
    bool any_alive = __any(alive);
    while (any_alive) {
    do {
        ++cnt.wave_trips;                 // one attempt issued for the whole wavefront, whoever still needs it
        const R rem = T - t;
        R hh = h;
        hh = (rem < (R)2 * h) ? (R)0.5 * rem : hh;      // (selects, not branches: the loop is long enough without them)
        hh = (rem <= c11 * h) ? rem : hh;
        const bool last_chance = (trip + 1 >= max_steps);
        if (last_chance) hh = rem;
        if constexpr (STIFF) {
            // (opts.stiff_pair) no attempt reaches further than SIMPLYP_STIFF_CAP relaxation times of the reach; the rate is the carried
            // state y[9] = cQ Qr**b_Q itself
            const R hcap = (R)SIMPLYP_STIFF_CAP * sp_rcp1(y[9]);
            hh = (!last_chance && hh > hcap) ? hcap : hh;
        }
        if (!alive) hh = 0;

        R k1[NS], k2[NS], k3[NS], k4[NS], k5[NS], k6[NS], kq[4], yt[NS];
        R sq[4], eq[4];                          // sum b_s kq_s, sum e_s kq_s
        // stage weights premultiplied by the step (per lane): one FMA per (component, earlier stage)
        SYS::f(y, c, k1, kq);
        if constexpr (!STIFF) {
#pragma unroll
        for (int i = 0; i < 4; ++i) { sq[i] = b1 * kq[i]; if (SYS::QUAD_IN_NORM) eq[i] = e1 * kq[i]; }
        }
        bool targeted = false;
        bool stiff = false;      // this attempt is taken by the stability-optimised pair (STIFF only)
        R kfac = (R)1;       // factor on the error estimate of a step that crosses a knee of a gate (SYS::KINK_AWARE)
        if constexpr (SYS::KINK_AWARE) {
            // times to the knees of the gates along the first slope (SYS::KNEE_*, KINK_*): soil boxes (hs), groundwater (hg)
            R hs = huge, hg = huge, ugd = 0;
            const R tlo = (R)SYS::KNEE_LO * hh;
#pragma unroll
            for (int i = 0; i < 3; ++i) {
                const R g = (i < 2) ? y[i] - c.fc : sp_fma(y[2], c.invTg, -c.Qgmin);
                const R sl = (i < 2) ? k1[i] : k1[2] * c.invTg;
                const R gd = (i < 2) ? c.dgate : c.dgq;
                const R r = sp_rcp_fast(sl);
                const R gdg = gd - g;
                const R t0 = ((R)0 - g) * r, t1 = gdg * r;
                const R tn = sp_min_raw(sp_if_above(t0, tlo), sp_if_above(t1, tlo));
                if (i < 2) hs = sp_min_raw(hs, tn); else { hg = tn; ugd = gdg; }
            }
            // a knee inside the step ends the step just past it
            const R hcut = sp_min_raw(hs, hg) * (R)SYS::KNEE_OVER;
            targeted = alive && !last_chance && hcut < (R)SYS::KNEE_HI * hh;
            hh = targeted ? hcut : hh;
            // a knee within KINK_REACH x the step along the first slope (a step that, by its first slope, stops just short of a
            // knee may well cross it) that the step was not aimed at: its estimate is not trusted.  A soil box that starts to
            // drain within the step can lift Vg through its gate within the same step, which the slope at the step's start
            // (dVg/dt = -Qg there) cannot see: a soil knee within reach while Vg / T_g is below the upper knee of its gate
            // (gd - g > 0) counts as a groundwater knee too.  A step aimed at a groundwater knee keeps KNEE_GW.
            const R look = (R)SYS::KINK_REACH * hh;
            const bool cs = hs < look, cg = hg < look;
            const bool gw = cg || (cs && ugd > (R)0);
            kfac = gw ? (targeted ? (R)SYS::KNEE_GW : (R)SYS::KINK_GW) : ((cs && !targeted) ? (R)SYS::KINK_SOIL : (R)1);
            // expansive reach: the estimate of a step that starts there is multiplied by EXPAND
            kfac = kfac * ((c.bQ * k1[3] > y[3] * y[9]) ? (R)SYS::EXPAND : (R)1);
            if constexpr (STIFF) {
                // Which pair takes this attempt (oracle: erk_aug_day): Cash-Karp unless the step is longer than SIMPLYP_STIFF_Z_ON relaxation
                // times of the reach -- then the second pair, except for an attempt with a knee within reach or aimed at one (what the knee
                // rules were tuned on is Cash-Karp's estimate): that one is shortened to Cash-Karp's stability interval instead.  The choice
                // depends on the lane's own state only.
                const bool longstep = hh * y[9] > (R)SIMPLYP_STIFF_Z_ON;
                const bool kneeish = cs || gw || targeted;
                stiff = longstep && !kneeish;
                const R hz = (R)SIMPLYP_STIFF_Z_ON * sp_rcp1(y[9]);
                hh = (longstep && kneeish && !last_chance && alive) ? hz : hh;
                toff = stiff ? TAB_STRIDE : 0;
#pragma unroll
                for (int i = 0; i < 4; ++i) sq[i] = (stiff ? (R)SIMPLYP_STIFF_B1 : b1) * kq[i];      // (first use of the row: see c21 below)
            }
        }
        // STIFF: this lane's row of the tableau is read from LDS ONE STAGE AHEAD of its use -- what stage s+1 needs is requested before
        // the right-hand side of stage s is evaluated (~60 fp64 instructions: the read's latency), and a scheduling barrier keeps the
        // request there.  Left to the compiler each group of reads sat 5-8 instructions before its s_waitcnt (it sinks them to their
        // uses to save registers): eight exposed LDS round trips per attempt, ~12 % of the network kernel's time
        // (profiles/r04_c4: 2.26 ns per instruction against 2.00 for the single-reach kernel at the same clock and mix).
#define SP_FETCHED() do { if constexpr (STIFF) __builtin_amdgcn_sched_barrier(0); } while (0)
        // (what is needed the moment the pair is chosen -- a21, and b1 above -- is a select between two literals, not a read)
        const R c21 = (STIFF && stiff) ? (R)SIMPLYP_STIFF_A21 : a21;
        const R c31 = TC(a31, TA31), c32 = TC(a32, TA32);
        SP_FETCHED();
        {
            const R h21 = hh * c21;
#pragma unroll
            for (int i = 0; i < NS; ++i) yt[i] = sp_fma(h21, k1[i], y[i]);
        }
        const R c41 = TC(a41, TA41), c42 = TC(a42, TA42), c43 = TC(a43, TA43), cb3 = TC(b3, TB3);
        SP_FETCHED();
        SYS::f(yt, c, k2, kq);
        {
            const R h31 = hh * c31, h32 = hh * c32;
#pragma unroll
            for (int i = 0; i < NS; ++i) yt[i] = sp_fma(h32, k2[i], sp_fma(h31, k1[i], y[i]));
        }
        const R c51 = TC(a51, TA51), c52 = TC(a52, TA52), c53 = TC(a53, TA53), c54 = TC(a54, TA54), cb4 = TC(b4, TB4);
        SP_FETCHED();
        SYS::f(yt, c, k3, kq);
#pragma unroll
        for (int i = 0; i < 4; ++i) { sq[i] = sp_fma(cb3, kq[i], sq[i]); if (SYS::QUAD_IN_NORM) eq[i] = sp_fma(e3, kq[i], eq[i]); }
        {
            const R h41 = hh * c41, h42 = hh * c42, h43 = hh * c43;
#pragma unroll
            for (int i = 0; i < NS; ++i)
                yt[i] = sp_fma(h43, k3[i], sp_fma(h42, k2[i], sp_fma(h41, k1[i], y[i])));
        }
        const R c61 = TC(a61, TA61), c62 = TC(a62, TA62), c63 = TC(a63, TA63), c64 = TC(a64, TA64), c65 = TC(a65, TA65);
        SP_FETCHED();
        SYS::f(yt, c, k4, kq);
#pragma unroll
        for (int i = 0; i < 4; ++i) { sq[i] = sp_fma(cb4, kq[i], sq[i]); if (SYS::QUAD_IN_NORM) eq[i] = sp_fma(e4, kq[i], eq[i]); }
        {
            const R h51 = hh * c51, h52 = hh * c52, h53 = hh * c53, h54 = hh * c54;
#pragma unroll
            for (int i = 0; i < NS; ++i)
                yt[i] = sp_fma(h54, k4[i], sp_fma(h53, k3[i], sp_fma(h52, k2[i], sp_fma(h51, k1[i], y[i]))));
        }
        SYS::f(yt, c, k5, kq);
#pragma unroll
        for (int i = 0; i < 4; ++i) if (SYS::QUAD_IN_NORM) eq[i] = sp_fma(e5, kq[i], eq[i]);
        {
            const R h61 = hh * c61, h62 = hh * c62, h63 = hh * c63, h64 = hh * c64, h65 = hh * c65;
#pragma unroll
            for (int i = 0; i < NS; ++i)
                yt[i] = sp_fma(h65, k5[i], sp_fma(h64, k4[i], sp_fma(h63, k3[i],
                        sp_fma(h62, k2[i], sp_fma(h61, k1[i], y[i])))));
        }
        // (the error weights: requested before the last right-hand side, which all six stage derivatives share the register file with --
        // the update weights only after it)
        const R ce1 = TC(e1, TE1), ce3 = TC(e3, TE3), ce4 = TC(e4, TE4), ce5 = TC(e5, TE5), ce6 = TC(e6, TE6);
        SP_FETCHED();
        SYS::f(yt, c, k6, kq);
        const R cb1 = TC(b1, TB1), cb3u = TC(b3, TB3), cb4u = TC(b4, TB4), cb6 = TC(b6, TB6);
        SP_FETCHED();
        if constexpr (!STIFF) {
#pragma unroll
        for (int i = 0; i < 4; ++i) { sq[i] = sp_fma(cb6, kq[i], sq[i]); if (SYS::QUAD_IN_NORM) eq[i] = sp_fma(e6, kq[i], eq[i]); }
        }

        // embedded error estimate and scaled error norm.  The scale uses the Euler predictor y + h k1 for "the new value":
        // the 5th-order increment itself is only formed once the step is accepted (weights premultiplied by the accept mask).
        R err = 0, chk = 0;
        R err_fast = 0;          // STIFF: the estimate of what the reach forgets at its relaxation rate (flow, three masses, Qr**k_M)
        R dq[4];
        const R he1 = hh * ce1, he3 = hh * ce3, he4 = hh * ce4, he5 = hh * ce5, he6 = hh * ce6;
#pragma unroll
        for (int i = 0; i < NS; ++i) {
            if (i < SYS::N_ERR) {
                const R he = sp_fma(he1, k1[i], sp_fma(he3, k3[i], sp_fma(he4, k4[i],
                                  sp_fma(he5, k5[i], he6 * k6[i]))));
                if (i >= NS - (NS == 11 ? 8 : 4) && i < NS - (NS == 11 ? 4 : 0)) chk += he;
                R ref = y[i];
                if constexpr (SYS::SOIL_REL) { if (i < 2) ref = y[i] - c.fc; }          // soil boxes: distance from field capacity
                const R pred = sp_fma(hh, k1[i], ref);                                  // Euler predictor of the step's end
                R w = sp_absmax(ref, pred);
                if constexpr (SYS::SOIL_REL) { if (i < 2) w = sp_max(w, c.dgate); }
                const R sc = sp_fma(rtol, w, atol);
                const R ri = sp_abs(he) * sp_rcp_fast(sc);
                if (STIFF && i >= 3) err_fast = sp_max(err_fast, ri); else err = sp_max(err, ri);
            }
        }
        if constexpr (SYS::KINK_AWARE) {
            // Qr**k_M (z[10]) in the norm, at AUX_WEIGHT x the tolerance
            const R he = sp_fma(he1, k1[10], sp_fma(he3, k3[10], sp_fma(he4, k4[10], sp_fma(he5, k5[10], he6 * k6[10]))));
            const R sc = sp_fma(rtol_aux, sp_absmax(y[10], sp_fma(hh, k1[10], y[10])), atol_aux);
            const R ra = sp_abs(he) * sp_rcp_fast(sc);
            if constexpr (STIFF) {
                // Damping-aware weights (include/simplyp_controller.h, SIMPLYP_DAMP_*; oracle: erk_aug_day): what the reach forgets at the
                // rate lam = rate - b_Q (dQr/dt) / Qr is allowed F x the tolerance, F = clamp(min(lam T / PHI, 1 + lam (T - t - h)), 1, FMAX).
                // (Same operations in the same order as ck_day_quad's: the two kernels agree bit for bit.)
                err_fast = sp_max(err_fast, ra);
                const R qd = (c.bQ * k1[3]) * sp_rcp_fast(y[3]);
                R lam = y[9] - qd;
                lam = sp_min(lam, y[9]);
                R F = sp_min((lam * T) * (R)(1.0 / SIMPLYP_DAMP_PHI), sp_fma(lam, rem - hh, (R)1));
                F = sp_min(sp_max(F, (R)1), (R)SIMPLYP_DAMP_FMAX);
                err = sp_max(err, err_fast * sp_rcp_fast(F));
            } else {
                err = sp_max(err, ra);
            }
            err *= kfac;
        }
        if constexpr (STIFF) {
            // (the last term of the daily integrals' sums: here, behind the error norm, so that the read of b6 is covered by it)
#pragma unroll
            for (int i = 0; i < 4; ++i) sq[i] = sp_fma(cb6, kq[i], sq[i]);
        }
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            dq[i] = hh * sq[i];
            if (SYS::QUAD_IN_NORM) {
                const R yq0 = (R)yq[i], yqn = yq0 + dq[i];
                const R sc = sp_fma(rtol, sp_absmax(yq0, yqn), atol);
                err = sp_max(err, sp_abs(hh * eq[i]) * sp_rcp_fast(sc));
            }
        }
        // v_max_f64 drops NaNs, so the finiteness test is on the increment itself
        const bool bad = !(err < huge) || !(sp_abs(chk) < huge);

        ++trip;
        // 0.9 err^(-1/5) in fp32 (v_log_f32 / v_exp_f32): a step-size factor needs no more.  err == 0 gives +inf -> 5;
        // err == inf gives 0 -> 0.2 (err is never NaN: v_max_f64 drops NaNs).
        // (raw v_log_f32 / v_exp_f32: the library versions add denormal-range fix-ups, ~12 instructions, for arguments
        // whose factor is clamped to 5 anyway)
        // (an attempt of the second pair carries a third-order estimate: exponent -1/4)
        const float fexp = (STIFF && stiff) ? (float)SIMPLYP_STIFF_ERR_EXP : -0.2f;
        float fac = (float)SIMPLYP_CTRL_SAFETY * __builtin_amdgcn_exp2f(fexp * __builtin_amdgcn_logf((float)err));
        fac = fminf(fmaxf(fac, (float)SIMPLYP_CTRL_FAC_MIN), (float)SIMPLYP_CTRL_FAC_MAX);
        const bool accept = alive && !bad && (err <= (R)1 || last_chance);
        bool give_up = false;
        n_alive += alive ? 1u : 0u;
        n_acc += accept ? 1u : 0u;
        if (last_chance && alive) cnt.capped = true;
        t = accept ? ((hh == rem) ? T : t + hh) : t;
        // State update in place, only for lanes that accepted: y += (m h b1) k1 + (m h b3) k3 + (m h b4) k4 + (m h b6) k6 with
        // m = 1 / 0.  0 * NaN would poison a lane that merely rejected a non-finite trial, so the rare wave that has such a
        // lane first zeroes that lane's stage derivatives (0 * 0 leaves its state alone); that is also where a lane gives up.
        // ONE update expression for every lane and every wave: a member's result does not depend on its wave mates, and the
        // loop-carried state has a single definition (two alternative update paths cost ~25 register copies per attempt).
        const bool any_bad = __any(alive && bad);
        if (any_bad) {
            give_up = alive && bad && (last_chance || hh <= (R)1.0e-9 * T);
            if (alive && bad) fac = (float)SIMPLYP_CTRL_FAC_MIN;
#pragma unroll
            for (int i = 0; i < NS; ++i) {
                k1[i] = bad ? (R)0 : k1[i]; k3[i] = bad ? (R)0 : k3[i]; k4[i] = bad ? (R)0 : k4[i]; k6[i] = bad ? (R)0 : k6[i];
            }
#pragma unroll
            for (int i = 0; i < 4; ++i) dq[i] = bad ? (R)0 : dq[i];
        }
        {
            const R m = accept ? hh : (R)0;
            const R hb1 = m * cb1, hb3 = m * cb3u, hb4 = m * cb4u, hb6 = m * cb6;
#pragma unroll
            for (int i = 0; i < NS; ++i) y[i] = sp_fma(hb6, k6[i], sp_fma(hb4, k4[i], sp_fma(hb3, k3[i], sp_fma(hb1, k1[i], y[i]))));
            const R mq = accept ? (R)1 : (R)0;
#pragma unroll
            for (int i = 0; i < 4; ++i) yq[i] = __builtin_fma((double)mq, (double)dq[i], yq[i]);
        }
        if (any_bad) {
            if (give_up) {
#pragma unroll
                for (int i = 0; i < NS; ++i) y[i] = (R)__builtin_nanf("");
#pragma unroll
                for (int i = 0; i < 4; ++i) yq[i] = __builtin_nan("");
                cnt.poisoned = true; gave_up_today = true;
            }
        }
        {
            // (a step that was cut to end at a knee and accepted does not shorten the step size carried on)
            const R hn = hh * (R)fac;
            h = alive ? ((targeted && accept && hn < h) ? h : hn) : h;
        }
        alive = alive && !give_up && (t < T);
        any_alive = __any(alive);
    } while (any_alive && (SYS::RESYNC_EVERY == 0 || (trip % SYS::RESYNC_EVERY) != 0));
        if (SYS::RESYNC_EVERY > 0) {
            // after every RESYNC_EVERY-th attempt of the day: a wave-uniform test (see `trip`), while the schedule stays a
            // function of the lane's own history -- results do not depend on which members share a wavefront
            if (any_alive && (trip % SYS::RESYNC_EVERY) == 0) {
                if (alive) SYS::resync(y, c);
            }
        }
    }
    // every attempt of a lane ends accepted, rejected, or -- at most once, then the lane is dead -- given up
    cnt.rhs += 6u * n_alive;
    cnt.steps += n_acc;
    cnt.rejected += n_alive - n_acc - (gave_up_today ? 1u : 0u);
    h_carry = (double)h;
#undef TC
#undef SP_FETCHED
}

// ---------------------------------------------------------------------------------------
// One member over FOUR lanes (opts.lanes_per_member = 4; scheme 2 only).
//
// An ensemble that cannot fill the chip's 1024 SIMDs even with one member per wave-lane-slot (BASELINE config C2: 1024
// members; a 100 000-member ensemble split over 8 GPUs: 12 500 each) runs at the latency of ONE member's serial chain of
// attempts, with 63 (or 51) of a wave's 64 lanes idle.  Here a member's Cash-Karp attempt is spread over the four lanes of a
// DPP quad, component-wise:
//
//      lane j of the quad      slot 0 (store)   slot 1 (carried function)     slot 2 (reach mass)   daily integral
//      0  soil box A           VsA              EA = exp(-mu VsA)             Msus                  Msus Qr/Vr
//      1  soil box S           VsS              ES = exp(-mu VsS)             TDPr                  TDPr Qr/Vr
//      2  groundwater          Vg               pk = Qr**k_M                  PPr                   PPr Qr/Vr
//      3  reach                Qr               pbc = cQ Qr**b_Q              (none: stays 0)       Qr
//
// so the stage sums, the error norm and the state update cost 3 components per lane instead of 11, and the right-hand side
// is evaluated "one sub-system per lane": the three smooth-step gates (two soil boxes, groundwater) are the same
// instructions on different operands, likewise the three mass balances and the four carried-function derivatives; what a
// lane needs from its quad mates (QsA, QsS, Qg, pbc, pk, the reach's dQr/Qr) travels by DPP quad_perm moves (two 32-bit
// moves per double, no LDS).  Per-lane coefficient tables (QuadConst) with 0 / 1 / -1 entries select each lane's formula
// through FMAs that are exact for those entries, so EVERY value is produced by the same IEEE operations in the same order as
// in SysAug::f / ck_day<SysAug>: results are bit-identical to the one-lane kernels (tested), step sequence included.
// 505 instructions per attempt instead of 747 (tools/isa_stats.py); measured 1.42 x (profiles/r02_experiments.md).

template <int CTRL>
__device__ __forceinline__ double quad_perm(double v)
{
    const long long b = __double_as_longlong(v);
    int lo = (int)(b & 0xffffffffll), hi = (int)(b >> 32);
    lo = __builtin_amdgcn_update_dpp(0, lo, CTRL, 0xf, 0xf, true);
    hi = __builtin_amdgcn_update_dpp(0, hi, CTRL, 0xf, 0xf, true);
    return __longlong_as_double(((long long)hi << 32) | (long long)(unsigned)lo);
}
// quad_perm control words: lane j of every quad reads lane sel_j of the same quad
#define SP_QP(s0, s1, s2, s3) ((s0) | ((s1) << 2) | ((s2) << 4) | ((s3) << 6))
template <int J> __device__ __forceinline__ double quad_bcast(double v) { return quad_perm<SP_QP(J, J, J, J)>(v); }

struct QuadConst {
    // gate of slot 0:  w = fma(x0, cw, w0);  s = clamp(fma(x0, csx, fma(w, csw, s0c)));  Q = fma(s s (3 - 2 s), w, q0)
    // (soil lanes: csx = 1/(0.01 fc), csw = 0; groundwater lane: csx = 0, csw = 1/(0.01 Qg_min) -- the FMA with the zero
    // coefficient returns its addend exactly, so each lane's s is the one-lane kernel's, and no select is needed)
    double cw, w0, csx, csw, s0c, q0;
    // slot-0 derivative:  t = fma(cA1, x1, fma(cA2, Qsum, fma(nG, Qg, cB)));  d0 = fma(nZ, x0, fma(nQ, Q, fma(mP, Qg, t))) * fma(mP, x1, m012)
    // (soil lanes: cA1 = alpha PET, cA2 = 0; the others: cA1 = 0, cA2 = beta | 1 - beta)
    double cA1, cA2, cB, nG, mP, nQ, nZ, m012;
    // slot-1 derivative:  d1 = (c1 x1) * rate
    double c1;
    // slot-2 derivative:  d2 = fma(mK, pk, fma(mA, QsA, fma(mS, QsS, fma(mG, Qg, m4)))) - x2 kap
    double mG, m4, mS, mA, mK;
    // error scale of slot 0: rtol * max(|x0 - eoff|, efloor) -- the soil lanes measure from field capacity (SysAug::SOIL_REL);
    // 0 / 0 on the other lanes, where x - 0 and max(., 0) change nothing
    double eoff, efloor;
    // gate argument of slot 0 for the knee test (SysAug::KINK_AWARE): g = fma(x0 - eoff, gs, g0), knees at 0 and gd.  Soil lanes:
    // the identity (gs = 1, g0 = -0.0) on Vs - fc, gd = 0.01 fc; groundwater lane: Vg / T_g - Qg_min, gd = 0.01 Qg_min; reach
    // lane: constant 1 (no gate).  auxm: 1 on the lane whose slot 1 is Qr**k_M (in the norm at AUX_WEIGHT x the tolerance), else 0
    double gs, g0, gd, auxm;
    int kbit, lowbit;    // what this lane reports to its quad: 1 = a soil knee crossed, 2 = a groundwater knee crossed (kbit);
                         // 4 = Vg / T_g starts below the upper knee of its gate (lowbit, groundwater lane only)
    // the same for every lane
    double fA, fS, invKvc;
};

__device__ __forceinline__ double sel4(int j, double a0, double a1, double a2, double a3)
{
    return j == 0 ? a0 : (j == 1 ? a1 : (j == 2 ? a2 : a3));
}

__device__ __forceinline__ QuadConst quad_const(const DayConst& c, int j)
{
    QuadConst k;
    k.cw = sel4(j, c.invTsA, c.invTsS, c.invTg, 0.0);
    k.w0 = sel4(j, c.wA0, c.wS0, -c.Qgmin, 0.0);
    k.csx = sel4(j, c.inv_d, c.inv_d, 0.0, 0.0);
    k.csw = sel4(j, 0.0, 0.0, c.inv_dg, 0.0);
    k.s0c = sel4(j, c.s0, c.s0, 0.0, 0.0);
    k.q0 = sel4(j, -0.0, -0.0, c.Qgmin, 0.0);       // fma(g, w, -0.0) == g * w, sign of zero included
    k.cA1 = sel4(j, c.aE, c.aE, 0.0, 0.0);
    k.cA2 = sel4(j, 0.0, 0.0, c.beta, c.omb);
    k.cB = sel4(j, c.c0m, c.c0m, 0.0, c.qin);
    k.nG = sel4(j, 0.0, 0.0, -1.0, 0.0);
    k.mP = sel4(j, 0.0, 0.0, 0.0, 1.0);
    k.nQ = sel4(j, -1.0, -1.0, 0.0, 0.0);
    k.nZ = sel4(j, 0.0, 0.0, 0.0, -1.0);
    k.m012 = sel4(j, 1.0, 1.0, 1.0, 0.0);
    k.c1 = sel4(j, -c.mu, -c.mu, c.kM, c.bQ);
    k.mG = sel4(j, 0.0, c.tg, 0.0, 0.0);
    k.m4 = sel4(j, c.MsusUS, c.tconst, c.PPrUS, 0.0);
    k.mS = sel4(j, 0.0, c.tS, 0.0, 0.0);
    k.mA = sel4(j, 0.0, c.tA, 0.0, 0.0);
    k.mK = sel4(j, c.Esum, 0.0, c.cPP, 0.0);
    k.eoff = sel4(j, c.fc, c.fc, 0.0, 0.0);
    k.efloor = sel4(j, c.dgate, c.dgate, 0.0, 0.0);
    k.gs = sel4(j, 1.0, 1.0, c.invTg, 0.0);
    k.g0 = sel4(j, -0.0, -0.0, -c.Qgmin, 1.0);
    k.gd = sel4(j, c.dgate, c.dgate, c.dgq, 0.0);
    k.auxm = sel4(j, 0.0, 0.0, 1.0, 0.0);
    k.kbit = j < 2 ? 1 : (j == 2 ? 2 : 0);
    k.lowbit = j == 2 ? 4 : 0;
    k.fA = c.fA; k.fS = c.fS; k.invKvc = c.invKvc;
    return k;
}

// SysAug::f, one sub-system per lane.  x = this lane's three slots, d = their derivatives, qv = this lane's daily integrand.
__device__ __forceinline__ void quad_rhs(const double (&x)[3], const QuadConst& k, const bool j_lt2, const bool j_eq2,
                                         double (&d)[3], double& qv)
{
    // smooth-step gate of the store in slot 0: QsA, QsS (:105, :109), Qg (:121-122); lane 3 gets 0
    const double w = __builtin_fma(x[0], k.cw, k.w0);
    double s = __builtin_fma(x[0], k.csx, __builtin_fma(w, k.csw, k.s0c));
    s = __builtin_fmin(__builtin_fmax(s, 0.0), 1.0);
    const double Q = __builtin_fma((s * s) * __builtin_fma(-2.0, s, 3.0), w, k.q0);
    const double QsA = quad_bcast<0>(Q), QsS = quad_bcast<1>(Q), Qg = quad_bcast<2>(Q);
    const double Qsum = __builtin_fma(k.fA, QsA, k.fS * QsS);
    // slot 0: dVsA, dVsS (:106, :110), dVg (:124), dQr (:127-130)
    const double t = __builtin_fma(k.cA1, x[1], __builtin_fma(k.cA2, Qsum, __builtin_fma(k.nG, Qg, k.cB)));
    const double u = __builtin_fma(k.mP, Qg, t);
    const double v = __builtin_fma(k.nZ, x[0], __builtin_fma(k.nQ, Q, u));
    d[0] = v * __builtin_fma(k.mP, x[1], k.m012);
    // slot 1: d exp(-mu Vs) = -mu E dVs;  d (cQ Qr**b_Q), d Qr**k_M = (b_Q | k_M) p dQr/Qr  -- the reach lane's dQr/Qr goes to lane 2
    const double rq = sp_rcp1(x[0]);
    const double rr = d[0] * (j_lt2 ? 1.0 : rq);
    const double rate = quad_perm<SP_QP(0, 1, 3, 3)>(rr);
    d[1] = (k.c1 * x[1]) * rate;
    // slot 2: the three reach masses (:141-145, :154-166, :171-178) against Qr/Vr = pbc invKvc
    const double pbc = quad_bcast<3>(x[1]), pk = quad_bcast<2>(x[1]);
    const double kap = pbc * k.invKvc;
    const double o = x[2] * kap;
    d[2] = __builtin_fma(k.mK, pk, __builtin_fma(k.mA, QsA, __builtin_fma(k.mS, QsS, __builtin_fma(k.mG, Qg, k.m4)))) - o;
    // daily integrands (:132, :147, :168, :180): Qr on the reach lane (whose slot 2 is empty: o == 0), the mass fluxes on the others
    qv = __builtin_fma(k.mP, x[0], o);
}

// ck_day<SysAug> for a quad.  z[7] (the physical states VsA VsS Vg Qr Msus TDPr PPr) and yq[4] hold the member's state on every
// lane of the quad, on entry and on return; aux1 is THIS lane's carried function for the day's start (lane 0: exp(-mu VsA),
// 1: exp(-mu VsS), 2: Qr**k_M, 3: cQ Qr**b_Q -- evaluated by run_slot at the day boundary, one exponential per lane).
template <bool STIFF = false>
__device__ __forceinline__ void ck_day_quad(double (&z)[7], const double aux1, double (&yq)[4], const DayConst& c,
                                            const double T, const double rtol, const double atol, int max_steps, double& h_carry,
                                            CkCounters& cnt, const bool lane_active, const int j, const double* s_tab = nullptr)
{
    typedef double R;
    const bool j_lt2 = j < 2, j_eq2 = j == 2, j_eq3 = j == 3;
    const QuadConst k = quad_const(c, j);
    const R rtol_aux = SysAug::AUX_WEIGHT * rtol, atol_aux = SysAug::AUX_WEIGHT * atol;
    R a21 = 1.0 / 5;
    R a31 = 3.0 / 40, a32 = 9.0 / 40;
    R a41 = 3.0 / 10, a42 = -9.0 / 10, a43 = 6.0 / 5;
    R a51 = -11.0 / 54, a52 = 5.0 / 2, a53 = -70.0 / 27, a54 = 35.0 / 27;
    R a61 = 1631.0 / 55296, a62 = 175.0 / 512, a63 = 575.0 / 13824, a64 = 44275.0 / 110592, a65 = 253.0 / 4096;
    R b1 = 37.0 / 378, b3 = 250.0 / 621, b4 = 125.0 / 594, b6 = 512.0 / 1771;
    R e1 = 37.0 / 378 - 2825.0 / 27648, e3 = 250.0 / 621 - 18575.0 / 48384,
      e4 = 125.0 / 594 - 13525.0 / 55296, e5 = -277.0 / 14336, e6 = 512.0 / 1771 - 1.0 / 4;
#define SP_KEEP_SCALAR(x) asm volatile("" : "+s"(x))
    if constexpr (!STIFF) {
    SP_KEEP_SCALAR(a21); SP_KEEP_SCALAR(a31); SP_KEEP_SCALAR(a32); SP_KEEP_SCALAR(a41); SP_KEEP_SCALAR(a42); SP_KEEP_SCALAR(a43);
    SP_KEEP_SCALAR(a51); SP_KEEP_SCALAR(a52); SP_KEEP_SCALAR(a53); SP_KEEP_SCALAR(a54);
    SP_KEEP_SCALAR(a61); SP_KEEP_SCALAR(a62); SP_KEEP_SCALAR(a63); SP_KEEP_SCALAR(a64); SP_KEEP_SCALAR(a65);
    SP_KEEP_SCALAR(b1); SP_KEEP_SCALAR(b3); SP_KEEP_SCALAR(b4); SP_KEEP_SCALAR(b6);
    SP_KEEP_SCALAR(e1); SP_KEEP_SCALAR(e3); SP_KEEP_SCALAR(e4); SP_KEEP_SCALAR(e5); SP_KEEP_SCALAR(e6);
    }
    R huge = sp_huge<R>(), c11 = (R)1.1;
    SP_KEEP_SCALAR(huge); SP_KEEP_SCALAR(c11);
#undef SP_KEEP_SCALAR
#define TC(name, idx) (STIFF ? s_tab[toff + (idx)] : name)
    int toff = 0;
    R t = 0, h = (R)h_carry * SysAug::DAY_START;
    if (!(h > (R)0) || h > T) h = T;
    if constexpr (STIFF) {
        // ck_day<SysAug, true>'s cap on the day's first step: the reach lane's carried function is the rate cQ Qr**b_Q
        const R h0 = quad_bcast<3>(SIMPLYP_STIFF_Z_START * sp_rcp1(aux1));
        h = (h > h0) ? h0 : h;
    }
    int trip = 0;
    unsigned n_alive = 0, n_acc = 0;
    bool gave_up_today = false;
    bool alive = true;
#pragma unroll
    for (int i = 0; i < 7; ++i) alive = alive && (sp_abs(z[i]) < sp_huge<R>());
    {
        // ... and the four carried functions, one per lane: a member whose state is already non-finite is not integrated further
        int dead = (sp_abs(aux1) < sp_huge<R>()) ? 0 : 1;
        dead |= __builtin_amdgcn_update_dpp(0, dead, SP_QP(1, 0, 3, 2), 0xf, 0xf, true);
        dead |= __builtin_amdgcn_update_dpp(0, dead, SP_QP(2, 3, 0, 1), 0xf, 0xf, true);
        alive = alive && (dead == 0);
    }
    if (!alive) {
        cnt.poisoned = true;
#pragma unroll
        for (int i = 0; i < 4; ++i) yq[i] = __builtin_nan("");
    }
    alive = alive && lane_active;

    // this lane's slots of the member's state, and its daily integral
    R y[3];
    y[0] = sel4(j, z[0], z[1], z[2], z[3]);
    y[1] = aux1;
    y[2] = sel4(j, z[4], z[5], z[6], 0.0);
    double yqv = sel4(j, yq[1], yq[2], yq[3], yq[0]);

    bool any_alive = __any(alive);
    while (any_alive) {
    do {
        ++cnt.wave_trips;
        const R rem = T - t;
        R hh = h;
        hh = (rem < (R)2 * h) ? (R)0.5 * rem : hh;
        hh = (rem <= c11 * h) ? rem : hh;
        const bool last_chance = (trip + 1 >= max_steps);
        if (last_chance) hh = rem;
        if constexpr (STIFF) {
            // ck_day<SysAug, true>'s cap: the reach lane holds the rate cQ Qr**b_Q (its slot 1)
            const R hcap = quad_bcast<3>(SIMPLYP_STIFF_CAP * sp_rcp1(y[1]));
            hh = (!last_chance && hh > hcap) ? hcap : hh;
        }
        if (!alive) hh = 0;

        R k1[3], k2[3], k3[3], k4[3], k5[3], k6[3], kq, yt[3];
        R sq;
        quad_rhs(y, k, j_lt2, j_eq2, k1, kq);
        if constexpr (!STIFF) sq = b1 * kq;
        bool targeted = false;
        bool kneeish = false, stiff = false;
        R kfac = 1.0;
        // Knee logic of ck_day<SysAug>, with a wave-uniform shortcut in front.  This lane's gate argument moves along the first
        // slope as g(t) = g + sl t; it has a knee (g = 0 or g = gd) at a time in (a, b) exactly when g(a) and g(b), or g(a) - gd
        // and g(b) - gd, differ in sign -- four FMAs and two sign tests, no reciprocal.  With (a, b) = (0.9 KNEE_LO hh,
        // 1.1 KINK_REACH hh), a window 10 % wider on both sides than everything the logic below looks at (its times carry the
        // 5e-8 of the hardware reciprocal), "no lane of the wave sees a knee in the window" implies that the full logic would
        // leave the step alone (no targeting, factor 1): skipping it then changes nothing, bit for bit.  The waves this kernel
        // exists for carry 1 to 16 members and most of their attempts are far from every knee (a 64-member wave of the one-lane
        // kernel nearly always has some lane near one: no shortcut there).
        const R g = sp_fma(y[0] - k.eoff, k.gs, k.g0), sl = k1[0] * k.gs;
        const R gdg = k.gd - g;
        bool near_knee;
        {
            const R ta = (0.9 * SysAug::KNEE_LO) * hh, tb = (1.1 * SysAug::KINK_REACH) * hh;
            const R e0 = sp_fma(sl, ta, g), e1 = sp_fma(sl, tb, g);
            const R f0 = sp_fma(sl, ta, -gdg), f1 = sp_fma(sl, tb, -gdg);
            near_knee = (sp_sign_xor(e0, e1) | sp_sign_xor(f0, f1)) < 0;
        }
        if (__any(near_knee)) {
            // times to the knees along the first slope, as in ck_day<SysAug>: this lane's gate, then the quad's minimum
            const R tlo = SysAug::KNEE_LO * hh;
            const R r = sp_rcp_fast(sl);
            const R t0 = (0.0 - g) * r, t1 = gdg * r;
            const R tn = sp_min_raw(sp_if_above(t0, tlo), sp_if_above(t1, tlo));
            R hk = sp_min_raw(huge, tn);
            hk = sp_min_raw(hk, quad_perm<SP_QP(1, 0, 3, 2)>(hk));
            hk = sp_min_raw(hk, quad_perm<SP_QP(2, 3, 0, 1)>(hk));
            const R hcut = hk * SysAug::KNEE_OVER;
            targeted = alive && !last_chance && hcut < SysAug::KNEE_HI * hh;
            hh = targeted ? hcut : hh;
            // what the lanes of the quad see within KINK_REACH x the step (1: a soil knee, 2: a groundwater knee, 4: Vg / T_g
            // below the upper knee of its gate), OR-ed; then the one-lane kernel's choice of factor
            const R look = SysAug::KINK_REACH * hh;
            int kink = (tn < look ? k.kbit : 0) | (gdg > 0.0 ? k.lowbit : 0);
            kink |= __builtin_amdgcn_update_dpp(0, kink, SP_QP(1, 0, 3, 2), 0xf, 0xf, true);
            kink |= __builtin_amdgcn_update_dpp(0, kink, SP_QP(2, 3, 0, 1), 0xf, 0xf, true);
            const bool gw = (kink & 2) != 0 || (kink & 5) == 5;
            kfac = gw ? (targeted ? SysAug::KNEE_GW : SysAug::KINK_GW) : (((kink & 1) && !targeted) ? SysAug::KINK_SOIL : 1.0);
            kneeish = (kink & 1) != 0 || gw || targeted;
        }
        {
            // expansive reach (SysAug::EXPAND): the reach lane holds Qr, cQ Qr**b_Q and the first slope of Qr; its verdict goes to the quad
            const int expanding = __builtin_amdgcn_update_dpp(0, (k.c1 * k1[0] > y[0] * y[1]) ? 1 : 0, SP_QP(3, 3, 3, 3), 0xf, 0xf, true);
            kfac = kfac * (expanding ? SysAug::EXPAND : 1.0);
        }
        if constexpr (STIFF) {
            // ck_day<SysAug, true>'s choice of the pair, from the reach lane's rate (same operations: bit-identical decisions)
            const bool longstep = quad_bcast<3>(hh * y[1]) > SIMPLYP_STIFF_Z_ON;
            stiff = longstep && !kneeish;
            const R hz = quad_bcast<3>(SIMPLYP_STIFF_Z_ON * sp_rcp1(y[1]));
            hh = (longstep && kneeish && !last_chance && alive) ? hz : hh;
            toff = stiff ? TAB_STRIDE : 0;
            sq = (stiff ? (R)SIMPLYP_STIFF_B1 : b1) * kq;
        }
        // (STIFF: the tableau row is read from LDS one stage ahead of its use, as in ck_day)
#define SP_FETCHED() do { if constexpr (STIFF) __builtin_amdgcn_sched_barrier(0); } while (0)
        const R c21 = (STIFF && stiff) ? (R)SIMPLYP_STIFF_A21 : a21;
        const R c31 = TC(a31, TA31), c32 = TC(a32, TA32);
        SP_FETCHED();
        {
            const R h21 = hh * c21;
#pragma unroll
            for (int i = 0; i < 3; ++i) yt[i] = sp_fma(h21, k1[i], y[i]);
        }
        const R c41 = TC(a41, TA41), c42 = TC(a42, TA42), c43 = TC(a43, TA43), cb3 = TC(b3, TB3);
        SP_FETCHED();
        quad_rhs(yt, k, j_lt2, j_eq2, k2, kq);
        {
            const R h31 = hh * c31, h32 = hh * c32;
#pragma unroll
            for (int i = 0; i < 3; ++i) yt[i] = sp_fma(h32, k2[i], sp_fma(h31, k1[i], y[i]));
        }
        const R c51 = TC(a51, TA51), c52 = TC(a52, TA52), c53 = TC(a53, TA53), c54 = TC(a54, TA54), cb4 = TC(b4, TB4);
        SP_FETCHED();
        quad_rhs(yt, k, j_lt2, j_eq2, k3, kq);
        sq = sp_fma(cb3, kq, sq);
        {
            const R h41 = hh * c41, h42 = hh * c42, h43 = hh * c43;
#pragma unroll
            for (int i = 0; i < 3; ++i)
                yt[i] = sp_fma(h43, k3[i], sp_fma(h42, k2[i], sp_fma(h41, k1[i], y[i])));
        }
        const R c61 = TC(a61, TA61), c62 = TC(a62, TA62), c63 = TC(a63, TA63), c64 = TC(a64, TA64), c65 = TC(a65, TA65);
        SP_FETCHED();
        quad_rhs(yt, k, j_lt2, j_eq2, k4, kq);
        sq = sp_fma(cb4, kq, sq);
        {
            const R h51 = hh * c51, h52 = hh * c52, h53 = hh * c53, h54 = hh * c54;
#pragma unroll
            for (int i = 0; i < 3; ++i)
                yt[i] = sp_fma(h54, k4[i], sp_fma(h53, k3[i], sp_fma(h52, k2[i], sp_fma(h51, k1[i], y[i]))));
        }
        const R ce1 = TC(e1, TE1), ce3 = TC(e3, TE3), ce4 = TC(e4, TE4), ce5 = TC(e5, TE5), ce6 = TC(e6, TE6);
        SP_FETCHED();
        quad_rhs(yt, k, j_lt2, j_eq2, k5, kq);
        {
            const R h61 = hh * c61, h62 = hh * c62, h63 = hh * c63, h64 = hh * c64, h65 = hh * c65;
#pragma unroll
            for (int i = 0; i < 3; ++i)
                yt[i] = sp_fma(h65, k5[i], sp_fma(h64, k4[i], sp_fma(h63, k3[i],
                        sp_fma(h62, k2[i], sp_fma(h61, k1[i], y[i])))));
        }
        const R cb1 = TC(b1, TB1), cb3u = TC(b3, TB3), cb4u = TC(b4, TB4), cb6 = TC(b6, TB6);
        SP_FETCHED();
        quad_rhs(yt, k, j_lt2, j_eq2, k6, kq);
        sq = sp_fma(cb6, kq, sq);

        // error norm over the 7 physical states = slots 0 and 2 of the quad (lane 3's slot 2 is identically 0 and adds
        // nothing); the maximum over the quad is exact in any order.  The finiteness test looks at the increments of the reach
        // states (Qr, Msus, TDPr, PPr), as ck_day<SysAug>'s `chk` does: a non-finite one makes the lane's error infinite.
        const R he1 = hh * ce1, he3 = hh * ce3, he4 = hh * ce4, he5 = hh * ce5, he6 = hh * ce6;
        R err = 0, err_fast = 0;
        R he_s[3];
#pragma unroll
        for (int i = 0; i < 3; i += 2) {
            const R he = sp_fma(he1, k1[i], sp_fma(he3, k3[i], sp_fma(he4, k4[i], sp_fma(he5, k5[i], he6 * k6[i]))));
            he_s[i] = he;
            // slot 0 of the soil lanes is measured from field capacity, like ck_day<SysAug> does for components 0 and 1
            const R ref = (i == 0) ? y[0] - k.eoff : y[i];
            const R pred = sp_fma(hh, k1[i], ref);
            R w = sp_absmax(ref, pred);
            if (i == 0) w = sp_max(w, k.efloor);
            const R sc = sp_fma(rtol, w, atol);
            const R ri = sp_abs(he) * sp_rcp_fast(sc);
            // (STIFF: slot 0 of the reach lane -- Qr -- and slot 2 -- the masses -- are what the reach forgets: see below)
            if (STIFF && (i == 2 || j_eq3)) err_fast = sp_max(err_fast, ri); else err = sp_max(err, ri);
        }
        {
            // Qr**k_M (slot 1 of the groundwater lane) in the norm at AUX_WEIGHT x the tolerance; x 0 on the other lanes
            const R he = sp_fma(he1, k1[1], sp_fma(he3, k3[1], sp_fma(he4, k4[1], sp_fma(he5, k5[1], he6 * k6[1]))));
            const R sc = sp_fma(rtol_aux, sp_absmax(y[1], sp_fma(hh, k1[1], y[1])), atol_aux);
            const R ra = (sp_abs(he) * sp_rcp_fast(sc)) * k.auxm;
            if constexpr (STIFF) {
                // ck_day<SysAug, true>'s damping-aware weights: the factor from the reach lane's Qr (slot 0), rate (slot 1) and dQr/dt
                err_fast = sp_max(err_fast, ra);
                const R qd = (c.bQ * k1[0]) * sp_rcp_fast(y[0]);
                R lam = y[1] - qd;
                lam = sp_min(lam, y[1]);
                R F = sp_min((lam * T) * (R)(1.0 / SIMPLYP_DAMP_PHI), sp_fma(lam, rem - hh, (R)1));
                F = sp_min(sp_max(F, (R)1), (R)SIMPLYP_DAMP_FMAX);
                const R G = quad_bcast<3>(sp_rcp_fast(F));
                err = sp_max(err, err_fast * G);
            } else {
                err = sp_max(err, ra);
            }
        }
        {
            const R chk = j_eq3 ? he_s[0] : he_s[2];
            if (!(sp_abs(chk) < huge)) err = __builtin_inf();
        }
        err = sp_max(err, quad_perm<SP_QP(1, 0, 3, 2)>(err));
        err = sp_max(err, quad_perm<SP_QP(2, 3, 0, 1)>(err));
        err *= kfac;
        R dq = hh * sq;
        const bool bad = !(err < huge);

        ++trip;
        const float fexp = (STIFF && stiff) ? (float)SIMPLYP_STIFF_ERR_EXP : -0.2f;
        float fac = (float)SIMPLYP_CTRL_SAFETY * __builtin_amdgcn_exp2f(fexp * __builtin_amdgcn_logf((float)err));
        fac = fminf(fmaxf(fac, (float)SIMPLYP_CTRL_FAC_MIN), (float)SIMPLYP_CTRL_FAC_MAX);
        const bool accept = alive && !bad && (err <= (R)1 || last_chance);
        bool give_up = false;
        n_alive += alive ? 1u : 0u;
        n_acc += accept ? 1u : 0u;
        if (last_chance && alive) cnt.capped = true;
        t = accept ? ((hh == rem) ? T : t + hh) : t;
        const bool any_bad = __any(alive && bad);
        if (any_bad) {
            give_up = alive && bad && (last_chance || hh <= (R)1.0e-9 * T);
            if (alive && bad) fac = (float)SIMPLYP_CTRL_FAC_MIN;
#pragma unroll
            for (int i = 0; i < 3; ++i) {
                k1[i] = bad ? (R)0 : k1[i]; k3[i] = bad ? (R)0 : k3[i]; k4[i] = bad ? (R)0 : k4[i]; k6[i] = bad ? (R)0 : k6[i];
            }
            dq = bad ? (R)0 : dq;
        }
        {
            const R m = accept ? hh : (R)0;
            const R hb1 = m * cb1, hb3 = m * cb3u, hb4 = m * cb4u, hb6 = m * cb6;
#pragma unroll
            for (int i = 0; i < 3; ++i) y[i] = sp_fma(hb6, k6[i], sp_fma(hb4, k4[i], sp_fma(hb3, k3[i], sp_fma(hb1, k1[i], y[i]))));
            const R mq = accept ? (R)1 : (R)0;
            yqv = __builtin_fma(mq, dq, yqv);
        }
        if (any_bad) {
            if (give_up) {
#pragma unroll
                for (int i = 0; i < 3; ++i) y[i] = (R)__builtin_nanf("");
                yqv = __builtin_nan("");
                cnt.poisoned = true; gave_up_today = true;
            }
        }
        {
            const R hn = hh * (R)fac;
            h = alive ? ((targeted && accept && hn < h) ? h : hn) : h;
        }
        alive = alive && !give_up && (t < T);
        any_alive = __any(alive);
    } while (any_alive && (trip % SysAug::RESYNC_EVERY) != 0);
        // SysAug::resync: pbc (lane 3) and pk (lane 2) from the reach lane's Qr, after every RESYNC_EVERY-th attempt
        if (any_alive && (trip % SysAug::RESYNC_EVERY) == 0) {
            const double Qr = quad_bcast<3>(y[0]);
            const double lq = sp_log(Qr);
            const double p = sp_exp((j_eq3 ? c.bQ : c.kM) * lq);
            const double y1 = j_eq3 ? c.cQ * p : (j_eq2 ? p : y[1]);
            y[1] = alive ? y1 : y[1];
        }
    }
    cnt.rhs += 6u * n_alive;
    cnt.steps += n_acc;
    cnt.rejected += n_alive - n_acc - (gave_up_today ? 1u : 0u);
    h_carry = (double)h;

    // the member's state back on every lane of the quad (the carried functions are re-evaluated at the day boundary)
    z[0] = quad_bcast<0>(y[0]); z[1] = quad_bcast<1>(y[0]); z[2] = quad_bcast<2>(y[0]); z[3] = quad_bcast<3>(y[0]);
    z[4] = quad_bcast<0>(y[2]); z[5] = quad_bcast<1>(y[2]); z[6] = quad_bcast<2>(y[2]);
    yq[0] = quad_bcast<3>(yqv); yq[1] = quad_bcast<0>(yqv); yq[2] = quad_bcast<1>(yqv); yq[3] = quad_bcast<2>(yqv);
#undef TC
#undef SP_FETCHED
}
#undef SP_QP

// ---------------------------------------------------------------------------------------

__device__ __forceinline__ int nc_type_of(double f_NC_A, double f_NC_S)     // model.py:325-334
{
    return f_NC_A > 0.0 ? 1 : (f_NC_S > 0.0 ? 2 : 0);
}

// discretized_soilP (model.py:39-56) followed by the >= 0 clamps (:696-699) and the soil-water concentration (:702-703), in two
// phases around the day boundary's one group of exponentials (sp_expn): soil_p_rate gives b (:43), whose exp(-b) the caller
// evaluates together with the day's other exponentials; soil_p_update does the rest.  The reference's nine divisions per call
// (by b, by Vs, by b*Vs) share two reciprocals, 1/Vs and 1/b (sp_rcp: hardware seed + two Newton steps, correctly rounded on
// every sample tried, tools/micro/rcp_accuracy.hip): a * (1/b) instead of a / b differs in the last bit at most -- eleven orders
// of magnitude below the parity bar -- and an IEEE division is ~12 instructions on this chip, a quarter of the day-boundary
// code before round 3.  Vs == 0 (never met with water in the soil box, but the reference defines it): b = inf there, so the
// reference gets TDPs = 0 + (TDPs - 0) * 0, sorp = 0 (:50), Plab unchanged and conc_TDPs = 0/0 = NaN; the reciprocals of 0 are
// NaN after their Newton steps, so that case is one select on TDPs (sorp is guarded as in the reference; conc = TDPs * NaN).
struct SoilPRate { double rVs, b; };

__device__ __forceinline__ SoilPRate soil_p_rate(double KfMsoil, double Qs, double Qq, double Vs)
{
    SoilPRate r;
    r.rVs = sp_rcp(Vs);
    r.b = (KfMsoil + Qs + Qq) * r.rVs;                                            // :43
    return r;
}

// aP = P_netInput * A_catch * 100 / 365 (:42, constant over the run); emb = exp(max(-b, -700))
__device__ __forceinline__ void soil_p_update(double aP, double KfMsoil, double EPC0, double Vs, const SoilPRate& r, double emb,
                                              double& TDPs, double& Plab, double& conc)
{
    const double a = aP + KfMsoil * EPC0;                                         // :42
    const double rb = sp_rcp(r.b);
    const double aob = a * rb;
    double T = aob + (TDPs - aob) * emb;                                          // :44
    T = (Vs == 0.0) ? (TDPs - 0.0) * 0.0 : T;                                     // b = inf: a/b = 0, exp(-b) = 0
    double sorp = 0.0;
    const double aobv = aob * r.rVs;                                              // a / (b Vs) = a / b0, :47
    if (Vs > 0.0)                                                                 // :50
        sorp = KfMsoil * (aobv - EPC0 + (rb * (T * r.rVs - aobv)) * (1.0 - emb));     // :51
    double Pl = Plab + sorp;                                                      // :54
    TDPs = (0.0 > T) ? 0.0 : T;                                                   // :696 (NaN stays NaN, like Python max)
    Plab = (0.0 > Pl) ? 0.0 : Pl;                                                 // :697
    conc = TDPs * r.rVs;                                                          // :702-703
}

// Everything one lane does for one member over days [d_begin, d_end) of every reach of its chain.
// `ckpt` ([CKPT_N][E], slot-major) carries a member's state across a time-chunk boundary for the queue
// kernel (single-reach problems); the chain kernel passes nullptr and runs all days of every reach.
constexpr int CKPT_N = 16;    // y[8], Plab_A, TDPs_A, Plab_NC, TDPs_NC, conc_A, conc_NC, h_carry, snow depth

template <int INTEG, bool SNOW, int TEAM, bool STIFF = false>
__device__ __forceinline__ void run_slot(const KernelArgs& a, double* s_P, double* s_E, double* s_T, int* s_doy,
                                         const int lane, const int slot_raw, const int* reaches, const int n_reaches,
                                         const int d_begin, const int d_end, double* ckpt, const double* s_tab = nullptr)
{
    static_assert(!STIFF || INTEG == SIMPLYP_INTEG_CASHKARP_AUG, "the second pair (opts.stiff_pair) exists for integrator 2 only");
    static_assert(TEAM == 1 || (TEAM == 4 && INTEG == SIMPLYP_INTEG_CASHKARP_AUG), "four lanes per member: scheme 2 only");
    const bool active = slot_raw < a.E;
    // TEAM == 4: the four lanes of a quad share a member slot.  Everything outside the day's integration (day constants, soil P,
    // end-of-day flows) is computed by all four alike -- same addresses, same values; one of them (`writer`) stores and counts.
    const bool writer = active && (TEAM == 1 || (lane & (TEAM - 1)) == 0);
    const int slot = active ? slot_raw : a.E - 1;     // inactive lanes shadow the last slot, store nothing
    // Load balancing (host side, simplyp_hip.hip): lane slots may be handed members in order of expected
    // cost.  Parameters, outputs and status are addressed by member id e (a gather / scatter of 8-byte
    // words, a few per day against ~1e5 cycles of arithmetic); the routing series stay in slot order.
    const int e = a.perm ? a.perm[slot] : slot;
    const size_t E = (size_t)a.E;
    const int S = a.S, D = a.D;
    const size_t Dst = (size_t)a.D_stride;            // days per forcing row (>= D: the pilot run uses fewer days)

    const int pe = a.params_by_slot ? slot : e;       // column of the parameter arrays
#define MPv(idx) (a.mp[(size_t)(idx) * E + pe])
#define RPv(idx, s) (a.rp[((size_t)(idx) * S + (s)) * E + pe])

    const int set = a.forcing_of_member ? a.forcing_of_member[pe] : 0;
    const bool shared_forcing = (a.forcing_of_member == nullptr);
    // Rows of this member's forcing set: P (Precipitation when SNOW), PET, T_air (SNOW only).  The set may differ per lane, so
    // these are per-lane 64-bit pointers; left to the compiler they are hoisted out of the day loop and held in 6 VGPRs across the
    // attempt loop (the snow instantiation of the chain kernel then spilled 20 B per lane to scratch).  Formed from the set number
    // where they are used instead -- once per 256-day tile, or per day for per-member forcing -- behind an opaque copy of it.
#define SP_FORCING_ROWS \
    int set_o = set; asm volatile("" : "+v"(set_o)); \
    const double* Pser = a.forcing + (size_t)set_o * (SNOW ? 3 : 2) * Dst; \
    const double* Eser = Pser + Dst; \
    const double* Tser = Eser + Dst; (void)Tser;

    // ---- member constants (model.py:349-361, 377-390) ----
    const double fc = MPv(SIMPLYP_PM_FC), f_quick = MPv(SIMPLYP_PM_F_QUICK), alpha = MPv(SIMPLYP_PM_ALPHA);
    const double beta = MPv(SIMPLYP_PM_BETA), T_g = MPv(SIMPLYP_PM_T_G), Qg_min = MPv(SIMPLYP_PM_QG_MIN);
    const double a_Q = MPv(SIMPLYP_PM_A_Q), b_Q = MPv(SIMPLYP_PM_B_Q), k_M = MPv(SIMPLYP_PM_K_M);
    const double T_s_A = MPv(SIMPLYP_PM_T_S_A), T_s_S = MPv(SIMPLYP_PM_T_S_S);
    const double mu = -log(0.01) / fc;                                                            // :349
    const double Qr_init = MPv(SIMPLYP_PM_QR0_INIT) * 86400 / (1000 * RPv(SIMPLYP_PR_A_CATCH, a.sc_qr0));   // :386

    // NC type of the last sub-catchment: the reference's leaked loop variable (:330-334 -> :442,:676)
    const int nc_leak = nc_type_of(RPv(SIMPLYP_PR_F_AR, S - 1) * RPv(SIMPLYP_PR_F_NC_AR, S - 1)
                                   + RPv(SIMPLYP_PR_F_NC_IG, S - 1) * RPv(SIMPLYP_PR_F_IG, S - 1),
                                   RPv(SIMPLYP_PR_F_NC_S, S - 1));

    unsigned n_rhs = 0, n_steps = 0, n_rej = 0, n_trips = 0;
    int stat = 0;

    const size_t RD = (size_t)a.route_days;           // rows of a routing series buffer
    for (int ci = 0; ci < n_reaches; ++ci) {
        const int s = reaches[ci];

        // ---- per-reach constants and initial conditions (model.py:377-463) ----
        const double A_catch = RPv(SIMPLYP_PR_A_CATCH, s);
        const double f_Ar = RPv(SIMPLYP_PR_F_AR, s), f_IG = RPv(SIMPLYP_PR_F_IG, s), f_S = RPv(SIMPLYP_PR_F_S, s);
        const double f_NC_Ar = RPv(SIMPLYP_PR_F_NC_AR, s), f_NC_IG = RPv(SIMPLYP_PR_F_NC_IG, s);
        const double f_NC_S = RPv(SIMPLYP_PR_F_NC_S, s);
        const double f_A = f_IG + f_Ar;                                                           // :318
        const double f_NC_A = (f_Ar * f_NC_Ar) + (f_NC_IG * f_IG);                                // :319
        const int nc_type = nc_type_of(f_NC_A, f_NC_S);
        const double L_reach = RPv(SIMPLYP_PR_L_REACH, s);
        const double Msoil = MPv(SIMPLYP_PM_MSOIL_M2) * 1000000 * A_catch;                        // :404
        const double P_inactive = 1e-6 * MPv(SIMPLYP_PM_SOILPCONC_S) * Msoil;                     // :407
        const double EPC0_0_A = MPv(SIMPLYP_PM_EPC0_INIT_A) * A_catch;                            // :412
        const double EPC0_0_S = MPv(SIMPLYP_PM_EPC0_INIT_S) * A_catch;
        const double Kf = a.run_mode_cal
            ? 1e-6 * (MPv(SIMPLYP_PM_SOILPCONC_A) - MPv(SIMPLYP_PM_SOILPCONC_S)) / EPC0_0_A       // :451
            : MPv(SIMPLYP_PM_KF);                                                                 // :453
        const double KfMsoil = Kf * Msoil;
        double TDPeff = RPv(SIMPLYP_PR_TDPEFF, s);
        if (TDPeff != TDPeff) TDPeff = 0.0;                                                       // :462-463
        const double Kv = L_reach / (a_Q * 8.64 * 10000);     // Vr = Kv Qr^(1-b_Q)              // :457-459

        double y[8];
        y[0] = fc; y[1] = fc;                                                                     // :377-378
        y[4] = Qr_init;
        y[2] = (beta * Qr_init) * T_g;                                                            // :389-390
        y[3] = L_reach / (a_Q * pow(Qr_init, b_Q) * 8.64 * 10000) * Qr_init;                      // :457-459
        y[5] = 0.0; y[6] = 0.0; y[7] = 0.0;                                                       // :396
        double Plab_A = 1e-6 * (MPv(SIMPLYP_PM_SOILPCONC_A) - MPv(SIMPLYP_PM_SOILPCONC_S)) * Msoil;   // :415,:426
        double TDPs_A = EPC0_0_A * fc;                                                            // :420
        double Plab_NC = (nc_type == 2) ? Plab_A : 0.0;                                           // :429-434
        double TDPs_NC = (nc_type == 2) ? TDPs_A : 0.0;
        double conc_A = TDPs_A / fc;                                                              // :438
        double conc_NC = TDPs_NC / fc;                                                            // :446 (VsA0 == VsS0)
        double h_carry = a.step_len / (double)(a.substeps > 0 ? a.substeps : 1);
        double D_snow = SNOW ? MPv(SIMPLYP_PM_D_SNOW_0) : 0.0;                                    // inputs.py:198
        if (ckpt && d_begin > 0) {      // resume from the previous time chunk (written by whichever wave ran it)
            const double* k = ckpt + slot;
#pragma unroll
            for (int i = 0; i < 8; ++i) y[i] = k[(size_t)i * E];
            Plab_A = k[(size_t)8 * E]; TDPs_A = k[(size_t)9 * E]; Plab_NC = k[(size_t)10 * E]; TDPs_NC = k[(size_t)11 * E];
            conc_A = k[(size_t)12 * E]; conc_NC = k[(size_t)13 * E]; h_carry = k[(size_t)14 * E];
            if (SNOW) D_snow = k[(size_t)15 * E];
        }

        // per-reach pieces of the day constants
        DayConst c;
        c.mu = mu; c.fc = fc; c.inv_d = 1.0 / (0.01 * fc);
        c.invTsA = 1.0 / T_s_A; c.invTsS = 1.0 / T_s_S;
        c.invTg = 1.0 / T_g; c.Qgmin = Qg_min;
        c.inv_dg = (Qg_min * 0.01 > 0.0) ? 1.0 / (Qg_min * 0.01) : 1.0e300;   // threshold 0 -> plain step
        c.beta = beta; c.fA = f_A; c.fS = f_S;
        c.omb = 1.0 - beta;
        c.cQ = a_Q * (8.64 * 10000) / ((1 - b_Q) * L_reach);
        c.bQ = b_Q; c.kM = k_M;
        c.tg = MPv(SIMPLYP_PM_TDPG) * A_catch;                                                    // :163
        c.invKv = 1.0 / Kv;
        c.wA0 = -fc * c.invTsA; c.wS0 = -fc * c.invTsS; c.s0 = -fc * c.inv_d;
        c.invKvc = c.invKv / c.cQ;
        c.dgate = 0.01 * fc;
        c.dgq = 0.01 * Qg_min;

        const double slopeA = RPv(SIMPLYP_PR_S_AR, s), slopeIG = RPv(SIMPLYP_PR_S_IG, s), slopeS = RPv(SIMPLYP_PR_S_SN, s);
        const double ES = MPv(SIMPLYP_PM_E_M) * RPv(SIMPLYP_PR_S_REACH, s);
        const double C_cover_A0 = MPv(SIMPLYP_PM_C_COVER_A);
        const double Esus_S = ES * slopeS * MPv(SIMPLYP_PM_C_COVER_S) * (1 - MPv(SIMPLYP_PM_C_MEAS_S));      // :591-594
        const double Esus_IG = ES * slopeIG * MPv(SIMPLYP_PM_C_COVER_IG) * (1 - MPv(SIMPLYP_PM_C_MEAS_IG));
        const double EA_fac = ES * slopeA * (1 - MPv(SIMPLYP_PM_C_MEAS_A));
        const double f_spr = RPv(SIMPLYP_PR_F_SPR, s);
        const double dmid0 = MPv(SIMPLYP_PM_D_MAXE_SPR), dmid1 = MPv(SIMPLYP_PM_D_MAXE_AUT);
        const double C_out = C_cover_A0 - (60.0 * (1 - C_cover_A0) / (2 * (365 - 60.0)));         // :575-576
        const double E_PP = MPv(SIMPLYP_PM_E_PP);
        const double wA = f_A * (1 - f_NC_A);                 // TDP source weights (:155-161)
        const double wNC = f_A * f_NC_A + f_S * f_NC_S;
        // Divisions by run constants become multiplications by their reciprocals, formed once per reach (IEEE divisions here);
        // an IEEE fp64 division is ~12 instructions and the day boundary had 22 of them (a quarter of its issue slots).
        const double aP_A = MPv(SIMPLYP_PM_P_NETINPUT_A) * A_catch * 100.0 / 365.;                // :42, first term
        const double aP_NC = MPv(SIMPLYP_PM_P_NETINPUT_NC) * A_catch * 100.0 / 365.;
        const double invKfMsoil = 1.0 / KfMsoil, invMsoil = 1.0 / Msoil;
        const double p0 = P_inactive * invMsoil;

        const int up_lo = a.up_ptr[s], up_hi = a.up_ptr[s + 1];
        const int rslot = a.route_slot[s];
        const int oslot = a.out_slot[s];
        double* route_w = rslot >= 0 ? a.route + (size_t)rslot * 4 * RD * E : nullptr;
        // output rows: row r of column k of this member sits at o_member + (k n_rows + r) row_stride
        const bool reduce = a.n_periods > 0;
        const size_t row_stride = (size_t)a.n_out_reaches * E;
        const size_t col_stride = (size_t)(reduce ? a.n_periods : D) * row_stride;
        double* const o_member = a.out + (size_t)(oslot >= 0 ? oslot : 0) * E + (a.out_by_slot ? slot : e);
        double* o_day = o_member + (size_t)d_begin * row_stride;       // daily rows: advanced by one row per day

        // Scheme 2 (and its fp32-stage variant): the auxiliary states for the day's start -- exp(-mu VsA), exp(-mu VsS), Qr**b_Q
        // (x cQ in the fp64 scheme: SysAug::f), Qr**k_M -- are exact functions of the carried state.  They are evaluated at the END
        // of the previous day, in one group with that day's other exponentials (below), and before the first day here; same
        // operations on the same inputs either way, so a run cut into time chunks reproduces them bit for bit.
        // TEAM == 4: a lane carries only its own slot-1 value (lane 0: EA, 1: ES, 2: pk, 3: cQ pb).
        constexpr bool HAS_AUX = (INTEG == SIMPLYP_INTEG_CASHKARP_AUG || INTEG == SIMPLYP_INTEG_CASHKARP_AUG_F32);
        constexpr bool AUX_CQ = (INTEG == SIMPLYP_INTEG_CASHKARP_AUG);
        const int qj = lane & 3;
        double aux[4] = {0.0, 0.0, 0.0, 0.0};                 // TEAM == 4: aux[0] only
        double lq = 0.0;                                      // log Qr of the carried state
        if (HAS_AUX) {
            lq = sp_log(y[4]);
            if (TEAM == 4) {
                aux[0] = sp_exp(sel4(qj, -mu * y[0], -mu * y[1], k_M * lq, b_Q * lq)) * (qj == 3 ? c.cQ : 1.0);
            } else {
                const double xs[4] = {-mu * y[0], -mu * y[1], b_Q * lq, k_M * lq};
                double rs[4];
                sp_expn<4>(xs, rs);
                aux[0] = rs[0]; aux[1] = rs[1]; aux[2] = AUX_CQ ? rs[2] * c.cQ : rs[2]; aux[3] = rs[3];
            }
        }

        for (int d0 = d_begin; d0 < d_end; d0 += TILE_D) {
            const int nd = min(TILE_D, d_end - d0);

            if (shared_forcing) {
                SP_FORCING_ROWS
                __syncthreads();
                for (int i = lane; i < nd; i += WAVE) { s_P[i] = Pser[d0 + i]; s_E[i] = Eser[d0 + i]; }
                if (SNOW) for (int i = lane; i < nd; i += WAVE) s_T[i] = Tser[d0 + i];
                if (a.dynamic_erod) for (int i = lane; i < nd; i += WAVE) s_doy[i] = a.doy[d0 + i];
                __syncthreads();
            }
            for (int dd = 0; dd < nd; ++dd) {
                const int d = d0 + dd;
                double P, PET, T_air = 0.0;
                if (shared_forcing) {
                    P = s_P[dd]; PET = s_E[dd]; if (SNOW) T_air = s_T[dd];                        // :497-498
                } else {
                    SP_FORCING_ROWS
                    P = Pser[d]; PET = Eser[d]; if (SNOW) T_air = Tser[d];
                }
                if (SNOW) {
                    // snow_hydrol_inputs (inputs.py:183-208) for this member: P so far is the day's precipitation
                    const double P_snow = (T_air < 0.0) ? P : 0.0;                                // :183-184
                    const double P_rain = P - P_snow;                                             // :187
                    double P_melt = MPv(SIMPLYP_PM_F_DDSM) * (T_air - 0);                         // :190
                    if (P_melt < 0.0) P_melt = 0.0;                                               // :191
                    P_melt = (D_snow < P_melt) ? D_snow : P_melt;                                 // :199, :204 melt limited by depth
                    D_snow = D_snow + P_snow - P_melt;                                            // :200, :205
                    P = P_rain + P_melt;                                                          // :208
                }
                const double Qq = f_quick * P;                                                    // :501

                // upstream inputs (:508-544): same-day daily means / fluxes of the reaches above
                double QrUS = 0.0, MsusUS = 0.0, TDPrUS = 0.0, PPrUS = 0.0;
                for (int k = up_lo; k < up_hi; ++k) {
                    const int u = a.up_idx[k];
                    const double* r = a.route + (size_t)a.route_slot[u] * 4 * RD * E + ((size_t)d % RD) * E + slot;
                    QrUS += r[0] * (RPv(SIMPLYP_PR_A_CATCH, u) / A_catch);                        // :524-525
                    MsusUS += r[RD * E];
                    TDPrUS += r[2 * RD * E];
                    PPrUS += r[3 * RD * E];                                                       // :526-528
                }

                // sediment input coefficients (:549-594)
                double C_cover_A = C_cover_A0;
                if (a.dynamic_erod) {
                    const int dayNo = shared_forcing ? s_doy[dd] : a.doy[d];
                    double Cs[2];
#pragma unroll
                    for (int q = 0; q < 2; ++q) {
                        const double dmid = q == 0 ? dmid0 : dmid1;
                        const double dstart = dmid - 60.0 / 2., dend = dmid + 60.0 / 2.;          // :358-359
                        const double kk = (double)dayNo - dstart;
                        const bool in_window = kk >= 0.0 && kk == floor(kk) && kk < ceil(dend - dstart)
                                               && dstart + kk == (double)dayNo;                   // :567
                        double v = C_out;
                        if (in_window)
                            v = ((double)dayNo < dmid)
                                ? C_cover_A0 + (1.0 - C_cover_A0) * (dayNo - dstart) / (dmid - dstart)   // :569-570
                                : 1.0 + (C_cover_A0 - 1.0) * (dayNo - dmid) / (dend - dmid);             // :572-573
                        Cs[q] = v;
                    }
                    C_cover_A = f_spr * Cs[0] + (1 - f_spr) * Cs[1];                              // :579-580
                }
                const double Esus_A = EA_fac * C_cover_A;

                double EPC0_A, EPC0_NC;
                if (a.dynamic_epc0) {
                    EPC0_A = __builtin_fmax(Plab_A * invKfMsoil, 0.0);                            // :602
                    EPC0_NC = __builtin_fmax(Plab_NC * invKfMsoil, 0.0);                          // :603
                } else {
                    EPC0_A = EPC0_0_A;                                                            // :607
                    EPC0_NC = (nc_type == 2) ? EPC0_0_A : EPC0_0_S;                               // :608-611
                }

                // ---- day constants of the right-hand side ----
                c.c0 = P * (1 - f_quick);
                c.aE = alpha * PET;
                c.c0m = c.c0 - c.aE;
                c.qin = Qq + QrUS;
                c.Esum = f_Ar * Esus_A + f_IG * Esus_IG + f_S * Esus_S;
                c.MsusUS = MsusUS;
                {
                    const double tNC = c.omb * wNC * conc_NC;             // soil-water TDP via QsNC (:156-157)
                    c.tA = c.omb * wA * conc_A + (nc_type == 1 ? tNC : 0.0);
                    c.tS = (nc_type == 1 ? 0.0 : tNC);
                    c.tconst = Qq * (wA * conc_A + wNC * conc_NC) + TDPeff + TDPrUS;      // :159-165
                }
                {
                    const double pA = (Plab_A + P_inactive) * invMsoil, pN = (Plab_NC + P_inactive) * invMsoil;
                    c.cPP = E_PP * (f_Ar * Esus_A * ((1 - f_NC_Ar) * pA + f_NC_Ar * pN)
                                    + f_IG * Esus_IG * ((1 - f_NC_IG) * pA + f_NC_IG * pN)
                                    + f_S * Esus_S * ((1 - f_NC_S) * p0 + f_NC_S * pN));           // :171-176
                }
                c.PPrUS = PPrUS;

                // ---- integrate the day (replaces odeint, model.py:640) ----
                double yq[4] = {0.0, 0.0, 0.0, 0.0};                                              // :618
                if (INTEG == SIMPLYP_INTEG_RK4) {
                    rk4_day(y, yq, c, a.step_len, a.substeps);
                    n_rhs += 4u * (unsigned)a.substeps; n_steps += (unsigned)a.substeps;
                } else if (INTEG == SIMPLYP_INTEG_CASHKARP) {
                    CkCounters cnt = {0u, 0u, 0u, 0u, false, false};
                    ck_day<SysLiteral>(y, yq, c, a.step_len, a.rtol, a.atol, a.max_steps, h_carry, cnt, active);
                    n_rhs += cnt.rhs; n_steps += cnt.steps; n_rej += cnt.rejected; n_trips += cnt.wave_trips;
                    if (cnt.capped) stat |= SIMPLYP_STATUS_STEPCAP;
                    if (cnt.poisoned) stat |= SIMPLYP_STATUS_NONFINITE;
                } else if (INTEG == SIMPLYP_INTEG_CASHKARP_AUG_F32) {
                    // fp32 stages: carried state, auxiliary states and day constants are rounded to float for the day's integration
                    float z[11];
                    z[0] = (float)y[0]; z[1] = (float)y[1]; z[2] = (float)y[2]; z[3] = (float)y[4];
                    z[4] = (float)y[5]; z[5] = (float)y[6]; z[6] = (float)y[7];
                    z[7] = (float)aux[0]; z[8] = (float)aux[1]; z[9] = (float)aux[2]; z[10] = (float)aux[3];
                    const DayConstF cf(c);
                    CkCounters cnt = {0u, 0u, 0u, 0u, false, false};
                    ck_day<SysAugF>(z, yq, cf, a.step_len, a.rtol, a.atol, a.max_steps, h_carry, cnt, active);
                    n_rhs += cnt.rhs; n_steps += cnt.steps; n_rej += cnt.rejected; n_trips += cnt.wave_trips;
                    if (cnt.capped) stat |= SIMPLYP_STATUS_STEPCAP;
                    if (cnt.poisoned) stat |= SIMPLYP_STATUS_NONFINITE;
                    y[0] = z[0]; y[1] = z[1]; y[2] = z[2]; y[4] = z[3]; y[5] = z[4]; y[6] = z[5]; y[7] = z[6];
                } else {
                    // augmented form
                    CkCounters cnt = {0u, 0u, 0u, 0u, false, false};
                    if (TEAM == 4) {
                        double z[7];
                        z[0] = y[0]; z[1] = y[1]; z[2] = y[2]; z[3] = y[4]; z[4] = y[5]; z[5] = y[6]; z[6] = y[7];
                        ck_day_quad<STIFF>(z, aux[0], yq, c, a.step_len, a.rtol, a.atol, a.max_steps, h_carry, cnt, active, qj, s_tab);
                        y[0] = z[0]; y[1] = z[1]; y[2] = z[2]; y[4] = z[3]; y[5] = z[4]; y[6] = z[5]; y[7] = z[6];
                    } else {
                        double z[11];
                        z[0] = y[0]; z[1] = y[1]; z[2] = y[2]; z[3] = y[4]; z[4] = y[5]; z[5] = y[6]; z[6] = y[7];
                        z[7] = aux[0]; z[8] = aux[1]; z[9] = aux[2]; z[10] = aux[3];
                        ck_day<SysAug, STIFF>(z, yq, c, a.step_len, a.rtol, a.atol, a.max_steps, h_carry, cnt, active, s_tab);
                        y[0] = z[0]; y[1] = z[1]; y[2] = z[2]; y[4] = z[3]; y[5] = z[4]; y[6] = z[5]; y[7] = z[6];
                    }
                    n_rhs += cnt.rhs; n_steps += cnt.steps; n_rej += cnt.rejected; n_trips += cnt.wave_trips;
                    if (cnt.capped) stat |= SIMPLYP_STATUS_STEPCAP;
                    if (cnt.poisoned) stat |= SIMPLYP_STATUS_NONFINITE;
                }
                const double Vg_ode = y[2];                                                       // stored 'Vg' (:644)

                // ---- end-of-day explicit flows and resets (:663-670) ----
                const double uA = y[0] - fc, uS = y[1] - fc;
                const double QsA = uA * gate(uA, c.inv_d) * c.invTsA;                             // :663
                const double QsS = uS * gate(uS, c.inv_d) * c.invTsS;                             // :664
                const double ug = __builtin_fma(y[2], c.invTg, -Qg_min);
                const double Qg = Qg_min + gate(ug, c.inv_dg) * ug;                               // :668-669
                y[2] = Qg * T_g;                                                                  // :670

                // ---- the day boundary's exponentials, one group (sp_expn) ----
                //   Vr on its invariant: Kv Qr**(1-b_Q) (schemes 2, 3 always; 0, 1 with opts.project_vr: see the oracle);
                //   exp(-b) of the two discretized_soilP calls (:44, :51);
                //   schemes 2, 3: the auxiliary states of the next day's start.
                const bool want_vr = HAS_AUX || a.project_vr;
                if (want_vr) lq = sp_log(y[4]);
                const double VsNC = (nc_leak == 1) ? y[0] : y[1];                                 // :676-681
                const double QsNC = (nc_leak == 1) ? QsA : QsS;
                SoilPRate rA = {0.0, 0.0}, rNC = {0.0, 0.0};
                if (a.dynamic_epc0) {
                    rA = soil_p_rate(KfMsoil, QsA, Qq, y[0]);
                    rNC = soil_p_rate(KfMsoil, QsNC, Qq, VsNC);
                }
                const double xA = __builtin_fmax(-rA.b, -700.0), xNC = __builtin_fmax(-rNC.b, -700.0);
                const double xVr = (1.0 - b_Q) * lq;
                double embA, embNC, eVr;
                if (TEAM == 4) {
                    const double xs[2] = {sel4(qj, -mu * y[0], -mu * y[1], k_M * lq, b_Q * lq), sel4(qj, xA, xNC, xVr, 0.0)};
                    double rs[2];
                    sp_expn<2>(xs, rs);
                    aux[0] = rs[0] * (qj == 3 ? c.cQ : 1.0);
                    embA = quad_bcast<0>(rs[1]); embNC = quad_bcast<1>(rs[1]); eVr = quad_bcast<2>(rs[1]);
                } else if (HAS_AUX) {
                    const double xs[7] = {-mu * y[0], -mu * y[1], b_Q * lq, k_M * lq, xVr, xA, xNC};
                    double rs[7];
                    sp_expn<7>(xs, rs);
                    aux[0] = rs[0]; aux[1] = rs[1]; aux[2] = AUX_CQ ? rs[2] * c.cQ : rs[2]; aux[3] = rs[3];
                    eVr = rs[4]; embA = rs[5]; embNC = rs[6];
                } else {
                    const double xs[3] = {xVr, xA, xNC};
                    double rs[3];
                    sp_expn<3>(xs, rs);
                    eVr = rs[0]; embA = rs[1]; embNC = rs[2];
                }
                if (want_vr) y[3] = Kv * eVr;                                                     // Vr = Kv Qr**(1-b_Q)
                {
                    bool fin = true;
#pragma unroll
                    for (int i = 0; i < 8; ++i) fin = fin && (__builtin_fabs(y[i]) < 1.0e300);
                    fin = fin && (__builtin_fabs(Vg_ode) < 1.0e300);
                    if (!fin) stat |= SIMPLYP_STATUS_NONFINITE;
                }

                // ---- soil P (:676-715) ----
                if (a.dynamic_epc0) {
                    soil_p_update(aP_A, KfMsoil, EPC0_A, y[0], rA, embA, TDPs_A, Plab_A, conc_A);         // :688-689, :696-697, :702
                    soil_p_update(aP_NC, KfMsoil, EPC0_NC, VsNC, rNC, embNC, TDPs_NC, Plab_NC, conc_NC);  // :692-693, :698-699, :703
                } else {
                    conc_A = EPC0_A;                                                              // :711
                    conc_NC = EPC0_NC;                                                            // :715
                }

                // ---- hand the daily series downstream and store the requested columns ----
                if (writer) {
                    if (route_w) {
                        double* r = route_w + ((size_t)d % RD) * E + slot;
                        r[0] = yq[0]; r[RD * E] = yq[1]; r[2 * RD * E] = yq[2]; r[3 * RD * E] = yq[3];
                    }
                    if (oslot >= 0 && a.out_mask) {
                        // daily rows, or (time-reduced output) running sums over the day's period: the row is owned
                        // by this member, so the read-modify-write needs no atomics; consecutive days hit the same
                        // L2-resident line, and across time chunks the queue kernel's release/acquire covers it
                        double* o = reduce ? o_member + (size_t)a.period_of_day[d] * row_stride : o_day;
                        unsigned m = a.out_mask;
#define PUT(col, val) if (m & (1u << (col))) { *o = reduce ? *o + (val) : (val); o += col_stride; }
                        PUT(SIMPLYP_OUT_VSA, y[0]) PUT(SIMPLYP_OUT_VSS, y[1]) PUT(SIMPLYP_OUT_VG, Vg_ode)
                        PUT(SIMPLYP_OUT_VR, y[3]) PUT(SIMPLYP_OUT_QR_END, y[4]) PUT(SIMPLYP_OUT_QR, yq[0])
                        PUT(SIMPLYP_OUT_MSUS_END, y[5]) PUT(SIMPLYP_OUT_MSUS_FLUX, yq[1])
                        PUT(SIMPLYP_OUT_TDPR_END, y[6]) PUT(SIMPLYP_OUT_TDP_FLUX, yq[2])
                        PUT(SIMPLYP_OUT_PPR_END, y[7]) PUT(SIMPLYP_OUT_PP_FLUX, yq[3])
                        PUT(SIMPLYP_OUT_QQ, Qq) PUT(SIMPLYP_OUT_QSA, QsA) PUT(SIMPLYP_OUT_QSS, QsS) PUT(SIMPLYP_OUT_QG, Qg)
                        PUT(SIMPLYP_OUT_C_COVER_A, C_cover_A) PUT(SIMPLYP_OUT_EPC0_A, EPC0_A) PUT(SIMPLYP_OUT_EPC0_NC, EPC0_NC)
                        PUT(SIMPLYP_OUT_TDPS_A, TDPs_A) PUT(SIMPLYP_OUT_PLAB_A, Plab_A) PUT(SIMPLYP_OUT_CONC_TDPS_A, conc_A)
                        PUT(SIMPLYP_OUT_TDPS_NC, TDPs_NC) PUT(SIMPLYP_OUT_PLAB_NC, Plab_NC) PUT(SIMPLYP_OUT_CONC_TDPS_NC, conc_NC)
                        if (SNOW) { PUT(SIMPLYP_OUT_D_SNOW, D_snow) }      // met_df['D_snow_end'] (inputs.py:200, :205) -> df_TC['D_snow'] (model.py:775-776)
#undef PUT
                    }
                }
                o_day += row_stride;
            }
        }
        if (ckpt && d_end < D && writer) {      // hand the state to whichever wave runs the next time chunk
            double* k = ckpt + slot;
#pragma unroll
            for (int i = 0; i < 8; ++i) k[(size_t)i * E] = y[i];
            k[(size_t)8 * E] = Plab_A; k[(size_t)9 * E] = TDPs_A; k[(size_t)10 * E] = Plab_NC; k[(size_t)11 * E] = TDPs_NC;
            k[(size_t)12 * E] = conc_A; k[(size_t)13 * E] = conc_NC; k[(size_t)14 * E] = h_carry;
            if (SNOW) k[(size_t)15 * E] = D_snow;
        }
    }
#undef MPv
#undef RPv
#undef SP_FORCING_ROWS

    // ---- per-wave solver statistics and member status ----
    if (!writer) { n_rhs = 0; n_steps = 0; n_rej = 0; }
    unsigned long long v0 = n_rhs, v1 = n_steps, v2 = n_rej;
    for (int off = 32; off > 0; off >>= 1) {
        v0 += __shfl_down(v0, off); v1 += __shfl_down(v1, off); v2 += __shfl_down(v2, off);
    }
    if (lane == 0) {
        atomicAdd(&a.counters[0], v0); atomicAdd(&a.counters[1], v1); atomicAdd(&a.counters[2], v2);
        atomicAdd(&a.counters[3], (unsigned long long)n_trips);      // wave-level attempts (lane 0's count = the wave's)
    }
    if (writer && stat) atomicOr(&a.status[e], stat);
    if (writer && a.member_rhs) atomicAdd(&a.member_rhs[e], n_rhs);      // several chains / time chunks per member
}

// Member slot of a lane: group g carries a.lanes consecutive slots; the other lanes of the wave get E (= no member).
// Fewer than 64 members per wave is what a small ensemble wants: a wave's day costs the attempts of its slowest lane, and an
// ensemble that cannot fill the chip's 1024 SIMDs with full waves loses nothing by spreading over more, thinner ones.
__device__ __forceinline__ int slot_of_lane(const KernelArgs& a, int group, int lane)
{
    const int l = lane >> a.team_shift;
    return l < a.lanes ? group * a.lanes + l : a.E;
}

// One wave per SIMD for every instantiation (__launch_bounds__(64, 1)): the fp64 schemes need ~450-500 registers for the
// Cash-Karp stages, and the fp32-stage scheme gains nothing from a second wave -- the chip issues v_fma_f32 every 1.29 ns for a
// lone wave and every 2.15 ns per wave with a partner on the SIMD, i.e. co-residency is worth 1.20 x on a bare FMA stream
// (tools/micro/valu_rates.hip, profiles/r02_valu_rates.log), while fitting two waves (256 registers each, with the fp64 day-level
// state beside the float stages) spilled 712 B per lane to scratch and ran 22 % slower (profiles/r02_experiments.md).  What
// makes BASELINE config C5 fast is its tolerance (44 instead of 84 right-hand sides per catchment-day), not the fp32 rate.
template <int INTEG, bool SNOW, int TEAM, bool STIFF = false>
__global__ __launch_bounds__(WAVE, 1) void simplyp_chain_kernel(const KernelArgs a)
{
    __shared__ double s_P[TILE_D];
    __shared__ double s_E[TILE_D];
    __shared__ double s_T[TILE_D];
    __shared__ int s_doy[TILE_D];
    __shared__ double s_tab[STIFF ? 2 * TAB_STRIDE : 1];
    if (STIFF) tableau_to_lds(s_tab, threadIdx.x);
    const int c0 = a.chain_ptr[blockIdx.y], c1 = a.chain_ptr[blockIdx.y + 1];
    if (a.win_stride > 0) {      // pilot windows: the same short run over different stretches of the forcing
        KernelArgs w = a;
        const size_t off = (size_t)blockIdx.z * a.win_stride;
        w.forcing = a.forcing + off;
        w.doy = a.doy ? a.doy + off : nullptr;
        w.member_rhs = a.member_rhs + (size_t)blockIdx.z * a.E;
        w.route = a.route ? a.route + (size_t)blockIdx.z * a.win_route_stride : nullptr;
        run_slot<INTEG, SNOW, TEAM, STIFF>(w, s_P, s_E, s_T, s_doy, threadIdx.x, slot_of_lane(a, blockIdx.x, threadIdx.x), a.chain_reach + c0, c1 - c0, 0, a.D, nullptr, s_tab);
        return;
    }
    run_slot<INTEG, SNOW, TEAM, STIFF>(a, s_P, s_E, s_T, s_doy, threadIdx.x, slot_of_lane(a, blockIdx.x, threadIdx.x), a.chain_reach + c0, c1 - c0, 0, a.D, nullptr, s_tab);
}

// ---------------------------------------------------------------------------------------
// Work-conserving, pipelined variant: one persistent wave per SIMD pulls tasks from a ticket counter.
//
// A task is (reach s, time chunk c, member group g of 64 lane slots).  It needs
//   (s, c-1, g)            the group's own state at the chunk boundary, left in `ckpt` by whichever wave ran it;
//   (u, c, g), u in up(s)  the upstream reaches' daily series for the chunk (routing ring buffers);
// and, because those ring buffers hold only `ring_chunks` chunks per reach,
//   (dn, c-ring_chunks, g), dn in down(s)   every reader of the ring rows this task is about to overwrite.
// The host numbers the tasks by (level(s) + c, then reach, then group) -- level = longest distance from a
// headwater -- and sizes the ring so that each of these dependencies has a SMALLER number.  Tickets are handed
// out in task order, so whatever a wave waits for is already running on a resident wave (or finished) and itself
// waits only on still smaller tickets: no assumption about dispatch order or placement, no deadlock.  A chain of
// 256 reaches therefore runs as a pipeline 256 deep (reach s works on chunk c while reach s+1 works on chunk c-1),
// and a single-reach ensemble simply keeps every SIMD busy until the tickets run out, whatever the members'
// relative costs.
//
// Hand-off (MI355X_MICROARCH.md, "Valid forms"): producer = plain stores, s_waitcnt vmcnt(0), agent-scope
// release, s_waitcnt vmcnt(0) (asm, so the compiler cannot drop it), relaxed agent-scope flag store;
// consumer = relaxed agent-scope polls (wave-uniform address, with s_sleep), agent-scope acquire, s_waitcnt vmcnt(0),
// then plain loads.  The spin is bounded (by lack of progress of the whole run, queue_wait): the wave then raises q.error and
// every worker drains.
struct QueueArgs {
    unsigned* ticket;          // next task number
    unsigned* done;            // [S][n_groups] chunks completed per (reach, member group)
    unsigned* error;           // set to 1 on a wait timeout
    unsigned* progress;        // tasks completed so far, by anybody: what a waiting wave watches (queue_wait)
    double* ckpt;              // [S][CKPT_N][E]
    const int* task_reach;     // [n_pairs] reach of the (reach, chunk) pair, in dependency order
    const int* task_chunk;     // [n_pairs]
    const int* down_ptr;       // [S+1] CSR of directly-downstream reaches
    const int* down_idx;
    int n_groups, n_pairs, chunk_days, ring_chunks;
    unsigned max_polls;
    // streamed output (simplyp_stream_out): tasks finished per time chunk; the wave whose task completes a chunk raises the
    // chunk's flag in host-pinned memory, and a host thread then copies that chunk's rows of `out` to the host on a second
    // stream while later chunks compute.  nullptr = off.
    unsigned* chunk_count;     // [n_chunks] device
    unsigned* host_ready;      // [n_chunks] host-pinned, device-visible
    unsigned tasks_per_chunk;  // S * n_groups
};

// Wait until *flag >= need.  Executed by the whole wave on a wave-uniform address (the 64 identical loads are
// one request) and made scalar with readfirstlane, so the spin is a scalar loop: no lane-masked control flow
// around it.
// The bound is on PROGRESS, not on time: the wait fails only when no task of the whole run has completed for q.max_polls
// polls in a row (`q.progress` is bumped by every finished task).  A run that is merely slow -- a GPU shared with another
// process, a profiler that turns the streamed copies into shader blits -- keeps completing tasks and never trips it; a
// dependency that can no longer be satisfied (the deadlock the ticket order rules out, a worker lost to a fault) does, after
// max_polls x ~2 us.  (Until round 3 the bound was the wait's own poll count: any healthy run slowed down enough would have
// failed with "timed out".)
struct QueueWaitStats { unsigned waits, longest_wait, longest_stall; };

__device__ __forceinline__ unsigned queue_load(const unsigned* p)
{
    return (unsigned)__builtin_amdgcn_readfirstlane((int)__hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT));
}

__device__ __forceinline__ bool queue_wait(const QueueArgs& q, const unsigned* flag, unsigned need, QueueWaitStats& ws)
{
    if (queue_load(flag) >= need) return true;
    ++ws.waits;
    unsigned polls = 0, stall = 0, seen = queue_load(q.progress);
    bool ok;
    for (;;) {
        __builtin_amdgcn_s_sleep(64);
        ++polls;
        if (queue_load(flag) >= need) { ok = true; break; }
        const unsigned now = queue_load(q.progress);
        stall = (now != seen) ? 0u : stall + 1u;
        seen = now;
        ws.longest_stall = max(ws.longest_stall, stall);
        if (stall > q.max_polls || queue_load(q.error) != 0u) { ok = false; break; }
    }
    ws.longest_wait = max(ws.longest_wait, polls);
    return ok;
}

__device__ __forceinline__ unsigned queue_take_ticket(const QueueArgs& q, int lane)
{
    unsigned k = 0;
    if (lane == 0) k = atomicAdd(q.ticket, 1u);
    return (unsigned)__builtin_amdgcn_readfirstlane((int)k);
}

// The persistent loop has a single exit, at its head, on a scalar: with `break`s in the body hipcc (ROCm 7.2)
// structurised the ticket loop into a lane-masked inner loop that re-ran task 0 forever.
template <int INTEG, bool SNOW, int TEAM, bool STIFF = false>
__global__ __launch_bounds__(WAVE, 1) void simplyp_queue_kernel(const KernelArgs a, const QueueArgs q)
{
    __shared__ double s_P[TILE_D];
    __shared__ double s_E[TILE_D];
    __shared__ double s_T[TILE_D];
    __shared__ int s_doy[TILE_D];
    __shared__ double s_tab[STIFF ? 2 * TAB_STRIDE : 1];
    if (STIFF) tableau_to_lds(s_tab, threadIdx.x);
    const int lane = threadIdx.x;
    const unsigned G = (unsigned)q.n_groups;
    const unsigned n_tasks = (unsigned)q.n_pairs * G;
    QueueWaitStats ws = {0u, 0u, 0u};
    unsigned k = queue_take_ticket(q, lane);
    while (k < n_tasks) {
        const int pair = (int)(k / G), g = (int)(k % G);
        const int s = __builtin_amdgcn_readfirstlane(q.task_reach[pair]);
        const int c = __builtin_amdgcn_readfirstlane(q.task_chunk[pair]);
        bool ok = true;                       // everything here is wave-uniform
        if (c > 0) ok = queue_wait(q, &q.done[(size_t)s * G + g], (unsigned)c, ws);
        for (int i = a.up_ptr[s]; ok && i < a.up_ptr[s + 1]; ++i)
            ok = queue_wait(q, &q.done[(size_t)a.up_idx[i] * G + g], (unsigned)(c + 1), ws);
        if (c >= q.ring_chunks)
            for (int i = q.down_ptr[s]; ok && i < q.down_ptr[s + 1]; ++i)
                ok = queue_wait(q, &q.done[(size_t)q.down_idx[i] * G + g], (unsigned)(c - q.ring_chunks + 1), ws);
        if (ok) {
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            const int d_begin = c * q.chunk_days;
            const int d_end = min(a.D, d_begin + q.chunk_days);
            run_slot<INTEG, SNOW, TEAM, STIFF>(a, s_P, s_E, s_T, s_doy, lane, slot_of_lane(a, g, lane), q.task_reach + pair, 1, d_begin, d_end,
                                   q.ckpt + (size_t)s * CKPT_N * (size_t)a.E, s_tab);
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            if (lane == 0) {
                __hip_atomic_store(&q.done[(size_t)s * G + g], (unsigned)(c + 1), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                __hip_atomic_fetch_add(q.progress, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);      // what waiting waves watch
            }
            if (q.chunk_count) {
                // every task adds to its chunk's counter AFTER its own release (above: the XCD's L2 has been written back and
                // the wait has drained), so when the last add arrives all rows of the chunk are in memory, where the copy engine
                // reads them; the flag itself goes to host memory with a system-scope release
                unsigned old = 0;
                if (lane == 0) old = atomicAdd(&q.chunk_count[c], 1u);
                old = (unsigned)__builtin_amdgcn_readfirstlane((int)old);
                if (old + 1u == q.tasks_per_chunk && lane == 0)
                    __hip_atomic_store(&q.host_ready[c], 1u, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
            }
            k = queue_take_ticket(q, lane);
        } else {
            if (lane == 0) __hip_atomic_store(q.error, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            k = n_tasks;                      // a wait found the run stalled (or another wave raised the error flag): drain
        }
    }
    if (lane == 0 && ws.waits) {
        atomicAdd(&a.counters[4], (unsigned long long)ws.waits);
        atomicMax(&a.counters[5], (unsigned long long)ws.longest_wait);
        atomicMax(&a.counters[6], (unsigned long long)ws.longest_stall);
    }
}

// Gather the columns of a [rows][E] array into lane-slot order: dst[r][i] = src[r][perm[i]].  One pass over the
// parameter tables (tens of MB) so that the time-stepping kernels read them coalesced, however often.
template <typename T>
__global__ void gather_columns_kernel(const T* __restrict__ src, T* __restrict__ dst, const int* __restrict__ perm,
                                      int rows, int E)
{
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= E) return;
    const int e = perm[i];
    for (int r = blockIdx.y; r < rows; r += gridDim.y) dst[(size_t)r * E + i] = src[(size_t)r * E + e];
}

// The scalar device functions of the path on caller-given arguments (simplyp_eval_units): what lets the tests pin the DEVICE
// restatements of f_x (model.py:23-37) and discretized_soilP (:39-56, with the clamps of :696-699 and the concentration of
// :702-703) to the vectors the unmodified reference functions produced, Vs == 0 included.
//   which 0: in [n][2] = x, threshold           -> out [n][2] = f_x by gate() (end-of-day flows), f_x by the fused form of SysAug::f
//   which 1: in [n][10] = P_netInput, A_catch, Kf, Msoil, EPC0, Qs, Qq, Vs, TDPs, Plab  -> out [n][3] = TDPs, Plab, conc_TDPs
__global__ void eval_units_kernel(int which, int n, const double* __restrict__ in, double* __restrict__ out)
{
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    if (which == 0) {
        const double x = in[2 * i], th = in[2 * i + 1];
        const double inv_d = (th * 0.01 > 0.0) ? 1.0 / (0.01 * th) : 1.0e300;      // threshold 0 -> plain step (run_slot: c.inv_dg)
        out[2 * i] = gate(x - th, inv_d);
        double s = __builtin_fma(x, inv_d, -th * inv_d);                           // SysAug::f: s = fma(Vs, inv_d, s0), s0 = -fc inv_d
        s = __builtin_fmin(__builtin_fmax(s, 0.0), 1.0);
        out[2 * i + 1] = (s * s) * __builtin_fma(-2.0, s, 3.0);
    } else {
        const double* a = in + 10 * (size_t)i;
        const double aP = a[0] * a[1] * 100.0 / 365.;                              // run_slot: aP_A
        const double KfMsoil = a[2] * a[3];
        const SoilPRate r = soil_p_rate(KfMsoil, a[5], a[6], a[7]);
        const double emb = sp_exp(__builtin_fmax(-r.b, -700.0));
        double TDPs = a[8], Plab = a[9], conc;
        soil_p_update(aP, KfMsoil, a[4], a[7], r, emb, TDPs, Plab, conc);
        out[3 * i] = TDPs; out[3 * i + 1] = Plab; out[3 * i + 2] = conc;
    }
}

}  // namespace simplyp
