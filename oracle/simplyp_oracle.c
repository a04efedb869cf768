/*
 * simplyp_oracle.c -- TEST INFRASTRUCTURE ONLY.  Not part of the product.
 *
 * A scalar fp64 CPU restatement of the reference's time-stepping path
 * (/root/reference/Current_Release/v0-2A/simplyP/model.py), written formula by formula in
 * the reference's own order, so that the HIP kernel can be checked against it on a machine
 * where the Python reference does not exist.  Only tests/, __graft_entry__.smoke() and
 * bench.py's cpu_baseline leg may load this; the product path never does.
 *
 * Pinning (tests/test_oracle_*.py, fixtures in tests/golden/ made by make_golden.py from the
 * unmodified reference):
 *   - oracle_fx / oracle_soilp / oracle_ode_f  ==  reference f_x / discretized_soilP / ode_f
 *     on the committed in/out vectors (1e-13 rel);
 *   - the full driver with a converged integrator == reference run_simply_p with
 *     odeint(rtol=atol=1e-12) on 4 scenarios (tolerance stated in the tests).
 * The integrator itself is NOT the reference's: the reference calls SciPy odeint = ODEPACK
 * LSODA (scipy is un-vendored; reference pins scipy 1.2.0 in README.md:39, 1.15.3 here) at
 * rtol=0.01 (model.py:640).  LSODA's step/order heuristics are not restated; the oracle
 * integrates the same right-hand side with classical RK4 or Cash-Karp 5(4) (literal system, or the
 * augmented transcendental-free form described at ode_aug), the schemes the device kernel implements.  For the LSODA-at-rtol=0.01 trajectory itself: parity
 * unpinned (the reference's own shipped CSVs pin it only to ~3e-3).
 */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>
#ifdef _OPENMP
#include <omp.h>
#endif

#include "../include/simplyp.h"
#include "../include/simplyp_controller.h"    /* the step controller's constants, shared with the device kernels */

/* ------------------------------------------------------------------------------------- */
/* f_x, model.py:23-37.  threshold == 0 gives d == 0: the reference then divides 0/0 on the
 * single point x == 0; defined here (and in the kernel) as a plain step.                  */
static double f_x(double x, double threshold, double reld)
{
    double d = threshold * reld;
    if (x < threshold) return 0.0;
    if (x > threshold + d) return 1.0;
    if (d == 0.0) return 0.0;            /* x == threshold == 0 */
    double s = (x - threshold) / d;
    return -2.0 * s * s * s + 3.0 * s * s;                      /* :33 */
}

/* discretized_soilP, model.py:39-56 */
static void discretized_soilP(double P_netInput, double A_catch, double Kf, double Msoil, double EPC0,
                              double Qs, double Qq, double Vs, double TDPs, double Plab,
                              double* TDPs_out, double* Plab_out)
{
    double a = P_netInput * A_catch * 100.0 / 365. + Kf * Msoil * EPC0;          /* :42 */
    double b = (Kf * Msoil + Qs + Qq) / Vs;                                      /* :43 */
    TDPs = a / b + (TDPs - a / b) * exp(-b);                                     /* :44 */
    double b0 = b * Vs;                                                          /* :47 */
    double sorp;
    if (Vs > 0)                                                                  /* :50 */
        sorp = Kf * Msoil * (a / b0 - EPC0 + (1 / b) * (TDPs / Vs - a / b0) * (1 - exp(-b)));   /* :51 */
    else
        sorp = 0.;
    *TDPs_out = TDPs;
    *Plab_out = Plab + sorp;                                                     /* :54 */
}

/* Everything ode_f unpacks from its 41-tuple (model.py:74-78) that it actually uses. */
typedef struct {
    double P, E, mu, Qq_i, Qr_US_i, Esus_A, Esus_S, Esus_IG, Msus_US_i, TDPr_US_i, PPr_US_i;
    double f_A, f_Ar, f_IG, f_S, f_NC_A, f_NC_Ar, f_NC_IG, f_NC_S;
    int    NC_type;          /* 0 'None', 1 'A', 2 'S' */
    double f_quick, alpha, beta, T_s_A, T_s_S, T_g, fc, L_reach, A_catch, a_Q, b_Q, k_M;
    double conc_TDPs_A, conc_TDPs_NC, PlabA_i, PlabNC_i, Msoil, TDPeff, TDPg, E_PP, P_inactive, Qg_min;
} ode_params;

/* ode_f, model.py:58-187.  y and dy have the reference's 12 slots. */
static void ode_f(const double* y, const ode_params* p, double* dy)
{
    double VsA_i = y[0], VsS_i = y[1], Vg_i = y[2], Vr_i = y[3], Qr_i = y[4];
    double Msus_i = y[6], TDPr_i = y[8], PPr_i = y[10];

    double QsA_i = (VsA_i - p->fc) * f_x(VsA_i, p->fc, 0.01) / p->T_s_A;                         /* :105 */
    double dVsA_dt = p->P * (1 - p->f_quick) - p->alpha * p->E * (1 - exp(-p->mu * VsA_i)) - QsA_i;  /* :106 */
    double QsS_i = (VsS_i - p->fc) * f_x(VsS_i, p->fc, 0.01) / p->T_s_S;                         /* :109 */
    double dVsS_dt = p->P * (1 - p->f_quick) - p->alpha * p->E * (1 - exp(-p->mu * VsS_i)) - QsS_i;  /* :110 */

    double QsNC_i = (p->NC_type == 1) ? QsA_i : QsS_i;                                           /* :113-118 */

    double f_Qg = f_x(Vg_i / p->T_g, p->Qg_min, 0.01);                                           /* :121 */
    double Qg_i = (1 - f_Qg) * p->Qg_min + f_Qg * (Vg_i / p->T_g);                               /* :122 */
    double dVg_dt = p->beta * (p->f_A * QsA_i + p->f_S * QsS_i) - Qg_i;                          /* :124 */

    double inflow = p->Qq_i + (1 - p->beta) * (p->f_A * QsA_i + p->f_S * QsS_i) + Qg_i + p->Qr_US_i - Qr_i;
    double dQr_dt = inflow * p->a_Q * pow(Qr_i, p->b_Q) * (8.64 * 10000) / ((1 - p->b_Q) * (p->L_reach));  /* :127-130 */
    double dVr_dt = inflow;                                                                      /* :131 */
    double dQr_av_dt = Qr_i;                                                                     /* :132 */

    double QrkM = pow(Qr_i, p->k_M);
    double Msus_in_A = p->Esus_A * QrkM, Msus_in_S = p->Esus_S * QrkM, Msus_in_IG = p->Esus_IG * QrkM;  /* :138 */
    double dMsus_dt = p->f_Ar * Msus_in_A + p->f_IG * Msus_in_IG + p->f_S * Msus_in_S
                      + p->Msus_US_i - (Msus_i / Vr_i) * Qr_i;                                   /* :141-145 */
    double dMsus_out_dt = (Msus_i / Vr_i) * Qr_i;                                                /* :147 */

    double dTDPr_dt = ((1 - p->beta) *
                       (p->f_A * (1 - p->f_NC_A) * QsA_i * (p->conc_TDPs_A)
                        + p->f_A * p->f_NC_A * QsNC_i * (p->conc_TDPs_NC)
                        + p->f_S * p->f_NC_S * QsNC_i * (p->conc_TDPs_NC))
                       + p->f_A * (1 - p->f_NC_A) * p->Qq_i * (p->conc_TDPs_A)
                       + p->f_A * p->f_NC_A * p->Qq_i * (p->conc_TDPs_NC)
                       + p->f_S * p->f_NC_S * p->Qq_i * (p->conc_TDPs_NC)
                       + Qg_i * (p->TDPg * p->A_catch)
                       + p->TDPeff
                       + p->TDPr_US_i
                       - Qr_i * (TDPr_i / Vr_i));                                                /* :154-166 */
    double dTDPr_out_dt = Qr_i * (TDPr_i / Vr_i);                                                /* :168 */

    double dPPr_dt = (p->E_PP * (p->f_Ar * (1 - p->f_NC_Ar) * Msus_in_A * (p->PlabA_i + p->P_inactive) / p->Msoil
                                 + p->f_IG * (1 - p->f_NC_IG) * Msus_in_IG * (p->PlabA_i + p->P_inactive) / p->Msoil
                                 + p->f_S * (1 - p->f_NC_S) * Msus_in_S * p->P_inactive / p->Msoil
                                 + p->f_Ar * p->f_NC_Ar * Msus_in_A * (p->PlabNC_i + p->P_inactive) / p->Msoil
                                 + p->f_IG * p->f_NC_IG * Msus_in_IG * (p->PlabNC_i + p->P_inactive) / p->Msoil
                                 + p->f_S * p->f_NC_S * Msus_in_S * (p->PlabNC_i + p->P_inactive) / p->Msoil)
                      + p->PPr_US_i
                      - Qr_i * (PPr_i / Vr_i));                                                  /* :171-178 */
    double dPPr_out_dt = Qr_i * PPr_i / Vr_i;                                                    /* :180 */

    dy[0] = dVsA_dt; dy[1] = dVsS_dt; dy[2] = dVg_dt; dy[3] = dVr_dt; dy[4] = dQr_dt; dy[5] = dQr_av_dt;
    dy[6] = dMsus_dt; dy[7] = dMsus_out_dt; dy[8] = dTDPr_dt; dy[9] = dTDPr_out_dt; dy[10] = dPPr_dt;
    dy[11] = dPPr_out_dt;                                                                        /* :184-185 */
}

/* ------------------------------------------------------------------------------------- */
/* Integrators over [0, T] for the autonomous 12-system.                                  */

#define NY 12

typedef struct { uint64_t rhs, steps, rejected; int capped, poisoned; } integ_stats;

static void rk4_day(double* y, const ode_params* p, double T, int n, integ_stats* st)
{
    double h = T / n, k1[NY], k2[NY], k3[NY], k4[NY], yt[NY];
    for (int s = 0; s < n; ++s) {
        ode_f(y, p, k1);
        for (int i = 0; i < NY; ++i) yt[i] = y[i] + 0.5 * h * k1[i];
        ode_f(yt, p, k2);
        for (int i = 0; i < NY; ++i) yt[i] = y[i] + 0.5 * h * k2[i];
        ode_f(yt, p, k3);
        for (int i = 0; i < NY; ++i) yt[i] = y[i] + h * k3[i];
        ode_f(yt, p, k4);
        for (int i = 0; i < NY; ++i) y[i] += h * (1.0 / 6.0) * (k1[i] + 2.0 * k2[i] + 2.0 * k3[i] + k4[i]);
    }
    st->rhs += 4u * (uint64_t)n; st->steps += (uint64_t)n;
}

/* Cash-Karp 5(4) embedded pair (Cash & Karp, ACM TOMS 16 (1990) 201-222). */
static const double CK_A[6][5] = {
    {0},
    {1.0 / 5},
    {3.0 / 40, 9.0 / 40},
    {3.0 / 10, -9.0 / 10, 6.0 / 5},
    {-11.0 / 54, 5.0 / 2, -70.0 / 27, 35.0 / 27},
    {1631.0 / 55296, 175.0 / 512, 575.0 / 13824, 44275.0 / 110592, 253.0 / 4096}};
static const double CK_B[6] = {37.0 / 378, 0, 250.0 / 621, 125.0 / 594, 0, 512.0 / 1771};
static const double CK_E[6] = {37.0 / 378 - 2825.0 / 27648, 0, 250.0 / 621 - 18575.0 / 48384,
                               125.0 / 594 - 13525.0 / 55296, -277.0 / 14336, 512.0 / 1771 - 1.0 / 4};   /* b - bhat */

/* An explicit embedded Runge-Kutta pair as data: what cashkarp_aug_day's driver (erk_aug_day) steps with.  Cash-Karp is the one
 * the device kernels implement; Tsitouras 5(4) exists here for the round-3 probe only (tools/probe_pair.py: would another pair
 * need fewer right-hand sides under the same knee-aware controller?  oracle-only integrator id ORACLE_INTEG_TSIT5_AUG). */
#define ERK_MAX_STAGES 13
typedef struct {
    int ns;                          /* stages */
    int fsal;                        /* 1: the last stage is the derivative at the new point (b == last row of A) */
    double A[ERK_MAX_STAGES][ERK_MAX_STAGES - 1];
    double B[ERK_MAX_STAGES], E[ERK_MAX_STAGES];      /* weights of the higher-order solution; b - bhat */
    double E3[ERK_MAX_STAGES];       /* second estimator (Dormand-Prince 8(5,3) only; all zero otherwise): the error of a component is then
                                        |e5|^2 / sqrt(e5^2 + 0.01 e3^2), Hairer's combination, applied per component under the max norm */
    int two_est;
    double err_exp;                  /* step-size factor = SAFETY * err^err_exp: -1/5 for the 5(4) pairs, -1/8 for the 8(5,3) pair */
} erk_tableau;

static const erk_tableau TAB_CASHKARP = {
    6, 0,
    {{0}, {1.0 / 5}, {3.0 / 40, 9.0 / 40}, {3.0 / 10, -9.0 / 10, 6.0 / 5}, {-11.0 / 54, 5.0 / 2, -70.0 / 27, 35.0 / 27},
     {1631.0 / 55296, 175.0 / 512, 575.0 / 13824, 44275.0 / 110592, 253.0 / 4096}},
    {37.0 / 378, 0, 250.0 / 621, 125.0 / 594, 0, 512.0 / 1771},
    {37.0 / 378 - 2825.0 / 27648, 0, 250.0 / 621 - 18575.0 / 48384, 125.0 / 594 - 13525.0 / 55296, -277.0 / 14336, 512.0 / 1771 - 1.0 / 4},
    {0}, 0, -0.2};

/* Ch. Tsitouras, "Runge-Kutta pairs of order 5(4) satisfying only the first column simplifying assumption", Computers &
 * Mathematics with Applications 62 (2011) 770-775.  7 stages, FSAL: 6 new right-hand sides per step.  (The 17 order
 * conditions of order 5 hold for B to 1e-15 and those of order 4 for B - E: checked when the table was typed in.) */
static const erk_tableau TAB_TSIT5 = {
    7, 1,
    {{0}, {0.161}, {-0.008480655492356989, 0.335480655492357},
     {2.8971530571054935, -6.359448489975075, 4.3622954328695815},
     {5.325864828439257, -11.748883564062828, 7.4955393428898365, -0.09249506636175525},
     {5.86145544294642, -12.92096931784711, 8.159367898576159, -0.071584973281401, -0.028269050394068383},
     {0.09646076681806523, 0.01, 0.4798896504144996, 1.379008574103742, -3.290069515436081, 2.324710524099774}},
    {0.09646076681806523, 0.01, 0.4798896504144996, 1.379008574103742, -3.290069515436081, 2.324710524099774, 0.0},
    {-0.00178001105222577714, -0.0008164344596567469, 0.007880878010261995, -0.1447110071732629, 0.5823571654525552,
     -0.45808210592918697, 0.015151515151515152},
    {0}, 0, -0.2};
#define ORACLE_INTEG_TSIT5_AUG 12      /* not in include/simplyp.h: a probe of this file only */
#define ORACLE_INTEG_DOP853_AUG 13     /* likewise: Dormand-Prince 8(5,3) under the same controller */
#include "dop853_tableau.inc"
static erk_tableau TAB_DOP853;
static int tab_dop853_ready = 0;
static const erk_tableau* tab_dop853(void)
{
    if (!tab_dop853_ready) {
#pragma omp critical
        {
            memset(&TAB_DOP853, 0, sizeof(TAB_DOP853));
            TAB_DOP853.ns = 13; TAB_DOP853.fsal = 1; TAB_DOP853.two_est = 1; TAB_DOP853.err_exp = -0.125;
            for (int s = 0; s < 13; ++s) {
                for (int j = 0; j < 12; ++j) TAB_DOP853.A[s][j] = DOP853_A[s][j];
                TAB_DOP853.B[s] = DOP853_B[s]; TAB_DOP853.E[s] = DOP853_E5[s]; TAB_DOP853.E3[s] = DOP853_E3[s];
            }
            tab_dop853_ready = 1;
        }
    }
    return &TAB_DOP853;
}

static int state_finite(const double* y)
{   /* the 8 carried states (slots 0-4, 6, 8, 10) */
    static const int idx[8] = {0, 1, 2, 3, 4, 6, 8, 10};
    for (int i = 0; i < 8; ++i) if (!(fabs(y[idx[i]]) < 1.0e300)) return 0;
    return 1;
}

/*
 * Adaptive step control -- the exact rule the device kernel implements (ck_day in
 * simplyp_amd/csrc/simplyp_kernels.hip.h):
 *   a day whose start state is non-finite is not integrated (member poisoned);
 *   trial step hh = h, except  rem <= 1.1 h -> hh = rem;  rem < 2 h -> hh = rem/2   (rem = T - t);
 *   the max_steps-th attempt of a day takes hh = rem and is accepted whatever its error (STEPCAP);
 *   err = max_i |hh e_i| / (atol + rtol * max(|y_i|, |y_i + hh k1_i|)) over all 12 components (Euler predictor in the
 *   scale; the four daily integrals use their new value);
 *   a non-finite trial is rejected with factor 0.2; at hh <= 1e-9 T (or on the last attempt) the
 *   member is poisoned (state := NaN) instead;
 *   accept iff err <= 1;  h <- hh * clamp(0.9 * err^(-1/5), 0.2, 5)   (err == 0 -> 5);
 *   *h_carry (the controller's h) persists from day to day.
 */
static void cashkarp_day(double* y, const ode_params* p, double T, double rtol, double atol,
                         int max_steps, double* h_carry, integ_stats* st)
{
    double k[6][NY], yt[NY], yn[NY];
    double t = 0.0, h = *h_carry;
    int attempts = 0;
    if (!(h > 0.0) || h > T) h = T;
    if (!state_finite(y)) { y[5] = y[7] = y[9] = y[11] = NAN; st->poisoned = 1; return; }
    while (t < T) {
        double rem = T - t, hh = h;
        if (rem <= 1.1 * h) hh = rem; else if (rem < 2.0 * h) hh = 0.5 * rem;
        int last_chance = (attempts + 1 >= max_steps);
        if (last_chance) hh = rem;
        ode_f(y, p, k[0]);
        for (int s = 1; s < 6; ++s) {
            for (int i = 0; i < NY; ++i) {
                double acc = 0.0;
                for (int j = 0; j < s; ++j) acc += CK_A[s][j] * k[j][i];
                yt[i] = y[i] + hh * acc;
            }
            ode_f(yt, p, k[s]);
        }
        st->rhs += 6;
        double err = 0.0; int bad = 0;
        for (int i = 0; i < NY; ++i) {
            double inc = 0.0, ee = 0.0;
            for (int s = 0; s < 6; ++s) { inc += CK_B[s] * k[s][i]; ee += CK_E[s] * k[s][i]; }
            yn[i] = y[i] + hh * inc;
            /* scale: the state and its Euler predictor (the kernel forms the 5th-order increment only for accepted steps);
             * the four daily integrals (5, 7, 9, 11), accumulated outside the stages there, use their new value */
            const int is_quad = (i == 5 || i == 7 || i == 9 || i == 11);
            double sc = atol + rtol * fmax(fabs(y[i]), is_quad ? fabs(yn[i]) : fabs(y[i] + hh * k[0][i]));
            double r = fabs(hh * ee) / sc;
            if (r > err) err = r;                 /* NaN terms drop out, like v_max_f64 */
        }
        if (!(err < 1.0e300) || !state_finite(yn)) bad = 1;
        ++attempts;
        if (last_chance) st->capped = 1;
        if (bad && (last_chance || hh <= 1.0e-9 * T)) {
            for (int i = 0; i < NY; ++i) y[i] = NAN;
            st->poisoned = 1;
            break;
        } else if (!bad && (err <= 1.0 || last_chance)) {
            memcpy(y, yn, sizeof(yn));
            t = (hh == rem) ? T : t + hh;
            st->steps++;
        } else {
            st->rejected++;
        }
        double fac;
        if (bad) fac = SIMPLYP_CTRL_FAC_MIN;
        else if (err == 0.0) fac = SIMPLYP_CTRL_FAC_MAX;
        else { fac = SIMPLYP_CTRL_SAFETY * pow(err, -0.2); if (fac < SIMPLYP_CTRL_FAC_MIN) fac = SIMPLYP_CTRL_FAC_MIN; if (fac > SIMPLYP_CTRL_FAC_MAX) fac = SIMPLYP_CTRL_FAC_MAX; }
        h = hh * fac;
    }
    *h_carry = h;
}

/* ------------------------------------------------------------------------------------- */
/*
 * SIMPLYP_INTEG_CASHKARP_AUG -- Cash-Karp on an augmented, transcendental-free form of the same system.
 *
 * ode_f spends almost all of its arithmetic on exp(-mu*Vs) (x2, :106,:110) and Qr**b_Q, Qr**k_M (:130,:138).
 * Each of those is a smooth function of a state variable, so it obeys an ODE that is an exact consequence
 * of the reference's equations:
 *     EA = exp(-mu VsA)   ->  dEA/dt = -mu EA dVsA/dt        (same for ES)
 *     pb = Qr**b_Q        ->  dpb/dt = b_Q pb (dQr/dt) / Qr   (same for pk = Qr**k_M)
 * and Vr is not a free variable at all: (:127-131) and Vr0 (:457-459) give Vr == L Qr**(1-b_Q) / (a_Q 86400),
 * i.e. Qr/Vr = pb a_Q 86400 / L.  Carrying EA, ES, pb, pk as four extra Runge-Kutta states (re-evaluated
 * exactly from VsA, VsS, Qr at the start of every day, pb and pk also after every 8th attempted step within a day,
 * so they cannot drift) leaves a right-hand side of
 * ~70 multiply-adds and one reciprocal.  The solution is the same function of time; the truncation error
 * is of the same order and is held by the same error controller; its norm runs over the 7 physical states
 * (the auxiliary states are functions of them, the 4 daily integrals are quadratures of them: including those
 * 8 components costs 8 % more steps and buys no accuracy on any output, measured on the 30-year series and on
 * the Monte-Carlo members).  State vector z: VsA VsS Vg Qr Msus TDPr PPr EA ES pb pk | Qr_av Msus_out TDP_out PP_out.
 */
#define NZ 15
/* the controller's constants: include/simplyp_controller.h, the same macros the kernels' SysAug::* are defined from */
#define AUG_RESYNC SIMPLYP_CTRL_RESYNC_EVERY
#define AUG_NERR SIMPLYP_CTRL_N_ERR
#define AUG_AUX_WEIGHT SIMPLYP_CTRL_AUX_WEIGHT       /* SysAug::AUX_WEIGHT */
#define AUG_KINK_SOIL SIMPLYP_CTRL_KINK_SOIL         /* SysAug::KINK_SOIL: a step across a knee of a soil-water gate */
#define AUG_KINK_REACH SIMPLYP_CTRL_KINK_REACH       /* SysAug::KINK_REACH: the end of the step is looked for this far along the first slope */
#define AUG_KNEE_LO SIMPLYP_CTRL_KNEE_LO             /* SysAug::KNEE_LO, KNEE_HI, KNEE_OVER: a knee between 2 % and 90 % of the step, along the */
#define AUG_KNEE_HI SIMPLYP_CTRL_KNEE_HI             /* first slope, ends the step 5 % past the knee (no inflation for such a step)            */
#define AUG_KNEE_OVER SIMPLYP_CTRL_KNEE_OVER
#define AUG_DAY_START SIMPLYP_CTRL_DAY_START         /* SysAug::DAY_START: share of the carried step size a new day starts with */
#define AUG_KNEE_GW SIMPLYP_CTRL_KNEE_GW             /* SysAug::KNEE_GW: a step aimed at a knee of the groundwater gate keeps this factor */
#define AUG_KINK_GW SIMPLYP_CTRL_KINK_GW             /* SysAug::KINK_GW: across a knee of the groundwater gate (zone 1 % of Qg_min wide) */
static void ode_aug(const double* z, const ode_params* p, double invKv, double* dz)
{
    double VsA = z[0], VsS = z[1], Vg = z[2], Qr = z[3], Msus = z[4], TDPr = z[5], PPr = z[6];
    double EA = z[7], ES = z[8], pb = z[9], pk = z[10];
    double QsA = (VsA - p->fc) * f_x(VsA, p->fc, 0.01) / p->T_s_A;                              /* :105 */
    double dVsA = p->P * (1 - p->f_quick) - p->alpha * p->E * (1 - EA) - QsA;                   /* :106 */
    double QsS = (VsS - p->fc) * f_x(VsS, p->fc, 0.01) / p->T_s_S;                              /* :109 */
    double dVsS = p->P * (1 - p->f_quick) - p->alpha * p->E * (1 - ES) - QsS;                   /* :110 */
    double QsNC = (p->NC_type == 1) ? QsA : QsS;
    double f_Qg = f_x(Vg / p->T_g, p->Qg_min, 0.01);
    double Qg = (1 - f_Qg) * p->Qg_min + f_Qg * (Vg / p->T_g);                                  /* :121-122 */
    double dVg = p->beta * (p->f_A * QsA + p->f_S * QsS) - Qg;                                  /* :124 */
    double inflow = p->Qq_i + (1 - p->beta) * (p->f_A * QsA + p->f_S * QsS) + Qg + p->Qr_US_i - Qr;
    double dQr = inflow * p->a_Q * pb * (8.64 * 10000) / ((1 - p->b_Q) * (p->L_reach));         /* :127-130 */
    double kap = pb * invKv;                                                                    /* Qr / Vr */
    double MA = p->Esus_A * pk, MS = p->Esus_S * pk, MIG = p->Esus_IG * pk;                     /* :138 */
    double oM = Msus * kap, oT = TDPr * kap, oP = PPr * kap;
    double dMsus = p->f_Ar * MA + p->f_IG * MIG + p->f_S * MS + p->Msus_US_i - oM;              /* :141-145 */
    double dTDPr = ((1 - p->beta) * (p->f_A * (1 - p->f_NC_A) * QsA * p->conc_TDPs_A
                                     + p->f_A * p->f_NC_A * QsNC * p->conc_TDPs_NC
                                     + p->f_S * p->f_NC_S * QsNC * p->conc_TDPs_NC)
                    + p->f_A * (1 - p->f_NC_A) * p->Qq_i * p->conc_TDPs_A
                    + p->f_A * p->f_NC_A * p->Qq_i * p->conc_TDPs_NC
                    + p->f_S * p->f_NC_S * p->Qq_i * p->conc_TDPs_NC
                    + Qg * (p->TDPg * p->A_catch) + p->TDPeff + p->TDPr_US_i - oT);             /* :154-166 */
    double dPPr = (p->E_PP * (p->f_Ar * (1 - p->f_NC_Ar) * MA * (p->PlabA_i + p->P_inactive) / p->Msoil
                              + p->f_IG * (1 - p->f_NC_IG) * MIG * (p->PlabA_i + p->P_inactive) / p->Msoil
                              + p->f_S * (1 - p->f_NC_S) * MS * p->P_inactive / p->Msoil
                              + p->f_Ar * p->f_NC_Ar * MA * (p->PlabNC_i + p->P_inactive) / p->Msoil
                              + p->f_IG * p->f_NC_IG * MIG * (p->PlabNC_i + p->P_inactive) / p->Msoil
                              + p->f_S * p->f_NC_S * MS * (p->PlabNC_i + p->P_inactive) / p->Msoil)
                   + p->PPr_US_i - oP);                                                         /* :171-178 */
    double r = dQr / Qr;
    dz[0] = dVsA; dz[1] = dVsS; dz[2] = dVg; dz[3] = dQr; dz[4] = dMsus; dz[5] = dTDPr; dz[6] = dPPr;
    dz[7] = -p->mu * EA * dVsA; dz[8] = -p->mu * ES * dVsS;
    dz[9] = p->b_Q * pb * r; dz[10] = p->k_M * pk * r;
    dz[11] = Qr; dz[12] = oM; dz[13] = oT; dz[14] = oP;                                         /* :132,:147,:168,:180 */
}

/* Same step-size rule as cashkarp_day (the kernel mirrors both).  y is the reference's 12-vector; slot 3
 * (Vr) is returned on its invariant. */
/* Probe hooks of tools/probe_c4_steps.py (round 4: where do a reach network's attempts go?), off by default.
 *   reach trace: [S][12] doubles per reach -- [0] attempts, [1] rejected, [2..7] accepted steps binned by h x (cQ Qr**b_Q), the step in
 *   units of the reach's relaxation time: < 0.5, < 1, < 2, < 3, < 4, >= 4 (Cash-Karp's real stability interval ends at 3.73);
 *   [8] sum of that rate over the attempts, [9] rejections with h x rate >= 2.  (Counters are shared by the OpenMP threads: a benign
 *   race, the probe's figures are statistics.)
 *   stability cap: attempts are cut to h x rate <= cap (0 = off): what a controller that never probes the stability limit would do. */
static double* g_reach_trace = NULL;
static _Thread_local int g_trace_reach = 0;
static double g_stab_cap = 0.0;
void simplyp_oracle_set_reach_trace(double* buf) { g_reach_trace = buf; }
void simplyp_oracle_set_stab_cap(double c) { g_stab_cap = c; }

/* The stability-optimised pair of integrator 2 (opts.stiff_pair; include/simplyp_controller.h SIMPLYP_STIFF_*, derived by
 * tools/derive_stiff_pair.py): 6 stages, order 4 with an embedded order 3, stable on the real axis down to -9.01 with every
 * stage polynomial <= 1.5 there, the sparsity of Cash-Karp's weights.  erk_aug_day uses it, attempt by attempt, where the
 * step is bound by the stability of Cash-Karp (|h x rate| <= 3.73, rate = cQ Qr**b_Q: a reach far down a network), not by
 * accuracy -- the rule is in erk_aug_day and mirrored by ck_day<SysAug, true> / ck_day_quad<true> in the kernels. */
static const erk_tableau TAB_STIFF = {
    6, 0,
    {{0}, {SIMPLYP_STIFF_A21}, {SIMPLYP_STIFF_A31, SIMPLYP_STIFF_A32}, {SIMPLYP_STIFF_A41, SIMPLYP_STIFF_A42, SIMPLYP_STIFF_A43},
     {SIMPLYP_STIFF_A51, SIMPLYP_STIFF_A52, SIMPLYP_STIFF_A53, SIMPLYP_STIFF_A54},
     {SIMPLYP_STIFF_A61, SIMPLYP_STIFF_A62, SIMPLYP_STIFF_A63, SIMPLYP_STIFF_A64, SIMPLYP_STIFF_A65}},
    {SIMPLYP_STIFF_B1, 0, SIMPLYP_STIFF_B3, SIMPLYP_STIFF_B4, 0, SIMPLYP_STIFF_B6},
    {SIMPLYP_STIFF_E1, 0, SIMPLYP_STIFF_E3, SIMPLYP_STIFF_E4, SIMPLYP_STIFF_E5, SIMPLYP_STIFF_E6},
    {0}, 0, SIMPLYP_STIFF_ERR_EXP};

static void erk_aug_day(const erk_tableau* tab, const erk_tableau* stiff, double* y, const ode_params* p, double T, double rtol, double atol,
                        int max_steps, double* h_carry, integ_stats* st)
{
    double k[ERK_MAX_STAGES][NZ], zt[NZ], zn[NZ], z[NZ];
    const int ns = tab->ns;
    const erk_tableau* const tab_default = tab;
    int have_k0 = 0;                 /* FSAL pairs: k[0] already holds the derivative at the current point */
    /* (the step size carried over the day boundary belongs to the smooth end of a day; the forcing jumps at midnight and the first
     * attempt of the new day with it was rejected on 85 % of the member-days: SysAug::DAY_START of it is the better guess) */
    double t = 0.0, h = *h_carry * AUG_DAY_START;
    int attempts = 0;
    const double Kv = p->L_reach / (p->a_Q * 8.64 * 10000), invKv = 1.0 / Kv;
    if (!(h > 0.0) || h > T) h = T;
    if (!state_finite(y)) { y[5] = y[7] = y[9] = y[11] = NAN; st->poisoned = 1; return; }
    z[0] = y[0]; z[1] = y[1]; z[2] = y[2]; z[3] = y[4]; z[4] = y[6]; z[5] = y[8]; z[6] = y[10];
    z[7] = exp(-p->mu * y[0]); z[8] = exp(-p->mu * y[1]);
    z[9] = pow(y[4], p->b_Q); z[10] = pow(y[4], p->k_M);
    z[11] = z[12] = z[13] = z[14] = 0.0;
    if (stiff) {      /* opts.stiff_pair: the day's first step reaches at most SIMPLYP_STIFF_Z_START relaxation times of the reach */
        const double rate0 = (p->a_Q * (8.64 * 10000) / ((1 - p->b_Q) * (p->L_reach))) * z[9];
        if (h * rate0 > SIMPLYP_STIFF_Z_START) h = SIMPLYP_STIFF_Z_START / rate0;
    }
    while (t < T) {
        double rem = T - t, hh = h;
        if (rem <= 1.1 * h) hh = rem; else if (rem < 2.0 * h) hh = 0.5 * rem;
        int last_chance = (attempts + 1 >= max_steps);
        if (last_chance) hh = rem;
        /* opts.stiff_pair: an attempt never reaches further than SIMPLYP_STIFF_CAP relaxation times of the reach (the second pair's real
         * stability interval ends at 9.01); the rate is the carried state cQ Qr**b_Q itself */
        const double stiff_rate = (p->a_Q * (8.64 * 10000) / ((1 - p->b_Q) * (p->L_reach))) * z[9];
        if (stiff && !last_chance && hh * stiff_rate > SIMPLYP_STIFF_CAP) hh = SIMPLYP_STIFF_CAP / stiff_rate;
        if (g_stab_cap > 0.0 && !last_chance && hh * stiff_rate > g_stab_cap) hh = g_stab_cap / stiff_rate;      /* (probe hook) */
        if (!have_k0) { ode_aug(z, p, invKv, k[0]); st->rhs += 1; }
        /* Aim at the knee (SysAug::KNEE_*): time to the nearest knee of a gate along the first slope; a knee inside the step
         * ends the step just past it, so that the right-hand side is smooth over all but its last few percent. */
        int targeted = 0, kink = 0, kink_gw = 0;
        {
            double hs = 1.0e300, hg = 1.0e300, ugd = 0.0;      /* soil boxes, groundwater; Qg gate: upper knee minus Vg / T_g - Qg_min */
            const double tlo = AUG_KNEE_LO * hh, invTg = 1.0 / p->T_g;
            for (int i = 0; i < 3; ++i) {
                const double g = (i < 2) ? z[i] - p->fc : z[2] * invTg - p->Qg_min;
                const double sl = (i < 2) ? k[0][i] : k[0][2] * invTg;
                const double gd = (i < 2) ? 0.01 * p->fc : 0.01 * p->Qg_min;
                const double gdg = gd - g;
                const double t0 = (0.0 - g) / sl, t1 = gdg / sl;
                double tn = 1.0e300;
                if (t0 > tlo && t0 < tn) tn = t0;
                if (t1 > tlo && t1 < tn) tn = t1;
                if (i < 2) { if (tn < hs) hs = tn; } else { hg = tn; ugd = gdg; }
            }
            const double hk = fmin(hs, hg);
            if (!last_chance && hk * AUG_KNEE_OVER < AUG_KNEE_HI * hh) { hh = hk * AUG_KNEE_OVER; targeted = 1; }
            /* A knee within AUG_KINK_REACH x the step along the first slope that the step was not aimed at: its estimate is not
             * trusted (SysAug::KINK_*).  A soil box that starts to drain within the step can lift Vg through its gate within the
             * same step, which the slope at the step's start (-Qg) cannot see: a soil knee within reach while Vg / T_g is below
             * the upper knee of its gate counts as a groundwater knee too. */
            const double look = AUG_KINK_REACH * hh;
            kink = hs < look;
            kink_gw = (hg < look) | (kink & (ugd > 0.0));
        }
        /* Which pair takes this attempt.  Cash-Karp (5th order) unless the step is longer than SIMPLYP_STIFF_Z_ON relaxation times of the
         * reach, where Cash-Karp's own stability ends (3.73): then the stability-optimised 4(3) pair -- except for an attempt that has a
         * knee of a gate within reach or was aimed at one: its estimate is what the knee rules were tuned on (with the second pair there a
         * storm day of a 256-reach chain came out at 4.7e-7 instead of 1.2e-7), so such an attempt is shortened to Cash-Karp's interval
         * instead.  A lane's choice depends on its own state only: results do not depend on which members share a wavefront. */
        tab = tab_default;
        if (stiff && hh * stiff_rate > SIMPLYP_STIFF_Z_ON) {
            if (!(kink || kink_gw || targeted)) tab = stiff;
            else if (!last_chance) hh = SIMPLYP_STIFF_Z_ON / stiff_rate;
        }
        for (int s = 1; s < ns; ++s) {
            for (int i = 0; i < NZ; ++i) {
                double acc = 0.0;
                for (int j = 0; j < s; ++j) acc += tab->A[s][j] * k[j][i];
                zt[i] = z[i] + hh * acc;
            }
            ode_aug(zt, p, invKv, k[s]);
        }
        st->rhs += (uint64_t)(ns - 1);
        double err = 0.0, err_fast = 0.0; int bad = 0;
        for (int i = 0; i < NZ; ++i) {
            double inc = 0.0, ee = 0.0, e3 = 0.0;
            for (int s = 0; s < ns; ++s) { inc += tab->B[s] * k[s][i]; ee += tab->E[s] * k[s][i]; e3 += tab->E3[s] * k[s][i]; }
            zn[i] = z[i] + hh * inc;
            /* error norm: the 7 physical states (see below), and Qr**k_M (z[10]) at AUG_AUX_WEIGHT x the tolerance: on a day
             * when a nearly dry reach is wetted it grows 200-fold, and its own truncation error then showed in the sediment
             * flux (SysAug::AUX_WEIGHT in simplyp_kernels.hip.h) */
            if (i >= AUG_NERR && i != 10) continue;
            /* Error scale: rtol * max(|z|, |Euler predictor|) + atol, as in the kernel -- with the two soil boxes measured from
             * field capacity and floored by the gate width 0.01 fc: what the rest of the system sees of a soil box is
             * Vs - fc (gate argument and flow, model.py:105-110), ~1 mm out of ~300, so an error of 1e-8 |Vs| would be
             * 3e-6 of the flow.  (SysAug::SOIL_REL in simplyp_kernels.hip.h.) */
            const double dgate = 0.01 * p->fc;
            double ref = (i < 2) ? z[i] - p->fc : z[i];
            double pred = ref + hh * k[0][i];
            double w = fmax(fabs(ref), fabs(pred));
            if (i < 2) w = fmax(w, dgate);
            double sc = atol + rtol * w;
            if (i == 10) sc = AUG_AUX_WEIGHT * atol + (AUG_AUX_WEIGHT * rtol) * w;
            double r = fabs(hh * ee) / sc;
            if (tab->two_est) {
                const double r3 = fabs(hh * e3) / sc, den = sqrt(r * r + 0.01 * r3 * r3);
                r = den > 0.0 ? r * r / den : 0.0;
            }
            /* (the reach's flow, its three masses and Qr**k_M apart: their estimate may be discounted, below) */
            if ((i >= 3 && i <= 6) || i == 10) { if (r > err_fast) err_fast = r; }
            else if (r > err) err = r;
        }
        if (stiff) {
            /* opts.stiff_pair: damping-aware weights (include/simplyp_controller.h, SIMPLYP_DAMP_*) -- what the reach forgets at rate lam is
             * allowed F x the tolerance */
            double lam = stiff_rate - p->b_Q * k[0][3] / z[3];
            if (lam > stiff_rate) lam = stiff_rate;
            double F = fmin(lam * T * (1.0 / SIMPLYP_DAMP_PHI), 1.0 + lam * (rem - hh));
            F = fmin(fmax(F, 1.0), SIMPLYP_DAMP_FMAX);
            err_fast = err_fast / F;
        }
        if (err_fast > err) err = err_fast;
        {   /* expansive reach (SIMPLYP_CTRL_EXPAND, SysAug::EXPAND): d(dQr/dt)/dQr > 0  <=>  b_Q x (net inflow) > Qr; k[0][3] = inflow cQ pb */
            const double cQ = p->a_Q * (8.64 * 10000) / ((1 - p->b_Q) * (p->L_reach));
            if (p->b_Q * k[0][3] > z[3] * (cQ * z[9])) err *= SIMPLYP_CTRL_EXPAND;
        }
        if (kink_gw) err *= targeted ? AUG_KNEE_GW : AUG_KINK_GW;     /* a crossing the first slope did not announce: accepted only if short */
        else if (kink && !targeted) err *= AUG_KINK_SOIL;            /* (a step that ends at a soil knee: no inflation) */
        if (!(err < 1.0e300)) bad = 1;
        for (int i = 0; i < 11; ++i) if (!(fabs(zn[i]) < 1.0e300)) bad = 1;
        ++attempts;
        if (g_reach_trace) {      /* (probe hook) */
            double* tr = g_reach_trace + 12 * (size_t)g_trace_reach;
            const double hl = hh * stiff_rate;
            const int acc = !bad && (err <= 1.0 || last_chance);
            tr[0] += 1; tr[8] += stiff_rate;
            if (!acc) { tr[1] += 1; if (hl >= 2.0) tr[9] += 1; }
            else tr[2 + (hl < 0.5 ? 0 : hl < 1 ? 1 : hl < 2 ? 2 : hl < 3 ? 3 : hl < 4 ? 4 : 5)] += 1;
        }
        if (last_chance) st->capped = 1;
        if (bad && (last_chance || hh <= 1.0e-9 * T)) {
            for (int i = 0; i < NY; ++i) y[i] = NAN;
            st->poisoned = 1;
            *h_carry = h;
            return;
        } else if (!bad && (err <= 1.0 || last_chance)) {
            memcpy(z, zn, sizeof(zn));
            t = (hh == rem) ? T : t + hh;
            st->steps++;
            /* first same as last: the last stage was evaluated at the new point */
            if (tab->fsal) { memcpy(k[0], k[ns - 1], sizeof(k[0])); have_k0 = 1; }
        } else {
            st->rejected++;              /* (k[0] stays valid: same point) */
            if (tab->fsal) have_k0 = 1;
        }
        /* pb, pk ride a neutrally stable manifold (nothing damps their drift from Qr**b, Qr**k) and are not in the error
         * norm, so on a storm day of 100+ steps the local errors would add up: re-evaluate them after every AUG_RESYNC-th
         * attempt of the day, accepted or not (a member's attempts are its own history; in the kernel the lanes of a
         * wavefront attempt in lockstep, so this test is wave-uniform there) */
        if (attempts % AUG_RESYNC == 0 && t < T) { z[9] = pow(z[3], p->b_Q); z[10] = pow(z[3], p->k_M); have_k0 = 0; }
        double fac;
        if (bad) fac = SIMPLYP_CTRL_FAC_MIN;
        else if (err == 0.0) fac = SIMPLYP_CTRL_FAC_MAX;
        else { fac = SIMPLYP_CTRL_SAFETY * pow(err, tab->err_exp); if (fac < SIMPLYP_CTRL_FAC_MIN) fac = SIMPLYP_CTRL_FAC_MIN; if (fac > SIMPLYP_CTRL_FAC_MAX) fac = SIMPLYP_CTRL_FAC_MAX; }
        /* (a step that was cut to end at a knee and accepted does not shorten the step size carried on) */
        if (!(targeted && !bad && err <= 1.0 && hh * fac < h)) h = hh * fac;
    }
    *h_carry = h;
    y[0] = z[0]; y[1] = z[1]; y[2] = z[2]; y[4] = z[3]; y[6] = z[4]; y[8] = z[5]; y[10] = z[6];
    y[5] = z[11]; y[7] = z[12]; y[9] = z[13]; y[11] = z[14];
    y[3] = Kv * pow(y[4], 1.0 - p->b_Q);
}

static void cashkarp_aug_day(double* y, const ode_params* p, double T, double rtol, double atol,
                             int max_steps, double* h_carry, integ_stats* st, int stiff)
{
    erk_aug_day(&TAB_CASHKARP, stiff ? &TAB_STIFF : NULL, y, p, T, rtol, atol, max_steps, h_carry, st);
}

/* ------------------------------------------------------------------------------------- */
/*
 * SIMPLYP_INTEG_CASHKARP_AUG_F32 -- the same-arithmetic mirror of the kernel's reduced-precision mode (BASELINE config
 * C5: "fp32 state + fp64 mass-balance accumulators"; SysAugF / DayConstF / ck_day<SysAugF> in
 * simplyp_amd/csrc/simplyp_kernels.hip.h).  NOT a parity-grade mode: fp32 stages cannot resolve rtol below ~3e-6; this
 * mirror exists so that the fp32 kernel has a checker that does the same thing, not one that is merely close.
 *   - the 11 Runge-Kutta states, every stage, the error norm and the step-size control are `float`;
 *   - the day constants are formed in double exactly as the kernel hoists them (DayConst) and rounded to float once per day;
 *   - the auxiliary states start each day from double-precision exp / pow rounded to float, and are re-evaluated in float
 *     after every 8th attempt;
 *   - the four daily integrals are accumulated in double from float increments; the state carried to the next day, Vr on
 *     its invariant, the soil-P update and everything else outside the day's integration stay double.
 * Remaining differences to the kernel: libm logf/expf/powf against v_log_f32 / v_exp_f32 (1-2 ulp of float), a true float
 * division against v_rcp_f32 (+ one Newton step), and 0.9 err^-0.2 through powf -- all far below the step controller's
 * own tolerance; an accept/reject decision may flip on them, after which the two runs are two valid integrations at the
 * same rtol.  Tests compare at 10 x rtol.
 */
typedef struct {
    float c0, aE, mu, fc, inv_d, invTsA, invTsS, invTg, Qgmin, inv_dg, beta, fA, fS, qin, omb, cQ, bQ, kM,
          Esum, MsusUS, tA, tS, tg, tconst, cPP, PPrUS, invKv;
} dayconst_f;

static float gate_f(float u, float inv_d)            /* f_x as one clamped cubic, like the kernel's gate() */
{
    float s = u * inv_d;
    s = fminf(fmaxf(s, 0.0f), 1.0f);
    return s * s * fmaf(-2.0f, s, 3.0f);
}

static void dayconst_from_params(const ode_params* p, dayconst_f* c)
{
    const double omb = 1.0 - p->beta;
    const double wA = p->f_A * (1 - p->f_NC_A);                       /* TDP source weights (:155-161) */
    const double wNC = p->f_A * p->f_NC_A + p->f_S * p->f_NC_S;
    const double tNC = omb * wNC * p->conc_TDPs_NC;
    const double Kv = p->L_reach / (p->a_Q * 8.64 * 10000);
    const double pA = (p->PlabA_i + p->P_inactive) / p->Msoil, pN = (p->PlabNC_i + p->P_inactive) / p->Msoil;
    const double p0 = p->P_inactive / p->Msoil;
    const double inv_dg = (p->Qg_min * 0.01 > 0.0) ? 1.0 / (p->Qg_min * 0.01) : 1.0e300;
    c->c0 = (float)(p->P * (1 - p->f_quick));
    c->aE = (float)(p->alpha * p->E);
    c->mu = (float)p->mu; c->fc = (float)p->fc; c->inv_d = (float)(1.0 / (0.01 * p->fc));
    c->invTsA = (float)(1.0 / p->T_s_A); c->invTsS = (float)(1.0 / p->T_s_S);
    c->invTg = (float)(1.0 / p->T_g); c->Qgmin = (float)p->Qg_min;
    c->inv_dg = (float)fmin(inv_dg, 1.0e30);
    c->beta = (float)p->beta; c->fA = (float)p->f_A; c->fS = (float)p->f_S;
    c->qin = (float)(p->Qq_i + p->Qr_US_i);
    c->omb = (float)omb;
    c->cQ = (float)(p->a_Q * (8.64 * 10000) / ((1 - p->b_Q) * p->L_reach));
    c->bQ = (float)p->b_Q; c->kM = (float)p->k_M;
    c->Esum = (float)(p->f_Ar * p->Esus_A + p->f_IG * p->Esus_IG + p->f_S * p->Esus_S);
    c->MsusUS = (float)p->Msus_US_i;
    c->tA = (float)(omb * wA * p->conc_TDPs_A + (p->NC_type == 1 ? tNC : 0.0));
    c->tS = (float)(p->NC_type == 1 ? 0.0 : tNC);
    c->tg = (float)(p->TDPg * p->A_catch);
    c->tconst = (float)(p->Qq_i * (wA * p->conc_TDPs_A + wNC * p->conc_TDPs_NC) + p->TDPeff + p->TDPr_US_i);
    c->cPP = (float)(p->E_PP * (p->f_Ar * p->Esus_A * ((1 - p->f_NC_Ar) * pA + p->f_NC_Ar * pN)
                                + p->f_IG * p->Esus_IG * ((1 - p->f_NC_IG) * pA + p->f_NC_IG * pN)
                                + p->f_S * p->Esus_S * ((1 - p->f_NC_S) * p0 + p->f_NC_S * pN)));
    c->PPrUS = (float)p->PPr_US_i;
    c->invKv = (float)(1.0 / Kv);
}

/* SysAugF::f, operation for operation.  z[11] = VsA VsS Vg Qr Msus TDPr PPr EA ES pb pk; q[4] = integrands of the
 * daily integrals. */
static void ode_aug_f32(const float* z, const dayconst_f* c, float* dz, float* q)
{
    const float uA = z[0] - c->fc, uS = z[1] - c->fc;
    const float QsA = uA * gate_f(uA, c->inv_d) * c->invTsA;
    const float QsS = uS * gate_f(uS, c->inv_d) * c->invTsS;
    dz[0] = fmaf(c->aE, z[7] - 1.0f, c->c0) - QsA;
    dz[1] = fmaf(c->aE, z[8] - 1.0f, c->c0) - QsS;
    const float Qsum = fmaf(c->fA, QsA, c->fS * QsS);
    const float ug = fmaf(z[2], c->invTg, -c->Qgmin);
    const float Qg = fmaf(gate_f(ug, c->inv_dg), ug, c->Qgmin);
    dz[2] = fmaf(c->beta, Qsum, -Qg);
    const float Qr = z[3], pb = z[9], pk = z[10];
    const float inflow = fmaf(c->omb, Qsum, c->qin) + Qg - Qr;
    const float dQr = inflow * c->cQ * pb;
    dz[3] = dQr;
    const float kap = pb * c->invKv;
    const float oM = z[4] * kap, oT = z[5] * kap, oP = z[6] * kap;
    dz[4] = fmaf(c->Esum, pk, c->MsusUS) - oM;
    dz[5] = fmaf(c->tA, QsA, fmaf(c->tS, QsS, fmaf(c->tg, Qg, c->tconst))) - oT;
    dz[6] = fmaf(c->cPP, pk, c->PPrUS) - oP;
    dz[7] = -c->mu * z[7] * dz[0];
    dz[8] = -c->mu * z[8] * dz[1];
    const float r = dQr * (1.0f / Qr);
    dz[9] = c->bQ * pb * r;
    dz[10] = c->kM * pk * r;
    q[0] = Qr; q[1] = oM; q[2] = oT; q[3] = oP;
}

static void cashkarp_aug_f32_day(double* y, const ode_params* p, double T_, double rtol_, double atol_,
                                 int max_steps, double* h_carry, integ_stats* st)
{
    enum { NS = 11 };
    const float T = (float)T_, rtol = (float)rtol_, atol = (float)atol_;
    float A[6][5], B[6], Ee[6];
    for (int s = 0; s < 6; ++s) { B[s] = (float)CK_B[s]; Ee[s] = (float)CK_E[s]; for (int j = 0; j < 5; ++j) A[s][j] = (float)CK_A[s][j]; }
    float z[NS], zt[NS], k[6][NS], kq[4], sq[4];
    double yq[4] = {0.0, 0.0, 0.0, 0.0};
    float t = 0.0f, h = (float)*h_carry;
    int attempts = 0;
    const double Kv = p->L_reach / (p->a_Q * 8.64 * 10000);
    if (!(h > 0.0f) || h > T) h = T;
    dayconst_f c;
    dayconst_from_params(p, &c);
    z[0] = (float)y[0]; z[1] = (float)y[1]; z[2] = (float)y[2]; z[3] = (float)y[4];
    z[4] = (float)y[6]; z[5] = (float)y[8]; z[6] = (float)y[10];
    z[7] = (float)exp(-p->mu * y[0]); z[8] = (float)exp(-p->mu * y[1]);
    z[9] = (float)pow(y[4], p->b_Q); z[10] = (float)pow(y[4], p->k_M);
    int alive = 1;
    for (int i = 0; i < NS; ++i) if (!(fabsf(z[i]) < 1.0e30f)) alive = 0;      /* (the kernel tests the float states) */
    if (!alive) { y[5] = y[7] = y[9] = y[11] = NAN; st->poisoned = 1; return; }
    while (t < T) {
        const float rem = T - t;
        float hh = h;
        if (rem < 2.0f * h) hh = 0.5f * rem;
        if (rem <= 1.1f * h) hh = rem;
        const int last_chance = (attempts + 1 >= max_steps);
        if (last_chance) hh = rem;
        ode_aug_f32(z, &c, k[0], kq);
        for (int i = 0; i < 4; ++i) sq[i] = B[0] * kq[i];
        for (int s = 1; s < 6; ++s) {
            /* stage weights premultiplied by the step, nested FMAs from the first stage outwards -- the kernel's order */
            float hw[5];
            for (int j = 0; j < s; ++j) hw[j] = hh * A[s][j];
            for (int i = 0; i < NS; ++i) {
                float acc = z[i];
                for (int j = 0; j < s; ++j) acc = fmaf(hw[j], k[j][i], acc);
                zt[i] = acc;
            }
            ode_aug_f32(zt, &c, k[s], kq);
            if (s == 2 || s == 3 || s == 5) for (int i = 0; i < 4; ++i) sq[i] = fmaf(B[s], kq[i], sq[i]);
        }
        st->rhs += 6;
        float err = 0.0f, chk = 0.0f;
        const float he[6] = {hh * Ee[0], 0.0f, hh * Ee[2], hh * Ee[3], hh * Ee[4], hh * Ee[5]};
        for (int i = 0; i < AUG_NERR; ++i) {          /* the plain controller: SysAugF has no knee logic (see the kernel) */
            const float e = fmaf(he[0], k[0][i], fmaf(he[2], k[2][i], fmaf(he[3], k[3][i], fmaf(he[4], k[4][i], he[5] * k[5][i]))));
            if (i >= 3) chk += e;
            const float sc = fmaf(rtol, fmaxf(fabsf(z[i]), fabsf(fmaf(hh, k[0][i], z[i]))), atol);
            const float r = fabsf(e) * (1.0f / sc);
            if (r > err) err = r;
        }
        const int bad = !(err < 1.0e30f) || !(fabsf(chk) < 1.0e30f);
        ++attempts;
        if (last_chance) st->capped = 1;
        const int accept = !bad && (err <= 1.0f || last_chance);
        if (bad && (last_chance || hh <= 1.0e-9f * T)) {
            for (int i = 0; i < NY; ++i) y[i] = NAN;
            st->poisoned = 1;
            *h_carry = (double)h;
            return;
        }
        if (accept) {
            const float hb0 = hh * B[0], hb2 = hh * B[2], hb3 = hh * B[3], hb5 = hh * B[5];
            for (int i = 0; i < NS; ++i)
                z[i] = fmaf(hb5, k[5][i], fmaf(hb3, k[3][i], fmaf(hb2, k[2][i], fmaf(hb0, k[0][i], z[i]))));
            for (int i = 0; i < 4; ++i) yq[i] = fma(1.0, (double)(hh * sq[i]), yq[i]);
            t = (hh == rem) ? T : t + hh;
            st->steps++;
        } else {
            st->rejected++;
        }
        float fac;
        if (bad) fac = (float)SIMPLYP_CTRL_FAC_MIN;
        else {
            fac = (float)SIMPLYP_CTRL_SAFETY * powf(err, -0.2f);               /* err == 0 -> +inf -> 5 */
            fac = fminf(fmaxf(fac, (float)SIMPLYP_CTRL_FAC_MIN), (float)SIMPLYP_CTRL_FAC_MAX);
        }
        h = hh * fac;
        if (attempts % AUG_RESYNC == 0 && t < T) {
            const float lq = logf(z[3]);
            z[9] = expf(c.bQ * lq); z[10] = expf(c.kM * lq);
        }
    }
    *h_carry = (double)h;
    y[0] = z[0]; y[1] = z[1]; y[2] = z[2]; y[4] = z[3]; y[6] = z[4]; y[8] = z[5]; y[10] = z[6];
    y[5] = yq[0]; y[7] = yq[1]; y[9] = yq[2]; y[11] = yq[3];
    y[3] = Kv * pow(y[4], 1.0 - p->b_Q);
}


/* ------------------------------------------------------------------------------------- */
/*
 * ORACLE_INTEG_SPLIT_AUG -- a PROBE of this file only (tools/probe_split.py; not in include/simplyp.h, no kernel): scheme 2 with the
 * slow stores split off, the numerical design behind DESIGN.md section 7 "Next for C4".
 *
 * The two soil boxes and the groundwater store do not depend on the reach's own states (model.py:105-124: forcing, member parameters
 * and the land-use shares only), and they are slow: a handful of steps a day resolve them where the reach -- whose flow equation
 * relaxes in 1/400 to 1/15 of a day and restarts a transient at every midnight -- takes ~30.  A day is integrated in two passes per
 * segment:
 *   slow pass:  w = VsA VsS Vg EA ES by Cash-Karp under the knee-aware controller (all three gates live here), tolerances
 *               g_split_slow_tol x the run's; each accepted step leaves one interval of a record: its length and the Hermite
 *               polynomials (cubic, or quintic with the second derivatives), in the time since the interval's start, of the two
 *               combinations the reach sees --
 *                   L = (1 - beta)(f_A QsA + f_S QsS) + Qg      (land-phase inflow, :127-129)
 *                   M = tA QsA + tS QsS + tg Qg                 (its TDP load, :154-163)
 *               with their exact time derivatives at the knots (chain rule through the gates);
 *   reach pass: z = Qr Msus TDPr PPr pb pk (+ the four daily integrals) over the record's intervals, one after the other, a step
 *               never crossing an interval's end -- same pairs, same controller as scheme 2 for what is left: no gate, hence no knee rule.
 * A segment holds at most g_split_ni intervals (a kernel would keep the record in LDS).
 * What it showed (profiles/r04_c4/split_prototype.log): the split itself is sound (2.5e-10 from the converged scheme at rtol 1e-11),
 * the reach pass needs 6 instead of 11 states and a right-hand side a third the size -- but the record, not the slow integration,
 * sets the slow pass's step: a gate's zone is 1 % of its threshold wide and the flow through it a quartic of the store, so a cubic
 * record needs ~14-20 knots a day to hold the references' fixtures at 2e-7, and a quintic one (exact on the quartic) is thrown by the
 * jump of L'' at a gate's upper knee unless it has as many.
 */
#define ORACLE_INTEG_SPLIT_AUG 14
#define SPLIT_NI_MAX 64
typedef struct { int n; double hk[SPLIT_NI_MAX], L[SPLIT_NI_MAX][6], M[SPLIT_NI_MAX][6]; } split_record;
static double g_split_slow_tol = 1.0e-4;
static int g_split_ni = 6, g_split_carry = 1, g_split_cut_keeps = 1;
static double g_split_h0 = 0.05;
static int g_split_order = 5;
void simplyp_oracle_split_order(int o) { g_split_order = o; }
static uint64_t g_split_slow_attempts = 0, g_split_reach_attempts = 0, g_split_segments = 0, g_split_slow_max = 0;
void simplyp_oracle_split_config(double slow_tol, int ni, int carry, int cut_keeps, double h0)
{
    g_split_slow_tol = slow_tol; g_split_ni = ni; g_split_carry = carry; g_split_cut_keeps = cut_keeps; g_split_h0 = h0;
    g_split_slow_attempts = g_split_reach_attempts = g_split_segments = g_split_slow_max = 0;
}
void simplyp_oracle_split_counts(uint64_t* c4) { c4[0] = g_split_slow_attempts; c4[1] = g_split_reach_attempts; c4[2] = g_split_segments; c4[3] = g_split_slow_max; }

typedef struct {            /* the day's constants in the kernel's grouping (DayConst) */
    double c0, aE, mu, fc, inv_d, invTsA, invTsS, invTg, Qgmin, inv_dg, beta, fA, fS, omb, tA, tS, tg, dgate, dgq;
} slow_const;

static double gate_d(double u, double inv_d, double* dg, double* d2g)      /* f_x as one clamped cubic; its first and second derivative in u */
{
    double sc = u * inv_d;
    const int inside = sc > 0.0 && sc < 1.0;
    sc = fmin(fmax(sc, 0.0), 1.0);
    *dg = 6.0 * sc * (1.0 - sc) * inv_d;
    *d2g = inside ? (6.0 - 12.0 * sc) * inv_d * inv_d : 0.0;
    return sc * sc * (3.0 - 2.0 * sc);
}

/* dw/dt of the slow stores; lm = L, M and (dlm) their first and second time derivatives at this point: dlm = L', M', L'', M'' */
static void slow_rhs(const double* w, const slow_const* c, double* dw, double* lm, double* dlm)
{
    double dgA, dgS, dgG, hgA, hgS, hgG;
    const double uA = w[0] - c->fc, uS = w[1] - c->fc;
    const double gA = gate_d(uA, c->inv_d, &dgA, &hgA), gS = gate_d(uS, c->inv_d, &dgS, &hgS);
    const double QsA = uA * c->invTsA * gA, QsS = uS * c->invTsS * gS;
    dw[0] = c->c0 + c->aE * (w[3] - 1.0) - QsA;
    dw[1] = c->c0 + c->aE * (w[4] - 1.0) - QsS;
    const double Qsum = c->fA * QsA + c->fS * QsS;
    const double ug = w[2] * c->invTg - c->Qgmin;
    const double gG = gate_d(ug, c->inv_dg, &dgG, &hgG);
    const double Qg = c->Qgmin + gG * ug;
    dw[2] = c->beta * Qsum - Qg;
    dw[3] = -c->mu * w[3] * dw[0];
    dw[4] = -c->mu * w[4] * dw[1];
    lm[0] = c->omb * Qsum + Qg;
    lm[1] = c->tA * QsA + c->tS * QsS + c->tg * Qg;
    if (dlm) {
        const double qA1 = (gA + uA * dgA) * c->invTsA, qS1 = (gS + uS * dgS) * c->invTsS, qG1 = gG + ug * dgG;       /* dQ/du */
        const double qA2 = (2.0 * dgA + uA * hgA) * c->invTsA, qS2 = (2.0 * dgS + uS * hgS) * c->invTsS, qG2 = 2.0 * dgG + ug * hgG;
        const double dQsA = qA1 * dw[0], dQsS = qS1 * dw[1];
        const double ugd = c->invTg * dw[2];
        const double dQg = qG1 * ugd;
        const double uA2 = c->aE * dw[3] - dQsA, uS2 = c->aE * dw[4] - dQsS;           /* d2 Vs / dt2 */
        const double ug2 = c->invTg * (c->beta * (c->fA * dQsA + c->fS * dQsS) - dQg);
        const double d2QsA = qA2 * dw[0] * dw[0] + qA1 * uA2, d2QsS = qS2 * dw[1] * dw[1] + qS1 * uS2;
        const double d2Qg = qG2 * ugd * ugd + qG1 * ug2;
        dlm[0] = c->omb * (c->fA * dQsA + c->fS * dQsS) + dQg;
        dlm[1] = c->tA * dQsA + c->tS * dQsS + c->tg * dQg;
        dlm[2] = c->omb * (c->fA * d2QsA + c->fS * d2QsS) + d2Qg;
        dlm[3] = c->tA * d2QsA + c->tS * d2QsS + c->tg * d2Qg;
    }
}

/* Hermite interpolant on [0, hk] in powers of the time since the interval's start: cubic from values and slopes (order 3), quintic
 * with the second derivatives too (order 5; c[4], c[5] are 0 for the cubic) */
static void hermite(int order, double f0, double d0, double s0, double f1, double d1, double s1, double hk, double* c)
{
    const double ih = 1.0 / hk, sl = (f1 - f0) * ih;
    c[0] = f0; c[1] = d0;
    if (order == 3) {
        c[2] = (3.0 * sl - 2.0 * d0 - d1) * ih;
        c[3] = ((d0 + d1) - 2.0 * sl) * ih * ih;
        c[4] = c[5] = 0.0;
    } else {
        const double a = s0 * hk, b = s1 * hk;      /* second derivatives x hk: same units as the slopes */
        c[2] = 0.5 * s0;
        c[3] = 0.5 * (20.0 * sl - 8.0 * d1 - 12.0 * d0 - (3.0 * a - b)) * ih * ih;
        c[4] = 0.5 * (-30.0 * sl + 14.0 * d1 + 16.0 * d0 + (3.0 * a - 2.0 * b)) * ih * ih * ih;
        c[5] = 0.5 * (12.0 * sl - 6.0 * (d1 + d0) - (a - b)) * ih * ih * ih * ih;
    }
}

typedef struct { double qin, cQ, bQ, kM, Esum, MsusUS, tconst, cPP, PPrUS, invKv; } reach_const;
#define NRZ 10     /* Qr Msus TDPr PPr pb pk | Qr_av Msus_out TDP_out PP_out */
static void reach_rhs(double tau, const double* z, const reach_const* c, const double* cl, const double* cm, double* dz)
{
    const double L = cl[0] + tau * (cl[1] + tau * (cl[2] + tau * (cl[3] + tau * (cl[4] + tau * cl[5]))));
    const double M = cm[0] + tau * (cm[1] + tau * (cm[2] + tau * (cm[3] + tau * (cm[4] + tau * cm[5]))));
    const double Qr = z[0], pb = z[4], pk = z[5];
    const double inflow = (L + c->qin) - Qr;
    const double dQr = inflow * (c->cQ * pb);
    const double kap = pb * c->invKv;
    const double oM = z[1] * kap, oT = z[2] * kap, oP = z[3] * kap;
    dz[0] = dQr;
    dz[1] = c->Esum * pk + c->MsusUS - oM;
    dz[2] = (M + c->tconst) - oT;
    dz[3] = c->cPP * pk + c->PPrUS - oP;
    const double r = dQr / Qr;
    dz[4] = c->bQ * pb * r; dz[5] = c->kM * pk * r;
    dz[6] = Qr; dz[7] = oM; dz[8] = oT; dz[9] = oP;
}

static void split_day(double* y, const ode_params* p, double T, double rtol, double atol, int max_steps,
                      double* h_carry, double* h_slow, integ_stats* st, int use_stiff)
{
    static const double CN[6] = {0.0, 1.0 / 5, 3.0 / 10, 3.0 / 5, 1.0, 7.0 / 8};       /* Cash-Karp's nodes */
    double cS[6]; { cS[0] = 0.0; for (int s = 1; s < 6; ++s) { double a = 0; for (int j = 0; j < s; ++j) a += TAB_STIFF.A[s][j]; cS[s] = a; } }
    slow_const sc; reach_const rc;
    if (!state_finite(y)) { y[5] = y[7] = y[9] = y[11] = NAN; st->poisoned = 1; return; }
    {
        const double wA = p->f_A * (1 - p->f_NC_A), wNC = p->f_A * p->f_NC_A + p->f_S * p->f_NC_S;
        sc.c0 = p->P * (1 - p->f_quick); sc.aE = p->alpha * p->E; sc.mu = p->mu; sc.fc = p->fc; sc.inv_d = 1.0 / (0.01 * p->fc);
        sc.invTsA = 1.0 / p->T_s_A; sc.invTsS = 1.0 / p->T_s_S; sc.invTg = 1.0 / p->T_g; sc.Qgmin = p->Qg_min;
        sc.inv_dg = (p->Qg_min * 0.01 > 0.0) ? 1.0 / (p->Qg_min * 0.01) : 1.0e300;
        sc.beta = p->beta; sc.fA = p->f_A; sc.fS = p->f_S; sc.omb = 1.0 - p->beta;
        const double tNC = sc.omb * wNC * p->conc_TDPs_NC;
        sc.tA = sc.omb * wA * p->conc_TDPs_A + (p->NC_type == 1 ? tNC : 0.0);
        sc.tS = (p->NC_type == 1 ? 0.0 : tNC);
        sc.tg = p->TDPg * p->A_catch; sc.dgate = 0.01 * p->fc; sc.dgq = 0.01 * p->Qg_min;
        rc.qin = p->Qq_i + p->Qr_US_i;
        rc.cQ = p->a_Q * (8.64 * 10000) / ((1 - p->b_Q) * (p->L_reach)); rc.bQ = p->b_Q; rc.kM = p->k_M;
        rc.Esum = p->f_Ar * p->Esus_A + p->f_IG * p->Esus_IG + p->f_S * p->Esus_S; rc.MsusUS = p->Msus_US_i;
        rc.tconst = p->Qq_i * (wA * p->conc_TDPs_A + wNC * p->conc_TDPs_NC) + p->TDPeff + p->TDPr_US_i;
        const double pA = (p->PlabA_i + p->P_inactive) / p->Msoil, pN = (p->PlabNC_i + p->P_inactive) / p->Msoil, p0 = p->P_inactive / p->Msoil;
        rc.cPP = p->E_PP * (p->f_Ar * p->Esus_A * ((1 - p->f_NC_Ar) * pA + p->f_NC_Ar * pN)
                            + p->f_IG * p->Esus_IG * ((1 - p->f_NC_IG) * pA + p->f_NC_IG * pN)
                            + p->f_S * p->Esus_S * ((1 - p->f_NC_S) * p0 + p->f_NC_S * pN));
        rc.PPrUS = p->PPr_US_i;
        rc.invKv = (p->a_Q * 8.64 * 10000) / p->L_reach;
    }
    const double rtol_s = rtol * g_split_slow_tol, atol_s = atol * g_split_slow_tol;
    double w[5] = {y[0], y[1], y[2], exp(-p->mu * y[0]), exp(-p->mu * y[1])};
    double z[NRZ] = {y[4], y[6], y[8], y[10], pow(y[4], p->b_Q), pow(y[4], p->k_M), 0.0, 0.0, 0.0, 0.0};
    double ts = 0.0, hs = g_split_carry ? *h_slow : g_split_h0 * T;
    double h = *h_carry * AUG_DAY_START;
    int slow_attempts = 0, attempts = 0;
    if (!(hs > 0.0) || hs > T) hs = T;
    if (!(h > 0.0) || h > T) h = T;
    if (use_stiff && h * rc.cQ * z[4] > SIMPLYP_STIFF_Z_START) h = SIMPLYP_STIFF_Z_START / (rc.cQ * z[4]);
    split_record rec;
    while (ts < T) {
        /* ---- slow pass: up to g_split_ni accepted steps ---- */
        double k[6][5], wt[5], wn[5], F[2], dF[4], Fn[2], dFn[4], kn[5], lms[2];
        const erk_tableau* ck = &TAB_CASHKARP;
        rec.n = 0; ++g_split_segments;
        slow_rhs(w, &sc, k[0], F, dF);
        while (ts < T && rec.n < g_split_ni) {
            double rem = T - ts, hh = hs;
            if (rem <= 1.1 * hs) hh = rem; else if (rem < 2.0 * hs) hh = 0.5 * rem;
            const int last_chance = (slow_attempts + 1 >= max_steps);
            if (last_chance) hh = rem;
            int targeted = 0, kink = 0, kink_gw = 0;
            {
                double hsl = 1.0e300, hg = 1.0e300, ugd = 0.0;
                const double tlo = AUG_KNEE_LO * hh;
                for (int i = 0; i < 3; ++i) {
                    const double g = (i < 2) ? w[i] - sc.fc : w[2] * sc.invTg - sc.Qgmin;
                    const double sl = (i < 2) ? k[0][i] : k[0][2] * sc.invTg;
                    const double gd = (i < 2) ? sc.dgate : sc.dgq;
                    const double gdg = gd - g;
                    const double t0 = (0.0 - g) / sl, t1 = gdg / sl;
                    double tn = 1.0e300;
                    if (t0 > tlo && t0 < tn) tn = t0;
                    if (t1 > tlo && t1 < tn) tn = t1;
                    if (i < 2) { if (tn < hsl) hsl = tn; } else { hg = tn; ugd = gdg; }
                }
                const double hk = fmin(hsl, hg);
                if (!last_chance && hk * AUG_KNEE_OVER < AUG_KNEE_HI * hh) { hh = hk * AUG_KNEE_OVER; targeted = 1; }
                const double look = AUG_KINK_REACH * hh;
                kink = hsl < look;
                kink_gw = (hg < look) | (kink & (ugd > 0.0));
            }
            for (int s = 1; s < 6; ++s) {
                for (int i = 0; i < 5; ++i) {
                    double acc = 0.0;
                    for (int j = 0; j < s; ++j) acc += ck->A[s][j] * k[j][i];
                    wt[i] = w[i] + hh * acc;
                }
                slow_rhs(wt, &sc, k[s], lms, NULL);
            }
            double err = 0.0; int bad = 0;
            for (int i = 0; i < 5; ++i) {
                double inc = 0.0, ee = 0.0;
                for (int s = 0; s < 6; ++s) { inc += ck->B[s] * k[s][i]; ee += ck->E[s] * k[s][i]; }
                wn[i] = w[i] + hh * inc;
                if (!(fabs(wn[i]) < 1.0e300)) bad = 1;
                if (i >= 3) continue;
                const double ref = (i < 2) ? w[i] - sc.fc : w[i], pred = ref + hh * k[0][i];
                double wgt = fmax(fabs(ref), fabs(pred));
                if (i < 2) wgt = fmax(wgt, sc.dgate);
                const double r = fabs(hh * ee) / (atol_s + rtol_s * wgt);
                if (r > err) err = r;
            }
            if (kink_gw) err *= targeted ? AUG_KNEE_GW : AUG_KINK_GW;
            else if (kink && !targeted) err *= AUG_KINK_SOIL;
            if (!(err < 1.0e300)) bad = 1;
            ++slow_attempts; ++g_split_slow_attempts; st->rhs += 6;
            if (last_chance) st->capped = 1;
            if (bad && (last_chance || hh <= 1.0e-9 * T)) { for (int i = 0; i < NY; ++i) y[i] = NAN; st->poisoned = 1; return; }
            else if (!bad && (err <= 1.0 || last_chance)) {
                slow_rhs(wn, &sc, kn, Fn, dFn);
                const int n = rec.n;
                rec.hk[n] = hh;
                hermite(g_split_order, F[0], dF[0], dF[2], Fn[0], dFn[0], dFn[2], hh, rec.L[n]);
                hermite(g_split_order, F[1], dF[1], dF[3], Fn[1], dFn[1], dFn[3], hh, rec.M[n]);
                rec.n = n + 1;
                memcpy(w, wn, sizeof(wn)); memcpy(k[0], kn, sizeof(kn)); memcpy(F, Fn, sizeof(Fn)); memcpy(dF, dFn, sizeof(dFn));
                ts = (hh == rem) ? T : ts + hh;
                st->steps++;
            } else st->rejected++;
            double fac;
            if (bad) fac = SIMPLYP_CTRL_FAC_MIN;
            else if (err == 0.0) fac = SIMPLYP_CTRL_FAC_MAX;
            else { fac = SIMPLYP_CTRL_SAFETY * pow(err, -0.2); if (fac < SIMPLYP_CTRL_FAC_MIN) fac = SIMPLYP_CTRL_FAC_MIN; if (fac > SIMPLYP_CTRL_FAC_MAX) fac = SIMPLYP_CTRL_FAC_MAX; }
            if (!(targeted && !bad && err <= 1.0 && hh * fac < hs)) hs = hh * fac;
        }
        /* ---- reach pass over the record ---- */
        for (int iv = 0; iv < rec.n; ++iv) {
            const double hk = rec.hk[iv];
            const double* cl = rec.L[iv]; const double* cm = rec.M[iv];
            double tau = 0.0, kz[6][NRZ], zt[NRZ], zn[NRZ];
            while (tau < hk) {
                double rem = hk - tau, hh = h;
                if (rem <= 1.1 * h) hh = rem; else if (rem < 2.0 * h) hh = 0.5 * rem;
                const int last_chance = (attempts + 1 >= max_steps);
                if (last_chance) hh = rem;
                const double rate = rc.cQ * z[4];
                if (use_stiff && !last_chance && hh * rate > SIMPLYP_STIFF_CAP) hh = SIMPLYP_STIFF_CAP / rate;
                const int cut = hh < h;
                const erk_tableau* tab = &TAB_CASHKARP;
                const double* cn = CN;
                if (use_stiff && hh * rate > SIMPLYP_STIFF_Z_ON) { tab = &TAB_STIFF; cn = cS; }
                reach_rhs(tau, z, &rc, cl, cm, kz[0]);
                for (int s = 1; s < 6; ++s) {
                    for (int i = 0; i < NRZ; ++i) {
                        double acc = 0.0;
                        for (int j = 0; j < s; ++j) acc += tab->A[s][j] * kz[j][i];
                        zt[i] = z[i] + hh * acc;
                    }
                    reach_rhs(tau + cn[s] * hh, zt, &rc, cl, cm, kz[s]);
                }
                st->rhs += 6;
                double err = 0.0; int bad = 0;
                for (int i = 0; i < NRZ; ++i) {
                    double inc = 0.0, ee = 0.0;
                    for (int s = 0; s < 6; ++s) { inc += tab->B[s] * kz[s][i]; ee += tab->E[s] * kz[s][i]; }
                    zn[i] = z[i] + hh * inc;
                    if (i < 6 && !(fabs(zn[i]) < 1.0e300)) bad = 1;
                    if (i >= 4 && i != 5) continue;
                    const double wgt = fmax(fabs(z[i]), fabs(z[i] + hh * kz[0][i]));
                    double scl = atol + rtol * wgt;
                    if (i == 5) scl = AUG_AUX_WEIGHT * atol + (AUG_AUX_WEIGHT * rtol) * wgt;
                    const double r = fabs(hh * ee) / scl;
                    if (r > err) err = r;
                }
                if (rc.bQ * kz[0][0] > z[0] * rate) err *= SIMPLYP_CTRL_EXPAND;
                if (!(err < 1.0e300)) bad = 1;
                ++attempts; ++g_split_reach_attempts;
                if (last_chance) st->capped = 1;
                const int accept = !bad && (err <= 1.0 || last_chance);
                if (bad && (last_chance || hh <= 1.0e-9 * T)) { for (int i = 0; i < NY; ++i) y[i] = NAN; st->poisoned = 1; *h_carry = h; return; }
                else if (accept) { memcpy(z, zn, sizeof(zn)); tau = (hh == rem) ? hk : tau + hh; st->steps++; }
                else st->rejected++;
                if (attempts % AUG_RESYNC == 0) { z[4] = pow(z[0], rc.bQ); z[5] = pow(z[0], rc.kM); }
                double fac;
                if (bad) fac = SIMPLYP_CTRL_FAC_MIN;
                else if (err == 0.0) fac = SIMPLYP_CTRL_FAC_MAX;
                else { fac = SIMPLYP_CTRL_SAFETY * pow(err, tab->err_exp); if (fac < SIMPLYP_CTRL_FAC_MIN) fac = SIMPLYP_CTRL_FAC_MIN; if (fac > SIMPLYP_CTRL_FAC_MAX) fac = SIMPLYP_CTRL_FAC_MAX; }
                /* (a step that was cut short -- by the interval's end or the cap -- and accepted does not shorten the step size carried on) */
                if (!(g_split_cut_keeps && cut && accept && hh * fac < h)) h = hh * fac;
            }
        }
    }
    if ((uint64_t)slow_attempts > g_split_slow_max) g_split_slow_max = (uint64_t)slow_attempts;
    *h_carry = h; *h_slow = hs;
    y[0] = w[0]; y[1] = w[1]; y[2] = w[2];
    y[4] = z[0]; y[6] = z[1]; y[8] = z[2]; y[10] = z[3];
    y[5] = z[6]; y[7] = z[7]; y[9] = z[8]; y[11] = z[9];
    y[3] = p->L_reach / (p->a_Q * 8.64 * 10000) * pow(y[4], 1.0 - p->b_Q);
}

/* ------------------------------------------------------------------------------------- */
/* One member: the SC loop (model.py:365) around the day loop (model.py:491).              */

/* Optional trace for tools/probe_async.py: attempted steps (accepted + rejected) of every member and day, [E][D] uint16,
 * single-reach runs.  NULL = off. */
static uint16_t* g_day_attempts = NULL;
void simplyp_oracle_set_trace(uint16_t* day_attempts) { g_day_attempts = day_attempts; }

#define MP(name) (mp[(size_t)SIMPLYP_PM_##name * E + e])
#define RP(name, s) (rp[((size_t)SIMPLYP_PR_##name * S + (s)) * E + e])

static int nc_type_of(double f_NC_A, double f_NC_S)     /* model.py:325-334 */
{
    if (f_NC_A > 0) return 1;
    else if (f_NC_S > 0) return 2;
    return 0;
}

static void run_member(int e, const simplyp_dims* dims, const simplyp_opts* o, const double* forcing,
                       const int32_t* doy, const int32_t* fom, const double* mp, const double* rp,
                       const int32_t* up_ptr, const int32_t* up_idx, const int32_t* out_slot,
                       int n_out_reaches, int n_integ, double* out, int32_t* status, integ_stats* st)
{
    const int E = dims->E, S = dims->S, D = dims->D;
    /* opts.stiff_pair: 0 = auto (reach networks), > 0 on, < 0 off -- SIMPLYP_STIFF_PAIR_ON in include/simplyp_controller.h, the rule the library uses */
    const int stiff_pair = SIMPLYP_STIFF_PAIR_ON(o->stiff_pair, S) && n_integ == SIMPLYP_INTEG_CASHKARP_AUG;
    const double* Pser = forcing + (size_t)(fom ? fom[e] : 0) * (o->snow ? 3 : 2) * D;   /* Precipitation when o->snow */
    const double* Eser = Pser + D;
    const double* Tser = Eser + D;                                                        /* T_air (o->snow only) */
    /* daily series of every reach of this member that a downstream reach may read (:524-528) */
    double* route = (double*)malloc(sizeof(double) * 4 * (size_t)S * D);

    int col_of[SIMPLYP_N_OUT], ncols = 0;
    for (int c = 0; c < SIMPLYP_N_OUT; ++c) col_of[c] = (o->out_mask >> c) & 1u ? ncols++ : -1;
    int32_t stat = 0;

    const double fc = MP(FC), f_quick = MP(F_QUICK), alpha = MP(ALPHA), beta = MP(BETA), T_g = MP(T_G);
    const double Qg_min = MP(QG_MIN), a_Q = MP(A_Q), b_Q = MP(B_Q), E_M = MP(E_M), k_M = MP(K_M);
    const double T_s_A = MP(T_S_A), T_s_S = MP(T_S_S);
    const double mu = -log(0.01) / fc;                                                           /* :349 */
    const double E_risk_period = 60.0;                                                           /* :354 */
    const double d_mid[2] = {MP(D_MAXE_SPR), MP(D_MAXE_AUT)};
    const double d_start[2] = {d_mid[0] - E_risk_period / 2., d_mid[1] - E_risk_period / 2.};    /* :358 */
    const double d_end[2] = {d_mid[0] + E_risk_period / 2., d_mid[1] + E_risk_period / 2.};      /* :359 */

    /* the Python variable NC_type left over from the validation loop (:321-335) = last SC's */
    const int NC_type_leak = nc_type_of(RP(F_AR, S - 1) * RP(F_NC_AR, S - 1) + RP(F_NC_IG, S - 1) * RP(F_IG, S - 1),
                                        RP(F_NC_S, S - 1));

    for (int s = 0; s < S; ++s) {                                                                /* :365 */
        const double A_catch = RP(A_CATCH, s);
        const double f_Ar = RP(F_AR, s), f_IG = RP(F_IG, s), f_S = RP(F_S, s);
        const double f_NC_Ar = RP(F_NC_AR, s), f_NC_IG = RP(F_NC_IG, s), f_NC_S = RP(F_NC_S, s);
        const double f_A = f_IG + f_Ar;                                                          /* :318 */
        const double f_NC_A = (f_Ar * f_NC_Ar) + (f_NC_IG * f_IG);                               /* :319 */
        const int NC_type = nc_type_of(f_NC_A, f_NC_S);

        double VsA0 = fc, VsS0 = VsA0;                                                           /* :377-378 */
        double Qr0 = MP(QR0_INIT) * 86400 / (1000 * RP(A_CATCH, o->sc_qr0));                     /* :386, hf.py:29 */
        double Qg0 = beta * Qr0;                                                                 /* :389 */
        double Vg0 = Qg0 * T_g;                                                                  /* :390 */
        double TDPr0 = 0.0, PPr0 = 0.0, Msus0 = 0.0;                                             /* :396 */
        const double Msoil = MP(MSOIL_M2) * 1000000 * A_catch;                                   /* :404 */
        const double P_inactive = 1e-6 * MP(SOILPCONC_S) * Msoil;                                /* :407 */
        const double EPC0_0_A = MP(EPC0_INIT_A) * A_catch;                                       /* :412, hf.py:56 */
        const double EPC0_0_S = MP(EPC0_INIT_S) * A_catch;
        const double Plab0_A_init = 1e-6 * (MP(SOILPCONC_A) - MP(SOILPCONC_S)) * Msoil;          /* :415 */
        const double TDPs0_A_init = EPC0_0_A * VsA0;                                             /* :420 */
        const double TDPs0_S_init = 0;                                                           /* :422 */
        double Plab0_A = Plab0_A_init, TDPs0_A = TDPs0_A_init;                                   /* :426 */
        double Plab0_NC, TDPs0_NC;
        if (NC_type == 2) { Plab0_NC = Plab0_A; TDPs0_NC = TDPs0_A; }                            /* :429-431 */
        else { Plab0_NC = 0.0; TDPs0_NC = TDPs0_S_init; }                                        /* :433-434 */
        double conc_TDPs_A = TDPs0_A / VsA0;                                                     /* :438 */
        double VsNC0 = (NC_type_leak == 1) ? VsA0 : VsS0;                                        /* :442-445 */
        double conc_TDPs_NC = TDPs0_NC / VsNC0;                                                  /* :446 */
        double Kf;
        if (o->run_mode_cal) Kf = 1e-6 * (MP(SOILPCONC_A) - MP(SOILPCONC_S)) / EPC0_0_A;         /* :451 */
        else Kf = MP(KF);                                                                        /* :453 */
        const double L_reach = RP(L_REACH, s);
        double Tr0 = L_reach / (a_Q * pow(Qr0, b_Q) * 8.64 * 10000);                             /* :457-458 */
        double Vr0 = Tr0 * Qr0;                                                                  /* :459 */
        double TDPeff = RP(TDPEFF, s);
        if (isnan(TDPeff)) TDPeff = 0.;                                                          /* :462-463 */
        const double slope_A = RP(S_AR, s), slope_IG = RP(S_IG, s), slope_S = RP(S_SN, s);       /* :469 */
        const double S_reach = RP(S_REACH, s), f_spr = RP(F_SPR, s);

        double h_carry = o->step_len / (o->substeps > 0 ? o->substeps : 1);
        double h_slow = g_split_h0 * o->step_len;        /* SIMPLYP_INTEG_SPLIT: the slow stores' own step size */
        double D_snow = o->snow ? MP(D_SNOW_0) : 0.0;                                            /* inputs.py:198 */

        for (int idx = 0; idx < D; ++idx) {                                                      /* :491 */
            double P = Pser[idx], Ev = Eser[idx];                                                /* :497-498 */
            if (o->snow) {      /* snow_hydrol_inputs, inputs.py:183-208, for this member and day */
                const double T_air = Tser[idx];
                const double P_snow = (T_air < 0.0) ? P : 0.0;                                   /* :183-184 */
                const double P_rain = P - P_snow;                                                /* :187 */
                double P_melt = MP(F_DDSM) * (T_air - 0);                                        /* :190 */
                if (P_melt < 0.0) P_melt = 0.0;                                                  /* :191 */
                P_melt = (D_snow < P_melt) ? D_snow : P_melt;                                    /* :199, :204 */
                D_snow = D_snow + P_snow - P_melt;                                               /* :200, :205 */
                P = P_rain + P_melt;                                                             /* :208 */
            }
            double Qq_i = f_quick * P;                                                           /* :501 */

            double Qr_US_i = 0.0, Msus_US_i = 0.0, TDPr_US_i = 0.0, PPr_US_i = 0.0;              /* :544 */
            for (int k = up_ptr[s]; k < up_ptr[s + 1]; ++k) {                                    /* :521-538 */
                int u = up_idx[k];
                const double* r = route + ((size_t)u * D + idx) * 4;
                Qr_US_i += r[0] * (RP(A_CATCH, u) / A_catch);                                    /* :524-525 */
                Msus_US_i += r[1]; TDPr_US_i += r[2]; PPr_US_i += r[3];                          /* :526-528 */
            }

            /* sediment input coefficients, :549-594 */
            double C_cover_A;
            if (o->dynamic_erod) {
                int dayNo = doy[idx];                                                            /* :550 */
                double C_season[2];
                for (int q = 0; q < 2; ++q) {
                    /* `dayNo in np.arange(d_start, d_end)` (:567): exact float membership */
                    double kk = (double)dayNo - d_start[q];
                    int in_window = (kk >= 0.0 && kk == floor(kk) && kk < ceil(d_end[q] - d_start[q])
                                     && d_start[q] + kk == (double)dayNo);
                    if (in_window) {
                        if (dayNo < d_mid[q])                                                    /* :568-570, hf.py:90 */
                            C_season[q] = MP(C_COVER_A) + (1.0 - MP(C_COVER_A)) * (dayNo - d_start[q]) / (d_mid[q] - d_start[q]);
                        else                                                                     /* :572-573 */
                            C_season[q] = 1.0 + (MP(C_COVER_A) - 1.0) * (dayNo - d_mid[q]) / (d_end[q] - d_mid[q]);
                    } else {
                        C_season[q] = (MP(C_COVER_A) - (E_risk_period * (1 - MP(C_COVER_A))
                                                       / (2 * (365 - E_risk_period))));          /* :575-576 */
                    }
                }
                C_cover_A = (f_spr * C_season[0] + (1 - f_spr) * C_season[1]);                   /* :579-580 */
            } else {
                C_cover_A = MP(C_COVER_A);                                                       /* :583 */
            }
            double Esus_A = (E_M * S_reach * slope_A * C_cover_A * (1 - MP(C_MEAS_A)));          /* :591-594 */
            double Esus_S = (E_M * S_reach * slope_S * MP(C_COVER_S) * (1 - MP(C_MEAS_S)));
            double Esus_IG = (E_M * S_reach * slope_IG * MP(C_COVER_IG) * (1 - MP(C_MEAS_IG)));

            double EPC0_A_i, EPC0_NC_i;
            if (o->dynamic_epc0) {
                EPC0_A_i = fmax(Plab0_A / (Kf * Msoil), 0);                                      /* :602 */
                EPC0_NC_i = fmax(Plab0_NC / (Kf * Msoil), 0);                                    /* :603 */
            } else {
                EPC0_A_i = EPC0_0_A;                                                             /* :607 */
                EPC0_NC_i = (NC_type == 2) ? EPC0_0_A : EPC0_0_S;                                /* :608-611 */
            }

            double y[NY] = {VsA0, VsS0, Vg0, Vr0, Qr0, 0.0, Msus0, 0.0, TDPr0, 0.0, PPr0, 0.0};  /* :618 */
            ode_params op;                                                                       /* :622-632 */
            op.P = P; op.E = Ev; op.mu = mu; op.Qq_i = Qq_i; op.Qr_US_i = Qr_US_i;
            op.Esus_A = Esus_A; op.Esus_S = Esus_S; op.Esus_IG = Esus_IG;
            op.Msus_US_i = Msus_US_i; op.TDPr_US_i = TDPr_US_i; op.PPr_US_i = PPr_US_i;
            op.f_A = f_A; op.f_Ar = f_Ar; op.f_IG = f_IG; op.f_S = f_S; op.f_NC_A = f_NC_A;
            op.f_NC_Ar = f_NC_Ar; op.f_NC_IG = f_NC_IG; op.f_NC_S = f_NC_S; op.NC_type = NC_type;
            op.f_quick = f_quick; op.alpha = alpha; op.beta = beta; op.T_s_A = T_s_A; op.T_s_S = T_s_S;
            op.T_g = T_g; op.fc = fc; op.L_reach = L_reach; op.A_catch = A_catch; op.a_Q = a_Q; op.b_Q = b_Q;
            op.k_M = k_M; op.conc_TDPs_A = conc_TDPs_A; op.conc_TDPs_NC = conc_TDPs_NC;
            op.PlabA_i = Plab0_A; op.PlabNC_i = Plab0_NC; op.Msoil = Msoil; op.TDPeff = TDPeff;
            op.TDPg = MP(TDPG); op.E_PP = MP(E_PP); op.P_inactive = P_inactive; op.Qg_min = Qg_min;

            /* model.py:640 -- the one place that is not a restatement (see header) */
            const uint64_t attempts_before = st->steps + st->rejected;
            g_trace_reach = s;
            if (n_integ == SIMPLYP_INTEG_RK4) rk4_day(y, &op, o->step_len, o->substeps, st);
            else if (n_integ == SIMPLYP_INTEG_CASHKARP_AUG)
                cashkarp_aug_day(y, &op, o->step_len, o->rtol, o->atol, o->max_steps, &h_carry, st, stiff_pair);
            else if (n_integ == SIMPLYP_INTEG_CASHKARP_AUG_F32)
                cashkarp_aug_f32_day(y, &op, o->step_len, o->rtol, o->atol, o->max_steps, &h_carry, st);
            else if (n_integ == ORACLE_INTEG_SPLIT_AUG)
                split_day(y, &op, o->step_len, o->rtol, o->atol, o->max_steps, &h_carry, &h_slow, st, 1);
            else if (n_integ == ORACLE_INTEG_TSIT5_AUG)
                erk_aug_day(&TAB_TSIT5, NULL, y, &op, o->step_len, o->rtol, o->atol, o->max_steps, &h_carry, st);
            else if (n_integ == ORACLE_INTEG_DOP853_AUG)
                erk_aug_day(tab_dop853(), NULL, y, &op, o->step_len, o->rtol, o->atol, o->max_steps, &h_carry, st);
            else cashkarp_day(y, &op, o->step_len, o->rtol, o->atol, o->max_steps, &h_carry, st);
            if (o->project_vr && n_integ != SIMPLYP_INTEG_CASHKARP_AUG && n_integ != SIMPLYP_INTEG_CASHKARP_AUG_F32 && n_integ != ORACLE_INTEG_TSIT5_AUG && n_integ != ORACLE_INTEG_DOP853_AUG && n_integ != ORACLE_INTEG_SPLIT_AUG) {
                /* Drift control (not in the reference).  The reference's own equations (:127-131) imply
                 * dVr = dQr * (1-b_Q) L / (a_Q 86400 Qr^b_Q), and Vr0 (:457-459) starts on that curve, so
                 * Vr == L Qr^(1-b_Q) / (a_Q 86400) for all t; Vr has no restoring term and a one-step
                 * integrator random-walks off it.  Re-impose it once per day. */
                y[3] = L_reach * pow(y[4], 1.0 - b_Q) / (a_Q * 8.64 * 10000);
            }
            if (g_day_attempts && S == 1) g_day_attempts[(size_t)e * D + idx] = (uint16_t)(st->steps + st->rejected - attempts_before);
            const double* res = y;                                                               /* :643 */
            for (int i = 0; i < NY; ++i) if (!isfinite(res[i])) stat |= SIMPLYP_STATUS_NONFINITE;

            VsA0 = res[0]; VsS0 = res[1]; Vg0 = res[2]; Vr0 = res[3]; Qr0 = res[4];              /* :648-652 */
            Msus0 = res[6]; TDPr0 = res[8]; PPr0 = res[10];                                      /* :654-658 */

            double QsA0 = (VsA0 - fc) * f_x(VsA0, fc, 0.01) / T_s_A;                             /* :663 */
            double QsS0 = (VsS0 - fc) * f_x(VsS0, fc, 0.01) / T_s_S;                             /* :664 */
            double f_Qg = f_x(Vg0 / T_g, Qg_min, 0.01);                                          /* :668 */
            Qg0 = (1 - f_Qg) * Qg_min + f_Qg * (Vg0 / T_g);                                      /* :669 */
            Vg0 = Qg0 * T_g;                                                                     /* :670 */

            double QsNC0;
            if (NC_type_leak == 1) { VsNC0 = VsA0; QsNC0 = QsA0; }                               /* :676-678 */
            else { VsNC0 = VsS0; QsNC0 = QsS0; }                                                 /* :680-681 */

            if (o->dynamic_epc0) {                                                               /* :684 */
                discretized_soilP(MP(P_NETINPUT_A), A_catch, Kf, Msoil, EPC0_A_i, QsA0, Qq_i, VsA0,
                                  TDPs0_A, Plab0_A, &TDPs0_A, &Plab0_A);                         /* :688-689 */
                discretized_soilP(MP(P_NETINPUT_NC), A_catch, Kf, Msoil, EPC0_NC_i, QsNC0, Qq_i, VsNC0,
                                  TDPs0_NC, Plab0_NC, &TDPs0_NC, &Plab0_NC);                     /* :692-693 */
                /* Python max(a, 0.) returns a when a is NaN; fmax would return 0 */
                TDPs0_A = (0. > TDPs0_A) ? 0. : TDPs0_A;                                         /* :696 */
                Plab0_A = (0. > Plab0_A) ? 0. : Plab0_A;                                         /* :697 */
                TDPs0_NC = (0. > TDPs0_NC) ? 0. : TDPs0_NC;                                      /* :698 */
                Plab0_NC = (0. > Plab0_NC) ? 0. : Plab0_NC;                                      /* :699 */
                conc_TDPs_A = TDPs0_A / VsA0;                                                    /* :702 */
                conc_TDPs_NC = TDPs0_NC / VsNC0;                                                 /* :703 */
            } else {
                conc_TDPs_A = EPC0_A_i;                                                          /* :711 */
                conc_TDPs_NC = EPC0_NC_i;                                                        /* :715 */
            }

            double* r = route + ((size_t)s * D + idx) * 4;
            r[0] = res[5]; r[1] = res[7]; r[2] = res[9]; r[3] = res[11];

            int slot = out_slot[s];
            if (slot >= 0) {
                double row[SIMPLYP_N_OUT];
                for (int i = 0; i < NY; ++i) row[i] = res[i];                                    /* :644 */
                row[12] = Qq_i; row[13] = QsA0; row[14] = QsS0; row[15] = Qg0; row[16] = C_cover_A;   /* :721 */
                row[17] = EPC0_A_i; row[18] = EPC0_NC_i; row[19] = TDPs0_A; row[20] = Plab0_A;
                row[21] = conc_TDPs_A; row[22] = TDPs0_NC; row[23] = Plab0_NC; row[24] = conc_TDPs_NC; /* :722-723 */
                row[SIMPLYP_OUT_D_SNOW] = D_snow;      /* met_df['D_snow_end'], inputs.py:200, :205 (written only with o->snow) */
                for (int c = 0; c < SIMPLYP_N_OUT; ++c)
                    if (col_of[c] >= 0)
                        out[(((size_t)col_of[c] * D + idx) * n_out_reaches + slot) * E + e] = row[c];
            }
        }
    }
    if (st->capped) stat |= SIMPLYP_STATUS_STEPCAP;
    if (st->poisoned) stat |= SIMPLYP_STATUS_NONFINITE;
    status[e] = stat;
    free(route);
}

/* Same arguments as simplyp_run() minus the context; every pointer is a host pointer.
 * n_threads > 1 runs members in parallel with OpenMP (cpu_baseline 'all cores' leg). */
int simplyp_oracle_run(const simplyp_dims* dims, const simplyp_opts* opts,
                       const double* forcing, const int32_t* doy, const int32_t* forcing_of_member,
                       const double* member_params, const double* reach_params,
                       const int32_t* up_ptr, const int32_t* up_idx,
                       const int32_t* out_reaches, int32_t n_out_reaches,
                       double* out, int32_t* member_status, simplyp_stats* stats, int n_threads)
{
    if (!dims || !opts || dims->E <= 0 || dims->S <= 0 || dims->D <= 0) return SIMPLYP_ERR_ARG;
    const int S = dims->S;
    int32_t* out_slot = (int32_t*)malloc(sizeof(int32_t) * S);
    for (int s = 0; s < S; ++s) out_slot[s] = out_reaches ? -1 : s;
    if (out_reaches) for (int k = 0; k < n_out_reaches; ++k) out_slot[out_reaches[k]] = k;
    else n_out_reaches = S;
    for (int s = 0; s < S; ++s)
        for (int k = up_ptr[s]; k < up_ptr[s + 1]; ++k)
            if (up_idx[k] < 0 || up_idx[k] >= s) { free(out_slot); return SIMPLYP_ERR_TOPOLOGY; }
    uint64_t rhs = 0, steps = 0, rej = 0;
    (void)n_threads;
#ifdef _OPENMP
    if (n_threads > 0) omp_set_num_threads(n_threads);
#pragma omp parallel for schedule(dynamic, 4) reduction(+ : rhs, steps, rej)
#endif
    for (int e = 0; e < dims->E; ++e) {
        integ_stats st = {0, 0, 0, 0, 0};
        run_member(e, dims, opts, forcing, doy, forcing_of_member, member_params, reach_params, up_ptr, up_idx,
                   out_slot, n_out_reaches, opts->integrator, out, member_status, &st);
        rhs += st.rhs; steps += st.steps; rej += st.rejected;
    }
    if (stats) { memset(stats, 0, sizeof(*stats)); stats->rhs_evals = rhs; stats->steps = steps; stats->rejected = rej; }
    free(out_slot);
    return SIMPLYP_OK;
}

/* ---- scalar entry points used to pin the restatement against tests/golden/unit_vectors.npz */
double simplyp_oracle_fx(double x, double threshold, double reld) { return f_x(x, threshold, reld); }

void simplyp_oracle_soilp(const double* in10, double* out2)
{   /* in10 = P_netInput, A_catch, Kf, Msoil, EPC0, Qs, Qq, Vs, TDPs, Plab */
    discretized_soilP(in10[0], in10[1], in10[2], in10[3], in10[4], in10[5], in10[6], in10[7], in10[8], in10[9],
                      &out2[0], &out2[1]);
}

void simplyp_oracle_ode_f(const double* y12, const double* p43, double* dy12)
{   /* p43 in the order of unit_vectors.npz 'ode_p_names' */
    ode_params p;
    p.P = p43[0]; p.E = p43[1]; p.mu = p43[2]; p.Qq_i = p43[3]; p.Qr_US_i = p43[4];
    p.Esus_A = p43[5]; p.Esus_S = p43[6]; p.Esus_IG = p43[7];
    p.Msus_US_i = p43[8]; p.TDPr_US_i = p43[9]; p.PPr_US_i = p43[10];
    p.f_A = p43[11]; p.f_Ar = p43[12]; p.f_IG = p43[13]; p.f_S = p43[14]; p.f_NC_A = p43[15];
    p.f_NC_Ar = p43[16]; p.f_NC_IG = p43[17]; p.f_NC_S = p43[18]; p.NC_type = (int)p43[19];
    p.f_quick = p43[20]; p.alpha = p43[21]; p.beta = p43[22]; p.T_s_A = p43[23]; p.T_s_S = p43[24];
    p.T_g = p43[25]; p.fc = p43[26]; p.L_reach = p43[27]; p.A_catch = p43[28]; p.a_Q = p43[29];
    p.b_Q = p43[30]; /* p43[31] = E_M, unused inside ode_f */ p.k_M = p43[32];
    p.conc_TDPs_A = p43[33]; p.conc_TDPs_NC = p43[34]; p.PlabA_i = p43[35]; p.PlabNC_i = p43[36];
    p.Msoil = p43[37]; p.TDPeff = p43[38]; p.TDPg = p43[39]; p.E_PP = p43[40]; p.P_inactive = p43[41];
    p.Qg_min = p43[42];
    ode_f(y12, &p, dy12);
}
