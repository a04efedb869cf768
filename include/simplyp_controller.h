/*
 * simplyp_controller.h -- the constants of the knee-aware step controller of integrator 2
 * (SIMPLYP_INTEG_CASHKARP_AUG), in ONE place.
 *
 * Plain C macros, no code: included by the device kernels (simplyp_amd/csrc/simplyp_kernels.hip.h: SysAug::*, read by
 * ck_day<SysAug> and ck_day_quad) and by the CPU oracle (oracle/simplyp_oracle.c: cashkarp_aug_day), so that the copies of
 * the controller cannot drift apart.  Not part of the C ABI: nothing here is visible to a caller of libsimplyp_hip.so.
 *
 * What they steer: the reference's gate f_x (model.py:23-37) is a smooth step, C1 only -- its second
 * derivative jumps at both ends ("knees") of its 1 %-wide zone.  Three gates sit in the right-hand side: soil box A and S at
 * fc ... 1.01 fc (model.py:105, :109), groundwater at Qg_min ... 1.01 Qg_min (:121).  A 5(4) pair that steps across a knee
 * drops to third order and its embedded estimate no longer bounds the error.
 */
#ifndef SIMPLYP_CONTROLLER_H
#define SIMPLYP_CONTROLLER_H

/* pb = Qr**b_Q, pk = Qr**k_M are only neutrally stable about their exact values and are not in the error norm: re-evaluate
 * them after every RESYNC-th attempt of the day (a storm day can take 100+) */
#define SIMPLYP_CTRL_RESYNC_EVERY 8
/* error norm over the 7 physical states VsA VsS Vg Qr Msus TDPr PPr (the auxiliary states are functions of them, the four
 * daily integrals quadratures of them) ... */
#define SIMPLYP_CTRL_N_ERR 7
/* ... plus Qr**k_M at AUX_WEIGHT x the tolerance (on a day when a nearly dry reach is wetted it grows 200-fold and its own
 * truncation error showed in the sediment flux) */
#define SIMPLYP_CTRL_AUX_WEIGHT 3.0
/* A step that, along its first slope, has a knee within KINK_REACH x its length without having been aimed at it has its error
 * estimate multiplied by KINK_SOIL (soil-water gates, crossed on most wet days) or KINK_GW (groundwater gate: zone 1 % of
 * Qg_min wide, crossed a few times a year by members with a low Qg_min): it is accepted only if it is short. */
#define SIMPLYP_CTRL_KINK_SOIL 10.0
#define SIMPLYP_CTRL_KINK_GW 100.0
#define SIMPLYP_CTRL_KINK_REACH 1.25
/* Aim at the knee: a knee between KNEE_LO and KNEE_HI of the step (along the first slope) ends the step KNEE_OVER x that far,
 * i.e. just past it; no inflation for such a step, except that one aimed at a groundwater knee keeps a factor KNEE_GW.
 * Knees within the first KNEE_LO of a step are left alone. */
#define SIMPLYP_CTRL_KNEE_LO 0.02
#define SIMPLYP_CTRL_KNEE_HI 0.9
#define SIMPLYP_CTRL_KNEE_OVER 1.05
#define SIMPLYP_CTRL_KNEE_GW 10.0
/* Expansive reach.  The flow equation dQr/dt = (I - Qr) cQ Qr**b_Q (model.py:127-130) damps an error in Qr as long as the reach is
 * near its quasi-steady state Qr ~ I -- the normal case, and the reason why local errors of 1e-7 give a global error of a few 1e-7.
 * Its derivative with respect to Qr is cQ Qr**(b_Q - 1) (b_Q (I - Qr) - Qr): POSITIVE when b_Q x (net inflow) > Qr, i.e. when a nearly
 * dry reach is wetted (flow a fraction of its input): there the equation amplifies what earlier steps of the day left behind, and the
 * daily outputs came out 10-80 x above the tolerance (found in round 3 on a drier climate than Tarland's: 31 of 100 000 members above
 * 1e-6, worst 8.6e-6; on the Tarland ensemble it was the worst member's mechanism too).  A step that STARTS in that regime has its error
 * estimate multiplied by EXPAND: accepted only if ~2 x shorter.  The regime is rare: 0.05 % more right-hand sides on the Tarland
 * ensemble (worst member of 100 000: 4.0e-7 -> 3.1e-7 with 10, the same with 30), 13 % more on the 256 driest members of a climate
 * with 0.6 x the precipitation and 1.67 x the PET (worst of 100 000: 8.6e-6 -> 5.2e-7 with 10, 4.2e-7 with 30; at 0.4 x / 2.5 x:
 * 8.0e-7 with 10, 5.1e-7 with 30, the same with 100) -- profiles/r03_experiments.md, profiles/r03_tolerance. */
#define SIMPLYP_CTRL_EXPAND 30.0
/* a day starts with this share of the step size carried over midnight (that size belongs to the smooth end of the previous
 * day; the forcing jumps at midnight) */
#define SIMPLYP_CTRL_DAY_START 0.2
/* step-size rule shared by all adaptive schemes: h <- h * clamp(SAFETY * err^(-1/5), FAC_MIN, FAC_MAX) */
#define SIMPLYP_CTRL_SAFETY 0.9
#define SIMPLYP_CTRL_FAC_MIN 0.2
#define SIMPLYP_CTRL_FAC_MAX 5.0


/* ---- the stability-optimised second pair of integrator 2 (opts.stiff_pair) -----------------------------------
 * Far down a reach network the flow relaxes at rate = cQ Qr**b_Q of up to several hundred per day; once the transient that follows
 * midnight has died away, Cash-Karp's steps there are bound by its real stability interval (|h x rate| <= 3.73), not by accuracy:
 * half of config C4's attempts (tools/probe_c4_steps.py).  For an attempt whose step is longer than Z_ON relaxation times -- and has no
 * knee of a gate within reach -- a lane switches to this pair: 6 stages, order 4 with an embedded order 3, |R| <= 0.97 and
 * |R_hat| <= 1 on [-9.01, 0], every internal stage polynomial <= 1.5 there, Cash-Karp's sparsity (b2 = b5 = 0, e2 = 0), so the attempt
 * loop runs the same instructions with other constants.  Steps are capped at CAP relaxation times.  Derivation (least squares from
 * random starts, then a projection onto the 12 order conditions): tools/derive_stiff_pair.py --beta 9 --pmax 1.5 --seed 34 --trial 53.
 * E_i = b_i - bhat_i.  The step-size factor of such an attempt is SAFETY * err^ERR_EXP (a third-order estimate). */
#define SIMPLYP_STIFF_Z_ON 3.4
#define SIMPLYP_STIFF_CAP 8.1
/* ... and a day starts with a step of at most Z_START relaxation times: the step carried over midnight (x DAY_START) belongs to the
 * capped, quasi-steady end of the previous day, while the new day opens with the reach's transient, which a 5th-order pair follows
 * with steps of 0.1 ... 0.3 relaxation times -- without this the first one or two attempts of a stiff reach's day were rejected
 * (rejections 5.2 -> 1.3 % of the attempts, attempts -3.6 %, on config C4's chain). */
#define SIMPLYP_STIFF_Z_START 0.3
#define SIMPLYP_STIFF_ERR_EXP (-0.25)
/* Damping-aware error weights (same switch, opts.stiff_pair: the network scheme).  The step controller bounds LOCAL errors; what the
 * bar is about are the day's outputs.  A local error d of the reach's flow -- or of one of its three masses, or of Qr**k_M -- does not
 * persist like one of a store without a restoring term: the flow equation forgets it at the rate
 *     lam = -d(dQr/dt)/dQr = rate (1 - b_Q (net inflow) / Qr),      rate = cQ Qr**b_Q,
 * so it reaches the end-of-day state as d exp(-lam (T - t)) and a daily mean or flux as d / (lam T) of a day's worth, against d and
 * d (T - t) / T for a component that keeps it.  Far down a network lam T is 100 ... 450: the controller was resolving the transient that
 * follows midnight ~10 x more finely than the outputs can tell.  The estimate of those five components is therefore divided by
 *     F = clamp(min(lam T / DAMP_PHI, 1 + lam x (what is left of the day after this step)), 1, DAMP_FMAX)
 * (1 + x <= exp(x): the second term is the end-of-day state's bound; lam from the attempt's first stage, at most rate; F = 1 in the
 * expansive regime, where lam < 0).  A reach that relaxes 15 times a day or less -- every single-reach problem of the benchmarks -- has
 * F = 1.  On config C4's chain (oracle, 4 members x 256 reaches x 200 days): attempts 31.0 -> 26.6 per reach-day, worst error against
 * the converged solution 1.1e-7 -> 2.1e-7; against the reference's tables: stiff 12-reach chain 1.5e-7 -> 2.1e-7, C4's members 1.5e-7
 * -> 1.4e-7 (profiles/r04_experiments.md section 7; PHI 30: 28.2 attempts, PHI 10: 25.8 and 2.8e-7 on the stiff chain). */
#define SIMPLYP_DAMP_PHI 15.0
#define SIMPLYP_DAMP_FMAX 16.0
#define SIMPLYP_STIFF_A21 0.12853527643260251
#define SIMPLYP_STIFF_A31 0.16351994308561854
#define SIMPLYP_STIFF_A32 0.19797306142138452
#define SIMPLYP_STIFF_A41 0.049733203403888065
#define SIMPLYP_STIFF_A42 -0.031561611797594993
#define SIMPLYP_STIFF_A43 0.41110922959004331
#define SIMPLYP_STIFF_A51 -0.24502075792174899
#define SIMPLYP_STIFF_A52 0.34132654630326525
#define SIMPLYP_STIFF_A53 0.29025955186572305
#define SIMPLYP_STIFF_A54 0.43141124249147206
#define SIMPLYP_STIFF_A61 -0.035331946161144254
#define SIMPLYP_STIFF_A62 0.29197800440542732
#define SIMPLYP_STIFF_A63 0.058008315542356415
#define SIMPLYP_STIFF_A64 0.35949958123859993
#define SIMPLYP_STIFF_A65 0.1960184892363403
#define SIMPLYP_STIFF_B1 0.12617507659604874
#define SIMPLYP_STIFF_B3 0.29065812274966046
#define SIMPLYP_STIFF_B4 0.25522498620491596
#define SIMPLYP_STIFF_B6 0.32794181444937487
#define SIMPLYP_STIFF_E1 -0.0080682924634790099
#define SIMPLYP_STIFF_E3 0.011178207329606782
#define SIMPLYP_STIFF_E4 0.018364840372072311
#define SIMPLYP_STIFF_E5 -0.12955476832078688
#define SIMPLYP_STIFF_E6 0.10808001308258683
/* opts.stiff_pair: 0 = auto (on for a reach network, S > 1: a single reach is never far from its headwater), > 0 on, < 0 off;
 * integrator 2 only */
#define SIMPLYP_STIFF_PAIR_ON(opt, S) ((opt) > 0 || ((opt) == 0 && (S) > 1))

#endif /* SIMPLYP_CONTROLLER_H */
