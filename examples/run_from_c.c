/* run_from_c.c -- the C ABI (include/simplyp.h) driven from plain C, no Python, no torch: device buffers from
 * simplyp_device_alloc, pinned staging from simplyp_host_alloc, simplyp_stream_out (the output table arrives in pinned host
 * memory while the kernel runs, as the reference leaves its results in host memory), simplyp_run_async + simplyp_sync.
 *
 * One sub-catchment with the Tarland workbook's parameters, synthetic forcing, E members that
 * differ in T_g.  Prints the outlet's mean daily flow per member and the solver statistics.
 *
 *   gcc -O2 -Iinclude examples/run_from_c.c -o run_from_c -Lsimplyp_amd/csrc -lsimplyp_hip -Wl,-rpath,$PWD/simplyp_amd/csrc -lm
 *   ./run_from_c [E] [D]
 */
#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include "simplyp.h"

#define CHECK(call)                                                                                       \
    do {                                                                                                  \
        int rc__ = (call);                                                                                \
        if (rc__ != SIMPLYP_OK) {                                                                         \
            fprintf(stderr, "%s failed (%d): %s\n", #call, rc__, simplyp_last_error(ctx));                \
            return 1;                                                                                     \
        }                                                                                                 \
    } while (0)

int main(int argc, char** argv)
{
    const int E = argc > 1 ? atoi(argv[1]) : 128, D = argc > 2 ? atoi(argv[2]) : 730, S = 1;
    simplyp_ctx* ctx = NULL;
    if (simplyp_abi_version() != SIMPLYP_ABI_VERSION) { fprintf(stderr, "ABI mismatch\n"); return 1; }
    if (simplyp_device_count() < 1) { fprintf(stderr, "no HIP device\n"); return 2; }
    CHECK(simplyp_ctx_create(0, &ctx));

    /* host side: pinned staging buffers */
    const size_t n_f = (size_t)2 * D, n_mp = (size_t)SIMPLYP_NP_M * E, n_rp = (size_t)SIMPLYP_NP_R * S * E;
    double* forcing = (double*)simplyp_host_alloc((int64_t)(n_f * sizeof(double)));
    int32_t* doy = (int32_t*)simplyp_host_alloc((int64_t)((size_t)D * sizeof(int32_t)));
    double* mp = (double*)simplyp_host_alloc((int64_t)(n_mp * sizeof(double)));
    double* rp = (double*)simplyp_host_alloc((int64_t)(n_rp * sizeof(double)));
    if (!forcing || !doy || !mp || !rp) { fprintf(stderr, "host alloc failed\n"); return 1; }
    for (int d = 0; d < D; ++d) {
        const double season = 0.5 - 0.5 * cos(2.0 * M_PI * d / 365.25);
        forcing[d] = (d % 5 == 0) ? 9.0 + 6.0 * sin(0.37 * d) : ((d % 3 == 0) ? 1.5 : 0.0);      /* P, mm/day   */
        forcing[D + d] = 0.2 + 2.8 * season;                                                    /* PET, mm/day */
        doy[d] = d % 365 + 1;
    }
    static const double pm[SIMPLYP_NP_M] = {
        /* f_quick alpha fc beta T_g Qg_min a_Q b_Q Qr0_init Msoil_m2 Kf */ 0.02, 1, 290, 0.7, 65, 0.4, 0.5, 0.42, 1, 95, 1.131528046e-4,
        /* TDPg E_PP E_M k_M d_maxE_spr d_maxE_aut */ 0.02, 1.6, 1500, 2, 60, 304,
        /* T_s A,S  SoilPconc A,S  P_netInput A,NC  EPC0_init A,S */ 2, 10, 1458, 873, 10, 10, 0.1, 0,
        /* C_cover A,S,IG  C_measures A,S,IG */ 0.2, 0.021, 0.09, 0, 0, 0,
        /* f_DDSM D_snow_0 */ 2.74, 0};
    static const double pr[SIMPLYP_NP_R] = {/* A_catch f_Ar f_IG f_S f_NC_Ar f_NC_IG f_NC_S f_spr */ 51.7, 0.2, 0.3, 0.5, 0, 0, 0, 0.65,
                                            /* S_Ar S_IG S_SN L_reach S_reach TDPeff */ 4, 4, 10, 10000, 0.8, 0.1};
    for (int i = 0; i < SIMPLYP_NP_M; ++i) for (int e = 0; e < E; ++e) mp[(size_t)i * E + e] = pm[i];
    for (int i = 0; i < SIMPLYP_NP_R; ++i) for (int e = 0; e < E; ++e) rp[(size_t)i * E + e] = pr[i];
    for (int e = 0; e < E; ++e) mp[(size_t)SIMPLYP_PM_T_G * E + e] = 40.0 + 60.0 * e / (E > 1 ? E - 1 : 1);

    simplyp_dims dims = {E, S, D, 1};
    simplyp_opts opts;
    memset(&opts, 0, sizeof(opts));
    opts.integrator = SIMPLYP_INTEG_CASHKARP_AUG; opts.substeps = 8; opts.rtol = 1e-7; opts.atol = 1e-12; opts.max_steps = 4000;
    opts.dynamic_epc0 = 1; opts.run_mode_cal = 1; opts.out_mask = SIMPLYP_MASK_REACH5; opts.step_len = 1.0; opts.project_vr = 1;
    opts.balance = 2;
    const int32_t up_ptr[2] = {0, 0};

    /* device side */
    const int64_t out_bytes = simplyp_out_bytes(&dims, &opts, 1);
    double* d_forcing = (double*)simplyp_device_alloc(ctx, (int64_t)(n_f * sizeof(double)));
    int32_t* d_doy = (int32_t*)simplyp_device_alloc(ctx, (int64_t)((size_t)D * sizeof(int32_t)));
    double* d_mp = (double*)simplyp_device_alloc(ctx, (int64_t)(n_mp * sizeof(double)));
    double* d_rp = (double*)simplyp_device_alloc(ctx, (int64_t)(n_rp * sizeof(double)));
    double* d_out = (double*)simplyp_device_alloc(ctx, out_bytes);
    int32_t* d_status = (int32_t*)simplyp_device_alloc(ctx, (int64_t)((size_t)E * sizeof(int32_t)));
    if (!d_forcing || !d_doy || !d_mp || !d_rp || !d_out || !d_status) { fprintf(stderr, "device alloc failed: %s\n", simplyp_last_error(ctx)); return 1; }
    CHECK(simplyp_memcpy_h2d(ctx, d_forcing, forcing, (int64_t)(n_f * sizeof(double))));
    CHECK(simplyp_memcpy_h2d(ctx, d_doy, doy, (int64_t)((size_t)D * sizeof(int32_t))));
    CHECK(simplyp_memcpy_h2d(ctx, d_mp, mp, (int64_t)(n_mp * sizeof(double))));
    CHECK(simplyp_memcpy_h2d(ctx, d_rp, rp, (int64_t)(n_rp * sizeof(double))));

    /* the table is delivered to `out` (pinned host memory) chunk by chunk while later chunks compute */
    double* out = (double*)simplyp_host_alloc(out_bytes);
    double* check = (double*)simplyp_host_alloc(out_bytes);
    int32_t* status = (int32_t*)simplyp_host_alloc((int64_t)((size_t)E * sizeof(int32_t)));
    if (!out || !check || !status) { fprintf(stderr, "host alloc failed\n"); return 1; }
    simplyp_stats stats;
    CHECK(simplyp_stream_out(ctx, out, out_bytes));
    CHECK(simplyp_run_async(ctx, &dims, &opts, d_forcing, d_doy, NULL, NULL, d_mp, d_rp, up_ptr, NULL, NULL, 0, d_out, d_status, NULL, NULL));
    CHECK(simplyp_sync(ctx, &stats));                 /* returns when the last byte has arrived */
    CHECK(simplyp_memcpy_d2h(ctx, check, d_out, out_bytes));
    CHECK(simplyp_memcpy_d2h(ctx, status, d_status, (int64_t)((size_t)E * sizeof(int32_t))));
    if (memcmp(out, check, (size_t)out_bytes) != 0) { fprintf(stderr, "streamed table differs from the device table\n"); return 4; }

    /* REACH-5 columns in ascending SIMPLYP_OUT_* order: Vr, Qr, Msus_kg/day, TDP_kg/day, PP_kg/day; out[c][d][0][e] */
    int flagged = 0;
    for (int e = 0; e < E; ++e) flagged += status[e] != 0;
    const int show[3] = {0, E / 2, E - 1};
    for (int k = 0; k < 3; ++k) {
        const int e = show[k];
        double q = 0.0, tdp = 0.0;
        for (int d = 0; d < D; ++d) { q += out[((size_t)1 * D + d) * E + e]; tdp += out[((size_t)3 * D + d) * E + e]; }
        printf("member %d: T_g %.1f d  mean Qr %.6f mm/d  TDP flux %.6f kg\n", e, mp[(size_t)SIMPLYP_PM_T_G * E + e], q / D, tdp);
    }
    printf("E=%d D=%d flagged=%d rhs_evals=%llu (%.1f per catchment-day) kernel %.3f ms launches %d\n", E, D, flagged,
           (unsigned long long)stats.rhs_evals, (double)stats.rhs_evals / ((double)E * D), stats.kernel_ms, stats.n_launches);
    printf("streamed: %d chunk(s) copied beside the kernel, %.3f ms of copy after the last launch, wall %.3f ms\n",
           stats.streamed_chunks, stats.d2h_tail_ms, stats.wall_ms);

    simplyp_device_free(ctx, d_forcing); simplyp_device_free(ctx, d_doy); simplyp_device_free(ctx, d_mp);
    simplyp_device_free(ctx, d_rp); simplyp_device_free(ctx, d_out); simplyp_device_free(ctx, d_status);
    simplyp_host_free(forcing); simplyp_host_free(doy); simplyp_host_free(mp); simplyp_host_free(rp);
    simplyp_host_free(out); simplyp_host_free(check); simplyp_host_free(status);
    simplyp_ctx_destroy(ctx);
    return flagged ? 3 : 0;
}
