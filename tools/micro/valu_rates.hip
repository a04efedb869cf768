// Issue cost of the instructions the time-stepping kernel leans on, measured the way that kernel runs: ONE wave per SIMD
// (1024 single-wave workgroups), 16 independent chains, 64 instructions per loop trip.  Prints cycles per wave64 instruction.
//   hipcc --offload-arch=gfx950 -O3 tools/micro/valu_rates.hip -o tools/micro/valu_rates && tools/micro/valu_rates
#include <hip/hip_runtime.h>
#include <cstdio>

#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)

enum { FMA64_DEP1, FMA64_DEP2, FMA64_DEP4, FMA64, MUL64, ADD64, MAX64, MIN64, RCP64, CMPSEL64, CVT6432, FMA32, PKFMA32, RCP32, EXP32, LOG32, FMA64_SMOV2, FMA64_SNOP, DPP32X2, DPP64, DPP64_FMA, CNDMASK2, N_OPS };
typedef float float2v __attribute__((ext_vector_type(2)));

template <int OP>
__global__ __launch_bounds__(64, 1) void k(double* out, long long* cyc, int iters, double seed)
{
    double a[16];
    float f[16];
    float2v g[16];
    for (int i = 0; i < 16; ++i) { a[i] = seed + threadIdx.x * 1e-3 + i * 0.01; f[i] = (float)a[i]; g[i] = float2v{f[i], f[i] * 0.5f}; }
    const double c1 = seed * 0.666, c2 = seed * 1e-9;
    const long long t0 = clock64();
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int r = 0; r < 4; ++r)
#pragma unroll
            for (int i = 0; i < 16; ++i) {
                if (OP == FMA64_DEP1) a[0] = __builtin_fma(a[0], c1, c2);                     // one chain: every FMA waits for the one before
                if (OP == FMA64_DEP2) a[i & 1] = __builtin_fma(a[i & 1], c1, c2);             // two interleaved chains
                if (OP == FMA64_DEP4) a[i & 3] = __builtin_fma(a[i & 3], c1, c2);             // four
                if (OP == FMA64) a[i] = __builtin_fma(a[i], c1, c2);
                if (OP == MUL64) a[i] = a[i] * c1;
                if (OP == ADD64) a[i] = a[i] + c2;
                if (OP == MAX64) a[i] = __builtin_fmax(a[i], a[(i + 1) & 15]);
                if (OP == MIN64) a[i] = __builtin_fmin(a[i], a[(i + 1) & 15]);
                if (OP == RCP64) a[i] = __builtin_amdgcn_rcp(a[i]);
                if (OP == CMPSEL64) a[i] = (a[i] > a[(i + 1) & 15]) ? a[i] : c1;           // v_cmp + 2 v_cndmask
                if (OP == CVT6432) f[i] = (float)a[i];                                       // (dead-code proof below)
                if (OP == FMA32) f[i] = __builtin_fmaf(f[i], 0.999f, 1e-3f);
                if (OP == PKFMA32) asm volatile("v_pk_fma_f32 %0, %0, %1, %2" : "+v"(g[i]) : "v"(float2v{0.999f, 0.998f}), "v"(float2v{1e-3f, 2e-3f}));   // two fp32 FMAs per lane
                if (OP == RCP32) f[i] = __builtin_amdgcn_rcpf(f[i]);
                if (OP == EXP32) f[i] = __builtin_amdgcn_exp2f(f[i]);
                if (OP == LOG32) f[i] = __builtin_amdgcn_logf(f[i]);
                if (OP == FMA64_SMOV2) { a[i] = __builtin_fma(a[i], c1, c2); asm volatile("s_mov_b32 s40, 0x1234\n\ts_mov_b32 s41, 0x5678" ::: "s40", "s41"); }
                if (OP == FMA64_SNOP) { a[i] = __builtin_fma(a[i], c1, c2); asm volatile("s_nop 0"); }
                if (OP == CVT6432) a[i] += 1.0;                                              // keeps the conversions distinct: + v_add_f64
                // a double handed to the other lanes of a quad: two 32-bit DPP moves (quad_perm) ...
                if (OP == DPP32X2) asm volatile("v_mov_b32_dpp %0, %2 quad_perm:[2,2,2,2] row_mask:0xf bank_mask:0xf bound_ctrl:1\n\t"
                                                "v_mov_b32_dpp %1, %3 quad_perm:[2,2,2,2] row_mask:0xf bank_mask:0xf bound_ctrl:1"
                                                : "=v"(f[i]), "=v"(f[(i + 1) & 15]) : "v"(f[(i + 2) & 15]), "v"(f[(i + 3) & 15]));
                // ... or one 64-bit DPP move (gfx90a+: row_newbcast only -- a lane of each row of 16 to the whole row)
                if (OP == DPP64) asm volatile("v_mov_b64_dpp %0, %1 row_newbcast:2 row_mask:0xf bank_mask:0xf bound_ctrl:1" : "=v"(a[i]) : "v"(a[(i + 1) & 15]));
                if (OP == DPP64_FMA) { asm volatile("v_mov_b64_dpp %0, %1 row_newbcast:2 row_mask:0xf bank_mask:0xf bound_ctrl:1" : "=v"(a[i]) : "v"(a[(i + 1) & 15])); a[(i + 5) & 15] = __builtin_fma(a[(i + 5) & 15], c1, a[i]); }
                if (OP == CNDMASK2) a[i] = (threadIdx.x & 1) ? a[i] : a[(i + 1) & 15];          // 2 v_cndmask_b32 on a lane-constant condition
            }
    }
    const long long t1 = clock64();
    double s = 0;
    for (int i = 0; i < 16; ++i) s += a[i] + f[i] + g[i].x + g[i].y;
    out[blockIdx.x * 64 + threadIdx.x] = s;
    if (threadIdx.x == 0) cyc[blockIdx.x] = t1 - t0;
}

template <int OP>
int run1(const char* name, double* out, long long* cyc, int blocks, double* ns_out)
{
    const int iters = 4000;
    hipEvent_t e0, e1;
    CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
    float ms = 0;
    for (int rep = 0; rep < 2; ++rep) {
        CHECK(hipEventRecord(e0));
        hipLaunchKernelGGL(k<OP>, dim3(blocks), dim3(64), 0, 0, out, cyc, iters, 1.5);
        CHECK(hipEventRecord(e1)); CHECK(hipEventSynchronize(e1));
        CHECK(hipEventElapsedTime(&ms, e0, e1));
    }
    *ns_out = ms * 1e6 / iters / 64;
    return 0;
}

// one wave per SIMD (1024 single-wave workgroups: the time-stepping kernel's regime) and two (2048): ns per instruction of ONE
// wave's stream; with two waves sharing the SIMD the same figure means twice the throughput
template <int OP>
int run(const char* name, double* out, long long* cyc, double)
{
    double a = 0, b = 0;
    if (run1<OP>(name, out, cyc, 1024, &a) || run1<OP>(name, out, cyc, 2048, &b)) return 1;
    printf("%-34s %6.2f ns per instruction alone on the SIMD | %6.2f ns with a second wave (%.2fx the throughput)\n", name, a, b, 2.0 * a / b);
    return 0;
}

int main()
{
    double* out; long long* cyc;
    CHECK(hipMalloc(&out, 2048 * 64 * sizeof(double)));
    CHECK(hipMalloc(&cyc, 2048 * sizeof(long long)));
    int dev = 0; hipDeviceProp_t p; CHECK(hipGetDeviceProperties(&p, dev));
    printf("%s, %d CUs, clock %.0f MHz (1 cycle = %.3f ns at that clock)\n", p.name, p.multiProcessorCount, p.clockRate / 1e3, 1e6 / p.clockRate);
    run<FMA64_DEP1>("v_fma_f64, 1 dependent chain", out, cyc, 0);
    run<FMA64_DEP2>("v_fma_f64, 2 interleaved chains", out, cyc, 0);
    run<FMA64_DEP4>("v_fma_f64, 4 interleaved chains", out, cyc, 0);
    run<FMA64>("v_fma_f64", out, cyc, 0);
    run<MUL64>("v_mul_f64", out, cyc, 0);
    run<ADD64>("v_add_f64", out, cyc, 0);
    run<MAX64>("v_max_f64", out, cyc, 0);
    run<MIN64>("v_min_f64", out, cyc, 0);
    run<RCP64>("v_rcp_f64", out, cyc, 0);
    run<CMPSEL64>("v_cmp_gt_f64 + 2 v_cndmask_b32", out, cyc, 0);
    run<CVT6432>("v_cvt_f32_f64 + v_add_f64", out, cyc, 0);
    run<FMA32>("v_fma_f32", out, cyc, 0);
    run<PKFMA32>("v_pk_fma_f32 (2 FMAs per lane)", out, cyc, 0);
    run<RCP32>("v_rcp_f32", out, cyc, 0);
    run<EXP32>("v_exp_f32", out, cyc, 0);
    run<LOG32>("v_log_f32", out, cyc, 0);
    run<FMA64_SMOV2>("v_fma_f64 + 2 s_mov_b32", out, cyc, 0);
    run<FMA64_SNOP>("v_fma_f64 + s_nop 0", out, cyc, 0);
    run<DPP32X2>("2 v_mov_b32_dpp quad_perm (a double)", out, cyc, 0);
    run<DPP64>("v_mov_b64_dpp row_newbcast (a double)", out, cyc, 0);
    run<DPP64_FMA>("v_mov_b64_dpp row_newbcast + v_fma_f64", out, cyc, 0);
    run<CNDMASK2>("2 v_cndmask_b32 (select a double)", out, cyc, 0);
    return 0;
}
