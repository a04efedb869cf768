// Relative error of v_rcp_f64 raw, with one and with two Newton steps (sp_rcp uses two).
//   hipcc --offload-arch=gfx950 -O3 tools/micro/rcp_accuracy.hip -o tools/micro/rcp_accuracy && tools/micro/rcp_accuracy
#include <hip/hip_runtime.h>
#include <cmath>
#include <cstdio>
#include <random>
#include <vector>
__global__ void k(const double* x, double* r0, double* r1, double* r2, int n)
{
    int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    double a = x[i], r = __builtin_amdgcn_rcp(a);
    r0[i] = r;
    r = __builtin_fma(__builtin_fma(-a, r, 1.0), r, r);
    r1[i] = r;
    r = __builtin_fma(__builtin_fma(-a, r, 1.0), r, r);
    r2[i] = r;
}
int main()
{
    const int n = 1 << 20;
    std::vector<double> x(n), a(n), b(n), c(n);
    std::mt19937_64 g(7); std::uniform_real_distribution<double> u(-3.0, 3.0);
    for (int i = 0; i < n; ++i) x[i] = std::pow(10.0, u(g)) * (1.0 + 1e-3 * u(g));
    double *dx, *d0, *d1, *d2;
    hipMalloc(&dx, n * 8); hipMalloc(&d0, n * 8); hipMalloc(&d1, n * 8); hipMalloc(&d2, n * 8);
    hipMemcpy(dx, x.data(), n * 8, hipMemcpyHostToDevice);
    hipLaunchKernelGGL(k, dim3(n / 256), dim3(256), 0, 0, dx, d0, d1, d2, n);
    hipMemcpy(a.data(), d0, n * 8, hipMemcpyDeviceToHost); hipMemcpy(b.data(), d1, n * 8, hipMemcpyDeviceToHost); hipMemcpy(c.data(), d2, n * 8, hipMemcpyDeviceToHost);
    double e0 = 0, e1 = 0, e2 = 0;
    for (int i = 0; i < n; ++i) {
        const long double t = 1.0L / (long double)x[i];
        e0 = std::fmax(e0, (double)fabsl(((long double)a[i] - t) / t));
        e1 = std::fmax(e1, (double)fabsl(((long double)b[i] - t) / t));
        e2 = std::fmax(e2, (double)fabsl(((long double)c[i] - t) / t));
    }
    printf("max relative error of 1/x: v_rcp_f64 raw %.3e, + 1 Newton step %.3e, + 2 Newton steps %.3e (eps = 1.1e-16)\n", e0, e1, e2);
    return 0;
}
