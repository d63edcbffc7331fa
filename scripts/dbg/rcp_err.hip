// accuracy of the bare v_rcp_f64 (and of one / two Newton steps on it): max relative error over a sweep of doubles
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cmath>
__global__ void k(int n, double *e0, double *e1, double *e2) {
    int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    double x = 1.0 + (double)i / (double)n;            // [1, 2)
    x *= (i & 1) ? 3.7e5 : 1.3e-3;
    double y = __builtin_amdgcn_rcp(x);
    double r = 1.0 / x;
    e0[i] = fabs(y - r) / r;
    double e = __builtin_fma(-x, y, 1.0); y = __builtin_fma(y, e, y);
    e1[i] = fabs(y - r) / r;
    e = __builtin_fma(-x, y, 1.0); y = __builtin_fma(y, e, y);
    e2[i] = fabs(y - r) / r;
}
int main() {
    const int n = 1 << 22;
    double *d[3], *h = new double[n];
    for (auto &p : d) hipMalloc(&p, n * sizeof(double));
    hipLaunchKernelGGL(k, dim3(n / 256), dim3(256), 0, 0, n, d[0], d[1], d[2]);
    for (int j = 0; j < 3; ++j) {
        hipMemcpy(h, d[j], n * sizeof(double), hipMemcpyDeviceToHost);
        double m = 0; for (int i = 0; i < n; ++i) m = fmax(m, h[i]);
        printf("newton steps %d: max relative error %.3e (2^%.1f)\n", j, m, m > 0 ? log2(m) : -99.0);
    }
    return 0;
}
