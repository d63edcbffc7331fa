// fp64 dependent-chain latency vs throughput on one CU-load of waves: NCH independent Horner-like chains per thread
#include <hip/hip_runtime.h>
#include <cstdio>
template <int NCH> __global__ __launch_bounds__(1024) void k(double *out, double x, int n) {
    double a[NCH];
    for (int j = 0; j < NCH; ++j) a[j] = 1.0 + threadIdx.x * 1e-9 + j;
    for (int i = 0; i < n; ++i) {
#pragma unroll
        for (int j = 0; j < NCH; ++j) a[j] = __builtin_fma(a[j], x, 0.5);
    }
    double s = 0; for (int j = 0; j < NCH; ++j) s += a[j];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}
template <int NCH> void run(int threads, const char *what) {
    double *d; hipMalloc(&d, 256 * 1024 * 8);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    const int n = 200000;
    hipLaunchKernelGGL(k<NCH>, dim3(256), dim3(threads), 0, 0, d, 0.999999, 1000);
    hipDeviceSynchronize();
    hipEventRecord(e0); hipLaunchKernelGGL(k<NCH>, dim3(256), dim3(threads), 0, 0, d, 0.999999, n); hipEventRecord(e1);
    hipEventSynchronize(e1); float ms; hipEventElapsedTime(&ms, e0, e1);
    // per SIMD: waves = threads/64/4; instructions per wave = n*NCH
    double cyc = ms * 1e-3 * 2.4e9;
    printf("%s threads/WG %4d (waves/SIMD %d) chains %d: %.3f ms, cycles per FMA per wave %.2f, per SIMD-issue %.2f\n", what, threads, threads / 256, NCH, ms,
           cyc / ((double)n * NCH), cyc / ((double)n * NCH * (threads / 256)));
    hipFree(d);
}
int main() {
    for (int t : {256, 512, 1024}) { run<1>(t, "dep"); run<2>(t, "dep"); run<4>(t, "dep"); }
    return 0;
}
