// microbenchmarks: f64 MFMA issue rate and HBM stream bandwidth on the box
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef double v4d __attribute__((ext_vector_type(4)));
__global__ __launch_bounds__(256) void k_mfma(double* out, int iters, int nacc) {
    v4d acc[8];
    for (int i = 0; i < 8; ++i) acc[i] = (v4d){0, 0, 0, 0};
    double a = threadIdx.x * 1e-3, b = 1.0 + blockIdx.x * 1e-6;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int i = 0; i < 8; ++i) acc[i] = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, acc[i], 0, 0, 0);
    }
    double s = 0;
    for (int i = 0; i < 8; ++i) s += acc[i][0] + acc[i][1] + acc[i][2] + acc[i][3];
    out[blockIdx.x * 256 + threadIdx.x] = s;
}
__global__ __launch_bounds__(256) void k_copy(const double2* __restrict__ x, double2* __restrict__ y, size_t n) {
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) y[i] = x[i];
}
__global__ __launch_bounds__(256) void k_read(const double2* __restrict__ x, double* out, size_t n) {
    double s = 0;
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) { double2 v = x[i]; s += v.x + v.y; }
    if (s == 123.456) out[0] = s;
}
int main() {
    double* out; hipMalloc(&out, 1 << 24);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    for (int waves = 1; waves <= 8; waves *= 2) {
        int blocks = 256 * waves, iters = 100000;
        k_mfma<<<blocks, 256>>>(out, 10, 8);
        hipEventRecord(e0); k_mfma<<<blocks, 256>>>(out, iters, 8); hipEventRecord(e1); hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1);
        double flops = (double)blocks * 4 * iters * 8 * 2048.0;
        printf("mfma f64 16x16x4: %d blocks x 4 waves: %.1f TFLOP/s (%.2f ms)\n", blocks, flops / ms * 1e-9, ms);
    }
    size_t n = (size_t)1 << 28;  // 4 GiB of double2
    double2 *x, *y; hipMalloc(&x, n * 16); hipMalloc(&y, n * 16); hipMemset(x, 0, n * 16); hipMemset(y, 0, n * 16);
    for (int g : {2048, 8192, 32768}) {
        k_copy<<<g, 256>>>(x, y, n);
        hipEventRecord(e0); for (int r = 0; r < 5; ++r) k_copy<<<g, 256>>>(x, y, n); hipEventRecord(e1); hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1);
        printf("copy grid %d: %.0f GB/s (read+write)\n", g, 5.0 * 2 * n * 16 / ms * 1e-6);
        k_read<<<g, 256>>>(x, out, n);
        hipEventRecord(e0); for (int r = 0; r < 5; ++r) k_read<<<g, 256>>>(x, out, n); hipEventRecord(e1); hipEventSynchronize(e1);
        hipEventElapsedTime(&ms, e0, e1);
        printf("read grid %d: %.0f GB/s\n", g, 5.0 * n * 16 / ms * 1e-6);
    }
    return 0;
}
