// micro: latency of a dependent chain of v_fma_f64 on one wave (alone on its SIMD / with a second wave on the same SIMD doing the same).
// hipcc --offload-arch=gfx950 -O3 tools/micro/f64_chain.hip -o /tmp/f64_chain && /tmp/f64_chain
#include <hip/hip_runtime.h>
#include <cstdio>
__global__ void chain(double *out, long *cycles, int n, double a, double b) {
    double x = out[threadIdx.x];
    long t0 = clock64();
#pragma unroll 1
    for (int i = 0; i < n; i++) {
#pragma unroll
        for (int k = 0; k < 16; k++) x = __builtin_fma(x, a, b);
    }
    long t1 = clock64();
    out[threadIdx.x + blockIdx.x * blockDim.x] = x;
    if (threadIdx.x % 64 == 0) cycles[blockIdx.x * (blockDim.x / 64) + threadIdx.x / 64] = t1 - t0;
}
__global__ void chain32(float *out, long *cycles, int n, float a, float b) {
    float x = out[threadIdx.x];
    long t0 = clock64();
#pragma unroll 1
    for (int i = 0; i < n; i++) {
#pragma unroll
        for (int k = 0; k < 16; k++) x = __builtin_fmaf(x, a, b);
    }
    long t1 = clock64();
    out[threadIdx.x + blockIdx.x * blockDim.x] = x;
    if (threadIdx.x % 64 == 0) cycles[blockIdx.x * (blockDim.x / 64) + threadIdx.x / 64] = t1 - t0;
}
int main() {
    double *d; long *c; float *f;
    hipMalloc(&d, 8 * 4096); hipMalloc(&c, 8 * 64); hipMalloc(&f, 4 * 4096);
    hipMemset(d, 0, 8 * 4096); hipMemset(f, 0, 4 * 4096);
    const int n = 4096;
    for (int threads : {64, 256, 512}) {
        hipLaunchKernelGGL(chain, dim3(1), dim3(threads), 0, 0, d, c, n, 0.999, 0.5);
        hipDeviceSynchronize();
        hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
        hipEventRecord(e0); hipLaunchKernelGGL(chain, dim3(1), dim3(threads), 0, 0, d, c, n, 0.999, 0.5); hipEventRecord(e1); hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1);
        long h[8]; hipMemcpy(h, c, 8 * 8, hipMemcpyDeviceToHost);
        printf("f64 %d threads (%d wave(s) per SIMD): %.2f ns per dependent fma (kernel), clock64 ticks per fma %.2f\n", threads, threads > 256 ? 2 : 1, ms * 1e6 / (n * 16.0), (double)h[0] / (n * 16.0));
        hipEventRecord(e0); hipLaunchKernelGGL(chain32, dim3(1), dim3(threads), 0, 0, f, c, n, 0.999f, 0.5f); hipEventRecord(e1); hipEventSynchronize(e1);
        hipEventElapsedTime(&ms, e0, e1);
        printf("f32 %d threads: %.2f ns per dependent fma\n", threads, ms * 1e6 / (n * 16.0));
    }
    return 0;
}
