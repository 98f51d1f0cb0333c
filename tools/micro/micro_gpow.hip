// micro-benchmark: latency of a dependent chain of gpow calls executed by one wave (development aid)
#include <hip/hip_runtime.h>
#include <cstdio>
#include "../../octa_autosegmentation_amd/csrc/gpow.h"
__global__ void chain(double *out, long *ticks, int n, int mode) {
    __shared__ double ltab[384];
    __shared__ uint64_t etab[256];
    for (int i = threadIdx.x; i < 384; i += blockDim.x) ltab[i] = octa_gpow::LOG_TAB[i];
    for (int i = threadIdx.x; i < 256; i += blockDim.x) etab[i] = octa_gpow::EXP_TAB[i];
    __syncthreads();
    double x = 0.0123 + 1e-6 * threadIdx.x * (mode == 2 ? 1 : 0);
    long t0 = wall_clock64();
    for (int i = 0; i < n; i++) {
        double s = (mode == 0) ? octa_gpow::gpow(x, 2.55) : octa_gpow::gpow_t(x, 2.55, ltab, etab);
        s = s + 1e-7;
        x = (mode == 0) ? octa_gpow::gpow(s, 1.0 / 2.55) : octa_gpow::gpow_t(s, 1.0 / 2.55, ltab, etab);
    }
    long t1 = wall_clock64();
    out[threadIdx.x] = x;
    if (threadIdx.x == 0) ticks[0] = t1 - t0;
}
int main() {
    double *d; long *t;
    hipMalloc(&d, 8 * 64); hipMalloc(&t, 8);
    for (int mode = 0; mode < 3; mode++) {
        int n = 20000;
        hipLaunchKernelGGL(chain, dim3(1), dim3(64), 0, 0, d, t, n, mode);
        hipDeviceSynchronize();
        long h; hipMemcpy(&h, t, 8, hipMemcpyDeviceToHost);
        printf("mode %d: %.1f ns per gpow (%d pairs, %.3f ms)\n", mode, h * 10.0 / (2.0 * n), n, h / 1e5);
    }
    return 0;
}
