// loss.hip -- DiceBCELoss forward / backward in one pass each ("fuse into one reduction kernel").
//
// Replaces utils/losses.py:111-121 (DiceBCELoss = (monai DiceLoss(sigmoid=True) + BCEWithLogitsLoss) / 2) in the training
// step (models/base_model_abc.py:152-167). torch evaluates it as a dozen elementwise and reduction kernels over the
// 1216 x 1216 logits; here the forward is ONE read of logits and labels producing four sums per sample
// (sum p*y, sum p, sum y, sum bce with p = sigmoid(x)), the backward ONE read + one write:
//   d/dx_i = g/2 * [ -(2 y_i D - N) / (B D^2) * p_i (1 - p_i) + (p_i - y_i) / (B n) ],  N = 2 S_py + eps, D = S_p + S_y + eps.
// HBM-bound streaming kernels; sums in double through atomics (order-dependent in the last bits, like torch's own).

#include "common.h"

namespace {

template <class T> __device__ __forceinline__ float ldl(const T *p);
template <> __device__ __forceinline__ float ldl<float>(const float *p) { return *p; }
template <> __device__ __forceinline__ float ldl<unsigned short>(const unsigned short *p) { return __uint_as_float((unsigned)*p << 16); }
template <class T> __device__ __forceinline__ void stl(T *p, float v);
template <> __device__ __forceinline__ void stl<float>(float *p, float v) { *p = v; }
template <> __device__ __forceinline__ void stl<unsigned short>(unsigned short *p, float v) {
    unsigned u = __float_as_uint(v);
    if ((u & 0x7fffffffu) > 0x7f800000u) { *p = (unsigned short)((u >> 16) | 0x40); return; }
    u += 0x7fffu + ((u >> 16) & 1u);
    *p = (unsigned short)(u >> 16);
}

template <class T>
__global__ void __launch_bounds__(256)
dice_bce_fwd_kernel(const T *__restrict__ x, const float *__restrict__ y, long n, double *__restrict__ sums) {
    __shared__ double sh[4 * 4];
    const int b = blockIdx.y;
    const T *px = x + (size_t)b * n;
    const float *py = y + (size_t)b * n;
    float s_py = 0.f, s_p = 0.f, s_y = 0.f, s_b = 0.f;
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long)gridDim.x * 256) {
        const float v = ldl(px + i), t = py[i];
        const float e = __expf(-fabsf(v));                 // exp(-|x|)
        const float p = v >= 0.f ? 1.f / (1.f + e) : e / (1.f + e);
        s_py += p * t; s_p += p; s_y += t;
        s_b += fmaxf(v, 0.f) - v * t + log1pf(e);          // BCEWithLogits, the stable form torch uses
    }
    double a[4] = {s_py, s_p, s_y, s_b};
#pragma unroll
    for (int k = 0; k < 4; k++)
#pragma unroll
        for (int d = 32; d > 0; d >>= 1) a[k] += __shfl_xor(a[k], d, 64);
    const int wv = threadIdx.x >> 6;
    if ((threadIdx.x & 63) == 0)
        for (int k = 0; k < 4; k++) sh[wv * 4 + k] = a[k];
    __syncthreads();
    if (threadIdx.x < 4) atomicAdd(&sums[b * 4 + threadIdx.x], sh[threadIdx.x] + sh[4 + threadIdx.x] + sh[8 + threadIdx.x] + sh[12 + threadIdx.x]);
}

template <class T>
__global__ void __launch_bounds__(256)
dice_bce_bwd_kernel(const T *__restrict__ x, const float *__restrict__ y, long n, int B, const double *__restrict__ sums,
                    const float *__restrict__ gout, float eps_nr, float eps_dr, T *__restrict__ dx) {
    const int b = blockIdx.y;
    const double N = 2.0 * sums[b * 4] + eps_nr, D = sums[b * 4 + 1] + sums[b * 4 + 2] + eps_dr;
    const float g = 0.5f * gout[0];
    const float kd = (float)(1.0 / ((double)B * D * D)), fN = (float)N, fD = (float)D, kb = (float)(1.0 / ((double)B * (double)n));
    const T *px = x + (size_t)b * n;
    const float *py = y + (size_t)b * n;
    T *pd = dx + (size_t)b * n;
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long)gridDim.x * 256) {
        const float v = ldl(px + i), t = py[i];
        const float e = __expf(-fabsf(v));
        const float p = v >= 0.f ? 1.f / (1.f + e) : e / (1.f + e);
        const float ddice = -(2.f * t * fD - fN) * kd * p * (1.f - p);
        stl(pd + i, g * (ddice + (p - t) * kb));
    }
}

}  // namespace

extern "C" int octa_dice_bce_fwd(octa_ctx *ctx, const void *d_logits, int dtype, const float *d_y, int B, int64_t n, double *d_sums, void *stream_) {
    if (!ctx || !d_logits || !d_y || !d_sums || B <= 0 || n <= 0 || B > 65535 || (dtype != 0 && dtype != 1)) { octa::set_error("octa_dice_bce_fwd: bad arguments"); return -2; }
    hipStream_t stream = (hipStream_t)stream_;
    OCTA_HIP_CHECK(hipSetDevice(ctx->device));
    OCTA_HIP_CHECK(hipMemsetAsync(d_sums, 0, sizeof(double) * 4 * B, stream));
    long gx = (n + 256 * 8 - 1) / (256 * 8);
    const long cap = (4L * ctx->num_cus + B - 1) / B;
    if (gx > cap) gx = cap;
    if (gx < 1) gx = 1;
    dim3 grid((unsigned)gx, (unsigned)B);
    if (dtype == 0) hipLaunchKernelGGL(dice_bce_fwd_kernel<float>, grid, dim3(256), 0, stream, static_cast<const float *>(d_logits), d_y, (long)n, d_sums);
    else hipLaunchKernelGGL(dice_bce_fwd_kernel<unsigned short>, grid, dim3(256), 0, stream, static_cast<const unsigned short *>(d_logits), d_y, (long)n, d_sums);
    OCTA_HIP_CHECK(hipGetLastError());
    return 0;
}

// loss = (mean_b(1 - (2 s0 + nr) / (s1 + s2 + dr)) + sum_b s3 / (B n)) / 2 from the sums of octa_dice_bce_fwd, in double like the torch
// expression it replaces (ten scalar launches between the forward pass and the backward pass of every step), rounded to float once
namespace {
__global__ void dice_bce_finish_kernel(const double *__restrict__ sums, int B, double n, double nr, double dr, float *__restrict__ out) {
    if (threadIdx.x != 0 || blockIdx.x != 0) return;
    double dice = 0, bce = 0;
    for (int b = 0; b < B; b++) {
        dice += 1.0 - (2.0 * sums[4 * b] + nr) / (sums[4 * b + 1] + sums[4 * b + 2] + dr);
        bce += sums[4 * b + 3];
    }
    out[0] = (float)((dice / (double)B + bce / ((double)B * n)) / 2.0);
}
}  // namespace

extern "C" int octa_dice_bce_finish(octa_ctx *ctx, const double *d_sums, int B, int64_t n, double smooth_nr, double smooth_dr, float *d_loss, void *stream_) {
    if (!ctx || !d_sums || !d_loss || B <= 0 || n <= 0) { octa::set_error("octa_dice_bce_finish: bad arguments"); return -2; }
    hipStream_t stream = (hipStream_t)stream_;
    OCTA_HIP_CHECK(hipSetDevice(ctx->device));
    hipLaunchKernelGGL(dice_bce_finish_kernel, dim3(1), dim3(64), 0, stream, d_sums, B, (double)n, smooth_nr, smooth_dr, d_loss);
    OCTA_HIP_CHECK(hipGetLastError());
    return 0;
}

extern "C" int octa_dice_bce_bwd(octa_ctx *ctx, const void *d_logits, int dtype, const float *d_y, int B, int64_t n, const double *d_sums,
                                 const float *d_grad_out, float smooth_nr, float smooth_dr, void *d_dlogits, void *stream_) {
    if (!ctx || !d_logits || !d_y || !d_sums || !d_grad_out || !d_dlogits || B <= 0 || n <= 0 || B > 65535 || (dtype != 0 && dtype != 1)) { octa::set_error("octa_dice_bce_bwd: bad arguments"); return -2; }
    hipStream_t stream = (hipStream_t)stream_;
    OCTA_HIP_CHECK(hipSetDevice(ctx->device));
    long gx = (n + 256 * 8 - 1) / (256 * 8);
    const long cap = (8L * ctx->num_cus + B - 1) / B;
    if (gx > cap) gx = cap;
    if (gx < 1) gx = 1;
    dim3 grid((unsigned)gx, (unsigned)B);
    if (dtype == 0) hipLaunchKernelGGL(dice_bce_bwd_kernel<float>, grid, dim3(256), 0, stream, static_cast<const float *>(d_logits), d_y, (long)n, B, d_sums, d_grad_out, smooth_nr, smooth_dr, static_cast<float *>(d_dlogits));
    else hipLaunchKernelGGL(dice_bce_bwd_kernel<unsigned short>, grid, dim3(256), 0, stream, static_cast<const unsigned short *>(d_logits), d_y, (long)n, B, d_sums, d_grad_out, smooth_nr, smooth_dr, static_cast<unsigned short *>(d_dlogits));
    OCTA_HIP_CHECK(hipGetLastError());
    return 0;
}
