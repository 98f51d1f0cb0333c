// blur.hip -- the anti-aliased resampling layers and the reflection pad of the GAN networks (SURVEY.md 8b N8 /
// a19 / a20: "depth-wise 3x3/4x4 blur (Downsample/Upsample)").
//
// Reference: models/networks.py:244-262 (Upsample: replicate pad 1, depth-wise transposed [1 3 3 1]^2/64 * 4 filter,
// stride 2, crop), :264-289 (Downsample: reflect pad 1, depth-wise [1 2 1]^2/16 filter, stride 2), and the
// nn.ReflectionPad2d in front of the 3x3 / 7x7 convolutions (:366-368, :404-421). The reference runs each as a pad
// kernel, a grouped (transposed) convolution and slices; here every layer is ONE streaming kernel per direction that
// folds pad, filter, stride and crop into its tap addresses. HBM-bound: algorithmic bytes = one read of the larger
// tensor + one write of the smaller one (the 9 / 4 taps of neighbouring outputs meet in L2 / TCP).
//
// Layout: planes [B][H][W][C] with C innermost. NCHW tensors are passed as B' = B*C planes with C' = 1, NHWC tensors as
// they are, so one kernel serves both layouts and consecutive lanes always touch consecutive addresses. Elements are
// float32 or bf16; arithmetic is fp32, rounded once on store.
//
// Closed forms (derived from the reference's conv / conv_transpose index arithmetic, pinned against torch in
// tests/test_blur_gpu.py):
//   down : out[y][x] = sum_{i,j<3} k3[i] k3[j] in[refl(2y+i-1)][refl(2x+j-1)],   k3 = [1 2 1]/4,  Ho = (H-1)/2 + 1
//   up   : out[2m]   = (3 in[m] + in[max(m-1, 0)]) / 4,  out[2m+1] = (3 in[m] + in[min(m+1, H-1)]) / 4  per axis
// and the backward kernels gather the transposed taps (no atomics, deterministic).

#include "common.h"

namespace {

typedef unsigned short bf16_t;

template <class T> __device__ __forceinline__ float ld(const T *p);
template <> __device__ __forceinline__ float ld<float>(const float *p) { return *p; }
template <> __device__ __forceinline__ float ld<bf16_t>(const bf16_t *p) { return __uint_as_float((unsigned)*p << 16); }
template <class T> __device__ __forceinline__ void st(T *p, float v);
template <> __device__ __forceinline__ void st<float>(float *p, float v) { *p = v; }
template <> __device__ __forceinline__ void st<bf16_t>(bf16_t *p, float v) {   // round to nearest even (torch's conversion)
    unsigned u = __float_as_uint(v);
    if ((u & 0x7fffffffu) > 0x7f800000u) { *p = (bf16_t)((u >> 16) | 0x40); return; }
    u += 0x7fffu + ((u >> 16) & 1u);
    *p = (bf16_t)(u >> 16);
}

// V elements per thread: 1, or 8 bf16 channels of an NHWC tensor as one 16-byte access (C % 8 == 0; round 3: the scalar form moved
// 2 bytes per lane and spent its time in address arithmetic -- blur_down_bwd 230 us per launch in the GAN-seg step)
template <class T, int V> __device__ __forceinline__ void ldv(const T *p, float (&o)[V]) {
    if constexpr (V == 1) o[0] = ld(p);
    else {
        const uint4 u = *reinterpret_cast<const uint4 *>(p);
        const unsigned w[4] = {u.x, u.y, u.z, u.w};
#pragma unroll
        for (int k = 0; k < 4; k++) { o[2 * k] = __uint_as_float(w[k] << 16); o[2 * k + 1] = __uint_as_float(w[k] & 0xffff0000u); }
    }
}
template <class T, int V> __device__ __forceinline__ void stv(T *p, const float (&v)[V]) {
    if constexpr (V == 1) st(p, v[0]);
    else {
        bf16_t h[8];
#pragma unroll
        for (int k = 0; k < 8; k++) st(&h[k], v[k]);
        *reinterpret_cast<uint4 *>(p) = make_uint4(h[0] | ((unsigned)h[1] << 16), h[2] | ((unsigned)h[3] << 16), h[4] | ((unsigned)h[5] << 16), h[6] | ((unsigned)h[7] << 16));
    }
}

__device__ __forceinline__ int refl(int p, int n) { return p < 0 ? -p : (p >= n ? 2 * (n - 1) - p : p); }

struct Idx { int b, y, x, c; };
// i enumerates (b, y, x, c / V); .c is the first channel of the thread's group
template <int V> __device__ __forceinline__ Idx split(long long i, int Hh, int Ww, int C) {
    Idx r;
    const int CV = C / V;
    if (i <= 0x7fffffffLL) {
        // the tensors of these networks: 32-bit unsigned divisions (round 5: the three 64-bit divide / modulo pairs were most of the
        // instructions of these kernels -- ~100 each, against a few 16-byte loads and one store per thread)
        unsigned u = (unsigned)i;
        r.c = (int)(u % (unsigned)CV) * V; u /= (unsigned)CV;
        r.x = (int)(u % (unsigned)Ww); u /= (unsigned)Ww;
        r.y = (int)(u % (unsigned)Hh);
        r.b = (int)(u / (unsigned)Hh);
        return r;
    }
    r.c = (int)(i % CV) * V; i /= CV;
    r.x = (int)(i % Ww); i /= Ww;
    r.y = (int)(i % Hh);
    r.b = (int)(i / Hh);
    return r;
}

// ---- reflection pad -------------------------------------------------------------------------------------------------
template <class T, int V>
__global__ void __launch_bounds__(256) reflect_pad_fwd_kernel(const T *__restrict__ in, T *__restrict__ out, int H, int W, int C, int pad, long long total) {
    const long long i = (long long)blockIdx.x * 256 + threadIdx.x;
    if (i >= total) return;
    const Idx o = split<V>(i, H + 2 * pad, W + 2 * pad, C);
    const int y = refl(o.y - pad, H), x = refl(o.x - pad, W);
    const T *src = in + (((long long)o.b * H + y) * W + x) * C + o.c;
    if constexpr (V == 1) out[i] = *src;
    else *reinterpret_cast<uint4 *>(out + i * V) = *reinterpret_cast<const uint4 *>(src);
}

// Padded positions that read source index r: r + pad itself, the mirror across the low border (r in 1..pad) and the
// mirror across the high border (r in n-1-pad..n-2), in padded coordinates.
__device__ __forceinline__ int pad_sources(int r, int n, int pad, int *p) {
    int k = 0;
    p[k++] = r + pad;
    if (r >= 1 && r <= pad) p[k++] = pad - r;
    if (r <= n - 2 && r >= n - 1 - pad) p[k++] = pad + 2 * (n - 1) - r;
    return k;
}

template <class T, int V>
__global__ void __launch_bounds__(256) reflect_pad_bwd_kernel(const T *__restrict__ g, T *__restrict__ dx, int H, int W, int C, int pad, long long total) {
    const long long i = (long long)blockIdx.x * 256 + threadIdx.x;
    if (i >= total) return;
    const Idx o = split<V>(i, H, W, C);
    int py[3], px[3];
    const int ny = pad_sources(o.y, H, pad, py), nx = pad_sources(o.x, W, pad, px);
    const int Hp = H + 2 * pad, Wp = W + 2 * pad;
    float acc[V];
#pragma unroll
    for (int v = 0; v < V; v++) acc[v] = 0.f;
    for (int a = 0; a < ny; ++a)
        for (int b = 0; b < nx; ++b) {
            float t[V];
            ldv<T, V>(g + (((long long)o.b * Hp + py[a]) * Wp + px[b]) * C + o.c, t);
#pragma unroll
            for (int v = 0; v < V; v++) acc[v] += t[v];
        }
    stv<T, V>(dx + i * V, acc);
}

// ---- blur + stride 2 ------------------------------------------------------------------------------------------------
template <class T, int V>
__global__ void __launch_bounds__(256) blur_down_fwd_kernel(const T *__restrict__ in, T *__restrict__ out, int H, int W, int C, int Ho, int Wo, long long total) {
    const long long i = (long long)blockIdx.x * 256 + threadIdx.x;
    if (i >= total) return;
    const Idx o = split<V>(i, Ho, Wo, C);
    const T *img = in + (long long)o.b * H * W * C + o.c;
    int ys[3], xs[3];
    for (int t = 0; t < 3; ++t) { ys[t] = refl(2 * o.y + t - 1, H); xs[t] = refl(2 * o.x + t - 1, W); }
    float acc[V];
#pragma unroll
    for (int v = 0; v < V; v++) acc[v] = 0.f;
    for (int a = 0; a < 3; ++a) {
        const T *row = img + (long long)ys[a] * W * C;
        float r0[V], r1[V], r2[V];
        ldv<T, V>(row + (long long)xs[0] * C, r0); ldv<T, V>(row + (long long)xs[1] * C, r1); ldv<T, V>(row + (long long)xs[2] * C, r2);
#pragma unroll
        for (int v = 0; v < V; v++) {
            const float r = r0[v] + 2.f * r1[v] + r2[v];
            acc[v] += (a == 1 ? 2.f : 1.f) * r;
        }
    }
#pragma unroll
    for (int v = 0; v < V; v++) acc[v] *= (1.f / 16.f);
    stv<T, V>(out + i * V, acc);
}

// Outputs (index, weight*4) whose taps land on source index r of an axis of length n (no of outputs: no).
__device__ __forceinline__ int down_taps(int r, int n, int no, int *idx, float *w) {
    int p[3], k = 0;
    const int np = pad_sources(r, n, 1, p);   // padded coordinates = tap position 2y + i
    for (int a = 0; a < np; ++a)
        for (int t = 0; t < 3; ++t) {
            const int e = p[a] - t;
            if (e >= 0 && !(e & 1) && (e >> 1) < no) { idx[k] = e >> 1; w[k] = t == 1 ? 2.f : 1.f; ++k; }
        }
    return k;
}

template <class T, int V>
__global__ void __launch_bounds__(256) blur_down_bwd_kernel(const T *__restrict__ g, T *__restrict__ dx, int H, int W, int C, int Ho, int Wo, long long total) {
    const long long i = (long long)blockIdx.x * 256 + threadIdx.x;
    if (i >= total) return;
    const Idx o = split<V>(i, H, W, C);
    int iy[9], ix[9];
    float wy[9], wx[9];
    const int ny = down_taps(o.y, H, Ho, iy, wy), nx = down_taps(o.x, W, Wo, ix, wx);
    const T *gi = g + (long long)o.b * Ho * Wo * C + o.c;
    float acc[V];
#pragma unroll
    for (int v = 0; v < V; v++) acc[v] = 0.f;
    for (int a = 0; a < ny; ++a) {
        float r[V];
#pragma unroll
        for (int v = 0; v < V; v++) r[v] = 0.f;
        for (int b = 0; b < nx; ++b) {
            float t[V];
            ldv<T, V>(gi + ((long long)iy[a] * Wo + ix[b]) * C, t);
#pragma unroll
            for (int v = 0; v < V; v++) r[v] += wx[b] * t[v];
        }
#pragma unroll
        for (int v = 0; v < V; v++) acc[v] += wy[a] * r[v];
    }
#pragma unroll
    for (int v = 0; v < V; v++) acc[v] *= (1.f / 16.f);
    stv<T, V>(dx + i * V, acc);
}

// ---- x2 upsampling with the [1 3 3 1] filter ------------------------------------------------------------------------
template <class T, int V>
__global__ void __launch_bounds__(256) blur_up_fwd_kernel(const T *__restrict__ in, T *__restrict__ out, int H, int W, int C, long long total) {
    const long long i = (long long)blockIdx.x * 256 + threadIdx.x;
    if (i >= total) return;
    const Idx o = split<V>(i, 2 * H, 2 * W, C);
    const int my = o.y >> 1, mx = o.x >> 1;
    const int ny = (o.y & 1) ? min(my + 1, H - 1) : max(my - 1, 0), nx = (o.x & 1) ? min(mx + 1, W - 1) : max(mx - 1, 0);
    const T *img = in + (long long)o.b * H * W * C + o.c;
    float a[V], b[V], c[V], d[V], r[V];
    ldv<T, V>(img + ((long long)my * W + mx) * C, a); ldv<T, V>(img + ((long long)my * W + nx) * C, b);
    ldv<T, V>(img + ((long long)ny * W + mx) * C, c); ldv<T, V>(img + ((long long)ny * W + nx) * C, d);
#pragma unroll
    for (int v = 0; v < V; v++) r[v] = (3.f * (3.f * a[v] + b[v]) + (3.f * c[v] + d[v])) * (1.f / 16.f);
    stv<T, V>(out + i * V, r);
}

__device__ __forceinline__ int up_taps(int r, int n, int *idx, float *w) {
    int k = 0;
    idx[k] = 2 * r; w[k++] = 3.f;
    idx[k] = 2 * r + 1; w[k++] = 3.f;
    if (r + 1 < n) { idx[k] = 2 * r + 2; w[k++] = 1.f; } else { idx[k] = 2 * n - 1; w[k++] = 1.f; }   // neighbour above, or the replicated border
    if (r >= 1) { idx[k] = 2 * r - 1; w[k++] = 1.f; } else { idx[k] = 0; w[k++] = 1.f; }
    return k;
}

template <class T, int V>
__global__ void __launch_bounds__(256) blur_up_bwd_kernel(const T *__restrict__ g, T *__restrict__ dx, int H, int W, int C, long long total) {
    const long long i = (long long)blockIdx.x * 256 + threadIdx.x;
    if (i >= total) return;
    const Idx o = split<V>(i, H, W, C);
    int iy[4], ix[4];
    float wy[4], wx[4];
    up_taps(o.y, H, iy, wy);
    up_taps(o.x, W, ix, wx);
    const T *gi = g + (long long)o.b * 4 * H * W * C + o.c;
    float acc[V];
#pragma unroll
    for (int v = 0; v < V; v++) acc[v] = 0.f;
    for (int a = 0; a < 4; ++a) {
        float r[V];
#pragma unroll
        for (int v = 0; v < V; v++) r[v] = 0.f;
        for (int b = 0; b < 4; ++b) {
            float t[V];
            ldv<T, V>(gi + ((long long)iy[a] * 2 * W + ix[b]) * C, t);
#pragma unroll
            for (int v = 0; v < V; v++) r[v] += wx[b] * t[v];
        }
#pragma unroll
        for (int v = 0; v < V; v++) acc[v] += wy[a] * r[v];
    }
#pragma unroll
    for (int v = 0; v < V; v++) acc[v] *= (1.f / 16.f);
    stv<T, V>(dx + i * V, acc);
}

inline unsigned blocks(long long total) { return (unsigned)((total + 255) / 256); }

bool bad(octa_ctx *ctx, const void *a, const void *b, int dtype, int B, int H, int W, int C, const char *who) {
    if (!ctx || !a || !b || B <= 0 || H <= 0 || W <= 0 || C <= 0 || (dtype != 0 && dtype != 1) || a == b) {
        octa::set_error("%s: bad arguments (dtype must be 0 = float32 or 1 = bf16)", who);
        return true;
    }
    return false;
}

}  // namespace

// TOTAL = number of ELEMENTS; the bf16 NHWC form with C % 8 == 0 runs one thread per 8 channels
#define OCTA_BLUR_LAUNCH(KERNEL, TOTAL, ...)                                                                                    \
    do {                                                                                                                        \
        OCTA_HIP_CHECK(hipSetDevice(ctx->device));                                                                              \
        if ((TOTAL) > 0x7fffffffLL * 256) { octa::set_error("blur: tensor too large"); return -2; }                             \
        if (dtype == 0) hipLaunchKernelGGL((KERNEL<float, 1>), dim3(blocks(TOTAL)), dim3(256), 0, (hipStream_t)stream, static_cast<const float *>(d_in), static_cast<float *>(d_out), __VA_ARGS__, (long long)(TOTAL)); \
        else if (C % 8 == 0 && ((uintptr_t)d_in % 16 == 0) && ((uintptr_t)d_out % 16 == 0)) hipLaunchKernelGGL((KERNEL<bf16_t, 8>), dim3(blocks((TOTAL) / 8)), dim3(256), 0, (hipStream_t)stream, static_cast<const bf16_t *>(d_in), static_cast<bf16_t *>(d_out), __VA_ARGS__, (long long)(TOTAL) / 8); \
        else hipLaunchKernelGGL((KERNEL<bf16_t, 1>), dim3(blocks(TOTAL)), dim3(256), 0, (hipStream_t)stream, static_cast<const bf16_t *>(d_in), static_cast<bf16_t *>(d_out), __VA_ARGS__, (long long)(TOTAL)); \
        OCTA_HIP_CHECK(hipGetLastError());                                                                                      \
        return 0;                                                                                                               \
    } while (0)

extern "C" int octa_reflect_pad_fwd(octa_ctx *ctx, const void *d_in, void *d_out, int dtype, int B, int H, int W, int C, int pad, void *stream) {
    if (bad(ctx, d_in, d_out, dtype, B, H, W, C, "octa_reflect_pad_fwd")) return -2;
    if (pad < 1 || pad >= H || pad >= W) { octa::set_error("octa_reflect_pad_fwd: pad must be in [1, min(H, W))"); return -2; }
    const long long total = (long long)B * (H + 2 * pad) * (W + 2 * pad) * C;
    OCTA_BLUR_LAUNCH(reflect_pad_fwd_kernel, total, H, W, C, pad);
}

extern "C" int octa_reflect_pad_bwd(octa_ctx *ctx, const void *d_in, void *d_out, int dtype, int B, int H, int W, int C, int pad, void *stream) {
    if (bad(ctx, d_in, d_out, dtype, B, H, W, C, "octa_reflect_pad_bwd")) return -2;
    if (pad < 1 || pad >= H || pad >= W) { octa::set_error("octa_reflect_pad_bwd: pad must be in [1, min(H, W))"); return -2; }
    const long long total = (long long)B * H * W * C;
    OCTA_BLUR_LAUNCH(reflect_pad_bwd_kernel, total, H, W, C, pad);
}

extern "C" int octa_blur_down_fwd(octa_ctx *ctx, const void *d_in, void *d_out, int dtype, int B, int H, int W, int C, void *stream) {
    if (bad(ctx, d_in, d_out, dtype, B, H, W, C, "octa_blur_down_fwd")) return -2;
    if (H < 2 || W < 2) { octa::set_error("octa_blur_down_fwd: reflect pad 1 needs H, W >= 2"); return -2; }
    const int Ho = (H - 1) / 2 + 1, Wo = (W - 1) / 2 + 1;
    const long long total = (long long)B * Ho * Wo * C;
    OCTA_BLUR_LAUNCH(blur_down_fwd_kernel, total, H, W, C, Ho, Wo);
}

extern "C" int octa_blur_down_bwd(octa_ctx *ctx, const void *d_in, void *d_out, int dtype, int B, int H, int W, int C, void *stream) {
    if (bad(ctx, d_in, d_out, dtype, B, H, W, C, "octa_blur_down_bwd")) return -2;
    if (H < 2 || W < 2) { octa::set_error("octa_blur_down_bwd: reflect pad 1 needs H, W >= 2"); return -2; }
    const int Ho = (H - 1) / 2 + 1, Wo = (W - 1) / 2 + 1;
    const long long total = (long long)B * H * W * C;
    OCTA_BLUR_LAUNCH(blur_down_bwd_kernel, total, H, W, C, Ho, Wo);
}

extern "C" int octa_blur_up_fwd(octa_ctx *ctx, const void *d_in, void *d_out, int dtype, int B, int H, int W, int C, void *stream) {
    if (bad(ctx, d_in, d_out, dtype, B, H, W, C, "octa_blur_up_fwd")) return -2;
    const long long total = (long long)B * 4 * H * W * C;
    OCTA_BLUR_LAUNCH(blur_up_fwd_kernel, total, H, W, C);
}

extern "C" int octa_blur_up_bwd(octa_ctx *ctx, const void *d_in, void *d_out, int dtype, int B, int H, int W, int C, void *stream) {
    if (bad(ctx, d_in, d_out, dtype, B, H, W, C, "octa_blur_up_bwd")) return -2;
    const long long total = (long long)B * H * W * C;
    OCTA_BLUR_LAUNCH(blur_up_bwd_kernel, total, H, W, C);
}
