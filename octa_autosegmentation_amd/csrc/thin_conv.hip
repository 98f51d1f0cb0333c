// thin_conv.hip -- the convolutions of the GAN networks that have ONE channel on one side:
//   ResnetGenerator   ReflectionPad2d(3) + Conv2d(1 -> 64, 7)  stem,  ReflectionPad2d(3) + Conv2d(64 -> 1, 7)  head   (models/networks.py:360-368)
//   NLayerDiscriminator  Conv2d(1 -> 64, 4, 1, 1) stem,  Conv2d(512 -> 1, 4, 1, 1) head                               (models/networks.py:433-442)
// They are not matrix-core shaped (a 1 x K*K or K*K x 1 GEMM side): the reference sends them to the vendor library, which
// runs them as im2col + GEMM + col2im (13 % of the GAN-seg step measured in round 1). Here each is a streaming kernel:
// HBM-bound, algorithmic bytes = one read / write of the C-channel tensor (12 MB per 304^2 x 64 bf16 image), the K*K taps
// of neighbouring pixels meet in LDS / L1.
//
// Three kernels cover forward, data gradient and weight gradient of both shapes (stride 1, zero padding `pad`; a
// reflection pad is applied by csrc/blur.hip's kernel in front):
//   expand : out[n][y][x][c] = act(b[c] + sum_t s[n][y+ky-pad][x+kx-pad] * w[c][t])      1 -> C   (stem forward, head data gradient)
//   squeeze: out[n][y][x]    = b + sum_t sum_c a[n][y+ky-pad][x+kx-pad][c] * w[c][t]      C -> 1   (head forward, stem data gradient)
//   wgrad  : g[c][t]         = sum_{n,y,x} a[n][y][x][c] * s[n][y+ky-pad][x+kx-pad]                (both weight gradients)
// with t = ky*K + kx, and `flip` addressing w / g at K*K-1-t: the data gradient of a convolution is the convolution of
// the output gradient with the flipped kernel and pad' = K-1-pad, and the weight gradient of the C -> 1 layer is the wgrad
// kernel with the roles of input and output gradient exchanged (derivation in models/thin_conv.py).
// Layouts: s, squeeze-out [N][H][W] bf16 (one channel: NHWC == NCHW); a, expand-out [N][H][W][C] bf16; w, g [C][K*K] fp32
// (= Conv2d.weight [C][1][K][K] or [1][C][K][K]). Arithmetic fp32, one rounding on store. wgrad is deterministic: per-block
// partial sums, then one reduction kernel.

#include "common.h"

namespace {

typedef unsigned short bf16_t;

__device__ __forceinline__ float bf2f(bf16_t v) { return __uint_as_float((unsigned)v << 16); }
__device__ __forceinline__ bf16_t f2bf(float v) {   // round to nearest even (torch's conversion)
    unsigned u = __float_as_uint(v);
    if ((u & 0x7fffffffu) > 0x7f800000u) return (bf16_t)((u >> 16) | 0x40);
    u += 0x7fffu + ((u >> 16) & 1u);
    return (bf16_t)(u >> 16);
}

constexpr int TC_THREADS = 256;
constexpr int EXPAND_SEGS = 8;   // row segments one expand block walks with the weights resident in LDS

// ---- expand: 1 -> C -----------------------------------------------------------------------------------------------------
// A thread owns one output pixel x 8 channels (one 16-byte store); a block of 256 threads = 256 / (C / 8) consecutive pixels
// of a row per segment. LDS: weights [K*K][C] fp32 (tap-major: the 8 channels of a thread are contiguous), the K x (pixels + K - 1)
// window of s as fp32.
template <int K>
__global__ void __launch_bounds__(TC_THREADS)
thin_expand_kernel(const bf16_t *__restrict__ s, const float *__restrict__ w, const float *__restrict__ bias, bf16_t *__restrict__ out,
                   int N, int Hs, int Ws, int Ho, int Wo, int C, int pad, int flip, float slope) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    constexpr int KK = K * K;
    float *wl = reinterpret_cast<float *>(smem);                 // [KK][C]
    const int g8 = C >> 3, ppw = TC_THREADS / g8;                // channel groups, pixels per segment
    float *win = wl + KK * C;                                    // [K][ppw + K - 1]
    const int wrow = ppw + K - 1;
    for (int i = threadIdx.x; i < KK * C; i += TC_THREADS) {
        const int c = i % C, t = i / C;
        wl[i] = w[(size_t)c * KK + (flip ? KK - 1 - t : t)];
    }
    const int g = threadIdx.x % g8, p = threadIdx.x / g8;
    float b[8];
#pragma unroll
    for (int j = 0; j < 8; j++) b[j] = bias ? bias[8 * g + j] : 0.f;
    const int segs_per_row = (Wo + ppw - 1) / ppw;
    const long long n_seg = (long long)N * Ho * segs_per_row;
    for (int k = 0; k < EXPAND_SEGS; k++) {
        const long long seg = (long long)blockIdx.x * EXPAND_SEGS + k;
        if (seg >= n_seg) break;
        const int sx = (int)(seg % segs_per_row), oy = (int)((seg / segs_per_row) % Ho), n = (int)(seg / ((long long)segs_per_row * Ho));
        const int ox0 = sx * ppw;
        __syncthreads();                                         // weights staged / previous window consumed
        for (int i = threadIdx.x; i < K * wrow; i += TC_THREADS) {
            const int ky = i / wrow, j = i % wrow;
            const int iy = oy + ky - pad, ix = ox0 + j - pad;
            win[i] = (iy >= 0 && iy < Hs && ix >= 0 && ix < Ws) ? bf2f(s[((size_t)n * Hs + iy) * Ws + ix]) : 0.f;
        }
        __syncthreads();
        const int ox = ox0 + p;
        if (ox >= Wo) continue;
        float acc[8];
#pragma unroll
        for (int j = 0; j < 8; j++) acc[j] = b[j];
#pragma unroll
        for (int ky = 0; ky < K; ky++)
#pragma unroll
            for (int kx = 0; kx < K; kx++) {
                const float sv = win[ky * wrow + p + kx];
                const float4 w0 = *reinterpret_cast<const float4 *>(wl + (ky * K + kx) * C + 8 * g);
                const float4 w1 = *reinterpret_cast<const float4 *>(wl + (ky * K + kx) * C + 8 * g + 4);
                acc[0] += sv * w0.x; acc[1] += sv * w0.y; acc[2] += sv * w0.z; acc[3] += sv * w0.w;
                acc[4] += sv * w1.x; acc[5] += sv * w1.y; acc[6] += sv * w1.z; acc[7] += sv * w1.w;
            }
        union { bf16_t h[8]; uint4 v; } o;
#pragma unroll
        for (int j = 0; j < 8; j++) { float v = acc[j]; v = v > 0.f ? v : v * slope; o.h[j] = f2bf(v); }
        *reinterpret_cast<uint4 *>(out + (((size_t)n * Ho + oy) * Wo + ox) * C + 8 * g) = o.v;
    }
}

// ---- squeeze: C -> 1 ----------------------------------------------------------------------------------------------------
// A wave owns P consecutive output pixels of a row, lane = channel of the current 64-channel chunk with its K*K weights in
// registers: every input pixel of the K x (P + K - 1) window is loaded once (one coalesced 128-byte load) and feeds the up to K
// outputs it belongs to; the P sums are reduced over the lanes at the end.
template <int K, int P>
__global__ void __launch_bounds__(TC_THREADS)
thin_squeeze_kernel(const bf16_t *__restrict__ a, const float *__restrict__ w, const float *__restrict__ bias, bf16_t *__restrict__ out,
                    int N, int Ha, int Wa, int Ho, int Wo, int C, int pad, int flip) {
    constexpr int KK = K * K;
    const int lane = threadIdx.x & 63;
    const int groups_per_row = (Wo + P - 1) / P;
    const long long grp = (long long)blockIdx.x * (TC_THREADS / 64) + (threadIdx.x >> 6);
    if (grp >= (long long)N * Ho * groups_per_row) return;
    const int gx = (int)(grp % groups_per_row), oy = (int)((grp / groups_per_row) % Ho), n = (int)(grp / ((long long)groups_per_row * Ho));
    const int ox0 = gx * P;
    float acc[P];
#pragma unroll
    for (int o = 0; o < P; o++) acc[o] = 0.f;
    for (int c0 = 0; c0 < C; c0 += 64) {
        float wr[KK];
#pragma unroll
        for (int t = 0; t < KK; t++) wr[t] = w[(size_t)(c0 + lane) * KK + (flip ? KK - 1 - t : t)];
#pragma unroll
        for (int ky = 0; ky < K; ky++) {
            const int iy = oy + ky - pad;
            if (iy < 0 || iy >= Ha) continue;
            const bf16_t *row = a + (((size_t)n * Ha + iy) * Wa) * C + c0 + lane;
#pragma unroll
            for (int j = 0; j < P + K - 1; j++) {
                const int ix = ox0 + j - pad;
                const float xv = (ix >= 0 && ix < Wa) ? bf2f(row[(size_t)ix * C]) : 0.f;
#pragma unroll
                for (int kx = 0; kx < K; kx++) {
                    const int o = j - kx;
                    if (o >= 0 && o < P) acc[o] += xv * wr[ky * K + kx];
                }
            }
        }
    }
#pragma unroll
    for (int o = 0; o < P; o++) {
        float v = acc[o];
#pragma unroll
        for (int d = 32; d >= 1; d >>= 1) v += __shfl_xor(v, d, 64);
        acc[o] = v;
    }
    const float b = bias ? bias[0] : 0.f;
#pragma unroll
    for (int o = 0; o < P; o++)
        if (lane == o && ox0 + o < Wo) out[((size_t)n * Ho + oy) * Wo + ox0 + o] = f2bf(acc[o] + b);
}

// ---- wgrad --------------------------------------------------------------------------------------------------------------
// g[c][t] = sum a[n][y][x][c] * s[n][y+ky-pad][x+kx-pad]. Block = (rows of a) x (64-channel chunk); thread = channel x every
// fourth tap (accumulators in registers). The K rows of s a row of a meets are staged in LDS as fp32 (zero outside). Per-block
// partial sums go to `partial` [gridDim.x][C][K*K (+1: the plain sum of a, the bias gradient of the 1 -> C layer)].
constexpr int WG_ROWS = 2;
template <int K>
__global__ void __launch_bounds__(TC_THREADS)
thin_wgrad_kernel(const bf16_t *__restrict__ a, const bf16_t *__restrict__ s, float *__restrict__ partial,
                  int N, int Ha, int Wa, int Hs, int Ws, int C, int pad, int flip) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    constexpr int KK = K * K, NT = (KK + 3) / 4;
    float *srow = reinterpret_cast<float *>(smem);               // [K][Wa + K - 1]
    const int wrow = Wa + K - 1;
    const int c = threadIdx.x & 63, tq = threadIdx.x >> 6;
    const int c0 = blockIdx.y * 64;
    float acc[NT], asum = 0.f;
    int off[NT];
#pragma unroll
    for (int j = 0; j < NT; j++) {
        acc[j] = 0.f;
        const int t = tq + 4 * j;
        off[j] = t < KK ? (t / K) * wrow + (t % K) : 0;
    }
    for (int r = 0; r < WG_ROWS; r++) {
        const long long rowid = (long long)blockIdx.x * WG_ROWS + r;
        if (rowid >= (long long)N * Ha) break;
        const int n = (int)(rowid / Ha), qy = (int)(rowid % Ha);
        __syncthreads();
        for (int i = threadIdx.x; i < K * wrow; i += TC_THREADS) {
            const int ky = i / wrow, j = i % wrow;
            const int iy = qy + ky - pad, ix = j - pad;
            srow[i] = (iy >= 0 && iy < Hs && ix >= 0 && ix < Ws) ? bf2f(s[((size_t)n * Hs + iy) * Ws + ix]) : 0.f;
        }
        __syncthreads();
        const bf16_t *arow = a + (((size_t)n * Ha + qy) * Wa) * C + c0 + c;
#pragma unroll 4
        for (int qx = 0; qx < Wa; qx++) {
            const float av = bf2f(arow[(size_t)qx * C]);
            asum += av;
#pragma unroll
            for (int j = 0; j < NT; j++) acc[j] += av * srow[off[j] + qx];
        }
    }
    float *dst = partial + ((size_t)blockIdx.x * C + c0 + c) * (KK + 1);
#pragma unroll
    for (int j = 0; j < NT; j++) {
        const int t = tq + 4 * j;
        if (t < KK) dst[flip ? KK - 1 - t : t] = acc[j];
    }
    if (tq == 0) dst[KK] = asum;
}

__global__ void __launch_bounds__(TC_THREADS)
thin_wgrad_reduce_kernel(const float *__restrict__ partial, int n_blocks, int C, int KK, float *__restrict__ g, float *__restrict__ asum) {
    const int i = blockIdx.x * TC_THREADS + threadIdx.x;        // over C * (KK + 1)
    if (i >= C * (KK + 1)) return;
    float v = 0.f;
    for (int b = 0; b < n_blocks; b++) v += partial[(size_t)b * C * (KK + 1) + i];
    const int c = i / (KK + 1), t = i % (KK + 1);
    if (t < KK) g[(size_t)c * KK + t] = v;
    else if (asum) asum[c] = v;
}

bool shape_ok(const char *fn, int N, int H, int W, int C, int K, int pad) {
    if (N <= 0 || H <= 0 || W <= 0 || (K != 4 && K != 7) || pad < 0 || pad >= K || H + 2 * pad < K || W + 2 * pad < K) {
        octa::set_error("%s: unsupported shape (N %d, H %d, W %d, K %d, pad %d; K is 4 or 7)", fn, N, H, W, K, pad);
        return false;
    }
    if (C < 64 || C % 64 || C > 1024) { octa::set_error("%s: the wide side needs a multiple of 64 channels up to 1024 (got %d)", fn, C); return false; }
    return true;
}

}  // namespace

extern "C" int octa_thinconv_expand(octa_ctx *ctx, const void *d_s, const void *d_w, const void *d_bias, void *d_out, int N, int Hs, int Ws, int C,
                                    int K, int pad, int flip, float slope, void *stream_) {
    if (!ctx || !d_s || !d_w || !d_out) { octa::set_error("octa_thinconv_expand: null argument"); return -2; }
    if (!shape_ok("octa_thinconv_expand", N, Hs, Ws, C, K, pad)) return -2;
    OCTA_HIP_CHECK(hipSetDevice(ctx->device));
    hipStream_t stream = (hipStream_t)stream_;
    const int Ho = Hs + 2 * pad - K + 1, Wo = Ws + 2 * pad - K + 1;
    const int g8 = C / 8, ppw = TC_THREADS / g8;
    if (ppw < 1 || TC_THREADS % g8) { octa::set_error("octa_thinconv_expand: C / 8 must divide %d", TC_THREADS); return -2; }
    const long long n_seg = (long long)N * Ho * ((Wo + ppw - 1) / ppw);
    const unsigned grid = (unsigned)((n_seg + EXPAND_SEGS - 1) / EXPAND_SEGS);
    const size_t lds = ((size_t)K * K * C + (size_t)K * (ppw + K - 1)) * sizeof(float);
    if (lds > 64 * 1024) { octa::set_error("octa_thinconv_expand: %d x %d weights of %d channels do not fit the LDS", K, K, C); return -2; }
    if (K == 7) hipLaunchKernelGGL(thin_expand_kernel<7>, dim3(grid), dim3(TC_THREADS), lds, stream, (const bf16_t *)d_s, (const float *)d_w, (const float *)d_bias, (bf16_t *)d_out, N, Hs, Ws, Ho, Wo, C, pad, flip, slope);
    else hipLaunchKernelGGL(thin_expand_kernel<4>, dim3(grid), dim3(TC_THREADS), lds, stream, (const bf16_t *)d_s, (const float *)d_w, (const float *)d_bias, (bf16_t *)d_out, N, Hs, Ws, Ho, Wo, C, pad, flip, slope);
    OCTA_HIP_CHECK(hipGetLastError());
    return 0;
}

extern "C" int octa_thinconv_squeeze(octa_ctx *ctx, const void *d_a, const void *d_w, const void *d_bias, void *d_out, int N, int Ha, int Wa, int C,
                                     int K, int pad, int flip, void *stream_) {
    if (!ctx || !d_a || !d_w || !d_out) { octa::set_error("octa_thinconv_squeeze: null argument"); return -2; }
    if (!shape_ok("octa_thinconv_squeeze", N, Ha, Wa, C, K, pad)) return -2;
    OCTA_HIP_CHECK(hipSetDevice(ctx->device));
    hipStream_t stream = (hipStream_t)stream_;
    const int Ho = Ha + 2 * pad - K + 1, Wo = Wa + 2 * pad - K + 1;
    constexpr int P = 8;
    const long long groups = (long long)N * Ho * ((Wo + P - 1) / P);
    const unsigned grid = (unsigned)((groups + TC_THREADS / 64 - 1) / (TC_THREADS / 64));
    if (K == 7) hipLaunchKernelGGL((thin_squeeze_kernel<7, P>), dim3(grid), dim3(TC_THREADS), 0, stream, (const bf16_t *)d_a, (const float *)d_w, (const float *)d_bias, (bf16_t *)d_out, N, Ha, Wa, Ho, Wo, C, pad, flip);
    else hipLaunchKernelGGL((thin_squeeze_kernel<4, P>), dim3(grid), dim3(TC_THREADS), 0, stream, (const bf16_t *)d_a, (const float *)d_w, (const float *)d_bias, (bf16_t *)d_out, N, Ha, Wa, Ho, Wo, C, pad, flip);
    OCTA_HIP_CHECK(hipGetLastError());
    return 0;
}

extern "C" long long octa_thinconv_wgrad_scratch_floats(int N, int Ha, int C, int K) {
    if (N <= 0 || Ha <= 0 || C <= 0 || K <= 0) return 0;
    const long long blocks = ((long long)N * Ha + WG_ROWS - 1) / WG_ROWS;
    return blocks * C * (K * K + 1);
}

extern "C" int octa_thinconv_wgrad(octa_ctx *ctx, const void *d_a, const void *d_s, void *d_scratch, void *d_g, void *d_asum, int N, int Ha, int Wa,
                                   int Hs, int Ws, int C, int K, int pad, int flip, void *stream_) {
    if (!ctx || !d_a || !d_s || !d_scratch || !d_g) { octa::set_error("octa_thinconv_wgrad: null argument"); return -2; }
    if (!shape_ok("octa_thinconv_wgrad", N, Ha, Wa, C, K, pad) || Hs <= 0 || Ws <= 0) { if (Hs <= 0 || Ws <= 0) octa::set_error("octa_thinconv_wgrad: bad s extent"); return -2; }
    OCTA_HIP_CHECK(hipSetDevice(ctx->device));
    hipStream_t stream = (hipStream_t)stream_;
    const unsigned blocks = (unsigned)(((long long)N * Ha + WG_ROWS - 1) / WG_ROWS);
    const size_t lds = (size_t)K * (Wa + K - 1) * sizeof(float);
    if (lds > 60 * 1024) { octa::set_error("octa_thinconv_wgrad: rows of %d pixels do not fit the LDS window", Wa); return -2; }
    if (K == 7) hipLaunchKernelGGL(thin_wgrad_kernel<7>, dim3(blocks, C / 64), dim3(TC_THREADS), lds, stream, (const bf16_t *)d_a, (const bf16_t *)d_s, (float *)d_scratch, N, Ha, Wa, Hs, Ws, C, pad, flip);
    else hipLaunchKernelGGL(thin_wgrad_kernel<4>, dim3(blocks, C / 64), dim3(TC_THREADS), lds, stream, (const bf16_t *)d_a, (const bf16_t *)d_s, (float *)d_scratch, N, Ha, Wa, Hs, Ws, C, pad, flip);
    const int total = C * (K * K + 1);
    hipLaunchKernelGGL(thin_wgrad_reduce_kernel, dim3((total + TC_THREADS - 1) / TC_THREADS), dim3(TC_THREADS), 0, stream, (const float *)d_scratch, (int)blocks, C, K * K, (float *)d_g, (float *)d_asum);
    OCTA_HIP_CHECK(hipGetLastError());
    return 0;
}
