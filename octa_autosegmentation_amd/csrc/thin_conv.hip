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
constexpr int WG_ROWS = 2;       // rows of `a` one wgrad block folds into its partial sums

// All three kernels put the 64 channels of a chunk on the 64 lanes of a wave and walk along an image row, so that every access
// to the wide tensor is one coalesced 128-byte line per pixel and the one-channel operand is wave-uniform (LDS broadcast). The
// K x K window slides in registers: the pixel loop is unrolled K times and column (u + kx) % K of the window is a compile-time
// register, so a step costs K loads (one per window row), not K*K.

// weights of the chunk, [64][KK] in global memory (contiguous) -> registers of lane c, through LDS so that the global read is coalesced
template <int KK>
__device__ __forceinline__ void load_weights(const float *__restrict__ w, int c0, int flip, float *wl /* LDS [64][KK + 1] */, float (&wr)[KK]) {
    for (int i = threadIdx.x; i < 64 * KK; i += blockDim.x) wl[(i / KK) * (KK + 1) + i % KK] = w[(size_t)c0 * KK + i];
    __syncthreads();
    const int lane = threadIdx.x & 63;
#pragma unroll
    for (int t = 0; t < KK; t++) wr[t] = wl[lane * (KK + 1) + (flip ? KK - 1 - t : t)];
}

// ---- expand: 1 -> C -----------------------------------------------------------------------------------------------------
// block = one output row x one 64-channel chunk; its 4 waves take a quarter of the row each. LDS: the K rows of s the output row
// meets (fp32, zero outside) + the weight staging area.
template <int K>
__global__ void __launch_bounds__(TC_THREADS)
thin_expand_kernel(const bf16_t *__restrict__ s, const float *__restrict__ w, const float *__restrict__ bias, bf16_t *__restrict__ out,
                   int N, int Hs, int Ws, int Ho, int Wo, int C, int pad, int flip, float slope) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    constexpr int KK = K * K;
    float *wl = reinterpret_cast<float *>(smem);                 // [64][KK + 1]
    float *srow = wl + 64 * (KK + 1);                            // [K][Wo + K - 1]
    const int wrow = Wo + K - 1;
    const int c0 = blockIdx.y * 64, lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int oy = blockIdx.x % Ho, n = blockIdx.x / Ho;
    float wr[KK];
    load_weights<KK>(w, c0, flip, wl, wr);
    for (int i = threadIdx.x; i < K * wrow; i += TC_THREADS) {
        const int ky = i / wrow, j = i % wrow;
        const int iy = oy + ky - pad, ix = j - pad;
        srow[i] = (iy >= 0 && iy < Hs && ix >= 0 && ix < Ws) ? bf2f(s[((size_t)n * Hs + iy) * Ws + ix]) : 0.f;
    }
    __syncthreads();
    const float b = bias ? bias[c0 + lane] : 0.f;
    const int per = (Wo + 3) / 4;
    const int x0 = wave * per, x1 = (x0 + per < Wo) ? x0 + per : Wo;
    if (x0 >= x1) return;
    float win[K][K];                                             // win[ky][col]: column (x + kx) lives in register (x - x0 + kx) % K
#pragma unroll
    for (int ky = 0; ky < K; ky++)
#pragma unroll
        for (int kx = 0; kx < K - 1; kx++) win[ky][kx] = srow[ky * wrow + x0 + kx];
    bf16_t *orow = out + (((size_t)n * Ho + oy) * Wo) * C + c0 + lane;
    for (int xb = x0; xb < x1; xb += K) {
#pragma unroll
        for (int u = 0; u < K; u++) {
            const int x = xb + u;
            if (x < x1) {
#pragma unroll
                for (int ky = 0; ky < K; ky++) win[ky][(u + K - 1) % K] = srow[ky * wrow + x + K - 1];
                float acc = b;
#pragma unroll
                for (int ky = 0; ky < K; ky++)
#pragma unroll
                    for (int kx = 0; kx < K; kx++) acc = __builtin_fmaf(win[ky][(u + kx) % K], wr[ky * K + kx], acc);
                acc = acc > 0.f ? acc : acc * slope;
                orow[(size_t)x * C] = f2bf(acc);
            }
        }
    }
}

// ---- squeeze: C -> 1 ----------------------------------------------------------------------------------------------------
// block = one output row; the weights of ALL chunks sit in LDS ([C][KK + 1]). A wave takes groups of P consecutive output pixels:
// per 64-channel chunk it loads the K x (P + K - 1) window of the wide tensor -- every load independent of the others, coordinates
// clamped and masked instead of branched on -- and each loaded pixel feeds the up to K outputs it belongs to; the P sums are reduced
// over the lanes at the end of the group.
template <int K, int P>
__global__ void __launch_bounds__(TC_THREADS)
thin_squeeze_kernel(const bf16_t *__restrict__ a, const float *__restrict__ w, const float *__restrict__ bias, bf16_t *__restrict__ out,
                    int N, int Ha, int Wa, int Ho, int Wo, int C, int pad, int flip) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    constexpr int KK = K * K;
    float *wl = reinterpret_cast<float *>(smem);                 // [C][KK + 1]
    for (int i = threadIdx.x; i < C * KK; i += TC_THREADS) wl[(i / KK) * (KK + 1) + i % KK] = w[i];
    __syncthreads();
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int oy = blockIdx.x % Ho, n = blockIdx.x / Ho;
    const float b = bias ? bias[0] : 0.f;
    const int groups = (Wo + P - 1) / P;
    for (int g = wave; g < groups; g += TC_THREADS / 64) {
        const int ox0 = g * P;
        // columns of the window: clamped index + 0/1 factor (wave-uniform), so that no load depends on a branch or a select mask
        int ixc[P + K - 1];
        float mcol[P + K - 1];
#pragma unroll
        for (int j = 0; j < P + K - 1; j++) {
            const int ix = ox0 + j - pad;
            const bool ok = ix >= 0 && ix < Wa;
            ixc[j] = ok ? ix : 0;
            mcol[j] = ok ? 1.f : 0.f;
        }
        float acc[P];
#pragma unroll
        for (int o = 0; o < P; o++) acc[o] = 0.f;
        for (int c0 = 0; c0 < C; c0 += 64) {
            float wr[KK];
#pragma unroll
            for (int t = 0; t < KK; t++) wr[t] = wl[(c0 + lane) * (KK + 1) + (flip ? KK - 1 - t : t)];
            float xv[K][P + K - 1];
#pragma unroll
            for (int ky = 0; ky < K; ky++) {
                const int iy = oy + ky - pad;
                const bool rowok = iy >= 0 && iy < Ha;
                const bf16_t *row = a + (((size_t)n * Ha + (rowok ? iy : 0)) * Wa) * C + c0 + lane;
                if (!rowok) {                                       // the row lies in the zero padding: its weights do not count
#pragma unroll
                    for (int kx = 0; kx < K; kx++) wr[ky * K + kx] = 0.f;
                }
#pragma unroll
                for (int j = 0; j < P + K - 1; j++) xv[ky][j] = bf2f(row[(size_t)ixc[j] * C]) * mcol[j];
            }
#pragma unroll
            for (int ky = 0; ky < K; ky++)
#pragma unroll
                for (int j = 0; j < P + K - 1; j++)
#pragma unroll
                    for (int kx = 0; kx < K; kx++) {
                        const int o = j - kx;
                        if (o >= 0 && o < P) acc[o] = __builtin_fmaf(xv[ky][j], wr[ky * K + kx], acc[o]);
                    }
        }
#pragma unroll
        for (int o = 0; o < P; o++) {
            float v = acc[o];
#pragma unroll
            for (int d = 32; d >= 1; d >>= 1) v += __shfl_xor(v, d, 64);
            acc[o] = v;
        }
        float mine = 0.f;
#pragma unroll
        for (int o = 0; o < P; o++) mine = lane == o ? acc[o] : mine;
        if (lane < P && ox0 + lane < Wo) out[((size_t)n * Ho + oy) * Wo + ox0 + lane] = f2bf(mine + b);
    }
}

// ---- squeeze, wide form (round 5) ---------------------------------------------------------------------------------------
// The kernel above puts a channel on a lane: 2 bytes per lane and load, 98 loads and a 64-lane butterfly per 8 outputs -- 0.50 ms for the
// generator's 64 -> 1 head at 8 x 304^2, a tenth of the vector peak (profiles/r05_gan_steady_kernel_stats.csv). Here a thread owns 8 channels
// (ONE 16-byte piece) of PW consecutive output pixels: per kernel row it loads the PW + K - 1 pieces of the window once (a wave's load is
// 128 contiguous bytes per pixel), holds the row's K weight octets in registers (tap-major table in LDS, 16-byte reads), accumulates channel
// PAIRS (v_pk_fma_f32) and folds the C / 8 lanes of a pixel group at the end.
typedef float tc_f2 __attribute__((ext_vector_type(2)));

template <int K, int PW>
__global__ void __launch_bounds__(TC_THREADS)
thin_squeeze_wide_kernel(const bf16_t *__restrict__ a, const float *__restrict__ w, const float *__restrict__ bias, bf16_t *__restrict__ out,
                         int N, int Ha, int Wa, int Ho, int Wo, int C, int pad, int flip) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    constexpr int KK = K * K;
    float *wl = reinterpret_cast<float *>(smem);                 // [KK][C]: tap-major (flipped here if asked), channel fastest
    for (int i = threadIdx.x; i < C * KK; i += TC_THREADS) {
        const int c = i / KK, t = i % KK;
        wl[(flip ? KK - 1 - t : t) * C + c] = w[i];
    }
    __syncthreads();
    const int groups = C / 8, gshift = 31 - __clz(groups);       // a power of two <= 64 (host-checked)
    const int g = threadIdx.x & (groups - 1), q0 = threadIdx.x >> gshift, qstep = TC_THREADS >> gshift;
    const int oy = blockIdx.x % Ho, n = blockIdx.x / Ho;
    const float b = bias ? bias[0] : 0.f;
    const int quads = (Wo + PW - 1) / PW;
    for (int qd = q0; qd < quads; qd += qstep) {
        const int ox0 = qd * PW;
        tc_f2 acc[PW][4];
#pragma unroll
        for (int o = 0; o < PW; o++)
#pragma unroll
            for (int k = 0; k < 4; k++) acc[o][k] = tc_f2{0.f, 0.f};
#pragma unroll
        for (int ky = 0; ky < K; ky++) {
            const int iy = oy + ky - pad;
            if (iy < 0 || iy >= Ha) continue;                    // a row of zero padding (uniform over the block)
            tc_f2 wk[K][4];
#pragma unroll
            for (int kx = 0; kx < K; kx++) {
                const float4 w0 = *reinterpret_cast<const float4 *>(wl + (ky * K + kx) * C + g * 8), w1 = *reinterpret_cast<const float4 *>(wl + (ky * K + kx) * C + g * 8 + 4);
                wk[kx][0] = tc_f2{w0.x, w0.y}; wk[kx][1] = tc_f2{w0.z, w0.w}; wk[kx][2] = tc_f2{w1.x, w1.y}; wk[kx][3] = tc_f2{w1.z, w1.w};
            }
            const bf16_t *row = a + (((size_t)n * Ha + iy) * Wa) * C + g * 8;
            uint4 v[PW + K - 1];
#pragma unroll
            for (int j = 0; j < PW + K - 1; j++) {               // all loads of the row first: independent, index clamped, masked afterwards
                const int ix = ox0 + j - pad;
                v[j] = *reinterpret_cast<const uint4 *>(row + (size_t)(ix < 0 ? 0 : (ix >= Wa ? Wa - 1 : ix)) * C);
            }
#pragma unroll
            for (int j = 0; j < PW + K - 1; j++) {
                const int ix = ox0 + j - pad;
                const float m = (ix >= 0 && ix < Wa) ? 1.f : 0.f;
                const unsigned u[4] = {v[j].x, v[j].y, v[j].z, v[j].w};
                tc_f2 x2[4];
#pragma unroll
                for (int k = 0; k < 4; k++) x2[k] = tc_f2{__uint_as_float(u[k] << 16) * m, __uint_as_float(u[k] & 0xffff0000u) * m};
#pragma unroll
                for (int kx = 0; kx < K; kx++) {
                    const int o = j - kx;
                    if (o >= 0 && o < PW) {
#pragma unroll
                        for (int k = 0; k < 4; k++) acc[o][k] = __builtin_elementwise_fma(x2[k], wk[kx][k], acc[o][k]);
                    }
                }
            }
        }
        float r[PW];
#pragma unroll
        for (int o = 0; o < PW; o++) {
            const tc_f2 t = (acc[o][0] + acc[o][1]) + (acc[o][2] + acc[o][3]);
            float vsum = t.x + t.y;
            for (int d = 1; d < groups; d <<= 1) vsum += __shfl_xor(vsum, d, 64);
            r[o] = vsum + b;
        }
        if (g == 0) {
            bf16_t *dst = out + ((size_t)n * Ho + oy) * Wo + ox0;
#pragma unroll
            for (int o = 0; o < PW; o++)
                if (ox0 + o < Wo) dst[o] = f2bf(r[o]);
        }
    }
}

// ---- wgrad --------------------------------------------------------------------------------------------------------------
// g[c][t] = sum a[n][y][x][c] * s[n][y+ky-pad][x+kx-pad]. block = WG_ROWS rows of a x one 64-channel chunk, K waves: wave ky
// owns window row ky (K accumulators per lane), its window of s slides along the row (one LDS broadcast per pixel); the values of
// a are fetched 2K pixels at a time (independent loads, index clamped and masked). Per-block partial sums go to `partial`
// [gridDim.x][C][K*K (+1: the plain sum of a, the bias gradient of the 1 -> C layer)].
template <int K>
__global__ void __launch_bounds__(K * 64)
thin_wgrad_kernel(const bf16_t *__restrict__ a, const bf16_t *__restrict__ s, float *__restrict__ partial,
                  int N, int Ha, int Wa, int Hs, int Ws, int C, int pad, int flip) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    constexpr int KK = K * K, U = 2 * K;
    float *srow = reinterpret_cast<float *>(smem);               // [K][Wa + K - 1 + U]: U zero columns of slack behind each row
    const int wrow = Wa + K - 1 + U;
    const int c = threadIdx.x & 63, ky = threadIdx.x >> 6;
    const int c0 = blockIdx.y * 64;
    float acc[K], asum = 0.f;
#pragma unroll
    for (int kx = 0; kx < K; kx++) acc[kx] = 0.f;
    for (int r = 0; r < WG_ROWS; r++) {
        const long long rowid = (long long)blockIdx.x * WG_ROWS + r;
        if (rowid >= (long long)N * Ha) break;
        const int n = (int)(rowid / Ha), qy = (int)(rowid % Ha);
        __syncthreads();
        for (int i = threadIdx.x; i < K * wrow; i += K * 64) {
            const int wy = i / wrow, j = i % wrow;
            const int iy = qy + wy - pad, ix = j - pad;
            srow[i] = (iy >= 0 && iy < Hs && ix >= 0 && ix < Ws) ? bf2f(s[((size_t)n * Hs + iy) * Ws + ix]) : 0.f;
        }
        __syncthreads();
        const bf16_t *arow = a + (((size_t)n * Ha + qy) * Wa) * C + c0 + c;
        const float *sr = srow + ky * wrow;
        float sw[K];
#pragma unroll
        for (int kx = 0; kx < K - 1; kx++) sw[kx] = sr[kx];
        for (int xb = 0; xb < Wa; xb += U) {
            float av[U];
#pragma unroll
            for (int u = 0; u < U; u++) {
                const int qx = xb + u;
                const float v = bf2f(arow[(size_t)(qx < Wa ? qx : Wa - 1) * C]);
                av[u] = qx < Wa ? v : 0.f;
            }
#pragma unroll
            for (int u = 0; u < U; u++) {
                sw[(u + K - 1) % K] = sr[xb + u + K - 1];
                asum += av[u];
#pragma unroll
                for (int kx = 0; kx < K; kx++) acc[kx] = __builtin_fmaf(av[u], sw[(u + kx) % K], acc[kx]);
            }
        }
    }
    float *dst = partial + ((size_t)blockIdx.x * C + c0 + c) * (KK + 1);
#pragma unroll
    for (int kx = 0; kx < K; kx++) {
        const int t = ky * K + kx;
        dst[flip ? KK - 1 - t : t] = acc[kx];
    }
    if (ky == 0) dst[KK] = asum;
}

// ---- wgrad, wide form (round 5) -------------------------------------------------------------------------------------------
// As the squeeze kernel: a lane per channel meant 2 bytes per lane and load and K waves re-reading the row (0.54 + 0.42 ms per GAN-seg step
// for the generator's two 7 x 7 layers). Here a thread owns 8 channels (one 16-byte piece) of 4 consecutive pixels; wave ky still owns
// window row ky (K x 4 pair accumulators per thread), the window of s comes from the staged rows in LDS, the pixel lanes of a channel group
// are folded once per block and the partial sums keep the layout the reduction kernel expects. Rows per block: WG_ROWS_WIDE.
constexpr int WG_ROWS_WIDE = 4;

template <int K>
__global__ void __launch_bounds__(K * 64)
thin_wgrad_wide_kernel(const bf16_t *__restrict__ a, const bf16_t *__restrict__ s, float *__restrict__ partial,
                       int N, int Ha, int Wa, int Hs, int Ws, int C, int pad, int flip) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    constexpr int KK = K * K, PW = 4;
    const int wrow = Wa + K - 1 + PW;                            // PW columns of slack behind each row (a quad may hang over the row's end)
    float *srow = reinterpret_cast<float *>(smem);               // [K][wrow]
    const int lane = threadIdx.x & 63, ky = threadIdx.x >> 6;
    const int groups = C / 8, gshift = 31 - __clz(groups);       // a power of two <= 64 (host-checked)
    const int g = lane & (groups - 1), q0 = lane >> gshift, qstep = 64 >> gshift;
    tc_f2 acc[K][4], asum[4];
#pragma unroll
    for (int kx = 0; kx < K; kx++)
#pragma unroll
        for (int k = 0; k < 4; k++) acc[kx][k] = tc_f2{0.f, 0.f};
#pragma unroll
    for (int k = 0; k < 4; k++) asum[k] = tc_f2{0.f, 0.f};
    const int quads = (Wa + PW - 1) / PW;
    for (int r = 0; r < WG_ROWS_WIDE; r++) {
        const long long rowid = (long long)blockIdx.x * WG_ROWS_WIDE + r;
        if (rowid >= (long long)N * Ha) break;
        const int n = (int)(rowid / Ha), qy = (int)(rowid % Ha);
        __syncthreads();
        for (int i = threadIdx.x; i < K * wrow; i += K * 64) {
            const int wy = i / wrow, j = i % wrow;
            const int iy = qy + wy - pad, ix = j - pad;
            srow[i] = (iy >= 0 && iy < Hs && ix >= 0 && ix < Ws) ? bf2f(s[((size_t)n * Hs + iy) * Ws + ix]) : 0.f;
        }
        __syncthreads();
        const bf16_t *arow = a + (((size_t)n * Ha + qy) * Wa) * C + g * 8;
        const float *sr = srow + ky * wrow;
        for (int qd = q0; qd < quads; qd += qstep) {
            const int x0 = qd * PW;
            uint4 v[PW];
#pragma unroll
            for (int o = 0; o < PW; o++) v[o] = *reinterpret_cast<const uint4 *>(arow + (size_t)(x0 + o < Wa ? x0 + o : Wa - 1) * C);
            float sw[PW + K - 1];
#pragma unroll
            for (int j = 0; j < PW + K - 1; j++) sw[j] = sr[x0 + j];
#pragma unroll
            for (int o = 0; o < PW; o++) {
                const float m = x0 + o < Wa ? 1.f : 0.f;
                const unsigned u[4] = {v[o].x, v[o].y, v[o].z, v[o].w};
                tc_f2 a2[4];
#pragma unroll
                for (int k = 0; k < 4; k++) { a2[k] = tc_f2{__uint_as_float(u[k] << 16) * m, __uint_as_float(u[k] & 0xffff0000u) * m}; asum[k] += a2[k]; }
#pragma unroll
                for (int kx = 0; kx < K; kx++) {
                    const tc_f2 s2 = {sw[o + kx], sw[o + kx]};
#pragma unroll
                    for (int k = 0; k < 4; k++) acc[kx][k] = __builtin_elementwise_fma(a2[k], s2, acc[kx][k]);
                }
            }
        }
    }
    // fold the pixel lanes of every channel group (lanes g, g + groups, ...), then lane g writes its 8 channels
#pragma unroll
    for (int kx = 0; kx < K; kx++)
#pragma unroll
        for (int k = 0; k < 4; k++)
            for (int d = groups; d < 64; d <<= 1) { acc[kx][k].x += __shfl_xor(acc[kx][k].x, d, 64); acc[kx][k].y += __shfl_xor(acc[kx][k].y, d, 64); }
    if (ky == 0) {
#pragma unroll
        for (int k = 0; k < 4; k++)
            for (int d = groups; d < 64; d <<= 1) { asum[k].x += __shfl_xor(asum[k].x, d, 64); asum[k].y += __shfl_xor(asum[k].y, d, 64); }
    }
    if (lane < groups) {
#pragma unroll
        for (int k = 0; k < 4; k++)
#pragma unroll
            for (int h = 0; h < 2; h++) {
                float *dst = partial + ((size_t)blockIdx.x * C + g * 8 + 2 * k + h) * (KK + 1);
#pragma unroll
                for (int kx = 0; kx < K; kx++) {
                    const int t = ky * K + kx;
                    dst[flip ? KK - 1 - t : t] = h ? acc[kx][k].y : acc[kx][k].x;
                }
                if (ky == 0) dst[KK] = h ? asum[k].y : asum[k].x;
            }
    }
}

// sum of the per-block partials: a block takes 16 consecutive outputs x 16 slices of the partial list (round 5: 64 x 4 left a thread a
// dependent chain of 300 loads: 82 us for 3 200 outputs)
__global__ void __launch_bounds__(TC_THREADS)
thin_wgrad_reduce_kernel(const float *__restrict__ partial, int n_blocks, int C, int KK, float *__restrict__ g, float *__restrict__ asum) {
    __shared__ float part[16][16];
    const int total = C * (KK + 1);
    const int el = threadIdx.x & 15, slice = threadIdx.x >> 4, i = blockIdx.x * 16 + el;
    float v0 = 0.f, v1 = 0.f;
    if (i < total) {
        int b = slice;
        for (; b + 16 < n_blocks; b += 32) { v0 += partial[(size_t)b * total + i]; v1 += partial[(size_t)(b + 16) * total + i]; }
        if (b < n_blocks) v0 += partial[(size_t)b * total + i];
    }
    part[slice][el] = v0 + v1;
    __syncthreads();
    if (slice != 0 || i >= total) return;
    float v = 0.f;
#pragma unroll
    for (int k = 0; k < 16; k++) v += part[k][el];
    const int c = i / (KK + 1), t = i % (KK + 1);
    if (t < KK) g[(size_t)c * KK + t] = v;
    else if (asum) asum[c] = v;
}

// LeakyReLU' applied to an incoming gradient by the sign of the layer's OUTPUT (slope > 0: the output's sign is its input's): out = y > 0 ? dy :
// dy * slope, one rounding. The expand layer fuses the activation into its forward; its backward used three torch launches for this (compare,
// scale, select: 125 us per 94 MB gradient of the discriminator's stem against ~40 us here).
__global__ void __launch_bounds__(256)
lrelu_bwd_kernel(const bf16_t *__restrict__ y, const bf16_t *__restrict__ dy, bf16_t *__restrict__ out, long n8, long n, float slope) {
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < n8; i += (long)gridDim.x * 256) {
        const uint4 a = reinterpret_cast<const uint4 *>(y)[i], g = reinterpret_cast<const uint4 *>(dy)[i];
        const unsigned av[4] = {a.x, a.y, a.z, a.w}, gv[4] = {g.x, g.y, g.z, g.w};
        unsigned r[4];
#pragma unroll
        for (int k = 0; k < 4; k++) {
            const bf16_t y0 = (bf16_t)(av[k] & 0xffffu), y1 = (bf16_t)(av[k] >> 16), g0 = (bf16_t)(gv[k] & 0xffffu), g1 = (bf16_t)(gv[k] >> 16);
            const bf16_t o0 = bf2f(y0) > 0.f ? g0 : f2bf(bf2f(g0) * slope), o1 = bf2f(y1) > 0.f ? g1 : f2bf(bf2f(g1) * slope);
            r[k] = (unsigned)o0 | ((unsigned)o1 << 16);
        }
        reinterpret_cast<uint4 *>(out)[i] = make_uint4(r[0], r[1], r[2], r[3]);
    }
    if (blockIdx.x == 0 && threadIdx.x < (int)(n - 8 * n8)) {          // the last n % 8 elements
        const long i = 8 * n8 + threadIdx.x;
        out[i] = bf2f(y[i]) > 0.f ? dy[i] : f2bf(bf2f(dy[i]) * slope);
    }
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
    const size_t lds = ((size_t)64 * (K * K + 1) + (size_t)K * (Wo + K - 1)) * sizeof(float);
    if (lds > 64 * 1024) { octa::set_error("octa_thinconv_expand: rows of %d pixels do not fit the LDS window", Wo); return -2; }
    const dim3 grid((unsigned)((long long)N * Ho), (unsigned)(C / 64));
    if (K == 7) hipLaunchKernelGGL(thin_expand_kernel<7>, grid, dim3(TC_THREADS), lds, stream, (const bf16_t *)d_s, (const float *)d_w, (const float *)d_bias, (bf16_t *)d_out, N, Hs, Ws, Ho, Wo, C, pad, flip, slope);
    else hipLaunchKernelGGL(thin_expand_kernel<4>, grid, dim3(TC_THREADS), lds, stream, (const bf16_t *)d_s, (const float *)d_w, (const float *)d_bias, (bf16_t *)d_out, N, Hs, Ws, Ho, Wo, C, pad, flip, slope);
    OCTA_HIP_CHECK(hipGetLastError());
    return 0;
}

extern "C" int octa_thinconv_squeeze(octa_ctx *ctx, const void *d_a, const void *d_w, const void *d_bias, void *d_out, int N, int Ha, int Wa, int C,
                                     int K, int pad, int flip, void *stream_) {
    if (!ctx || !d_a || !d_w || !d_out) { octa::set_error("octa_thinconv_squeeze: null argument"); return -2; }
    if (K == 3) {
        // the data gradient of a one-channel 3 x 3 first layer (C -> 1; csrc/conv.hip's c1 kernels are its forward and weight gradient): the
        // segmentor's input in the GAN-seg step is the generator's image, so the gradient has to reach it. Wide form only: 8 - 64 channels
        if (N <= 0 || Ha <= 0 || Wa <= 0 || pad < 0 || pad >= K || Ha + 2 * pad < K || Wa + 2 * pad < K || (C != 8 && C != 16 && C != 32 && C != 64)) {
            octa::set_error("octa_thinconv_squeeze: K = 3 needs 8, 16, 32 or 64 channels (N %d, H %d, W %d, C %d, pad %d)", N, Ha, Wa, C, pad);
            return -2;
        }
        OCTA_HIP_CHECK(hipSetDevice(ctx->device));
        const int Ho3 = Ha + 2 * pad - K + 1, Wo3 = Wa + 2 * pad - K + 1;
        hipLaunchKernelGGL((thin_squeeze_wide_kernel<3, 4>), dim3((unsigned)((long long)N * Ho3)), dim3(TC_THREADS), (size_t)C * 9 * sizeof(float), (hipStream_t)stream_,
                           (const bf16_t *)d_a, (const float *)d_w, (const float *)d_bias, (bf16_t *)d_out, N, Ha, Wa, Ho3, Wo3, C, pad, flip);
        OCTA_HIP_CHECK(hipGetLastError());
        return 0;
    }
    if (!shape_ok("octa_thinconv_squeeze", N, Ha, Wa, C, K, pad)) return -2;
    OCTA_HIP_CHECK(hipSetDevice(ctx->device));
    hipStream_t stream = (hipStream_t)stream_;
    const int Ho = Ha + 2 * pad - K + 1, Wo = Wa + 2 * pad - K + 1;
    const unsigned grid = (unsigned)((long long)N * Ho);
    const size_t lds = (size_t)C * (K * K + 1) * sizeof(float);
    if (lds > 64 * 1024) { octa::set_error("octa_thinconv_squeeze: %d x %d weights of %d channels do not fit the LDS", K, K, C); return -2; }
    // wide form: a thread per 8 channels of 4 output pixels (OCTA_THIN_WIDE=0: the lane-per-channel kernels of rounds 1-4)
    constexpr int wide = 1;
    const int groups = C / 8;
    if (wide && C % 8 == 0 && groups <= 64 && (groups & (groups - 1)) == 0 && (size_t)C * K * K * sizeof(float) <= 64 * 1024) {
        const size_t wlds = (size_t)C * K * K * sizeof(float);
        if (K == 7) hipLaunchKernelGGL((thin_squeeze_wide_kernel<7, 4>), dim3(grid), dim3(TC_THREADS), wlds, stream, (const bf16_t *)d_a, (const float *)d_w, (const float *)d_bias, (bf16_t *)d_out, N, Ha, Wa, Ho, Wo, C, pad, flip);
        else hipLaunchKernelGGL((thin_squeeze_wide_kernel<4, 4>), dim3(grid), dim3(TC_THREADS), wlds, stream, (const bf16_t *)d_a, (const float *)d_w, (const float *)d_bias, (bf16_t *)d_out, N, Ha, Wa, Ho, Wo, C, pad, flip);
        OCTA_HIP_CHECK(hipGetLastError());
        return 0;
    }
    if (K == 7) hipLaunchKernelGGL((thin_squeeze_kernel<7, 8>), dim3(grid), dim3(TC_THREADS), lds, stream, (const bf16_t *)d_a, (const float *)d_w, (const float *)d_bias, (bf16_t *)d_out, N, Ha, Wa, Ho, Wo, C, pad, flip);
    else hipLaunchKernelGGL((thin_squeeze_kernel<4, 8>), dim3(grid), dim3(TC_THREADS), lds, stream, (const bf16_t *)d_a, (const float *)d_w, (const float *)d_bias, (bf16_t *)d_out, N, Ha, Wa, Ho, Wo, C, pad, flip);
    OCTA_HIP_CHECK(hipGetLastError());
    return 0;
}

extern "C" int octa_lrelu_bwd_bf16(octa_ctx *ctx, const void *d_y, const void *d_dy, void *d_out, int64_t n, float slope, void *stream_) {
    if (!ctx || !d_y || !d_dy || !d_out || n <= 0) { octa::set_error("octa_lrelu_bwd_bf16: bad arguments"); return -2; }
    if ((reinterpret_cast<size_t>(d_y) | reinterpret_cast<size_t>(d_dy) | reinterpret_cast<size_t>(d_out)) & 15) { octa::set_error("octa_lrelu_bwd_bf16: the tensors must be 16-byte aligned"); return -2; }
    OCTA_HIP_CHECK(hipSetDevice(ctx->device));
    const long n8 = (long)(n / 8);
    long blocks = (n8 + 255) / 256;
    if (blocks > 16L * ctx->num_cus) blocks = 16L * ctx->num_cus;
    if (blocks < 1) blocks = 1;
    hipLaunchKernelGGL(lrelu_bwd_kernel, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream_, (const bf16_t *)d_y, (const bf16_t *)d_dy, (bf16_t *)d_out, n8, (long)n, slope);
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
    if (!shape_ok("octa_thinconv_wgrad", N, Ha, Wa, C, K, pad)) return -2;
    if (Hs <= 0 || Ws <= 0) { octa::set_error("octa_thinconv_wgrad: bad s extent"); return -2; }
    OCTA_HIP_CHECK(hipSetDevice(ctx->device));
    hipStream_t stream = (hipStream_t)stream_;
    unsigned blocks = (unsigned)(((long long)N * Ha + WG_ROWS - 1) / WG_ROWS);
    const size_t lds = (size_t)K * (Wa + K - 1 + 2 * K) * sizeof(float);
    if (lds > 60 * 1024) { octa::set_error("octa_thinconv_wgrad: rows of %d pixels do not fit the LDS window", Wa); return -2; }
    constexpr int wide = 1;
    const int groups = C / 8;
    const int total = C * (K * K + 1);
    if (wide && groups <= 64 && (groups & (groups - 1)) == 0) {
        // (fewer, larger blocks than octa_thinconv_wgrad_scratch_floats sized the scratch for)
        blocks = (unsigned)(((long long)N * Ha + WG_ROWS_WIDE - 1) / WG_ROWS_WIDE);
        if (K == 7) hipLaunchKernelGGL(thin_wgrad_wide_kernel<7>, dim3(blocks), dim3(7 * 64), lds, stream, (const bf16_t *)d_a, (const bf16_t *)d_s, (float *)d_scratch, N, Ha, Wa, Hs, Ws, C, pad, flip);
        else hipLaunchKernelGGL(thin_wgrad_wide_kernel<4>, dim3(blocks), dim3(4 * 64), lds, stream, (const bf16_t *)d_a, (const bf16_t *)d_s, (float *)d_scratch, N, Ha, Wa, Hs, Ws, C, pad, flip);
        hipLaunchKernelGGL(thin_wgrad_reduce_kernel, dim3((total + 15) / 16), dim3(TC_THREADS), 0, stream, (const float *)d_scratch, (int)blocks, C, K * K, (float *)d_g, (float *)d_asum);
        OCTA_HIP_CHECK(hipGetLastError());
        return 0;
    }
    if (K == 7) hipLaunchKernelGGL(thin_wgrad_kernel<7>, dim3(blocks, C / 64), dim3(7 * 64), lds, stream, (const bf16_t *)d_a, (const bf16_t *)d_s, (float *)d_scratch, N, Ha, Wa, Hs, Ws, C, pad, flip);
    else hipLaunchKernelGGL(thin_wgrad_kernel<4>, dim3(blocks, C / 64), dim3(4 * 64), lds, stream, (const bf16_t *)d_a, (const bf16_t *)d_s, (float *)d_scratch, N, Ha, Wa, Hs, Ws, C, pad, flip);
    hipLaunchKernelGGL(thin_wgrad_reduce_kernel, dim3((total + 15) / 16), dim3(TC_THREADS), 0, stream, (const float *)d_scratch, (int)blocks, C, K * K, (float *)d_g, (float *)d_asum);
    OCTA_HIP_CHECK(hipGetLastError());
    return 0;
}
