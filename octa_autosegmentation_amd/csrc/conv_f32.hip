// conv_f32.hip -- single-precision convolution on the gfx950 matrix cores (v_mfma_f32_32x32x2_f32: fp32 operands, fp32
// accumulation -- no bf16 / xf32 rounding anywhere), NCHW, for the paths of the reference that run WITHOUT mixed precision:
//   * test.py:79 and validate.py (model.inference outside torch.cuda.amp.autocast): every convolution of DynUNet
//     (models/networks.py:6 -> MONAI DynUNet: 3x3 stride 1 / 2, 2x2 stride-2 and 1x1 transposed, 1x1 head with bias);
//   * `General.amp: false` forward passes that need no gradient;
//   * round 6: the same for the GAN networks -- test.py with `General.inference: G` and the frozen generator of the docker pipeline
//     (test.py:75-82, docker/dockershell.sh:14-16) run ResnetGenerator in fp32: 7x7 stem / head behind a reflection pad, 3x3 layers with
//     zero padding or none (behind ReflectionPad2d(1)); NLayerDiscriminator's 4x4 stride-1 layers.
// north_star asks for segmentation logits within 1e-4 of the reference's fp32 CPU path: with exact fp32 products and fp32 sums the
// only difference left is the summation order (tests/test_conv_f32_gpu.py: 1x1x1216x1216 DynUNet logits against the CPU modules).
//
// Implicit GEMM, D[co][pixel] += W[co][ci, tap] * X[ci][pixel + tap]: a 256-thread workgroup owns 32 * MB output channels x an
// 8 x 32 output-pixel tile; a wave owns two tile rows (two 32-pixel N-blocks) x MB M-blocks. Per slice of KC = 8 (stride 2: 4) input channels
// the halo tile [KC][IH][IW] and the weight slice [KC][K*K][32 * MB] are staged in LDS (zero-filled outside the image / beyond Cin /
// beyond Cout); an MFMA consumes two input channels of one tap: lane l supplies W[co = l % 32][ci + l / 32] and
// X[ci + l / 32][pixel l % 32] (one ds_read_b32 each, conflict-free: consecutive lanes read consecutive words). The accumulator
// fragment holds, per register, 32 consecutive pixels of one output channel, so the NCHW stores are 128-byte rows straight from
// registers. A 2x2 stride-2 transposed convolution is four 1x1 launches with a scattered store (osc = 2: one output parity each).
//
// Roofline: MFMA-bound on paper (dense fp32 matrix peak 157 TFLOP/s, MI355X_MICROARCH.md); algorithmic HBM bytes = input + output
// activations once (fp32) + weights. Measured figures: DESIGN.md section 4.2c'.

#include "common.h"

namespace {

typedef __attribute__((ext_vector_type(16))) float f32x16;

constexpr int F_TH = 8, F_TW = 32, F_THREADS = 256;

// X: [N][Cin][H][W]; Wp: [Cin][K*K][CoutW] (packed by the caller: output channel innermost); Y: [N][CoutY][Ho*osc][Wo*osc] written at
// (oy * osc + ooy, ox * osc + oox). out(co, oy, ox) = bias[co] + sum_{ci, r, s} X(ci, oy * S + r - pad, ox * S + s - pad) * Wp[ci][r * K + s][co].
template <int K, int S, int MB, bool TR = false>
__global__ void __launch_bounds__(F_THREADS)
conv_f32_kernel(const float *__restrict__ X, const float *__restrict__ Wp, const float *__restrict__ bias, float *__restrict__ Y,
                int Cin, int H, int W, int Cout, int CoutW, int Ho, int Wo, int pad, int tiles_x, int osc, int ooy, int oox, const float *__restrict__ zero) {
    // input channels per slice (stride 2: the halo tile is 4x the output tile, half the depth keeps the prefetch in registers; the 4x4 and 7x7
    // layers of the GAN networks, round 6: 16 / 49 taps per channel -- 4 / 2 channels keep the weight slice at 16 / 25 KB of LDS)
    constexpr int KC = K >= 7 ? 2 : ((S == 2 || K >= 4) ? 4 : 8);
    constexpr int IH = (F_TH - 1) * S + K, IW = (F_TW - 1) * S + K;
    constexpr int IWP = IW | 1;                          // odd row pitch: the two half-waves (channels ci, ci + 1) start on different banks
    constexpr int BM = 32 * MB, KK = K * K;
    constexpr int IN_FLOATS = KC * IH * IWP, W_FLOATS = KC * KK * BM;
    __shared__ float s_in[IN_FLOATS];
    __shared__ float s_w[W_FLOATS];
    static_assert(!TR || (K == 1 && S == 1 && MB == 2), "the fused 2x2 transposed convolution is a 1x1 product with two M-blocks");
    // TR: one launch of a 2x2 stride-2 transposed convolution. blockIdx.y = (block of 32 output channels, output row parity ta);
    // M-block mb holds the SAME 32 channels for output column parity mb, so a lane owns both pixels of an output pair (8-byte stores).
    const int tile = blockIdx.x, n = blockIdx.z, co0 = TR ? (blockIdx.y >> 1) * 32 : blockIdx.y * BM, ta = TR ? (blockIdx.y & 1) : 0;
    const int ty0 = (tile / tiles_x) * F_TH, tx0 = (tile % tiles_x) * F_TW;
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    const int m = lane & 31, kg = lane >> 5;
    const int iy0 = ty0 * S - pad, ix0 = tx0 * S - pad;
    const float *img = X + (size_t)n * Cin * H * W;

    f32x16 acc[2][MB];
#pragma unroll
    for (int a = 0; a < 2; a++)
#pragma unroll
        for (int b = 0; b < MB; b++)
#pragma unroll
            for (int k = 0; k < 16; k++) acc[a][b][k] = 0.f;

    // A slice is fetched into REGISTERS one slice ahead (all loads of a thread issued back to back, nothing waits on them until the
    // MFMAs of the current slice have been issued) and written to LDS after the compute: the first version loaded and stored element
    // by element in a loop, ~20 dependent global-load latencies per slice against ~2 us of MFMA work.
    constexpr int NIN = (KC * IH * IW + F_THREADS - 1) / F_THREADS, NW = (W_FLOATS + F_THREADS - 1) / F_THREADS;
    float pin[NIN], pw[NW];
    auto fetch = [&](int c0) {
#pragma unroll
        for (int j = 0; j < NIN; j++) {
            const int i = threadIdx.x + j * F_THREADS;
            const int c = i / (IH * IW), rem = i % (IH * IW), hy = rem / IW, hx = rem % IW;
            const int ci = c0 + c, yy = iy0 + hy, xx = ix0 + hx;
            const bool ok = i < KC * IH * IW && ci < Cin && yy >= 0 && yy < H && xx >= 0 && xx < W;
            pin[j] = *(ok ? img + ((ci * H + yy) * W + xx) : zero);   // padding reads a zero word: the predicate dies before the load is issued (Cin * H * W < 2^31: entry point)
        }
#pragma unroll
        for (int j = 0; j < NW; j++) {
            const int i = threadIdx.x + j * F_THREADS;
            const int row = i / BM, mm = i % BM;                      // row = channel of the slice * K*K + tap (BM is a power of two)
            const int col = TR ? co0 + (mm & 31) : co0 + mm;           // TR: Wp is [Cin][tap 2 ta + mb][Cout]
            const bool ok = i < W_FLOATS && row < (Cin - c0) * KK && col < Cout;
            pw[j] = *(ok ? Wp + ((size_t)(c0 * KK + row) * CoutW + (TR ? (2 * ta + (mm >> 5)) * Cout : 0) + col) : zero);
        }
    };
    auto stash = [&]() {
#pragma unroll
        for (int j = 0; j < NIN; j++) {
            const int i = threadIdx.x + j * F_THREADS;
            const int c = i / (IH * IW), rem = i % (IH * IW), hy = rem / IW, hx = rem % IW;
            if (i < KC * IH * IW) s_in[(c * IH + hy) * IWP + hx] = pin[j];
        }
#pragma unroll
        for (int j = 0; j < NW; j++)
            if (W_FLOATS % F_THREADS == 0 || threadIdx.x + j * F_THREADS < W_FLOATS) s_w[threadIdx.x + j * F_THREADS] = pw[j];
    };

    fetch(0);
    for (int c0 = 0; c0 < Cin; c0 += KC) {
        stash();
        __syncthreads();
        if (c0 + KC < Cin) fetch(c0 + KC);
#pragma unroll
        for (int cp = 0; cp < KC; cp += 2) {
            const int c = cp + kg;                        // this lane's input channel of the pair
#pragma unroll
            for (int r = 0; r < K; r++)
#pragma unroll
                for (int s = 0; s < K; s++) {
                    float a[MB], b[2];
#pragma unroll
                    for (int mb = 0; mb < MB; mb++) a[mb] = s_w[(c * KK + r * K + s) * BM + mb * 32 + m];
#pragma unroll
                    for (int rr = 0; rr < 2; rr++) b[rr] = s_in[(c * IH + (2 * wv + rr) * S + r) * IWP + m * S + s];
#pragma unroll
                    for (int rr = 0; rr < 2; rr++)
#pragma unroll
                        for (int mb = 0; mb < MB; mb++)
                            acc[rr][mb] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[mb], b[rr], acc[rr][mb], 0, 0, 0);
                }
        }
        __syncthreads();                                  // the slice has been consumed
    }
    // D[co][pixel]: register k of lane (m, kg) holds output channel (k & 3) + 8 * (k >> 2) + 4 * kg of the M-block, pixel column m
    const int Hy = Ho * osc, Wy = Wo * osc;
    const int ox = tx0 + m;
#pragma unroll
    for (int rr = 0; rr < 2; rr++) {
        const int oy = ty0 + 2 * wv + rr;
        if (oy >= Ho || ox >= Wo) continue;
        if constexpr (TR) {
#pragma unroll
            for (int k = 0; k < 16; k++) {
                const int co = co0 + (k & 3) + 8 * (k >> 2) + 4 * kg;
                if (co < Cout)
                    *reinterpret_cast<float2 *>(Y + (((size_t)n * Cout + co) * Hy + (oy * 2 + ta)) * Wy + ox * 2) = make_float2(acc[rr][0][k], acc[rr][1][k]);
            }
            continue;
        }
#pragma unroll
        for (int mb = 0; mb < MB; mb++)
#pragma unroll
            for (int k = 0; k < 16; k++) {
                const int co = co0 + mb * 32 + (k & 3) + 8 * (k >> 2) + 4 * kg;
                if (co < Cout) {
                    const float v = acc[rr][mb][k] + (bias ? bias[co] : 0.f);
                    Y[(((size_t)n * Cout + co) * Hy + (oy * osc + ooy)) * Wy + ox * osc + oox] = v;
                }
            }
    }
}

template <int K, int S, int MB, bool TR = false>
int launch_f32(const float *zero, const float *X, const float *Wp, const float *bias, float *Y, int N, int Cin, int H, int W, int Cout, int CoutW, int Ho, int Wo, int pad,
               int osc, int ooy, int oox, hipStream_t stream) {
    const int tiles_x = (Wo + F_TW - 1) / F_TW, tiles_y = (Ho + F_TH - 1) / F_TH;
    dim3 grid((unsigned)(tiles_x * tiles_y), TR ? (unsigned)(2 * ((Cout + 31) / 32)) : (unsigned)((Cout + 32 * MB - 1) / (32 * MB)), (unsigned)N);
    hipLaunchKernelGGL((conv_f32_kernel<K, S, MB, TR>), grid, dim3(F_THREADS), 0, stream, X, Wp, bias, Y, Cin, H, W, Cout, CoutW, Ho, Wo, pad, tiles_x, osc, ooy, oox, zero);
    OCTA_HIP_CHECK(hipGetLastError());
    return 0;
}

const float *zero_word(octa_ctx *ctx) {               // what the padding lanes of a slice fetch read
    if (!ctx->zero_page.p) {
        if (ctx->zero_page.reserve(256)) return nullptr;
        // hipMemset on device memory may return before the fill has run, and it runs on the NULL stream, which torch's (non-blocking) streams do
        // not wait for: the first DMA-staged launch of a fresh context could fetch its padding from an uncleared page. Wait for the device once.
        if (hipMemset(ctx->zero_page.p, 0, ctx->zero_page.cap) != hipSuccess || hipDeviceSynchronize() != hipSuccess) { octa::set_error("conv_f32: zero page memset failed"); ctx->zero_page.release(); return nullptr; }
    }
    return ctx->zero_page.as<float>();
}

}  // namespace

// Single-precision convolution, NCHW (see the file header). d_wp: weights packed as [Cin][K*K][cout_w] with cout_w >= Cout (a
// view of a larger packed tensor may be passed: rows are cout_w apart); d_bias: [Cout] or NULL. K / stride in {1/1, 3/1, 3/2, 4/1, 7/1}.
// The output tensor is [N][Cout][Ho * osc][Wo * osc]; osc = 1 writes it densely, osc = 2 writes the pixels of
// parity (ooy, oox) only (one of the four 1x1 products of a 2x2 stride-2 transposed convolution).
extern "C" int octa_conv2d_f32_nchw(octa_ctx *ctx, const float *d_x, const float *d_wp, const float *d_bias, float *d_y, int N, int Cin, int H, int W,
                                    int Cout, int cout_w, int K, int stride, int pad, int Ho, int Wo, int osc, int ooy, int oox, void *stream_) {
    if (!ctx || !d_x || !d_wp || !d_y || N <= 0 || Cin <= 0 || H <= 0 || W <= 0 || Cout <= 0 || cout_w < Cout || Ho <= 0 || Wo <= 0 || pad < 0) {
        octa::set_error("octa_conv2d_f32_nchw: bad arguments");
        return -2;
    }
    if ((osc != 1 && osc != 2) || ooy < 0 || ooy >= osc || oox < 0 || oox >= osc) { octa::set_error("octa_conv2d_f32_nchw: bad output scatter"); return -2; }
    if ((long)(Ho - 1) * stride + K - pad > (long)H + pad || (long)(Wo - 1) * stride + K - pad > (long)W + pad) {
        octa::set_error("octa_conv2d_f32_nchw: output size %dx%d reads beyond the padded input", Ho, Wo);
        return -2;
    }
    if ((long)Cin * H * W >= (1L << 31)) { octa::set_error("octa_conv2d_f32_nchw: one image of the input exceeds 2^31 elements"); return -2; }
    hipStream_t stream = (hipStream_t)stream_;
    OCTA_HIP_CHECK(hipSetDevice(ctx->device));
    const float *zero = zero_word(ctx);
    if (!zero) return -1;
    // 64 output channels per workgroup halve the input staging per product, but the workgroups of a launch are all resident at once
    // and share the CUs' matrix pipes: 380 workgroups on 256 CUs run at the pace of the CUs that hold two (74 %), 760 of half the
    // size at 99 %. Take the narrow variant when it balances the CUs better by more than its extra staging costs.
    auto balance = [&](int mb) {
        const long wgs = (long)((Wo + F_TW - 1) / F_TW) * ((Ho + F_TH - 1) / F_TH) * ((Cout + 32 * mb - 1) / (32 * mb)) * N;
        const long cus = ctx->num_cus > 0 ? ctx->num_cus : 256;
        return (double)wgs / (double)(cus * ((wgs + cus - 1) / cus));
    };
    const bool wide = Cout > 32 && (stride != 1 || balance(2) >= 0.9 * balance(1));     // stride 2 (half-depth slices) measured slower when narrow
#define OCTA_F32_CASE(KK_, SS_)                                                                                                              \
    if (K == KK_ && stride == SS_)                                                                                                           \
        return wide ? launch_f32<KK_, SS_, 2>(zero, d_x, d_wp, d_bias, d_y, N, Cin, H, W, Cout, cout_w, Ho, Wo, pad, osc, ooy, oox, stream)       \
                    : launch_f32<KK_, SS_, 1>(zero, d_x, d_wp, d_bias, d_y, N, Cin, H, W, Cout, cout_w, Ho, Wo, pad, osc, ooy, oox, stream);
    OCTA_F32_CASE(1, 1)
    OCTA_F32_CASE(3, 1)
    OCTA_F32_CASE(3, 2)
    OCTA_F32_CASE(4, 1)          // PatchGAN (models/networks.py:445-506: 4x4, stride 1, padding 1)
    OCTA_F32_CASE(7, 1)          // the generator's stem and head (models/networks.py:404-421: 7x7 behind ReflectionPad2d(3))
#undef OCTA_F32_CASE
    octa::set_error("octa_conv2d_f32_nchw: kernel size %d with stride %d is not instantiated (1/1, 3/1, 3/2, 4/1, 7/1)", K, stride);
    return -2;
}

// 2x2 stride-2 transposed convolution (torch.nn.ConvTranspose2d(Cin, Cout, 2, 2, bias=False): DynUNet's upsampling, MONAI
// UnetUpBlock.transp_conv) in ONE launch: d_wp = the weights packed [Cin][4][Cout] (tap 2 a + b, output channel innermost),
// d_y [N][Cout][2H][2W], y(co, 2 y + a, 2 x + b) = sum_ci x(ci, y, x) w(ci, co, a, b). Four 1x1 launches of octa_conv2d_f32_nchw with
// osc = 2 give the same numbers (same products, same summation order) with 4-byte stores two pixels apart.
extern "C" int octa_convtranspose2x2_f32_nchw(octa_ctx *ctx, const float *d_x, const float *d_wp, float *d_y, int N, int Cin, int H, int W, int Cout, void *stream_) {
    if (!ctx || !d_x || !d_wp || !d_y || N <= 0 || Cin <= 0 || H <= 0 || W <= 0 || Cout <= 0) { octa::set_error("octa_convtranspose2x2_f32_nchw: bad arguments"); return -2; }
    if ((long)Cin * H * W >= (1L << 31)) { octa::set_error("octa_convtranspose2x2_f32_nchw: one image of the input exceeds 2^31 elements"); return -2; }
    if (((uintptr_t)d_y & 7) != 0) { octa::set_error("octa_convtranspose2x2_f32_nchw: the output must be 8-byte aligned"); return -2; }
    hipStream_t stream = (hipStream_t)stream_;
    OCTA_HIP_CHECK(hipSetDevice(ctx->device));
    const float *zero = zero_word(ctx);
    if (!zero) return -1;
    return launch_f32<1, 1, 2, true>(zero, d_x, d_wp, nullptr, d_y, N, Cin, H, W, Cout, 4 * Cout, H, W, 0, 2, 0, 0, stream);
}
