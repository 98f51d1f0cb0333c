// conv.hip -- 3x3 convolution for NHWC bf16 activations on the gfx950 matrix cores (MFMA), forward and
// data-gradient; fp32 accumulation.
//
// Replaces the 3x3 convolutions of DynUNet's UnetBasicBlock / UnetUpBlock (MONAI, imported at
// models/networks.py:6; architecture in SURVEY.md a18: filters [32,64,128,256,512], strides [1,2,2,2,1],
// bias-free) in the training step of models/base_model_abc.py:152-167. torch/MIOpen runs these layers through
// NCHW<->NHWC transposes around its implicit-GEMM kernels (14 % of the step, profiles/r01_train_kernel_stats.csv);
// here the activations stay NHWC end to end.
//
// Implicit GEMM, one 256-thread workgroup per 8 x 32 output-pixel tile and BN output channels:
//   * per 32-channel slice of the input: the (8*st+2) x (32*st+2) halo tile is staged once in LDS
//     ([y][x][32 ch], 80-byte pixel pitch: a 16-lane ds_read_b128 group then touches all 64 banks once) and
//     reused by the nine taps; the weight slice [9][BN][32] is staged beside it (same pitch);
//   * a wave owns two tile rows (2 x 32 pixels = two MFMA M-blocks) x BN/32 N-blocks and issues
//     v_mfma_f32_32x32x16_bf16: A = pixels x 16 channels, B = 16 channels x 32 output channels, 8 bf16 per lane
//     (lane&31 = row / column, lane>>5 = which 8 of the 16 channels);
//   * epilogue: fp32 -> bf16 (round to nearest even), NHWC store; lanes 0-31 of a register hold the 32
//     consecutive output channels of one pixel (64 contiguous bytes).
// Data gradient of a stride-1 layer = the same kernel on flipped, transposed weights; of a stride-2 layer = the
// same kernel reading the output gradient through a virtual zero insertion (dil = 2), so no scatter is needed.
//
// Roofline: MFMA-bound on paper (2 * 9 * Cin * Cout flop per output pixel); algorithmic HBM bytes per launch =
// input + output activations once (bf16) + weights.

#include "common.h"

namespace {

typedef __attribute__((ext_vector_type(8))) short bf16x8;   // 8 bf16 = 4 VGPRs (MFMA A / B operand)
typedef __attribute__((ext_vector_type(16))) float f32x16;  // 32x32 accumulator fragment

constexpr int TH = 8, TW = 32;     // output pixels per workgroup tile
constexpr int KC = 32;             // input channels per LDS slice
constexpr int PITCH = 80;          // bytes per pixel / per weight row in LDS (64 B of data + 16 B pad)
constexpr int CONV_THREADS = 256;

__device__ __forceinline__ unsigned short f2bf(float f) {
    unsigned u = __float_as_uint(f);
    if ((u & 0x7fffffffu) > 0x7f800000u) return (unsigned short)((u >> 16) | 0x40);
    u += 0x7fffu + ((u >> 16) & 1u);
    return (unsigned short)(u >> 16);
}

// X: [N][H][W][Cin] (bf16 bits), Wt: [9][Cout][Cin], Y: [N][Ho][Wo][Cout].
// out(y, x) = sum_{r,s,ci} Xv(y*st + r - 1, x*st + s - 1, ci) * Wt[3r+s][co][ci], where the virtual input is
// Xv(yy, xx) = X[yy/dil][xx/dil] if 0 <= yy < H*dil, 0 <= xx < W*dil and yy, xx multiples of dil, else 0.
template <int BN, int ST>
__global__ void __launch_bounds__(CONV_THREADS)
conv3x3_nhwc_kernel(const unsigned short *__restrict__ X, const unsigned short *__restrict__ Wt, unsigned short *__restrict__ Y,
                    int H, int W, int Cin, int Ho, int Wo, int Cout, int dil, int tiles_x) {
    constexpr int IH = (TH - 1) * ST + 3, IW = (TW - 1) * ST + 3;  // halo tile
    constexpr int NB = BN / 32;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    unsigned char *s_in = smem;                          // [IH][IW] pixels x PITCH
    unsigned char *s_w = smem + IH * IW * PITCH;         // [9][BN] rows x PITCH
    const int tile = blockIdx.x, n = blockIdx.z, co0 = blockIdx.y * BN;
    const int ty0 = (tile / tiles_x) * TH, tx0 = (tile % tiles_x) * TW;
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    const int Hv = H * dil, Wv = W * dil;
    const int iy0 = ty0 * ST - 1, ix0 = tx0 * ST - 1;    // virtual coordinates of the halo tile's origin
    f32x16 acc[2][NB];
#pragma unroll
    for (int a = 0; a < 2; a++)
#pragma unroll
        for (int b = 0; b < NB; b++)
#pragma unroll
            for (int k = 0; k < 16; k++) acc[a][b][k] = 0.f;

    const int m = lane & 31, kg = lane >> 5;
    for (int c0 = 0; c0 < Cin; c0 += KC) {
        __syncthreads();
        // stage the input slice: 16-byte pieces (8 channels), 4 per pixel
        for (int i = threadIdx.x; i < IH * IW * 4; i += CONV_THREADS) {
            const int p = i >> 2, q = i & 3;
            const int yy = iy0 + p / IW, xx = ix0 + p % IW;
            uint4 v = make_uint4(0u, 0u, 0u, 0u);
            bool ok = yy >= 0 && yy < Hv && xx >= 0 && xx < Wv;
            if (ok && dil == 2) ok = !((yy | xx) & 1);
            if (ok) {
                const int sy = dil == 2 ? yy >> 1 : yy, sx = dil == 2 ? xx >> 1 : xx;
                v = *reinterpret_cast<const uint4 *>(X + (((size_t)n * H + sy) * W + sx) * Cin + c0 + q * 8);
            }
            *reinterpret_cast<uint4 *>(s_in + p * PITCH + q * 16) = v;
        }
        // stage the weight slice [9][BN][32]
        for (int i = threadIdx.x; i < 9 * BN * 4; i += CONV_THREADS) {
            const int row = i >> 2, q = i & 3;
            const int tap = row / BN, co = row % BN;
            const uint4 v = *reinterpret_cast<const uint4 *>(Wt + ((size_t)tap * Cout + co0 + co) * Cin + c0 + q * 8);
            *reinterpret_cast<uint4 *>(s_w + row * PITCH + q * 16) = v;
        }
        __syncthreads();
#pragma unroll
        for (int r = 0; r < 3; r++)
#pragma unroll
            for (int s = 0; s < 3; s++)
#pragma unroll
                for (int ks = 0; ks < KC / 16; ks++) {
                    bf16x8 a[2], b[NB];
#pragma unroll
                    for (int rr = 0; rr < 2; rr++) {
                        const int py = (2 * wv + rr) * ST + r, px = m * ST + s;
                        a[rr] = *reinterpret_cast<const bf16x8 *>(s_in + (py * IW + px) * PITCH + ks * 32 + kg * 16);
                    }
#pragma unroll
                    for (int nb = 0; nb < NB; nb++)
                        b[nb] = *reinterpret_cast<const bf16x8 *>(s_w + ((3 * r + s) * BN + nb * 32 + m) * PITCH + ks * 32 + kg * 16);
#pragma unroll
                    for (int rr = 0; rr < 2; rr++)
#pragma unroll
                        for (int nb = 0; nb < NB; nb++)
                            acc[rr][nb] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[rr], b[nb], acc[rr][nb], 0, 0, 0);
                }
    }
    // epilogue: D[row = pixel x][col = output channel]; row = (k&3) + 8*(k>>2) + 4*(lane>>5), col = lane&31
#pragma unroll
    for (int rr = 0; rr < 2; rr++) {
        const int oy = ty0 + 2 * wv + rr;
        if (oy >= Ho) continue;
#pragma unroll
        for (int nb = 0; nb < NB; nb++)
#pragma unroll
            for (int k = 0; k < 16; k++) {
                const int ox = tx0 + (k & 3) + 8 * (k >> 2) + 4 * kg;
                if (ox < Wo) Y[(((size_t)n * Ho + oy) * Wo + ox) * Cout + co0 + nb * 32 + m] = f2bf(acc[rr][nb][k]);
            }
    }
}

template <int BN, int ST>
int launch_conv(const unsigned short *X, const unsigned short *Wt, unsigned short *Y, int N, int H, int W, int Cin, int Ho, int Wo,
                int Cout, int dil, hipStream_t stream) {
    constexpr int IH = (TH - 1) * ST + 3, IW = (TW - 1) * ST + 3;
    const size_t lds = (size_t)IH * IW * PITCH + (size_t)9 * BN * PITCH;
    const int tiles_x = (Wo + TW - 1) / TW, tiles_y = (Ho + TH - 1) / TH;
    auto kern = conv3x3_nhwc_kernel<BN, ST>;
    OCTA_HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void *>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    dim3 grid((unsigned)(tiles_x * tiles_y), (unsigned)(Cout / BN), (unsigned)N);
    hipLaunchKernelGGL(kern, grid, dim3(CONV_THREADS), lds, stream, X, Wt, Y, H, W, Cin, Ho, Wo, Cout, dil, tiles_x);
    OCTA_HIP_CHECK(hipGetLastError());
    return 0;
}

}  // namespace

extern "C" int octa_conv3x3_nhwc_fwd(octa_ctx *ctx, const void *d_x, const void *d_w, void *d_y, int N, int H, int W, int Cin,
                                     int Cout, int stride, int in_dilation, void *stream_) {
    if (!ctx || !d_x || !d_w || !d_y) { octa::set_error("octa_conv3x3_nhwc_fwd: null pointer"); return -2; }
    if (N <= 0 || H <= 0 || W <= 0) { octa::set_error("octa_conv3x3_nhwc_fwd: bad shape"); return -2; }
    if (Cin % 32 || Cout % 32 || Cin <= 0 || Cout <= 0) { octa::set_error("octa_conv3x3_nhwc_fwd: Cin and Cout must be multiples of 32 (got %d, %d)", Cin, Cout); return -2; }
    if ((stride != 1 && stride != 2) || (in_dilation != 1 && in_dilation != 2) || (stride == 2 && in_dilation == 2)) {
        octa::set_error("octa_conv3x3_nhwc_fwd: stride %d / input dilation %d not supported", stride, in_dilation);
        return -2;
    }
    if (N > 65535) { octa::set_error("octa_conv3x3_nhwc_fwd: N > 65535"); return -2; }
    hipStream_t stream = (hipStream_t)stream_;
    OCTA_HIP_CHECK(hipSetDevice(ctx->device));
    const int Hv = H * in_dilation, Wv = W * in_dilation;
    const int Ho = (Hv + 2 - 3) / stride + 1, Wo = (Wv + 2 - 3) / stride + 1;
    const unsigned short *X = static_cast<const unsigned short *>(d_x), *Wt = static_cast<const unsigned short *>(d_w);
    unsigned short *Y = static_cast<unsigned short *>(d_y);
    const bool wide = (Cout % 64 == 0);
    if (stride == 1) return wide ? launch_conv<64, 1>(X, Wt, Y, N, H, W, Cin, Ho, Wo, Cout, in_dilation, stream)
                                 : launch_conv<32, 1>(X, Wt, Y, N, H, W, Cin, Ho, Wo, Cout, in_dilation, stream);
    return wide ? launch_conv<64, 2>(X, Wt, Y, N, H, W, Cin, Ho, Wo, Cout, in_dilation, stream)
                : launch_conv<32, 2>(X, Wt, Y, N, H, W, Cin, Ho, Wo, Cout, in_dilation, stream);
}
