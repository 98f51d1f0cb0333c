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
#include <cstdlib>

namespace {

typedef __attribute__((ext_vector_type(8))) short bf16x8;   // 8 bf16 = 4 VGPRs (MFMA A / B operand)
typedef __attribute__((ext_vector_type(16))) float f32x16;  // 32x32 accumulator fragment

constexpr int TH = 8, TW = 32;     // output pixels per workgroup tile
constexpr int KC = 32;             // input channels per LDS slice
constexpr int PITCH = 80;          // bytes per pixel / per weight row in LDS (64 B of data + 16 B pad)
constexpr int CONV_THREADS = 256;
#ifndef OCTA_DMA_PARTS
#define OCTA_DMA_PARTS 4
#endif

// 256 zero bytes in HBM: the source of padding pixels for loads that must not be branched around
const unsigned short *zero_page(octa_ctx *ctx) {
    if (!ctx->zero_page.p) {
        if (ctx->zero_page.reserve(256)) return nullptr;
        // hipMemset on device memory may return before the fill has run, and it runs on the NULL stream, which torch's (non-blocking) streams do
        // not wait for: the first DMA-staged launch of a fresh context could fetch its padding from an uncleared page. Wait for the device once.
        if (hipMemset(ctx->zero_page.p, 0, ctx->zero_page.cap) != hipSuccess || hipDeviceSynchronize() != hipSuccess) { octa::set_error("conv: zero page memset failed"); ctx->zero_page.release(); return nullptr; }
    }
    return ctx->zero_page.as<unsigned short>();
}

__device__ __forceinline__ unsigned short f2bf(float f) { return octa_f2bf(f); }

// X: [N][H][W][Cin] (bf16 bits), Wt: [9][Cout][Cin], Y: [N][Ho][Wo][Cout].
// out(y, x) = sum_{r,s,ci} Xv(y*st + r - 1, x*st + s - 1, ci) * Wt[3r+s][co][ci], where the virtual input is
// Xv(yy, xx) = X[yy/dil][xx/dil] if 0 <= yy < H*dil, 0 <= xx < W*dil and yy, xx multiples of dil, else 0.
// EXTRA = false compiles the plain kernel; true adds normalise-on-load and the statistics epilogue (both measured
// slower in the U-Net step, kept for experiments -- as a template flag they cost the plain kernel nothing).
// ---- packed weights: SLICE-MAJOR storage (round 5) ------------------------------------------------------------------------------
// Every kernel below reads its weights 16 input channels at a time (one K-step of v_mfma_f32_32x32x16_bf16) for all taps and a block of
// output channels. Rounds 1-4 kept them tap-major, [KK][Cout][Cin] with Cin fastest: a 16-channel slice of a row is then 32 bytes out
// of Cin * 2, every DMA lane group touches its own cache line for 32 useful bytes, and the vector memory path moved twice to four times
// the bytes the LDS received (measured: the same loads pointed at one contiguous range took 152^2 512->512 from 929 to 1114 TFLOP/s, no
// loads in the loop 1815). The storage order is now [Cin/16][KK][Cout][16]: the slice (s, all taps, Cout block) of a workgroup is KK
// contiguous runs of BN * 32 bytes. Element (tap t, output channel co, input channel ci) lives at wt_off(): callers keep passing the
// nominal shape [KK][Cout][Cin]; mfma_conv.slice_major() / octa_pack_conv_weights produce the order.
__host__ __device__ __forceinline__ size_t wt_off(int t, int co, int ci, int KK, int Cout) {
    return (((size_t)(ci >> 4) * KK + t) * Cout + co) * 16 + (ci & 15);
}

template <int BN, int ST, bool EXTRA, bool MASKED>
__global__ void __launch_bounds__(CONV_THREADS)
conv3x3_nhwc_kernel(const unsigned short *__restrict__ X, const unsigned short *__restrict__ X2, int C1,
                    const unsigned short *__restrict__ Wt, unsigned short *__restrict__ Y, unsigned short *__restrict__ Y2, int CY1,
                    int H, int W, int Cin, int Ho, int Wo, int Cout, int dil, int tiles_x, int tap_mask, int osc, int ooy, int oox,
                    const float *__restrict__ sc1, const float *__restrict__ sh1, const float *__restrict__ sc2,
                    const float *__restrict__ sh2, float slope, float *__restrict__ part) {
    constexpr int IH = (TH - 1) * ST + 3, IW = (TW - 1) * ST + 3;  // halo tile
    constexpr int NB = BN / 32;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    unsigned char *s_in = smem;                          // [IH][IW] pixels x PITCH
    unsigned char *s_w = smem + IH * IW * PITCH;         // [9][BN] rows x PITCH
    float *s_ss = reinterpret_cast<float *>(s_w + 9 * BN * PITCH);  // [2][KC] scale / shift of the current input slice
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
        // virtual concatenation: channels [0, C1) come from X, [C1, Cin) from X2 (C1 is a multiple of KC)
        const unsigned short *Xs = c0 < C1 ? X : X2;
        const int cs = c0 < C1 ? C1 : Cin - C1, cb = c0 < C1 ? c0 : c0 - C1;
        // Normalise-on-load: an input that is the RAW output of an earlier convolution is turned into
        // lrelu(x * scale + shift) (InstanceNorm affine + LeakyReLU, per image and channel) while it is staged, rounded
        // to bf16 exactly like the materialised tensor would have been; padding stays zero.
        const float *scp = c0 < C1 ? sc1 : sc2, *shp = c0 < C1 ? sh1 : sh2;
        const bool xform = EXTRA && scp != nullptr;
        if (xform) {
            if (threadIdx.x < KC) s_ss[threadIdx.x] = scp[(size_t)n * cs + cb + threadIdx.x];
            else if (threadIdx.x < 2 * KC) s_ss[threadIdx.x] = shp[(size_t)n * cs + cb + threadIdx.x - KC];
            __syncthreads();
        }
        // stage the input slice: 16-byte pieces (8 channels), 4 per pixel
        for (int i = threadIdx.x; i < IH * IW * 4; i += CONV_THREADS) {
            const int p = i >> 2, q = i & 3;
            const int yy = iy0 + p / IW, xx = ix0 + p % IW;
            uint4 v = make_uint4(0u, 0u, 0u, 0u);
            bool ok = yy >= 0 && yy < Hv && xx >= 0 && xx < Wv;
            if (ok && dil == 2) ok = !((yy | xx) & 1);
            if (ok) {
                const int sy = dil == 2 ? yy >> 1 : yy, sx = dil == 2 ? xx >> 1 : xx;
                v = *reinterpret_cast<const uint4 *>(Xs + (((size_t)n * H + sy) * W + sx) * cs + cb + q * 8);
                if (xform) {
                    unsigned u[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
                    for (int k = 0; k < 4; k++) {
                        float z0 = __uint_as_float(u[k] << 16) * s_ss[q * 8 + 2 * k] + s_ss[KC + q * 8 + 2 * k];
                        float z1 = __uint_as_float(u[k] & 0xffff0000u) * s_ss[q * 8 + 2 * k + 1] + s_ss[KC + q * 8 + 2 * k + 1];
                        z0 = z0 > 0.f ? z0 : z0 * slope;
                        z1 = z1 > 0.f ? z1 : z1 * slope;
                        u[k] = octa_pack_bf16x2(z0, z1);
                    }
                    v = make_uint4(u[0], u[1], u[2], u[3]);
                }
            }
            *reinterpret_cast<uint4 *>(s_in + p * PITCH + q * 16) = v;
        }
        // stage the weight slice [9][BN][32]
        for (int i = threadIdx.x; i < 9 * BN * 4; i += CONV_THREADS) {
            const int row = i >> 2, q = i & 3;
            const int tap = row / BN, co = row % BN;
            if (MASKED && !((tap_mask >> tap) & 1)) continue;   // unused taps are neither staged nor multiplied
            const uint4 v = *reinterpret_cast<const uint4 *>(Wt + wt_off(tap, co0 + co, c0 + q * 8, 9, Cout));
            *reinterpret_cast<uint4 *>(s_w + row * PITCH + q * 16) = v;
        }
        __syncthreads();
#pragma unroll
        for (int r = 0; r < 3; r++)
#pragma unroll
            for (int s = 0; s < 3; s++) {
                if (MASKED && !((tap_mask >> (3 * r + s)) & 1)) continue;   // taps whose weights are structurally zero
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
    }
    // epilogue: D[row = pixel x][col = output channel]; row = (k&3) + 8*(k>>2) + 4*(lane>>5), col = lane&31.
    // Split output: channels [0, CY1) go to Y, the rest to Y2 (a BN block never straddles CY1).
    // Scattered output: result pixel (oy, ox) is stored at (oy * osc + ooy, ox * osc + oox) of an image osc times
    // larger (osc = 2: one parity class of a zero-insertion-free stride-2 data gradient / 2x2 transposed convolution).
    unsigned short *Yo = co0 < CY1 ? Y : Y2;
    const int ys = co0 < CY1 ? CY1 : Cout - CY1, yb = co0 < CY1 ? co0 : co0 - CY1;
    if (!EXTRA) { osc = 1; ooy = 0; oox = 0; }   // the scattered store is compiled into the EXTRA variant only
    const int HoF = Ho * osc, WoF = Wo * osc;
    if (!EXTRA) {
        // Coalesced store: the tile goes through LDS ([pixel][BN channels], 16-byte padded rows) so that consecutive
        // threads write consecutive 16-byte pieces -- whole 64/128-byte pixel rows, whole tile rows back to back --
        // instead of 2-byte elements 64 bytes apart.
        constexpr int OP = BN * 2 + 16;                 // bytes per pixel row in LDS
        unsigned char *s_out = smem;                    // [TH*TW] rows; the operand tiles are dead now
        __syncthreads();
#pragma unroll
        for (int rr = 0; rr < 2; rr++)
#pragma unroll
            for (int nb = 0; nb < NB; nb++)
#pragma unroll
                for (int k = 0; k < 16; k++) {
                    const int px = (k & 3) + 8 * (k >> 2) + 4 * kg;
                    *reinterpret_cast<unsigned short *>(s_out + ((2 * wv + rr) * TW + px) * OP + (nb * 32 + m) * 2) = f2bf(acc[rr][nb][k]);
                }
        __syncthreads();
        constexpr int PIECES = BN / 8;                  // 16-byte pieces per pixel
        for (int i = threadIdx.x; i < TH * TW * PIECES; i += CONV_THREADS) {
            const int p = i / PIECES, q = i % PIECES;
            const int oy = ty0 + p / TW, ox = tx0 + p % TW;
            if (oy < Ho && ox < Wo)
                *reinterpret_cast<uint4 *>(Yo + (((size_t)n * Ho + oy) * Wo + ox) * ys + yb + q * 8) = *reinterpret_cast<const uint4 *>(s_out + p * OP + q * 16);
        }
    } else {
#pragma unroll
    for (int rr = 0; rr < 2; rr++) {
        const int oy = ty0 + 2 * wv + rr;
        if (oy >= Ho) continue;
#pragma unroll
        for (int nb = 0; nb < NB; nb++)
#pragma unroll
            for (int k = 0; k < 16; k++) {
                const int ox = tx0 + (k & 3) + 8 * (k >> 2) + 4 * kg;
                if (ox < Wo) Yo[(((size_t)n * HoF + oy * osc + ooy) * WoF + ox * osc + oox) * ys + yb + nb * 32 + m] = f2bf(acc[rr][nb][k]);
            }
    }
    }
    // InstanceNorm statistics of the layer that follows, for free: per tile and output channel the sum and the sum of
    // squares of the bf16-ROUNDED results (what the norm kernels would read back) -> part[n][tile][Cout][2]
    if (EXTRA && part) {
        float s1[NB], s2[NB];
#pragma unroll
        for (int nb = 0; nb < NB; nb++) { s1[nb] = 0.f; s2[nb] = 0.f; }
#pragma unroll
        for (int rr = 0; rr < 2; rr++) {
            const int oy = ty0 + 2 * wv + rr;
#pragma unroll
            for (int nb = 0; nb < NB; nb++)
#pragma unroll
                for (int k = 0; k < 16; k++) {
                    const int ox = tx0 + (k & 3) + 8 * (k >> 2) + 4 * kg;
                    if (oy < Ho && ox < Wo) {
                        const float v = __uint_as_float((unsigned)f2bf(acc[rr][nb][k]) << 16);
                        s1[nb] += v; s2[nb] += v * v;
                    }
                }
        }
        float *s_red = reinterpret_cast<float *>(smem);   // [4 waves][BN][2]; the operand tiles are dead now
        __syncthreads();
#pragma unroll
        for (int nb = 0; nb < NB; nb++) {
            const float a1 = s1[nb] + __shfl_xor(s1[nb], 32, 64), a2 = s2[nb] + __shfl_xor(s2[nb], 32, 64);
            if (kg == 0) { s_red[(wv * BN + nb * 32 + m) * 2] = a1; s_red[(wv * BN + nb * 32 + m) * 2 + 1] = a2; }
        }
        __syncthreads();
        if (threadIdx.x < BN) {
            float a1 = 0.f, a2 = 0.f;
#pragma unroll
            for (int w4 = 0; w4 < 4; w4++) { a1 += s_red[(w4 * BN + threadIdx.x) * 2]; a2 += s_red[(w4 * BN + threadIdx.x) * 2 + 1]; }
            float *dst = part + (((size_t)n * gridDim.x + tile) * Cout + co0 + threadIdx.x) * 2;
            dst[0] = a1; dst[1] = a2;
        }
    }
}

template <int BN, int ST, bool EXTRA, bool MASKED>
int launch_conv_impl(const unsigned short *X, const unsigned short *X2, int C1, const unsigned short *Wt, unsigned short *Y, unsigned short *Y2,
                int CY1, int N, int H, int W, int Cin, int Ho, int Wo, int Cout, int dil, int tap_mask, int osc, int ooy, int oox,
                const float *sc1, const float *sh1, const float *sc2, const float *sh2, float slope, float *part, hipStream_t stream) {
    constexpr int IH = (TH - 1) * ST + 3, IW = (TW - 1) * ST + 3;
    const size_t lds = (size_t)IH * IW * PITCH + (size_t)9 * BN * PITCH + 2 * KC * sizeof(float);
    const int tiles_x = (Wo + TW - 1) / TW, tiles_y = (Ho + TH - 1) / TH;
    auto kern = conv3x3_nhwc_kernel<BN, ST, EXTRA, MASKED>;
    OCTA_HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void *>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    dim3 grid((unsigned)(tiles_x * tiles_y), (unsigned)(Cout / BN), (unsigned)N);
    hipLaunchKernelGGL(kern, grid, dim3(CONV_THREADS), lds, stream, X, X2, C1, Wt, Y, Y2, CY1, H, W, Cin, Ho, Wo, Cout, dil, tiles_x, tap_mask, osc, ooy, oox,
                       sc1, sh1, sc2, sh2, slope, part);
    OCTA_HIP_CHECK(hipGetLastError());
    return 0;
}

// ---- stride-1 kernel with asynchronous global -> LDS staging (global_load_lds, 16 bytes per lane) -------------------------
// Same tile, same MFMA schedule and same epilogue as conv3x3_nhwc_kernel<BN, 1, false, false>, but the operand slices
// never pass through VGPRs: every wave issues global_load_lds_dwordx4 for the NEXT slice into the other LDS buffer, then
// multiplies the current one, so the HBM/L2 latency of a slice hides under the 72 (BN = 64) MFMAs of the previous slice
// and the ds_write_b128 pass (~79 B/clk/CU) disappears. One barrier per slice.
// The DMA writes LDS lane-linearly (wave-uniform base + lane * 16), so the LDS image is an unpadded array of 16-byte
// pieces [row][PP] (row = halo pixel or weight row, PP = KCV / 8 pieces) and the bank spread comes from an XOR swizzle
// applied to BOTH the source address and the read: piece q of row p lives in slot p * PP + (q ^ swz(p)), swz(p) =
// (p >> 2) & 3 for PP = 4: the 16 rows of one ds_read_b128 lane group then cover all 64 banks once.
// Out-of-image halo pixels (padding, the zero rows of a virtual zero insertion) are fetched from a 16-byte zero page.
template <int PP> __device__ __forceinline__ int glds_swz(int p) { return PP == 4 ? (p >> 2) & 3 : (p >> 3) & 1; }

__device__ __forceinline__ void glds16(const void *gsrc, unsigned char *lds_wave_base) {
    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void *)gsrc,
                                     (__attribute__((address_space(3))) void *)lds_wave_base, 16, 0, 0);
}

// ST = 2 (stride-2 layers): the halo is (2*TH+1) x (2*TW+1) pixels and a wave's 32 output columns read every OTHER halo
// column, so the LDS image keeps the even and the odd halo columns of a row in two planes ([row][parity][33 pixels]):
// the 32 lanes of a tap then read 32 CONSECUTIVE LDS pixels again and the swizzle stays conflict-free. One workgroup
// per CU (2 x 54 KB of LDS), the loads of the next slice still overlap the MFMAs of the current one.
template <int BN, int KCV, bool STATS, bool MASKED, int ST, int KS, int THT>
__global__ void __launch_bounds__(CONV_THREADS, (THT == 16 ? 2 : 1))
conv3x3_nhwc_glds_kernel(const unsigned short *__restrict__ X, const unsigned short *__restrict__ X2, int C1,
                         const unsigned short *__restrict__ Wt, unsigned short *__restrict__ Y, unsigned short *__restrict__ Y2, int CY1,
                         int H, int W, int Cin, int Ho, int Wo, int Cout, int dil, int tiles_x, const unsigned short *__restrict__ zero16,
                         float *__restrict__ part, int tap_mask, int osc, int ooy, int oox, int pad, const unsigned short *__restrict__ R, int reflect,
                         int nslot) {
    constexpr int RPW = THT / 4;                           // tile rows per wave (THT = 8: two, THT = 16: four)
    constexpr int IH = (THT - 1) * ST + KS, IW = (TW - 1) * ST + KS;   // KS x KS taps (3: the U-Net / generator layers, 4: the PatchGAN)
    constexpr int PW = (IW + ST - 1) / ST;                 // pixels per LDS plane row (ST = 2: 33 even / 32 odd columns)
    constexpr int LPIX = IH * ST * PW;                      // pixels of the LDS image
    constexpr int PP = KCV / 8, NB = BN / 32;
    constexpr int IN_INSTR = (LPIX * PP + 63) / 64, W_INSTR = KS * KS * BN * PP / 64;
    constexpr int IN_BYTES = IN_INSTR * 1024, BUF = IN_BYTES + W_INSTR * 1024;
    constexpr int IN_PW = (IN_INSTR + 3) / 4, W_PW = (W_INSTR + 3) / 4;   // wave-instructions per wave and slice
    static_assert((KS * KS * BN * PP) % 64 == 0, "weight slice must be whole wave-instructions");
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const int tile = blockIdx.x, n = blockIdx.z, co0 = blockIdx.y * BN;
    const int ty0 = (tile / tiles_x) * THT, tx0 = (tile % tiles_x) * TW;
    const int lane = threadIdx.x & 63, wv = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int Hv = H * dil, Wv = W * dil;
    const int iy0 = ty0 * ST - pad, ix0 = tx0 * ST - pad;
    const int m = lane & 31, kg = lane >> 5;

    // per-lane sources of this wave's DMA slots (the same pixels / weight rows for every channel slice)
    int in_src[IN_PW], w_src[W_PW];
#pragma unroll
    for (int i = 0; i < IN_PW; i++) {
        const int slot = (wv + 4 * i) * 64 + lane, p = slot / PP, q = (slot % PP) ^ glds_swz<PP>(p);
        // LDS pixel p = (halo row * ST + column parity) * PW + column / ST
        const int prow = p / PW, hy = prow / ST, hx = (p % PW) * ST + prow % ST;
        int yy = iy0 + hy, xx = ix0 + hx;
        if (reflect) {      // nn.ReflectionPad2d(pad) in front of the convolution: the halo mirrors the image instead of padding it with zeros
            yy = yy < 0 ? -yy : (yy >= Hv ? 2 * (Hv - 1) - yy : yy);
            xx = xx < 0 ? -xx : (xx >= Wv ? 2 * (Wv - 1) - xx : xx);
        }
        bool ok = p < LPIX && hx < IW && yy >= 0 && yy < Hv && xx >= 0 && xx < Wv;
        if (ok && dil == 2) ok = !((yy | xx) & 1);
        const int sy = dil == 2 ? yy >> 1 : yy, sx = dil == 2 ? xx >> 1 : xx;
        in_src[i] = ok ? ((sy * W + sx) << 2) | q : -1;
    }
#pragma unroll
    for (int i = 0; i < W_PW; i++) {
        const int slot = (wv + 4 * i) * 64 + lane, rw = slot / PP, q = (slot % PP) ^ glds_swz<PP>(rw);
        w_src[i] = (int)wt_off(rw / BN, co0 + rw % BN, q * 8, KS * KS, Cout);      // + the slice: c0 / 16 * (KK * Cout * 16) elements
    }
    // `part` / `nparts`: issue the wave's DMA instructions i with i % nparts == part (all of them for nparts = 1) -- the main loop spreads
    // them between its MFMA groups, so that their issue (M0 write, address, the VMEM slot: 60-180 cycles each) runs in the matrix
    // pipe's shadow instead of in front of the slice's first MFMA
    auto issue = [&](int c0, unsigned char *buf, int part = 0, int nparts = 1) {
        const unsigned short *Xs = c0 < C1 ? X : X2;
        const int cs = c0 < C1 ? C1 : Cin - C1, cb = c0 < C1 ? c0 : c0 - C1;
        const unsigned short *img = Xs + (size_t)n * H * W * cs + cb;
#pragma unroll
        for (int i = 0; i < IN_PW; i++) {
            const int j = wv + 4 * i;
            if (i % nparts == part && j < IN_INSTR) {
                const unsigned short *src = in_src[i] < 0 ? zero16 : img + (size_t)(in_src[i] >> 2) * cs + (in_src[i] & 3) * 8;
                glds16(src, buf + j * 1024);
            }
        }
#pragma unroll
        for (int i = 0; i < W_PW; i++) {
            const int j = wv + 4 * i;
            // (a wave-instruction of weights covers 64 / PP rows of ONE tap, BN being a multiple of that: masked taps are not fetched)
            if ((IN_PW + i) % nparts == part && j < W_INSTR && (!MASKED || ((tap_mask >> (j * (64 / PP) / BN)) & 1))) glds16(Wt + w_src[i] + (size_t)c0 * (KS * KS) * Cout, buf + IN_BYTES + j * 1024);
        }
    };

    f32x16 acc[RPW][NB];
#pragma unroll
    for (int a = 0; a < RPW; a++)
#pragma unroll
        for (int b = 0; b < NB; b++)
#pragma unroll
            for (int k = 0; k < 16; k++) acc[a][b][k] = 0.f;

    const int nsl = Cin / KCV;
    issue(0, smem);
    for (int k = 0; k < nsl; k++) {
        __syncthreads();   // slice k has landed (every wave drained its own DMA queue first); buffer (k+1)&1 is free again
        constexpr bool SPREAD = !MASKED && ST == 1;           // the DMA of the next slice goes out between this slice's MFMA groups
        const bool more = k + 1 < nsl;
        unsigned char *nbuf = smem + ((k + 1) & 1) * BUF;
        if (!SPREAD && more) issue((k + 1) * KCV, nbuf);
        const unsigned char *s_in = smem + (k & 1) * BUF, *s_w = s_in + IN_BYTES;
        if constexpr (!MASKED && ST == 1 && RPW == 2) {
            // column-major tap order: the RPW + KS - 1 halo rows a wave's RPW tile rows touch at tap column s are read ONCE and serve all KS
            // tap rows (tile row rr at tap row r reads halo row rr + r): RPW + KS - 1 + KS * NB operand reads per KS * RPW * NB MFMAs --
            // 10 per 12 at RPW = 2, 12 per 24 at RPW = 4. The reads of tap column s + 1 are issued BEFORE the MFMAs of column s (two
            // operand sets in registers; the compiler's own schedule kept four operands and waited for LDS every 2-4 MFMAs).
            constexpr int PH = (KCV / 16) * KS, NA = RPW + KS - 1;
            bf16x8 a[2][NA], b[2][NB];
            auto load_a = [&](int ph, bf16x8 *ad) {
                const int ks = ph / KS, sc = ph % KS, qa = ks * 2 + kg;
#pragma unroll
                for (int hr = 0; hr < NA; hr++) {
                    const int p = (RPW * wv + hr) * PW + m + sc;
                    ad[hr] = *reinterpret_cast<const bf16x8 *>(s_in + (p * PP + (qa ^ glds_swz<PP>(p))) * 16);
                }
            };
            auto load_b = [&](int t, bf16x8 *bd) {           // step t = (phase, tap row r)
                const int ph = t / KS, r = t % KS, ks = ph / KS, sc = ph % KS, qa = ks * 2 + kg;
#pragma unroll
                for (int nb = 0; nb < NB; nb++) {
                    const int rw = (KS * r + sc) * BN + nb * 32 + m;
                    bd[nb] = *reinterpret_cast<const bf16x8 *>(s_w + (rw * PP + (qa ^ glds_swz<PP>(rw))) * 16);
                }
            };
            load_a(0, a[0]);
            load_b(0, b[0]);
#pragma unroll
            for (int t = 0; t < PH * KS; t++) {
                const int ph = t / KS, r = t % KS;
                if (t + 1 < PH * KS) load_b(t + 1, b[(t + 1) & 1]);
                if (r == 0 && ph + 1 < PH) load_a(ph + 1, a[(ph + 1) & 1]);
                __builtin_amdgcn_sched_barrier(0);
                constexpr int NPARTS = OCTA_DMA_PARTS;   // steps of the slice that carry DMA issues: the data must land before the next barrier
#pragma unroll
                for (int rr = 0; rr < RPW; rr++)
#pragma unroll
                    for (int nb = 0; nb < NB; nb++) {
                        acc[rr][nb] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[ph & 1][rr + r], b[t & 1][nb], acc[rr][nb], 0, 0, 0);
                        if (rr == 0 && nb == 0 && t < NPARTS) {
                            __builtin_amdgcn_sched_barrier(0);
                            if (more) issue((k + 1) * KCV, nbuf, t, NPARTS);
                            __builtin_amdgcn_sched_barrier(0);
                        }
                    }
                __builtin_amdgcn_sched_barrier(0);
            }
        } else if constexpr (!MASKED && ST == 1) {
            // 16-row tiles (RPW = 4): the same column-major order without the second operand set -- two sets on top of 128 accumulator
            // registers spill at two waves per SIMD (measured: 304^2 128->128 0.122 -> 0.144 ms with 6-12 spilled registers)
#pragma unroll
            for (int ks = 0; ks < KCV / 16; ks++)
#pragma unroll
                for (int s = 0; s < KS; s++) {
                    const int qa = ks * 2 + kg;
                    bf16x8 a[RPW + KS - 1];
#pragma unroll
                    for (int hr = 0; hr < RPW + KS - 1; hr++) {
                        const int p = (RPW * wv + hr) * PW + m + s;
                        a[hr] = *reinterpret_cast<const bf16x8 *>(s_in + (p * PP + (qa ^ glds_swz<PP>(p))) * 16);
                    }
#pragma unroll
                    for (int r = 0; r < KS; r++) {
                        bf16x8 b[NB];
#pragma unroll
                        for (int nb = 0; nb < NB; nb++) {
                            const int rw = (KS * r + s) * BN + nb * 32 + m;
                            b[nb] = *reinterpret_cast<const bf16x8 *>(s_w + (rw * PP + (qa ^ glds_swz<PP>(rw))) * 16);
                        }
                        constexpr int NPARTS = OCTA_DMA_PARTS;
                        const int t = (ks * KS + s) * KS + r;
#pragma unroll
                        for (int rr = 0; rr < RPW; rr++)
#pragma unroll
                            for (int nb = 0; nb < NB; nb++) {
                                acc[rr][nb] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[rr + r], b[nb], acc[rr][nb], 0, 0, 0);
                                if (rr == 0 && nb == 0 && t < NPARTS) {
                                    __builtin_amdgcn_sched_barrier(0);
                                    if (more) issue((k + 1) * KCV, nbuf, t, NPARTS);
                                    __builtin_amdgcn_sched_barrier(0);
                                }
                            }
                    }
                }
        } else
#pragma unroll
        for (int r = 0; r < KS; r++)
#pragma unroll
            for (int s = 0; s < KS; s++)
#pragma unroll
                for (int ks = 0; ks < KCV / 16; ks++) {
                    if (MASKED && !((tap_mask >> (KS * r + s)) & 1)) continue;   // taps whose weights are structurally zero
                    const int qa = ks * 2 + kg;
                    bf16x8 a[RPW], b[NB];
#pragma unroll
                    for (int rr = 0; rr < RPW; rr++) {
                        const int hy = (RPW * wv + rr) * ST + r;                       // halo row; halo column = m * ST + s
                        const int p = (hy * ST + (ST == 2 ? (s & 1) : 0)) * PW + m + (ST == 2 ? (s >> 1) : s);
                        a[rr] = *reinterpret_cast<const bf16x8 *>(s_in + (p * PP + (qa ^ glds_swz<PP>(p))) * 16);
                    }
#pragma unroll
                    for (int nb = 0; nb < NB; nb++) {
                        const int rw = (KS * r + s) * BN + nb * 32 + m;
                        b[nb] = *reinterpret_cast<const bf16x8 *>(s_w + (rw * PP + (qa ^ glds_swz<PP>(rw))) * 16);
                    }
#pragma unroll
                    for (int rr = 0; rr < RPW; rr++)
#pragma unroll
                        for (int nb = 0; nb < NB; nb++)
                            acc[rr][nb] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[rr], b[nb], acc[rr][nb], 0, 0, 0);
                }
    }
    if (!MASKED) { osc = 1; ooy = 0; oox = 0; }   // the scattered store is compiled into the masked variant only (its only user)
    // epilogue of the plain kernel: tile through LDS, 16-byte coalesced NHWC stores
    unsigned short *Yo = co0 < CY1 ? Y : Y2;
    const int ys = co0 < CY1 ? CY1 : Cout - CY1, yb = co0 < CY1 ? co0 : co0 - CY1;
    constexpr int OP = BN * 2 + 16;
    unsigned char *s_out = smem;
    // InstanceNorm statistics of the layer that follows, slot form (round 5; nslot > 0): the sum and the sum of squares of the bf16-ROUNDED
    // results per output channel ride in the conversion loop -- a lane holds ONE channel (m) of 16 pixels per fragment, v_cvt_pk_bf16_f32
    // rounds two of them at once and two v_dot2c_f32_bf16 (pair . (1, 1), pair . pair) add both to the running sums: 1.5 VALU instructions
    // per value where the per-tile form below spends ~5 (conversion again, range masks, multiply-add). Results of an edge tile that lie
    // outside the image are never stored: they are cleared first, so the sums need no masks.
    const bool stat_slots = STATS && nslot > 0;
    float q1[NB], q2[NB];
#pragma unroll
    for (int nb = 0; nb < NB; nb++) { q1[nb] = 0.f; q2[nb] = 0.f; }
    if (stat_slots && !(ty0 + THT <= Ho && tx0 + TW <= Wo)) {
#pragma unroll
        for (int rr = 0; rr < RPW; rr++) {
            const bool row_in = ty0 + RPW * wv + rr < Ho;
#pragma unroll
            for (int nb = 0; nb < NB; nb++)
#pragma unroll
                for (int k = 0; k < 16; k++)
                    if (!(row_in && tx0 + (k & 3) + 8 * (k >> 2) + 4 * kg < Wo)) acc[rr][nb][k] = 0.f;
        }
    }
    __syncthreads();
#pragma unroll
    for (int rr = 0; rr < RPW; rr++)
#pragma unroll
        for (int nb = 0; nb < NB; nb++)
#pragma unroll
            for (int k = 0; k < 16; k += 2) {
                const int px = (k & 3) + 8 * (k >> 2) + 4 * kg;          // k even: pixels px and px + 1 of the lane's channel
                const unsigned pk = octa_pack_bf16x2(acc[rr][nb][k], acc[rr][nb][k + 1]);
                unsigned char *d = s_out + ((RPW * wv + rr) * TW + px) * OP + (nb * 32 + m) * 2;
                *reinterpret_cast<unsigned short *>(d) = (unsigned short)(pk & 0xffffu);
                *reinterpret_cast<unsigned short *>(d + OP) = (unsigned short)(pk >> 16);
                if (stat_slots) {
                    typedef __attribute__((ext_vector_type(2))) __bf16 bf2;
                    const bf2 v = __builtin_bit_cast(bf2, pk), one = __builtin_bit_cast(bf2, 0x3f803f80u);
                    q1[nb] = __builtin_amdgcn_fdot2_f32_bf16(v, one, q1[nb], false);
                    q2[nb] = __builtin_amdgcn_fdot2_f32_bf16(v, v, q2[nb], false);
                }
            }
    // the four waves' sums meet in LDS BEHIND the output tile (free since the last slice was consumed), so that the barrier the tile needs
    // anyway also publishes them: no barrier of its own
    float *s_red = reinterpret_cast<float *>(smem + ((THT * TW * OP + 15) / 16) * 16);   // [4 waves][BN][2]
    if (stat_slots) {
#pragma unroll
        for (int nb = 0; nb < NB; nb++) {
            const float a1 = q1[nb] + __shfl_xor(q1[nb], 32, 64), a2 = q2[nb] + __shfl_xor(q2[nb], 32, 64);
            if (kg == 0) { s_red[(wv * BN + nb * 32 + m) * 2] = a1; s_red[(wv * BN + nb * 32 + m) * 2 + 1] = a2; }
        }
    }
    __syncthreads();
    if (stat_slots && threadIdx.x < BN) {
        // ONE pair of double atomics per channel and tile goes to slot (tile % nslot) of sums[nslot][N][Cout][2] (pre-zeroed by the caller):
        // the norm's apply pass adds the slots up (csrc/norm.hip). Spreading the tiles of an image over the slots keeps the chain of
        // same-address atomics short (5776 tiles per 1216^2 image -> 361 per address at 16 slots). Issued in front of the tile's stores.
        float a1 = 0.f, a2 = 0.f;
#pragma unroll
        for (int w4 = 0; w4 < 4; w4++) { a1 += s_red[(w4 * BN + threadIdx.x) * 2]; a2 += s_red[(w4 * BN + threadIdx.x) * 2 + 1]; }
        double *dst = reinterpret_cast<double *>(part) + (((size_t)(tile % nslot) * gridDim.z + n) * Cout + co0 + threadIdx.x) * 2;
        atomicAdd(dst, (double)a1);
        atomicAdd(dst + 1, (double)a2);
    }
    constexpr int PIECES = BN / 8;
    for (int i = threadIdx.x; i < THT * TW * PIECES; i += CONV_THREADS) {
        const int p = i / PIECES, q = i % PIECES;
        const int oy = ty0 + p / TW, ox = tx0 + p % TW;
        // scattered output (osc = 2): result pixel (oy, ox) lands at (oy * 2 + ooy, ox * 2 + oox) of an image twice as large --
        // one parity class of a zero-insertion-free stride-2 data gradient / 2x2 transposed convolution
        if (oy < Ho && ox < Wo) {
            const size_t off = (((size_t)n * Ho * osc + oy * osc + ooy) * (Wo * osc) + ox * osc + oox) * ys + yb + q * 8;
            uint4 v = *reinterpret_cast<const uint4 *>(s_out + p * OP + q * 16);
            if (R) {
                // residual: the other gradient of a tensor with two consumers (a U-Net skip connection), added to the bf16-rounded
                // result in fp32 and rounded again -- bit for bit what a separate bf16 tensor addition would store
                const uint4 r = *reinterpret_cast<const uint4 *>(R + off);
                unsigned a[4] = {v.x, v.y, v.z, v.w};
                const unsigned b[4] = {r.x, r.y, r.z, r.w};
#pragma unroll
                for (int k = 0; k < 4; k++)
                    a[k] = octa_pack_bf16x2(__uint_as_float(a[k] << 16) + __uint_as_float(b[k] << 16),
                                            __uint_as_float(a[k] & 0xffff0000u) + __uint_as_float(b[k] & 0xffff0000u));
                v = make_uint4(a[0], a[1], a[2], a[3]);
            }
            *reinterpret_cast<uint4 *>(Yo + off) = v;
        }
    }
    // InstanceNorm statistics of the layer that follows: per tile and output channel the sum and the sum of squares of the
    // bf16-ROUNDED results (what a statistics pass would read back) -> part[n][tile][Cout][2]; saves that pass over Y.
    if (STATS && nslot == 0) {
        float s1[NB], s2[NB];
#pragma unroll
        for (int nb = 0; nb < NB; nb++) { s1[nb] = 0.f; s2[nb] = 0.f; }
#pragma unroll
        for (int rr = 0; rr < RPW; rr++) {
            const int oy = ty0 + RPW * wv + rr;
#pragma unroll
            for (int nb = 0; nb < NB; nb++)
#pragma unroll
                for (int k = 0; k < 16; k++) {
                    const int ox = tx0 + (k & 3) + 8 * (k >> 2) + 4 * kg;
                    const float v = (oy < Ho && ox < Wo) ? __uint_as_float((unsigned)f2bf(acc[rr][nb][k]) << 16) : 0.f;
                    s1[nb] += v; s2[nb] += v * v;
                }
        }
        float *s_red = reinterpret_cast<float *>(smem);   // [4 waves][BN][2]
        __syncthreads();                                  // the output tile has been read out of LDS
#pragma unroll
        for (int nb = 0; nb < NB; nb++) {
            const float a1 = s1[nb] + __shfl_xor(s1[nb], 32, 64), a2 = s2[nb] + __shfl_xor(s2[nb], 32, 64);
            if (kg == 0) { s_red[(wv * BN + nb * 32 + m) * 2] = a1; s_red[(wv * BN + nb * 32 + m) * 2 + 1] = a2; }
        }
        __syncthreads();
        if (THT == 8) {
            if (threadIdx.x < BN) {
                float a1 = 0.f, a2 = 0.f;
#pragma unroll
                for (int w4 = 0; w4 < 4; w4++) { a1 += s_red[(w4 * BN + threadIdx.x) * 2]; a2 += s_red[(w4 * BN + threadIdx.x) * 2 + 1]; }
                float *dst = part + (((size_t)n * gridDim.x + tile) * Cout + co0 + threadIdx.x) * 2;
                dst[0] = a1; dst[1] = a2;
            }
        } else if (threadIdx.x < 2 * BN) {
            // 16-row tiles: `part` keeps the layout of the 8-row tiling (octa_conv_stat_tiles): waves 0-1 hold the upper half's rows,
            // waves 2-3 the lower half's, each half is one 8-row tile of that grid
            const int half = threadIdx.x / BN, c = threadIdx.x % BN;
            const int ty8 = (tile / tiles_x) * 2 + half, tiles_y8 = (Ho + 7) / 8;
            if (ty8 < tiles_y8) {
                const float a1 = s_red[((2 * half) * BN + c) * 2] + s_red[((2 * half + 1) * BN + c) * 2];
                const float a2 = s_red[((2 * half) * BN + c) * 2 + 1] + s_red[((2 * half + 1) * BN + c) * 2 + 1];
                float *dst = part + (((size_t)n * tiles_y8 * tiles_x + (size_t)ty8 * tiles_x + tile % tiles_x) * Cout + co0 + c) * 2;
                dst[0] = a1; dst[1] = a2;
            }
        }
    }
}

template <int BN, int KCV, int ST = 1, int KS = 3, int THT = TH>
int launch_conv_glds(const unsigned short *X, const unsigned short *X2, int C1, const unsigned short *Wt, unsigned short *Y, unsigned short *Y2,
                     int CY1, int N, int H, int W, int Cin, int Ho, int Wo, int Cout, int dil, const unsigned short *zero16, float *part, int tap_mask,
                     int osc, int ooy, int oox, hipStream_t stream, int pad = 1, const unsigned short *R = nullptr, int reflect = 0, int nslot = 0) {
    constexpr int IH = (THT - 1) * ST + KS, IW = (TW - 1) * ST + KS, PP = KCV / 8;
    constexpr int BUF = ((IH * ST * ((IW + ST - 1) / ST) * PP + 63) / 64 + KS * KS * BN * PP / 64) * 1024;
    constexpr int OUT = THT * TW * (BN * 2 + 16) + 16 + 4 * BN * 2 * 4;      // output tile + the statistics epilogue's [4 waves][BN][2] floats behind it
    const size_t lds = 2 * BUF > OUT ? 2 * BUF : OUT;
    const int tiles_x = (Wo + TW - 1) / TW, tiles_y = (Ho + THT - 1) / THT;
    auto kern = conv3x3_nhwc_glds_kernel<BN, KCV, false, false, ST, KS, THT>;
    if constexpr (KS == 3) {
        if (part) kern = tap_mask != 0x1ff || osc != 1 ? conv3x3_nhwc_glds_kernel<BN, KCV, true, true, ST, 3, THT> : conv3x3_nhwc_glds_kernel<BN, KCV, true, false, ST, 3, THT>;
        else if (tap_mask != 0x1ff || osc != 1) kern = conv3x3_nhwc_glds_kernel<BN, KCV, false, true, ST, 3, THT>;
    }
    OCTA_HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void *>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    dim3 grid((unsigned)(tiles_x * tiles_y), (unsigned)(Cout / BN), (unsigned)N);
    hipLaunchKernelGGL(kern, grid, dim3(CONV_THREADS), lds, stream, X, X2, C1, Wt, Y, Y2, CY1, H, W, Cin, Ho, Wo, Cout, dil, tiles_x, zero16, part, tap_mask, osc, ooy, oox, pad, R, reflect, nslot);
    OCTA_HIP_CHECK(hipGetLastError());
    return 0;
}

template <int BN, int ST>
int launch_conv(const unsigned short *X, const unsigned short *X2, int C1, const unsigned short *Wt, unsigned short *Y, unsigned short *Y2,
                int CY1, int N, int H, int W, int Cin, int Ho, int Wo, int Cout, int dil, int tap_mask, int osc, int ooy, int oox,
                const float *sc1, const float *sh1, const float *sc2, const float *sh2, float slope, float *part, hipStream_t stream) {
    if (sc1 || sc2 || part || osc != 1)
        return launch_conv_impl<BN, ST, true, true>(X, X2, C1, Wt, Y, Y2, CY1, N, H, W, Cin, Ho, Wo, Cout, dil, tap_mask, osc, ooy, oox, sc1, sh1, sc2,
                                                    sh2, slope, part, stream);
    if (tap_mask != 0x1ff)
        return launch_conv_impl<BN, ST, false, true>(X, X2, C1, Wt, Y, Y2, CY1, N, H, W, Cin, Ho, Wo, Cout, dil, tap_mask, 1, 0, 0, nullptr, nullptr,
                                                     nullptr, nullptr, 0.f, nullptr, stream);
    return launch_conv_impl<BN, ST, false, false>(X, X2, C1, Wt, Y, Y2, CY1, N, H, W, Cin, Ho, Wo, Cout, dil, tap_mask, 1, 0, 0, nullptr, nullptr,
                                                  nullptr, nullptr, 0.f, nullptr, stream);
}

// ---- stride-2 "up" convolution with the four output parities fused (round 5) ------------------------------------------------------
// Two layers of the U-Net produce an image TWICE the size of their input: the data gradient of a stride-2 3x3 convolution and the 2x2
// stride-2 transposed convolution (= the data gradient of a stride-2 layer whose taps r, s = 0 are zero; MONAI UnetBasicBlock stride 2 /
// UnetUpBlock.transp_conv, imported at models/networks.py:6). Rounds 1-4 ran them through the stride-1 kernel on a VIRTUALLY zero-inserted
// input (dil = 2): three of four multiply-adds hit an inserted zero. Here a workgroup takes an 8 x 32 tile of the SMALL image and produces
// the 16 x 64 output pixels above it: out[2h + a][2w + b] = sum over the taps of parity class (a, b) of in[h + oy][w + ox] . Wt[3 kr + ks]
// with, per axis, tap k = 1 -> class 0, offset 0; k = 0 -> class 1, offset 0; k = 2 -> class 1, offset +1 (Wt = the SAME flipped,
// transposed pack the zero-insertion form reads: mfma_conv.pack_weight_dgrad / pack_convt2x2()[1]). One accumulator set per class (4 x RPW x NB
// fragments), 9 MFMA groups per 16-channel slice for four times the output pixels: a quarter of the matrix work, the input read once,
// the output leaving as whole rows through LDS. DMA staging, swizzle and operand layout as conv3x3_nhwc_glds_kernel (KCV = 16).
template <int BN>
__global__ void __launch_bounds__(CONV_THREADS, 2)
conv3x3_s2t_kernel(const unsigned short *__restrict__ X, const unsigned short *__restrict__ Wt, unsigned short *__restrict__ Y, int H, int W, int Cin,
                   int Cout, int tiles_x, const unsigned short *__restrict__ zero16, int tap_mask, const unsigned short *__restrict__ R) {
    constexpr int KCV = 16, PP = 2, NB = BN / 32, RPW = 2;
    constexpr int IH = TH + 1, IW = TW + 1, LPIX = IH * IW;
    constexpr int IN_INSTR = (LPIX * PP + 63) / 64, W_INSTR = 9 * BN * PP / 64;
    constexpr int IN_BYTES = IN_INSTR * 1024, BUF = IN_BYTES + W_INSTR * 1024;
    constexpr int IN_PW = (IN_INSTR + 3) / 4, W_PW = (W_INSTR + 3) / 4;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const int tile = blockIdx.x, n = blockIdx.z, co0 = blockIdx.y * BN;
    const int ty0 = (tile / tiles_x) * TH, tx0 = (tile % tiles_x) * TW;            // small-image coordinates
    const int lane = threadIdx.x & 63, wv = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int m = lane & 31, kg = lane >> 5;
    int in_src[IN_PW], w_src[W_PW];
#pragma unroll
    for (int i = 0; i < IN_PW; i++) {
        const int slot = (wv + 4 * i) * 64 + lane, p = slot / PP, q = (slot % PP) ^ glds_swz<PP>(p);
        const int yy = ty0 + p / IW, xx = tx0 + p % IW;
        const bool ok = p < LPIX && yy < H && xx < W;
        in_src[i] = ok ? ((yy * W + xx) << 2) | q : -1;
    }
#pragma unroll
    for (int i = 0; i < W_PW; i++) {
        const int slot = (wv + 4 * i) * 64 + lane, rw = slot / PP, q = (slot % PP) ^ glds_swz<PP>(rw);
        w_src[i] = (int)wt_off(rw / BN, co0 + rw % BN, q * 8, 9, Cout);
    }
    const unsigned short *img = X + (size_t)n * H * W * Cin;
    auto issue = [&](int c0, unsigned char *buf) {
#pragma unroll
        for (int i = 0; i < IN_PW; i++) {
            const int j = wv + 4 * i;
            if (j < IN_INSTR) {
                const unsigned short *src = in_src[i] < 0 ? zero16 : img + (size_t)(in_src[i] >> 2) * Cin + c0 + (in_src[i] & 3) * 8;
                glds16(src, buf + j * 1024);
            }
        }
#pragma unroll
        for (int i = 0; i < W_PW; i++) {
            const int j = wv + 4 * i;
            if (j < W_INSTR && ((tap_mask >> (j * (64 / PP) / BN)) & 1)) glds16(Wt + w_src[i] + (size_t)c0 * 9 * Cout, buf + IN_BYTES + j * 1024);
        }
    };
    f32x16 acc[4][RPW][NB];
#pragma unroll
    for (int c = 0; c < 4; c++)
#pragma unroll
        for (int a = 0; a < RPW; a++)
#pragma unroll
            for (int b = 0; b < NB; b++)
#pragma unroll
                for (int k = 0; k < 16; k++) acc[c][a][b][k] = 0.f;
    const int nsl = Cin / KCV;
    issue(0, smem);
    for (int k = 0; k < nsl; k++) {
        __syncthreads();
        if (k + 1 < nsl) issue((k + 1) * KCV, smem + ((k + 1) & 1) * BUF);
        const unsigned char *s_in = smem + (k & 1) * BUF, *s_w = s_in + IN_BYTES;
        // the six operand fragments of this wave's two small rows: halo rows 2 wv .. 2 wv + 2, column shifts 0 / 1
        bf16x8 a[3][2];
#pragma unroll
        for (int hr = 0; hr < 3; hr++)
#pragma unroll
            for (int ox = 0; ox < 2; ox++) {
                const int p = (RPW * wv + hr) * IW + m + ox;
                a[hr][ox] = *reinterpret_cast<const bf16x8 *>(s_in + (p * PP + (kg ^ glds_swz<PP>(p))) * 16);
            }
#pragma unroll
        for (int kr = 0; kr < 3; kr++)
#pragma unroll
            for (int ks = 0; ks < 3; ks++) {
                if (!((tap_mask >> (3 * kr + ks)) & 1)) continue;          // wave-uniform: the transposed convolution has one tap per class
                const int cls = (kr != 1 ? 2 : 0) + (ks != 1 ? 1 : 0), oy = kr == 2 ? 1 : 0, ox = ks == 2 ? 1 : 0;
#pragma unroll
                for (int nb = 0; nb < NB; nb++) {
                    const int rw = (3 * kr + ks) * BN + nb * 32 + m;
                    const bf16x8 b = *reinterpret_cast<const bf16x8 *>(s_w + (rw * PP + (kg ^ glds_swz<PP>(rw))) * 16);
#pragma unroll
                    for (int rr = 0; rr < RPW; rr++)
                        acc[cls][rr][nb] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[rr + oy][ox], b, acc[cls][rr][nb], 0, 0, 0);
                }
            }
    }
    // epilogue: the 16 x 64 output pixels in two passes of 8 output rows (waves 2 p, 2 p + 1 hold them) through LDS, whole rows out
    const int Ho = 2 * H, Wo = 2 * W;
    constexpr int OP = BN * 2 + 16;
    unsigned char *s_out = smem;
    constexpr int PIECES = BN / 8;
#pragma unroll
    for (int pass = 0; pass < 2; pass++) {
        __syncthreads();
        if ((wv >> 1) == pass) {
#pragma unroll
            for (int cls = 0; cls < 4; cls++)
#pragma unroll
                for (int rr = 0; rr < RPW; rr++) {
                    const int orow = 2 * (RPW * (wv & 1) + rr) + (cls >> 1);           // 0 .. 7 inside the pass
#pragma unroll
                    for (int nb = 0; nb < NB; nb++)
#pragma unroll
                        for (int k = 0; k < 16; k += 2) {
                            const int px = (k & 3) + 8 * (k >> 2) + 4 * kg;              // small-image column; k + 1 is px + 1
                            const unsigned pk = octa_pack_bf16x2(acc[cls][rr][nb][k], acc[cls][rr][nb][k + 1]);
                            unsigned char *d = s_out + (orow * (2 * TW) + 2 * px + (cls & 1)) * OP + (nb * 32 + m) * 2;
                            *reinterpret_cast<unsigned short *>(d) = (unsigned short)(pk & 0xffffu);
                            *reinterpret_cast<unsigned short *>(d + 2 * OP) = (unsigned short)(pk >> 16);
                        }
                }
        }
        __syncthreads();
        for (int i = threadIdx.x; i < 8 * (2 * TW) * PIECES; i += CONV_THREADS) {
            const int p = i / PIECES, q = i % PIECES;
            const int oy = 2 * ty0 + 8 * pass + p / (2 * TW), ox = 2 * tx0 + p % (2 * TW);
            if (oy < Ho && ox < Wo) {
                const size_t off = (((size_t)n * Ho + oy) * Wo + ox) * Cout + co0 + q * 8;
                uint4 v = *reinterpret_cast<const uint4 *>(s_out + p * OP + q * 16);
                if (R) {        // the other gradient of a skip tensor, added as octa_conv3x3_nhwc_fwd6 adds it (fp32 add of the rounded values, rounded again)
                    const uint4 r = *reinterpret_cast<const uint4 *>(R + off);
                    unsigned av[4] = {v.x, v.y, v.z, v.w};
                    const unsigned bv[4] = {r.x, r.y, r.z, r.w};
#pragma unroll
                    for (int k = 0; k < 4; k++)
                        av[k] = octa_pack_bf16x2(__uint_as_float(av[k] << 16) + __uint_as_float(bv[k] << 16),
                                                 __uint_as_float(av[k] & 0xffff0000u) + __uint_as_float(bv[k] & 0xffff0000u));
                    v = make_uint4(av[0], av[1], av[2], av[3]);
                }
                *reinterpret_cast<uint4 *>(Y + off) = v;
            }
        }
    }
}

template <int BN>
int launch_conv_s2t(const unsigned short *X, const unsigned short *Wt, unsigned short *Y, int N, int H, int W, int Cin, int Cout, const unsigned short *zero16,
                    int tap_mask, const unsigned short *R, hipStream_t stream) {
    constexpr int PP = 2;
    constexpr int BUF = (((TH + 1) * (TW + 1) * PP + 63) / 64 + 9 * BN * PP / 64) * 1024;
    constexpr int OUT = 8 * (2 * TW) * (BN * 2 + 16);
    const size_t lds = 2 * BUF > OUT ? 2 * BUF : OUT;
    const int tiles_x = (W + TW - 1) / TW, tiles_y = (H + TH - 1) / TH;
    auto kern = conv3x3_s2t_kernel<BN>;
    OCTA_HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void *>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    dim3 grid((unsigned)(tiles_x * tiles_y), (unsigned)(Cout / BN), (unsigned)N);
    hipLaunchKernelGGL(kern, grid, dim3(CONV_THREADS), lds, stream, X, Wt, Y, H, W, Cin, Cout, tiles_x, zero16, tap_mask, R);
    OCTA_HIP_CHECK(hipGetLastError());
    return 0;
}

}  // namespace

extern "C" int octa_conv_stat_tiles(int Ho, int Wo) { return ((Wo + TW - 1) / TW) * ((Ho + TH - 1) / TH); }

namespace {
int conv3x3_fwd_impl(octa_ctx *ctx, const void *d_x, const void *d_x2, int C1, const void *d_w, void *d_y, void *d_y2,
                     int CY1, int N, int H, int W, int Cin, int Cout, int stride, int in_dilation, int tap_mask,
                     int out_scale, int out_off_y, int out_off_x, const float *d_scale1, const float *d_shift1,
                     const float *d_scale2, const float *d_shift2, float slope, float *d_stat_partials,
                     const void *d_residual, double *d_stat_slots, int nslot, void *stream_) {
    if (!ctx || !d_x || !d_w || !d_y) { octa::set_error("octa_conv3x3_nhwc_fwd: null pointer"); return -2; }
    if (N <= 0 || H <= 0 || W <= 0) { octa::set_error("octa_conv3x3_nhwc_fwd: bad shape"); return -2; }
    if (Cin % 32 || Cout % 32 || Cin <= 0 || Cout <= 0) { octa::set_error("octa_conv3x3_nhwc_fwd: Cin and Cout must be multiples of 32 (got %d, %d)", Cin, Cout); return -2; }
    if ((stride != 1 && stride != 2) || (in_dilation != 1 && in_dilation != 2) || (stride == 2 && in_dilation == 2)) {
        octa::set_error("octa_conv3x3_nhwc_fwd: stride %d / input dilation %d not supported", stride, in_dilation);
        return -2;
    }
    if (N > 65535) { octa::set_error("octa_conv3x3_nhwc_fwd: N > 65535"); return -2; }
    if (!d_x2) C1 = Cin;
    if (!d_y2) CY1 = Cout;
    tap_mask &= 0x1ff;
    if (tap_mask == 0) { octa::set_error("octa_conv3x3_nhwc_fwd: empty tap mask"); return -2; }
    if (out_scale < 1 || out_scale > 2 || out_off_y < 0 || out_off_y >= out_scale || out_off_x < 0 || out_off_x >= out_scale) {
        octa::set_error("octa_conv3x3_nhwc_fwd: output scatter must be scale 1 or 2 with offsets below the scale");
        return -2;
    }
    if (d_stat_slots && (nslot <= 0 || nslot > 1024 || d_stat_partials || d_scale1 || d_scale2)) { octa::set_error("octa_conv3x3_nhwc_fwd: statistics slots need 1..1024 slots, no per-tile partials beside them and the DMA-staged kernel"); return -2; }
    if (!d_stat_slots) nslot = 0;
    if (d_stat_slots) d_stat_partials = reinterpret_cast<float *>(d_stat_slots);      // one kernel argument: read as double slots when nslot > 0
    if (d_stat_partials && (out_scale != 1 || d_y2)) { octa::set_error("octa_conv3x3_nhwc_fwd: statistics need a plain single output"); return -2; }
    const unsigned short *Rz = static_cast<const unsigned short *>(d_residual);
    if (Rz && (out_scale != 1 || d_y2 || d_scale1 || d_scale2 || d_stat_partials || d_residual == d_y)) {
        octa::set_error("octa_conv3x3_nhwc_fwd: the residual needs a plain single output on the DMA-staged kernel");
        return -2;
    }
    if ((d_scale1 == nullptr) != (d_shift1 == nullptr) || (d_scale2 == nullptr) != (d_shift2 == nullptr)) { octa::set_error("octa_conv3x3_nhwc_fwd: scale and shift come in pairs"); return -2; }
    if (C1 <= 0 || C1 > Cin || C1 % 32 || CY1 <= 0 || CY1 > Cout || CY1 % 32) { octa::set_error("octa_conv3x3_nhwc_fwd: channel splits must be multiples of 32 inside the channel range"); return -2; }
    hipStream_t stream = (hipStream_t)stream_;
    OCTA_HIP_CHECK(hipSetDevice(ctx->device));
    const int Hv = H * in_dilation, Wv = W * in_dilation;
    const int Ho = (Hv + 2 - 3) / stride + 1, Wo = (Wv + 2 - 3) / stride + 1;
    const unsigned short *X = static_cast<const unsigned short *>(d_x), *X2 = static_cast<const unsigned short *>(d_x2);
    const unsigned short *Wt = static_cast<const unsigned short *>(d_w);
    unsigned short *Y = static_cast<unsigned short *>(d_y), *Y2 = static_cast<unsigned short *>(d_y2);
    const bool wide = (Cout % 64 == 0) && (CY1 % 64 == 0);   // a 64-channel block must not straddle the output split
    // plain stride-1 layers: the DMA-staged kernel (OCTA_CONV_GLDS=0 selects the register-staged one, =16 (default) / =32 the slice depth)
    constexpr int glds_mode = 16;        // DMA-staged kernels with 16-channel slices (rounds 3-5 A/B'd 0 / 16 / 32 through OCTA_CONV_GLDS; 16 ships)
    if (glds_mode && !d_scale1 && !d_scale2) {
        const unsigned short *z = zero_page(ctx);
        if (!z) return -1;
        if (stride == 2)
            return wide ? launch_conv_glds<64, 16, 2>(X, X2, C1, Wt, Y, Y2, CY1, N, H, W, Cin, Ho, Wo, Cout, 1, z, d_stat_partials, tap_mask, out_scale, out_off_y, out_off_x, stream, 1, Rz, 0, nslot)
                        : launch_conv_glds<32, 16, 2>(X, X2, C1, Wt, Y, Y2, CY1, N, H, W, Cin, Ho, Wo, Cout, 1, z, d_stat_partials, tap_mask, out_scale, out_off_y, out_off_x, stream, 1, Rz, 0, nslot);
        // 16-row tiles (a wave owns four tile rows: 6 operand reads per 8 MFMAs instead of 4 per 4, 72 MFMAs per barrier) from
        // 200 output rows up: 8-14 % faster on the 304^2 / 608^2 layers (256->128 at 304^2: 1.0 PFLOP/s), no gain at 152^2
        // (half as many workgroups: tail effects) and on the HBM-bound 1216^2 layers. OCTA_CONV_TALL=0 disables.
        constexpr int tall = 200;
        if (glds_mode == 16 && tall && wide && stride == 1 && tap_mask == 0x1ff && out_scale == 1 && Ho >= tall)
            return launch_conv_glds<64, 16, 1, 3, 16>(X, X2, C1, Wt, Y, Y2, CY1, N, H, W, Cin, Ho, Wo, Cout, in_dilation, z, d_stat_partials, tap_mask, 1, 0, 0, stream, 1, Rz, 0, nslot);
        if (glds_mode == 16)
            return wide ? launch_conv_glds<64, 16>(X, X2, C1, Wt, Y, Y2, CY1, N, H, W, Cin, Ho, Wo, Cout, in_dilation, z, d_stat_partials, tap_mask, out_scale, out_off_y, out_off_x, stream, 1, Rz, 0, nslot)
                        : launch_conv_glds<32, 16>(X, X2, C1, Wt, Y, Y2, CY1, N, H, W, Cin, Ho, Wo, Cout, in_dilation, z, d_stat_partials, tap_mask, out_scale, out_off_y, out_off_x, stream, 1, Rz, 0, nslot);
        return wide ? launch_conv_glds<64, 32>(X, X2, C1, Wt, Y, Y2, CY1, N, H, W, Cin, Ho, Wo, Cout, in_dilation, z, d_stat_partials, tap_mask, out_scale, out_off_y, out_off_x, stream, 1, Rz, 0, nslot)
                    : launch_conv_glds<32, 32>(X, X2, C1, Wt, Y, Y2, CY1, N, H, W, Cin, Ho, Wo, Cout, in_dilation, z, d_stat_partials, tap_mask, out_scale, out_off_y, out_off_x, stream, 1, Rz, 0, nslot);
    }
    if (Rz) { octa::set_error("octa_conv3x3_nhwc_fwd: the residual is implemented in the DMA-staged kernel only (OCTA_CONV_GLDS=0 is set)"); return -2; }
    if (nslot) { octa::set_error("octa_conv3x3_nhwc_fwd: statistics slots are implemented in the DMA-staged kernel only (OCTA_CONV_GLDS=0 is set)"); return -2; }
    if (stride == 1) return wide ? launch_conv<64, 1>(X, X2, C1, Wt, Y, Y2, CY1, N, H, W, Cin, Ho, Wo, Cout, in_dilation, tap_mask, out_scale, out_off_y, out_off_x, d_scale1, d_shift1, d_scale2, d_shift2, slope, d_stat_partials, stream)
                                 : launch_conv<32, 1>(X, X2, C1, Wt, Y, Y2, CY1, N, H, W, Cin, Ho, Wo, Cout, in_dilation, tap_mask, out_scale, out_off_y, out_off_x, d_scale1, d_shift1, d_scale2, d_shift2, slope, d_stat_partials, stream);
    return wide ? launch_conv<64, 2>(X, X2, C1, Wt, Y, Y2, CY1, N, H, W, Cin, Ho, Wo, Cout, in_dilation, tap_mask, out_scale, out_off_y, out_off_x, d_scale1, d_shift1, d_scale2, d_shift2, slope, d_stat_partials, stream)
                : launch_conv<32, 2>(X, X2, C1, Wt, Y, Y2, CY1, N, H, W, Cin, Ho, Wo, Cout, in_dilation, tap_mask, out_scale, out_off_y, out_off_x, d_scale1, d_shift1, d_scale2, d_shift2, slope, d_stat_partials, stream);
}
}  // namespace

extern "C" int octa_conv3x3_nhwc_fwd6(octa_ctx *ctx, const void *d_x, const void *d_x2, int C1, const void *d_w, void *d_y, void *d_y2,
                                      int CY1, int N, int H, int W, int Cin, int Cout, int stride, int in_dilation, int tap_mask,
                                      int out_scale, int out_off_y, int out_off_x, const float *d_scale1, const float *d_shift1,
                                      const float *d_scale2, const float *d_shift2, float slope, float *d_stat_partials,
                                      const void *d_residual, void *stream_) {
    return conv3x3_fwd_impl(ctx, d_x, d_x2, C1, d_w, d_y, d_y2, CY1, N, H, W, Cin, Cout, stride, in_dilation, tap_mask, out_scale, out_off_y, out_off_x,
                            d_scale1, d_shift1, d_scale2, d_shift2, slope, d_stat_partials, d_residual, nullptr, 0, stream_);
}

// fwd7 = fwd2 + the InstanceNorm statistics of the RESULT in slot form (round 5): d_stat_slots is double[nslot][N][Cout][2], zero on entry;
// every output tile adds the sum and the sum of squares of its bf16-rounded results per channel to slot (tile % nslot). Replaces the
// statistics pass of the norm layer that follows (reference: MONAI UnetBasicBlock conv -> InstanceNorm, models/networks.py:6);
// octa_instnorm_lrelu_nhwc_fwd_s consumes the slots. Plain single output, stride 1 or 2, no input dilation mask restrictions beyond fwd2's.
extern "C" int octa_conv3x3_nhwc_fwd7(octa_ctx *ctx, const void *d_x, const void *d_x2, int C1, const void *d_w, void *d_y, int N, int H, int W,
                                      int Cin, int Cout, int stride, double *d_stat_slots, int nslot, void *stream_) {
    if (!d_stat_slots) { octa::set_error("octa_conv3x3_nhwc_fwd7: null statistics slots"); return -2; }
    return conv3x3_fwd_impl(ctx, d_x, d_x2, C1, d_w, d_y, nullptr, Cout, N, H, W, Cin, Cout, stride, 1, 0x1ff, 1, 0, 0, nullptr, nullptr, nullptr, nullptr,
                            0.f, nullptr, nullptr, d_stat_slots, nslot, stream_);
}

extern "C" int octa_conv3x3_nhwc_fwd5(octa_ctx *ctx, const void *d_x, const void *d_x2, int C1, const void *d_w, void *d_y, void *d_y2,
                                      int CY1, int N, int H, int W, int Cin, int Cout, int stride, int in_dilation, int tap_mask,
                                      int out_scale, int out_off_y, int out_off_x, const float *d_scale1, const float *d_shift1,
                                      const float *d_scale2, const float *d_shift2, float slope, float *d_stat_partials, void *stream_) {
    return octa_conv3x3_nhwc_fwd6(ctx, d_x, d_x2, C1, d_w, d_y, d_y2, CY1, N, H, W, Cin, Cout, stride, in_dilation, tap_mask, out_scale, out_off_y,
                                  out_off_x, d_scale1, d_shift1, d_scale2, d_shift2, slope, d_stat_partials, nullptr, stream_);
}

extern "C" int octa_conv3x3_nhwc_fwd4(octa_ctx *ctx, const void *d_x, const void *d_x2, int C1, const void *d_w, void *d_y, void *d_y2,
                                      int CY1, int N, int H, int W, int Cin, int Cout, int stride, int in_dilation, int tap_mask,
                                      int out_scale, int out_off_y, int out_off_x, const float *d_scale1, const float *d_shift1,
                                      const float *d_scale2, const float *d_shift2, float slope, void *stream_) {
    return octa_conv3x3_nhwc_fwd5(ctx, d_x, d_x2, C1, d_w, d_y, d_y2, CY1, N, H, W, Cin, Cout, stride, in_dilation, tap_mask, out_scale, out_off_y,
                                  out_off_x, d_scale1, d_shift1, d_scale2, d_shift2, slope, nullptr, stream_);
}

extern "C" int octa_conv3x3_nhwc_fwd3(octa_ctx *ctx, const void *d_x, const void *d_x2, int C1, const void *d_w, void *d_y, void *d_y2,
                                      int CY1, int N, int H, int W, int Cin, int Cout, int stride, int in_dilation, int tap_mask,
                                      int out_scale, int out_off_y, int out_off_x, void *stream_) {
    return octa_conv3x3_nhwc_fwd4(ctx, d_x, d_x2, C1, d_w, d_y, d_y2, CY1, N, H, W, Cin, Cout, stride, in_dilation, tap_mask, out_scale, out_off_y,
                                  out_off_x, nullptr, nullptr, nullptr, nullptr, 0.f, stream_);
}

extern "C" int octa_conv3x3_nhwc_fwd2(octa_ctx *ctx, const void *d_x, const void *d_x2, int C1, const void *d_w, void *d_y, void *d_y2,
                                      int CY1, int N, int H, int W, int Cin, int Cout, int stride, int in_dilation, int tap_mask,
                                      void *stream_) {
    return octa_conv3x3_nhwc_fwd3(ctx, d_x, d_x2, C1, d_w, d_y, d_y2, CY1, N, H, W, Cin, Cout, stride, in_dilation, tap_mask, 1, 0, 0, stream_);
}

extern "C" int octa_conv3x3_nhwc_fwd(octa_ctx *ctx, const void *d_x, const void *d_w, void *d_y, int N, int H, int W, int Cin,
                                     int Cout, int stride, int in_dilation, void *stream_) {
    return octa_conv3x3_nhwc_fwd2(ctx, d_x, nullptr, Cin, d_w, d_y, nullptr, Cout, N, H, W, Cin, Cout, stride, in_dilation, 0x1ff, stream_);
}

// d_x [N][H][W][Cin] bf16 (the SMALL image), d_w [9][Cout][Cin] bf16 packed as for the zero-insertion form (octa_conv3x3_nhwc_fwd2 with
// in_dilation = 2 gives the same result: tests/test_conv_gpu.py), d_y [N][2H][2W][Cout] bf16; tap_mask as there (0x1ff: data gradient of a
// stride-2 3x3 layer; 0b000011011: the 2x2 stride-2 transposed convolution); d_residual (shape of d_y, may be NULL) is added as in _fwd6.
extern "C" int octa_conv3x3_s2t_nhwc(octa_ctx *ctx, const void *d_x, const void *d_w, void *d_y, int N, int H, int W, int Cin, int Cout, int tap_mask,
                                     const void *d_residual, void *stream_) {
    if (!ctx || !d_x || !d_w || !d_y) { octa::set_error("octa_conv3x3_s2t_nhwc: null pointer"); return -2; }
    if (N <= 0 || N > 65535 || H <= 0 || W <= 0) { octa::set_error("octa_conv3x3_s2t_nhwc: bad shape"); return -2; }
    if (Cin % 32 || Cout % 32 || Cin <= 0 || Cout <= 0) { octa::set_error("octa_conv3x3_s2t_nhwc: Cin and Cout must be multiples of 32 (got %d, %d)", Cin, Cout); return -2; }
    tap_mask &= 0x1ff;
    if (tap_mask == 0 || d_residual == d_y) { octa::set_error("octa_conv3x3_s2t_nhwc: empty tap mask or residual aliasing the output"); return -2; }
    hipStream_t stream = (hipStream_t)stream_;
    OCTA_HIP_CHECK(hipSetDevice(ctx->device));
    const unsigned short *z = zero_page(ctx);
    if (!z) return -1;
    const unsigned short *X = static_cast<const unsigned short *>(d_x), *Wt = static_cast<const unsigned short *>(d_w), *R = static_cast<const unsigned short *>(d_residual);
    unsigned short *Y = static_cast<unsigned short *>(d_y);
    // 32 output channels per workgroup everywhere: 128 accumulator registers, two workgroups per CU. The 64-channel form (256 accumulator
    // registers in AGPRs, one workgroup of four waves per CU) measured slower on every layer (data gradients 159 / 238 / 260 us against
    // 123 / 169 / 262, transposed convolutions 90 / 118 / 128 against 74 / 96 / 129; zero-insertion form: 180 / 212 / 296 and 112 / 131 / 245).
    return launch_conv_s2t<32>(X, Wt, Y, N, H, W, Cin, Cout, z, tap_mask, R, stream);
}

extern "C" int octa_conv4x4_nhwc_fwd(octa_ctx *ctx, const void *d_x, const void *d_w, void *d_y, int N, int H, int W, int Cin, int Cout, int pad,
                                     void *stream_) {
    if (!ctx || !d_x || !d_w || !d_y) { octa::set_error("octa_conv4x4_nhwc_fwd: null pointer"); return -2; }
    if (N <= 0 || N > 65535 || H <= 0 || W <= 0 || pad < 0 || pad > 3 || H + 2 * pad < 4 || W + 2 * pad < 4) { octa::set_error("octa_conv4x4_nhwc_fwd: bad shape"); return -2; }
    if (Cin % 32 || Cout % 32 || Cin <= 0 || Cout <= 0) { octa::set_error("octa_conv4x4_nhwc_fwd: Cin and Cout must be multiples of 32 (got %d, %d)", Cin, Cout); return -2; }
    hipStream_t stream = (hipStream_t)stream_;
    OCTA_HIP_CHECK(hipSetDevice(ctx->device));
    const unsigned short *z = zero_page(ctx);
    if (!z) return -1;
    const int Ho = H + 2 * pad - 3, Wo = W + 2 * pad - 3;
    const unsigned short *X = static_cast<const unsigned short *>(d_x), *Wt = static_cast<const unsigned short *>(d_w);
    unsigned short *Y = static_cast<unsigned short *>(d_y);
    if (Cout % 64 == 0)
        return launch_conv_glds<64, 16, 1, 4>(X, nullptr, Cin, Wt, Y, nullptr, Cout, N, H, W, Cin, Ho, Wo, Cout, 1, z, nullptr, 0xffff, 1, 0, 0, stream, pad);
    return launch_conv_glds<32, 16, 1, 4>(X, nullptr, Cin, Wt, Y, nullptr, Cout, N, H, W, Cin, Ho, Wo, Cout, 1, z, nullptr, 0xffff, 1, 0, 0, stream, pad);
}

// 3 x 3 convolution, stride 1, with an explicit padding: pad = 0 (valid), 1 (same) or 2 (full: what the data gradient of a valid
// convolution is), zeros outside the image -- or, reflect = 1 with pad = 1, nn.ReflectionPad2d(1) fused into the halo fetch (the
// ResNet blocks of the generator, models/networks.py:_resblock_nhwc: no padded copy, no cropped copy). Output (H + 2 pad - 2)^2.
extern "C" int octa_conv3x3_nhwc_fwd_pad_s(octa_ctx *ctx, const void *d_x, const void *d_w, void *d_y, int N, int H, int W, int Cin, int Cout, int pad,
                                           int reflect, double *d_stat_slots, int nslot, void *stream_);
extern "C" int octa_conv3x3_nhwc_fwd_pad(octa_ctx *ctx, const void *d_x, const void *d_w, void *d_y, int N, int H, int W, int Cin, int Cout, int pad,
                                         int reflect, void *stream_) {
    return octa_conv3x3_nhwc_fwd_pad_s(ctx, d_x, d_w, d_y, N, H, W, Cin, Cout, pad, reflect, nullptr, 0, stream_);
}

// ... with the InstanceNorm statistics of the result accumulated by the epilogue in slot form (d_stat_slots: double [nslot][N][Cout][2], zeroed by
// the caller; as octa_conv3x3_nhwc_fwd7): the generator's residual blocks are reflect-padded convolution -> InstanceNorm, 18 of them per pass,
// and each paid a statistics launch over a tensor that had just been written (round 5). NULL slots = octa_conv3x3_nhwc_fwd_pad.
extern "C" int octa_conv3x3_nhwc_fwd_pad_s(octa_ctx *ctx, const void *d_x, const void *d_w, void *d_y, int N, int H, int W, int Cin, int Cout, int pad,
                                           int reflect, double *d_stat_slots, int nslot, void *stream_) {
    if (!ctx || !d_x || !d_w || !d_y) { octa::set_error("octa_conv3x3_nhwc_fwd_pad: null pointer"); return -2; }
    if (d_stat_slots && (nslot <= 0 || nslot > 1024)) { octa::set_error("octa_conv3x3_nhwc_fwd_pad_s: statistics slots need 1..1024 slots"); return -2; }
    if (!d_stat_slots) nslot = 0;
    float *part = reinterpret_cast<float *>(d_stat_slots);      // one kernel argument: read as double slots when nslot > 0
    if (N <= 0 || N > 65535 || H <= 0 || W <= 0 || pad < 0 || pad > 2 || H + 2 * pad < 3 || W + 2 * pad < 3) { octa::set_error("octa_conv3x3_nhwc_fwd_pad: bad shape"); return -2; }
    if (reflect && (pad != 1 || H < 2 || W < 2)) { octa::set_error("octa_conv3x3_nhwc_fwd_pad: reflection needs pad = 1 and an image of at least 2 x 2"); return -2; }
    if (Cin % 32 || Cout % 32 || Cin <= 0 || Cout <= 0) { octa::set_error("octa_conv3x3_nhwc_fwd_pad: Cin and Cout must be multiples of 32 (got %d, %d)", Cin, Cout); return -2; }
    hipStream_t stream = (hipStream_t)stream_;
    OCTA_HIP_CHECK(hipSetDevice(ctx->device));
    const unsigned short *z = zero_page(ctx);
    if (!z) return -1;
    const int Ho = H + 2 * pad - 2, Wo = W + 2 * pad - 2;
    const unsigned short *X = static_cast<const unsigned short *>(d_x), *Wt = static_cast<const unsigned short *>(d_w);
    unsigned short *Y = static_cast<unsigned short *>(d_y);
    if (Cout % 64 == 0) {
        if (Ho >= 200) return launch_conv_glds<64, 16, 1, 3, 16>(X, nullptr, Cin, Wt, Y, nullptr, Cout, N, H, W, Cin, Ho, Wo, Cout, 1, z, part, 0x1ff, 1, 0, 0, stream, pad, nullptr, reflect, nslot);
        return launch_conv_glds<64, 16>(X, nullptr, Cin, Wt, Y, nullptr, Cout, N, H, W, Cin, Ho, Wo, Cout, 1, z, part, 0x1ff, 1, 0, 0, stream, pad, nullptr, reflect, nslot);
    }
    return launch_conv_glds<32, 16>(X, nullptr, Cin, Wt, Y, nullptr, Cout, N, H, W, Cin, Ho, Wo, Cout, 1, z, part, 0x1ff, 1, 0, 0, stream, pad, nullptr, reflect, nslot);
}

// ---- weight gradient (stride 1) ----------------------------------------------------------------------------------
// dW[tap][co][ci] = sum over pixels p of dY[p][co] * X[p + tap - (1,1)][ci]: a GEMM whose contraction index is
// the PIXEL, while NHWC memory is contiguous in the channel. The tiles are therefore transposed on their way
// into LDS (16-byte global loads of 8 channels, eight 2-byte LDS stores each), so that an MFMA operand -- 8
// consecutive pixels of one channel -- is one aligned ds_read_b128; the +-1 pixel shift of the taps s = 0, 2
// is a funnel shift of five dwords in registers. One workgroup = 4 waves works on COB x CIB channels: every
// wave owns one 32 x 32 (co, ci) pair and all nine taps (144 accumulator registers) over a share of the tile's
// rows; workgroups are persistent over the pixel tiles and add their fp32 partial sums to dW with atomics once.
namespace {

constexpr int WTH = 4;                 // tile rows of the weight-gradient kernel (4 x 32 = 128 pixels per tile)
// LDS row pitches of the transposed tiles are padded to 16 (mod 256) bytes: the 16 lanes of a ds_read_b128 group read
// the same pixel group of 16 DIFFERENT channel rows, so the pitch must walk the 64 banks four at a time (a 256-byte
// pitch would put all 16 rows on the same four banks: a 16-way conflict on every operand read).
constexpr int wg_pad_pitch(int bytes) { return ((bytes - 16 + 255) / 256) * 256 + 16; }
constexpr int WG_ROWP = wg_pad_pitch(TW * 2 * WTH);  // bytes per channel row of the transposed dY tile
constexpr int HALO_W = 40;             // halo row pitch in pixels (34 used; 80 B keeps rows 16-byte aligned)

// ST = 2: weight gradient of a stride-2 layer. dY has Ho x Wo pixels, the input 2Ho x 2Wo; tap (r, s) pairs output
// pixel (y, x) with input pixel (2y + r - 1, 2x + s - 1). The transposed X tile keeps its ODD and EVEN halo columns in
// two planes per row, so the eight input pixels of a K-group are contiguous again: tap s = 0 reads the odd plane, s = 1
// the even plane, s = 2 the odd plane one element further (funnel shift).
// Workgroups per CU: the 32 x 32-channel stride-1 tile (the 1216^2 level: HBM / latency-bound, 18 MFMAs per wave and tile)
// is compiled for two resident workgroups -- twice the loads in flight; the larger tiles need the whole register file.
template <int COB, int CIB, int ST, int KS = 3> constexpr int wgrad_wgs_per_cu() { return COB == 32 && CIB == 32 && ST == 1 && KS == 3 ? 2 : 1; }

template <int COB, int CIB, bool MASKED, int ST, bool XFORM, int KS>
__global__ void __launch_bounds__(CONV_THREADS, (wgrad_wgs_per_cu<COB, CIB, ST, KS>()))
conv3x3_nhwc_wgrad_kernel(const unsigned short *__restrict__ X, const unsigned short *__restrict__ X2, int C1,
                          const unsigned short *__restrict__ dY, float *__restrict__ dW,
                          int N, int H, int W, int Ho, int Wo, int Cin, int Cout, int tiles_x, int tiles_y, int tap_mask,
                          const float *__restrict__ sc1, const float *__restrict__ sh1, const float *__restrict__ sc2,
                          const float *__restrict__ sh2, float slope, const unsigned short *__restrict__ zero16) {
    constexpr int PAIRS = (COB / 32) * (CIB / 32);
    constexpr int KSPLIT = 4 / PAIRS;            // waves sharing one (co, ci) pair split the tile rows
    constexpr int ROWS_PER_WAVE = WTH / KSPLIT;
    constexpr int SLOTS = ROWS_PER_WAVE * (TW / 16);   // MFMA groups of the compute loop = places to tuck LDS stores
    constexpr int XROWS = ST == 1 ? WTH + KS - 1 : 2 * WTH + 1, XCOLS = ST == 1 ? TW + KS - 1 : 2 * TW + 1;   // KS x KS taps (4: the PatchGAN, stride 1 only)
    static_assert(KS == 3 || (KS == 4 && ST == 1 && !MASKED && !XFORM), "4x4 taps: plain stride-1 variant only");
    constexpr int XRP = ST == 1 ? HALO_W * 2 : 2 * HALO_W * 2;      // bytes per halo row (two column-parity planes for ST = 2)
    constexpr int WG_XROW = wg_pad_pitch(XROWS * XRP);               // bytes per channel row of the transposed X tile
    constexpr int DYPIX = WTH * TW, XPIX = XROWS * XCOLS;
    constexpr int DY_ITEMS = DYPIX * (COB / 8), X_ITEMS = XPIX * (CIB / 8);   // 16-byte pieces per tile
    constexpr int DY_PT = (DY_ITEMS + CONV_THREADS - 1) / CONV_THREADS, X_PT = (X_ITEMS + CONV_THREADS - 1) / CONV_THREADS;
    constexpr int BUF = COB * WG_ROWP + CIB * WG_XROW;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const int co0 = (blockIdx.y / (Cin / CIB)) * COB, ci0 = (blockIdx.y % (Cin / CIB)) * CIB;
    // virtual concatenation of the input: a CIB block lies entirely in X (channels < C1) or in X2
    const unsigned short *Xs = ci0 < C1 ? X : X2;
    const int xcs = ci0 < C1 ? C1 : Cin - C1, xcb = ci0 < C1 ? ci0 : ci0 - C1;
    // normalise-on-load of the layer input (see the forward kernel): per image and channel scale / shift
    const float *scp = ci0 < C1 ? sc1 : sc2, *shp = ci0 < C1 ? sh1 : sh2;
    const bool xform = XFORM && scp != nullptr;   // compiled into the XFORM variant only (measured slower in the U-Net step)
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    const int pair = wv % PAIRS, kpart = wv / PAIRS;
    const int cob = (pair % (COB / 32)) * 32, cib = (pair / (COB / 32)) * 32;
    const int m = lane & 31, kg = lane >> 5;
    f32x16 acc[KS * KS];
#pragma unroll
    for (int t = 0; t < KS * KS; t++)
#pragma unroll
        for (int k = 0; k < 16; k++) acc[t][k] = 0.f;
    const int n_tiles = tiles_x * tiles_y * N;
    // Pipeline over the tiles of this workgroup: tile t is contracted out of LDS buffer t&1 while the registers
    // holding tile t+1 (loaded from HBM during tile t-1) are written, transposed, into the other buffer BETWEEN the
    // MFMA groups -- LDS stores issue in the shadow of the matrix pipe -- and the loads of tile t+2 are issued.
    // Item i of a tile: pixel i % NPIX, 8-channel group i / NPIX, so consecutive lanes hold consecutive pixels and
    // every 2-byte LDS store of a wave covers contiguous bytes of one channel row (no bank conflicts).
    uint4 r_dy[DY_PT], r_x[X_PT];
    struct TilePos { int n, ty0, tx0; };
    auto tile_pos = [&](int tile) {
        const int n = tile / (tiles_x * tiles_y), tt = tile % (tiles_x * tiles_y);
        return TilePos{n, (tt / tiles_x) * WTH, (tt % tiles_x) * TW};
    };
    auto fetch_dy = [&](const TilePos &tp, int k) {
        const int i = threadIdx.x + k * CONV_THREADS, p = i % DYPIX, q = i / DYPIX;
        const int y = tp.ty0 + p / TW, x = tp.tx0 + p % TW;
        const bool ok = i < DY_ITEMS && y < Ho && x < Wo;
        // the load itself is unconditional (padding comes from a zero page): a branch around it would make the number of
        // loads in flight path-dependent and the compiler would then wait for ALL of them (vmcnt(0)) at every use
        r_dy[k] = *reinterpret_cast<const uint4 *>(ok ? dY + (((size_t)tp.n * Ho + y) * Wo + x) * Cout + co0 + q * 8 : zero16);
    };
    auto fetch_x = [&](const TilePos &tp, int k) {
        const int n = tp.n;
        const int i = threadIdx.x + k * CONV_THREADS, p = i % XPIX, q = i / XPIX;
        const int y = tp.ty0 * ST - 1 + p / XCOLS, x = tp.tx0 * ST - 1 + p % XCOLS;
        const bool ok = i < X_ITEMS && y >= 0 && y < H && x >= 0 && x < W;
        uint4 v = *reinterpret_cast<const uint4 *>(ok ? Xs + (((size_t)n * H + y) * W + x) * xcs + xcb + q * 8 : zero16);
        if (XFORM && xform && ok) {
            const float4 s0 = *reinterpret_cast<const float4 *>(scp + (size_t)n * xcs + xcb + q * 8), s1 = *reinterpret_cast<const float4 *>(scp + (size_t)n * xcs + xcb + q * 8 + 4);
            const float4 h0 = *reinterpret_cast<const float4 *>(shp + (size_t)n * xcs + xcb + q * 8), h1 = *reinterpret_cast<const float4 *>(shp + (size_t)n * xcs + xcb + q * 8 + 4);
            const float sc[8] = {s0.x, s0.y, s0.z, s0.w, s1.x, s1.y, s1.z, s1.w}, sh[8] = {h0.x, h0.y, h0.z, h0.w, h1.x, h1.y, h1.z, h1.w};
            unsigned u[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
            for (int j = 0; j < 4; j++) {
                float z0 = __uint_as_float(u[j] << 16) * sc[2 * j] + sh[2 * j];
                float z1 = __uint_as_float(u[j] & 0xffff0000u) * sc[2 * j + 1] + sh[2 * j + 1];
                z0 = z0 > 0.f ? z0 : z0 * slope;
                z1 = z1 > 0.f ? z1 : z1 * slope;
                u[j] = octa_pack_bf16x2(z0, z1);
            }
            v = make_uint4(u[0], u[1], u[2], u[3]);
        }
        r_x[k] = v;
    };
    auto fetch = [&](int tile) {
        const TilePos tp = tile_pos(tile);
#pragma unroll
        for (int k = 0; k < DY_PT; k++) fetch_dy(tp, k);
#pragma unroll
        for (int k = 0; k < X_PT; k++) fetch_x(tp, k);
    };
    auto stash_dy = [&](unsigned char *buf, int k) {
        const int i = threadIdx.x + k * CONV_THREADS, p = i % DYPIX, q = i / DYPIX;
        if (i < DY_ITEMS) {
            const unsigned w4[4] = {r_dy[k].x, r_dy[k].y, r_dy[k].z, r_dy[k].w};
#pragma unroll
            for (int j = 0; j < 8; j++)
                *reinterpret_cast<unsigned short *>(buf + (q * 8 + j) * WG_ROWP + p * 2) = (unsigned short)(w4[j >> 1] >> ((j & 1) * 16));
        }
    };
    auto stash_x = [&](unsigned char *buf, int k) {
        const int i = threadIdx.x + k * CONV_THREADS, p = i % XPIX, q = i / XPIX;
        if (i < X_ITEMS) {
            const int hy = p / XCOLS, hx = p % XCOLS;
            // ST = 2: halo column hx is input column 2*tx0 - 1 + hx: even hx -> odd plane (0), odd hx -> even plane (1)
            const int off = ST == 1 ? hy * XRP + hx * 2 : hy * XRP + (hx & 1) * (HALO_W * 2) + (hx >> 1) * 2;
            const unsigned w4[4] = {r_x[k].x, r_x[k].y, r_x[k].z, r_x[k].w};
#pragma unroll
            for (int j = 0; j < 8; j++)
                *reinterpret_cast<unsigned short *>(buf + COB * WG_ROWP + (q * 8 + j) * WG_XROW + off) = (unsigned short)(w4[j >> 1] >> ((j & 1) * 16));
        }
    };
    int tile = blockIdx.x;
    if (tile < n_tiles) {
        fetch(tile);
#pragma unroll
        for (int k = 0; k < DY_PT; k++) stash_dy(smem, k);
#pragma unroll
        for (int k = 0; k < X_PT; k++) stash_x(smem, k);
        // staging registers <- tile after this one, issued in the order the loop below re-issues them (slot by slot), so
        // that the in-order load counter is the same on the loop's entry edge and on its back edge
        const TilePos tp1 = tile_pos(tile + (int)gridDim.x < n_tiles ? tile + (int)gridDim.x : tile);
#pragma unroll
        for (int slot = 0; slot < SLOTS; slot++) {
#pragma unroll
            for (int k = slot; k < DY_PT; k += SLOTS) fetch_dy(tp1, k);
#pragma unroll
            for (int k = slot; k < X_PT; k += SLOTS) fetch_x(tp1, k);
        }
    }
    __syncthreads();
    int cur = 0;
    for (; tile < n_tiles; tile += gridDim.x) {
        const unsigned char *s_dy = smem + cur * BUF, *s_x = s_dy + COB * WG_ROWP;
        unsigned char *nxt = smem + (cur ^ 1) * BUF;
        const TilePos tp2 = tile_pos(tile + 2 * (int)gridDim.x < n_tiles ? tile + 2 * (int)gridDim.x : tile);
        // Operands of MFMA group `slot`: the dY fragment and, per kernel row, the aligned 8-pixel group of the halo row plus
        // the dword (ST = 1) / the other parity plane (ST = 2) the shifted taps need. They are read one group AHEAD, before
        // the MFMAs of the current group and before its share of the transposed stores (which the compiler must assume to
        // alias), so the LDS latency hides under nine MFMAs even with one wave per SIMD.
        struct Ops { bf16x8 a; uint4 d[KS]; unsigned e[KS]; unsigned e2[KS]; uint4 f[KS]; };
        // the dword behind the 16-byte group is fetched through an offset the compiler cannot see: it would otherwise fuse the
        // 20 contiguous bytes into ds_read_b96 + ds_read2_b32 (12 LDS cycles) instead of ds_read_b128 + ds_read_b32 (6)
        int e_off = 16;
        asm volatile("" : "+v"(e_off));
        auto load_ops = [&](int slot, Ops &o) {
            const int y = kpart * ROWS_PER_WAVE + slot / (TW / 16), xs = (slot % (TW / 16)) * 16;
            o.a = *reinterpret_cast<const bf16x8 *>(s_dy + (cob + m) * WG_ROWP + (y * TW + xs + kg * 8) * 2);
#pragma unroll
            for (int r = 0; r < KS; r++) {
                if (MASKED && !((tap_mask >> (3 * r)) & 7)) continue;   // no tap of this kernel row is wanted
                // ST = 1: halo columns xs + kg*8 + s .. +7 of halo row y + r. ST = 2: input row 2y + r; s = 0: odd plane at
                // xs.., s = 1: even plane at xs.., s = 2: odd plane at xs+1..
                const unsigned char *row = s_x + (cib + m) * WG_XROW + (ST * y + r) * XRP + (xs + kg * 8) * 2;
                o.d[r] = *reinterpret_cast<const uint4 *>(row);
                o.e[r] = *reinterpret_cast<const unsigned *>(row + e_off);
                if (KS == 4) o.e2[r] = *reinterpret_cast<const unsigned *>(row + e_off + 4);   // pixels +10, +11 for the shift by three
                if (ST == 2) o.f[r] = *reinterpret_cast<const uint4 *>(row + HALO_W * 2);
            }
        };
        Ops ops[2];
        load_ops(0, ops[0]);
#pragma unroll
        for (int slot = 0; slot < SLOTS; slot++) {
            if (slot + 1 < SLOTS) load_ops(slot + 1, ops[(slot + 1) & 1]);
            const Ops &o = ops[slot & 1];
#pragma unroll
            for (int r = 0; r < KS; r++) {
                if (MASKED && !((tap_mask >> (3 * r)) & 7)) continue;
                union { uint4 u; bf16x8 v; } b0, b1, b2;
                const uint4 d = o.d[r];
                const uint4 sh = make_uint4(__builtin_amdgcn_alignbit(d.y, d.x, 16), __builtin_amdgcn_alignbit(d.z, d.y, 16),
                                            __builtin_amdgcn_alignbit(d.w, d.z, 16), __builtin_amdgcn_alignbit(o.e[r], d.w, 16));
                b0.u = d;
                if (ST == 1) { b1.u = sh; b2.u = make_uint4(d.y, d.z, d.w, o.e[r]); }
                else { b1.u = o.f[r]; b2.u = sh; }
                if (!MASKED || ((tap_mask >> (3 * r)) & 1)) acc[KS * r + 0] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(o.a, b0.v, acc[KS * r + 0], 0, 0, 0);
                if (!MASKED || ((tap_mask >> (3 * r + 1)) & 1)) acc[KS * r + 1] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(o.a, b1.v, acc[KS * r + 1], 0, 0, 0);
                if (!MASKED || ((tap_mask >> (3 * r + 2)) & 1)) acc[KS * r + 2] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(o.a, b2.v, acc[KS * r + 2], 0, 0, 0);
                if (KS == 4) {   // shift by three pixels: the funnel shift of the group that starts one dword further
                    union { uint4 u; bf16x8 v; } b3;
                    b3.u = make_uint4(__builtin_amdgcn_alignbit(d.z, d.y, 16), __builtin_amdgcn_alignbit(d.w, d.z, 16),
                                      __builtin_amdgcn_alignbit(o.e[r], d.w, 16), __builtin_amdgcn_alignbit(o.e2[r], o.e[r], 16));
                    acc[KS * r + 3] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(o.a, b3.v, acc[KS * r + 3], 0, 0, 0);
                }
            }
            // a share of the next tile's transposed stores, tucked behind this MFMA group; every staging register is
            // refilled with its piece of the tile after next as soon as it has been stored, so each load has a whole tile
            // of MFMAs to land before its turn comes round again
            // (unconditional: on the last tiles the stores fill a buffer nobody reads and the loads re-read a valid tile --
            // branch-free code keeps the compiler's count of loads in flight exact, so it waits with vmcnt(N), not vmcnt(0))
#pragma unroll
            for (int k = slot; k < DY_PT; k += SLOTS) { stash_dy(nxt, k); fetch_dy(tp2, k); }
#pragma unroll
            for (int k = slot; k < X_PT; k += SLOTS) { stash_x(nxt, k); fetch_x(tp2, k); }
        }
        __syncthreads();
        cur ^= 1;
    }
    // D[row = co][col = ci]: row = (k&3) + 8*(k>>2) + 4*kg, col = m
#pragma unroll
    for (int t = 0; t < KS * KS; t++) {
        if (MASKED && !((tap_mask >> t) & 1)) continue;
#pragma unroll
        for (int k = 0; k < 16; k++) {
            const int co = co0 + cob + (k & 3) + 8 * (k >> 2) + 4 * kg;
            atomicAdd(dW + ((size_t)t * Cout + co) * Cin + ci0 + cib + m, acc[t][k]);
        }
    }
}

// ---- weight gradient, stride 1, 3x3, all taps: raw NHWC tiles in LDS + transposing reads (round 3) ------------------------
// The kernel above transposes its tiles on the way INTO the LDS (eight 2-byte stores per 16-byte piece: 55 % of its LDS cycles were
// bank conflicts, and the +-1 pixel shift of the taps is a funnel shift of registers: 440 VALU instructions per 72 MFMAs). Here the
// tiles stay as they are in memory -- [pixel][32 channels], 64 bytes per pixel, one plane per 32 channels -- and are fetched by
// the DMA (global_load_lds, 16 bytes per lane, a wave-instruction = 16 pixels of one plane); gfx950's ds_read_b64_tr_b16 transposes
// on the way OUT: lane i of a 16-lane group addresses 4 channels of pixel i / 4 and receives pixels 0..3 of channel i -- the
// K-major fragment the MFMA wants (two reads = 8 pixels of one channel per lane). A tap is then a constant added to the pixel
// index, i.e. an immediate offset of the read: every operand read of the kernel is `base VGPR + immediate`, no shifts, no
// register transposition. A 32-lane half of the read covers 4 pixels x 64 bytes = 256 contiguous bytes: all 64 banks once.
// Work split as above: a wave owns one 32 x 32 (co, ci) pair and all nine taps over its share of the tile's rows; the X
// fragments of a 16-pixel column group are read once per halo row and tap column and serve every tile row they touch
// ((rows + 2) x 3 fragments for rows x 9 MFMAs).
typedef __attribute__((ext_vector_type(4))) short s16x4;

__device__ __forceinline__ bf16x8 tr_read8(const unsigned char *p) {
    // two transposing reads, pixels +0..3 and +4..7 (256 bytes further) of this lane's channel
    union { struct { s16x4 lo, hi; } h; bf16x8 v; } u;
    u.h.lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4 *)(p));
    u.h.hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4 *)(p + 256));
    return u.v;
}

// TH_: tile rows (4 or 8; 8 halves the barriers and the halo share of the DMA where the double buffer fits).
// ST = 2 (stride-2 layers; dY has Ho x Wo = H / 2 x W / 2 pixels): tap (r, s) pairs output pixel (y, x) with input pixel
// (2y + r - 1, 2x + s - 1); the X tile keeps the even and the odd halo columns of a row in two planes of 33 pixels ([row][parity][33]),
// so the 16 input pixels of a K-chunk are consecutive LDS pixels again and a tap is still an immediate offset:
// ((2y + r) * 2 + s % 2) * 33 + x + s / 2. MASKED (the 2 x 2 transposed convolution's weight gradient wants four of the nine taps):
// all nine are multiplied -- skipping reads and MFMAs behind run-time tests measured slower than the full loop (0.158 against 0.131 ms
// at 1216^2 32->64) -- and the taps outside tap_mask are left out of the result.
// REFLECT (stride 1): the halo mirrors the image -- nn.ReflectionPad2d(1) in front of the convolution (octa_conv3x3_nhwc_fwd_pad); a template
// parameter because the issue code below sits BETWEEN the MFMAs of a wave that has its SIMD to itself: every instruction of it is a cycle
// the matrix pipe may idle (round 5: with the mirror arithmetic, the dY / X choice and the range tests behind run-time branches one DMA
// instruction cost 53-84 instructions, 1355 per 144 MFMAs; MFMA busy 32 %, and removing the in-loop DMA alone took 512->512 at 152^2
// from 0.593 to 0.337 ms).
// TMASK: taps known at COMPILE time (the 2 x 2 transposed convolution's weight gradient wants taps r, s in {1, 2} of a stride-2 layer:
// 0b110110000): the MFMAs and operand reads of the other taps are not generated -- 4 of 9 MFMAs per pixel group; the run-time mask of
// MASKED only leaves taps out of the result.
template <int COB, int CIB, int TH_, int ST, bool MASKED, bool REFLECT = false, int TMASK = 0x1ff>
__global__ void __launch_bounds__(CONV_THREADS, (COB == 32 && CIB == 32 && ST == 1 ? 2 : 1))
conv3x3_nhwc_wgrad_tr_kernel(const unsigned short *__restrict__ X, const unsigned short *__restrict__ X2, int C1,
                             const unsigned short *__restrict__ dY, float *__restrict__ dW,
                             int N, int H, int W, int Ho, int Wo, int Cin, int Cout, int tiles_x, int tiles_y, int tap_mask,
                             const unsigned short *__restrict__ zero16, float *__restrict__ ws, int pad) {
    constexpr int PAIRS = (COB / 32) * (CIB / 32), KSPLIT = 4 / PAIRS, RPW = TH_ / KSPLIT;   // tile rows per wave
    constexpr int RG = RPW < 4 ? RPW : 4, NG = RPW / RG;            // rows per operand set (register budget), sets per column group
    constexpr int XROWS = ST * (TH_ - 1) + 3;                                                // halo rows
    constexpr int XCOLS = ST == 1 ? TW + 2 : 2 * (TW + 1);                                   // LDS pixels per halo row (ST = 2: two parity planes of 33)
    constexpr int XPIX = XROWS * XCOLS;
    constexpr int NBR = ST * (RG - 1) + 3;                                                   // halo rows one operand set touches
    constexpr int DY_IPP = TH_ * TW / 16, X_IPP = (XPIX + 15) / 16;                          // DMA instructions (16 pixels) per 32-channel plane
    constexpr int DY_PLANE = DY_IPP * 1024, X_PLANE = X_IPP * 1024;
    constexpr int DY_INSTR = (COB / 32) * DY_IPP, X_INSTR = (CIB / 32) * X_IPP, N_INSTR = DY_INSTR + X_INSTR;
    constexpr int IPW = (N_INSTR + 3) / 4;                                                   // per wave and tile
    constexpr int BUF = (COB / 32) * DY_PLANE + (CIB / 32) * X_PLANE;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const int co0 = (blockIdx.y / (Cin / CIB)) * COB, ci0 = (blockIdx.y % (Cin / CIB)) * CIB;
    const unsigned short *Xs = ci0 < C1 ? X : X2;
    const int xcs = ci0 < C1 ? C1 : Cin - C1, xcb = ci0 < C1 ? ci0 : ci0 - C1;
    const int lane = threadIdx.x & 63, wv = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int pair = wv % PAIRS, kpart = wv / PAIRS;
    const int cobp = pair % (COB / 32), cibp = pair / (COB / 32);        // this wave's 32-channel planes
    const int m = lane & 31, kg = lane >> 5;
    f32x16 acc[9];
#pragma unroll
    for (int t = 0; t < 9; t++)
#pragma unroll
        for (int k = 0; k < 16; k++) acc[t][k] = 0.f;
    const int n_tiles = tiles_x * tiles_y * N;

    // DMA slots of this wave: instruction j = wv + 4 i; lane -> pixel 16 (j % IPP) + lane / 4 of plane j / IPP, channels 8 (lane % 4) ..
    // Per slot and lane, once: the pixel's row / column relative to the tile origin and its element offset from the origin pixel;
    // per tile only the origin (scalar), two unsigned range tests and one 64-bit add per slot are left.
    const int lp = lane >> 2, lq = lane & 3;
    int s_rel[IPW], s_py[IPW], s_px[IPW];
#pragma unroll
    for (int i = 0; i < IPW; i++) {
        const int j = wv + 4 * i;
        if (j < DY_INSTR) {
            const int plane = j / DY_IPP, p = (j % DY_IPP) * 16 + lp;
            s_py[i] = p / TW; s_px[i] = p % TW;
            s_rel[i] = (s_py[i] * Wo + s_px[i]) * Cout + plane * 32 + lq * 8;
        } else {
            const int jj = j - DY_INSTR, plane = jj / X_IPP, p = (jj % X_IPP) * 16 + lp;
            const int hy = p / XCOLS, rem = p % XCOLS;
            const int hx = ST == 1 ? rem : 2 * (rem % (TW + 1)) + rem / (TW + 1);          // halo column of LDS pixel `rem` of the row
            s_py[i] = (p < XPIX && hx <= ST * TW + 1 - (ST == 1 ? 0 : 1)) ? hy - pad : -(1 << 20); s_px[i] = hx - pad;   // pad = 1 but for the valid / full forms (stride 1)
            s_rel[i] = (s_py[i] * W + s_px[i]) * xcs + plane * 32 + lq * 8;
        }
    }
    struct TilePos { int n, ty0, tx0; };
    auto tile_pos = [&](int tile) {
        const int n = tile / (tiles_x * tiles_y), tt = tile % (tiles_x * tiles_y);
        return TilePos{n, (tt / tiles_x) * TH_, (tt % tiles_x) * TW};      // in output (dY) pixels
    };
    // part < 0: all of the wave's instructions; otherwise only instruction `part` (the main loop spreads them between its MFMAs).
    // Instruction j = wv + 4 i: DY_INSTR is a multiple of four, so slot i is a dY slot for every wave or for none (compile time), and only the
    // last slot can lie beyond N_INSTR. Per slot: two adds, two unsigned range tests, one 64-bit add, two selects, M0, the load.
    static_assert(DY_INSTR % 4 == 0, "a DMA slot must not mix dY and X instructions across the waves");
    auto issue = [&](const TilePos &tp, unsigned char *buf, int part) {
        const unsigned short *dy0 = dY + (((size_t)tp.n * Ho + tp.ty0) * Wo + tp.tx0) * Cout + co0;
        const unsigned short *x0 = Xs + (((size_t)tp.n * H + ST * tp.ty0) * W + ST * tp.tx0) * xcs + xcb;
        const int xy0 = ST * tp.ty0, xx0 = ST * tp.tx0;
#pragma unroll
        for (int i = 0; i < IPW; i++) {
            if (part >= 0 && i != part) continue;
            const int j = wv + 4 * i;
            if (4 * i + 3 >= N_INSTR && j >= N_INSTR) continue;
            if (i < DY_INSTR / 4) {
                const bool ok = (unsigned)(tp.ty0 + s_py[i]) < (unsigned)Ho && (unsigned)(tp.tx0 + s_px[i]) < (unsigned)Wo;
                glds16(ok ? dy0 + s_rel[i] : zero16, buf + j * 1024);
            } else if (!REFLECT) {
                const bool ok = (unsigned)(xy0 + s_py[i]) < (unsigned)H && (unsigned)(xx0 + s_px[i]) < (unsigned)W;
                glds16(ok ? x0 + s_rel[i] : zero16, buf + j * 1024);
            } else {
                int y = xy0 + s_py[i], x = xx0 + s_px[i];
                y = y < 0 ? -y : (y >= H ? 2 * (H - 1) - y : y);
                x = x < 0 ? -x : (x >= W ? 2 * (W - 1) - x : x);
                // LDS pixels that belong to no halo position (s_py = -2^20) and tiles beyond the image edge (further than one mirror) read zeros
                const bool in = (unsigned)y < (unsigned)H && (unsigned)x < (unsigned)W;
                const unsigned short *src = Xs + (((size_t)tp.n * H + (in ? y : 0)) * W + (in ? x : 0)) * xcs + xcb + ((j - DY_INSTR) / X_IPP) * 32 + lq * 8;
                glds16(in ? src : zero16, buf + j * 1024);
            }
        }
    };

    // lane part of every operand address: pixel (lane / 32) * 8 + (lane % 16) / 4, channels 16 * ((lane / 16) % 2) + 4 * (lane % 4)
    const int la = (((lane >> 5) * 8 + ((lane & 15) >> 2)) * 64) + (16 * ((lane >> 4) & 1) + 4 * (lane & 3)) * 2;
    (void)m; (void)kg;

    constexpr int NSTEP = (TW / 16) * NG;                           // operand sets per tile: (column group, row group)
    constexpr int NMF = (TW / 16) * RPW * 9;                       // MFMAs per wave and tile
    constexpr int ISTRIDE = (NMF * 2 / 3) / IPW > 0 ? (NMF * 2 / 3) / IPW : 1;   // one DMA instruction every ISTRIDE MFMAs, the last third of the tile carries none
    struct Frags { bf16x8 b[NBR][3], a[RG]; };
    // operand reads of one 16-pixel column group in the order of their first use (the compiler's lgkmcnt then lets the first tile
    // row's MFMAs start while the later fragments are still on their way)
    auto load_frags = [&](const unsigned char *a_base, const unsigned char *b_base, int step, Frags &f) {
        const int xc = step / NG, row0 = kpart * RPW + (step % NG) * RG;
#pragma unroll
        for (int hr = 0; hr < NBR; hr++) {
            if (ST == 1 ? (hr >= 2 && hr - 2 < RG) : (hr % 2 == 0 && hr / 2 < RG))
                f.a[ST == 1 ? hr - 2 : hr / 2] = tr_read8(a_base + ((row0 + (ST == 1 ? hr - 2 : hr / 2)) * TW + xc * 16) * 64);
#pragma unroll
            for (int sx = 0; sx < 3; sx++) {
                if (!((TMASK >> sx) & 0x49)) continue;           // no wanted tap in this tap column (bits sx, sx + 3, sx + 6)
                const int e = ST == 1 ? (row0 + hr) * XCOLS + xc * 16 + sx : ((ST * row0 + hr) * 2 + (sx & 1)) * (TW + 1) + xc * 16 + (sx >> 1);
                f.b[hr][sx] = tr_read8(b_base + e * 64);
            }
        }
    };
    int tile = blockIdx.x, cur = 0;
    if (tile < n_tiles) issue(tile_pos(tile), smem, -1);
    for (; tile < n_tiles; tile += gridDim.x) {
        __syncthreads();        // this tile has landed (every wave drained its own DMA queue), the other buffer is free
        const int nxt = tile + (int)gridDim.x;
        const bool more = nxt < n_tiles;
        const TilePos tpn = tile_pos(more ? nxt : tile);
        unsigned char *nbuf = smem + (cur ^ 1) * BUF;
        const unsigned char *a_base = smem + cur * BUF + cobp * DY_PLANE + la;
        const unsigned char *b_base = smem + cur * BUF + (COB / 32) * DY_PLANE + cibp * X_PLANE + la;
        // two operand sets (the next set's reads ahead of this one's MFMAs) where one wave per SIMD has the registers for it; the
        // 32 x 32 block runs two workgroups per CU on 8-row tiles (HBM-bound layers: bytes in flight matter, not the matrix pipe)
        constexpr bool PIPE = !(COB == 32 && CIB == 32);
        Frags fr[PIPE ? 2 : 1];
        load_frags(a_base, b_base, 0, fr[0]);
#pragma unroll
        for (int st = 0; st < NSTEP; st++) {
            if (PIPE && st + 1 < NSTEP) load_frags(a_base, b_base, st + 1, fr[(st + 1) & 1]);
            if (!PIPE && st > 0) load_frags(a_base, b_base, st, fr[0]);
            __builtin_amdgcn_sched_barrier(0);
            const Frags &f = fr[PIPE ? (st & 1) : 0];
#pragma unroll
            for (int rr = 0; rr < RG; rr++)
#pragma unroll
                for (int r = 0; r < 3; r++)
#pragma unroll
                    for (int sx = 0; sx < 3; sx++) {
                        if ((TMASK >> (3 * r + sx)) & 1)
                            acc[3 * r + sx] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(f.a[rr], f.b[ST * rr + r][sx], acc[3 * r + sx], 0, 0, 0);
                        const int q = (st * RG + rr) * 9 + 3 * r + sx;
                        if (q % ISTRIDE == ISTRIDE - 1 && q / ISTRIDE < IPW) {
                            __builtin_amdgcn_sched_barrier(0);
                            if (more) issue(tpn, nbuf, q / ISTRIDE);
                            __builtin_amdgcn_sched_barrier(0);
                        }
                    }
            __builtin_amdgcn_sched_barrier(0);
        }
        cur ^= 1;
    }
    // D[row = co][col = ci]: row = (k&3) + 8*(k>>2) + 4*(lane/32), col = lane % 32
    if (KSPLIT > 1) {
        // the waves that shared a (co, ci) pair fold their partial sums in the LDS first (every lane owns the same 144 addresses in
        // each of them: plain read-modify-write, one barrier per wave): 1 / KSPLIT of the atomics / workspace traffic below
        __syncthreads();                                   // every wave is done with the operand buffers
        float *red = reinterpret_cast<float *>(smem) + pair * (9 * 32 * 32);
#pragma unroll
        for (int w = 0; w < KSPLIT; w++) {
            if (kpart == w) {
#pragma unroll
                for (int t = 0; t < 9; t++)
#pragma unroll
                    for (int k = 0; k < 16; k++) {
                        float *q = red + (t * 16 + k) * 64 + lane;
                        if (w == 0) *q = acc[t][k]; else if (w + 1 < KSPLIT) *q += acc[t][k]; else acc[t][k] += *q;
                    }
            }
            if (w + 1 < KSPLIT) __syncthreads();
        }
        if (kpart != KSPLIT - 1) return;                   // the last wave of the pair holds the total
    }
    if (ws) {
        // many workgroups per channel block (the wide, few-channel layers: up to 512 workgroups adding into the same 9216 weights):
        // plain stores of this pair's partial sums, folded by wgrad_tr_reduce_kernel
        float *mine = ws + (((size_t)blockIdx.y * gridDim.x + blockIdx.x) * PAIRS + pair) * (9 * 32 * 32);
#pragma unroll
        for (int t = 0; t < 9; t++)
#pragma unroll
            for (int k = 0; k < 16; k++)
                mine[(t * 32 + (k & 3) + 8 * (k >> 2) + 4 * (lane >> 5)) * 32 + (lane & 31)] = (!MASKED || ((tap_mask >> t) & 1)) ? acc[t][k] : 0.f;
        return;
    }
#pragma unroll
    for (int t = 0; t < 9; t++) {
        if (MASKED && !((tap_mask >> t) & 1)) continue;      // dW was cleared: masked taps come back as zero
#pragma unroll
        for (int k = 0; k < 16; k++) {
            const int co = co0 + cobp * 32 + (k & 3) + 8 * (k >> 2) + 4 * (lane >> 5);
            atomicAdd(dW + ((size_t)t * Cout + co) * Cin + ci0 + cibp * 32 + (lane & 31), acc[t][k]);
        }
    }
}

// dW[t][co][ci] = sum over the workgroups of the channel block of their partial 32 x 32 tiles (ws: [block][workgroup][pair][9][32][32]);
// thread = one weight, consecutive threads = consecutive ci
template <int COB, int CIB>
__global__ void __launch_bounds__(256)
wgrad_tr_reduce_kernel(const float *__restrict__ ws, float *__restrict__ dW, int Cin, int Cout, int per_block) {
    constexpr int PAIRS = (COB / 32) * (CIB / 32);
    const int e = blockIdx.x * 256 + threadIdx.x;                 // (t, co, ci) of the whole weight tensor
    if (e >= 9 * Cout * Cin) return;
    const int ci = e % Cin, co = (e / Cin) % Cout, t = e / (Cin * Cout);
    const int blk = (co / COB) * (Cin / CIB) + ci / CIB;
    const int pair = (co % COB) / 32 + (COB / 32) * ((ci % CIB) / 32);
    const float *p = ws + ((size_t)blk * per_block * PAIRS + pair) * (9 * 32 * 32) + (t * 32 + co % 32) * 32 + ci % 32;
    float s0 = 0.f, s1 = 0.f, s2 = 0.f, s3 = 0.f;
    int g = 0;
    for (; g + 3 < per_block; g += 4) {
        s0 += p[(size_t)g * PAIRS * (9 * 32 * 32)];
        s1 += p[(size_t)(g + 1) * PAIRS * (9 * 32 * 32)];
        s2 += p[(size_t)(g + 2) * PAIRS * (9 * 32 * 32)];
        s3 += p[(size_t)(g + 3) * PAIRS * (9 * 32 * 32)];
    }
    for (; g < per_block; g++) s0 += p[(size_t)g * PAIRS * (9 * 32 * 32)];
    dW[e] = (s0 + s1) + (s2 + s3);
}

// The partial tiles of the workspace summed STRAIGHT into a gradient buffer in the parameter's own layout, [Cout][Cin][3][3] float32
// (torch / MONAI state-dict layout), ADDING to what it holds: the training step hands `weight.grad` (a view of the flat gradient arena,
// cleared once per step) -- no [9][Cout][Cin] temporary, no fill, no layout copy, no autograd accumulation launch per layer (round 5:
// 21 copies + 15 fills per U-Net step). A workgroup owns (co, 32 ci, all taps): coalesced reads of the partials (ci fastest), an LDS
// transposition, 288 contiguous floats out. gridDim.z > 1 (many partials: the 1216^2 layers, 512 workgroups): slices of the partials,
// folded by atomics. accumulate = 0: the buffer is overwritten (cleared first where the slices need atomics).
template <int COB, int CIB>
__global__ void __launch_bounds__(288)
wgrad_tr_acc_kernel(const float *__restrict__ ws, float *__restrict__ grad, int Cin, int Cout, int per_block, int accumulate) {
    constexpr int PAIRS = (COB / 32) * (CIB / 32);
    __shared__ float s_t[32][10];
    const int t = threadIdx.x >> 5, cil = threadIdx.x & 31;
    const int co = blockIdx.x, ci = blockIdx.y * 32 + cil;
    const int blk = (co / COB) * (Cin / CIB) + ci / CIB;
    const int pair = (co % COB) / 32 + (COB / 32) * ((ci % CIB) / 32);
    const int chunk = (per_block + (int)gridDim.z - 1) / (int)gridDim.z, g0 = blockIdx.z * chunk, g1 = g0 + chunk < per_block ? g0 + chunk : per_block;
    const float *p = ws + ((size_t)blk * per_block * PAIRS + pair) * (9 * 32 * 32) + (t * 32 + co % 32) * 32 + ci % 32;
    float s0 = 0.f, s1 = 0.f, s2 = 0.f, s3 = 0.f;
    int g = g0;
    for (; g + 3 < g1; g += 4) {
        s0 += p[(size_t)g * PAIRS * (9 * 32 * 32)];
        s1 += p[(size_t)(g + 1) * PAIRS * (9 * 32 * 32)];
        s2 += p[(size_t)(g + 2) * PAIRS * (9 * 32 * 32)];
        s3 += p[(size_t)(g + 3) * PAIRS * (9 * 32 * 32)];
    }
    for (; g < g1; g++) s0 += p[(size_t)g * PAIRS * (9 * 32 * 32)];
    s_t[cil][t] = (s0 + s1) + (s2 + s3);
    __syncthreads();
    float *out = grad + ((size_t)co * Cin + blockIdx.y * 32) * 9 + threadIdx.x;      // element (ci = threadIdx.x / 9, tap = threadIdx.x % 9) of this run
    const float v = s_t[threadIdx.x / 9][threadIdx.x % 9];
    if (gridDim.z > 1) atomicAdd(out, v); else if (accumulate) *out += v; else *out = v;
}

template <int COB, int CIB, int ST = 1>
int launch_wgrad_tr(octa_ctx *ctx, const unsigned short *X, const unsigned short *X2, int C1, const unsigned short *dY, float *dW, int N, int H, int W, int Cin,
                    int Cout, int num_cus, int tap_mask, const unsigned short *zero16, hipStream_t stream, int pad = 1, int reflect = 0, float *acc = nullptr, int accumulate = 1) {
    // stride 1: 64 x 64 blocks: the 8-row double buffer fits one CU (152 KB); 32 x 32: two workgroups of 76 KB; the mixed blocks keep 4 rows
    // and two workgroups. Stride 2 (a 9 x 66-pixel halo per 4 output rows): 4 rows, one workgroup per CU
    constexpr int TH_ = (ST == 1 && COB == CIB) ? 8 : 4;
    constexpr int XPIX = (ST * (TH_ - 1) + 3) * (ST == 1 ? TW + 2 : 2 * (TW + 1));
    constexpr int BUF = (COB / 32) * (TH_ * TW / 16) * 1024 + (CIB / 32) * ((XPIX + 15) / 16) * 1024;
    constexpr int PAIRS_ = (COB / 32) * (CIB / 32);
    constexpr size_t FOLD = PAIRS_ < 4 ? (size_t)PAIRS_ * 9 * 32 * 32 * sizeof(float) : 0;     // LDS fold of the waves that share a pair
    const size_t lds = 2 * (size_t)BUF > FOLD ? 2 * (size_t)BUF : FOLD;
    const int Ho = ST == 1 ? H + 2 * pad - 2 : H / ST, Wo = ST == 1 ? W + 2 * pad - 2 : W / ST;
    const int tiles_x = (Wo + TW - 1) / TW, tiles_y = (Ho + TH_ - 1) / TH_;
    const int blocks = (Cout / COB) * (Cin / CIB);
    int per_block = (num_cus * (COB == 32 && CIB == 32 && ST == 1 ? 2 : 1) + blocks - 1) / blocks;
    const int n_tiles = tiles_x * tiles_y * N;
    if (per_block > n_tiles) per_block = n_tiles;
    if (per_block < 1) per_block = 1;
    // 16 - 64 workgroups per channel block: partial sums through a workspace + one reduction launch instead of atomics (152^2 256->256
    // 0.130 -> 0.119 ms, 304^2 128->128 0.133 -> 0.130); with more workgroups per block the reduction's per-thread loop over the partials
    // costs more than the atomics (1216^2 32->32, 512 workgroups: 0.238 against 0.201 ms), with fewer there is little contention.
    // (Measured with the atomics removed altogether: 0.188 / 0.112 / 0.408 ms for 1216^2 32->32 / 152^2 256->256 / 512->512.)
    constexpr int ws_from = 16;
    float *ws = nullptr;
    if (acc || (ws_from > 0 && per_block >= ws_from && per_block <= 64)) {      // acc: always through the workspace (wgrad_tr_acc_kernel folds it)
        if (ctx->wgrad_ws.reserve((size_t)blocks * per_block * PAIRS_ * 9 * 32 * 32 * sizeof(float))) return -1;
        ws = ctx->wgrad_ws.as<float>();
    } else {
        OCTA_HIP_CHECK(hipMemsetAsync(dW, 0, sizeof(float) * 9 * (size_t)Cout * Cin, stream));
    }
    auto kern = conv3x3_nhwc_wgrad_tr_kernel<COB, CIB, TH_, ST, false>;
    if (tap_mask != 0x1ff) kern = conv3x3_nhwc_wgrad_tr_kernel<COB, CIB, TH_, ST, true>;
    if constexpr (ST == 2) { if (tap_mask == 0x1b0) kern = conv3x3_nhwc_wgrad_tr_kernel<COB, CIB, TH_, 2, true, false, 0x1b0>; }      // ConvTranspose2d(k = s = 2)
    if constexpr (ST == 1) { if (reflect) kern = conv3x3_nhwc_wgrad_tr_kernel<COB, CIB, TH_, 1, false, true>; }   // (the mirrored form is never masked: octa_conv3x3_nhwc_wgrad_pad)
    OCTA_HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void *>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    hipLaunchKernelGGL(kern, dim3((unsigned)per_block, (unsigned)blocks), dim3(CONV_THREADS), lds, stream, X, X2, C1, dY, dW, N, H, W, Ho, Wo, Cin, Cout,
                       tiles_x, tiles_y, tap_mask, zero16, ws, pad);
    OCTA_HIP_CHECK(hipGetLastError());
    if (acc) {
        const int slices = per_block > 64 ? (per_block / 32 < 16 ? per_block / 32 : 16) : 1;
        if (!accumulate && slices > 1) OCTA_HIP_CHECK(hipMemsetAsync(acc, 0, sizeof(float) * 9 * (size_t)Cout * Cin, stream));
        hipLaunchKernelGGL((wgrad_tr_acc_kernel<COB, CIB>), dim3((unsigned)Cout, (unsigned)(Cin / 32), (unsigned)slices), dim3(288), 0, stream, ws, acc, Cin, Cout, per_block, accumulate);
        OCTA_HIP_CHECK(hipGetLastError());
    } else if (ws) {
        const int total = 9 * Cout * Cin;
        hipLaunchKernelGGL((wgrad_tr_reduce_kernel<COB, CIB>), dim3((unsigned)((total + 255) / 256)), dim3(256), 0, stream, ws, dW, Cin, Cout, per_block);
        OCTA_HIP_CHECK(hipGetLastError());
    }
    return 0;
}

template <int COB, int CIB, bool MASKED, int ST, int KS = 3>
int launch_wgrad_impl(const unsigned short *X, const unsigned short *X2, int C1, const unsigned short *dY, float *dW, int N, int H, int W, int Cin,
                 int Cout, int num_cus, int tap_mask, const float *sc1, const float *sh1, const float *sc2, const float *sh2, float slope,
                 const unsigned short *zero16, hipStream_t stream) {
    constexpr int XROWS = ST == 1 ? WTH + KS - 1 : 2 * WTH + 1;
    constexpr int WG_XROW = wg_pad_pitch(XROWS * (ST == 1 ? HALO_W * 2 : 2 * HALO_W * 2));
    const size_t lds = 2 * ((size_t)COB * WG_ROWP + (size_t)CIB * WG_XROW);   // double buffered
    const int Ho = ST == 1 ? H + 3 - KS : H / ST, Wo = ST == 1 ? W + 3 - KS : W / ST;   // padding 1
    const int tiles_x = (Wo + TW - 1) / TW, tiles_y = (Ho + WTH - 1) / WTH;
    const int blocks = (Cout / COB) * (Cin / CIB);
    int per_block = (num_cus * wgrad_wgs_per_cu<COB, CIB, ST, KS>() + blocks - 1) / blocks;   // one round of resident workgroups: fewer atomics
    const int n_tiles = tiles_x * tiles_y * N;
    if (per_block > n_tiles) per_block = n_tiles;
    if (per_block < 1) per_block = 1;
    auto kern = conv3x3_nhwc_wgrad_kernel<COB, CIB, MASKED, ST, false, KS>;
    if constexpr (KS == 3) { if (sc1 || sc2) kern = conv3x3_nhwc_wgrad_kernel<COB, CIB, MASKED, ST, true, 3>; }
    OCTA_HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void *>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    hipLaunchKernelGGL(kern, dim3((unsigned)per_block, (unsigned)blocks), dim3(CONV_THREADS), lds, stream, X, X2, C1, dY, dW, N, H, W, Ho, Wo, Cin,
                       Cout, tiles_x, tiles_y, tap_mask, sc1, sh1, sc2, sh2, slope, zero16);
    OCTA_HIP_CHECK(hipGetLastError());
    return 0;
}

template <int COB, int CIB, int ST = 1>
int launch_wgrad(const unsigned short *X, const unsigned short *X2, int C1, const unsigned short *dY, float *dW, int N, int H, int W, int Cin,
                 int Cout, int num_cus, int tap_mask, const float *sc1, const float *sh1, const float *sc2, const float *sh2, float slope,
                 const unsigned short *zero16, hipStream_t stream) {
    if (tap_mask != 0x1ff) return launch_wgrad_impl<COB, CIB, true, ST>(X, X2, C1, dY, dW, N, H, W, Cin, Cout, num_cus, tap_mask, sc1, sh1, sc2, sh2, slope, zero16, stream);
    return launch_wgrad_impl<COB, CIB, false, ST>(X, X2, C1, dY, dW, N, H, W, Cin, Cout, num_cus, tap_mask, sc1, sh1, sc2, sh2, slope, zero16, stream);
}

}  // namespace

// acc: d_dw is a gradient buffer in the PARAMETER layout [Cout][Cin][3][3] that receives the result ADDED to its contents (wgrad_tr_acc_kernel)
static int wgrad4_impl(octa_ctx *ctx, const void *d_x, const void *d_x2, int C1, const void *d_dy, float *d_dw, int N, int H,
                       int W, int Cin, int Cout, int stride, int tap_mask, const float *d_scale1, const float *d_shift1,
                       const float *d_scale2, const float *d_shift2, float slope, void *stream_, int acc /* 0: d_dw [9][Cout][Cin]; 1: add to / 2: overwrite a parameter-layout buffer */) {
    if (!ctx || !d_x || !d_dy || !d_dw) { octa::set_error("octa_conv3x3_nhwc_wgrad: null pointer"); return -2; }
    tap_mask &= 0x1ff;
    if (tap_mask == 0) { octa::set_error("octa_conv3x3_nhwc_wgrad: empty tap mask"); return -2; }
    if (N <= 0 || H <= 0 || W <= 0) { octa::set_error("octa_conv3x3_nhwc_wgrad: bad shape"); return -2; }
    if (Cin % 32 || Cout % 32 || Cin <= 0 || Cout <= 0) { octa::set_error("octa_conv3x3_nhwc_wgrad: Cin and Cout must be multiples of 32 (got %d, %d)", Cin, Cout); return -2; }
    if (!d_x2) C1 = Cin;
    if (C1 <= 0 || C1 > Cin || C1 % 32) { octa::set_error("octa_conv3x3_nhwc_wgrad: the input split must be a multiple of 32 inside the channel range"); return -2; }
    hipStream_t stream = (hipStream_t)stream_;
    OCTA_HIP_CHECK(hipSetDevice(ctx->device));
    const unsigned short *X = static_cast<const unsigned short *>(d_x), *X2 = static_cast<const unsigned short *>(d_x2);
    const unsigned short *dY = static_cast<const unsigned short *>(d_dy);
    const unsigned short *z = zero_page(ctx);
    if (!z) return -1;
    const bool co64 = Cout % 64 == 0, ci64 = Cin % 64 == 0 && C1 % 64 == 0;   // a 64-channel block must not straddle the split
    constexpr int use_tr = 2;           // transposing-read weight-gradient kernels at stride 1 and 2 (the round-3/4 switch OCTA_WGRAD_TR is gone)
    if (use_tr && !d_scale1 && !d_scale2 && (stride == 1 || (stride == 2 && H % 2 == 0 && W % 2 == 0))) {
        // raw tiles + transposing reads (see conv3x3_nhwc_wgrad_tr_kernel); OCTA_WGRAD_TR=1: stride 1 only, =2 (default): stride 2 as well
        if (stride == 1) {
            if (co64 && ci64) return launch_wgrad_tr<64, 64>(ctx, X, X2, C1, dY, d_dw, N, H, W, Cin, Cout, ctx->num_cus, tap_mask, z, stream, 1, 0, acc ? d_dw : nullptr, acc == 1);
            if (co64) return launch_wgrad_tr<64, 32>(ctx, X, X2, C1, dY, d_dw, N, H, W, Cin, Cout, ctx->num_cus, tap_mask, z, stream, 1, 0, acc ? d_dw : nullptr, acc == 1);
            if (ci64) return launch_wgrad_tr<32, 64>(ctx, X, X2, C1, dY, d_dw, N, H, W, Cin, Cout, ctx->num_cus, tap_mask, z, stream, 1, 0, acc ? d_dw : nullptr, acc == 1);
            return launch_wgrad_tr<32, 32>(ctx, X, X2, C1, dY, d_dw, N, H, W, Cin, Cout, ctx->num_cus, tap_mask, z, stream, 1, 0, acc ? d_dw : nullptr, acc == 1);
        }
        if (use_tr >= 2) {
            if (co64) return launch_wgrad_tr<64, 32, 2>(ctx, X, X2, C1, dY, d_dw, N, H, W, Cin, Cout, ctx->num_cus, tap_mask, z, stream, 1, 0, acc ? d_dw : nullptr, acc == 1);
            return launch_wgrad_tr<32, 32, 2>(ctx, X, X2, C1, dY, d_dw, N, H, W, Cin, Cout, ctx->num_cus, tap_mask, z, stream, 1, 0, acc ? d_dw : nullptr, acc == 1);
        }
    }
    if (acc) { octa::set_error("octa_conv3x3_nhwc_wgrad_acc: needs the transposing-read kernels (no normalise-on-load operands, OCTA_WGRAD_TR unset, even sizes at stride 2)"); return -2; }
    OCTA_HIP_CHECK(hipMemsetAsync(d_dw, 0, sizeof(float) * 9 * (size_t)Cout * Cin, stream));
    if (stride == 2) {
        if (H % 2 || W % 2) { octa::set_error("octa_conv3x3_nhwc_wgrad: stride-2 layers need even input sizes"); return -2; }
        // two column-parity planes per halo row: 32 input channels per workgroup keep the double buffer inside the LDS
        if (co64) return launch_wgrad<64, 32, 2>(X, X2, C1, dY, d_dw, N, H, W, Cin, Cout, ctx->num_cus, tap_mask, d_scale1, d_shift1, d_scale2, d_shift2, slope, z, stream);
        return launch_wgrad<32, 32, 2>(X, X2, C1, dY, d_dw, N, H, W, Cin, Cout, ctx->num_cus, tap_mask, d_scale1, d_shift1, d_scale2, d_shift2, slope, z, stream);
    }
    if (stride != 1) { octa::set_error("octa_conv3x3_nhwc_wgrad: stride must be 1 or 2"); return -2; }
    if (co64 && ci64) return launch_wgrad<64, 64>(X, X2, C1, dY, d_dw, N, H, W, Cin, Cout, ctx->num_cus, tap_mask, d_scale1, d_shift1, d_scale2, d_shift2, slope, z, stream);
    if (co64) return launch_wgrad<64, 32>(X, X2, C1, dY, d_dw, N, H, W, Cin, Cout, ctx->num_cus, tap_mask, d_scale1, d_shift1, d_scale2, d_shift2, slope, z, stream);
    if (ci64) return launch_wgrad<32, 64>(X, X2, C1, dY, d_dw, N, H, W, Cin, Cout, ctx->num_cus, tap_mask, d_scale1, d_shift1, d_scale2, d_shift2, slope, z, stream);
    return launch_wgrad<32, 32>(X, X2, C1, dY, d_dw, N, H, W, Cin, Cout, ctx->num_cus, tap_mask, d_scale1, d_shift1, d_scale2, d_shift2, slope, z, stream);
}

extern "C" int octa_conv3x3_nhwc_wgrad4(octa_ctx *ctx, const void *d_x, const void *d_x2, int C1, const void *d_dy, float *d_dw, int N, int H,
                                        int W, int Cin, int Cout, int stride, int tap_mask, const float *d_scale1, const float *d_shift1,
                                        const float *d_scale2, const float *d_shift2, float slope, void *stream_) {
    return wgrad4_impl(ctx, d_x, d_x2, C1, d_dy, d_dw, N, H, W, Cin, Cout, stride, tap_mask, d_scale1, d_shift1, d_scale2, d_shift2, slope, stream_, 0);
}

// octa_conv3x3_nhwc_wgrad4 (no normalise-on-load operands) with the result ADDED to d_grad in the parameter's own layout, float32
// [Cout][Cin][3][3] (models/networks.py / MONAI state-dict layout): what the training step's `weight.grad` is (accumulate = 0: overwritten
// instead). Taps outside tap_mask contribute zero.
extern "C" int octa_conv3x3_nhwc_wgrad_acc(octa_ctx *ctx, const void *d_x, const void *d_x2, int C1, const void *d_dy, float *d_grad, int N, int H,
                                           int W, int Cin, int Cout, int stride, int tap_mask, int accumulate, void *stream_) {
    return wgrad4_impl(ctx, d_x, d_x2, C1, d_dy, d_grad, N, H, W, Cin, Cout, stride, tap_mask, nullptr, nullptr, nullptr, nullptr, 0.f, stream_, accumulate ? 1 : 2);
}

// Weight gradient of octa_conv3x3_nhwc_fwd_pad: d_x [N][H][W][Cin], d_dy [N][H + 2 pad - 2][W + 2 pad - 2][Cout], stride 1; reflect = 1
// (pad = 1): the input was mirrored at its borders (nn.ReflectionPad2d(1) in front of the convolution).
static int wgrad_pad_impl(octa_ctx *ctx, const void *d_x, const void *d_dy, float *d_dw, int N, int H, int W, int Cin, int Cout, int pad,
                          int reflect, void *stream_, int acc) {
    if (!ctx || !d_x || !d_dy || !d_dw) { octa::set_error("octa_conv3x3_nhwc_wgrad_pad: null pointer"); return -2; }
    if (N <= 0 || H <= 0 || W <= 0 || pad < 0 || pad > 2 || H + 2 * pad < 3 || W + 2 * pad < 3) { octa::set_error("octa_conv3x3_nhwc_wgrad_pad: bad shape"); return -2; }
    if (reflect && (pad != 1 || H < 2 || W < 2)) { octa::set_error("octa_conv3x3_nhwc_wgrad_pad: reflection needs pad = 1 and an image of at least 2 x 2"); return -2; }
    if (Cin % 32 || Cout % 32 || Cin <= 0 || Cout <= 0) { octa::set_error("octa_conv3x3_nhwc_wgrad_pad: Cin and Cout must be multiples of 32 (got %d, %d)", Cin, Cout); return -2; }
    hipStream_t stream = (hipStream_t)stream_;
    OCTA_HIP_CHECK(hipSetDevice(ctx->device));
    const unsigned short *z = zero_page(ctx);
    if (!z) return -1;
    const unsigned short *X = static_cast<const unsigned short *>(d_x), *dY = static_cast<const unsigned short *>(d_dy);
    const bool co64 = Cout % 64 == 0, ci64 = Cin % 64 == 0;
    if (co64 && ci64) return launch_wgrad_tr<64, 64>(ctx, X, nullptr, Cin, dY, d_dw, N, H, W, Cin, Cout, ctx->num_cus, 0x1ff, z, stream, pad, reflect, acc ? d_dw : nullptr, acc == 1);
    if (co64) return launch_wgrad_tr<64, 32>(ctx, X, nullptr, Cin, dY, d_dw, N, H, W, Cin, Cout, ctx->num_cus, 0x1ff, z, stream, pad, reflect, acc ? d_dw : nullptr, acc == 1);
    if (ci64) return launch_wgrad_tr<32, 64>(ctx, X, nullptr, Cin, dY, d_dw, N, H, W, Cin, Cout, ctx->num_cus, 0x1ff, z, stream, pad, reflect, acc ? d_dw : nullptr, acc == 1);
    return launch_wgrad_tr<32, 32>(ctx, X, nullptr, Cin, dY, d_dw, N, H, W, Cin, Cout, ctx->num_cus, 0x1ff, z, stream, pad, reflect, acc ? d_dw : nullptr, acc == 1);
}

extern "C" int octa_conv3x3_nhwc_wgrad_pad(octa_ctx *ctx, const void *d_x, const void *d_dy, float *d_dw, int N, int H, int W, int Cin, int Cout, int pad,
                                           int reflect, void *stream_) {
    return wgrad_pad_impl(ctx, d_x, d_dy, d_dw, N, H, W, Cin, Cout, pad, reflect, stream_, 0);
}

// octa_conv3x3_nhwc_wgrad_pad with the result ADDED to d_grad in the parameter layout [Cout][Cin][3][3] (see octa_conv3x3_nhwc_wgrad_acc)
extern "C" int octa_conv3x3_nhwc_wgrad_pad_acc(octa_ctx *ctx, const void *d_x, const void *d_dy, float *d_grad, int N, int H, int W, int Cin, int Cout, int pad,
                                               int reflect, int accumulate, void *stream_) {
    return wgrad_pad_impl(ctx, d_x, d_dy, d_grad, N, H, W, Cin, Cout, pad, reflect, stream_, accumulate ? 1 : 2);
}

extern "C" int octa_conv4x4_nhwc_wgrad(octa_ctx *ctx, const void *d_x, const void *d_dy, float *d_dw, int N, int H, int W, int Cin, int Cout,
                                       void *stream_) {
    if (!ctx || !d_x || !d_dy || !d_dw) { octa::set_error("octa_conv4x4_nhwc_wgrad: null pointer"); return -2; }
    if (N <= 0 || H < 2 || W < 2) { octa::set_error("octa_conv4x4_nhwc_wgrad: bad shape"); return -2; }
    if (Cin % 32 || Cout % 32 || Cin <= 0 || Cout <= 0) { octa::set_error("octa_conv4x4_nhwc_wgrad: Cin and Cout must be multiples of 32 (got %d, %d)", Cin, Cout); return -2; }
    hipStream_t stream = (hipStream_t)stream_;
    OCTA_HIP_CHECK(hipSetDevice(ctx->device));
    OCTA_HIP_CHECK(hipMemsetAsync(d_dw, 0, sizeof(float) * 16 * (size_t)Cout * Cin, stream));
    const unsigned short *z = zero_page(ctx);
    if (!z) return -1;
    const unsigned short *X = static_cast<const unsigned short *>(d_x), *dY = static_cast<const unsigned short *>(d_dy);
    if (Cout % 64 == 0 && Cin % 64 == 0)
        return launch_wgrad_impl<64, 64, false, 1, 4>(X, nullptr, Cin, dY, d_dw, N, H, W, Cin, Cout, ctx->num_cus, 0xffff, nullptr, nullptr, nullptr, nullptr, 0.f, z, stream);
    return launch_wgrad_impl<32, 32, false, 1, 4>(X, nullptr, Cin, dY, d_dw, N, H, W, Cin, Cout, ctx->num_cus, 0xffff, nullptr, nullptr, nullptr, nullptr, 0.f, z, stream);
}

extern "C" int octa_conv3x3_nhwc_wgrad3(octa_ctx *ctx, const void *d_x, const void *d_x2, int C1, const void *d_dy, float *d_dw, int N, int H,
                                        int W, int Cin, int Cout, int tap_mask, const float *d_scale1, const float *d_shift1,
                                        const float *d_scale2, const float *d_shift2, float slope, void *stream_) {
    return octa_conv3x3_nhwc_wgrad4(ctx, d_x, d_x2, C1, d_dy, d_dw, N, H, W, Cin, Cout, 1, tap_mask, d_scale1, d_shift1, d_scale2, d_shift2, slope, stream_);
}

extern "C" int octa_conv3x3_nhwc_wgrad2(octa_ctx *ctx, const void *d_x, const void *d_x2, int C1, const void *d_dy, float *d_dw, int N, int H,
                                        int W, int Cin, int Cout, int tap_mask, void *stream_) {
    return octa_conv3x3_nhwc_wgrad3(ctx, d_x, d_x2, C1, d_dy, d_dw, N, H, W, Cin, Cout, tap_mask, nullptr, nullptr, nullptr, nullptr, 0.f, stream_);
}

extern "C" int octa_conv3x3_nhwc_wgrad(octa_ctx *ctx, const void *d_x, const void *d_dy, float *d_dw, int N, int H, int W, int Cin,
                                       int Cout, void *stream_) {
    return octa_conv3x3_nhwc_wgrad2(ctx, d_x, nullptr, Cin, d_dy, d_dw, N, H, W, Cin, Cout, 0x1ff, stream_);
}

// ---- 1x1 head with one output channel (UnetOutBlock, 32 -> 1 with bias): HBM-bound streaming kernels ----------
namespace {

__device__ __forceinline__ float bf2f(unsigned short h) { return __uint_as_float((unsigned)h << 16); }

// y[p] = b + sum_c x[p][c] w[c];  C multiple of 8, <= 256. One thread per pixel group of 16 bytes x (C/8).
__global__ void __launch_bounds__(256)
head1_fwd_kernel(const unsigned short *__restrict__ x, const float *__restrict__ w, float bias, const float *__restrict__ bias_p,
                 long npix, int C, unsigned short *__restrict__ y) {
    __shared__ float s_w[256];
    for (int c = threadIdx.x; c < C; c += 256) s_w[c] = w[c];
    __syncthreads();
    if (bias_p) bias += *bias_p;
    for (long p = (long)blockIdx.x * 256 + threadIdx.x; p < npix; p += (long)gridDim.x * 256) {
        float acc = bias;
        const uint4 *px = reinterpret_cast<const uint4 *>(x + p * C);
        for (int g = 0; g < C / 8; g++) {
            const uint4 v = px[g];
            const unsigned u[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
            for (int k = 0; k < 4; k++) {
                acc += __uint_as_float(u[k] << 16) * s_w[g * 8 + 2 * k];
                acc += __uint_as_float(u[k] & 0xffff0000u) * s_w[g * 8 + 2 * k + 1];
            }
        }
        y[p] = f2bf(acc);
    }
}

// dx[p][c] = dy[p] w[c];  dw[c] += sum_p x[p][c] dy[p];  db += sum_p dy[p]
__global__ void __launch_bounds__(256)
head1_bwd_kernel(const unsigned short *__restrict__ x, const unsigned short *__restrict__ dy, const float *__restrict__ w, long npix,
                 int C, unsigned short *__restrict__ dx, float *__restrict__ dw, float *__restrict__ db) {
    __shared__ float s_w[256];
    __shared__ float s_acc[257];
    for (int c = threadIdx.x; c < C; c += 256) { s_w[c] = w[c]; s_acc[c] = 0.f; }
    if (threadIdx.x == 0) s_acc[256] = 0.f;
    __syncthreads();
    // thread = (pixel lane, 8-channel group): consecutive threads cover one pixel's channels (coalesced)
    const int groups = C / 8, cg = threadIdx.x % groups, pl = threadIdx.x / groups, npl = 256 / groups;
    float a[8];
#pragma unroll
    for (int k = 0; k < 8; k++) a[k] = 0.f;
    float sdy = 0.f;
    if (pl < npl) {
        for (long p = (long)blockIdx.x * npl + pl; p < npix; p += (long)gridDim.x * npl) {
            const float d = bf2f(dy[p]);
            const uint4 v = *reinterpret_cast<const uint4 *>(x + p * C + cg * 8);
            const unsigned u[4] = {v.x, v.y, v.z, v.w};
            unsigned o[4];
#pragma unroll
            for (int k = 0; k < 4; k++) {
                a[2 * k] += __uint_as_float(u[k] << 16) * d;
                a[2 * k + 1] += __uint_as_float(u[k] & 0xffff0000u) * d;
                o[k] = octa_pack_bf16x2(d * s_w[cg * 8 + 2 * k], d * s_w[cg * 8 + 2 * k + 1]);
            }
            *reinterpret_cast<uint4 *>(dx + p * C + cg * 8) = make_uint4(o[0], o[1], o[2], o[3]);
            if (cg == 0) sdy += d;
        }
#pragma unroll
        for (int k = 0; k < 8; k++) atomicAdd(&s_acc[cg * 8 + k], a[k]);
        if (cg == 0) atomicAdd(&s_acc[256], sdy);
    }
    __syncthreads();
    for (int c = threadIdx.x; c < C; c += 256) atomicAdd(&dw[c], s_acc[c]);
    if (threadIdx.x == 0) atomicAdd(db, s_acc[256]);
}

}  // namespace

extern "C" int octa_head1_nhwc_fwd(octa_ctx *ctx, const void *d_x, const float *d_w, float bias, int64_t npix, int C, void *d_y,
                                   void *stream_) {
    if (!ctx || !d_x || !d_w || !d_y || npix <= 0 || C <= 0 || C % 8 || C > 256) { octa::set_error("octa_head1_nhwc_fwd: bad arguments (C must be a multiple of 8, <= 256)"); return -2; }
    hipStream_t stream = (hipStream_t)stream_;
    OCTA_HIP_CHECK(hipSetDevice(ctx->device));
    long blocks = (npix + 255) / 256;
    if (blocks > 16L * ctx->num_cus) blocks = 16L * ctx->num_cus;
    hipLaunchKernelGGL(head1_fwd_kernel, dim3((unsigned)blocks), dim3(256), 0, stream, static_cast<const unsigned short *>(d_x), d_w, bias,
                       (const float *)nullptr, (long)npix, C, static_cast<unsigned short *>(d_y));
    OCTA_HIP_CHECK(hipGetLastError());
    return 0;
}

extern "C" int octa_head1_nhwc_bwd(octa_ctx *ctx, const void *d_x, const void *d_dy, const float *d_w, int64_t npix, int C, void *d_dx,
                                   float *d_dw, float *d_db, void *stream_) {
    if (!ctx || !d_x || !d_dy || !d_w || !d_dx || !d_dw || !d_db || npix <= 0 || C <= 0 || C % 8 || C > 256 || 256 % (C / 8)) {
        octa::set_error("octa_head1_nhwc_bwd: bad arguments (C must be a multiple of 8 dividing 2048, <= 256)");
        return -2;
    }
    hipStream_t stream = (hipStream_t)stream_;
    OCTA_HIP_CHECK(hipSetDevice(ctx->device));
    OCTA_HIP_CHECK(hipMemsetAsync(d_dw, 0, sizeof(float) * C, stream));
    OCTA_HIP_CHECK(hipMemsetAsync(d_db, 0, sizeof(float), stream));
    const int npl = 256 / (C / 8);
    long blocks = (npix + npl - 1) / npl;
    if (blocks > 8L * ctx->num_cus) blocks = 8L * ctx->num_cus;
    hipLaunchKernelGGL(head1_bwd_kernel, dim3((unsigned)blocks), dim3(256), 0, stream, static_cast<const unsigned short *>(d_x),
                       static_cast<const unsigned short *>(d_dy), d_w, (long)npix, C, static_cast<unsigned short *>(d_dx), d_dw, d_db);
    OCTA_HIP_CHECK(hipGetLastError());
    return 0;
}

// ---- first layer: ONE input channel (the grey image) -> Cout channels, 3x3, padding 1, stride 1 ------------------------
// 9 multiply-adds per output value: not matrix-core work. Forward and weight gradient are single streaming passes over the Cout-channel
// tensor (HBM-bound) instead of padding the image to 32 zero channels for the MFMA kernels. Round 5 rewrite of both (rounds 1-4: 333
// vector instructions per 8-channel item around 72 FMAs -- bounds tests and 64-bit addresses of nine 2-byte global loads per item, the
// weights read from LDS per FMA; 0.178 ms forward / 0.205 ms weight gradient at 4 x 1216^2 x 32, i.e. 2 TB/s): a workgroup now owns a
// band of image rows (c1_band()), stages band + halo ONCE as floats with zero borders in LDS (16-byte global loads), a thread owns 8 output
// channels (weights / accumulators in registers) of pixels 64 / groups apart, so a wave's 16-byte stores / loads of the big tensor are 1 KB
// contiguous and the inner loop is LDS reads + FMAs only.
namespace {

__device__ __forceinline__ float bfu(unsigned short h) { return __uint_as_float((unsigned)h << 16); }
typedef float f32x2 __attribute__((ext_vector_type(2)));

constexpr int C1_BAND_MAX = 8;   // image rows per workgroup, at most
// Rows per workgroup: the launch should fill the chip's resident slots once (four 256-thread workgroups per CU by registers) -- the kernels
// stream, so waves in flight are bytes in flight (4 x 1216^2: 8 rows = 608 workgroups = 2.4 waves per SIMD, 5 rows = 976 = 3.8)
int c1_band(const octa_ctx *ctx, int N, int H) {
    long b = ((long)N * H + 4L * ctx->num_cus - 1) / (4L * ctx->num_cus);
    return (int)(b < 2 ? 2 : (b > C1_BAND_MAX ? C1_BAND_MAX : b));
}

// rows y0-1 .. y0+rows of image n as floats into s_x[(rows + 2)][W + 2] (column 0 / W+1 and rows outside the image: zero)
__device__ __forceinline__ void c1_stage_band(const unsigned short *__restrict__ X, float *s_x, long n, int y0, int rows, int H, int W) {
    const int WP = W + 2;
    if (W % 8 == 0) {
        const int pieces = W / 8;
        for (int i = threadIdx.x; i < (rows + 2) * pieces; i += 256) {
            const int r = i / pieces, xq = i % pieces, yy = y0 + r - 1;
            uint4 v = make_uint4(0, 0, 0, 0);
            if (yy >= 0 && yy < H) v = *reinterpret_cast<const uint4 *>(X + (n * H + yy) * W + xq * 8);
            const unsigned u[4] = {v.x, v.y, v.z, v.w};
            float *d = s_x + r * WP + 1 + xq * 8;
#pragma unroll
            for (int k = 0; k < 4; k++) { d[2 * k] = __uint_as_float(u[k] << 16); d[2 * k + 1] = __uint_as_float(u[k] & 0xffff0000u); }
        }
        for (int i = threadIdx.x; i < (rows + 2) * 2; i += 256) s_x[(i >> 1) * WP + (i & 1) * (W + 1)] = 0.f;
    } else {
        for (int i = threadIdx.x; i < (rows + 2) * WP; i += 256) {
            const int r = i / WP, xx = i % WP - 1, yy = y0 + r - 1;
            s_x[i] = (yy >= 0 && yy < H && xx >= 0 && xx < W) ? bfu(X[(n * H + yy) * W + xx]) : 0.f;
        }
    }
}

// stat != nullptr: also the InstanceNorm statistics of the result -- sum and sum of squares of the bf16-ROUNDED values per image and channel,
// added (double atomics) to slot blockIdx.x % nslot of stat[nslot][N][Cout][2] (pre-zeroed; the slot form of conv3x3_nhwc_glds_kernel): the
// norm that follows needs no statistics pass over the tensor.
__global__ void __launch_bounds__(256)
conv3x3_c1_fwd_kernel(const unsigned short *__restrict__ X, const float *__restrict__ Wf /* [Cout][9] */, unsigned short *__restrict__ Y,
                      int H, int W, int Cout, double *__restrict__ stat, int nslot, int C1_BAND) {
    extern __shared__ float s_mem[];
    float *s_x = s_mem;                                  // [C1_BAND + 2][W + 2]
    const int groups = Cout / 8, gshift = 31 - __clz(groups);   // groups is a power of two dividing 64
    const int q = threadIdx.x & (groups - 1), pix0 = threadIdx.x >> gshift, ppi = 256 >> gshift;   // pixels per block-iteration
    const long n = blockIdx.y;
    const int y0 = blockIdx.x * C1_BAND, rows = H - y0 < C1_BAND ? H - y0 : C1_BAND, WP = W + 2;
    // channel pairs as 2-vectors: v_pk_fma_f32 multiplies both by the (broadcast) pixel in one instruction -- 36 instead of 72 per item
    f32x2 wr[4][9];
#pragma unroll
    for (int k = 0; k < 4; k++)
#pragma unroll
        for (int t = 0; t < 9; t++) { wr[k][t].x = Wf[(q * 8 + 2 * k) * 9 + t]; wr[k][t].y = Wf[(q * 8 + 2 * k + 1) * 9 + t]; }
    c1_stage_band(X, s_x, n, y0, rows, H, W);
    __syncthreads();
    float s1[8], s2[8];
#pragma unroll
    for (int k = 0; k < 8; k++) { s1[k] = 0.f; s2[k] = 0.f; }
    for (int rr = 0; rr < rows; rr++) {
        const float *sx = s_x + rr * WP;
        unsigned short *yrow = Y + ((n * H + y0 + rr) * W) * Cout + q * 8;
        for (int x = pix0; x < W; x += ppi) {
            float in[9];
#pragma unroll
            for (int r = 0; r < 3; r++)
#pragma unroll
                for (int c = 0; c < 3; c++) in[3 * r + c] = sx[r * WP + x + c];
            unsigned o[4];
#pragma unroll
            for (int k = 0; k < 4; k++) {
                f32x2 a = {0.f, 0.f};
#pragma unroll
                for (int t = 0; t < 9; t++) { const f32x2 i2 = {in[t], in[t]}; a = __builtin_elementwise_fma(i2, wr[k][t], a); }
                o[k] = octa_pack_bf16x2(a.x, a.y);
                const float r0 = __uint_as_float(o[k] << 16), r1 = __uint_as_float(o[k] & 0xffff0000u);
                s1[2 * k] += r0; s2[2 * k] += r0 * r0; s1[2 * k + 1] += r1; s2[2 * k + 1] += r1 * r1;
            }
            *reinterpret_cast<uint4 *>(yrow + (long)x * Cout) = make_uint4(o[0], o[1], o[2], o[3]);
        }
    }
    if (stat) {
        // lanes with the same channel group: lane % groups; then the four waves through LDS, one pair of double atomics per channel
        __syncthreads();                                  // the band is consumed: its LDS is reused
        float *s_red = s_mem;                             // [4 waves][Cout][2]
#pragma unroll
        for (int k = 0; k < 8; k++) {
            float a = s1[k], b = s2[k];
            for (int dlt = groups; dlt < 64; dlt <<= 1) { a += __shfl_xor(a, dlt, 64); b += __shfl_xor(b, dlt, 64); }
            if ((int)(threadIdx.x & 63) < groups) { s_red[((threadIdx.x >> 6) * Cout + q * 8 + k) * 2] = a; s_red[((threadIdx.x >> 6) * Cout + q * 8 + k) * 2 + 1] = b; }
        }
        __syncthreads();
        if ((int)threadIdx.x < Cout) {
            float a = 0.f, b = 0.f;
#pragma unroll
            for (int w4 = 0; w4 < 4; w4++) { a += s_red[(w4 * Cout + threadIdx.x) * 2]; b += s_red[(w4 * Cout + threadIdx.x) * 2 + 1]; }
            double *dst = stat + (((size_t)(blockIdx.x % nslot) * gridDim.y + n) * Cout + threadIdx.x) * 2;
            atomicAdd(dst, (double)a);
            atomicAdd(dst + 1, (double)b);
        }
    }
}

// dW[co][t] = sum over the pixels of dY[p][co] * X[p + t]: per workgroup one band; partial sums [band][Cout][9] go to a workspace and are
// folded by c1_wgrad_reduce_kernel (rounds 1-4: one atomicAdd per workgroup and weight into 288 addresses -- 1216 workgroups queueing on each)
__global__ void __launch_bounds__(256)
conv3x3_c1_wgrad_kernel(const unsigned short *__restrict__ X, const unsigned short *__restrict__ dY, float *__restrict__ ws /* [gridDim.y * gridDim.x][Cout][9] */,
                        int H, int W, int Cout, int C1_BAND) {
    extern __shared__ float s_mem[];
    float *s_x = s_mem;                                  // [C1_BAND + 2][W + 2]
    const int groups = Cout / 8, gshift = 31 - __clz(groups);
    const int q = threadIdx.x & (groups - 1), pix0 = threadIdx.x >> gshift, ppi = 256 >> gshift;
    const long n = blockIdx.y;
    const int y0 = blockIdx.x * C1_BAND, rows = H - y0 < C1_BAND ? H - y0 : C1_BAND, WP = W + 2;
    f32x2 acc[4][9];                                     // channel pairs: v_pk_fma_f32
#pragma unroll
    for (int k = 0; k < 4; k++)
#pragma unroll
        for (int t = 0; t < 9; t++) acc[k][t] = f32x2{0.f, 0.f};
    c1_stage_band(X, s_x, n, y0, rows, H, W);
    __syncthreads();
    for (int rr = 0; rr < rows; rr++) {
        const float *sx = s_x + rr * WP;
        const unsigned short *drow = dY + ((n * H + y0 + rr) * W) * Cout + q * 8;
        // four 16-byte pieces of dY per trip, all loads issued before any is used (three workgroups of 48 KB per CU: the bytes in flight
        // per wave decide how much of the HBM latency is covered)
        for (int x = pix0; x < W; x += 4 * ppi) {
            uint4 v[4];
#pragma unroll
            for (int h = 0; h < 4; h++) { const int xx = x + h * ppi; v[h] = *reinterpret_cast<const uint4 *>(drow + (long)(xx < W ? xx : x) * Cout); }
#pragma unroll
            for (int h = 0; h < 4; h++) {
                const int xx = x + h * ppi;
                if (xx >= W) break;
                const unsigned u[4] = {v[h].x, v[h].y, v[h].z, v[h].w};
                f32x2 d[4];
#pragma unroll
                for (int k = 0; k < 4; k++) { d[k].x = __uint_as_float(u[k] << 16); d[k].y = __uint_as_float(u[k] & 0xffff0000u); }
#pragma unroll
                for (int r = 0; r < 3; r++)
#pragma unroll
                    for (int c = 0; c < 3; c++) {
                        const float in = sx[r * WP + xx + c];
                        const f32x2 i2 = {in, in};
#pragma unroll
                        for (int k = 0; k < 4; k++) acc[k][3 * r + c] = __builtin_elementwise_fma(d[k], i2, acc[k][3 * r + c]);
                    }
            }
        }
    }
    __syncthreads();
    float *s_red = s_mem;                                 // [4 waves][Cout][9]
#pragma unroll
    for (int k = 0; k < 8; k++)
#pragma unroll
        for (int t = 0; t < 9; t++) {
            float a = (k & 1) ? acc[k >> 1][t].y : acc[k >> 1][t].x;
            for (int dlt = groups; dlt < 64; dlt <<= 1) a += __shfl_xor(a, dlt, 64);
            if ((int)(threadIdx.x & 63) < groups) s_red[((threadIdx.x >> 6) * Cout + q * 8 + k) * 9 + t] = a;
        }
    __syncthreads();
    float *mine = ws + ((size_t)blockIdx.y * gridDim.x + blockIdx.x) * Cout * 9;
    for (int i = threadIdx.x; i < Cout * 9; i += 256) mine[i] = (s_red[i] + s_red[Cout * 9 + i]) + (s_red[2 * Cout * 9 + i] + s_red[3 * Cout * 9 + i]);
}

// dW[e] = sum of the bands' partial sums: 16 outputs x 16 slices of the partials per workgroup (a thread per output alone walks a
// dependent chain of 600 loads: 43 us)
__global__ void __launch_bounds__(256)
c1_wgrad_reduce_kernel(const float *__restrict__ ws, float *__restrict__ dW, int total, int parts) {
    __shared__ float s_p[256];
    const int el = threadIdx.x & 15, sl = threadIdx.x >> 4, e = blockIdx.x * 16 + el;
    float s0 = 0.f, s1 = 0.f;
    if (e < total) {
        int g = sl;
        for (; g + 16 < parts; g += 32) { s0 += ws[(size_t)g * total + e]; s1 += ws[(size_t)(g + 16) * total + e]; }
        if (g < parts) s0 += ws[(size_t)g * total + e];
    }
    s_p[threadIdx.x] = s0 + s1;
    __syncthreads();
    if (threadIdx.x < 16 && e < total) {
        float a = 0.f;
#pragma unroll
        for (int k = 0; k < 16; k++) a += s_p[k * 16 + threadIdx.x];
        dW[e] = a;
    }
}

}  // namespace

// d_stat / nslot: optional InstanceNorm statistics of the result in slot form (see conv3x3_c1_fwd_kernel; octa_conv3x3_nhwc_fwd7's contract)
extern "C" int octa_conv3x3_c1_fwd2(octa_ctx *ctx, const void *d_x, const float *d_w, void *d_y, int N, int H, int W, int Cout, double *d_stat,
                                    int nslot, void *stream_) {
    if (!ctx || !d_x || !d_w || !d_y || N <= 0 || H <= 0 || W <= 0) { octa::set_error("octa_conv3x3_c1_fwd: bad arguments"); return -2; }
    if (Cout != 8 && Cout != 16 && Cout != 32 && Cout != 64) { octa::set_error("octa_conv3x3_c1_fwd: Cout must be 8, 16, 32 or 64"); return -2; }
    if (W > 3840 || N > 65535) { octa::set_error("octa_conv3x3_c1_fwd: W > 3840 or N > 65535"); return -2; }
    if (d_stat && nslot <= 0) { octa::set_error("octa_conv3x3_c1_fwd: statistics need nslot >= 1"); return -2; }
    hipStream_t stream = (hipStream_t)stream_;
    OCTA_HIP_CHECK(hipSetDevice(ctx->device));
    const int band = c1_band(ctx, N, H);
    size_t lds = sizeof(float) * (size_t)(band + 2) * (W + 2);
    if (lds < sizeof(float) * 4 * Cout * 2) lds = sizeof(float) * 4 * Cout * 2;
    OCTA_HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void *>(conv3x3_c1_fwd_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    hipLaunchKernelGGL(conv3x3_c1_fwd_kernel, dim3((unsigned)((H + band - 1) / band), (unsigned)N), dim3(256), lds, stream,
                       static_cast<const unsigned short *>(d_x), d_w, static_cast<unsigned short *>(d_y), H, W, Cout, d_stat, nslot, band);
    OCTA_HIP_CHECK(hipGetLastError());
    return 0;
}

extern "C" int octa_conv3x3_c1_fwd(octa_ctx *ctx, const void *d_x, const float *d_w, void *d_y, int N, int H, int W, int Cout, void *stream_) {
    return octa_conv3x3_c1_fwd2(ctx, d_x, d_w, d_y, N, H, W, Cout, nullptr, 0, stream_);
}

extern "C" int octa_conv3x3_c1_wgrad(octa_ctx *ctx, const void *d_x, const void *d_dy, float *d_dw, int N, int H, int W, int Cout, void *stream_) {
    if (!ctx || !d_x || !d_dy || !d_dw || N <= 0 || H <= 0 || W <= 0) { octa::set_error("octa_conv3x3_c1_wgrad: bad arguments"); return -2; }
    if (Cout != 8 && Cout != 16 && Cout != 32 && Cout != 64) { octa::set_error("octa_conv3x3_c1_wgrad: Cout must be 8, 16, 32 or 64"); return -2; }
    if (W > 3840 || N > 65535) { octa::set_error("octa_conv3x3_c1_wgrad: W > 3840 or N > 65535"); return -2; }
    hipStream_t stream = (hipStream_t)stream_;
    OCTA_HIP_CHECK(hipSetDevice(ctx->device));
    const int band = c1_band(ctx, N, H), bands = (H + band - 1) / band, parts = bands * N, total = Cout * 9;
    if (ctx->wgrad_ws.reserve((size_t)parts * total * sizeof(float))) return -1;
    float *ws = ctx->wgrad_ws.as<float>();
    size_t lds = sizeof(float) * (size_t)(band + 2) * (W + 2);
    if (lds < sizeof(float) * 4 * total) lds = sizeof(float) * 4 * total;
    OCTA_HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void *>(conv3x3_c1_wgrad_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    hipLaunchKernelGGL(conv3x3_c1_wgrad_kernel, dim3((unsigned)bands, (unsigned)N), dim3(256), lds, stream, static_cast<const unsigned short *>(d_x),
                       static_cast<const unsigned short *>(d_dy), ws, H, W, Cout, band);
    OCTA_HIP_CHECK(hipGetLastError());
    hipLaunchKernelGGL(c1_wgrad_reduce_kernel, dim3((unsigned)((total + 15) / 16)), dim3(256), 0, stream, ws, d_dw, total, parts);
    OCTA_HIP_CHECK(hipGetLastError());
    return 0;
}

// ---- every KxK weight of a network into both MFMA layouts, ONE launch per optimiser step ---------------------------
// The master weights stay torch parameters (float32 [Cout][Cin][K][K]; MONAI / networks.py state-dict layout); the
// kernels want bf16 [K*K][Cout][CinP] (forward) and [K*K][CinP][Cout] with the taps reversed (data gradient), both in
// the slice-major storage order of wt_off(). Packing layer by layer with torch ops costs ~7 tiny launches per layer and step (flip, permute-copy,
// cast, zero fill: 0.66 ms of the 20.3 ms U-Net step, profiles/r01_train_mfma_kernel_stats.csv); here a table of
// descriptors drives one launch. HBM-bound, ~6 B per weight element; nothing to tile.
namespace {

struct PackDesc {          // one row of the int64 table [L][8]
    long src;              // const float *: [A][B][K][K], A = Cout, B = Cin (kind 0); kind 1: ConvTranspose2d(k = s = 2)
                           //   weight [A][B][2][2] read as the 3x3 kernel wc[a][b][r][s] = w[a][b][r-1][s-1] (r, s >= 1)
    long off_fwd;          // element offset in dst of [KK][A][BP]
    long off_dg;           // element offset in dst of [KK][BP][A], taps reversed
    long A, B, BP, KK, kind;
};

__device__ __forceinline__ float pack_src(const float *w, long a, long b, long B, int t, int KK, int kind) {
    if (kind == 0) return w[(a * B + b) * KK + t];
    const int r = t / 3, s = t % 3;                      // kind 1: KK == 9 over a 2x2 source
    return (r >= 1 && s >= 1) ? w[(a * B + b) * 4 + (r - 1) * 2 + (s - 1)] : 0.f;
}

__global__ void __launch_bounds__(256)
pack_weights_kernel(const long *__restrict__ table, unsigned short *__restrict__ dst) {
    const PackDesc d = *reinterpret_cast<const PackDesc *>(table + 8 * blockIdx.y);
    const float *w = reinterpret_cast<const float *>(d.src);
    const long pairs = d.A * d.BP;
    const int KK = (int)d.KK, kind = (int)d.kind;
    // forward layout: b fastest -> coalesced bf16 stores per tap, K*K contiguous floats read per thread
    for (long e = (long)blockIdx.x * 256 + threadIdx.x; e < pairs; e += (long)gridDim.x * 256) {
        const long a = e / d.BP, b = e % d.BP;
        for (int t = 0; t < KK; t++)
            dst[d.off_fwd + (long)wt_off(t, (int)a, (int)b, KK, (int)d.A)] = b < d.B ? f2bf(pack_src(w, a, b, d.B, t, KK, kind)) : (unsigned short)0;
    }
    // data-gradient layout: a fastest (source re-read through L2, stores coalesced), taps reversed. Its "input channel" is a: a layer whose A is
    // no multiple of 16 (the PatchGAN's 512 -> 1 head) has no MFMA data gradient and keeps the tap-major order [KK][BP][A]
    const bool dg_sliced = d.A % 16 == 0;
    for (long e = (long)blockIdx.x * 256 + threadIdx.x; e < pairs; e += (long)gridDim.x * 256) {
        const long b = e / d.A, a = e % d.A;
        for (int t = 0; t < KK; t++)
            dst[d.off_dg + (dg_sliced ? (long)wt_off(KK - 1 - t, (int)b, (int)a, KK, (int)d.BP) : (long)(KK - 1 - t) * pairs + e)] =
                b < d.B ? f2bf(pack_src(w, a, b, d.B, t, KK, kind)) : (unsigned short)0;
    }
}

}  // namespace

extern "C" int octa_pack_conv_weights(octa_ctx *ctx, const int64_t *d_table, int L, void *d_dst, void *stream_) {
    if (!ctx || !d_table || !d_dst || L <= 0 || L > 65535) { octa::set_error("octa_pack_conv_weights: bad arguments"); return -2; }
    hipStream_t stream = (hipStream_t)stream_;
    OCTA_HIP_CHECK(hipSetDevice(ctx->device));
    // 256 workgroups per layer: the launch lasts as long as the largest layer's share (512 x 512 x 9: 85 us with 64 workgroups on it, one per step)
    hipLaunchKernelGGL(pack_weights_kernel, dim3(256, (unsigned)L), dim3(256), 0, stream, reinterpret_cast<const long *>(d_table),
                       static_cast<unsigned short *>(d_dst));
    OCTA_HIP_CHECK(hipGetLastError());
    return 0;
}

// head forward with the bias read from device memory (no host read-back of the parameter in the training step)
extern "C" int octa_head1_nhwc_fwd_b(octa_ctx *ctx, const void *d_x, const float *d_w, const float *d_bias, int64_t npix, int C, void *d_y,
                                     void *stream_) {
    if (!ctx || !d_x || !d_w || !d_y || npix <= 0 || C <= 0 || C % 8 || C > 256) { octa::set_error("octa_head1_nhwc_fwd_b: bad arguments (C must be a multiple of 8, <= 256)"); return -2; }
    hipStream_t stream = (hipStream_t)stream_;
    OCTA_HIP_CHECK(hipSetDevice(ctx->device));
    long blocks = (npix + 255) / 256;
    if (blocks > 16L * ctx->num_cus) blocks = 16L * ctx->num_cus;
    hipLaunchKernelGGL(head1_fwd_kernel, dim3((unsigned)blocks), dim3(256), 0, stream, static_cast<const unsigned short *>(d_x), d_w, 0.f, d_bias,
                       (long)npix, C, static_cast<unsigned short *>(d_y));
    OCTA_HIP_CHECK(hipGetLastError());
    return 0;
}
