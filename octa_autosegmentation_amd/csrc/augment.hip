// augment.hip -- the GPU data-augmentation chain between the rasteriser and the network.
//
// Replaces, for a whole batch resident in HBM, the MONAI CPU transforms of the training configs
// (configs/config_ves_seg-S.yml:42-102, built by data/data_transforms.py:587-611 get_data_augmentations):
//   ScaleIntensityd(0, 1) -> Resized(1216, 1216, bilinear) -> RandFlipd -> RandRotate90d -> RandRotated(+-10 deg,
//   bilinear, zeros) -> AsDiscreted(label, 0.1) -> CastToTyped.
// Two streaming kernels (HBM-bound; algorithmic bytes = one read of the source + one write of the result):
//   resize_bilinear_kernel : uint8 / float32 [B][h][w] -> float32 [B][H][W], torch's upsample_bilinear2d arithmetic
//                            (align_corners = False), with the per-image affine intensity map of ScaleIntensity folded in;
//   rotate_kernel          : float32 [B][H][W] -> float32 [B][H][W]: rotation by a per-image angle about the image centre
//                            (F.affine_grid + F.grid_sample, bilinear, zeros padding, align_corners = False arithmetic),
//                            the exact index permutations of the preceding flip / rot90 folded into the tap addresses,
//                            optional threshold (AsDiscrete) on the way out.
// MONAI itself is not in the image (parity with it is unpinned); the kernels are pinned against the torch
// ops MONAI delegates to (tests/test_augment_gpu.py).

#include "common.h"

namespace {

template <class T> __device__ __forceinline__ float ldf(const T *p);
template <> __device__ __forceinline__ float ldf<unsigned char>(const unsigned char *p) { return (float)*p; }
template <> __device__ __forceinline__ float ldf<float>(const float *p) { return *p; }

// torch area_pixel_compute_source_index(scale, dst, align_corners = false, cubic = false)
__device__ __forceinline__ float src_index(float scale, int dst) {
    const float s = scale * ((float)dst + 0.5f) - 0.5f;
    return s < 0.f ? 0.f : s;
}

template <class T>
__global__ void __launch_bounds__(256)
resize_bilinear_kernel(const T *__restrict__ in, int h, int w, float *__restrict__ out, int H, int W, const float *__restrict__ mul,
                       const float *__restrict__ add) {
    const int b = blockIdx.z, Y = blockIdx.y, X = blockIdx.x * 256 + threadIdx.x;
    if (X >= W) return;
    const float rh = (float)h / (float)H, rw = (float)w / (float)W;
    const float h1r = src_index(rh, Y), w1r = src_index(rw, X);
    const int h1 = (int)h1r, w1 = (int)w1r;
    const int h1p = h1 < h - 1 ? 1 : 0, w1p = w1 < w - 1 ? 1 : 0;
    const float h1l = h1r - (float)h1, h0l = 1.f - h1l, w1l = w1r - (float)w1, w0l = 1.f - w1l;
    const T *p = in + ((size_t)b * h + h1) * w + w1;
    const float m = mul ? mul[b] : 1.f, a = add ? add[b] : 0.f;
    const float p00 = ldf(p) * m + a, p01 = ldf(p + w1p) * m + a, p10 = ldf(p + (size_t)h1p * w) * m + a, p11 = ldf(p + (size_t)h1p * w + w1p) * m + a;
    out[((size_t)b * H + Y) * W + X] = h0l * (w0l * p00 + w1l * p01) + h1l * (w0l * p10 + w1l * p11);
}

// Adjoint of resize_bilinear_kernel (identity intensity map): dx[iy][ix] = sum over the output pixels whose two-by-two taps include
// (iy, ix) of their weight for it times dy. Gather form: a thread owns one SOURCE pixel and walks the few output rows / columns that
// can name it (no atomics, fixed summation order); the weights are recomputed with the forward's own expressions, so forward and
// adjoint agree to the last bit of every weight (GanSegModel's 304 -> 1216 up-sampling in front of the segmentor,
// models/gan_seg_model.py:61,101-106).
__device__ __forceinline__ float tap_weight(float ratio, int src, int n_src, int dst) {      // weight of source index `src` in output index `dst`
    const float r = src_index(ratio, dst);
    const int i1 = (int)r;
    const int p = i1 < n_src - 1 ? 1 : 0;
    const float l1 = r - (float)i1, l0 = 1.f - l1;
    float wgt = 0.f;
    if (i1 == src) wgt += l0;
    if (i1 + p == src) wgt += l1;
    return wgt;
}
__global__ void __launch_bounds__(256)
resize_bilinear_bwd_kernel(const float *__restrict__ dy, int h, int w, int H, int W, float *__restrict__ dx) {
    const int b = blockIdx.z, iy = blockIdx.y, ix = blockIdx.x * 256 + threadIdx.x;
    if (ix >= w) return;
    const float rh = (float)h / (float)H, rw = (float)w / (float)W;
    // output indices that can touch source index i: src_index(dst) in (i - 1, i + 1)  =>  dst in ((i - 0.5) / ratio - 0.5, (i + 1.5) / ratio - 0.5)
    int y0 = (int)floorf(((float)iy - 0.5f) / rh - 0.5f) - 1, y1 = (int)ceilf(((float)iy + 1.5f) / rh - 0.5f) + 1;
    int x0 = (int)floorf(((float)ix - 0.5f) / rw - 0.5f) - 1, x1 = (int)ceilf(((float)ix + 1.5f) / rw - 0.5f) + 1;
    y0 = y0 < 0 ? 0 : y0; x0 = x0 < 0 ? 0 : x0;
    y1 = y1 > H - 1 ? H - 1 : y1; x1 = x1 > W - 1 ? W - 1 : x1;
    if (iy == 0) y0 = 0;                 // clamped source coordinates: every output row above the first source row maps onto it
    if (ix == 0) x0 = 0;
    float acc = 0.f;
    const float *g = dy + (size_t)b * H * W;
    for (int Y = y0; Y <= y1; Y++) {
        const float wy = tap_weight(rh, iy, h, Y);
        if (wy == 0.f) continue;
        float row = 0.f;
        for (int X = x0; X <= x1; X++) {
            const float wx = tap_weight(rw, ix, w, X);
            if (wx != 0.f) row += wx * g[(size_t)Y * W + X];
        }
        acc += wy * row;
    }
    dx[((size_t)b * h + iy) * w + ix] = acc;
}

// One tap of the rotated image's source: (iy, ix) indexes the image AFTER flip and rot90; map it back to the stored one.
// torch.rot90(x, k, (H, W)) on a square image: k = 1: out[i][j] = in[j][N-1-i]; k = 2: in[N-1-i][N-1-j]; k = 3: in[N-1-j][i].
__device__ __forceinline__ float tap(const float *img, int N, int iy, int ix, int k, int flip) {
    if (iy < 0 || iy >= N || ix < 0 || ix >= N) return 0.f;   // zeros padding
    int y = iy, x = ix;
    if (k == 1) { y = ix; x = N - 1 - iy; }
    else if (k == 2) { y = N - 1 - iy; x = N - 1 - ix; }
    else if (k == 3) { y = N - 1 - ix; x = iy; }
    if (flip) { y = N - 1 - y; x = N - 1 - x; }
    return img[(size_t)y * N + x];
}

// theta = [[c, -s, 0], [s, c, 0]] in grid_sample's normalised coordinates (x first), align_corners = False.
__global__ void __launch_bounds__(256)
rotate_kernel(const float *__restrict__ in, float *__restrict__ out, int N, const float *__restrict__ angle, const int *__restrict__ rot_k,
              const int *__restrict__ flip, float threshold, int use_threshold) {
    const int b = blockIdx.z, Y = blockIdx.y, X = blockIdx.x * 256 + threadIdx.x;
    if (X >= N) return;
    const float c = cosf(angle[b]), s = sinf(angle[b]);
    // affine_grid base coordinates: (2i + 1) / N - 1
    const float gx = (2.f * (float)X + 1.f) / (float)N - 1.f, gy = (2.f * (float)Y + 1.f) / (float)N - 1.f;
    const float sx = c * gx - s * gy, sy = s * gx + c * gy;
    // grid_sample unnormalise (align_corners = False): ((coord + 1) * N - 1) / 2
    const float fx = ((sx + 1.f) * (float)N - 1.f) * 0.5f, fy = ((sy + 1.f) * (float)N - 1.f) * 0.5f;
    const float x0f = floorf(fx), y0f = floorf(fy);
    const int x0 = (int)x0f, y0 = (int)y0f;
    const float tx = fx - x0f, ty = fy - y0f;
    const float *img = in + (size_t)b * N * N;
    const int k = rot_k ? rot_k[b] : 0, fl = flip ? flip[b] : 0;
    // grid_sample's weights: nw = (x1 - x)(y1 - y), ne = (x - x0)(y1 - y), sw = (x1 - x)(y - y0), se = (x - x0)(y - y0)
    const float nw = (1.f - tx) * (1.f - ty), ne = tx * (1.f - ty), sw = (1.f - tx) * ty, se = tx * ty;
    float v = tap(img, N, y0, x0, k, fl) * nw;
    v += tap(img, N, y0, x0 + 1, k, fl) * ne;
    v += tap(img, N, y0 + 1, x0, k, fl) * sw;
    v += tap(img, N, y0 + 1, x0 + 1, k, fl) * se;
    if (use_threshold) v = v >= threshold ? 1.f : 0.f;
    out[((size_t)b * N + Y) * N + X] = v;
}

// ---- noise model of the GAN configs (reference data/data_transforms.py:25-42 SpeckleBrightnesd, :498-516 AddRandomBackgroundNoised) ----
// Both are streaming, HBM-bound maps over [B][H][W] float32 images; the random numbers are the reference's own streams (numpy's
// global uniform stream for the background factor, torch's CPU generator for the speckle grid), drawn on the host and passed in.

// out = max(img, noise * u): the product is formed in float64 as torch's type promotion does in the reference (float32 tensor
// times float64 numpy array), then compared with the float32 image; out_f64 receives the reference's float64 result, out_f32 its
// CastToTyped(float32) -- either may be NULL.
__global__ void __launch_bounds__(256)
background_noise_kernel(const float *__restrict__ img, const float *__restrict__ noise, const double *__restrict__ u, long n, double *__restrict__ out_f64,
                        float *__restrict__ out_f32) {
    const long i = (long)blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    const double a = (double)img[i], b = (double)noise[i] * u[i];
    const double m = (a >= b || a != a) ? a : b;                 // torch.maximum propagates NaN of either operand; b is finite here
    if (out_f64) out_f64[i] = m;
    if (out_f32) out_f32[i] = (float)m;
}

// torch's upsample_bilinear2d of the 9x9 control grid c (align_corners = False) at pixel (Y, X) of an H x W image
__device__ __forceinline__ float speckle_grid(const float *c, int H, int W, int Y, int X) {
    const float rh = 9.f / (float)H, rw = 9.f / (float)W;
    const float h1r = src_index(rh, Y), w1r = src_index(rw, X);
    const int h1 = (int)h1r, w1 = (int)w1r;
    const int h1p = h1 < 8 ? 1 : 0, w1p = w1 < 8 ? 1 : 0;
    const float h1l = h1r - (float)h1, h0l = 1.f - h1l, w1l = w1r - (float)w1, w0l = 1.f - w1l;
    const float *p = c + h1 * 9 + w1;
    return h0l * (w0l * p[0] + w1l * p[w1p]) + h1l * (w0l * p[h1p * 9] + w1l * p[h1p * 9 + w1p]);
}

// pass 1: v = img * (C - u (1 - C)); per-image max and min of v through order-preserving integer atomics
__device__ __forceinline__ int f2ord(float f) { int i = __float_as_int(f); return i >= 0 ? i : i ^ 0x7fffffff; }
__device__ __forceinline__ float ord2f(int i) { return __int_as_float(i >= 0 ? i : i ^ 0x7fffffff); }

__global__ void __launch_bounds__(256)
speckle_pass1_kernel(const float *__restrict__ img, const float *__restrict__ grid9, const float *__restrict__ u, int H, int W, float *__restrict__ v_out,
                     int *__restrict__ minmax) {
    const int b = blockIdx.z, Y = blockIdx.y, X = blockIdx.x * 256 + threadIdx.x;
    __shared__ int smx, smn;
    if (threadIdx.x == 0) { smx = f2ord(-INFINITY); smn = f2ord(INFINITY); }
    __syncthreads();
    if (X < W) {
        const size_t i = ((size_t)b * H + Y) * W + X;
        const float C = speckle_grid(grid9 + (size_t)b * 81, H, W, Y, X);
        const float R = C - u[i] * (1.f - C);
        const float v = img[i] * R;
        v_out[i] = v;
        atomicMax(&smx, f2ord(v));
        atomicMin(&smn, f2ord(v));
    }
    __syncthreads();
    if (threadIdx.x == 0) { atomicMax(minmax + 2 * b, smx); atomicMin(minmax + 2 * b + 1, smn); }
}

// pass 2: img /= img.max(); img -= img.min()   (min of the quotients = quotient of the min: division by a positive constant is monotone)
__global__ void __launch_bounds__(256)
speckle_pass2_kernel(float *__restrict__ v, long per_image, const int *__restrict__ minmax) {
    const int b = blockIdx.y;
    const long i = (long)blockIdx.x * 256 + threadIdx.x;
    if (i >= per_image) return;
    const float mx = ord2f(minmax[2 * b]), mn = ord2f(minmax[2 * b + 1]);
    float *p = v + (size_t)b * per_image + i;
    *p = *p / mx - mn / mx;
}

__global__ void speckle_init_kernel(int *minmax, int B) {
    const int b = blockIdx.x * 64 + threadIdx.x;
    if (b < B) { minmax[2 * b] = f2ord(-INFINITY); minmax[2 * b + 1] = f2ord(INFINITY); }
}

}  // namespace

extern "C" int octa_background_noise(octa_ctx *ctx, const float *d_img, const float *d_noise, const double *d_u, int64_t n, double *d_out_f64,
                                     float *d_out_f32, void *stream_) {
    if (!ctx || !d_img || !d_noise || !d_u || n <= 0 || (!d_out_f64 && !d_out_f32)) { octa::set_error("octa_background_noise: bad arguments"); return -2; }
    hipStream_t stream = (hipStream_t)stream_;
    OCTA_HIP_CHECK(hipSetDevice(ctx->device));
    hipLaunchKernelGGL(background_noise_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, stream, d_img, d_noise, d_u, (long)n, d_out_f64, d_out_f32);
    OCTA_HIP_CHECK(hipGetLastError());
    return 0;
}

extern "C" int octa_speckle_brightness(octa_ctx *ctx, const float *d_img, const float *d_grid9, const float *d_u, int B, int H, int W, float *d_out,
                                       int *d_minmax, void *stream_) {
    if (!ctx || !d_img || !d_grid9 || !d_u || !d_out || !d_minmax || B <= 0 || H <= 0 || W <= 0 || B > 65535 || H > 65535) {
        octa::set_error("octa_speckle_brightness: bad arguments"); return -2;
    }
    hipStream_t stream = (hipStream_t)stream_;
    OCTA_HIP_CHECK(hipSetDevice(ctx->device));
    hipLaunchKernelGGL(speckle_init_kernel, dim3((unsigned)((B + 63) / 64)), dim3(64), 0, stream, d_minmax, B);
    hipLaunchKernelGGL(speckle_pass1_kernel, dim3((unsigned)((W + 255) / 256), (unsigned)H, (unsigned)B), dim3(256), 0, stream, d_img, d_grid9, d_u, H, W, d_out, d_minmax);
    const long per = (long)H * W;
    hipLaunchKernelGGL(speckle_pass2_kernel, dim3((unsigned)((per + 255) / 256), (unsigned)B), dim3(256), 0, stream, d_out, per, d_minmax);
    OCTA_HIP_CHECK(hipGetLastError());
    return 0;
}

extern "C" int octa_resize_bilinear(octa_ctx *ctx, const void *d_in, int in_dtype, int B, int h, int w, float *d_out, int H, int W,
                                    const float *d_mul, const float *d_add, void *stream_) {
    if (!ctx || !d_in || !d_out || B <= 0 || h <= 0 || w <= 0 || H <= 0 || W <= 0 || B > 65535 || H > 65535) { octa::set_error("octa_resize_bilinear: bad arguments"); return -2; }
    hipStream_t stream = (hipStream_t)stream_;
    OCTA_HIP_CHECK(hipSetDevice(ctx->device));
    dim3 grid((unsigned)((W + 255) / 256), (unsigned)H, (unsigned)B);
    if (in_dtype == 0) hipLaunchKernelGGL(resize_bilinear_kernel<unsigned char>, grid, dim3(256), 0, stream, static_cast<const unsigned char *>(d_in), h, w, d_out, H, W, d_mul, d_add);
    else if (in_dtype == 1) hipLaunchKernelGGL(resize_bilinear_kernel<float>, grid, dim3(256), 0, stream, static_cast<const float *>(d_in), h, w, d_out, H, W, d_mul, d_add);
    else { octa::set_error("octa_resize_bilinear: in_dtype must be 0 (uint8) or 1 (float32)"); return -2; }
    OCTA_HIP_CHECK(hipGetLastError());
    return 0;
}

extern "C" int octa_resize_bilinear_bwd(octa_ctx *ctx, const float *d_dy, int B, int h, int w, int H, int W, float *d_dx, void *stream_) {
    if (!ctx || !d_dy || !d_dx || B <= 0 || h <= 0 || w <= 0 || H <= 0 || W <= 0 || B > 65535 || h > 65535) { octa::set_error("octa_resize_bilinear_bwd: bad arguments"); return -2; }
    hipStream_t stream = (hipStream_t)stream_;
    OCTA_HIP_CHECK(hipSetDevice(ctx->device));
    dim3 grid((unsigned)((w + 255) / 256), (unsigned)h, (unsigned)B);
    hipLaunchKernelGGL(resize_bilinear_bwd_kernel, grid, dim3(256), 0, stream, d_dy, h, w, H, W, d_dx);
    OCTA_HIP_CHECK(hipGetLastError());
    return 0;
}

extern "C" int octa_flip_rot90_rotate(octa_ctx *ctx, const float *d_in, float *d_out, int B, int N, const float *d_angle, const int *d_rot_k,
                                      const int *d_flip, float threshold, int use_threshold, void *stream_) {
    if (!ctx || !d_in || !d_out || !d_angle || B <= 0 || N <= 0 || B > 65535 || N > 65535 || d_in == d_out) { octa::set_error("octa_flip_rot90_rotate: bad arguments"); return -2; }
    hipStream_t stream = (hipStream_t)stream_;
    OCTA_HIP_CHECK(hipSetDevice(ctx->device));
    dim3 grid((unsigned)((N + 255) / 256), (unsigned)N, (unsigned)B);
    hipLaunchKernelGGL(rotate_kernel, grid, dim3(256), 0, stream, d_in, d_out, N, d_angle, d_rot_k, d_flip, threshold, use_threshold);
    OCTA_HIP_CHECK(hipGetLastError());
    return 0;
}
