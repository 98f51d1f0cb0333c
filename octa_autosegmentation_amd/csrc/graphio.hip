// graphio.hip -- device-side emulation of the CSV round trip of node positions.
//
// The reference writes positions with str(np.ndarray) (generate_vessel_graph.py:59-66) and reads them back
// with float() per token (tree2img.py:73-76, visualize_vessel_graphs.py:71-75), so labels are rendered
// from positions rounded to numpy's default text: per 3-vector either fixed notation with 8 decimals or --
// when min|x| < 1e-4 or max|x|/min|x| > 1e3 over its non-zero entries, or max|x| >= 1e8 -- scientific
// notation with 8 mantissa decimals. Both roundings are computed exactly: x * 10^p as an error-free
// product (fma), rounded to the nearest integer k, then k / 10^p (IEEE division = the nearest double to
// the decimal value = what strtod returns). One thread per 3-vector; 56 B in, 56 B out per edge.
#include "common.h"

namespace {

__device__ __constant__ double kPow10[23] = {1e0, 1e1, 1e2, 1e3, 1e4, 1e5, 1e6, 1e7, 1e8, 1e9, 1e10, 1e11,
                                             1e12, 1e13, 1e14, 1e15, 1e16, 1e17, 1e18, 1e19, 1e20, 1e21, 1e22};

// nearest integer (ties to even) of the exact product x * s, returned as a double (|x*s| < 2^52)
__device__ __forceinline__ double round_product(double x, double s) {
    double hi = x * s;
    double lo = fma(x, s, -hi);
    double k = rint(hi);
    double d = (hi - k) + lo;         // exact: |hi - k| <= 0.5, lo tiny
    if (d > 0.5 || (d == 0.5 && fmod(k, 2.0) != 0.0)) k += 1.0;
    else if (d < -0.5 || (d == -0.5 && fmod(k, 2.0) != 0.0)) k -= 1.0;
    return k;
}

__global__ void __launch_bounds__(256)
read_back_kernel(const double *__restrict__ in, double *__restrict__ out, long n_vec, int *__restrict__ flag) {
    long v = (long)blockIdx.x * blockDim.x + threadIdx.x;   // 3-vector index: two per edge
    if (v >= n_vec) return;
    long e = v >> 1;
    int half = (int)(v & 1);
    const double *p = in + 7 * e + 3 * half;
    double *o = out + 7 * e + 3 * half;
    double x[3] = {p[0], p[1], p[2]};
    if (half == 0) out[7 * e + 6] = in[7 * e + 6];
    double mx = 0.0, mn = INFINITY;
    bool any = false, bad = false;
    for (int k = 0; k < 3; k++) {
        double a = fabs(x[k]);
        if (!(a == a) || a == INFINITY) bad = true;
        if (a > 0.0) { any = true; mx = fmax(mx, a); mn = fmin(mn, a); }
    }
    bool sci = any && (mx >= 1e8 || mn < 1e-4 || mx / mn > 1000.0);
    for (int k = 0; k < 3; k++) {
        double a = fabs(x[k]);
        double r;
        if (bad) { r = x[k]; }
        else if (!sci) {
            if (a >= 4e7) { bad = true; r = x[k]; }
            else r = round_product(x[k], 1e8) / 1e8;
        } else if (a == 0.0) {
            r = x[k];
        } else {
            // decade e10 with 10^e10 <= a < 10^(e10+1), exact comparisons against powers of ten
            int e10 = 0;
            if (a >= 1.0) { while (e10 < 22 && a >= kPow10[e10 + 1]) e10++; }
            else { e10 = -1; while (e10 > -15 && a < 1.0 / kPow10[-e10]) e10--; if (a < 1.0 / kPow10[-e10]) bad = true; }
            // 1/10^m is not exact: decide the decade by multiplication instead when a < 1
            if (a < 1.0 && !bad) {
                e10 = -1;
                while (e10 > -15 && a * kPow10[-e10] < 1.0) e10--;   // a * 10^m is exact enough only near decade edges: re-checked below
                double m1 = a * kPow10[-e10];
                if (m1 < 1.0) bad = true;
                if (m1 >= 10.0) e10++;
            }
            int pw = 8 - e10;
            if (bad || pw > 22 || pw < 0) { bad = true; r = x[k]; }
            else {
                double kk = round_product(x[k], kPow10[pw]);
                r = kk / kPow10[pw];
            }
        }
        o[k] = r;
    }
    if (bad) atomicAdd(flag, 1);
}

// mode "1" rows of binarised labels (visualize_vessel_graphs.py:99 -> PNG bit depth 1): bit 7 of byte x / 8 is pixel x, non-zero = white.
// One thread per output byte; eight pixels are one 64-bit load when the row allows it. 1.48 MB in, 185 KB out per 1216 x 1216 label:
// what crosses PCIe for the label file is an eighth of the byte image.
__global__ void __launch_bounds__(256)
pack_bits_kernel(const uint8_t *__restrict__ in, uint8_t *__restrict__ out, long n_rows, int width, int row_bytes) {
    const long t = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= n_rows * row_bytes) return;
    const long row = t / row_bytes;
    const int bx = (int)(t - row * row_bytes);
    const uint8_t *src = in + row * width + 8 * bx;
    unsigned b = 0;
    if (8 * bx + 8 <= width && ((size_t)src & 7) == 0) {
        unsigned long long v = *reinterpret_cast<const unsigned long long *>(src);
        v |= v >> 4; v |= v >> 2; v |= v >> 1;
        v &= 0x0101010101010101ull;
        b = (unsigned)((v * 0x8040201008040201ull) >> 56);
    } else {
        for (int k = 0; k < 8 && 8 * bx + k < width; k++) if (src[k]) b |= 0x80u >> k;
    }
    out[t] = (uint8_t)b;
}

}  // namespace

extern "C" int octa_pack_bits(octa_ctx *ctx, const uint8_t *d_in, uint8_t *d_out, int64_t n_rows, int width, void *stream_) {
    if (!ctx || n_rows < 0 || width <= 0 || (n_rows > 0 && (!d_in || !d_out))) { octa::set_error("octa_pack_bits: bad arguments"); return -2; }
    if (n_rows == 0) return 0;
    OCTA_HIP_CHECK(hipSetDevice(ctx->device));
    const int rb = (width + 7) / 8;
    const long total = (long)n_rows * rb;
    hipLaunchKernelGGL(pack_bits_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, (hipStream_t)stream_, d_in, d_out, (long)n_rows, width, rb);
    OCTA_HIP_CHECK(hipGetLastError());
    return 0;
}

extern "C" int octa_edges_read_back(octa_ctx *ctx, const double *d_edges, double *d_out, int64_t n_edges, int *h_n_unhandled,
                                    void *stream_) {
    if (!ctx || (n_edges > 0 && (!d_edges || !d_out))) { octa::set_error("octa_edges_read_back: null pointer"); return -2; }
    if (h_n_unhandled) *h_n_unhandled = 0;
    if (n_edges <= 0) return 0;
    hipStream_t stream = (hipStream_t)stream_;
    OCTA_HIP_CHECK(hipSetDevice(ctx->device));
    if (ctx->r_counters.reserve(sizeof(long) * 8)) return -1;
    int *flag = ctx->r_counters.as<int>() + 14;
    OCTA_HIP_CHECK(hipMemsetAsync(flag, 0, sizeof(int), stream));
    long n_vec = 2 * (long)n_edges;
    hipLaunchKernelGGL(read_back_kernel, dim3((unsigned)((n_vec + 255) / 256)), dim3(256), 0, stream, d_edges, d_out, n_vec, flag);
    OCTA_HIP_CHECK(hipGetLastError());
    if (h_n_unhandled) {
        OCTA_HIP_CHECK(hipMemcpyAsync(h_n_unhandled, flag, sizeof(int), hipMemcpyDeviceToHost, stream));
        OCTA_HIP_CHECK(hipStreamSynchronize(stream));
    }
    return 0;
}
