// voxel.hip -- N6: 3-D anti-aliased tube voxeliser (reference tree2img.py:176-280 voxelize_forest with
// the 'cuboid' getCrossSlice of :151-172).
//
// Every edge max-blends 1 - (dist - (r - sqrt(3)/2)) / sqrt(3) into the voxels of its padded AABB
// (distance to the segment where the projection falls inside it, distance to the nearer endpoint
// everywhere); the reference finally stores uint16(255 * clip(v, 0, 1)). The quantisation is monotone,
// so the kernel takes the maximum of the QUANTISED contributions: one workgroup per edge walks its
// AABB (z fastest = memory order) and folds uint8 values into the volume with a 32-bit CAS max; a
// second, purely streaming kernel widens uint8 -> uint16 (16 B loads / 32 B stores). Edge order is
// irrelevant (max is commutative), so there is no ordering machinery here. HBM-bound by design:
// algorithmic bytes = the volume written once (X*Y*Z*2 B) + 56 B per edge.
#include "common.h"

namespace {

struct VoxParams {
    int dims[3];    // requested volume_dimensions
    int pd[3];      // padded dims
    double scale;   // max(volume_dimensions)
    double corr[3]; // (padded - requested) / 2
    int ignore_z;
    double min_radius, max_radius;
};

__global__ void __launch_bounds__(256)
voxel_edges_kernel(const double *__restrict__ edges, const unsigned char *__restrict__ keep, const long *__restrict__ edge_img,
                   long n_total, VoxParams P, unsigned char *__restrict__ vol8) {
    const long e_i = blockIdx.x;
    if (e_i >= n_total) return;
    const double *e = edges + 7 * e_i;
    double radius = e[6];
    if (radius < P.min_radius || radius > P.max_radius) return;
    if (keep && !keep[e_i]) return;
    radius *= P.scale;
    double cur[3], prox[3];
    for (int k = 0; k < 3; k++) { cur[k] = e[k] * P.scale + P.corr[k]; prox[k] = e[3 + k] * P.scale + P.corr[k]; }
    if (P.ignore_z) { cur[2] = P.pd[2] / 2; prox[2] = P.pd[2] / 2; }
    const double off = radius * sqrt(2.0);
    int s[3], t[3];
    for (int k = 0; k < 3; k++) {
        double a = cur[k], b = prox[k];
        if (a > b) { double tmp = a; a = b; b = tmp; }
        double lo = floor(a - off), hi = ceil(b + off + 1);
        s[k] = lo < 0 ? 0 : (lo > 2e9 ? 2000000000 : (int)lo);
        t[k] = hi > P.pd[k] ? P.pd[k] : (hi < -2e9 ? -2000000000 : (int)hi);
    }
    if (t[0] <= s[0] || t[1] <= s[1] || t[2] <= s[2]) return;
    const double seg[3] = {cur[0] - prox[0], cur[1] - prox[1], cur[2] - prox[2]};
    const double den = fma(seg[2], seg[2], fma(seg[1], seg[1], seg[0] * seg[0]));
    const double voxel_diag = sqrt(3.0);
    const double rr = radius - voxel_diag / 2;
    const long nx = t[0] - s[0], ny = t[1] - s[1], nz = t[2] - s[2];
    const long total = nx * ny * nz;
    unsigned char *vol = vol8 + (size_t)edge_img[e_i] * P.pd[0] * P.pd[1] * P.pd[2];
    for (long l = threadIdx.x; l < total; l += blockDim.x) {
        int z = s[2] + (int)(l % nz);
        long r2 = l / nz;
        int y = s[1] + (int)(r2 % ny);
        int x = s[0] + (int)(r2 / ny);
        double p[3] = {x + .5, y + .5, z + .5};
        double v[3] = {p[0] - prox[0], p[1] - prox[1], p[2] - prox[2]};
        double sp = ((v[0] * seg[0] + v[1] * seg[1]) + v[2] * seg[2]) / den;
        double best = -INFINITY;
        if (sp > 0 && sp < 1) {
            double q0 = p[0] - (prox[0] + sp * seg[0]), q1 = p[1] - (prox[1] + sp * seg[1]), q2 = p[2] - (prox[2] + sp * seg[2]);
            double dist = sqrt((q0 * q0 + q1 * q1) + q2 * q2);
            best = 1 - ((dist - rr) / voxel_diag);
        }
        double c0 = p[0] - cur[0], c1 = p[1] - cur[1], c2 = p[2] - cur[2];
        double d1 = sqrt((c0 * c0 + c1 * c1) + c2 * c2);
        double d2 = sqrt((v[0] * v[0] + v[1] * v[1]) + v[2] * v[2]);
        double dist = d1 < d2 ? d1 : d2;
        double ends = 1 - ((dist - rr) / voxel_diag);
        if (ends > best) best = ends;
        if (best > 0) {
            double cl = best > 1 ? 1 : best;
            unsigned qv = (unsigned)(255 * cl);
            if (qv == 0) continue;
            size_t idx = ((size_t)x * P.pd[1] + y) * P.pd[2] + z;
            unsigned *w = reinterpret_cast<unsigned *>(vol + (idx & ~(size_t)3));
            const int sh = (int)(idx & 3) * 8;
            unsigned old = *w;
            while (((old >> sh) & 255u) < qv) {
                unsigned nw = (old & ~(255u << sh)) | (qv << sh);
                unsigned prev = atomicCAS(w, old, nw);
                if (prev == old) break;
                old = prev;
            }
        }
    }
}

__global__ void voxel_widen_kernel(const unsigned char *__restrict__ in, unsigned short *__restrict__ out, size_t n) {
    size_t i = ((size_t)blockIdx.x * blockDim.x + threadIdx.x) * 16;
    if (i + 16 <= n) {
        uint4 v = *reinterpret_cast<const uint4 *>(in + i);
        const unsigned w[4] = {v.x, v.y, v.z, v.w};
        uint4 o[2];
        unsigned *po = reinterpret_cast<unsigned *>(o);
#pragma unroll
        for (int k = 0; k < 4; k++) {
            po[2 * k] = (w[k] & 255u) | (((w[k] >> 8) & 255u) << 16);
            po[2 * k + 1] = ((w[k] >> 16) & 255u) | ((w[k] >> 24) << 16);
        }
        uint4 *dst = reinterpret_cast<uint4 *>(out + i);
        dst[0] = o[0];
        dst[1] = o[1];
    } else {
        for (size_t k = i; k < n; k++) out[k] = in[k];
    }
}

}  // namespace

extern "C" int octa_voxel_padded_dims(const int *dims3, int *padded3) {
    if (!dims3 || !padded3) return -2;
    int scale = dims3[0] > dims3[1] ? dims3[0] : dims3[1];
    if (dims3[2] > scale) scale = dims3[2];
    int min_dim = (int)ceil((1.0 / 76) * scale + 2 * 0.015 * scale);
    for (int k = 0; k < 3; k++) padded3[k] = dims3[k] > min_dim ? dims3[k] : min_dim;
    return 0;
}

extern "C" int octa_voxelize_3d(octa_ctx *ctx, int B, const double *d_edges, const int64_t *h_edge_off, const uint8_t *d_keep,
                                const int *dims3, double min_radius, double max_radius, int ignore_z, uint16_t *d_out,
                                void *stream_) {
    if (!ctx || !h_edge_off || !dims3 || !d_out) { octa::set_error("octa_voxelize_3d: null pointer"); return -2; }
    if (B <= 0) return 0;
    hipStream_t stream = (hipStream_t)stream_;
    OCTA_HIP_CHECK(hipSetDevice(ctx->device));
    VoxParams P;
    for (int k = 0; k < 3; k++) { if (dims3[k] <= 0 || dims3[k] > 8192) { octa::set_error("octa_voxelize_3d: bad dimension %d", dims3[k]); return -2; } P.dims[k] = dims3[k]; }
    octa_voxel_padded_dims(P.dims, P.pd);
    P.scale = (double)(P.dims[0] > P.dims[1] ? (P.dims[0] > P.dims[2] ? P.dims[0] : P.dims[2]) : (P.dims[1] > P.dims[2] ? P.dims[1] : P.dims[2]));
    for (int k = 0; k < 3; k++) P.corr[k] = (P.pd[k] - P.dims[k]) / 2.0;
    P.ignore_z = ignore_z; P.min_radius = min_radius; P.max_radius = max_radius;
    const long n_total = (long)h_edge_off[B];
    if (h_edge_off[0] != 0) { octa::set_error("octa_voxelize_3d: offsets must start at 0"); return -2; }
    const size_t vox = (size_t)P.pd[0] * P.pd[1] * P.pd[2];
    const size_t vol_bytes = ((vox * (size_t)B + 15) / 16) * 16;
    if (ctx->r_sides.reserve(vol_bytes + 16)) return -1;
    if (ctx->r_seg_total.reserve(sizeof(long) * (size_t)(n_total + 1))) return -1;
    OCTA_HIP_CHECK(hipMemsetAsync(ctx->r_sides.p, 0, vol_bytes, stream));
    if (n_total > 0) {
        if (!d_edges) { octa::set_error("octa_voxelize_3d: null edges"); return -2; }
        std::vector<long> img((size_t)n_total);
        for (int b = 0; b < B; b++) {
            if (h_edge_off[b + 1] < h_edge_off[b]) { octa::set_error("octa_voxelize_3d: offsets must be non-decreasing"); return -2; }
            for (long i = h_edge_off[b]; i < h_edge_off[b + 1]; i++) img[(size_t)i] = b;
        }
        OCTA_HIP_CHECK(hipMemcpyAsync(ctx->r_seg_total.p, img.data(), sizeof(long) * n_total, hipMemcpyHostToDevice, stream));
        OCTA_HIP_CHECK(hipStreamSynchronize(stream));  // img goes out of scope
        hipLaunchKernelGGL(voxel_edges_kernel, dim3((unsigned)n_total), dim3(256), 0, stream, d_edges, d_keep,
                           ctx->r_seg_total.as<long>(), n_total, P, ctx->r_sides.as<unsigned char>());
        OCTA_HIP_CHECK(hipGetLastError());
    }
    const size_t n = vox * (size_t)B;
    const size_t threads = (n + 15) / 16;
    hipLaunchKernelGGL(voxel_widen_kernel, dim3((unsigned)((threads + 255) / 256)), dim3(256), 0, stream,
                       ctx->r_sides.as<unsigned char>(), d_out, n);
    OCTA_HIP_CHECK(hipGetLastError());
    return 0;
}
