// common.h -- context, error reporting and scratch management shared by the HIP sources.
#pragma once
#include <hip/hip_runtime.h>
#include <cstdarg>
#include <cstdint>
#include <cstdio>
#include <cmath>
#include <cstring>
#include <vector>

#include "../../include/octa_hip.h"

namespace octa {

void set_error(const char *fmt, ...);

#define OCTA_HIP_CHECK(expr)                                                                   \
    do {                                                                                       \
        hipError_t _e = (expr);                                                                \
        if (_e != hipSuccess) {                                                                \
            octa::set_error("%s failed: %s (%s:%d)", #expr, hipGetErrorString(_e), __FILE__,   \
                            __LINE__);                                                         \
            return -1;                                                                         \
        }                                                                                      \
    } while (0)

// grow-only device buffer
struct DevBuf {
    void *p = nullptr;
    size_t cap = 0;
    int reserve(size_t bytes) {
        if (bytes <= cap) return 0;
        if (p) {
            // kernels launched a moment ago may still read the old buffer (on any stream of this context's owner): wait
            // for the device explicitly instead of relying on hipFree's implicit synchronisation
            hipError_t e = hipDeviceSynchronize();
            e = hipFree(p);
            (void)e;
            p = nullptr;
            cap = 0;
        }
        size_t want = bytes + bytes / 4 + 256;
        hipError_t e = hipMalloc(&p, want);
        if (e != hipSuccess) {
            set_error("hipMalloc(%zu) failed: %s", want, hipGetErrorString(e));
            p = nullptr;
            return -1;
        }
        cap = want;
        return 0;
    }
    void release() {
        if (p) {
            hipError_t e = hipFree(p);
            (void)e;
        }
        p = nullptr;
        cap = 0;
    }
    template <class T> T *as() const { return reinterpret_cast<T *>(p); }
};

}  // namespace octa

#if defined(__HIPCC__)
// fp32 -> bf16, round to nearest even, on gfx950's v_cvt_pk_bf16_f32 (one instruction per PAIR; the integer formulation costs
// five per value, which shows in the VALU-bound epilogues and streaming kernels). Bit-identical to it for every non-NaN input.
__device__ __forceinline__ unsigned octa_pack_bf16x2(float lo, float hi) {
    typedef __attribute__((ext_vector_type(2))) float octa_f2;
    typedef __attribute__((ext_vector_type(2))) __bf16 octa_bf2;
    const octa_f2 v = {lo, hi};
    const octa_bf2 r = __builtin_convertvector(v, octa_bf2);
    return __builtin_bit_cast(unsigned, r);
}
__device__ __forceinline__ unsigned short octa_f2bf(float f) { return (unsigned short)(octa_pack_bf16x2(f, 0.f) & 0xffffu); }
#endif

struct octa_ctx {
    int device = 0;
    int num_cus = 256;
    // rasteriser scratch
    octa::DevBuf r_edge_off;    // int64 [B+1]
    octa::DevBuf r_ucount;      // int32 [n_total] upper bound of sides per edge -> local exclusive scan
    octa::DevBuf r_seg_total;   // int64 [B] per-graph totals, then exclusive bases
    octa::DevBuf r_sides;       // int4  [sum U]
    octa::DevBuf r_edge_meta;   // EdgeMeta [n_total]
    octa::DevBuf r_tile_count;  // int32 [B*ntiles] -> local exclusive scan
    octa::DevBuf r_tile_fill;   // int32 [B*ntiles]
    octa::DevBuf r_tile_total;  // int64 [B]
    octa::DevBuf r_tile_list;   // int32 [sum tile counts]
    octa::DevBuf r_counters;    // int64 [8] misc device counters
    // what octa_rasterize_2d_plan left for octa_rasterize_2d_draw (raster.hip)
    struct RasterPlan { bool valid = false, rowbin = false; int B = 0, W = 0, H = 0; long n_total = 0; } r_plan;
    octa::DevBuf zero_page;     // 256 zero bytes: source of the padding pixels of the DMA-staged convolution (conv.hip)
    octa::DevBuf wgrad_ws;      // per-workgroup partial weight gradients of conv3x3_nhwc_wgrad_tr_kernel (conv.hip), stream-ordered like the rest
    // ring of pre-zeroed accumulator slots (norm.hip statistics): ONE memset per lap instead of one per launch. Stream-ordered
    // like every other scratch buffer of the context (one context = one stream at a time).
    octa::DevBuf zero_ring;
    size_t zero_ring_pos = 0;
    // returns `bytes` (multiple of 256 after rounding) of zeros that the caller may dirty; nullptr on failure
    void *zeroed(size_t bytes, hipStream_t stream) {
        bytes = (bytes + 255) & ~(size_t)255;
        const size_t want = bytes * 64 > ((size_t)4 << 20) ? bytes * 64 : ((size_t)4 << 20);
        if (zero_ring.cap < want) {
            if (zero_ring.reserve(want)) return nullptr;
            zero_ring_pos = zero_ring.cap;            // force a clearing lap
        }
        if (zero_ring_pos + bytes > zero_ring.cap) {
            if (hipMemsetAsync(zero_ring.p, 0, zero_ring.cap, stream) != hipSuccess) { octa::set_error("zero ring memset failed"); return nullptr; }
            zero_ring_pos = 0;
        }
        void *r = static_cast<char *>(zero_ring.p) + zero_ring_pos;
        zero_ring_pos += bytes;
        return r;
    }
    size_t scratch_bytes() const {
        return r_edge_off.cap + r_ucount.cap + r_seg_total.cap + r_sides.cap + r_edge_meta.cap + r_tile_count.cap +
               r_tile_fill.cap + r_tile_total.cap + r_tile_list.cap + r_counters.cap;
    }
};
