// raster.hip -- N5 Agg-exact 2-D graph rasteriser for gfx950, plus N7 Floyd-Steinberg dither.
//
// What is reproduced (bit-exact, integer after the 24.8 conversion): the image that the
// reference's tree2img.rasterize_forest (vessel_graph_generation/tree2img.py:12-114) obtains
// from matplotlib's Agg backend for a white round-capped anti-aliased LineCollection --
// PathClipper, PathSnapper, agg::conv_stroke round caps, rasterizer_sl_clip_dbl,
// rasterizer_cells_aa coverage, non-zero winding, fixed_blender_rgba_plain, in list order.
//
// How (MI355X-first, not a translation of Agg's scanline machinery):
//  * raster_meta_kernel: one thread per edge runs the per-path double pipeline (radius filter,
//    x1.3 width, data->display, Liang-Barsky clip, auto-snap) and emits a fixed 56-byte record
//    + an int16 pixel bbox. No variable-size intermediate ever goes to HBM.
//  * raster_render_kernel: one 1024-thread workgroup per 64x64 pixel super-tile. It streams the
//    graph's bbox array (coalesced 8 B/edge), keeps the edges that touch the tile IN LIST ORDER
//    with a ballot/prefix compaction into LDS, tessellates their stroke polygons into 24.8
//    fixed-point sides in LDS (one thread per polygon side), then every wave owns a 16 x (4*NPX) block
//    (NPX pixels per lane, same row) and folds the edges in order: the Agg cell sums (cover, area)
//    of a pixel are evaluated in CLOSED FORM per polygon side -- Agg's two nested integer DDAs
//    are exact floor divisions, so x at scanline boundary j is x1 + floor(((256-fy1)+256(j-1))dx/dy)
//    and likewise for y at cell boundaries inside a scanline -- so no per-edge cell list, no
//    sorting and no atomics are needed, and the pixel lives in a register until it is stored once.
//    HBM traffic = edge records + bbox stream + one byte per pixel.
//  * fs_dither_kernel: Pillow's L->1 error diffusion; rows are skewed by 2 pixels across the 64
//    lanes of a wave (row r at column x needs row r-1 at column x+1), error terms move lane to
//    lane with a DPP/shuffle, bands of 64 rows hand the last row's errors on through LDS.
//
// All floating point here is IEEE double compiled with -ffp-contract=off.

#include <algorithm>
#include <cstdlib>

#include "common.h"
#include "raster_core.h"

namespace {
using namespace octa_raster;

constexpr int NPX = 4;          // adjacent pixels of one row per lane (register pressure vs. division reuse)
constexpr int BLK_H = 4 * NPX;  // a wave owns a 16 x BLK_H pixel block (16/NPX lanes per row)
constexpr int ST = 64;          // super-tile width (pixels): 4 blocks
#ifndef OCTA_RASTER_WG
#define OCTA_RASTER_WG 512
#endif
// The render workgroup: 512 threads on a 64 x 32 super-tile with 53 KB of LDS -- three per CU, six waves per SIMD, and room beside ONE
// simulator workgroup (80 KB). Rounds 4-5 shipped 1024 threads on 64 x 64 with 145 KB (ONE per CU, four waves per SIMD, 55 % vector-pipe
// utilisation: 8.0 ms per 128 labels against 6.1 for this form) because in the generator's pipeline the dense form was a disaster: a render
// workgroup dispatched while a launch was being placed took the slot the launch's second workgroup of that CU was waiting for and was replaced
// by the next render workgroup when it left -- 846 against 1131 samples/s. Round 6 removed the cause instead of the symptom: a launch's
// rasterisation is ordered behind the NEXT launch on the device (csrc/order.hip) and runs at the lower queue priority (pipeline.py), so render
// workgroups only ever take what finished samples leave, half a CU at a time: 1168 - 1176 samples/s against 1134 - 1155 for the exclusive
// form on one box, alternating (profiles/r06_raster_dense_ab.log). -DOCTA_RASTER_WG=1024 -DOCTA_RASTER_SLOTS_PER_THREAD=4
// -DOCTA_RASTER_LIST_CAP=512 -DOCTA_RASTER_CELL_CAP=160 -DOCTA_RASTER_ITEM_CAP=256 builds the exclusive form.
constexpr int WG = OCTA_RASTER_WG;             // threads per render workgroup: 8 waves = a 64 x 32 super-tile (1024: 64 x 64)
constexpr int ST_Y = (WG / 64 / 4) * BLK_H;    // super-tile height: a wave per 16 x 16 block, four blocks across
constexpr int EPT = 4;          // edges tested per thread per scan round (8 measured slower in round 4: bin 602 -> 656 WG-ms per 128 labels)
#ifndef OCTA_RASTER_LIST_CAP
#define OCTA_RASTER_LIST_CAP 256
#endif
constexpr int LIST_CAP = OCTA_RASTER_LIST_CAP;   // edges per chunk (1024 until round 5: the batched fold's cell lists need the LDS; a 64 x 64 super-tile of a 13 k-edge graph lists 60 - 150)
#ifndef OCTA_RASTER_FB
#define OCTA_RASTER_FB 4
#endif
#ifndef OCTA_RASTER_CELL_CAP
#define OCTA_RASTER_CELL_CAP 128
#endif
constexpr int FB = OCTA_RASTER_FB;   // edges folded per batch (their list indices travel in one 64-bit word, 10 bits each: at most 6)
constexpr int FB_MAX_SLOTS = 32;// an edge with more side slots than this is folded on its own
constexpr int CELL_CAP = OCTA_RASTER_CELL_CAP;   // cells (incl. carried covers) per wave and batch
#ifndef OCTA_RASTER_SLOTS_PER_THREAD
#define OCTA_RASTER_SLOTS_PER_THREAD 2
#endif
constexpr int SLOT_CAP = OCTA_RASTER_SLOTS_PER_THREAD * WG;  // int4 side slots per chunk (64 KiB; 80 KiB until the fold's accumulators needed 33 KiB, round 4)
#ifndef OCTA_RASTER_ITEM_CAP
#define OCTA_RASTER_ITEM_CAP 192
#endif
constexpr int ITEM_CAP = OCTA_RASTER_ITEM_CAP;   // (side, scanline) work items of one edge inside one wave's block

// ---- kernel 1: per-edge record ---------------------------------------------------------------

// Inclusive prefix sum over the 64 lanes of a wave by DPP adds (four shifts inside the rows of 16 lanes, lane 15 of rows 0 / 2 onto rows
// 1 / 3, lane 31 onto rows 2 and 3): six VALU instructions where six `__shfl_up` steps are six ds_bpermute round trips.
__device__ __forceinline__ int wave_scan_incl(int v) {
    v += __builtin_amdgcn_update_dpp(0, v, 0x111, 0xf, 0xf, true);
    v += __builtin_amdgcn_update_dpp(0, v, 0x112, 0xf, 0xf, true);
    v += __builtin_amdgcn_update_dpp(0, v, 0x114, 0xf, 0xf, true);
    v += __builtin_amdgcn_update_dpp(0, v, 0x118, 0xf, 0xf, true);
    v += __builtin_amdgcn_update_dpp(0, v, 0x142, 0xa, 0xf, false);
    v += __builtin_amdgcn_update_dpp(0, v, 0x143, 0xc, 0xf, false);
    return v;
}

__global__ void __launch_bounds__(256)
raster_meta_kernel(const double *__restrict__ edges, const unsigned char *__restrict__ keep, long n_total, int W, int H,
                   int ax_x, int ax_y, double min_radius, double max_radius, EdgeMeta *__restrict__ meta,
                   BBox16 *__restrict__ bbox) {
    long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n_total) return;
    EdgeMeta m;
    BBox16 b;
    compute_edge_meta(edges + 7 * i, keep ? keep[i] != 0 : true, W, H, ax_x, ax_y, min_radius, max_radius, &m, &b);
    meta[i] = m;
    bbox[i] = b;
}

// ---- kernels 2-4: side offsets (exclusive scan of nv + EXTRA_SLOTS) and tessellation, once per edge ----

constexpr int SCAN_EPT = 4;                    // elements per thread
constexpr int SCAN_BLK = 1024 * SCAN_EPT;      // elements per scan block

__device__ __forceinline__ int edge_slots(const EdgeMeta &m) { return m.nv > 0 ? m.nv + EXTRA_SLOTS : 0; }

__global__ void __launch_bounds__(1024)
raster_scan_block_kernel(const EdgeMeta *__restrict__ meta, long n_total, int *__restrict__ off, int *__restrict__ block_sums) {
    __shared__ int s_w[17];
    const long base = (long)blockIdx.x * SCAN_BLK + (long)threadIdx.x * SCAN_EPT;
    int v[SCAN_EPT], sum = 0;
#pragma unroll
    for (int q = 0; q < SCAN_EPT; q++) {
        v[q] = (base + q < n_total) ? edge_slots(meta[base + q]) : 0;
        sum += v[q];
    }
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    int inc = sum;
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) {
        int u = __shfl_up(inc, d, 64);
        if (lane >= d) inc += u;
    }
    if (lane == 63) s_w[wv] = inc;
    __syncthreads();
    if (threadIdx.x == 0) {
        int run = 0;
        for (int k = 0; k < 16; k++) { int t = s_w[k]; s_w[k] = run; run += t; }
        s_w[16] = run;
    }
    __syncthreads();
    int run = s_w[wv] + inc - sum;
#pragma unroll
    for (int q = 0; q < SCAN_EPT; q++) {
        if (base + q < n_total) off[base + q] = run;
        run += v[q];
    }
    if (threadIdx.x == 0) block_sums[blockIdx.x] = s_w[16];
}

// exclusive scan of the block sums by one workgroup; total (int64) -> total_out[0]; > INT_MAX sets err
__global__ void __launch_bounds__(1024)
raster_scan_sums_kernel(int *__restrict__ block_sums, int nb, long *__restrict__ total_out, int *__restrict__ err_flag) {
    __shared__ long s_w[17];
    __shared__ long s_carry;
    if (threadIdx.x == 0) s_carry = 0;
    __syncthreads();
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    for (int c0 = 0; c0 < nb; c0 += 1024) {
        const int i = c0 + (int)threadIdx.x;
        const long v = i < nb ? (long)block_sums[i] : 0;
        long inc = v;
#pragma unroll
        for (int d = 1; d < 64; d <<= 1) {
            long u = __shfl_up(inc, d, 64);
            if (lane >= d) inc += u;
        }
        if (lane == 63) s_w[wv] = inc;
        __syncthreads();
        if (threadIdx.x == 0) {
            long run = s_carry;
            for (int k = 0; k < 16; k++) { long t = s_w[k]; s_w[k] = run; run += t; }
            s_w[16] = run;
        }
        __syncthreads();
        const long ex = s_w[wv] + inc - v;
        if (i < nb) {
            if (ex > 0x7fffffffL) atomicExch(err_flag, 3);
            block_sums[i] = (int)ex;
        }
        __syncthreads();
        if (threadIdx.x == 0) s_carry = s_w[16];
        __syncthreads();
    }
    if (threadIdx.x == 0) total_out[0] = s_carry;
}

// one thread per edge: the stroke polygon's sides, clipped and converted to 24.8 fixed point, into the edge's
// nv + EXTRA_SLOTS slots (side k in slot k, extra clip pieces behind them, unused slots zero).
// Round 5: the edges of a workgroup own ONE contiguous run of the side array (their offsets come from an exclusive scan), so the
// sides are staged in LDS and written out by consecutive threads, 16 bytes each. Thread-per-edge stores straight to HBM put every
// lane's 16 bytes on a line of their own: the counters showed 4.7 GB written per 512 graphs for 1.5 GB of sides.
#ifndef OCTA_TESS_STAGE
#define OCTA_TESS_STAGE 2048
#endif
constexpr int TESS_STAGE = OCTA_TESS_STAGE;               // int4 slots (32 KiB): 128 edges x 16 slots (a 1.8-pixel stroke has 10 sides + 4 spare); longer runs fall back to direct stores. 3072 (48 KiB, three workgroups per CU) measured 8.25 / 3.50 ms per 128 labels / image pairs against 8.13 / 3.45 here, 1536: 8.12 / 3.39
#ifndef OCTA_TESS_WG
#define OCTA_TESS_WG 128
#endif
constexpr int TESS_WG = OCTA_TESS_WG;

__global__ void __launch_bounds__(TESS_WG)
raster_tess_kernel(const EdgeMeta *__restrict__ meta, const int *__restrict__ off, const int *__restrict__ block_sums,
                   long n_total, int W, int H, int4 *__restrict__ sides, int *__restrict__ err_flag) {
    __shared__ int4 s_stage[TESS_STAGE];
    __shared__ long s_run[2];                  // first slot of the workgroup's run, one past its last slot
    const long i0 = (long)blockIdx.x * blockDim.x, i = i0 + threadIdx.x;
    const bool valid = i < n_total;
    EdgeMeta m;
    m.nv = 0;
    if (valid) m = meta[i];
    const int nv = m.nv > 0 ? m.nv : 0;
    const long goff = valid ? (long)off[i] + (long)block_sums[i / SCAN_BLK] : 0;
    if (threadIdx.x == 0) s_run[0] = goff;
    const long last = (i0 + blockDim.x < n_total ? i0 + blockDim.x : n_total) - 1;
    if (i == last) s_run[1] = goff + (nv > 0 ? nv + EXTRA_SLOTS : 0);
    __syncthreads();
    const long base = s_run[0];
    const int run = (int)(s_run[1] - base);
    const bool staged = run <= TESS_STAGE;
    if (nv > 0) {
        int4 *out = staged ? s_stage + (goff - base) : sides + goff;
        for (int k = nv; k < nv + EXTRA_SLOTS; k++) out[k] = make_int4(0, 0, 0, 0);
        const double ddx = m.x1 - m.x0, ddy = m.y1 - m.y0;
        const double len = sqrt(ddx * ddx + ddy * ddy);
        // The polygon's vertices in order, with stroke_vertex()'s arithmetic (raster_core.h) but WALKED: per cap the offset vector and the
        // start angle once, the angle by the same chain of additions (vertex i of a cap is reached after i additions of da, as
        // agg::math_stroke::calc_cap accumulates it) -- stroke_vertex(v) recomputes atan2 and the whole chain for every vertex (round 5:
        // 560 -> us per 128 labels for this kernel, a third of it those repeats).
        const int per = m.n + 2;
        const double da = M_PI / (m.n + 1);
        int wi = 0, wcap = 0;
        double wdx1 = 0, wdy1 = 0, wa1 = 0, wv0x = 0, wv0y = 0;
        auto next_vertex = [&](double *px, double *py) {
            if (wi == 0) {
                wv0x = wcap ? m.x1 : m.x0; wv0y = wcap ? m.y1 : m.y0;
                const double v1x = wcap ? m.x0 : m.x1, v1y = wcap ? m.y0 : m.y1;
                wdx1 = (v1y - wv0y) / len;
                wdy1 = (v1x - wv0x) / len;
                wdx1 *= m.w;
                wdy1 *= m.w;
                *px = wv0x - wdx1; *py = wv0y + wdy1;
            } else if (wi == per - 1) {
                *px = wv0x + wdx1; *py = wv0y - wdy1;
            } else {
                if (wi == 1) { wa1 = atan2(wdy1, -wdx1); wa1 += da; } else wa1 += da;
                *px = wv0x + cos(wa1) * m.w;
                *py = wv0y + sin(wa1) * m.w;
            }
            if (++wi == per) { wi = 0; wcap++; }
        };
        double fx, fy;
        next_vertex(&fx, &fy);
        double ax = fx, ay = fy;
        int extra = 0;
        for (int v = 1; v <= nv; v++) {
            double bx = fx, by = fy;
            if (v < nv) next_vertex(&bx, &by);
            SideSink sink;
            sink.n = 0;
            clip_side(sink, (double)W, (double)H, ax, ay, bx, by);
            out[v - 1] = sink.n > 0 ? sink.piece[0] : make_int4(0, 0, 0, 0);
            for (int q = 1; q < sink.n; q++) {
                if (extra < EXTRA_SLOTS) out[nv + extra++] = sink.piece[q];
                else atomicExch(err_flag, 2);
            }
            ax = bx; ay = by;
        }
    }
    if (staged) {
        __syncthreads();
        for (int j = threadIdx.x; j < run; j += blockDim.x) sides[base + j] = s_stage[j];
    }
}

// ---- kernel 4b (round 5): edges per ROW of super-tiles -----------------------------------------------------------
// The render kernel's workgroup used to stream the graph's WHOLE bounding-box array to find the edges of its 64 x 64 super-tile: 361
// super-tiles of a 1216^2 label each tested all 13.5 k edges (28 % of the kernel's workgroup time after the fold got faster). One pass
// per (image, row of super-tiles) now writes, IN LIST ORDER, the indices of the edges whose box meets the row's 64 scanlines (u16: graphs
// of more than 65 535 edges keep the old scan); a super-tile then tests the ~7 % of the edges that are in its row.
struct RowRec {          // 16 bytes: a candidate of a row of super-tiles, everything the render kernel's compaction needs of it
    unsigned short edge, nv;     // index inside the graph (< 65 536), polygon sides
    int side_off;                // first slot of the edge in the global side array
    BBox16 bb;
};

__global__ void __launch_bounds__(256)
raster_rowbin_kernel(const BBox16 *__restrict__ bbox, const long *__restrict__ edge_off, int H, int rows, RowRec *__restrict__ row_list,
                     int *__restrict__ row_cnt, const EdgeMeta *__restrict__ meta, const int *__restrict__ side_off, const int *__restrict__ side_block_sums) {
    // every wave takes a contiguous quarter of the graph's edges: count (ballots), ONE barrier for the four totals, then emit at the running
    // offset (ballot prefix) -- the boxes are read twice from the L2 instead of once with three barriers per 4096 edges (236 -> ~30 us per
    // 128 labels)
    __shared__ int s_tot[4];
    const int img = blockIdx.y, row = blockIdx.x;
    const long e_begin = edge_off[img], e_end = edge_off[img + 1];
    const int n = (int)(e_end - e_begin);
    const BBox16 *gb = bbox + e_begin;
    RowRec *out = row_list + (size_t)e_begin * rows + (size_t)row * n;
    const int y0 = row * ST_Y, y1 = min(y0 + ST_Y, H) - 1;
    const int lane = threadIdx.x & 63, wv = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int seg = ((n + 4 * 64 - 1) / (4 * 64)) * 64;
    const int s0 = wv * seg < n ? wv * seg : n, s1 = s0 + seg < n ? s0 + seg : n;
    auto hit_at = [&](int e) {
        if (e >= s1) return false;
        const BBox16 b = gb[e];
        return b.x0 <= b.x1 && b.y1 >= y0 && b.y0 <= y1;
    };
    int c = 0;
    for (int e0 = s0; e0 < s1; e0 += 64) c += (int)__popcll(__ballot(hit_at(e0 + lane)));
    if (lane == 0) s_tot[wv] = c;
    __syncthreads();
    int at = 0, tot = 0;
#pragma unroll
    for (int k = 0; k < 4; k++) { const int t = s_tot[k]; at += k < wv ? t : 0; tot += t; }
    for (int e0 = s0; e0 < s1; e0 += 64) {
        const bool h = hit_at(e0 + lane);
        const unsigned long long m = __ballot(h);
        if (h) {
            // the candidate's whole record (round 5, last change): the render kernel's compaction read the index here, then the box, the side
            // count and the side offset of that edge from three more arrays -- four dependent loads in front of its first block scan
            const int e = e0 + lane;
            RowRec r;
            r.edge = (unsigned short)e;
            r.nv = (unsigned short)meta[e_begin + e].nv;
            r.side_off = side_off[e_begin + e] + side_block_sums[(e_begin + e) / SCAN_BLK];
            r.bb = gb[e];
            out[at + (int)__popcll(m & ((1ull << lane) - 1ull))] = r;
        }
        at += (int)__popcll(m);
    }
    if (threadIdx.x == 0) row_cnt[img * rows + row] = tot;
}

// The lane index, recomputed where it is asked for: the empty asm keeps the compiler from hoisting it (and everything derived from it: row,
// column, LDS addresses) out of the fold's loops into registers that then live -- and spill -- across the whole kernel.
__device__ __forceinline__ int fresh_lane() {
    int l = (int)(threadIdx.x & 63u);
    asm volatile("" : "+v"(l));
    return l;
}

#ifdef OCTA_RASTER_PROF
#define RASTER_PROF_BEGIN long _tp = (long)wall_clock64();
#define RASTER_PROF_MARK(slot) if (threadIdx.x == 0) { long _t = (long)wall_clock64(); atomicAdd((unsigned long long *)(err_flag + (slot)), (unsigned long long)(_t - _tp)); _tp = _t; }
#else
#define RASTER_PROF_BEGIN
#define RASTER_PROF_MARK(slot)
#endif

// ---- kernel 5: render ------------------------------------------------------------------------

struct ListEntry {      // 24 bytes (48 until round 5: the segment's start / direction / reach for the per-lane miss test lived here; only
    int edge;           // the one-edge fallback path asks for them now and fetches the edge record itself). edge index inside the graph
    int slot_off;       // first LDS slot
    int nv;             // polygon sides (primary slots)
    int side_off;       // first slot of the edge in the global side array
    BBox16 bb;
};

// block-wide exclusive scan of two ints per thread over WG threads: DPP wave scans, the wave totals through LDS, every thread sums
// the totals in front of its wave itself (round 4: ds_bpermute scans, a serial pass of thread 0 and one more barrier until then)
__device__ __forceinline__ void block_scan2(int a, int b, int *sh /*[2*16+2]*/, int &ea, int &eb, int &ta, int &tb) {
    const int lane = threadIdx.x & 63, wv = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int ia = wave_scan_incl(a), ib = wave_scan_incl(b);
    if (lane == 63) { sh[wv] = ia; sh[16 + wv] = ib; }
    __syncthreads();
    int sa = 0, sb = 0, fa = 0, fb = 0;
#pragma unroll
    for (int k = 0; k < WG / 64; k++) {
        const int va = sh[k], vb = sh[16 + k];
        fa += k < wv ? va : 0; fb += k < wv ? vb : 0;
        sa += va; sb += vb;
    }
    ea = fa + ia - a;
    eb = fb + ib - b;
    ta = sa;
    tb = sb;
    __syncthreads();
}

// RB: the candidates come as records from raster_rowbin_kernel (the usual case); false: every edge of the graph is a candidate and its box,
// side count and side offset are looked up here (graphs of 65 536 edges and more, OCTA_RASTER_ROWBIN=0). Two instantiations: the look-up
// path's three array pointers cost the record path scalar registers it spills
#ifndef OCTA_RASTER_MIN_WGS
#define OCTA_RASTER_MIN_WGS 1
#endif
template <bool RB>
__global__ void __launch_bounds__(WG, OCTA_RASTER_MIN_WGS)
raster_render_kernel(const EdgeMeta *__restrict__ meta, const BBox16 *__restrict__ bbox,
                     const long *__restrict__ edge_off, const int4 *__restrict__ sides, const int *__restrict__ side_off,
                     const int *__restrict__ side_block_sums, int W, int H, int tiles_x, int tiles_y,
                     unsigned char *__restrict__ out, int *__restrict__ err_flag, const RowRec *__restrict__ row_list,
                     const int *__restrict__ row_cnt) {
    __shared__ int4 s_slots[SLOT_CAP];
    __shared__ ListEntry s_list[LIST_CAP];
    __shared__ unsigned short s_owner[(WG / 64) * ITEM_CAP];
    __shared__ int s_scan[34];
    __shared__ int s_ctl[4];  // 0: first overflow edge, 1: list_n, 2: slots_n
    // per-wave accumulators of the item-parallel fold (round 4): cover and area of every cell of the wave's 16 x 16 block for the edge
    // being folded, and per row the cover of the edge's cells left of the block
    __shared__ __attribute__((aligned(16))) int s_cov[(WG / 64) * BLK_H * 16];
    __shared__ __attribute__((aligned(16))) int s_area[(WG / 64) * BLK_H * 16];
    __shared__ int s_carry[(WG / 64) * BLK_H];
    // the batched fold's cell lists (round 5), per wave: key = row << 4 | column (0x100 | row: cover carried in from the left), cover, area
    __shared__ unsigned short s_ckey[(WG / 64) * CELL_CAP];
    __shared__ int s_ccov[(WG / 64) * CELL_CAP], s_carea[(WG / 64) * CELL_CAP];
    static_assert(NPX == 4 && BLK_H == 16, "the accumulator fold reads a lane's four cells as one 16-byte piece");

    const int img = blockIdx.y;
    // XCD-aware tile order (round 5): workgroups go to the 8 XCDs round robin (workgroup b -> XCD b % 8), each XCD has its own L2, and a
    // 128-byte line of the image is shared by two horizontally adjacent super-tiles (64 bytes each). In plain order the two halves were
    // written through two different L2s -- every line left for HBM twice, half-filled (counters: 1.76 x the image bytes). Workgroups b and
    // b + 8 run on the same XCD back to back: they take the tile pair (2k, 2k + 1), so the halves meet in one L2.
    int tile = blockIdx.x;
    {
        const int g16 = tile & ~15;
        if (g16 + 16 <= (int)gridDim.x) tile = g16 + 2 * (tile & 7) + ((tile >> 3) & 1);
    }
    const int tx0 = (tile % tiles_x) * ST, ty0 = (tile / tiles_x) * ST_Y;
    const int tx1 = min(tx0 + ST, W) - 1, ty1 = min(ty0 + ST_Y, H) - 1;
    const long e_begin = edge_off[img], e_end = edge_off[img + 1];
    // the candidates of this super-tile: the edges of its row of super-tiles (raster_rowbin_kernel), or the whole graph
    const int trow = tile / tiles_x;
    const RowRec *rl = RB ? row_list + (size_t)e_begin * tiles_y + (size_t)trow * (size_t)(e_end - e_begin) : nullptr;
    const int n_edges = RB ? row_cnt[img * tiles_y + trow] : (int)(e_end - e_begin);      // positions in rl (or edges)
    const EdgeMeta *gm = meta + e_begin;
    const BBox16 *gb = bbox + e_begin;

    // wave -> 16x16 block, lane -> 4 adjacent pixels of one row
    // the wave index is wave-uniform by construction; saying so keeps the block's origin and the per-wave LDS bases in scalar registers
    // (round 5: the batched fold had pushed the kernel over its 128 vector registers -- twelve loop invariants spilled at kernel entry)
    const int lane = threadIdx.x & 63, wv = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int bx0 = tx0 + (wv & 3) * 16, by0 = ty0 + (wv >> 2) * BLK_H;
    unsigned pix[NPX];
#pragma unroll
    for (int q = 0; q < NPX; q++) pix[q] = 0u;
    int *w_cov = s_cov + wv * (BLK_H * 16), *w_area = s_area + wv * (BLK_H * 16), *w_carry = s_carry + wv * BLK_H;
    for (int i = lane; i < BLK_H * 16; i += 64) { w_cov[i] = 0; w_area[i] = 0; }
    if (lane < BLK_H) w_carry[lane] = 0;

    int cursor = 0;
    while (cursor < n_edges) {
        // phase timers of the render kernel (octa_raster_prof, tools/time_raster.py): a diagnostic build only (-DOCTA_RASTER_PROF) since round 5 --
        // the kernel sits at its 128 vector / 106 scalar registers, and the timers' clock reads and 64-bit atomics had tipped it into spilling
        RASTER_PROF_BEGIN
        // ---- phase 1: ordered compaction of the edges that touch this super-tile
        if (threadIdx.x == 0) { s_ctl[1] = 0; s_ctl[2] = 0; }
        __syncthreads();
        int list_n = 0, slots_n = 0;
        bool stop = false;
        while (!stop && cursor < n_edges && list_n < LIST_CAP / 2) {
            if (threadIdx.x == 0) s_ctl[0] = 0x7fffffff;
            int e0 = cursor + threadIdx.x * EPT;           // position in the candidate list
            BBox16 bb[EPT];
            int cnt[EPT];
            int hits = 0, slots = 0;
#pragma unroll
            for (int q = 0; q < EPT; q++) {
                int e = e0 + q;
                cnt[q] = 0;
                if (e < n_edges) {
                    bb[q] = RB ? rl[e].bb : gb[e];                 // (the record's other half is read in the hit branch: the same 16-byte line)
                    bool hit = bb[q].x0 <= bb[q].x1 && bb[q].x1 >= tx0 && bb[q].x0 <= tx1 && bb[q].y1 >= ty0 && bb[q].y0 <= ty1;
                    if (hit) {
                        int nv = RB ? (int)rl[e].nv : gm[e].nv;
                        cnt[q] = nv + EXTRA_SLOTS;
                        hits++;
                        slots += cnt[q];
                    }
                }
            }
            int eh, es, th, ts;
            block_scan2(hits, slots, s_scan, eh, es, th, ts);
            int pos = list_n + eh, sp = slots_n + es;
#pragma unroll
            for (int q = 0; q < EPT; q++) {
                if (cnt[q] > 0) {
                    if (pos < LIST_CAP && sp + cnt[q] <= SLOT_CAP) {
                        ListEntry le;
                        le.slot_off = sp;
                        le.nv = cnt[q] - EXTRA_SLOTS;
                        le.bb = bb[q];
                        if constexpr (RB) {
                            const RowRec rc = rl[e0 + q];                  // read again (a cache hit) rather than kept: the kernel sits at its register limit
                            le.edge = rc.edge;
                            le.side_off = rc.side_off;
                        } else {
                            le.edge = e0 + q;
                            le.side_off = side_off[e_begin + e0 + q] + side_block_sums[(e_begin + e0 + q) / SCAN_BLK];
                        }
                        s_list[pos] = le;
                    } else {
                        atomicMin(&s_ctl[0], e0 + q);
                    }
                    pos++;
                    sp += cnt[q];
                }
            }
            __syncthreads();
            int first_over = s_ctl[0];
            if (first_over != 0x7fffffff) {
                // entries before first_over were accepted; count them
                int acc = 0, accs = 0;
#pragma unroll
                for (int q = 0; q < EPT; q++)
                    if (cnt[q] > 0 && e0 + q < first_over) { acc++; accs += cnt[q]; }
                int d0, d1, ta, tb;
                block_scan2(acc, accs, s_scan, d0, d1, ta, tb);
                if (ta == 0 && list_n == 0) {
                    // a single edge does not fit the LDS pool: flag and skip it
                    if (threadIdx.x == 0) atomicExch(err_flag, 1);
                    cursor = first_over + 1;
                } else {
                    cursor = first_over;
                }
                list_n += ta;
                slots_n += tb;
                stop = true;
            } else {
                list_n += th;
                slots_n += ts;
                cursor += WG * EPT;
            }
            __syncthreads();
        }
        RASTER_PROF_MARK(2)
        if (list_n == 0) continue;

        // ---- phase 2: copy the tessellated sides of the listed edges into LDS (16 B per lane, coalesced per edge)
        for (int item = threadIdx.x; item < slots_n; item += WG) {
            int lo = 0, hi = list_n - 1;
            while (lo < hi) {
                int mid = (lo + hi + 1) >> 1;
                if (s_list[mid].slot_off <= item) lo = mid; else hi = mid - 1;
            }
            s_slots[item] = sides[(long)s_list[lo].side_off + (item - s_list[lo].slot_off)];
        }
        __syncthreads();

        RASTER_PROF_MARK(4)
        // ---- phase 3: ordered fold of the edges over this wave's 16x16 block.
        // Per edge the work items are (polygon side, scanline) pairs: lane = side counts the rows of its side
        // inside the block, a wave prefix sum lays the items out, lane = item computes the scanline piece
        // ONCE (the two 64-bit floor divisions of Agg's outer DDA), per-row ballots tell every pixel lane
        // which items lie in its row, and it adds their cells with the cheap inner divisions. Integer sums,
        // so the order of the items is irrelevant; very wide strokes fall back to the side loop.
        unsigned short *w_owner = s_owner + wv * ITEM_CAP;
        unsigned short *w_ckey = s_ckey + wv * CELL_CAP;
        int *w_ccov = s_ccov + wv * CELL_CAP, *w_carea = s_carea + wv * CELL_CAP;
        // the pixel pass of ONE edge: the block's accumulators (cover / area per cell, carried cover per row) -> this lane's four (C, A),
        // accumulators cleared for the next edge
        auto pixel_pass = [&](int (&C)[NPX], int (&A)[NPX]) {
            const int lane = fresh_lane(), myrow = lane / (16 / NPX);
            const int4 cv = *reinterpret_cast<const int4 *>(w_cov + lane * 4);         // lane = row * 4 + column group: its four cells
            const int4 av = *reinterpret_cast<const int4 *>(w_area + lane * 4);
            const int s4 = cv.x + cv.y + cv.z + cv.w;
            int pre = w_carry[myrow];
            {   // the cover of the cells to the left inside the row: the three lanes in front, by DPP row shifts
                const int u1 = __builtin_amdgcn_update_dpp(0, s4, 0x111, 0xf, 0xf, true);
                const int u2 = __builtin_amdgcn_update_dpp(0, s4, 0x112, 0xf, 0xf, true);
                const int u3 = __builtin_amdgcn_update_dpp(0, s4, 0x113, 0xf, 0xf, true);
                const int q = lane & 3;
                pre += (q >= 1 ? u1 : 0) + (q >= 2 ? u2 : 0) + (q >= 3 ? u3 : 0);
            }
            C[0] = pre + cv.x; C[1] = C[0] + cv.y; C[2] = C[1] + cv.z; C[3] = C[2] + cv.w;
            A[0] = av.x; A[1] = av.y; A[2] = av.z; A[3] = av.w;
            __builtin_amdgcn_wave_barrier();
            *reinterpret_cast<int4 *>(w_cov + lane * 4) = make_int4(0, 0, 0, 0);
            *reinterpret_cast<int4 *>(w_area + lane * 4) = make_int4(0, 0, 0, 0);
            if ((lane & 3) == 0) w_carry[myrow] = 0;
            __builtin_amdgcn_wave_barrier();
        };
        auto blend4 = [&](const int (&C)[NPX], const int (&A)[NPX]) {
#pragma unroll
            for (int q = 0; q < NPX; q++) {
                int v = (C[q] << 9) - A[q];
                int c = v >> 9;
                if (c < 0) c = -c;
                if (c > 255) c = 255;
                pix[q] = blend_white(pix[q], (unsigned)c);
            }
        };
        // rows of polygon side `sd` inside this wave's block: first row (relative) and count
        auto side_rows = [&](const int4 sd, int &r0, int &nr) {
            r0 = 0; nr = 0;
            if (sd.y != sd.w) {
                const int ey1 = sd.y >> 8, ey2 = sd.w >> 8;
                int lo = ey1 < ey2 ? ey1 : ey2, hi = ey1 < ey2 ? ey2 : ey1;
                const int xmin = (sd.x < sd.z ? sd.x : sd.z) >> 8;  // pieces right of the block add nothing
                lo = lo > by0 ? lo : by0;
                hi = hi < by0 + BLK_H - 1 ? hi : by0 + BLK_H - 1;
                if (hi >= lo && xmin <= bx0 + 15) { r0 = lo - by0; nr = hi - lo + 1; }
            }
        };
        // ONE edge folded on its own (rounds 3-4's path; now the fallback for strokes with more sides than a batch takes, for item or cell
        // counts beyond the LDS lists, and -- via the side loop -- for very wide strokes)
        auto fold_one = [&](const ListEntry &le) {
            const int lane = fresh_lane();
            const int prow = by0 + lane / (16 / NPX), pcol = bx0 + (lane % (16 / NPX)) * NPX;
            // the NPX x 1 pixel span of this lane cannot be touched if its centre is farther from the segment than half width + half
            // diagonal of the span + slack for the fp32 test, the 1/256 vertex rounding and the snap of axis-aligned paths (already in
            // ax..vy); only the side loop of very wide strokes and the blend below ask
            bool touch = !(prow < le.bb.y0 || prow > le.bb.y1 || pcol + NPX - 1 < le.bb.x0 || pcol > le.bb.x1);
            if (touch) {
                const EdgeMeta &em = gm[le.edge];
                const float vx = (float)(em.x1 - em.x0), vy = (float)(em.y1 - em.y0), l2 = vx * vx + vy * vy;
                const float inv_len2 = l2 > 0.f ? 1.0f / l2 : 0.f, reach = (float)em.w + 0.25f;
                float cx = (float)pcol + 0.5f * NPX - (float)em.x0, cy = (float)prow + 0.5f - (float)em.y0;
                float t = (cx * vx + cy * vy) * inv_len2;
                t = fminf(fmaxf(t, 0.f), 1.f);
                float ex = cx - t * vx, ey = cy - t * vy;
                float lim = reach + 0.5f * sqrtf((float)(NPX * NPX + 1)) + 0.01f;
                if (ex * ex + ey * ey > lim * lim) touch = false;
            }
            if (!__any(touch)) return;
            int C[NPX], A[NPX];
#pragma unroll
            for (int q = 0; q < NPX; q++) { C[q] = 0; A[q] = 0; }
            const int ns = le.nv + EXTRA_SLOTS;
            const int4 *sl = s_slots + le.slot_off;
            bool done = false;
            if (ns <= 64) {
                int r0 = 0, nr = 0;
                if (lane < ns) side_rows(sl[lane], r0, nr);
                const int inc = wave_scan_incl(nr);
                const int total = __builtin_amdgcn_readlane(inc, 63);
                if (total <= ITEM_CAP) {
                    done = true;
                    const int ex0 = inc - nr;
                    for (int t = 0; t < nr; t++) w_owner[ex0 + t] = (unsigned short)((lane << 8) | (r0 + t));
                    __builtin_amdgcn_wave_barrier();
                    for (int base = 0; base < total; base += 64) {
                        if (base + lane < total) {
                            const unsigned ow = w_owner[base + lane];
                            const int r = (int)(ow & 255u);
                            int hx1, hy1, hx2, hy2;
                            if (side_row_piece(sl[ow >> 8], by0 + r, hx1, hy1, hx2, hy2)) {
                                int *cr = w_cov + r * 16 - bx0, *ar = w_area + r * 16 - bx0;
                                const int carry = hline_cells(hx1, hy1, hx2, hy2, hline_step(hx1, hy1, hx2, hy2), bx0, 16,
                                                              [&](int px, int c, int a) { atomicAdd(cr + px, c); atomicAdd(ar + px, a); });
                                if (carry) atomicAdd(w_carry + r, carry);
                            }
                        }
                    }
                    __builtin_amdgcn_wave_barrier();
                    pixel_pass(C, A);
                }
            }
            if (!done && touch) {
                for (int k = 0; k < ns; k++) {
                    int4 sd = sl[k];
                    if (sd.y == sd.w) continue;
                    side_eval<NPX>(sd, prow, pcol, C, A);
                }
            }
            if (touch) blend4(C, A);
        };
        // Round 5: edges folded FOUR AT A TIME up to the pixel pass. A pair (edge, block) has ~3 (side, scanline) items: taken one edge at a
        // time, the side scan, the item layout and the items' divisions ran with 3 - 14 of 64 lanes busy and were the larger part of the
        // ~600 instructions a pair cost. A batch lays the sides of up to FB edges out on the lanes (sum of their slots <= 64), the items of all
        // of them in one list, and every item writes its cells -- (row, column, cover, area), the carried cover of the cells left of the block
        // as a cell of its own -- into a per-wave cell list, edge by edge contiguous (items come in side order, sides in edge order). Then,
        // per edge IN LIST ORDER: its cells are added into the block's accumulators, one pixel pass turns them into alpha and blends. Integer
        // sums per edge as before, so the image is the same bit for bit; an edge without cells in the block skips its pixel pass.
        unsigned long long fb_pack = 0;   // list indices of the batched edges, 10 bits each (wave-uniform; no array: a dynamic index would go to scratch)
        auto fb_i = [&](int k) { return (int)((fb_pack >> (10 * k)) & 1023ull); };
        int fb_n = 0, fb_slots = 0;
        // the batched path for the pending edges; false = an item or cell list would overflow (nothing folded: the caller takes them one by one)
        auto try_batch = [&]() -> bool {
            if (fb_n == 0) return true;
            const int lane = fresh_lane();
            // lane -> (edge k of the batch, side j)
            int k_of = -1, slot = 0;
            {
                int o = 0;
#pragma unroll
                for (int k = 0; k < FB; k++)
                    if (k < fb_n) {
                        const int ns = s_list[fb_i(k)].nv + EXTRA_SLOTS;
                        if (lane >= o && lane < o + ns) { k_of = k; slot = s_list[fb_i(k)].slot_off + (lane - o); }
                        o += ns;
                    }
            }
            int r0 = 0, nr = 0;
            if (k_of >= 0) side_rows(s_slots[slot], r0, nr);
            const int inc = wave_scan_incl(nr);
            const int total = __builtin_amdgcn_readlane(inc, 63);
            bool batched = total <= ITEM_CAP;
            int cend[FB];
#pragma unroll
            for (int k = 0; k < FB; k++) cend[k] = 0;
            if (batched) {
                const int ex0 = inc - nr;
                for (int t = 0; t < nr; t++) w_owner[ex0 + t] = (unsigned short)((k_of << 10) | (lane << 4) | (r0 + t));     // edge, side lane, row
                __builtin_amdgcn_wave_barrier();
                int cbase = 0;
                for (int base = 0; base < total && batched; base += 64) {
                    const bool have = base + lane < total;
                    int r = 0, ke = 0, hx1 = 0, hy1 = 0, hx2 = 0, hy2 = 0, ncell = 0;
                    bool piece = false, left = false;
                    const unsigned ow = have ? w_owner[base + lane] : 0u;
                    r = (int)(ow & 15u);
                    ke = (int)(ow >> 10);
                    // the side's LDS slot is what the side's lane computed above: fetched with ALL lanes active (a lane that has no item in this
                    // chunk may be the side lane of one that has)
                    const int sslot = __builtin_amdgcn_ds_bpermute((int)((ow >> 4) & 63u) << 2, slot);
                    if (have) {
                        piece = side_row_piece(s_slots[sslot], by0 + r, hx1, hy1, hx2, hy2);
                        if (piece) {
                            // cells of the piece inside the block's columns: hline_cells emits exactly one per column of the clipped range,
                            // and returns a carried cover iff the piece has cells left of the block
                            const int ea = hx1 >> 8, eb = hx2 >> 8;
                            const int lo = ea < eb ? ea : eb, hi = ea < eb ? eb : ea;
                            left = lo < bx0;
                            const int c0 = lo > bx0 ? lo : bx0, c1 = hi < bx0 + 15 ? hi : bx0 + 15;
                            ncell = (c1 >= c0 ? c1 - c0 + 1 : 0) + (left ? 1 : 0);
                        }
                    }
                    const int cinc = wave_scan_incl(ncell);
                    const int ctot = __builtin_amdgcn_readlane(cinc, 63);
                    if (cbase + ctot > CELL_CAP) { batched = false; break; }
                    int pos = cbase + cinc - ncell;
                    if (piece) {
                        const int carry = hline_cells(hx1, hy1, hx2, hy2, hline_step(hx1, hy1, hx2, hy2), bx0, 16, [&](int px, int c, int a) {
                            w_ckey[pos] = (unsigned short)((r << 4) | (px - bx0)); w_ccov[pos] = c; w_carea[pos] = a; pos++;
                        });
                        if (left) { w_ckey[pos] = (unsigned short)(0x100 | r); w_ccov[pos] = carry; w_carea[pos] = 0; pos++; }
                    }
                    // where each edge's cells end (items are in edge order): the last item of edge k in this chunk
#pragma unroll
                    for (int k = 0; k < FB; k++) {
                        const unsigned long long mk = __ballot(have && ke == k);
                        if (mk) cend[k] = __builtin_amdgcn_readlane(cbase + cinc, 63 - __builtin_clzll(mk));
                    }
                    cbase += ctot;
                }
                __builtin_amdgcn_wave_barrier();
            }
            if (batched) {
                int cstart = 0;
#pragma unroll
                for (int k = 0; k < FB; k++) {
                    if (k >= fb_n) break;
                    const int ce = cend[k] > cstart ? cend[k] : cstart;        // an edge without items keeps the running end
                    if (ce > cstart) {
                        for (int c = cstart + lane; c < ce; c += 64) {
                            const unsigned key = w_ckey[c];
                            if (key & 0x100u) atomicAdd(w_carry + (key & 15u), w_ccov[c]);
                            else { atomicAdd(w_cov + (key >> 4) * 16 + (key & 15u), w_ccov[c]); atomicAdd(w_area + (key >> 4) * 16 + (key & 15u), w_carea[c]); }
                        }
                        __builtin_amdgcn_wave_barrier();
                        int C[NPX], A[NPX];
                        pixel_pass(C, A);
                        blend4(C, A);                 // a pixel without coverage blends with alpha 0: unchanged
                    }
                    cstart = ce;
                }
            }
            return batched;
        };
        // ONE site drains what is pending (each lambda is inlined where it is called: one copy of either path): the batch, or -- when its
        // lists would overflow -- its edges one by one, then `extra` (a wide stroke) on its own
        auto drain = [&](int extra_i) {
            const bool singles = !try_batch();
            unsigned long long pk = singles ? fb_pack : 0ull;
            int n = singles ? fb_n : 0;
            if (extra_i >= 0) { pk |= (unsigned long long)extra_i << (10 * n); n++; }
            for (int k = 0; k < n; k++) fold_one(s_list[(int)((pk >> (10 * k)) & 1023ull)]);
            fb_n = 0; fb_slots = 0; fb_pack = 0;
        };
        // the entries whose bounding box misses this wave's block (most of a super-tile's list) are dropped 64 at a time: lane j tests
        // entry base + j, the ballot's set bits are visited in order. One loop, one drain site: the last turn (no candidate left) drains
        // what is pending.
        {
            int base_i = 0;
            unsigned long long cand = 0;
            for (;;) {
                while (!cand && base_i < list_n) {
                    const int j = base_i + lane;
                    bool hit = false;
                    if (j < list_n) {
                        const BBox16 b = s_list[j].bb;
                        hit = !(b.x1 < bx0 || b.x0 > bx0 + 15 || b.y1 < by0 || b.y0 > by0 + BLK_H - 1);
                    }
                    cand = __ballot(hit);
                    if (!cand) base_i += 64;
                }
                int i = -1, ns = 0;
                if (cand) {
                    i = base_i + (int)__ffsll((long long)cand) - 1;
                    cand &= cand - 1ull;
                    if (!cand) base_i += 64;
                    // (rounds 3-4 tested every lane's 4-pixel span against the stroke here and dropped the edge when no lane could be touched:
                    // 3 % of the candidates, for three LDS reads and ~30 instructions on every one of them. The batch needs no such test -- a
                    // pixel without coverage blends with alpha 0, an edge without cells in the block skips its pixel pass.)
                    ns = s_list[i].nv + EXTRA_SLOTS;
                }
                const bool last = i < 0;
                const bool wide = !last && ns > FB_MAX_SLOTS;     // a wide stroke (many cap vertices): on its own, after what is pending
                if (last || wide || fb_n == FB || fb_slots + ns > 64) drain(wide ? i : -1);
                if (last) break;
                if (!wide) {
                    fb_pack |= (unsigned long long)i << (10 * fb_n);
                    fb_n++;
                    fb_slots += ns;
                }
            }
        }
        __syncthreads();
        RASTER_PROF_MARK(6)
    }

    // Store (round 5): a wave's block is 16 pixels wide, so its lanes' 4-byte pieces filled a 64-byte row segment of the super-tile from four
    // different waves at four different times (counters: 1.75 x the image bytes written). The finished super-tile goes through LDS (the
    // side slots are dead by now) and leaves as whole 64-byte row segments, 16 bytes per thread.
    __syncthreads();
    unsigned *s_tile = reinterpret_cast<unsigned *>(s_slots);              // [ST_Y rows][ST / 4 words]
    {
        unsigned v = 0;
#pragma unroll
        for (int q = 0; q < NPX; q++) v |= pix[q] << (8 * q);
        const int prow = by0 + lane / (16 / NPX), pcol = bx0 + (lane % (16 / NPX)) * NPX;
        s_tile[(prow - ty0) * (ST / 4) + (pcol - tx0) / 4] = v;
    }
    __syncthreads();
    static_assert(NPX == 4, "the staged store packs a lane's four pixels into one word");
    if (threadIdx.x < ST_Y * (ST / 16)) {
        const int r = threadIdx.x / (ST / 16), c16 = threadIdx.x % (ST / 16);
        const int y = ty0 + r, x = tx0 + 16 * c16;
        if (y < H && x < W) {
            const size_t o = ((size_t)img * H + y) * (size_t)W + x;
            const uint4 v = *reinterpret_cast<const uint4 *>(s_tile + r * (ST / 4) + 4 * c16);
            if (x + 15 < W && (o & 15) == 0) {
                *reinterpret_cast<uint4 *>(out + o) = v;
            } else {
                const unsigned wds[4] = {v.x, v.y, v.z, v.w};
                for (int k = 0; k < 16 && x + k < W; k++) out[o + k] = (unsigned char)(wds[k >> 2] >> (8 * (k & 3)));
            }
        }
    }
}

// ---- Floyd-Steinberg (Pillow L -> 1) -----------------------------------------------------------
// One wave per image. Lane r of a band handles row band+r at column x = t - 2r in step t, so the
// term errors[x+1] of the row above is exactly what lane r-1 produced one step earlier (shuffle).
// The band's input rows are staged in LDS with coalesced 16-byte loads; the last row of a band
// leaves its error terms in LDS for the first row of the next band.

template <bool ST4>
__global__ void __launch_bounds__(64)
fs_dither_kernel(const unsigned char *__restrict__ in, int W, int H, int band_rows, int rs, unsigned char *__restrict__ out) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    int *errors = reinterpret_cast<int *>(smem);                       // [W+1], padded to 16 B
    const int err_bytes = (((W + 1) * 4 + 15) / 16) * 16;
    unsigned char *s_in = smem + err_bytes;                            // [band_rows][rs]; rs = an ODD number of 32-bit words (the 64 lanes read 64 rows at once)
    const int img = blockIdx.x;
    const int lane = threadIdx.x;
    const unsigned char *src = in + (size_t)img * W * H;
    unsigned char *dst = out + (size_t)img * W * H;
    for (int i = lane; i <= W; i += 64) errors[i] = 0;
    __syncthreads();
    const bool vec_ok = (W % 16 == 0) && ((reinterpret_cast<size_t>(src) & 15) == 0);
    for (int band = 0; band < H; band += band_rows) {
        const int rows = min(band_rows, H - band);
        // stage rows [band, band+rows)
        if (vec_ok) {
            const int vec_per_row = W / 16;
            const int total = rows * vec_per_row;
            const uint4 *g = reinterpret_cast<const uint4 *>(src + (size_t)band * W);
            for (int i = lane; i < total; i += 64) {
                int r = i / vec_per_row, c = i - r * vec_per_row;
                const uint4 v = g[i];
                unsigned *d = reinterpret_cast<unsigned *>(s_in + r * rs + c * 16);
                d[0] = v.x; d[1] = v.y; d[2] = v.z; d[3] = v.w;
            }
        } else {
            const int total = rows * W;
            for (int i = lane; i < total; i += 64) {
                int r = i / W, c = i - r * W;
                s_in[r * rs + c] = src[(size_t)band * W + i];
            }
        }
        __syncthreads();
        const int y = band + lane;
        const bool row_ok = lane < rows;
        int l = 0, l0 = 0, l1 = 0;
        int e_out = 0;
        unsigned obuf = 0;
        const int steps = (W + 1) + 2 * (rows - 1);
        const unsigned *my_w = reinterpret_cast<const unsigned *>(s_in + (row_ok ? lane : 0) * rs);
        unsigned char *my_out_row = dst + (size_t)y * W;
        // Everything a step reads from LDS is fetched ONE STEP AHEAD (the word holding this lane's next input pixel; for lane 0 the
        // error term of the row above, left by the previous band), and the error term of the row above inside the band moves down a
        // lane with a DPP wave shift: the first version paid three dependent LDS round trips per step (ds_bpermute, the error read,
        // a byte read of the input), ~480 cycles per step for a chain of ~25 integer instructions.
        // The step itself is straight-line code on selects (one wave per image runs alone on its SIMD: every instruction and every
        // taken branch of the ~25 000 serial steps is paid in full).
        unsigned wcur = my_w[0];                                        // x <= 0 at t = 0 for every lane
        int perr = errors[1];                                           // lane 0, step 0: errors[x + 1]
        const bool last_row = lane == rows - 1;
        for (int t = 0; t < steps; t++) {
            const int x = t - 2 * lane;
            const int xn = min(max(x + 1, 0), W - 1);
            const unsigned wnext = my_w[xn >> 2];
            const int perr_next = errors[min(t + 2, W)];                // what lane 0 needs in step t + 1 (same address for all lanes: a broadcast)
            int up = __builtin_amdgcn_update_dpp(0, e_out, 0x138 /* wave_shr:1 */, 0xf, 0xf, false);
            if (lane == 0) up = (x < W) ? perr : 0;
            const bool active = row_ok && (unsigned)x < (unsigned)W;
            const int sh = 8 * (x & 3);
            const int v = (int)((wcur >> sh) & 255u) + (l + up) / 16;
            const int lc = v <= 0 ? 0 : (v < 256 ? v : 255);
            const int o = (lc > 128) ? 255 : 0;
            const int e = lc - o;
            // Pillow: l = 7e for the next pixel; errors[x] = 3e + l0; l0 = 5e + l1; l1 = e; at x == W the row leaves l0 behind
            const int my_out = active ? 3 * e + l0 : ((row_ok && x == W) ? l0 : e_out);
            l0 = active ? 5 * e + l1 : l0;
            l1 = active ? e : l1;
            l = active ? 7 * e : l;
            if (ST4) {
                obuf = active ? (obuf | ((unsigned)o << sh)) : obuf;
                if (active && (x & 3) == 3) *reinterpret_cast<unsigned *>(my_out_row + (x - 3)) = obuf;
                obuf = (x & 3) == 3 ? 0u : obuf;
            } else if (active) {
                my_out_row[x] = (unsigned char)o;
            }
            if (last_row && x >= 0 && x <= W) errors[x] = my_out;
            e_out = my_out;
            wcur = wnext;
            perr = perr_next;
        }
        __syncthreads();
    }
}

// ---- Floyd-Steinberg as a pipeline of waves (round 4) ---------------------------------------------------------------
// The kernel above runs the 19 bands of a 1216-row label one after the other on ONE wave: (W + 127) x 19 = 25 500 serial steps. Band
// b + 1 only needs, for its first row at column x, the error term the LAST row of band b left at column x + 1 -- which that row
// produces 2 x 63 steps after the band's first row reached the column. So the bands run as a pipeline: one workgroup per image, wave w
// takes bands w, w + NW, ...; the last row of a band leaves its error terms in the band's own LDS row (no reuse, hence no
// back-pressure) and publishes its step count every DP_PUB steps; the wave of the next band starts each chunk of DP_PUB steps once the
// producer is DP_LAG steps ahead. 18 x 144 + 1343 = 3 900 step times instead of 25 500. Nothing is staged in LDS but the error rows:
// a lane keeps the 16 pixels of its row it is working on and the next 16 in registers (two 16-byte loads in flight per lane).
// Same arithmetic, same order of operations per row as Pillow's (ImagingConvert L -> 1): bit-identical to fs_dither_kernel.
constexpr int DP_PUB = 16;                    // steps between two publications of a band's progress
constexpr int DP_LAG = 2 * 63 + DP_PUB + 2;   // steps the producer must be ahead of the start of a consumer chunk

template <int DQ>          // words (4 pixels each) per store burst: 4, 8 or 16
__global__ void __launch_bounds__(1024)
fs_dither_pipe_kernel(const unsigned char *__restrict__ in, int W, int H, int n_bands, unsigned char *__restrict__ out) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    int *progress = reinterpret_cast<int *>(smem);                              // [n_bands] steps completed by band b (INT_MAX: done)
    int *errs = reinterpret_cast<int *>(smem) + ((n_bands + 3) & ~3);           // [n_bands][W + 2]
    const int ES = W + 2;
    const int img = blockIdx.x;
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6, nw = blockDim.x >> 6;
    const unsigned char *src = in + (size_t)img * W * H;
    unsigned char *dst = out + (size_t)img * W * H;
    for (int i = threadIdx.x; i < n_bands; i += blockDim.x) progress[i] = 0;
    __syncthreads();
    for (int b = wv; b < n_bands; b += nw) {
        const int row0 = b * 64;
        const int rows = min(64, H - row0);
        const bool row_ok = lane < rows;
        const int y = row0 + (row_ok ? lane : 0);
        const uint4 *my_row = reinterpret_cast<const uint4 *>(src + (size_t)y * W);
        unsigned char *my_out_row = dst + (size_t)y * W;
        const int *err_in = b > 0 ? errs + (size_t)(b - 1) * ES : nullptr;
        int *err_out = errs + (size_t)b * ES;
        volatile int *prog_in = b > 0 ? progress + (b - 1) : nullptr;
        const int nblk = W / 16;
        uint4 cur = my_row[0];
        uint4 nxt = nblk > 1 ? my_row[1] : make_uint4(0, 0, 0, 0);
        int l = 0, l0 = 0, l1 = 0, e_out = 0;
        unsigned obuf = 0, ow[DQ];               // ow[0 .. DQ - 2]: the finished words of the current burst, oldest first (ow[DQ - 1] unused)
#pragma unroll
        for (int k = 0; k < DQ; k++) ow[k] = 0u;
        const int steps = (W + 1) + 2 * (rows - 1);
        const bool last_row = lane == rows - 1;
        int perr = 0;
        for (int t0 = 0; t0 < steps; t0 += DP_PUB) {
            if (prog_in) {                                    // the band above must be DP_LAG steps ahead of this chunk (or finished)
                const int need = t0 + DP_LAG;
                while (*prog_in < need) __builtin_amdgcn_s_sleep(2);
                __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
            }
            if (t0 == 0) perr = err_in ? err_in[1] : 0;       // lane 0, step 0: errors[x + 1]
            const int t1 = min(t0 + DP_PUB, steps);
            for (int t = t0; t < t1; t++) {
                const int x = t - 2 * lane;
                const int perr_next = err_in ? err_in[min(t + 2, W)] : 0;          // what lane 0 needs in step t + 1 (one address for all lanes)
                int up = __builtin_amdgcn_update_dpp(0, e_out, 0x138 /* wave_shr:1 */, 0xf, 0xf, false);
                if (lane == 0) up = (x < W) ? perr : 0;
                const bool active = row_ok && (unsigned)x < (unsigned)W;
                const int sh = 8 * (x & 3);
                const int v = (int)((cur.x >> sh) & 255u) + (l + up) / 16;
                const int lc = v <= 0 ? 0 : (v < 256 ? v : 255);
                const int o = (lc > 128) ? 255 : 0;
                const int e = lc - o;
                const int my_out = active ? 3 * e + l0 : ((row_ok && x == W) ? l0 : e_out);
                l0 = active ? 5 * e + l1 : l0;
                l1 = active ? e : l1;
                l = active ? 7 * e : l;
                obuf = active ? (obuf | ((unsigned)o << sh)) : obuf;
                // DQ words (16 / 32 / 64 pixels) per store burst (round 5; 4 pixels until then): the lanes of a wave write 64 different rows, so
                // every store opens a line of its own in the L2, and with 4-byte pieces the lines were evicted half-written again and again
                // (counters: 9 x the output bytes written, 5 x the input bytes fetched). The finished words wait in a shift queue of DQ - 1
                // registers; every DQ words DQ / 4 16-byte stores go out back to back (W % 16 == 0 and 16-byte aligned rows are the kernel's
                // precondition; the tail of a row whose width is no multiple of the burst goes out in 16-byte pieces).
                if ((x & 3) == 3) {
                    constexpr int PX = 4 * DQ;
                    if (active && ((x & (PX - 1)) == PX - 1 || x == W - 1)) {
                        const int nq = ((x & (PX - 1)) >> 4) + 1;             // 16-byte pieces queued (1 .. DQ / 4), the newest last
                        uint4 *o16 = reinterpret_cast<uint4 *>(my_out_row + (x & ~(PX - 1)));
                        // the queue is right-aligned: piece j of the burst (j = 0 oldest of a FULL burst) holds words 4j .. 4j + 3, the last word is obuf
#pragma unroll
                        for (int j = 0; j < DQ / 4; j++) {
                            const int jj = j - (DQ / 4 - nq);                 // position of piece j in this (possibly short) burst
                            if (jj >= 0) o16[jj] = make_uint4(ow[4 * j], ow[4 * j + 1], ow[4 * j + 2], j == DQ / 4 - 1 ? obuf : ow[4 * j + 3]);
                        }
                    }
#pragma unroll
                    for (int k = 0; k < DQ - 2; k++) ow[k] = ow[k + 1];
                    ow[DQ - 2] = obuf;
                    obuf = 0u;
                }
                if (last_row && x >= 0 && x <= W) err_out[x] = my_out;
                e_out = my_out;
                perr = perr_next;
                // the lane's pixel window: the next word every 4 pixels, the next 16-byte block every 16 (and the block after that goes in flight)
                if (x >= 0 && (x & 3) == 3) {
                    if ((x & 15) == 15) {
                        cur = nxt;
                        const int kb = (x >> 4) + 2;
                        if (kb < nblk && row_ok) nxt = my_row[kb];
                    } else {
                        cur.x = cur.y; cur.y = cur.z; cur.z = cur.w;
                    }
                }
            }
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
            if (lane == 0) *(volatile int *)(progress + b) = t1 >= steps ? 0x7fffffff : t1;
        }
    }
}

__global__ void max_u8_kernel(const unsigned char *a, const unsigned char *b, unsigned char *o, size_t n) {
    size_t i = ((size_t)blockIdx.x * blockDim.x + threadIdx.x) * 16;
    if (i + 16 <= n) {
        uint4 va = *reinterpret_cast<const uint4 *>(a + i), vb = *reinterpret_cast<const uint4 *>(b + i), vo;
        const unsigned *pa = &va.x, *pb = &vb.x;
        unsigned *po = &vo.x;
#pragma unroll
        for (int k = 0; k < 4; k++) {
            unsigned r = 0;
#pragma unroll
            for (int s = 0; s < 32; s += 8) {
                unsigned x = (pa[k] >> s) & 255u, y = (pb[k] >> s) & 255u;
                r |= (x > y ? x : y) << s;
            }
            po[k] = r;
        }
        *reinterpret_cast<uint4 *>(o + i) = vo;
    } else {
        for (size_t k = i; k < n; k++) o[k] = a[k] > b[k] ? a[k] : b[k];
    }
}

}  // namespace

// ---- C-ABI -----------------------------------------------------------------------------------

// The rasteriser in two calls (round 5). _plan: per-edge records, side offsets, and the ONE host synchronisation of the rasteriser (the
// side array is sized from the scan total) -- a few short kernels; _draw: tessellation, row binning and the ordered fold, enqueued
// without touching the host or the allocator. A pipeline that shares the GPU with a long kernel plans while the GPU is free and draws
// whenever it likes: nothing of the heavy part can end up queued behind a host wait (pipeline.py). octa_rasterize_2d = both.
extern "C" int octa_rasterize_2d_plan(octa_ctx *ctx, int B, const double *d_edges, const int64_t *h_edge_off,
                                      const uint8_t *d_keep, int no_pixels_x, int no_pixels_y, int mip_axis,
                                      double min_radius, double max_radius, void *stream_) {
    if (!ctx) { octa::set_error("octa_rasterize_2d: null ctx"); return -2; }
    ctx->r_plan = octa_ctx::RasterPlan();
    if (B <= 0) return 0;
    if (!h_edge_off) { octa::set_error("octa_rasterize_2d: null pointer"); return -2; }
    if (mip_axis < 0 || mip_axis > 2) { octa::set_error("octa_rasterize_2d: MIP_axis must be 0, 1 or 2"); return -2; }
    const int W = no_pixels_x, H = no_pixels_y;
    if (W <= 0 || H <= 0 || W > 16000 || H > 16000) { octa::set_error("octa_rasterize_2d: bad resolution %dx%d", W, H); return -2; }
    if (B > 65535) { octa::set_error("octa_rasterize_2d: B > 65535"); return -2; }
    hipStream_t stream = (hipStream_t)stream_;
    OCTA_HIP_CHECK(hipSetDevice(ctx->device));
    const long n_total = (long)h_edge_off[B];
    for (int b = 0; b < B; b++)
        if (h_edge_off[b + 1] < h_edge_off[b] || h_edge_off[0] != 0) { octa::set_error("octa_rasterize_2d: edge offsets must start at 0 and be non-decreasing"); return -2; }
    if (n_total > 0 && !d_edges) { octa::set_error("octa_rasterize_2d: null edges"); return -2; }
    int axes[2], k = 0;
    for (int a = 0; a < 3; a++) if (a != mip_axis) axes[k++] = a;
    // tree2img.py:85: segment = ((cur[axes[1]], cur[axes[0]]), ...) -> x from axes[1], y from axes[0]
    const int ax_x = axes[1], ax_y = axes[0];

    if (ctx->r_edge_off.reserve(sizeof(long) * (B + 1))) return -1;
    if (ctx->r_edge_meta.reserve(sizeof(EdgeMeta) * (size_t)(n_total + 1))) return -1;
    if (ctx->r_ucount.reserve(sizeof(BBox16) * (size_t)(n_total + 1))) return -1;
    if (ctx->r_counters.reserve(sizeof(long) * 8)) return -1;
    OCTA_HIP_CHECK(hipMemcpyAsync(ctx->r_edge_off.p, h_edge_off, sizeof(long) * (B + 1), hipMemcpyHostToDevice, stream));
    OCTA_HIP_CHECK(hipMemsetAsync(ctx->r_counters.p, 0, sizeof(long) * 8, stream));
    const int nb = (int)((n_total + SCAN_BLK - 1) / SCAN_BLK);
    if (ctx->r_tile_count.reserve(sizeof(int) * (size_t)(n_total + 1))) return -1;   // per-edge side offsets (block-local)
    if (ctx->r_seg_total.reserve(sizeof(int) * (size_t)(nb + 1))) return -1;          // scan block sums -> bases
    long total_sides = 0;
    if (n_total > 0) {
        dim3 g((unsigned)((n_total + 255) / 256));
        hipLaunchKernelGGL(raster_meta_kernel, g, dim3(256), 0, stream, d_edges, d_keep, n_total, W, H, ax_x, ax_y,
                           min_radius, max_radius, ctx->r_edge_meta.as<EdgeMeta>(), ctx->r_ucount.as<BBox16>());
        hipLaunchKernelGGL(raster_scan_block_kernel, dim3((unsigned)nb), dim3(1024), 0, stream, ctx->r_edge_meta.as<EdgeMeta>(),
                           n_total, ctx->r_tile_count.as<int>(), ctx->r_seg_total.as<int>());
        hipLaunchKernelGGL(raster_scan_sums_kernel, dim3(1), dim3(1024), 0, stream, ctx->r_seg_total.as<int>(), nb,
                           ctx->r_counters.as<long>() + 4, ctx->r_counters.as<int>());
        OCTA_HIP_CHECK(hipGetLastError());
        // the side array is sized from the scan total: one 8-byte read-back per call on this stream
        OCTA_HIP_CHECK(hipMemcpyAsync(&total_sides, ctx->r_counters.as<long>() + 4, sizeof(long), hipMemcpyDeviceToHost, stream));
        OCTA_HIP_CHECK(hipStreamSynchronize(stream));
        if (total_sides > 0x7fffffffL) { octa::set_error("octa_rasterize_2d: %ld polygon sides exceed the 32-bit offset range", total_sides); return -2; }
        if (ctx->r_sides.reserve(sizeof(int4) * (size_t)(total_sides + 1))) return -1;
    } else {
        if (ctx->r_sides.reserve(sizeof(int4))) return -1;
    }
    const int tiles_y = (H + ST_Y - 1) / ST_Y;
    // edges per row of super-tiles (u16 indices inside a graph): every graph must have fewer than 65 536 edges, and one row must be worth it
    bool rowbin = false;
    {
        constexpr bool rowbin_on = true;
        long max_graph = 0;
        for (int b = 0; b < B; b++) max_graph = std::max<long>(max_graph, (long)(h_edge_off[b + 1] - h_edge_off[b]));
        if (rowbin_on && n_total > 0 && tiles_y > 1 && max_graph < 65536) {
            if (ctx->r_tile_list.reserve(sizeof(RowRec) * (size_t)n_total * tiles_y + 16)) return -1;
            if (ctx->r_tile_fill.reserve(sizeof(int) * (size_t)B * tiles_y)) return -1;
            rowbin = true;
        }
    }
    ctx->r_plan.valid = true; ctx->r_plan.rowbin = rowbin;
    ctx->r_plan.B = B; ctx->r_plan.W = W; ctx->r_plan.H = H; ctx->r_plan.n_total = n_total;
    return 0;
}

extern "C" int octa_rasterize_2d_draw(octa_ctx *ctx, uint8_t *d_out, void *stream_) {
    if (!ctx) { octa::set_error("octa_rasterize_2d: null ctx"); return -2; }
    if (!ctx->r_plan.valid) { octa::set_error("octa_rasterize_2d_draw: no plan (octa_rasterize_2d_plan must precede every draw)"); return -2; }
    if (!d_out) { octa::set_error("octa_rasterize_2d: null pointer"); return -2; }
    const int B = ctx->r_plan.B, W = ctx->r_plan.W, H = ctx->r_plan.H;
    const long n_total = ctx->r_plan.n_total;
    const bool rowbin = ctx->r_plan.rowbin;
    ctx->r_plan.valid = false;                  // the scratch holds ONE plan; it is consumed here
    hipStream_t stream = (hipStream_t)stream_;
    OCTA_HIP_CHECK(hipSetDevice(ctx->device));
    if (n_total > 0)
        hipLaunchKernelGGL(raster_tess_kernel, dim3((unsigned)((n_total + TESS_WG - 1) / TESS_WG)), dim3(TESS_WG), 0, stream, ctx->r_edge_meta.as<EdgeMeta>(), ctx->r_tile_count.as<int>(),
                           ctx->r_seg_total.as<int>(), n_total, W, H, ctx->r_sides.as<int4>(), ctx->r_counters.as<int>());
    const int tiles_x = (W + ST - 1) / ST, tiles_y = (H + ST_Y - 1) / ST_Y;
    const RowRec *row_list = nullptr;
    const int *row_cnt = nullptr;
    {
        if (rowbin) {
            hipLaunchKernelGGL(raster_rowbin_kernel, dim3((unsigned)tiles_y, (unsigned)B), dim3(256), 0, stream, ctx->r_ucount.as<BBox16>(),
                               ctx->r_edge_off.as<long>(), H, tiles_y, ctx->r_tile_list.as<RowRec>(), ctx->r_tile_fill.as<int>(), ctx->r_edge_meta.as<EdgeMeta>(),
                               ctx->r_tile_count.as<int>(), ctx->r_seg_total.as<int>());
            row_list = ctx->r_tile_list.as<RowRec>();
            row_cnt = ctx->r_tile_fill.as<int>();
        }
    }
    dim3 grid((unsigned)(tiles_x * tiles_y), (unsigned)B);
    hipLaunchKernelGGL((row_list ? raster_render_kernel<true> : raster_render_kernel<false>), grid, dim3(WG), 0, stream, ctx->r_edge_meta.as<EdgeMeta>(),
                       ctx->r_ucount.as<BBox16>(), ctx->r_edge_off.as<long>(), ctx->r_sides.as<int4>(), ctx->r_tile_count.as<int>(),
                       ctx->r_seg_total.as<int>(), W, H, tiles_x, tiles_y, d_out, ctx->r_counters.as<int>(), row_list, row_cnt);
    OCTA_HIP_CHECK(hipGetLastError());
    return 0;
}

extern "C" int octa_rasterize_2d(octa_ctx *ctx, int B, const double *d_edges, const int64_t *h_edge_off,
                                 const uint8_t *d_keep, int no_pixels_x, int no_pixels_y, int mip_axis,
                                 double min_radius, double max_radius, uint8_t *d_out, void *stream_) {
    if (B > 0 && !d_out) { octa::set_error("octa_rasterize_2d: null pointer"); return -2; }
    const int rc = octa_rasterize_2d_plan(ctx, B, d_edges, h_edge_off, d_keep, no_pixels_x, no_pixels_y, mip_axis, min_radius, max_radius, stream_);
    if (rc || B <= 0) return rc;
    return octa_rasterize_2d_draw(ctx, d_out, stream_);
}

extern "C" int octa_fs_dither(octa_ctx *ctx, int B, const uint8_t *d_in, int W, int H, uint8_t *d_out, void *stream_) {
    if (!ctx) { octa::set_error("octa_fs_dither: null ctx"); return -2; }
    if (B <= 0) return 0;
    if (!d_in || !d_out || W <= 0 || H <= 0) { octa::set_error("octa_fs_dither: bad arguments"); return -2; }
    hipStream_t stream = (hipStream_t)stream_;
    OCTA_HIP_CHECK(hipSetDevice(ctx->device));
    {   // the pipelined form (round 4): rows 16-byte aligned, all bands' error rows in LDS; OCTA_DITHER_PIPE=0 keeps the one-wave kernel
        static const bool pipe_on = [] { const char *e = getenv("OCTA_DITHER_PIPE"); return !(e && e[0] == '0'); }();
        const int n_bands = (H + 63) / 64;
        const size_t lds_pipe = ((size_t)((n_bands + 3) & ~3) + (size_t)n_bands * (W + 2)) * sizeof(int);
        if (pipe_on && W % 16 == 0 && (reinterpret_cast<size_t>(d_in) & 15) == 0 && (reinterpret_cast<size_t>(d_out) & 15) == 0 && n_bands >= 2 &&
            lds_pipe <= 150 * 1024) {
            const int nw = n_bands < 16 ? n_bands : 16;
            // OCTA_DITHER_BURST = 16 | 32 | 64 pixels per store burst (development aid). Measured per 128 labels at 1216^2 (profiles/r05_raster_pmc.log):
            // 16: 488 MB written, 1.94 ms; 32: 306 MB, 1.98 ms; 64: 227 MB, 2.14 ms (the shift queue); 4 (round 4): 613 MB. Default 32.
            constexpr int burst = 32;      // pixels per store burst (16 / 32 / 64 measured in round 5: 32 ships)
            auto kern = burst == 64 ? fs_dither_pipe_kernel<16> : (burst == 32 ? fs_dither_pipe_kernel<8> : fs_dither_pipe_kernel<4>);
            OCTA_HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void *>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_pipe));
            hipLaunchKernelGGL(kern, dim3((unsigned)B), dim3(64 * nw), lds_pipe, stream, d_in, W, H, n_bands, d_out);
            OCTA_HIP_CHECK(hipGetLastError());
            return 0;
        }
    }
    const int err_bytes = (((W + 1) * 4 + 15) / 16) * 16;
    int rs_words = (W + 3) / 4 + 1;                                    // one spare word; an odd word count keeps the 64 rows on 64 different banks
    if (rs_words % 2 == 0) rs_words++;
    const int rs = 4 * rs_words;
    const int lds_budget = 150 * 1024;
    int band_rows = (lds_budget - err_bytes) / rs;
    if (band_rows > 64) band_rows = 64;
    if (band_rows < 1) { octa::set_error("octa_fs_dither: image too wide (%d) for the LDS-staged kernel", W); return -2; }
    const size_t lds = (size_t)err_bytes + (size_t)band_rows * rs;
    // 4-pixel stores when every image row starts on a 4-byte boundary
    if (W % 4 == 0 && (reinterpret_cast<size_t>(d_out) & 3) == 0) {
        OCTA_HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void *>(fs_dither_kernel<true>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
        hipLaunchKernelGGL(fs_dither_kernel<true>, dim3((unsigned)B), dim3(64), lds, stream, d_in, W, H, band_rows, rs, d_out);
    } else {
        OCTA_HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void *>(fs_dither_kernel<false>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
        hipLaunchKernelGGL(fs_dither_kernel<false>, dim3((unsigned)B), dim3(64), lds, stream, d_in, W, H, band_rows, rs, d_out);
    }
    OCTA_HIP_CHECK(hipGetLastError());
    return 0;
}

extern "C" int octa_max_u8(octa_ctx *ctx, const uint8_t *d_a, const uint8_t *d_b, uint8_t *d_out, size_t n, void *stream_) {
    if (!ctx) { octa::set_error("octa_max_u8: null ctx"); return -2; }
    if (n == 0) return 0;
    hipStream_t stream = (hipStream_t)stream_;
    OCTA_HIP_CHECK(hipSetDevice(ctx->device));
    size_t nthreads = (n + 15) / 16;
    hipLaunchKernelGGL(max_u8_kernel, dim3((unsigned)((nthreads + 255) / 256)), dim3(256), 0, stream, d_a, d_b, d_out, n);
    OCTA_HIP_CHECK(hipGetLastError());
    return 0;
}

// diagnostics: ticks (100 MHz) summed over workgroups for the three phases of the last raster launches
extern "C" int octa_raster_prof(octa_ctx *ctx, int64_t *h_out4) {
    if (!ctx || !h_out4) return -2;
    OCTA_HIP_CHECK(hipSetDevice(ctx->device));
    long tmp[8] = {0};
    if (ctx->r_counters.p) OCTA_HIP_CHECK(hipMemcpy(tmp, ctx->r_counters.p, sizeof(tmp), hipMemcpyDeviceToHost));
    h_out4[0] = tmp[0] & 0xffffffff; h_out4[1] = tmp[1]; h_out4[2] = tmp[2]; h_out4[3] = tmp[3];
    return 0;
}
