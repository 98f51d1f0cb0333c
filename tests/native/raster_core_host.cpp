// tests/native/raster_core_host.cpp -- TEST ONLY. Compiles csrc/raster_core.h as host C++ and
// evaluates every pixel the way the HIP render kernel does (closed-form cell sums per polygon
// side, 4 adjacent pixels per "lane", edges folded in order), so the kernel's arithmetic can be
// compared with the oracle on a machine without a GPU. Never used by the product path.
#include <cstdlib>
#include <cstring>
#include <vector>
#include "../../octa_autosegmentation_amd/csrc/raster_core.h"

using namespace octa_raster;

extern "C" long octa_corehost_rasterize(const double *edges, long n, int no_pixels_x, int no_pixels_y, int mip_axis,
                                        double min_radius, double max_radius, const unsigned char *keep,
                                        unsigned char *out) {
    const int W = no_pixels_x, H = no_pixels_y;
    int axes[2], k = 0;
    for (int a = 0; a < 3; a++) if (a != mip_axis) axes[k++] = a;
    const int ax_x = axes[1], ax_y = axes[0];
    memset(out, 0, (size_t)W * H);
    std::vector<int4> slots;
    long drawn = 0;
    for (long i = 0; i < n; i++) {
        EdgeMeta m;
        BBox16 bb;
        compute_edge_meta(edges + 7 * i, keep ? keep[i] != 0 : true, W, H, ax_x, ax_y, min_radius, max_radius, &m, &bb);
        if (m.nv == 0 || bb.x0 > bb.x1) continue;
        drawn++;
        slots.assign((size_t)m.nv + EXTRA_SLOTS, make_int4(0, 0, 0, 0));
        int extra = 0;
        double ddx = m.x1 - m.x0, ddy = m.y1 - m.y0;
        double len = sqrt(ddx * ddx + ddy * ddy);
        for (int s = 0; s < m.nv; s++) {
            double ax, ay, bx, by;
            stroke_vertex(m, len, s, &ax, &ay);
            stroke_vertex(m, len, (s + 1 == m.nv) ? 0 : s + 1, &bx, &by);
            SideSink sink;
            sink.n = 0;
            clip_side(sink, (double)W, (double)H, ax, ay, bx, by);
            for (int q = 0; q < sink.n; q++) {
                if (q == 0) slots[s] = sink.piece[0];
                else if (extra < EXTRA_SLOTS) slots[m.nv + extra++] = sink.piece[q];
                else return -100;  // convexity bound violated
            }
        }
        for (int py = bb.y0; py <= bb.y1; py++) {
            for (int px0 = (bb.x0 / 4) * 4; px0 <= bb.x1; px0 += 4) {
                int C[4] = {0, 0, 0, 0}, A[4] = {0, 0, 0, 0};
                for (size_t q = 0; q < slots.size(); q++) {
                    if (slots[q].y == slots[q].w) continue;
                    side_eval<4>(slots[q], py, px0, C, A);
                }
                for (int q = 0; q < 4; q++) {
                    int px = px0 + q;
                    if (px >= W || px < bb.x0 || px > bb.x1) continue;
                    int v = (C[q] << 9) - A[q];
                    int c = v >> 9;
                    if (c < 0) c = -c;
                    if (c > 255) c = 255;
                    unsigned char *p = out + (size_t)py * W + px;
                    *p = (unsigned char)blend_white(*p, (unsigned)c);
                }
            }
        }
    }
    return drawn;
}


// The same image through the ITEM-PARALLEL form of the fold (round 4: hline_cells into per-block accumulators, one blend pass per edge
// and 16 x 16 block), evaluated sequentially: must equal octa_corehost_rasterize pixel for pixel.
extern "C" long octa_corehost_rasterize_acc(const double *edges, long n, int no_pixels_x, int no_pixels_y, int mip_axis,
                                            double min_radius, double max_radius, const unsigned char *keep,
                                            unsigned char *out) {
    const int W = no_pixels_x, H = no_pixels_y;
    int axes[2], k = 0;
    for (int a = 0; a < 3; a++) if (a != mip_axis) axes[k++] = a;
    const int ax_x = axes[1], ax_y = axes[0];
    memset(out, 0, (size_t)W * H);
    std::vector<int4> slots;
    long drawn = 0;
    for (long i = 0; i < n; i++) {
        EdgeMeta m;
        BBox16 bb;
        compute_edge_meta(edges + 7 * i, keep ? keep[i] != 0 : true, W, H, ax_x, ax_y, min_radius, max_radius, &m, &bb);
        if (m.nv == 0 || bb.x0 > bb.x1) continue;
        drawn++;
        slots.assign((size_t)m.nv + EXTRA_SLOTS, make_int4(0, 0, 0, 0));
        int extra = 0;
        double ddx = m.x1 - m.x0, ddy = m.y1 - m.y0;
        double len = sqrt(ddx * ddx + ddy * ddy);
        for (int s = 0; s < m.nv; s++) {
            double ax, ay, bx, by;
            stroke_vertex(m, len, s, &ax, &ay);
            stroke_vertex(m, len, (s + 1 == m.nv) ? 0 : s + 1, &bx, &by);
            SideSink sink;
            sink.n = 0;
            clip_side(sink, (double)W, (double)H, ax, ay, bx, by);
            for (int q = 0; q < sink.n; q++) {
                if (q == 0) slots[s] = sink.piece[0];
                else if (extra < EXTRA_SLOTS) slots[m.nv + extra++] = sink.piece[q];
                else return -100;
            }
        }
        for (int by0 = (bb.y0 / 16) * 16; by0 <= bb.y1; by0 += 16)
            for (int bx0 = (bb.x0 / 16) * 16; bx0 <= bb.x1; bx0 += 16) {
                int cov[16][16], area[16][16], carry[16];
                memset(cov, 0, sizeof(cov)); memset(area, 0, sizeof(area)); memset(carry, 0, sizeof(carry));
                for (size_t q = 0; q < slots.size(); q++) {
                    if (slots[q].y == slots[q].w) continue;
                    for (int r = 0; r < 16; r++) {
                        int hx1, hy1, hx2, hy2;
                        if (!side_row_piece(slots[q], by0 + r, hx1, hy1, hx2, hy2)) continue;
                        carry[r] += hline_cells(hx1, hy1, hx2, hy2, hline_step(hx1, hy1, hx2, hy2), bx0, 16,
                                                [&](int px, int c, int a) { cov[r][px - bx0] += c; area[r][px - bx0] += a; });
                    }
                }
                for (int r = 0; r < 16; r++) {
                    int C = carry[r];
                    for (int c = 0; c < 16; c++) {
                        C += cov[r][c];
                        const int px = bx0 + c, py = by0 + r;
                        if (px >= W || py >= H || px < bb.x0 || px > bb.x1 || py < bb.y0 || py > bb.y1) continue;
                        int v = (C << 9) - area[r][c];
                        int cc = v >> 9;
                        if (cc < 0) cc = -cc;
                        if (cc > 255) cc = 255;
                        unsigned char *p = out + (size_t)py * W + px;
                        *p = (unsigned char)blend_white(*p, (unsigned)cc);
                    }
                }
            }
    }
    return drawn;
}
