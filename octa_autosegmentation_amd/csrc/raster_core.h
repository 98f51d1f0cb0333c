// raster_core.h -- the arithmetic of the Agg-exact rasteriser, shared by the HIP kernels
// (raster.hip) and, compiled as plain host C++, by tests/native/raster_core_host.cpp, which lets
// the closed-form cell evaluation be checked on a machine without a GPU. Not a CPU fallback: the
// library never calls these on the host.
#pragma once
#include <cmath>
#include <cstdint>

#if defined(__HIP__) || defined(__HIPCC__)
#define OCTA_HD __host__ __device__
#else
#define OCTA_HD
struct int4 { int x, y, z, w; };
static inline int4 make_int4(int x, int y, int z, int w) { int4 v = {x, y, z, w}; return v; }
#endif

namespace octa_raster {

// finite test without <cmath> overload differences between host and device compilers
OCTA_HD inline bool finite_d(double v) { return (v - v) == 0.0; }

constexpr int EXTRA_SLOTS = 4;  // clip pieces beyond one per polygon side (convexity bound)

struct __attribute__((aligned(8))) EdgeMeta {
    double x0, y0, x1, y1;  // display-space endpoints after clip + snap
    double w;               // half stroke width (pixels)
    int nv;                 // polygon vertices = 2*(n+2); 0 = nothing to draw
    int n;                  // arc points per cap
};

struct __attribute__((aligned(8))) BBox16 {
    short x0, y0, x1, y1;  // inclusive pixel bbox, x0 > x1 when empty
};

// ---- double pipeline pieces (same expression order as oracle/raster_oracle.c) -------------

OCTA_HD inline unsigned clip_flags(double x, double y, double cx1, double cy1, double cx2, double cy2) {
    return (unsigned)(x > cx2) | ((unsigned)(y > cy2) << 1) | ((unsigned)(x < cx1) << 2) | ((unsigned)(y < cy1) << 3);
}
OCTA_HD inline unsigned clip_flags_y(double y, double cy1, double cy2) {
    return ((unsigned)(y > cy2) << 1) | ((unsigned)(y < cy1) << 3);
}

OCTA_HD bool clip_move_point(double x1, double y1, double x2, double y2, double bx1, double by1, double bx2,
                                double by2, double *x, double *y, unsigned flags) {
    double bound;
    if (flags & 5) {
        if (x1 == x2) return false;
        bound = (flags & 4) ? bx1 : bx2;
        *y = (bound - x1) * (y2 - y1) / (x2 - x1) + y1;
        *x = bound;
    }
    flags = clip_flags_y(*y, by1, by2);
    if (flags & 10) {
        if (y1 == y2) return false;
        bound = (flags & 8) ? by1 : by2;
        *x = (bound - y1) * (x2 - x1) / (y2 - y1) + x1;
        *y = bound;
    }
    return true;
}

OCTA_HD unsigned clip_line_segment(double *x1, double *y1, double *x2, double *y2, double bx1, double by1,
                                      double bx2, double by2) {
    unsigned f1 = clip_flags(*x1, *y1, bx1, by1, bx2, by2);
    unsigned f2 = clip_flags(*x2, *y2, bx1, by1, bx2, by2);
    unsigned ret = 0;
    if ((f2 | f1) == 0) return 0;
    if ((f1 & 5) != 0 && (f1 & 5) == (f2 & 5)) return 4;
    if ((f1 & 10) != 0 && (f1 & 10) == (f2 & 10)) return 4;
    double tx1 = *x1, ty1 = *y1, tx2 = *x2, ty2 = *y2;
    if (f1) {
        if (!clip_move_point(tx1, ty1, tx2, ty2, bx1, by1, bx2, by2, x1, y1, f1)) return 4;
        if (*x1 == *x2 && *y1 == *y2) return 4;
        ret |= 1;
    }
    if (f2) {
        if (!clip_move_point(tx1, ty1, tx2, ty2, bx1, by1, bx2, by2, x2, y2, f2)) return 4;
        if (*x1 == *x2 && *y1 == *y2) return 4;
        ret |= 2;
    }
    return ret;
}

OCTA_HD inline int iround_d(double v) { return (int)((v < 0.0) ? v - 0.5 : v + 0.5); }

// Per-edge double pipeline up to the stroke parameters (tree2img.py:65-86 + matplotlib's
// transform / PathNanRemover / PathClipper / PathSnapper). e = 7 doubles.
OCTA_HD inline void compute_edge_meta(const double *e, bool kept, int W, int H, int ax_x, int ax_y, double min_radius,
                                      double max_radius, EdgeMeta *meta_out, BBox16 *bbox_out) {
    EdgeMeta m;
    m.x0 = m.y0 = m.x1 = m.y1 = m.w = 0.0;
    m.nv = 0;
    m.n = 0;
    BBox16 b;
    b.x0 = 1; b.y0 = 1; b.x1 = 0; b.y1 = 0;
    double radius = e[6];
    bool ok = !(radius < min_radius || radius > max_radius);
    if (!kept) ok = false;
    if (ok) {
        radius *= 1.3;
        const double scale_factor = (double)(W > H ? W : H);
        double thickness = radius * scale_factor;
        double lw_px = thickness * 100.0 / 72.0;
        double x0 = e[ax_x] * (double)W, y0 = e[ax_y] * (double)H;
        double x1 = e[3 + ax_x] * (double)W, y1 = e[3 + ax_y] * (double)H;
        ok = finite_d(x0) && finite_d(y0) && finite_d(x1) && finite_d(y1) && finite_d(lw_px);
        if (ok) {
            unsigned moved = clip_line_segment(&x0, &y0, &x1, &y1, -1.0, -1.0, W + 1.0, H + 1.0);
            ok = moved < 4;
        }
        if (ok) {
            if (fabs(x0 - x1) < 1e-4 || fabs(y0 - y1) < 1e-4) {
                int r = (int)(lw_px + ((lw_px >= 0.0) ? 0.5 : -0.5));
                double sv = (r % 2) ? 0.5 : 0.0;
                x0 = floor(x0 + 0.5) + sv; y0 = floor(y0 + 0.5) + sv;
                x1 = floor(x1 + 0.5) + sv; y1 = floor(y1 + 0.5) + sv;
            }
            double ddx = x1 - x0, ddy = y1 - y0;
            double len = sqrt(ddx * ddx + ddy * ddy);
            ok = len > 1e-14;
            if (ok) {
                double w = lw_px * 0.5;
                if (w < 0) w = -w;
                double da = acos(w / (w + 0.125)) * 2.0;
                int n = (int)(M_PI / da);
                // pixel bbox (conservative by one pixel), clamped to the canvas
                double mnx = fmin(x0, x1) - w, mxx = fmax(x0, x1) + w;
                double mny = fmin(y0, y1) - w, mxy = fmax(y0, y1) + w;
                double fx0 = floor(mnx) - 1.0, fx1 = floor(mxx) + 1.0, fy0 = floor(mny) - 1.0, fy1 = floor(mxy) + 1.0;
                int ix0 = fx0 < 0.0 ? 0 : (fx0 > (double)W ? W : (int)fx0);
                int iy0 = fy0 < 0.0 ? 0 : (fy0 > (double)H ? H : (int)fy0);
                int ix1 = fx1 > (double)(W - 1) ? W - 1 : (fx1 < -1.0 ? -1 : (int)fx1);
                int iy1 = fy1 > (double)(H - 1) ? H - 1 : (fy1 < -1.0 ? -1 : (int)fy1);
                if (ix1 >= ix0 && iy1 >= iy0 && n >= 0 && n < 100000000) {
                    m.x0 = x0; m.y0 = y0; m.x1 = x1; m.y1 = y1; m.w = w;
                    m.n = n;
                    m.nv = 2 * (n + 2);
                    b.x0 = (short)ix0; b.y0 = (short)iy0; b.x1 = (short)ix1; b.y1 = (short)iy1;
                }
            }
        }
    }
    *meta_out = m;
    *bbox_out = b;
}

// ---- stroke tessellation + rasteriser clip, one polygon side -------------------------------

// vertex v of the stroke polygon (agg::math_stroke::calc_cap, round cap, two caps back to back)
OCTA_HD void stroke_vertex(const EdgeMeta &m, double len, int v, double *px, double *py) {
    const int per = m.n + 2;
    int cap = v >= per ? 1 : 0;
    int i = v - cap * per;
    double v0x = cap ? m.x1 : m.x0, v0y = cap ? m.y1 : m.y0;
    double v1x = cap ? m.x0 : m.x1, v1y = cap ? m.y0 : m.y1;
    double dx1 = (v1y - v0y) / len;
    double dy1 = (v1x - v0x) / len;
    dx1 *= m.w;
    dy1 *= m.w;
    if (i == 0) {
        *px = v0x - dx1; *py = v0y + dy1;
    } else if (i == per - 1) {
        *px = v0x + dx1; *py = v0y - dy1;
    } else {
        double da = M_PI / (m.n + 1);
        double a1 = atan2(dy1, -dx1);
        a1 += da;
        for (int k = 1; k < i; k++) a1 += da;
        *px = v0x + cos(a1) * m.w;
        *py = v0y + sin(a1) * m.w;
    }
}

// up to 3 fixed-point pieces produced by clipping one polygon side
struct SideSink {
    int4 piece[3];
    int n;
};

OCTA_HD inline void emit_line(SideSink &s, int x1, int y1, int x2, int y2) {
    if (y1 == y2) return;  // horizontal in fixed point: contributes no cover and no area
    if (s.n < 3) s.piece[s.n++] = make_int4(x1, y1, x2, y2);
}

OCTA_HD void line_clip_y(SideSink &s, double H, double x1, double y1, double x2, double y2, unsigned f1, unsigned f2) {
    const double cy1 = 0.0, cy2 = H;
    f1 &= 10; f2 &= 10;
    if ((f1 | f2) == 0) {
        emit_line(s, iround_d(x1 * 256.0), iround_d(y1 * 256.0), iround_d(x2 * 256.0), iround_d(y2 * 256.0));
        return;
    }
    if (f1 == f2) return;
    double tx1 = x1, ty1 = y1, tx2 = x2, ty2 = y2;
    if (f1 & 8) { tx1 = x1 + (cy1 - y1) * (x2 - x1) / (y2 - y1); ty1 = cy1; }
    if (f1 & 2) { tx1 = x1 + (cy2 - y1) * (x2 - x1) / (y2 - y1); ty1 = cy2; }
    if (f2 & 8) { tx2 = x1 + (cy1 - y1) * (x2 - x1) / (y2 - y1); ty2 = cy1; }
    if (f2 & 2) { tx2 = x1 + (cy2 - y1) * (x2 - x1) / (y2 - y1); ty2 = cy2; }
    emit_line(s, iround_d(tx1 * 256.0), iround_d(ty1 * 256.0), iround_d(tx2 * 256.0), iround_d(ty2 * 256.0));
}

// agg::rasterizer_sl_clip<ras_conv_dbl>::line_to for one side, clip box [0,W]x[0,H]
OCTA_HD void clip_side(SideSink &s, double W, double H, double x1, double y1, double x2, double y2) {
    const double cx1 = 0.0, cy1 = 0.0, cx2 = W, cy2 = H;
    unsigned f1 = clip_flags(x1, y1, cx1, cy1, cx2, cy2);
    unsigned f2 = clip_flags(x2, y2, cx1, cy1, cx2, cy2);
    if ((f1 & 10) == (f2 & 10) && (f1 & 10) != 0) return;
    double y3, y4;
    unsigned f3, f4;
    switch (((f1 & 5) << 1) | (f2 & 5)) {
    case 0:
        line_clip_y(s, H, x1, y1, x2, y2, f1, f2);
        break;
    case 1:
        y3 = y1 + (cx2 - x1) * (y2 - y1) / (x2 - x1);
        f3 = clip_flags_y(y3, cy1, cy2);
        line_clip_y(s, H, x1, y1, cx2, y3, f1, f3);
        line_clip_y(s, H, cx2, y3, cx2, y2, f3, f2);
        break;
    case 2:
        y3 = y1 + (cx2 - x1) * (y2 - y1) / (x2 - x1);
        f3 = clip_flags_y(y3, cy1, cy2);
        line_clip_y(s, H, cx2, y1, cx2, y3, f1, f3);
        line_clip_y(s, H, cx2, y3, x2, y2, f3, f2);
        break;
    case 3:
        line_clip_y(s, H, cx2, y1, cx2, y2, f1, f2);
        break;
    case 4:
        y3 = y1 + (cx1 - x1) * (y2 - y1) / (x2 - x1);
        f3 = clip_flags_y(y3, cy1, cy2);
        line_clip_y(s, H, x1, y1, cx1, y3, f1, f3);
        line_clip_y(s, H, cx1, y3, cx1, y2, f3, f2);
        break;
    case 6:
        y3 = y1 + (cx2 - x1) * (y2 - y1) / (x2 - x1);
        y4 = y1 + (cx1 - x1) * (y2 - y1) / (x2 - x1);
        f3 = clip_flags_y(y3, cy1, cy2);
        f4 = clip_flags_y(y4, cy1, cy2);
        line_clip_y(s, H, cx2, y1, cx2, y3, f1, f3);
        line_clip_y(s, H, cx2, y3, cx1, y4, f3, f4);
        line_clip_y(s, H, cx1, y4, cx1, y2, f4, f2);
        break;
    case 8:
        y3 = y1 + (cx1 - x1) * (y2 - y1) / (x2 - x1);
        f3 = clip_flags_y(y3, cy1, cy2);
        line_clip_y(s, H, cx1, y1, cx1, y3, f1, f3);
        line_clip_y(s, H, cx1, y3, x2, y2, f3, f2);
        break;
    case 9:
        y3 = y1 + (cx1 - x1) * (y2 - y1) / (x2 - x1);
        y4 = y1 + (cx2 - x1) * (y2 - y1) / (x2 - x1);
        f3 = clip_flags_y(y3, cy1, cy2);
        f4 = clip_flags_y(y4, cy1, cy2);
        line_clip_y(s, H, cx1, y1, cx1, y3, f1, f3);
        line_clip_y(s, H, cx1, y3, cx2, y4, f3, f4);
        line_clip_y(s, H, cx2, y4, cx2, y2, f4, f2);
        break;
    case 12:
        line_clip_y(s, H, cx1, y1, cx1, y2, f1, f2);
        break;
    default:
        break;
    }
}

// ---- closed-form Agg cell sums ---------------------------------------------------------------

// floor(a / b) for b > 0, |a| < 2^52 and floor(a/b)*b < 2^52: the correctly rounded double quotient
// of two exactly representable integers cannot cross an integer boundary.
OCTA_HD inline double rcp_approx(double b) {
#if defined(__HIP_DEVICE_COMPILE__)
    return __builtin_amdgcn_rcp(b);   // v_rcp_f64, ~1 ulp: the quotient estimate is fixed up exactly below
#else
    return 1.0 / b;
#endif
}
// exact floor(a / b) for b > 0: reciprocal estimate (off by at most one for |a| < 2^50) + integer fix-up
OCTA_HD inline long floordiv_d(long a, long b) {
    long q = (long)floor((double)a * rcp_approx((double)b));
    long r = a - q * b;
    if (r < 0) { q--; r += b; }
    if (r < 0) { q--; r += b; }
    if (r >= b) { q++; r -= b; }
    if (r >= b) { q++; }
    return q;
}
OCTA_HD inline int floordiv_i(int a, int b) {
    int q = (int)floor((double)a * rcp_approx((double)b));
    int r = a - q * b;
    if (r < 0) { q--; r += b; }
    if (r >= b) { q++; }
    return q;
}
// floor(num * dx / dy), dy > 0, |num * dx| < 2^52: the same estimate + fix-up in double arithmetic -- the product, q * dy and the
// remainder are integers below 2^53, so fma(-q, dy, num * dx) is exact; replaces the 64-bit integer multiply / subtract / compare
// sequences of floordiv_d in the per-item code of the render kernel (two of these per scanline piece)
OCTA_HD inline int floordiv_prod(int num, long dx, long dy) {
    const double a = (double)num * (double)dx, b = (double)dy;
    double q = floor(a * rcp_approx(b));
    double r = fma(-q, b, a);
    if (r < 0.0) { q -= 1.0; r += b; }
    if (r < 0.0) { q -= 1.0; r += b; }
    if (r >= b) { q += 1.0; r -= b; }
    if (r >= b) { q += 1.0; }
    return (int)q;
}

// floor(a / b) and the remainder a - q * b in [0, b), b > 0 (|a|, b < 2^31)
OCTA_HD inline int floordivmod_i(int a, int b, int &rem) {
    int q = (int)floor((double)a * rcp_approx((double)b));
    int r = a - q * b;
    if (r < 0) { q--; r += b; }
    if (r >= b) { q++; r -= b; }
    rem = r;
    return q;
}

// The division of a scanline piece that does not depend on the pixel: Agg's DDA step. Crossing one more cell boundary adds
// 256 * dy to the numerator of y = hy1 + floor((.. + k * 256 * dy) / |dx|): quotient `lift` and remainder `rem` of that step.
struct HStep { int lift, rem; };
OCTA_HD inline HStep hline_step(int hx1, int hy1, int hx2, int hy2) {
    HStep st = {0, 0};
    if (hy1 == hy2 || (hx1 >> 8) == (hx2 >> 8)) return st;
    const int dxh = hx2 >= hx1 ? hx2 - hx1 : hx1 - hx2;
    st.lift = floordivmod_i(256 * (hy2 - hy1), dxh, st.rem);
    return st;
}

// One scanline piece (agg render_hline semantics) evaluated for NP adjacent pixels px0..px0+NP-1.
// Adds, per pixel, C += sum of cover over cells with ex <= px, A += area of cell px.
// Y[b] is the fractional y reached when the piece crosses the cell boundary x = (px0 + b) * 256, clamped
// to the piece's ends, in traversal direction; cell px then has cover = Y_out - Y_in.
// The boundaries inside the piece are y_k = hy1 + floor((base + k * 256 * dy) / |dx|), k = 0, 1, ..: ONE division for the first
// boundary of the span, the following ones by the DDA step `st` (quotient + remainder carry): floor((n + s) / d) =
// floor(n / d) + lift + [rem(n) + rem >= d] for any sign of s.
template <int NP>
OCTA_HD inline void hline_eval(int hx1, int hy1, int hx2, int hy2, HStep st, int px0, int (&C)[NP], int (&A)[NP]) {
    if (hy1 == hy2) return;
    const int ex1 = hx1 >> 8, ex2 = hx2 >> 8;
    const int fx1 = hx1 & 255, fx2 = hx2 & 255;
    const int dy = hy2 - hy1;
    if (ex1 == ex2) {
        // the piece stays inside one cell (most sides of a round cap): no boundary crossing, no division
#pragma unroll
        for (int q = 0; q < NP; q++) {
            const int px = px0 + q;
            if (px >= ex1) C[q] += dy;
            if (px == ex1) A[q] += (fx1 + fx2) * dy;
        }
        return;
    }
    int Y[NP + 1];
    if (hx2 >= hx1) {
        // cells ex1 .. ex2 left to right; boundary xb lies between cells xb-1 and xb; inside the piece for ex1 < xb <= ex2
        const int dxh = hx2 - hx1;
        const int k0 = px0 - ex1 - 1 > 0 ? px0 - ex1 - 1 : 0;       // first boundary of the span that can lie inside
        int r, q = floordivmod_i((256 - fx1) * dy + k0 * 256 * dy, dxh, r);
#pragma unroll
        for (int bq = 0; bq <= NP; bq++) {
            const int xb = px0 + bq;
            int v;
            if (xb <= ex1) v = hy1;
            else if (xb > ex2) v = hy2;
            else {
                v = hy1 + q;
                r += st.rem;
                q += st.lift;
                if (r >= dxh) { r -= dxh; q++; }
            }
            Y[bq] = v;
        }
#pragma unroll
        for (int q2 = 0; q2 < NP; q2++) {
            int px = px0 + q2;
            int cov = Y[q2 + 1] - Y[q2];
            C[q2] += Y[q2 + 1] - hy1;
            int fxin = (px == ex1) ? fx1 : 0;
            int fxout = (px == ex2) ? fx2 : 256;
            A[q2] += (fxin + fxout) * cov;
        }
    } else {
        // cells ex1 .. ex2 right to left; boundary xb is crossed when entering cell xb-1; inside the piece for ex2 < xb <= ex1,
        // k = ex1 - xb grows towards the LEFT: walk the span's boundaries from its right end
        const int dxh = hx1 - hx2;
        const int xr = px0 + NP < ex1 ? px0 + NP : ex1;             // rightmost boundary of the span that can lie inside
        int r, q = floordivmod_i(fx1 * dy + (ex1 - xr) * 256 * dy, dxh, r);
#pragma unroll
        for (int bq = NP; bq >= 0; bq--) {
            const int xb = px0 + bq;
            int v;
            if (xb > ex1) v = hy1;
            else if (xb <= ex2) v = hy2;
            else {
                v = hy1 + q;
                r += st.rem;
                q += st.lift;
                if (r >= dxh) { r -= dxh; q++; }
            }
            Y[bq] = v;
        }
#pragma unroll
        for (int q2 = 0; q2 < NP; q2++) {
            int px = px0 + q2;
            int cov = Y[q2] - Y[q2 + 1];
            C[q2] += hy2 - Y[q2 + 1];
            int fxin = (px == ex1) ? fx1 : 256;
            int fxout = (px == ex2) ? fx2 : 0;
            A[q2] += (fxin + fxout) * cov;
        }
    }
}

// One scanline piece walked CELL BY CELL (agg render_hline) inside the pixel columns [bx0, bx0 + bw): emit(px, cover, area) for every
// cell of the piece in that window; returns the cover of the piece's cells LEFT of bx0 (they reach the window's pixels only as cover
// carried in: a pixel's C is the cover of all cells at or left of it). Cells right of the window contribute nothing. Same boundary
// values as hline_eval (Y(xb) = hy1 + floor(... / |dx|), first boundary by one division, the following ones by the DDA step), so
// carry + sum of the emitted covers up to px equals hline_eval's C[px], and the emitted area its A[px] -- integer for integer. This is
// the item-parallel form of the fold (round 4): a lane owns one piece and adds its cells into a per-block accumulator.
template <class F>
OCTA_HD inline int hline_cells(int hx1, int hy1, int hx2, int hy2, HStep st, int bx0, int bw, F &&emit) {
    if (hy1 == hy2) return 0;
    const int ex1 = hx1 >> 8, ex2 = hx2 >> 8;
    const int fx1 = hx1 & 255, fx2 = hx2 & 255;
    const int dy = hy2 - hy1;
    const int bx1 = bx0 + bw - 1;
    if (ex1 == ex2) {
        if (ex1 < bx0) return dy;
        if (ex1 <= bx1) emit(ex1, dy, (fx1 + fx2) * dy);
        return 0;
    }
    if (hx2 >= hx1) {
        // left to right: boundary xb (between cells xb - 1 and xb) inside the piece for ex1 < xb <= ex2,
        // Y(xb) = hy1 + floor(((256 - fx1) + (xb - ex1 - 1) * 256) * dy / dxh)
        if (ex1 > bx1) return 0;
        if (ex2 < bx0) return dy;
        const int dxh = hx2 - hx1;
        int px = ex1 > bx0 ? ex1 : bx0;
        int r, q, yl, carry = 0;
        if (px > ex1) {
            q = floordivmod_i(((256 - fx1) + (px - ex1 - 1) * 256) * dy, dxh, r);      // Y(px) - hy1 = cover of the cells left of px
            yl = hy1 + q;
            carry = q;
            r += st.rem; q += st.lift;
            if (r >= dxh) { r -= dxh; q++; }
        } else {
            yl = hy1;
            q = floordivmod_i((256 - fx1) * dy, dxh, r);                               // Y(ex1 + 1) - hy1
        }
        const int last = ex2 < bx1 ? ex2 : bx1;
        for (; px <= last; px++) {
            const int yr = (px == ex2) ? hy2 : hy1 + q;
            const int cov = yr - yl;
            const int fxin = (px == ex1) ? fx1 : 0, fxout = (px == ex2) ? fx2 : 256;
            emit(px, cov, (fxin + fxout) * cov);
            yl = yr;
            r += st.rem; q += st.lift;
            if (r >= dxh) { r -= dxh; q++; }
        }
        return carry;
    }
    // right to left: cells ex1 down to ex2; boundary xb inside the piece for ex2 < xb <= ex1,
    // Y(xb) = hy1 + floor((fx1 + (ex1 - xb) * 256) * dy / dxh); a cell is entered at its right boundary and left at its left one
    if (ex2 > bx1) return 0;
    if (ex1 < bx0) return dy;
    const int dxh = hx1 - hx2;
    int px = ex1 < bx1 ? ex1 : bx1;
    int r, q, yr;
    if (px < ex1) {
        q = floordivmod_i((fx1 + (ex1 - px - 1) * 256) * dy, dxh, r);                  // Y(px + 1) - hy1
        yr = hy1 + q;
        r += st.rem; q += st.lift;
        if (r >= dxh) { r -= dxh; q++; }
    } else {
        yr = hy1;
        q = floordivmod_i(fx1 * dy, dxh, r);                                           // Y(ex1) - hy1
    }
    const int first = ex2 > bx0 ? ex2 : bx0;
    int yl = yr;
    for (; px >= first; px--) {
        yl = (px == ex2) ? hy2 : hy1 + q;
        const int cov = yl - yr;
        const int fxin = (px == ex1) ? fx1 : 256, fxout = (px == ex2) ? fx2 : 0;
        emit(px, cov, (fxin + fxout) * cov);
        yr = yl;
        r += st.rem; q += st.lift;
        if (r >= dxh) { r -= dxh; q++; }
    }
    return ex2 < bx0 ? hy2 - yl : 0;          // yl = Y(bx0) after the last cell of the window
}

// The piece of one polygon side (24.8 fixed point) that lies in scanline py, in Agg's render_hline terms:
// from (hx1, hy1) to (hx2, hy2) with hy in [0, 256]. Returns false if the side does not touch the scanline
// or the piece has no vertical extent (it then contributes nothing).
OCTA_HD inline bool side_row_piece(int4 s, int py, int &hx1, int &hy1, int &hx2, int &hy2) {
    const int x1 = s.x, y1 = s.y, x2 = s.z, y2 = s.w;
    const int ey1 = y1 >> 8, ey2 = y2 >> 8;
    const int lo = ey1 < ey2 ? ey1 : ey2, hi = ey1 < ey2 ? ey2 : ey1;
    if (py < lo || py > hi) return false;
    const int fy1 = y1 & 255, fy2 = y2 & 255;
    if (ey1 == ey2) {
        hx1 = x1; hy1 = fy1; hx2 = x2; hy2 = fy2;
        return hy1 != hy2;
    }
    const long dx = (long)x2 - (long)x1;
    if (y2 > y1) {
        const long dy = (long)y2 - (long)y1;
        const int K = ey2 - ey1, k = py - ey1;
        hx1 = (k == 0) ? x1 : x1 + floordiv_prod((256 - fy1) + 256 * (k - 1), dx, dy);
        hy1 = (k == 0) ? fy1 : 0;
        hx2 = (k == K) ? x2 : x1 + floordiv_prod((256 - fy1) + 256 * k, dx, dy);
        hy2 = (k == K) ? fy2 : 256;
    } else {
        const long dy = (long)y1 - (long)y2;
        const int K = ey1 - ey2, k = ey1 - py;
        hx1 = (k == 0) ? x1 : x1 + floordiv_prod(fy1 + 256 * (k - 1), dx, dy);
        hy1 = (k == 0) ? fy1 : 256;
        hx2 = (k == K) ? x2 : x1 + floordiv_prod(fy1 + 256 * k, dx, dy);
        hy2 = (k == K) ? fy2 : 0;
    }
    return hy1 != hy2;
}

// One polygon side for scanline py and NP adjacent pixels.
template <int NP>
OCTA_HD inline void side_eval(int4 s, int py, int px0, int (&C)[NP], int (&A)[NP]) {
    int hx1, hy1, hx2, hy2;
    if (side_row_piece(s, py, hx1, hy1, hx2, hy2)) hline_eval<NP>(hx1, hy1, hx2, hy2, hline_step(hx1, hy1, hx2, hy2), px0, C, A);
}

// matplotlib fixed_blender_rgba_plain: white with coverage alpha over an opaque grey p
OCTA_HD inline unsigned blend_white(unsigned p, unsigned alpha) {
    // branch-free (round 5: the four blends per edge and lane were a fifth of the fold's time -- two data-dependent branches and an IEEE
    // division per pixel): alpha 0 and 255 fall out of the select at the end, the quotient comes from the hardware reciprocal estimate
    unsigned r = p * 255u;
    unsigned a = alpha + 65280u;
    unsigned num = ((255u << 8) - r) * alpha + (r << 8);
    // exact num / a (a in [65280, 65535], num < 2^24 * 1.004): float estimate (num and the reciprocal each within ~1 ulp, so the
    // estimate is within 1 of the quotient) + fix-up by the exact remainder
#if defined(__HIP_DEVICE_COMPILE__)
    unsigned q = (unsigned)((float)num * __builtin_amdgcn_rcpf((float)a));
#else
    unsigned q = (unsigned)((float)num * (1.0f / (float)a));
#endif
    int rem = (int)num - (int)(q * a);
    if (rem < 0) { q--; rem += (int)a; }
    if (rem >= (int)a) { q++; }
    return alpha == 0u ? p : (alpha == 255u ? 255u : q);
}


}  // namespace octa_raster
