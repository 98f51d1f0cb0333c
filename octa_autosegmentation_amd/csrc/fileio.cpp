// fileio.cpp -- the reference's on-disk formats written and read natively (host C++, no GPU work):
//
//   * graph CSV  `node1,node2,radius`  (generate_vessel_graph.py:59-66, forest.py:196-207): positions are
//     str(np.ndarray) of three doubles -- numpy's default print options (precision 8, floatmode 'maxprec', Dragon4 in unique
//     mode) -- the radius is repr(float); rows end in "\r\n" (csv.writer's default dialect). The text is part of the drop-in
//     contract (labels are rendered from positions AS READ BACK from it), so the formatter restates numpy's rules
//     (numpy/_core/arrayprint.py FloatingFormat.fillFormat / __call__) exactly:
//       - the 3-vector is printed in scientific notation iff max|x| >= 1e8, min|x| < 1e-4 or max|x| / min|x| > 1000 over its
//         non-zero entries; otherwise in positional notation;
//       - positional: every entry correctly rounded to 8 decimals (ties to even on the exact binary value), trailing zeros
//         trimmed ("0.5", "1.", "0."), right-padded with blanks to the longest fraction of the vector, left-padded to the
//         longest integer part (sign included);
//       - scientific: all entries with the SAME number p of mantissa decimals, p = the longest mantissa of the vector after
//         rounding to 8 decimals and trimming trailing zeros, zero-filled, exponent of at least two digits.
//     repr(float): shortest digits that round-trip (std::to_chars = Ryu; CPython uses David Gay's dtoa mode 0: both give the
//     shortest, closest digit string), exponent form iff the decimal exponent is < -4 or >= 16, ".0" appended to integers.
//     Checked against numpy / CPython themselves on millions of vectors in tests/test_fileio.py.
//   * reading it back: the "Legacy" string branch of tree2img.py:73-76 / data_transforms.py:369-375 (split on blanks, float())
//     with std::from_chars (correctly rounded like float(); locale-independent).
//   * PNG: 8-bit grey (art_ven_img_gray.png, tree2img.py:282-292) and 1-bit (label PNGs, visualize_vessel_graphs.py:99) through
//     zlib's deflate; readers decode them to the pixels PIL writes (PNG is lossless; the compressed bytes are not part of the
//     contract).
#include <charconv>
#include <cerrno>
#include <cmath>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <atomic>
#include <mutex>
#include <string>
#include <thread>
#include <vector>

#include <sys/stat.h>
#include <sys/types.h>
#include <zlib.h>

#include "common.h"

namespace {

// ---- number formatting --------------------------------------------------------------------------------------------

// round-half-even of x * 1e8 for 0 <= x < 2^22, exactly (error-free product through fma); returns false outside the range
inline bool scaled8(double ax, uint64_t *k_out) {
    if (!(ax < 4194304.0)) return false;
    const double p = ax * 1e8;
    const double lo = std::fma(ax, 1e8, -p);           // ax * 1e8 == p + lo exactly (1e8 is a double), |lo| <= ulp(p) / 2
    double k = std::nearbyint(p);                      // ties to even ON p; hi = p - k is exact and |hi| <= 0.5
    const double hi = p - k;
    // |hi| < 0.5 cannot be carried across a half by lo (hi is a multiple of ulp(p)); hi == +-0.5 is a tie of p only:
    // the exact product lies above / below it according to the sign of lo, and is a true tie (already on the even k) for lo == 0
    if (hi == 0.5 && lo > 0.0) k += 1.0;
    else if (hi == -0.5 && lo < 0.0) k -= 1.0;
    *k_out = (uint64_t)k;
    return true;
}

struct Pos {          // one positional entry: sign, integer digits, fraction digits (trailing zeros trimmed)
    char ip[40];
    int ni;
    char fp[16];
    int nf;
};

inline void positional(double x, Pos *o) {
    const bool neg = std::signbit(x);
    const double ax = std::fabs(x);
    uint64_t k;
    int ni = 0;
    if (neg) o->ip[ni++] = '-';
    if (ax < 1e7 && scaled8(ax, &k)) {      // at most 7 + 8 = 15 significant digits: rounding to 8 decimals never outruns the
        uint64_t ipart = k / 100000000ull, f = k % 100000000ull;   // shortest round-trip digits (Dragon4's unique mode would stop there)
        char tmp[24];
        int nt = 0;
        do { tmp[nt++] = (char)('0' + ipart % 10); ipart /= 10; } while (ipart);
        while (nt) o->ip[ni++] = tmp[--nt];
        char fd[8];
        for (int i = 7; i >= 0; i--) { fd[i] = (char)('0' + f % 10); f /= 10; }
        int nf = 8;
        while (nf > 0 && fd[nf - 1] == '0') nf--;
        memcpy(o->fp, fd, nf);
        o->nf = nf;
    } else {
        // large values: numpy prints the shortest digits that round-trip when they end before the 8th decimal
        char buf[400];
        auto r = std::to_chars(buf, buf + sizeof(buf) - 1, ax, std::chars_format::fixed);
        int n = (int)(r.ptr - buf);
        const char *dot = (const char *)memchr(buf, '.', n);
        int nf_unique = dot ? n - (int)(dot - buf) - 1 : 0;
        if (nf_unique >= 8) {
            n = snprintf(buf, sizeof(buf), "%.8f", ax);             // glibc: exact, ties to even
            dot = (const char *)memchr(buf, '.', n);
        }
        const int nint = dot ? (int)(dot - buf) : n;
        memcpy(o->ip + ni, buf, nint > 36 ? 36 : nint);
        ni += nint > 36 ? 36 : nint;
        int nf = dot ? n - nint - 1 : 0;
        while (nf > 0 && dot[nf] == '0') nf--;
        if (nf) memcpy(o->fp, dot + 1, nf);
        o->nf = nf;
    }
    o->ni = ni;
}

// mantissa decimals numpy's first pass keeps for |x|: rounded to 8 decimals (unique mode stops there at the latest), trailing
// zeros trimmed
inline int trimmed_frac_digits(double ax) {
    char b[48];
    snprintf(b, sizeof(b), "%.8e", ax);                // d.dddddddde+XX
    int nf = 8;
    while (nf > 0 && b[1 + nf] == '0') nf--;
    return nf;
}

// str(np.array([v0, v1, v2])) appended to out; returns the new end
char *format_vec3(const double *v, char *out) {
    double mx = 0.0, mn = INFINITY;
    bool any = false, finite = true;
    for (int i = 0; i < 3; i++) {
        if (!std::isfinite(v[i])) { finite = false; continue; }
        const double a = std::fabs(v[i]);
        if (a != 0.0) { any = true; if (a > mx) mx = a; if (a < mn) mn = a; }
    }
    const bool sci = any && (mx >= 1e8 || mn < 0.0001 || mx / mn > 1000.0);
    *out++ = '[';
    if (!finite) {   // nan / inf never leave the simulator; keep the text recognisable rather than mimic numpy's padding
        for (int i = 0; i < 3; i++) out += sprintf(out, i ? " %g" : "%g", v[i]);
        *out++ = ']';
        return out;
    }
    if (!sci) {
        Pos p[3];
        int pl = 0, pr = 0;
        for (int i = 0; i < 3; i++) { positional(v[i], &p[i]); if (p[i].ni > pl) pl = p[i].ni; if (p[i].nf > pr) pr = p[i].nf; }
        for (int i = 0; i < 3; i++) {
            if (i) *out++ = ' ';
            for (int s = p[i].ni; s < pl; s++) *out++ = ' ';
            memcpy(out, p[i].ip, p[i].ni); out += p[i].ni;
            *out++ = '.';
            memcpy(out, p[i].fp, p[i].nf); out += p[i].nf;
            for (int s = p[i].nf; s < pr; s++) *out++ = ' ';
        }
    } else {
        int prec = 0, pl = 1, ed = 2;
        char w[3][48];
        for (int i = 0; i < 3; i++) { const int u = trimmed_frac_digits(std::fabs(v[i])); if (u > prec) prec = u; if (std::signbit(v[i])) pl = 2; }
        int elen[3];
        for (int i = 0; i < 3; i++) {
            snprintf(w[i], sizeof(w[i]), "%.*e", prec, std::fabs(v[i]));
            if (prec == 0) {                                        // numpy keeps the point: "1.e-05"
                char *e = strchr(w[i], 'e');
                memmove(e + 1, e, strlen(e) + 1);
                *e = '.';
            }
            const char *e = strchr(w[i], 'e');
            elen[i] = (int)strlen(e + 2);
            if (elen[i] > ed) ed = elen[i];
        }
        for (int i = 0; i < 3; i++) {
            if (i) *out++ = ' ';
            const bool neg = std::signbit(v[i]);
            for (int s = neg ? 2 : 1; s < pl; s++) *out++ = ' ';
            if (neg) *out++ = '-';
            const char *e = strchr(w[i], 'e');
            memcpy(out, w[i], e - w[i] + 2); out += e - w[i] + 2;    // mantissa, 'e', sign
            for (int s = elen[i]; s < ed; s++) *out++ = '0';
            memcpy(out, e + 2, elen[i]); out += elen[i];
        }
    }
    *out++ = ']';
    return out;
}

// repr(float)
char *format_repr(double x, char *out) {
    if (std::isnan(x)) { memcpy(out, "nan", 3); return out + 3; }
    if (std::isinf(x)) { const char *s = x < 0 ? "-inf" : "inf"; const size_t n = strlen(s); memcpy(out, s, n); return out + n; }
    if (std::signbit(x)) *out++ = '-';
    const double ax = std::fabs(x);
    if (ax == 0.0) { memcpy(out, "0.0", 3); return out + 3; }
    char b[40];
    auto r = std::to_chars(b, b + sizeof(b), ax, std::chars_format::scientific);
    *r.ptr = 0;
    char *e = strchr(b, 'e');
    const int exp10 = atoi(e + 1);
    char digits[24];
    int nd = 0;
    for (char *c = b; c < e; c++) if (*c != '.') digits[nd++] = *c;
    const int decpt = exp10 + 1;                        // value = 0.d1d2... * 10^decpt
    if (decpt > 16 || decpt < -3) {                     // float_repr_style 'short', format code 'r'
        *out++ = digits[0];
        if (nd > 1) { *out++ = '.'; memcpy(out, digits + 1, nd - 1); out += nd - 1; }
        out += sprintf(out, "e%c%02d", exp10 < 0 ? '-' : '+', std::abs(exp10));
    } else if (decpt <= 0) {
        *out++ = '0'; *out++ = '.';
        for (int i = 0; i < -decpt; i++) *out++ = '0';
        memcpy(out, digits, nd); out += nd;
    } else if (decpt >= nd) {
        memcpy(out, digits, nd); out += nd;
        for (int i = nd; i < decpt; i++) *out++ = '0';
        *out++ = '.'; *out++ = '0';
    } else {
        memcpy(out, digits, decpt); out += decpt;
        *out++ = '.';
        memcpy(out, digits + decpt, nd - decpt); out += nd - decpt;
    }
    return out;
}

constexpr int64_t ROW_BYTES_MAX = 512;     // generous: two vectors of three %.8e-style numbers, a repr, separators
const char CSV_HEADER[] = "node1,node2,radius\r\n";

// ---- PNG -----------------------------------------------------------------------------------------------------------

void put_u32(std::vector<unsigned char> &v, uint32_t x) { v.push_back(x >> 24); v.push_back(x >> 16); v.push_back(x >> 8); v.push_back(x); }

void png_chunk(std::vector<unsigned char> &png, const char *type, const unsigned char *data, size_t n) {
    put_u32(png, (uint32_t)n);
    const size_t at = png.size();
    png.insert(png.end(), type, type + 4);
    if (n) png.insert(png.end(), data, data + n);
    put_u32(png, (uint32_t)crc32(0, png.data() + at, (uInt)(n + 4)));
}

// scratch a caller may keep between images: a writer thread of octa_write_sample_files encodes hundreds of images of the same size, and
// allocating (and page-faulting) ~0.5 MB per image from eight threads at once serialises them on the process's memory-map lock
struct PngScratch { std::vector<unsigned char> raw, z; };

int png_encode(const unsigned char *rows, int w, int h, int bit_depth, size_t row_bytes, int level, std::vector<unsigned char> &png, PngScratch *scratch = nullptr) {
    static const unsigned char sig[8] = {0x89, 'P', 'N', 'G', 0x0d, 0x0a, 0x1a, 0x0a};
    png.assign(sig, sig + 8);
    unsigned char ihdr[13];
    ihdr[0] = w >> 24; ihdr[1] = w >> 16; ihdr[2] = w >> 8; ihdr[3] = w;
    ihdr[4] = h >> 24; ihdr[5] = h >> 16; ihdr[6] = h >> 8; ihdr[7] = h;
    ihdr[8] = (unsigned char)bit_depth; ihdr[9] = 0 /* greyscale */; ihdr[10] = 0; ihdr[11] = 0; ihdr[12] = 0;
    png_chunk(png, "IHDR", ihdr, 13);
    PngScratch local;
    PngScratch &S = scratch ? *scratch : local;
    std::vector<unsigned char> &raw = S.raw, &z = S.z;
    raw.resize((row_bytes + 1) * (size_t)h);
    for (int y = 0; y < h; y++) {
        raw[(row_bytes + 1) * y] = 0;                                      // filter type None
        memcpy(&raw[(row_bytes + 1) * y + 1], rows + row_bytes * y, row_bytes);
    }
    uLongf cap = compressBound((uLong)raw.size());
    if (z.size() < cap) z.resize(cap);
    if (level >= 0) {
        if (compress2(z.data(), &cap, raw.data(), (uLong)raw.size(), level) != Z_OK) return -1;
    } else {
        // "fastest sensible" (round 6): run-length matching only (Z_RLE). On the reference's own files -- a 1216 x 1216 1-bit label, a 304 x 304
        // grey image -- it takes 2.3 / 1.0 ms against 5.2 / 2.6 ms at level 1 and the result is no larger (104.6 / 87.2 KB against 113.6 / 87.3 KB)
        z_stream zs;
        memset(&zs, 0, sizeof(zs));
        if (deflateInit2(&zs, 1, Z_DEFLATED, 15, 8, Z_RLE) != Z_OK) return -1;
        zs.next_in = raw.data(); zs.avail_in = (uInt)raw.size();
        zs.next_out = z.data(); zs.avail_out = (uInt)cap;
        const int rc = deflate(&zs, Z_FINISH);
        cap = (uLongf)zs.total_out;
        deflateEnd(&zs);
        if (rc != Z_STREAM_END) return -1;
    }
    png_chunk(png, "IDAT", z.data(), cap);
    png_chunk(png, "IEND", nullptr, 0);
    return 0;
}

int write_file(const char *path, const void *data, size_t n) {
    FILE *f = fopen(path, "wb");
    if (!f) { octa::set_error("cannot open %s for writing: %s", path, strerror(errno)); return -1; }
    const size_t w = fwrite(data, 1, n, f);
    if (fclose(f) != 0 || w != n) { octa::set_error("short write to %s", path); return -1; }
    return 0;
}

}  // namespace

extern "C" int64_t octa_csv_format_edges(const double *h_edges, int64_t n, char *out, int64_t cap) {
    if (!h_edges || !out || n < 0) { octa::set_error("octa_csv_format_edges: bad arguments"); return -2; }
    if (cap < (int64_t)sizeof(CSV_HEADER) + n * ROW_BYTES_MAX) {
        octa::set_error("octa_csv_format_edges: buffer of %lld bytes is too small, need %lld", (long long)cap,
                        (long long)(sizeof(CSV_HEADER) + n * ROW_BYTES_MAX));
        return -2;
    }
    char *p = out;
    memcpy(p, CSV_HEADER, sizeof(CSV_HEADER) - 1); p += sizeof(CSV_HEADER) - 1;
    for (int64_t i = 0; i < n; i++) {
        const double *e = h_edges + 7 * i;
        p = format_vec3(e, p); *p++ = ',';
        p = format_vec3(e + 3, p); *p++ = ',';
        p = format_repr(e[6], p);
        *p++ = '\r'; *p++ = '\n';
    }
    return (int64_t)(p - out);
}

extern "C" int64_t octa_csv_bytes_bound(int64_t n) { return (int64_t)sizeof(CSV_HEADER) + (n < 0 ? 0 : n) * ROW_BYTES_MAX; }

extern "C" int octa_csv_write_file(const char *path, const double *h_edges, int64_t n) {
    if (!path || (!h_edges && n > 0) || n < 0) { octa::set_error("octa_csv_write_file: bad arguments"); return -2; }
    std::vector<char> buf((size_t)octa_csv_bytes_bound(n));
    const int64_t len = octa_csv_format_edges(h_edges, n, buf.data(), (int64_t)buf.size());
    if (len < 0) return (int)len;
    return write_file(path, buf.data(), (size_t)len);
}

// Rows of "[a b c],[d e f],r": every number with std::from_chars (float()'s value); blanks, brackets and commas separate.
extern "C" int64_t octa_csv_parse_edges(const char *text, int64_t len, double *h_out, int64_t cap_rows) {
    if (!text || len < 0 || (!h_out && cap_rows > 0)) { octa::set_error("octa_csv_parse_edges: bad arguments"); return -2; }
    const char *p = text, *end = text + len;
    // header line
    const char *nl = (const char *)memchr(p, '\n', end - p);
    if (!nl) return 0;
    if (strncmp(p, "node1,node2,radius", 18) != 0) { octa::set_error("octa_csv_parse_edges: header is not node1,node2,radius"); return -3; }
    p = nl + 1;
    int64_t rows = 0;
    std::string line;
    while (p < end) {
        nl = (const char *)memchr(p, '\n', end - p);
        const char *le = nl ? nl : end;
        if (le - p > 1 || (le - p == 1 && *p != '\r')) {
            line.assign(p, le);                        // NUL-terminated copy for strtod
            if (rows >= cap_rows) { octa::set_error("octa_csv_parse_edges: more than %lld rows", (long long)cap_rows); return -4; }
            char *c = &line[0];
            double *o = h_out + 7 * rows;
            int got = 0;
            while (*c && got < 7) {
                while (*c == ' ' || *c == '[' || *c == ']' || *c == ',' || *c == '\r' || *c == '"') c++;
                if (!*c) break;
                // std::from_chars: correctly rounded like float(), independent of the process locale (strtod honours LC_NUMERIC and
                // takes hex floats); a leading '+' is float()'s, not from_chars'
                const char *b = (*c == '+' && c[1] != '-' && c[1] != '+') ? c + 1 : c;
                double v = 0.0;
                const auto fc = std::from_chars(b, line.data() + line.size(), v, std::chars_format::general);
                char *q = (fc.ec == std::errc() || fc.ec == std::errc::result_out_of_range) ? const_cast<char *>(fc.ptr) : c;
                if (fc.ec == std::errc::result_out_of_range) v = strtod(b, nullptr);      // +-inf / 0 / subnormal exactly as float() gives
                if (q == c) { octa::set_error("octa_csv_parse_edges: row %lld: cannot parse '%.24s'", (long long)rows, c); return -3; }
                o[got++] = v;
                c = q;
            }
            if (got != 7) { octa::set_error("octa_csv_parse_edges: row %lld has %d numbers, expected 7", (long long)rows, got); return -3; }
            rows++;
        }
        if (!nl) break;
        p = nl + 1;
    }
    return rows;
}

extern "C" int64_t octa_csv_count_rows(const char *text, int64_t len) {
    if (!text || len < 0) return -2;
    int64_t n = 0;
    for (const char *p = text, *end = text + len; p < end;) {
        const char *nl = (const char *)memchr(p, '\n', end - p);
        const char *le = nl ? nl : end;
        if (le - p > 1 || (le - p == 1 && *p != '\r')) n++;
        if (!nl) break;
        p = nl + 1;
    }
    return n > 0 ? n - 1 : 0;   // minus the header
}

// CPython's random.random() consumed n times: the MT19937 state as random.getstate() reports it (624 words + position) advanced
// by 2 n outputs -- for replaying the draws of the reference's per-edge dropout test (tree2img.py:62,78) without a Python loop.
extern "C" int octa_py_random_advance(uint32_t *state625, int64_t n_draws) {
    if (!state625 || n_draws < 0) { octa::set_error("octa_py_random_advance: bad arguments"); return -2; }
    uint32_t *mt = state625;
    int64_t left = 2 * n_draws;
    uint32_t pos = mt[624];
    if (pos > 624) { octa::set_error("octa_py_random_advance: corrupt state"); return -2; }
    while (left > 0) {
        if (pos >= 624) {
            int kk;
            for (kk = 0; kk < 624 - 397; kk++) { uint32_t y = (mt[kk] & 0x80000000u) | (mt[kk + 1] & 0x7fffffffu); mt[kk] = mt[kk + 397] ^ (y >> 1) ^ ((y & 1u) ? 0x9908b0dfu : 0u); }
            for (; kk < 623; kk++) { uint32_t y = (mt[kk] & 0x80000000u) | (mt[kk + 1] & 0x7fffffffu); mt[kk] = mt[kk - 227] ^ (y >> 1) ^ ((y & 1u) ? 0x9908b0dfu : 0u); }
            uint32_t y = (mt[623] & 0x80000000u) | (mt[0] & 0x7fffffffu);
            mt[623] = mt[396] ^ (y >> 1) ^ ((y & 1u) ? 0x9908b0dfu : 0u);
            pos = 0;
        }
        const int64_t take = left < (int64_t)(624 - pos) ? left : (int64_t)(624 - pos);
        pos += (uint32_t)take;
        left -= take;
    }
    mt[624] = pos;
    return 0;
}

extern "C" int octa_png_write_gray8(const char *path, const uint8_t *h_pixels, int width, int height, int level) {
    if (!path || !h_pixels || width <= 0 || height <= 0) { octa::set_error("octa_png_write_gray8: bad arguments"); return -2; }
    std::vector<unsigned char> png;
    if (png_encode(h_pixels, width, height, 8, (size_t)width, level, png)) { octa::set_error("octa_png_write_gray8: deflate failed"); return -1; }
    return write_file(path, png.data(), png.size());
}

// one row of a mode "1" image: bit 7 of byte x / 8 is pixel x, non-zero = white; eight pixels per step
static void pack_row(const uint8_t *src, int width, unsigned char *dst) {
    int x = 0;
    for (; x + 8 <= width; x += 8) {
        uint64_t t;
        memcpy(&t, src + x, 8);
        t |= t >> 4; t |= t >> 2; t |= t >> 1;
        t &= 0x0101010101010101ull;
        dst[x >> 3] = (unsigned char)((t * 0x8040201008040201ull) >> 56);      // byte i (pixel x + i) -> bit 7 - i
    }
    if (x < width) {
        unsigned char b = 0;
        for (int k = 0; x + k < width; k++) if (src[x + k]) b |= (unsigned char)(0x80u >> k);
        dst[x >> 3] = b;
    }
}

// h_pixels: one byte per pixel, non-zero = white; written as a 1-bit greyscale PNG (Pillow mode "1")
extern "C" int octa_png_write_bits(const char *path, const uint8_t *h_pixels, int width, int height, int level) {
    if (!path || !h_pixels || width <= 0 || height <= 0) { octa::set_error("octa_png_write_bits: bad arguments"); return -2; }
    const size_t rb = ((size_t)width + 7) / 8;
    std::vector<unsigned char> packed(rb * (size_t)height, 0);
    for (int y = 0; y < height; y++) pack_row(h_pixels + (size_t)y * width, width, &packed[rb * y]);
    std::vector<unsigned char> png;
    if (png_encode(packed.data(), width, height, 1, rb, level, png)) { octa::set_error("octa_png_write_bits: deflate failed"); return -1; }
    return write_file(path, png.data(), png.size());
}


// ---- a whole batch's files, written by native threads --------------------------------------------------------------------------------
// generate_vessel_graph.py:43-86 writes, per sample, `<dir>/config.yml`, `<dir>/<name>.csv` and `<dir>/art_ven_img_gray.png`;
// visualize_vessel_graphs.py:95-101 adds the binarised label (here `<dir>/<name>_label.png`). Round 6: ONE call per batch. Submitted
// sample by sample from Python (four futures per sample) the on-disk rate was bound by the interpreter lock the generator threads need
// as well -- 410 - 730 triples/s depending on the host, against 1060 - 1150 complete in HBM. Here the samples are taken from an atomic
// counter by `threads` native threads; nothing of the loop runs under the lock.
namespace {

int mkdir_parents(const std::string &dir) {
    struct stat st;
    if (stat(dir.c_str(), &st) == 0) return S_ISDIR(st.st_mode) ? 0 : -1;
    const size_t slash = dir.find_last_of('/');
    if (slash != std::string::npos && slash > 0 && mkdir_parents(dir.substr(0, slash)) != 0) return -1;
    if (mkdir(dir.c_str(), 0777) != 0 && errno != EEXIST) return -1;
    return 0;
}

}  // namespace

extern "C" int octa_write_sample_files(int64_t n_samples, const char *const *dirs, const char *const *names, const double *h_edges,
                                       const int64_t *edge_off, const uint8_t *h_images, int image_w, int image_h, const uint8_t *h_labels,
                                       int label_w, int label_h, int labels_packed, const char *config_text, int64_t config_len, int png_level, int threads) {
    if (n_samples < 0 || !dirs || !names || (h_edges && !edge_off) || (h_images && (image_w <= 0 || image_h <= 0)) ||
        (h_labels && (label_w <= 0 || label_h <= 0)) || (config_text && config_len < 0)) {
        octa::set_error("octa_write_sample_files: bad arguments");
        return -2;
    }
    if (n_samples == 0) return 0;
    int nt = threads > 0 ? threads : 1;
    if ((int64_t)nt > n_samples) nt = (int)n_samples;
    const int level = png_level;
    std::atomic<int64_t> next{0};
    std::atomic<int> failed{0};
    std::mutex err_lock;
    std::string first_error;
    auto fail = [&](const std::string &what) {
        std::lock_guard<std::mutex> g(err_lock);
        if (!failed.exchange(1)) first_error = what;
    };
    auto work = [&]() {
        std::vector<char> text;
        std::vector<unsigned char> png, packed;
        PngScratch scratch;
        while (!failed.load(std::memory_order_relaxed)) {
            const int64_t k = next.fetch_add(1);
            if (k >= n_samples) break;
            if (!dirs[k] || !names[k]) { fail("octa_write_sample_files: null directory or name"); break; }
            const std::string dir(dirs[k]), name(names[k]);
            if (mkdir_parents(dir) != 0) { fail("cannot create directory " + dir + ": " + strerror(errno)); break; }
            if (config_text && write_file((dir + "/config.yml").c_str(), config_text, (size_t)config_len)) { fail(std::string(octa_last_error())); break; }
            if (h_edges) {
                const int64_t n = edge_off[k + 1] - edge_off[k];
                if (n < 0) { fail("octa_write_sample_files: edge offsets are not ascending"); break; }
                if (text.size() < (size_t)octa_csv_bytes_bound(n)) text.resize((size_t)octa_csv_bytes_bound(n));
                const int64_t len = octa_csv_format_edges(h_edges + 7 * edge_off[k], n, text.data(), (int64_t)text.size());
                if (len < 0 || write_file((dir + "/" + name + ".csv").c_str(), text.data(), (size_t)len)) { fail(std::string(octa_last_error())); break; }
            }
            if (h_images) {
                if (png_encode(h_images + (size_t)k * image_w * image_h, image_w, image_h, 8, (size_t)image_w, level, png, &scratch)) { fail("octa_write_sample_files: deflate failed"); break; }
                if (write_file((dir + "/art_ven_img_gray.png").c_str(), png.data(), png.size())) { fail(std::string(octa_last_error())); break; }
            }
            if (h_labels) {
                const size_t rb = ((size_t)label_w + 7) / 8;
                const unsigned char *rows = h_labels + (size_t)k * rb * label_h;      // packed on the device (octa_pack_bits)
                if (!labels_packed) {
                    packed.resize(rb * (size_t)label_h);
                    const uint8_t *src0 = h_labels + (size_t)k * label_w * label_h;
                    for (int y = 0; y < label_h; y++) pack_row(src0 + (size_t)y * label_w, label_w, &packed[rb * y]);
                    rows = packed.data();
                }
                if (png_encode(rows, label_w, label_h, 1, rb, level, png, &scratch)) { fail("octa_write_sample_files: deflate failed"); break; }
                if (write_file((dir + "/" + name + "_label.png").c_str(), png.data(), png.size())) { fail(std::string(octa_last_error())); break; }
            }
        }
    };
    std::vector<std::thread> pool;
    for (int t = 1; t < nt; t++) pool.emplace_back(work);
    work();
    for (auto &th : pool) th.join();
    if (failed.load()) { octa::set_error("%s", first_error.c_str()); return -1; }
    return 0;
}
