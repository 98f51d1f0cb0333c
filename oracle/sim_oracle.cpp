/*
 * oracle/sim_oracle.cpp -- TEST INFRASTRUCTURE ONLY (the checker, never the product).
 *
 * Sequential CPU restatement of the reference's space-colonisation vessel-graph simulator:
 *   vessel_graph_generation/greenhouse.py:17-366  (Greenhouse: parameters, develop_forest,
 *       simulation_space_expansion, grow_vessels, sample_oxygen_sinks, assignment)
 *   vessel_graph_generation/forest.py:68-181      (tree stumps on the lateral walls)
 *   vessel_graph_generation/arterial_tree.py:7-229 (Node status, Murray propagation, BFS order)
 *   vessel_graph_generation/simulation_space.py:36-110 (validity mask, candidate sinks)
 *   vessel_graph_generation/element_mesh.py:87-232 (KD_Tree semantics on scipy cKDTree)
 *   generate_vessel_graph.py:43-56                 (edge list order)
 * Third-party behaviour that is part of the result and is restated from the published
 * algorithms (none of it is under /root/reference):
 *   numpy legacy RandomState (MT19937 init_genrand, masked-rejection randint, polar gauss),
 *   CPython `random` (init_by_array seeding, getrandbits/_randbelow choice, 53-bit random()),
 *   scipy 1.15 cKDTree build order (leafsize 16, compact, median split by std::nth_element with
 *   the (value, index) comparator) -> order of query_ball_point results,
 *   CPython 3.10 float/tuple hashing and `set` open addressing -> iteration order of `to_add`.
 * Deliberately simple: brute-force nearest/ball queries over plain arrays (the product uses
 * spatial binning on the GPU), so this file is an independent check.
 * Leaf bifurcation needs numpy's cov + LAPACK dgeev (sign of the eigenvector is part of the
 * result): the Python driver (oracle/sim_oracle.py) passes a callback for it.
 *
 * Parity is PINNED by tests/test_sim_oracle.py against fixtures captured from the imported
 * reference (tools/make_golden_sim.py).
 * Floating point: IEEE double, -ffp-contract=off; fma() written out only where the reference
 * goes through OpenBLAS ddot (np.linalg.norm of a 1-D vector, np.dot of two vectors), SURVEY App. F.
 */
#include <algorithm>
#include <array>
#include <cmath>
#include <cstdint>
#include <cstring>
#include <vector>

namespace {

// ------------------------------------------------------------------ MT19937 streams
struct MT {
    uint32_t mt[624];
    int idx;
    long n_out = 0;
    void init_genrand(uint32_t s) {
        mt[0] = s;
        for (int i = 1; i < 624; i++) mt[i] = 1812433253u * (mt[i - 1] ^ (mt[i - 1] >> 30)) + (uint32_t)i;
        idx = 624;
    }
    void init_by_array(const uint32_t *key, int len) {
        init_genrand(19650218u);
        int i = 1, j = 0;
        int k = 624 > len ? 624 : len;
        for (; k; k--) {
            mt[i] = (mt[i] ^ ((mt[i - 1] ^ (mt[i - 1] >> 30)) * 1664525u)) + key[j] + (uint32_t)j;
            i++; j++;
            if (i >= 624) { mt[0] = mt[623]; i = 1; }
            if (j >= len) j = 0;
        }
        for (k = 623; k; k--) {
            mt[i] = (mt[i] ^ ((mt[i - 1] ^ (mt[i - 1] >> 30)) * 1566083941u)) - (uint32_t)i;
            i++;
            if (i >= 624) { mt[0] = mt[623]; i = 1; }
        }
        mt[0] = 0x80000000u;
        idx = 624;
    }
    uint32_t next() {
        if (idx >= 624) {
            int kk;
            for (kk = 0; kk < 624 - 397; kk++) {
                uint32_t y = (mt[kk] & 0x80000000u) | (mt[kk + 1] & 0x7fffffffu);
                mt[kk] = mt[kk + 397] ^ (y >> 1) ^ ((y & 1u) ? 0x9908b0dfu : 0u);
            }
            for (; kk < 623; kk++) {
                uint32_t y = (mt[kk] & 0x80000000u) | (mt[kk + 1] & 0x7fffffffu);
                mt[kk] = mt[kk + (397 - 624)] ^ (y >> 1) ^ ((y & 1u) ? 0x9908b0dfu : 0u);
            }
            uint32_t y = (mt[623] & 0x80000000u) | (mt[0] & 0x7fffffffu);
            mt[623] = mt[396] ^ (y >> 1) ^ ((y & 1u) ? 0x9908b0dfu : 0u);
            idx = 0;
        }
        uint32_t y = mt[idx++];
        n_out++;
        y ^= (y >> 11);
        y ^= (y << 7) & 0x9d2c5680u;
        y ^= (y << 15) & 0xefc60000u;
        y ^= (y >> 18);
        return y;
    }
    double next_double() {
        uint32_t a = next() >> 5, b = next() >> 6;
        return (a * 67108864.0 + b) / 9007199254740992.0;
    }
};

// numpy legacy RandomState
struct NpRandom {
    MT g;
    bool has_gauss = false;
    double gauss = 0.0;
    void seed(uint32_t s) { g.init_genrand(s); has_gauss = false; gauss = 0.0; }
    double uniform(double lo, double hi) { return lo + (hi - lo) * g.next_double(); }
    double legacy_gauss() {
        if (has_gauss) { double t = gauss; has_gauss = false; gauss = 0.0; return t; }
        double f, x1, x2, r2;
        do {
            x1 = 2.0 * g.next_double() - 1.0;
            x2 = 2.0 * g.next_double() - 1.0;
            r2 = x1 * x1 + x2 * x2;
        } while (r2 >= 1.0 || r2 == 0.0);
        f = std::sqrt(-2.0 * std::log(r2) / r2);
        gauss = f * x1;
        has_gauss = true;
        return f * x2;
    }
    double normal(double loc, double scale) { return loc + scale * legacy_gauss(); }
    // randint(0, K): masked rejection on 32-bit outputs (K-1 <= 0xffffffff)
    uint32_t randint(uint32_t K) {
        uint32_t rng = K - 1;
        if (rng == 0) return 0;
        uint32_t mask = rng;
        mask |= mask >> 1; mask |= mask >> 2; mask |= mask >> 4; mask |= mask >> 8; mask |= mask >> 16;
        uint32_t v;
        do { v = g.next() & mask; } while (v > rng);
        return v;
    }
};

// CPython random.Random
struct PyRandom {
    MT g;
    long n_random = 0;
    void seed(uint64_t a) {
        uint32_t key[2] = {(uint32_t)(a & 0xffffffffu), (uint32_t)(a >> 32)};
        g.init_by_array(key, key[1] ? 2 : 1);
    }
    double random() { n_random++; return g.next_double(); }
    uint32_t getrandbits(int k) { return g.next() >> (32 - k); }
    uint32_t randbelow(uint32_t n) {
        int k = 0;
        for (uint32_t t = n; t; t >>= 1) k++;
        uint32_t r = getrandbits(k);
        while (r >= n) r = getrandbits(k);
        return r;
    }
};

// ------------------------------------------------------------------ small vector helpers
typedef std::array<double, 3> V3;
inline V3 sub(const V3 &a, const V3 &b) { return {a[0] - b[0], a[1] - b[1], a[2] - b[2]}; }
inline V3 add(const V3 &a, const V3 &b) { return {a[0] + b[0], a[1] + b[1], a[2] + b[2]}; }
inline V3 mul(const V3 &a, double s) { return {a[0] * s, a[1] * s, a[2] * s}; }
inline V3 divs(const V3 &a, double s) { return {a[0] / s, a[1] / s, a[2] / s}; }
// np.linalg.norm of a 1-D vector = sqrt(ddot(x,x)) through OpenBLAS: an fma chain (App. F)
inline double norm3(const V3 &v) { return std::sqrt(std::fma(v[2], v[2], std::fma(v[1], v[1], v[0] * v[0]))); }
inline double norm2(double a, double b) { return std::sqrt(std::fma(b, b, a * a)); }
inline double dot3_blas(const V3 &a, const V3 &b) { return std::fma(a[2], b[2], std::fma(a[1], b[1], a[0] * b[0])); }
inline double dot2_blas(double a0, double a1, double b0, double b1) { return std::fma(a1, b1, a0 * b0); }
inline V3 unit(const V3 &v) { return divs(v, norm3(v)); }  // utilities.py:43 norm_vector
// np.linalg.norm(V, axis=1): plain sum of squares
inline double rownorm(const V3 &v) { return std::sqrt((v[0] * v[0] + v[1] * v[1]) + v[2] * v[2]); }
inline V3 cross(const V3 &a, const V3 &b) {
    return {a[1] * b[2] - a[2] * b[1], a[2] * b[0] - a[0] * b[2], a[0] * b[1] - a[1] * b[0]};
}
inline double sqdist(const V3 &a, const V3 &b) {
    double dx = a[0] - b[0], dy = a[1] - b[1], dz = a[2] - b[2];
    return (dx * dx + dy * dy) + dz * dz;  // scipy sqeuclidean_distance_double for m = 3
}
const double RAD2DEG = 180.0 / M_PI, DEG2RAD = M_PI / 180.0;
inline double clip1(double c) { return std::fmin(std::fmax(c, -1.0), 1.0); }
// utilities.py:46-50 angle (degrees) between u and every row of V; comparisons only
inline double angle_uv(const V3 &u, double nu, const V3 &v) {
    double c = ((u[0] * v[0] + u[1] * v[1]) + u[2] * v[2]) / nu / rownorm(v);
    return std::acos(clip1(c)) * RAD2DEG;
}
// utilities.py:52-56 for 2-vectors
inline double angle2(double u0, double u1, double v0, double v1) {
    double c = dot2_blas(u0, u1, v0, v1) / norm2(u0, u1) / norm2(v0, v1);
    return std::acos(clip1(c)) * RAD2DEG;
}

// ------------------------------------------------------------------ forest
struct Node {
    V3 pos;
    double radius;
    double kappa;
    int parent;
    int child[2];
    int nchild;
};
struct Forest {
    std::vector<Node> nodes;
    std::vector<int> roots;
    int add(const V3 &p, double r, int parent, double kappa) {
        Node n;
        n.pos = p; n.radius = r; n.kappa = kappa; n.parent = parent; n.child[0] = n.child[1] = -1; n.nchild = 0;
        nodes.push_back(n);
        int id = (int)nodes.size() - 1;
        if (parent >= 0) {
            Node &pa = nodes[parent];
            if (pa.nchild < 2) pa.child[pa.nchild] = id;
            pa.nchild++;
        }
        return id;
    }
    // arterial_tree.py:174-184
    void murray_to_root(int id, long *steps) {
        while (id >= 0) {
            Node &n = nodes[id];
            if (n.parent < 0 || n.nchild == 0) return;
            double s = 0.0;
            for (int c = 0; c < n.nchild && c < 2; c++) {
                double t = std::pow(nodes[n.child[c]].radius, n.kappa);
                s = (c == 0) ? t : s + t;
            }
            double rp = std::pow(s, 1.0 / n.kappa);
            if (steps) (*steps)++;
            if (n.radius == rp) return;
            n.radius = rp;
            id = n.parent;
        }
    }
};

// ------------------------------------------------------------------ scipy cKDTree index order
struct KdOrder {
    std::vector<int> indices, rank;
    const std::vector<V3> *pts;
    void build_rec(int s, int e) {
        if (e - s <= 16) return;
        double mx[3], mn[3];
        for (int k = 0; k < 3; k++) mx[k] = mn[k] = (*pts)[indices[s]][k];
        for (int j = s + 1; j < e; j++)
            for (int k = 0; k < 3; k++) {
                double t = (*pts)[indices[j]][k];
                mx[k] = mx[k] > t ? mx[k] : t;
                mn[k] = mn[k] < t ? mn[k] : t;
            }
        int d = 0;
        double size = 0;
        for (int k = 0; k < 3; k++)
            if (mx[k] - mn[k] > size) { d = k; size = mx[k] - mn[k]; }
        if (mx[d] == mn[d]) return;
        const std::vector<V3> &P = *pts;
        auto cmp = [&P, d](int a, int b) {
            double pa = P[a][d], pb = P[b][d];
            if (pa == pb) return a < b;
            return pa < pb;
        };
        int np = e - s;
        std::nth_element(indices.begin() + s, indices.begin() + s + np / 2, indices.begin() + e, cmp);
        build_rec(s, s + np / 2);
        build_rec(s + np / 2, e);
    }
    void build(const std::vector<V3> &points) {
        pts = &points;
        int n = (int)points.size();
        indices.resize(n);
        for (int i = 0; i < n; i++) indices[i] = i;
        if (n) build_rec(0, n);
        rank.resize(n);
        for (int i = 0; i < n; i++) rank[indices[i]] = i;
    }
};

// ------------------------------------------------------------------ CPython hashing + set order
inline uint64_t py_hash_double(double v) {
    const uint64_t MOD = ((uint64_t)1 << 61) - 1;
    int e;
    double m = std::frexp(v, &e);
    int sign = 1;
    if (m < 0) { sign = -1; m = -m; }
    uint64_t x = 0;
    while (m) {
        x = ((x << 28) & MOD) | x >> (61 - 28);
        m *= 268435456.0;
        e -= 28;
        uint64_t y = (uint64_t)m;
        m -= (double)y;
        x += y;
        if (x >= MOD) x -= MOD;
    }
    e = e >= 0 ? e % 61 : 61 - 1 - ((-1 - e) % 61);
    x = ((x << e) & MOD) | x >> (61 - e);
    x = x * (uint64_t)(int64_t)sign;
    if (x == (uint64_t)-1) x = (uint64_t)-2;
    return x;
}
inline uint64_t py_hash_tuple3(const V3 &t) {
    const uint64_t P1 = 11400714785074694791ULL, P2 = 14029467366897019727ULL, P5 = 2870177450012600261ULL;
    uint64_t acc = P5;
    for (int i = 0; i < 3; i++) {
        uint64_t lane = py_hash_double(t[i]);
        acc += lane * P2;
        acc = (acc << 31) | (acc >> 33);
        acc *= P1;
    }
    acc += 3ULL ^ (P5 ^ 3527539ULL);
    if (acc == (uint64_t)-1) return 1546275796ULL;
    return acc;
}
struct PySet {  // keys are element ids; no deletions
    struct Entry { int key; uint64_t hash; };
    std::vector<Entry> table;
    size_t mask, fill, used;
    PySet() { table.assign(8, Entry{-1, 0}); mask = 7; fill = used = 0; }
    static void insert_clean(std::vector<Entry> &t, size_t mask, int key, uint64_t hash) {
        size_t perturb = hash, i = (size_t)hash & mask;
        while (true) {
            size_t e = i;
            int probes = (i + 9 <= mask) ? 9 : 0;
            do {
                if (t[e].key < 0) { t[e].key = key; t[e].hash = hash; return; }
                e++;
            } while (probes--);
            perturb >>= 5;
            i = (i * 5 + 1 + perturb) & mask;
        }
    }
    void resize(size_t minused) {
        size_t newsize = 8;
        while (newsize <= minused) newsize <<= 1;
        std::vector<Entry> nt(newsize, Entry{-1, 0});
        for (size_t k = 0; k <= mask; k++)
            if (table[k].key >= 0) insert_clean(nt, newsize - 1, table[k].key, table[k].hash);
        table.swap(nt);
        mask = newsize - 1;
        fill = used;
    }
    void add(int key, uint64_t hash) {
        size_t perturb = hash, i = (size_t)hash & mask;
        while (true) {
            size_t e = i;
            int probes = (i + 9 <= mask) ? 9 : 0;
            do {
                if (table[e].key < 0) {
                    fill++; used++;
                    table[e].key = key; table[e].hash = hash;
                    if (fill * 5 < mask * 3) return;
                    resize(used > 50000 ? used * 2 : used * 4);
                    return;
                }
                if (table[e].hash == hash && table[e].key == key) return;  // same tuple object
                e++;
            } while (probes--);
            perturb >>= 5;
            i = (i * 5 + 1 + perturb) & mask;
        }
    }
};

// ------------------------------------------------------------------ parameters
struct Mode {
    int I, N;
    double eps_n, eps_s, eps_k, delta_art, delta_ven, gamma_art, gamma_ven, phi, omega, kappa, delta_sigma;
};

}  // namespace

extern "C" {

typedef void (*octa_bifurcation_cb)(const double *pos, const double *atts, int n, double r, double kappa, double d,
                                    double *out6);

struct octa_sim_params {
    double param_scale, d, r, faz_radius_mean, faz_radius_std, rotation_radius, faz_center[2];
    double size[3];
    int n_trees;
    int walls[4];  // x0, x1, y0, y1 enabled
    int n_modes;
    double modes[8][13];  // I, N, eps_n, eps_s, eps_k, delta_art, delta_ven, gamma_art, gamma_ven, phi, omega, kappa, delta_sigma
    int forest_type;      // 0 'stumps' (forest.py:68-181), 1 'nerve' (forest.py:38-66)
    double nerve_center[2], nerve_radius;   // as configured (greenhouse.py:28-29 divides them by param_scale)
    const unsigned char *geometry;          // oxygen_sample_geometry_path mask [g0][g1][g2], C order, or NULL (simulation_space.py:29-34)
    int geometry_shape[3];
    int n_source_walls;                     // > 0: the enabled walls in the order of the YAML mapping (forest.py:81-84), 0..5 = x0 x1 y0 y1 z0 z1
    int source_walls[6];                    //      (walls[] is then ignored); the z walls need a geometry file (simulation_space.py:82-87)
};

struct octa_sim_result {
    long n_art_edges, n_ven_edges;
    long n_iter;
    long np_u32_draws, py_random_draws, murray_steps, n_bifurcations, nn_queries, ball_queries;
    double faz_radius;
};

/*
 * Runs one sample. edges_out: capacity max_edges x 7 doubles (node1 xyz, node2 xyz, radius), arterial
 * trees first then venous, BFS per tree (generate_vessel_graph.py:43-56). trace_out (optional): per
 * iteration 4 longs = len(art nodes), len(oxy), len(ven nodes), len(co2) after the iteration
 * (greenhouse.py:129-134). Returns 0, or <0 on error (-2 capacity).
 */
int octa_oracle_simulate(const octa_sim_params *P, uint32_t np_seed, uint64_t py_seed, octa_bifurcation_cb bif_cb,
                         double *edges_out, long max_edges, long *trace_out, long max_trace, double *oxy_out,
                         long max_oxy, long *n_oxy_out, double *co2_out, long max_co2, long *n_co2_out,
                         octa_sim_result *res) {
    NpRandom np;
    PyRandom py;
    np.seed(np_seed);
    py.seed(py_seed);
    memset(res, 0, sizeof(*res));

    // ---- Greenhouse.__init__ (greenhouse.py:17-32)
    const double ps = P->param_scale;
    double d = P->d / ps;
    const double r = P->r / ps;
    const double FAZ_radius = np.normal(P->faz_radius_mean / ps, P->faz_radius_std / ps);
    res->faz_radius = FAZ_radius;
    const double rotation_radius = P->rotation_radius / ps;
    const double fc0 = P->faz_center[0], fc1 = P->faz_center[1];
    // ---- SimulationSpace.__init__ (simulation_space.py:36-54), no nerve disc for the docker config
    const bool fixed = P->geometry != nullptr;
    const int G0 = fixed ? P->geometry_shape[0] : 0, G1 = fixed ? P->geometry_shape[1] : 0, G2 = fixed ? P->geometry_shape[2] : 0;
    if (fixed && (G0 <= 0 || G1 <= 0 || G2 <= 0 || G0 > 65535 || G1 > 65535 || G2 > 65535)) return -4;
    // fixed geometry: geometry_size = max(geometry.shape), shape = geometry.shape / geometry_size (simulation_space.py:31-33)
    const int GS = fixed ? std::max(G0, std::max(G1, G2)) : 76;
    const double sx = fixed ? (double)G0 / GS : P->size[0], sy = fixed ? (double)G1 / GS : P->size[1], sz = fixed ? (double)G2 / GS : P->size[2];
    const int gy = (int)std::ceil(sx * GS), gx = (int)std::ceil(sy * GS);
    const double fcx = fc0 * GS, fcy = fc1 * GS, fr = FAZ_radius * GS * 0.5;
    // greenhouse.py:28-29, simulation_space.py:48-50: the optic-nerve disc leaves the mask when it is inside the field of view
    const double nerve_c0 = P->nerve_center[0] / ps, nerve_c1 = P->nerve_center[1] / ps, nerve_r = P->nerve_radius / ps;
    const bool disc = (nerve_c0 - nerve_r <= 1.0) && (nerve_c1 - nerve_r <= 1.0);
    const double ncx = nerve_c0 * GS, ncy = nerve_c1 * GS, nrr = nerve_r * GS;
    std::vector<std::array<int, 3>> valid;    // np.argwhere(geometry): C order
    // simulation_space.py:70-76: np.argwhere over face 0 along the wall's axis -- for every wall: `shape[axis] - 1` of the NORMALISED
    // shape lies in (-1, 0] and np.take truncates it to 0. face[a] = the two remaining voxel indices, C order of the 2-D face.
    std::vector<std::array<int, 2>> face[3];
    auto geo = [&](int i, int j, int k) { return P->geometry[((size_t)i * G1 + j) * G2 + k] != 0; };
    if (fixed) {
        for (int i = 0; i < G0; i++)
            for (int j = 0; j < G1; j++)
                for (int k = 0; k < G2; k++)
                    if (geo(i, j, k)) valid.push_back({i, j, k});
        for (int j = 0; j < G1; j++) for (int k = 0; k < G2; k++) if (geo(0, j, k)) face[0].push_back({j, k});
        for (int i = 0; i < G0; i++) for (int k = 0; k < G2; k++) if (geo(i, 0, k)) face[1].push_back({i, k});
        for (int i = 0; i < G0; i++) for (int j = 0; j < G1; j++) if (geo(i, j, 0)) face[2].push_back({i, j});
    } else
    for (int i = 0; i < gy; i++)
        for (int j = 0; j < gx; j++) {
            bool ok = (j - fcx) * (j - fcx) + (i - fcy) * (i - fcy) > fr * fr;
            if (ok && disc) ok = (j - ncx) * (j - ncx) + (i - ncy) * (i - ncy) > nrr * nrr;
            if (ok) valid.push_back({i, j, 0});
        }
    const uint32_t K = (uint32_t)valid.size();

    // ---- modes; first mode loaded in __init__ (unscaled values are live during iteration 0)
    std::vector<Mode> modes(P->n_modes);
    for (int m = 0; m < P->n_modes; m++) {
        const double *q = P->modes[m];
        modes[m] = Mode{(int)q[0], (int)q[1], q[2], q[3], q[4], q[5], q[6], q[7], q[8], q[9], q[10], q[11], q[12]};
    }
    double eps_n, eps_s, eps_k, delta_art, delta_ven, gamma_art, gamma_ven, phi, omega, kappa, delta_sigma, sigma_t;
    double orig_scale[6];
    int I, N;
    auto load_mode = [&](const Mode &m) {  // greenhouse.py:34-51
        I = m.I; N = m.N; eps_n = m.eps_n; eps_s = m.eps_s; eps_k = m.eps_k; delta_art = m.delta_art;
        delta_ven = m.delta_ven; gamma_art = m.gamma_art; gamma_ven = m.gamma_ven; phi = m.phi; omega = m.omega;
        kappa = m.kappa; delta_sigma = m.delta_sigma; sigma_t = 1;
        orig_scale[0] = eps_k / ps; orig_scale[1] = eps_n / ps; orig_scale[2] = eps_s / ps;
        orig_scale[3] = delta_art / ps; orig_scale[4] = delta_ven / ps; orig_scale[5] = d;
    };
    load_mode(modes[0]);

    // ---- Forest stumps (forest.py:68-181), arterial then venous
    Forest forest[2];
    std::vector<int> walls;
    if (P->n_source_walls > 0) {
        for (int w = 0; w < P->n_source_walls && w < 6; w++) {
            if (P->source_walls[w] < 0 || P->source_walls[w] > 5) return -3;
            if (P->source_walls[w] >= 4 && !fixed) return -3;   // simulation_space.py:82-87: AttributeError (`valid_pixels`) in the reference
            walls.push_back(P->source_walls[w]);
        }
    } else
        for (int w = 0; w < 4; w++) if (P->walls[w]) walls.push_back(w);
    if (walls.empty() && P->forest_type == 0) return -3;
    for (int f = 0; f < 2; f++) {
        for (int t = 0; t < P->n_trees && P->forest_type == 1; t++) {   // forest.py:38-66
            const double alpha = 2 * M_PI * py.random();
            const double rr = nerve_r * std::sqrt(py.random());
            const double x = rr * std::cos(alpha) + nerve_c1;
            const double y = rr * std::sin(alpha) + nerve_c0;
            const double z = py.random() * sz;
            V3 pos = {x, y, z};
            const double a = py.random() - 0.5, b = py.random() - 0.5;
            V3 dir = {a, b, 0.0};
            dir = mul(divs(dir, norm3(dir)), d);
            int root = forest[f].add(pos, r, -1, 4.0);
            forest[f].roots.push_back(root);
            forest[f].add(add(pos, dir), r, root, 4.0);
        }
        for (int t = 0; t < P->n_trees && P->forest_type == 0; t++) {
            int wall = walls[py.randbelow((uint32_t)walls.size())];
            V3 pos, dir;
            const double d0 = d;
            double fa = 0, fb = 0;
            if (fixed) {   // get_random_valid_position: random.choice over the face's valid voxels + np.random.uniform(0, 1, 3)
                const int axis = wall >> 1;
                const std::vector<std::array<int, 2>> &fc = face[axis];
                if (fc.empty()) return -3;
                const std::array<int, 2> v = fc[py.randbelow((uint32_t)fc.size())];
                const double u[3] = {np.uniform(0, 1), np.uniform(0, 1), np.uniform(0, 1)};
                const int ca = axis == 0 ? 1 : 0, cb = axis == 2 ? 1 : 2;   // the coordinates that stay after `del pos_3d[along_axis]`
                fa = (v[0] + u[ca]) / GS; fb = (v[1] + u[cb]) / GS;
            }
            if (wall == 0 || wall == 1) {
                double y = fixed ? fa : np.uniform(0, sy), z = fixed ? fb : np.uniform(0, sz);
                pos = {wall == 0 ? 0.0 : sx - 1e-6, y, z};
                double a = wall == 0 ? np.uniform(0.1, 1) : np.uniform(-1, -0.1);
                double b = np.uniform(y - d0 > 0 ? -1 : 0, y + d0 < sy ? 1 : 0);
                double c = np.uniform(z - d0 > 0 ? -1 : 0, z + d0 < sz ? 1 : 0);
                dir = {a, b, c};
            } else if (wall == 2 || wall == 3) {
                double x = fixed ? fa : np.uniform(0, sx), z = fixed ? fb : np.uniform(0, sz);
                pos = {x, wall == 2 ? 0.0 : sy - 1e-6, z};
                double a = np.uniform(x - d0 > 0 ? -1 : 0, x + d0 < sx ? 1 : 0);
                double b = wall == 2 ? np.uniform(0.1, 1) : np.uniform(-1, -0.1);
                double c = np.uniform(z - d0 > 0 ? -1 : 0, z + d0 < sz ? 1 : 0);
                dir = {a, b, c};
            } else {       // z0 / z1 (forest.py:153-181), fixed geometry only
                const double x = fa, y = fb;
                pos = {x, y, wall == 4 ? 0.0 : sz - 1e-6};
                double a = np.uniform(x - d0 > 0 ? -1 : 0, x + d0 < sx ? 1 : 0);
                double b = np.uniform(y - d0 > 0 ? -1 : 0, y + d0 < sy ? 1 : 0);
                double c = wall == 4 ? np.uniform(0.1, 1) : np.uniform(-1, -0.1);
                dir = {a, b, c};
            }
            dir = mul(divs(dir, norm3(dir)), d0);
            int root = forest[f].add(pos, r, -1, 4.0);
            forest[f].roots.push_back(root);
            forest[f].add(add(pos, dir), r, root, 4.0);
        }
    }

    // ---- meshes (element_mesh.py KD_Tree: parallel ordered lists)
    std::vector<int> all_nodes[2], active[2];
    for (int f = 0; f < 2; f++)
        for (int i = 0; i < (int)forest[f].nodes.size(); i++) { all_nodes[f].push_back(i); active[f].push_back(i); }
    std::vector<V3> oxy, co2;

    auto nearest_node = [&](const Forest &F, const std::vector<int> &mesh, const V3 &p, double max_dist) -> int {
        res->nn_queries++;
        int best = -1;
        double bd = INFINITY;
        for (int id : mesh) {
            double d2 = sqdist(F.nodes[id].pos, p);
            if (d2 < bd) { bd = d2; best = id; }
        }
        if (best < 0) return -1;
        return std::sqrt(bd) <= max_dist ? best : -1;
    };
    auto nearest_pt_within = [&](const std::vector<V3> &mesh, const V3 &p, double max_dist) -> bool {
        res->nn_queries++;
        double bd = INFINITY;
        for (const V3 &q : mesh) {
            double d2 = sqdist(q, p);
            if (d2 < bd) bd = d2;
        }
        return !mesh.empty() && std::sqrt(bd) <= max_dist;
    };

    // greenhouse.py:309-317
    auto oxygen_distance = [&](double rad) {
        const double c_oxygen = 203.9e-3, kap = 0.02 * c_oxygen, r0 = 3.5e-3;
        double q = rad * ps / r0;
        double c1 = kap * q * std::exp(1 - q);
        return c1 * 6 / ps;
    };

    // ---- growth (greenhouse.py:157-307)
    std::vector<std::vector<int>> asg_lists;
    auto grow = [&](int f, const std::vector<V3> &atts_mesh, double gamma, double delta, bool first_mode, int t,
                    std::vector<int> &new_nodes) {
        Forest &F = forest[f];
        std::vector<int> &act = active[f];
        new_nodes.clear();
        // assignment (greenhouse.py:343-366): insertion-ordered dict node -> attractor indices
        std::vector<int> order;
        std::vector<int> slot(F.nodes.size(), -1);
        asg_lists.clear();
        for (int a = 0; a < (int)atts_mesh.size(); a++) {
            int c = nearest_node(F, act, atts_mesh[a], delta);
            if (c < 0) continue;
            if (slot[c] < 0) { slot[c] = (int)order.size(); order.push_back(c); asg_lists.emplace_back(); }
            asg_lists[slot[c]].push_back(a);
        }
        for (size_t oi = 0; oi < order.size(); oi++) {
            const int id = order[oi];
            const std::vector<int> &al = asg_lists[oi];
            const V3 pos = F.nodes[id].pos;
            const double vc0 = fc0 - pos[0], vc1 = fc1 - pos[1];
            const double dist_to_center = norm2(vc0, vc1);
            const bool is_leaf = F.nodes[id].nchild == 0;
            const bool is_inter = F.nodes[id].parent >= 0 && F.nodes[id].nchild == 1;
            if (is_leaf) {
                V3 v = sub(pos, F.nodes[F.nodes[id].parent].pos);
                double nv = norm3(v);
                std::vector<int> kept;
                std::vector<double> ang;
                double lim = std::fmax(gamma / 2, 0.0);
                for (int a : al) {
                    double an = angle_uv(v, nv, sub(atts_mesh[a], pos));
                    if (an <= lim) { kept.push_back(a); ang.push_back(an); }
                }
                if (kept.empty()) continue;
                V3 avg = {0, 0, 0};
                for (size_t k = 0; k < kept.size(); k++) {
                    V3 u = unit(sub(atts_mesh[kept[k]], pos));
                    avg = k == 0 ? u : add(avg, u);
                }
                double mean = 0;
                for (double a : ang) mean += a;
                mean /= (double)ang.size();
                double var = 0;
                for (double a : ang) var += (a - mean) * (a - mean);
                double sd = std::sqrt(var / (double)ang.size());
                bool bif = false;
                if (sd > phi) {
                    if (FAZ_radius == 0) bif = true;
                    else {
                        double u = py.random();  // random.uniform(0,1)
                        if (std::pow(dist_to_center / (2 * FAZ_radius), 5.0) > u && angle2(vc0, vc1, avg[0], avg[1]) > 90) bif = true;
                    }
                }
                if (bif) {
                    std::vector<double> buf(kept.size() * 3);
                    for (size_t k = 0; k < kept.size(); k++)
                        for (int c = 0; c < 3; c++) buf[3 * k + c] = atts_mesh[kept[k]][c];
                    double out6[6];
                    bif_cb(pos.data(), buf.data(), (int)kept.size(), r, kappa, d, out6);
                    res->n_bifurcations++;
                    new_nodes.push_back(F.add({out6[0], out6[1], out6[2]}, r, id, kappa));
                    new_nodes.push_back(F.add({out6[3], out6[4], out6[5]}, r, id, kappa));
                    F.murray_to_root(id, &res->murray_steps);
                    act.erase(std::find(act.begin(), act.end(), id));
                } else {
                    V3 g = add(mul(unit(v), omega), mul(unit(avg), 1 - omega));
                    if (rotation_radius > 0 && t > 15) {
                        g = unit(g);
                        double cn = norm2(vc0, vc1);
                        double cv0 = vc0 / cn, cv1 = vc1 / cn;
                        V3 np_ = add(pos, mul(g, d));
                        double dist_new = norm2(fc0 - np_[0], fc1 - np_[1]);
                        double weight = std::fmax(first_mode ? 0.0 : 0.01, rotation_radius - dist_new);
                        weight = std::sqrt(weight);
                        V3 ort = {-cv1, cv0, 0};
                        if (angle2(g[0], g[1], ort[0], ort[1]) > 90) ort = mul(ort, -1.0);
                        V3 outv = {-cv0, -cv1, 0};
                        g = add(add(mul(g, 1 - weight), mul(ort, 0.7 * weight)), mul(outv, 0.3 * weight));
                    }
                    V3 pk = add(pos, mul(unit(g), d));
                    new_nodes.push_back(F.add(pk, r, id, kappa));
                }
            } else if (is_inter) {
                const int ch = F.nodes[id].child[0];
                double r1 = F.nodes[ch].radius, r2 = r;
                double rp = std::pow(std::pow(r1, kappa) + std::pow(r2, kappa), 1 / kappa);
                double rp4 = std::pow(rp, 4.0), rp2 = std::pow(rp, 2.0);
                double phi1 = std::acos((rp4 + std::pow(r1, 4.0) - std::pow(r2, 4.0)) / (2 * rp2 * std::pow(r1, 2.0))) * RAD2DEG;
                double phi2 = std::acos((rp4 + std::pow(r2, 4.0) - std::pow(r1, 4.0)) / (2 * rp2 * std::pow(r2, 2.0))) * RAD2DEG;
                V3 dist_seg = sub(F.nodes[ch].pos, pos);
                V3 prox_seg = sub(pos, F.nodes[F.nodes[id].parent].pos);
                double nd = norm3(dist_seg), npx = norm3(prox_seg);
                std::vector<int> kept;
                double lo = phi1 + phi2 - gamma / 2, hi = phi1 + phi2 + gamma / 2, pl = phi2 + gamma / 2;
                for (int a : al) {
                    V3 w = sub(atts_mesh[a], pos);
                    double ad = angle_uv(dist_seg, nd, w), ap = angle_uv(prox_seg, npx, w);
                    if (lo <= ad && ad <= hi && ap <= pl) kept.push_back(a);
                }
                if (kept.empty()) continue;
                V3 avg = {0, 0, 0};
                for (size_t k = 0; k < kept.size(); k++) {
                    V3 u = unit(sub(atts_mesh[kept[k]], pos));
                    avg = k == 0 ? u : add(avg, u);
                }
                V3 dv = unit(dist_seg);
                V3 cr = cross(dv, avg);
                if (cr[0] == 0 && cr[1] == 0 && cr[2] == 0) continue;
                double u = py.random();
                if (std::pow(dist_to_center / (2 * FAZ_radius), 5.0) <= u && angle2(vc0, vc1, avg[0], avg[1]) <= 90) continue;
                V3 k = unit(cr);
                double th = phi2 * DEG2RAD;
                double ct = std::cos(th), st = std::sin(th);
                V3 kxd = cross(k, dv);
                V3 vrot = add(add(mul(dv, ct), mul(kxd, st)), mul(mul(k, dot3_blas(k, dv)), 1 - ct));
                V3 g = add(mul(unit(vrot), omega), mul(unit(avg), 1 - omega));
                V3 pk = add(pos, mul(unit(g), d));
                new_nodes.push_back(F.add(pk, r, id, kappa));
                F.murray_to_root(id, &res->murray_steps);
                act.erase(std::find(act.begin(), act.end(), id));
            }
        }
    };

    // ---- main loop (greenhouse.py:57-137)
    int t = 0;
    long it = 0;
    std::vector<int> new_nodes;
    KdOrder kd;
    for (int m = 0; m < P->n_modes; m++) {
        if (m != 0) load_mode(modes[m]);  // greenhouse.py:84-85 (mode names are distinct)
        if (I <= 0) continue;
        const bool first_mode = (m == 0);
        const int t_end = t + I;
        for (int tt = t; tt < t_end; tt++) {
            t = tt;
            // 1. sample oxygen sinks (greenhouse.py:319-341)
            {
                const double en = std::fmax(eps_n, eps_k), es = eps_s;
                std::vector<uint32_t> idx(N);
                for (int i = 0; i < N; i++) idx[i] = np.randint(K);
                std::vector<V3> cand;
                for (int i = 0; i < N; i++) {
                    double u0 = np.uniform(0, 1), u1 = np.uniform(0, 1), u2 = np.uniform(0, 1);
                    V3 p = {(valid[idx[i]][0] + u0) / GS, (valid[idx[i]][1] + u1) / GS, (valid[idx[i]][2] + u2) / GS};
                    // simulation_space.py:89-98 (the FAZ test compares unit coordinates with the voxel-unit centre)
                    if (p[0] >= sx || p[1] >= sy || p[2] >= sz || p[0] < 0 || p[1] < 0 || p[2] < 0) continue;
                    if (fixed) {   // geometry[(pos * geometry_size).astype(np.uint16)] > 0: the product may land one voxel below
                        const int vi = (int)(uint16_t)(p[0] * GS), vj = (int)(uint16_t)(p[1] * GS), vk = (int)(uint16_t)(p[2] * GS);
                        if (vi >= G0 || vj >= G1 || vk >= G2 || !geo(vi, vj, vk)) continue;
                    } else {
                        double dd = std::sqrt((p[0] - fcx) * (p[0] - fcx) + (p[1] - fcy) * (p[1] - fcy));
                        if (!(dd > fr)) continue;
                    }
                    cand.push_back(p);
                }
                std::vector<V3> to_add;
                for (const V3 &c : cand) {
                    bool ok = true;
                    res->ball_queries++;
                    for (int id : all_nodes[0]) {
                        const Node &nd = forest[0].nodes[id];
                        double d2 = sqdist(nd.pos, c);
                        if (d2 <= en * en) {
                            if (!(std::sqrt(d2) > oxygen_distance(nd.radius))) { ok = false; break; }
                        }
                    }
                    if (!ok) continue;
                    if (nearest_pt_within(oxy, c, es)) continue;
                    for (const V3 &a : to_add)
                        if (!(rownorm(sub(c, a)) > es)) { ok = false; break; }
                    if (ok) to_add.push_back(c);
                }
                oxy.insert(oxy.end(), to_add.begin(), to_add.end());
            }
            // 2. arterial growth
            grow(0, oxy, gamma_art, delta_art, first_mode, t, new_nodes);
            all_nodes[0].insert(all_nodes[0].end(), new_nodes.begin(), new_nodes.end());
            active[0].insert(active[0].end(), new_nodes.begin(), new_nodes.end());
            // 3. satisfied sinks -> CO2 sources (greenhouse.py:98-112)
            {
                std::vector<char> removed(oxy.size(), 0);
                PySet to_add;
                if (!new_nodes.empty() && !oxy.empty()) kd.build(oxy);
                std::vector<std::pair<int, int>> hits;
                for (int nid : new_nodes) {
                    const V3 &np_ = forest[0].nodes[nid].pos;
                    res->ball_queries++;
                    hits.clear();
                    for (int o = 0; o < (int)oxy.size(); o++)
                        if (sqdist(oxy[o], np_) <= eps_k * eps_k) hits.push_back({kd.rank[o], o});
                    std::sort(hits.begin(), hits.end());
                    for (auto &h : hits) {
                        int o = h.second;
                        removed[o] = 1;
                        if (nearest_node(forest[1], all_nodes[1], oxy[o], eps_k) < 0) to_add.add(o, py_hash_tuple3(oxy[o]));
                    }
                }
                for (size_t k = 0; k <= to_add.mask; k++)
                    if (to_add.table[k].key >= 0) co2.push_back(oxy[to_add.table[k].key]);
                size_t w = 0;
                for (size_t o = 0; o < oxy.size(); o++)
                    if (!removed[o]) oxy[w++] = oxy[o];
                oxy.resize(w);
            }
            // 4. venous growth, 5. CO2 removal
            grow(1, co2, gamma_ven, delta_ven, first_mode, t, new_nodes);
            all_nodes[1].insert(all_nodes[1].end(), new_nodes.begin(), new_nodes.end());
            active[1].insert(active[1].end(), new_nodes.begin(), new_nodes.end());
            {
                std::vector<char> removed(co2.size(), 0);
                for (int nid : new_nodes) {
                    const V3 &np_ = forest[1].nodes[nid].pos;
                    res->ball_queries++;
                    for (size_t o = 0; o < co2.size(); o++)
                        if (sqdist(co2[o], np_) <= eps_k * eps_k) removed[o] = 1;
                }
                size_t w = 0;
                for (size_t o = 0; o < co2.size(); o++)
                    if (!removed[o]) co2[w++] = co2[o];
                co2.resize(w);
            }
            // 6. expansion (greenhouse.py:139-155)
            sigma_t = sigma_t + delta_sigma;
            eps_k = orig_scale[0] / sigma_t; eps_n = orig_scale[1] / sigma_t; eps_s = orig_scale[2] / sigma_t;
            delta_art = orig_scale[3] / sigma_t; delta_ven = orig_scale[4] / sigma_t; d = orig_scale[5] / sigma_t;
            d = std::fmax(d, 0.04 / ps);
            if (trace_out && it < max_trace) {
                trace_out[4 * it + 0] = (long)all_nodes[0].size();
                trace_out[4 * it + 1] = (long)oxy.size();
                trace_out[4 * it + 2] = (long)all_nodes[1].size();
                trace_out[4 * it + 3] = (long)co2.size();
            }
            it++;
        }
    }
    res->n_iter = it;
    res->np_u32_draws = np.g.n_out;
    res->py_random_draws = py.n_random;

    // ---- edge export: BFS per tree, exclude root (arterial_tree.py:226-229)
    long ne = 0;
    for (int f = 0; f < 2; f++) {
        const Forest &F = forest[f];
        for (int root : F.roots) {
            std::vector<int> q;
            q.push_back(root);
            for (size_t h = 0; h < q.size(); h++) {
                const Node &n = F.nodes[q[h]];
                if (n.parent >= 0) {
                    if (ne >= max_edges) return -2;
                    double *e = edges_out + 7 * ne;
                    for (int c = 0; c < 3; c++) { e[c] = n.pos[c]; e[3 + c] = F.nodes[n.parent].pos[c]; }
                    e[6] = n.radius;
                    ne++;
                }
                for (int c = 0; c < n.nchild && c < 2; c++) q.push_back(n.child[c]);
            }
        }
        if (f == 0) res->n_art_edges = ne;
    }
    res->n_ven_edges = ne - res->n_art_edges;
    if (n_oxy_out) *n_oxy_out = (long)oxy.size();
    if (oxy_out) for (size_t i = 0; i < oxy.size() && (long)i < max_oxy; i++) for (int c = 0; c < 3; c++) oxy_out[3 * i + c] = oxy[i][c];
    if (n_co2_out) *n_co2_out = (long)co2.size();
    if (co2_out) for (size_t i = 0; i < co2.size() && (long)i < max_co2; i++) for (int c = 0; c < 3; c++) co2_out[3 * i + c] = co2[i][c];
    return 0;
}

// known-answer hooks for the RNG / hashing restatements
void octa_oracle_np_stream(uint32_t seed, int n_u32, uint32_t *out_u32, int n_dbl, double *out_dbl, uint32_t K, int n_ri,
                           uint32_t *out_ri, int n_norm, double *out_norm) {
    NpRandom r; r.seed(seed);
    for (int i = 0; i < n_u32; i++) out_u32[i] = r.g.next();
    for (int i = 0; i < n_dbl; i++) out_dbl[i] = r.g.next_double();
    for (int i = 0; i < n_ri; i++) out_ri[i] = r.randint(K);
    for (int i = 0; i < n_norm; i++) out_norm[i] = r.normal(0.5, 2.0);
}
void octa_oracle_py_stream(uint64_t seed, int n_dbl, double *out_dbl, uint32_t n_choice, int n_ch, uint32_t *out_ch) {
    PyRandom r; r.seed(seed);
    for (int i = 0; i < n_dbl; i++) out_dbl[i] = r.random();
    for (int i = 0; i < n_ch; i++) out_ch[i] = r.randbelow(n_choice);
}
uint64_t octa_oracle_hash_tuple3(const double *t) { return py_hash_tuple3({t[0], t[1], t[2]}); }
// iteration order of a CPython set after inserting n tuples (ids 0..n-1, with repeats allowed through `ids`)
long octa_oracle_set_order(const double *tuples, const int *ids, int n_ins, int *out) {
    PySet s;
    for (int i = 0; i < n_ins; i++) s.add(ids[i], py_hash_tuple3({tuples[3 * ids[i]], tuples[3 * ids[i] + 1], tuples[3 * ids[i] + 2]}));
    long k = 0;
    for (size_t e = 0; e <= s.mask; e++) if (s.table[e].key >= 0) out[k++] = s.table[e].key;
    return k;
}
void octa_oracle_kd_indices(const double *pts, int n, int *out) {
    std::vector<V3> P(n);
    for (int i = 0; i < n; i++) P[i] = {pts[3 * i], pts[3 * i + 1], pts[3 * i + 2]};
    KdOrder kd; kd.build(P);
    for (int i = 0; i < n; i++) out[i] = kd.indices[i];
}

}  // extern "C"
