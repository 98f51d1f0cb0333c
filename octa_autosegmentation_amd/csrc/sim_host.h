// sim_host.h -- host-side pieces of the simulator: RNG streams, per-sample initialisation, the
// per-iteration parameter table, and the BFS edge export. Plain C++ (no HIP).
//
//   numpy legacy RandomState / CPython random streams   
//   Greenhouse.__init__, init_params_from_config          greenhouse.py:17-51
//   simulation_space_expansion as a table                 greenhouse.py:139-155 (+ loop quirks :78-90)
//   SimulationSpace validity mask                         simulation_space.py:36-54
//   Forest._initialize_tree_stumps                        forest.py:68-181
//   edge list order                                       generate_vessel_graph.py:43-56
#pragma once
#include <algorithm>
#include <cmath>
#include <cstdint>
#include <cstring>
#include <vector>

#include "sim_core.h"

namespace OCTA_SIMK {

struct Mt19937 {
    uint32_t mt[624];
    int idx;
    OCTA_HD void init_genrand(uint32_t s) {
        mt[0] = s;
        for (int i = 1; i < 624; i++) mt[i] = 1812433253u * (mt[i - 1] ^ (mt[i - 1] >> 30)) + (uint32_t)i;
        idx = 624;
    }
    OCTA_HD void init_by_array(const uint32_t *key, int len) {
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
    OCTA_HD void refill() {
        int kk;
        for (kk = 0; kk < 624 - 397; kk++) {
            uint32_t y = (mt[kk] & 0x80000000u) | (mt[kk + 1] & 0x7fffffffu);
            mt[kk] = mt[kk + 397] ^ (y >> 1) ^ ((y & 1u) ? 0x9908b0dfu : 0u);
        }
        for (; kk < 623; kk++) {
            uint32_t y = (mt[kk] & 0x80000000u) | (mt[kk + 1] & 0x7fffffffu);
            mt[kk] = mt[kk - 227] ^ (y >> 1) ^ ((y & 1u) ? 0x9908b0dfu : 0u);
        }
        uint32_t y = (mt[623] & 0x80000000u) | (mt[0] & 0x7fffffffu);
        mt[623] = mt[396] ^ (y >> 1) ^ ((y & 1u) ? 0x9908b0dfu : 0u);
        idx = 0;
    }
    OCTA_HD uint32_t next() {
        if (idx >= 624) refill();
        uint32_t y = mt[idx++];
        y ^= (y >> 11);
        y ^= (y << 7) & 0x9d2c5680u;
        y ^= (y << 15) & 0xefc60000u;
        y ^= (y >> 18);
        return y;
    }
    OCTA_HD double next_double() {
        uint32_t a = next() >> 5, b = next() >> 6;
        return (a * 67108864.0 + b) / 9007199254740992.0;
    }
};

// numpy: randint(0, K) by masked rejection; uniform(lo, hi); legacy polar gauss
OCTA_HD inline uint32_t np_randint(Mt19937 &g, uint32_t K) {
    uint32_t rng = K - 1;
    if (rng == 0) return 0;
    uint32_t mask = rng;
    mask |= mask >> 1; mask |= mask >> 2; mask |= mask >> 4; mask |= mask >> 8; mask |= mask >> 16;
    uint32_t v;
    do { v = g.next() & mask; } while (v > rng);
    return v;
}
inline double np_uniform(Mt19937 &g, double lo, double hi) { return lo + (hi - lo) * g.next_double(); }
inline double np_normal_first(Mt19937 &g, double loc, double scale) {
    double f, x1, x2, r2;
    do {
        x1 = 2.0 * g.next_double() - 1.0;
        x2 = 2.0 * g.next_double() - 1.0;
        r2 = x1 * x1 + x2 * x2;
    } while (r2 >= 1.0 || r2 == 0.0);
    f = std::sqrt(-2.0 * std::log(r2) / r2);
    return loc + scale * (f * x2);  // the cached f*x1 is never consumed by the simulator
}
inline void py_seed(Mt19937 &g, uint64_t a) {
    uint32_t key[2] = {(uint32_t)(a & 0xffffffffu), (uint32_t)(a >> 32)};
    g.init_by_array(key, key[1] ? 2 : 1);
}
inline uint32_t py_randbelow(Mt19937 &g, uint32_t n) {
    int k = 0;
    for (uint32_t t = n; t; t >>= 1) k++;
    uint32_t r = g.next() >> (32 - k);
    while (r >= n) r = g.next() >> (32 - k);
    return r;
}

// candidate stream of one sample: for every iteration N masked-rejection voxel picks, then N x 3
// uniforms; candidate = (voxel + u) / geometry_size (simulation_space.py:57-67,106-110). ONE thread.
OCTA_HD inline void gen_candidates(Mt19937 &g, const unsigned short *valid /*[K][3]*/, uint32_t K, const int *N_per_iter,
                                   int n_iter, int n_max, double *out /*[n_iter][n_max][3]*/, unsigned *idx_scratch /*[n_max]*/,
                                   double gs = 76.0) {
    for (int it = 0; it < n_iter; it++) {
        const int N = N_per_iter[it];
        for (int i = 0; i < N; i++) idx_scratch[i] = np_randint(g, K);
        double *o = out + (size_t)it * n_max * 3;
        for (int i = 0; i < N; i++) {
            double u0 = g.next_double(), u1 = g.next_double(), u2 = g.next_double();
            const unsigned short *v = valid + 3 * idx_scratch[i];
            o[3 * i] = ((double)v[0] + u0) / gs;
            o[3 * i + 1] = ((double)v[1] + u1) / gs;
            o[3 * i + 2] = ((double)v[2] + u2) / gs;
        }
    }
}

struct ModeCfg {
    int I, N;
    double eps_n, eps_s, eps_k, delta_art, delta_ven, gamma_art, gamma_ven, phi, omega, kappa, delta_sigma;
};
struct SimConfig {
    double param_scale, d, r, faz_mean, faz_std, rotation_radius, fc0, fc1, sx, sy, sz;
    int n_trees;
    int walls[4];
    std::vector<ModeCfg> modes;
    std::vector<unsigned char> geometry;   // fixed geometry [gshape[0]][gshape[1]][gshape[2]] (C order), empty = analytic mask
    int gshape[3] = {76, 76, 1};
    int n_wall_list = 0;                   // > 0: the enabled source walls in the order of the configuration's mapping (forest.py:81-84),
    int wall_list[6] = {0, 0, 0, 0, 0, 0}; //      0..5 = x0 x1 y0 y1 z0 z1; 0: walls[] (x0 x1 y0 y1 order)
    bool fixed() const { return !geometry.empty(); }
    // SimulationSpace.geometry_size (simulation_space.py:31, 41)
    int gs() const { return fixed() ? std::max(gshape[0], std::max(gshape[1], gshape[2])) : 76; }
    int forest_type = 0;               // 0 stumps, 1 nerve
    double nc0 = 1e30, nc1 = 1e30, nr = 0;   // nerve_center / nerve_radius as configured (before the division by param_scale)
};

// values live DURING each iteration (the reference loads a mode's raw values and only divides by
// param_scale at the first expansion; d carries over between modes)
inline std::vector<IterParams> build_iter_table(const SimConfig &cfg, SimConst *C) {
    std::vector<IterParams> tab;
    const double ps = cfg.param_scale;
    double d = cfg.d / ps;
    int t = 0;
    int n_max = 0;
    for (size_t m = 0; m < cfg.modes.size(); m++) {
        const ModeCfg &M = cfg.modes[m];
        double eps_n = M.eps_n, eps_s = M.eps_s, eps_k = M.eps_k, da = M.delta_art, dv = M.delta_ven, sigma = 1;
        double orig[6] = {eps_k / ps, eps_n / ps, eps_s / ps, da / ps, dv / ps, d};
        if (M.I <= 0) continue;
        const int t_end = t + M.I;
        for (int tt = t; tt < t_end; tt++) {
            t = tt;
            IterParams P;
            memset(&P, 0, sizeof(P));
            P.t = t; P.first_mode = (m == 0); P.N = M.N;
            P.eps_n = eps_n; P.eps_s = eps_s; P.eps_k = eps_k; P.delta_art = da; P.delta_ven = dv; P.d = d;
            P.gamma_art = M.gamma_art; P.gamma_ven = M.gamma_ven; P.phi = M.phi; P.omega = M.omega; P.kappa = M.kappa;
            tab.push_back(P);
            if (M.N > n_max) n_max = M.N;
            sigma = sigma + M.delta_sigma;
            eps_k = orig[0] / sigma; eps_n = orig[1] / sigma; eps_s = orig[2] / sigma;
            da = orig[3] / sigma; dv = orig[4] / sigma; d = orig[5] / sigma;
            d = std::fmax(d, 0.04 / ps);
        }
    }
    C->ps = ps; C->r = cfg.r / ps; C->rotation_radius = cfg.rotation_radius / ps; C->fc0 = cfg.fc0; C->fc1 = cfg.fc1;
    C->sx = cfg.sx; C->sy = cfg.sy; C->sz = cfg.sz; C->n_iter = (int)tab.size(); C->n_max = n_max;
    C->gs = (double)cfg.gs(); C->fixed = cfg.fixed() ? 1 : 0;
    for (int c = 0; c < 3; c++) C->gshape[c] = cfg.fixed() ? cfg.gshape[c] : 0;
    C->mask = nullptr;
    return tab;
}

struct SampleInit {
    double faz_radius;
    std::vector<unsigned short> valid;   // [K][3] (i, j, k): np.argwhere(geometry)
    Mt19937 np_state;                     // numpy stream after the forest stumps
    std::vector<double> py_u;             // PYCAP pre-drawn random.uniform(0,1) values after the stumps (host builds / tests only)
    Mt19937 py_state;                     // CPython's generator after the stumps: the device draws the uniforms itself (sim.hip)
    bool want_py_u = true;                // false: leave py_u empty (4.2 M draws per 128-sample batch that the device makes faster)
    // stump nodes per forest: root, child per tree
    std::vector<double> pos[2];           // xyz
    int n_nodes[2];
};

// sink-sampling mask of one sample (simulation_space.py:36-54) and, for a fixed geometry, the valid voxels of the wall faces:
// face[a] = pairs of the two remaining voxel indices of np.argwhere(np.take(geometry, 0, axis=a)), C order. (Face 0 for the far walls
// too: `self.shape[along_axis] - 1` of the normalised shape lies in (-1, 0] and np.take truncates it to 0, simulation_space.py:71-72.)
inline void init_mask(const SimConfig &cfg, double faz_radius, SampleInit *S, std::vector<int> face[3]) {
    const double ps = cfg.param_scale;
    const int GS = 76;
    const int gy = (int)std::ceil(cfg.sx * GS), gx = (int)std::ceil(cfg.sy * GS);
    const double fcx = cfg.fc0 * GS, fcy = cfg.fc1 * GS, fr = faz_radius * GS * 0.5;
    // optic-nerve disc: cut out of the mask only when it lies inside the field of view (simulation_space.py:48-50:
    // `all(nerve_center - nerve_radius <= 1)` on the values already divided by param_scale)
    const double nerve_c0 = cfg.nc0 / ps, nerve_c1 = cfg.nc1 / ps, nerve_r = cfg.nr / ps;
    const bool disc = (nerve_c0 - nerve_r <= 1.0) && (nerve_c1 - nerve_r <= 1.0);
    const double ncx = nerve_c0 * GS, ncy = nerve_c1 * GS, nrr = nerve_r * GS;
    S->valid.clear();
    for (int a = 0; a < 3; a++) face[a].clear();
    if (cfg.fixed()) {      // simulation_space.py:29-34: the mask comes from the geometry file
        const int G0 = cfg.gshape[0], G1 = cfg.gshape[1], G2 = cfg.gshape[2];
        auto geo = [&](int i, int j, int k) { return cfg.geometry[((size_t)i * G1 + j) * G2 + k] != 0; };
        for (int i = 0; i < G0; i++)
            for (int j = 0; j < G1; j++)
                for (int k = 0; k < G2; k++)
                    if (geo(i, j, k)) { S->valid.push_back((unsigned short)i); S->valid.push_back((unsigned short)j); S->valid.push_back((unsigned short)k); }
        for (int j = 0; j < G1; j++) for (int k = 0; k < G2; k++) if (geo(0, j, k)) { face[0].push_back(j); face[0].push_back(k); }
        for (int i = 0; i < G0; i++) for (int k = 0; k < G2; k++) if (geo(i, 0, k)) { face[1].push_back(i); face[1].push_back(k); }
        for (int i = 0; i < G0; i++) for (int j = 0; j < G1; j++) if (geo(i, j, 0)) { face[2].push_back(i); face[2].push_back(j); }
        return;
    }
    for (int i = 0; i < gy; i++)
        for (int j = 0; j < gx; j++) {
            bool ok = (j - fcx) * (j - fcx) + (i - fcy) * (i - fcy) > fr * fr;
            if (ok && disc) ok = (j - ncx) * (j - ncx) + (i - ncy) * (i - ncy) > nrr * nrr;
            if (ok) { S->valid.push_back((unsigned short)i); S->valid.push_back((unsigned short)j); S->valid.push_back(0); }
        }
}

// a sample whose generators stand where the caller's do: FAZ radius and stump nodes already drawn (the Greenhouse / Forest
// adapters draw them from the global numpy / CPython generators exactly as the reference's constructors do), `np` / `py` = the
// generators' states afterwards
inline void init_sample_given(const SimConfig &cfg, double faz_radius, const double *pos_art, const double *pos_ven, const Mt19937 &np,
                              Mt19937 py, SampleInit *S) {
    std::vector<int> face[3];
    S->np_state = np;
    S->faz_radius = faz_radius;
    init_mask(cfg, faz_radius, S, face);
    const size_t n = (size_t)2 * cfg.n_trees * 3;
    S->pos[0].assign(pos_art, pos_art + n);
    S->pos[1].assign(pos_ven, pos_ven + n);
    S->n_nodes[0] = S->n_nodes[1] = 2 * cfg.n_trees;
    S->py_state = py;
    if (S->want_py_u) {
        S->py_u.resize(PYCAP);
        for (int i = 0; i < PYCAP; i++) S->py_u[i] = py.next_double();
    }
}

inline void init_sample(const SimConfig &cfg, uint32_t np_seed, uint64_t py_seed_v, SampleInit *S) {
    Mt19937 &np = S->np_state;
    Mt19937 py;
    np.init_genrand(np_seed);
    py_seed(py, py_seed_v);
    const double ps = cfg.param_scale;
    const double d0 = cfg.d / ps;
    S->faz_radius = np_normal_first(np, cfg.faz_mean / ps, cfg.faz_std / ps);
    const int GS = cfg.gs();
    const double nerve_c0 = cfg.nc0 / ps, nerve_c1 = cfg.nc1 / ps, nerve_r = cfg.nr / ps;
    const bool fixed = cfg.fixed();
    std::vector<int> face[3];                      // valid voxels of face 0 along each axis
    init_mask(cfg, S->faz_radius, S, face);
    std::vector<int> walls;
    if (cfg.n_wall_list > 0) walls.assign(cfg.wall_list, cfg.wall_list + cfg.n_wall_list);
    else
        for (int w = 0; w < 4; w++) if (cfg.walls[w]) walls.push_back(w);
    for (int f = 0; f < 2; f++) {
        S->pos[f].clear();
        for (int t = 0; t < cfg.n_trees && cfg.forest_type == 1; t++) {
            // Forest._initialize_tree_stumps_from_nerve (forest.py:38-66): five draws of Python's random() per tree
            const double alpha = 2 * M_PI * py.next_double();
            const double rr = nerve_r * std::sqrt(py.next_double());
            double p[3], dir[3];
            p[0] = rr * std::cos(alpha) + nerve_c1;
            p[1] = rr * std::sin(alpha) + nerve_c0;
            p[2] = py.next_double() * cfg.sz;
            dir[0] = py.next_double() - 0.5;
            dir[1] = py.next_double() - 0.5;
            dir[2] = 0.0;
            const double nrm = std::sqrt(std::fma(dir[2], dir[2], std::fma(dir[1], dir[1], dir[0] * dir[0])));
            for (int c = 0; c < 3; c++) S->pos[f].push_back(p[c]);
            for (int c = 0; c < 3; c++) S->pos[f].push_back(p[c] + dir[c] / nrm * d0);
        }
        for (int t = 0; t < cfg.n_trees && cfg.forest_type == 0; t++) {
            int wall = walls[py_randbelow(py, (uint32_t)walls.size())];
            double p[3], dir[3];
            // fixed geometry (simulation_space.py:70-76): random.choice over the wall face's valid voxels, then the voxel + three
            // numpy uniforms (the one along the wall axis is drawn and dropped)
            double fa = 0, fb = 0;
            if (fixed) {
                const int axis = wall >> 1;
                const std::vector<int> &fc = face[axis];      // octa_sim_create refuses a wall whose face has no valid voxel
                const uint32_t pick = py_randbelow(py, (uint32_t)(fc.size() / 2));
                const double u[3] = {np.next_double(), np.next_double(), np.next_double()};
                const int ca = axis == 0 ? 1 : 0, cb = axis == 2 ? 1 : 2;   // the coordinates left after `del pos_3d[along_axis]`
                fa = (fc[2 * pick] + u[ca]) / GS; fb = (fc[2 * pick + 1] + u[cb]) / GS;
            }
            if (wall == 0 || wall == 1) {
                double y = fixed ? fa : np_uniform(np, 0, cfg.sy), z = fixed ? fb : np_uniform(np, 0, cfg.sz);
                p[0] = wall == 0 ? 0.0 : cfg.sx - 1e-6; p[1] = y; p[2] = z;
                dir[0] = wall == 0 ? np_uniform(np, 0.1, 1) : np_uniform(np, -1, -0.1);
                dir[1] = np_uniform(np, y - d0 > 0 ? -1 : 0, y + d0 < cfg.sy ? 1 : 0);
                dir[2] = np_uniform(np, z - d0 > 0 ? -1 : 0, z + d0 < cfg.sz ? 1 : 0);
            } else if (wall == 2 || wall == 3) {
                double x = fixed ? fa : np_uniform(np, 0, cfg.sx), z = fixed ? fb : np_uniform(np, 0, cfg.sz);
                p[0] = x; p[1] = wall == 2 ? 0.0 : cfg.sy - 1e-6; p[2] = z;
                dir[0] = np_uniform(np, x - d0 > 0 ? -1 : 0, x + d0 < cfg.sx ? 1 : 0);
                dir[1] = wall == 2 ? np_uniform(np, 0.1, 1) : np_uniform(np, -1, -0.1);
                dir[2] = np_uniform(np, z - d0 > 0 ? -1 : 0, z + d0 < cfg.sz ? 1 : 0);
            } else {   // z0 / z1 (forest.py:153-181): only with a geometry file (simulation_space.py:82-87 fails without one)
                const double x = fa, y = fb;
                p[0] = x; p[1] = y; p[2] = wall == 4 ? 0.0 : cfg.sz - 1e-6;
                dir[0] = np_uniform(np, x - d0 > 0 ? -1 : 0, x + d0 < cfg.sx ? 1 : 0);
                dir[1] = np_uniform(np, y - d0 > 0 ? -1 : 0, y + d0 < cfg.sy ? 1 : 0);
                dir[2] = wall == 4 ? np_uniform(np, 0.1, 1) : np_uniform(np, -1, -0.1);
            }
            double nrm = std::sqrt(std::fma(dir[2], dir[2], std::fma(dir[1], dir[1], dir[0] * dir[0])));
            for (int c = 0; c < 3; c++) S->pos[f].push_back(p[c]);
            for (int c = 0; c < 3; c++) S->pos[f].push_back(p[c] + dir[c] / nrm * d0);
        }
        S->n_nodes[f] = 2 * cfg.n_trees;
    }
    S->py_state = py;
    if (S->want_py_u) {
        S->py_u.resize(PYCAP);
        for (int i = 0; i < PYCAP; i++) S->py_u[i] = py.next_double();
    }
}

// BFS per tree, root excluded: rows (node xyz, parent xyz, radius); arterial then venous
inline long export_edges(const double *npos[2], const double *nrad[2], const int *npar[2], const int *nch0[2],
                         const int *nch1[2], const unsigned char *nnch[2], const int n_nodes[2], int n_trees,
                         double *edges, long max_edges, long *n_art_edges) {
    long ne = 0;
    std::vector<int> q;
    for (int f = 0; f < 2; f++) {
        for (int t = 0; t < n_trees; t++) {
            q.clear();
            q.push_back(2 * t);  // roots are nodes 0, 2, 4, ... (root, stump child per tree)
            for (size_t h = 0; h < q.size(); h++) {
                int id = q[h];
                int par = npar[f][id];
                if (par >= 0) {
                    if (ne >= max_edges) return -1;
                    double *e = edges + 7 * ne;
                    for (int c = 0; c < 3; c++) { e[c] = npos[f][3 * id + c]; e[3 + c] = npos[f][3 * par + c]; }
                    e[6] = nrad[f][id];
                    ne++;
                }
                int nc = nnch[f][id];
                if (nc >= 1) q.push_back(nch0[f][id]);
                if (nc >= 2) q.push_back(nch1[f][id]);
            }
        }
        if (f == 0 && n_art_edges) *n_art_edges = ne;
        (void)n_nodes;
    }
    return ne;
}

}  // namespace OCTA_SIMK
