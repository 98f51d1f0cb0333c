// tests/native/sim_core_host.cpp -- TEST ONLY. Runs the simulator phases of csrc/sim_core.h on the
// host with a one-thread "block" (same source the HIP kernels compile) so the phase logic, the
// std::nth_element / CPython-set / glibc-pow restatements and the host-side init can be compared
// with the oracle without a GPU. Never used by the product path.
#include <cstdio>
#include "../../octa_autosegmentation_amd/csrc/glibc_trig.h"
#include <cstdlib>
#include <vector>
#include "../../octa_autosegmentation_amd/csrc/sim_host.h"

using namespace OCTA_SIMK;      // octa_simk, or octa_simk_large when compiled with -DOCTA_SIM_LARGE=1 (the wide-field build)

extern "C" {

typedef void (*bif_cb_t)(const double *pos, const double *atts, int n, double r, double kappa, double d, double *out6);

struct host_sim_params {
    double param_scale, d, r, faz_mean, faz_std, rotation_radius, fc[2], size[3];
    int n_trees, walls[4], n_modes;
    double modes[8][13];
    int forest_type;                 // oracle/sim_oracle.py: SimParams (same layout)
    double nerve_center[2], nerve_radius;
    const unsigned char *geometry;   // geometry file's mask or NULL
    int geometry_shape[3];
    int n_source_walls, source_walls[6];
};

double octa_simcore_gpow(double x, double y) { return octa_gpow::gpow(x, y); }

// gpow vs the C library's pow on n pseudo-random inputs from the simulator's domain; returns mismatches
long octa_simcore_gpow_check(long n, unsigned long long seed) {
    const double ys[8] = {2.55, 2.9, 4.0, 1 / 2.55, 1 / 2.9, 0.25, 2.0, 5.0};
    unsigned long long s = seed ? seed : 88172645463325252ULL;
    long bad = 0;
    for (long i = 0; i < n; i++) {
        s ^= s << 13; s ^= s >> 7; s ^= s << 17;
        double u = (double)(s >> 11) / 9007199254740992.0;
        double y = ys[i & 7], x;
        if ((i & 7) >= 3 && (i & 7) <= 5) x = 1e-12 * pow(1e9, u);
        else if ((i & 7) == 7) x = u * 3 + 1e-9;
        else x = 1e-4 + u * 0.05;
        double a = pow(x, y), m = octa_gpow::gpow(x, y);
        if (a != m && !(a != a && m != m)) bad++;
    }
    return bad;
}

// nth_element restatement vs the real std::nth_element is checked from Python through this hook
void octa_simcore_kd_indices(const double *pts, int n, idx_t *out_idx) {
    std::vector<unsigned char> smem((size_t)SIM_LDS_BYTES + 64);
    Blk b = {0, 1, smem.data()};
    std::vector<idx_t> rank(n);
    std::vector<float> xy((size_t)2 * n + 2);
    double zlo = n ? pts[2] : 0, zhi = zlo;
    for (int i = 1; i < n; i++) { zlo = std::min(zlo, pts[3 * i + 2]); zhi = std::max(zhi, pts[3 * i + 2]); }
    kd_build(b, pts, n, out_idx, rank.data(), xy.data(), zlo, zhi);
}

long octa_simcore_set_order(const double *tuples, const int *ids, int n_ins, int *out) {
    std::vector<unsigned long long> h(SETCAP);
    std::vector<int> k(SETCAP);
    int err = 0;
    PySetView S;
    S.hash = h.data(); S.key = k.data(); S.err = &err; S.cap = (int)k.size();
    pyset_init(S);
    for (int i = 0; i < n_ins; i++) pyset_add(S, ids[i], py_hash_tuple3(v3(tuples[3 * ids[i]], tuples[3 * ids[i] + 1], tuples[3 * ids[i] + 2])));
    long c = 0;
    for (int e = 0; e <= S.mask; e++) if (S.key[e] >= 0) out[c++] = S.key[e];
    return err ? -1 : c;
}

int octa_simcore_host_run(const host_sim_params *hp, unsigned np_seed, unsigned long long py_seed_v, bif_cb_t cb,
                          double *edges_out, long max_edges, long *trace_out, long *info_out /*[8]*/) {
    SimConfig cfg;
    cfg.param_scale = hp->param_scale; cfg.d = hp->d; cfg.r = hp->r; cfg.faz_mean = hp->faz_mean; cfg.faz_std = hp->faz_std;
    cfg.rotation_radius = hp->rotation_radius; cfg.fc0 = hp->fc[0]; cfg.fc1 = hp->fc[1];
    cfg.sx = hp->size[0]; cfg.sy = hp->size[1]; cfg.sz = hp->size[2]; cfg.n_trees = hp->n_trees;
    for (int w = 0; w < 4; w++) cfg.walls[w] = hp->walls[w];
    cfg.forest_type = hp->forest_type; cfg.nc0 = hp->nerve_center[0]; cfg.nc1 = hp->nerve_center[1]; cfg.nr = hp->nerve_radius;
    cfg.n_wall_list = hp->n_source_walls;
    for (int w = 0; w < hp->n_source_walls && w < 6; w++) cfg.wall_list[w] = hp->source_walls[w];
    if (hp->geometry) {
        const int *g = hp->geometry_shape;
        for (int k = 0; k < 3; k++) cfg.gshape[k] = g[k];
        cfg.geometry.assign(hp->geometry, hp->geometry + (size_t)g[0] * g[1] * g[2]);
        const double gs = (double)cfg.gs();
        cfg.sx = g[0] / gs; cfg.sy = g[1] / gs; cfg.sz = g[2] / gs;
    }
    for (int m = 0; m < hp->n_modes; m++) {
        const double *q = hp->modes[m];
        cfg.modes.push_back(ModeCfg{(int)q[0], (int)q[1], q[2], q[3], q[4], q[5], q[6], q[7], q[8], q[9], q[10], q[11], q[12]});
    }
    SimConst C;
    std::vector<IterParams> tab = build_iter_table(cfg, &C);
    C.mask = cfg.fixed() ? cfg.geometry.data() : nullptr;
    SampleInit S;
    init_sample(cfg, np_seed, py_seed_v, &S);

    // arrays
    std::vector<double> npos[2], nrad[2], nkap[2];
    std::vector<int> npar[2], nch0[2], nch1[2];
    std::vector<unsigned char> nnch[2], nact[2];
    SimArrays A;
    SampleScalars sc;
    memset(&sc, 0, sizeof(sc));
    for (int f = 0; f < 2; f++) {
        npos[f].assign((size_t)NCAP * 3, 0); nrad[f].assign(NCAP, 0); nkap[f].assign(NCAP, 0);
        npar[f].assign(NCAP, -1); nch0[f].assign(NCAP, -1); nch1[f].assign(NCAP, -1); nnch[f].assign(NCAP, 0); nact[f].assign(NCAP, 0);
        A.npos[f] = npos[f].data(); A.nrad[f] = nrad[f].data(); A.nkap[f] = nkap[f].data(); A.npar[f] = npar[f].data();
        A.nch0[f] = nch0[f].data(); A.nch1[f] = nch1[f].data(); A.nnch[f] = nnch[f].data(); A.nact[f] = nact[f].data();
    }
    A.sc = &sc;
    sc.faz_radius = S.faz_radius;
    sc.py_cap = PYCAP;
    for (int f = 0; f < 2; f++)
        for (int t = 0; t < cfg.n_trees; t++) {
            int root = add_node(A, f, ld3(&S.pos[f][6 * t]), C.r, -1, 4.0);
            add_node(A, f, ld3(&S.pos[f][6 * t + 3]), C.r, root, 4.0);
        }
    std::vector<double> oxy((size_t)OCAP * 3), co2((size_t)CCAP * 3), cand((size_t)NCANDCAP * 3), tmp_dbl((size_t)OCAP * 3);
    std::vector<double> grid_pts((size_t)GRID_N * 3);
    A.grid_pts = grid_pts.data();
    std::vector<int> nn(OCAP), act_list(NCAP), gnode(GCAP), gstart(GCAP), gcount(GCAP), set_key(SETCAP), tmp_int(OCAP + 2 * NCANDCAP);
    std::vector<unsigned> sorted(SORTCAP), pairs(PCAP);
    std::vector<Rec> rec(GCAP);
    std::vector<int> glist(GCAP), child_group(NCAP, 0);
    A.glist = glist.data(); A.child_group = child_group.data();
    std::vector<idx_t> kd_idx(OCAP), kd_rank(OCAP);
    std::vector<unsigned char> removed(OCAP), ven_near(OCAP);
    std::vector<unsigned long long> hashes(OCAP), set_hash(SETCAP);
    A.oxy = oxy.data(); A.co2 = co2.data(); A.cand = cand.data(); A.py_u = S.py_u.data();
    A.nn = nn.data(); A.act_list = act_list.data(); A.sorted = sorted.data();
    A.gnode = gnode.data(); A.gstart = gstart.data(); A.gcount = gcount.data(); A.rec = rec.data();
    A.kd_idx = kd_idx.data(); A.kd_rank = kd_rank.data(); A.removed = removed.data(); A.ven_near = ven_near.data();
    A.hashes = hashes.data(); A.pairs = pairs.data(); A.set_hash = set_hash.data(); A.set_key = set_key.data();
    A.tmp_int = tmp_int.data(); A.tmp_dbl = tmp_dbl.data();
    std::vector<unsigned> idx_scratch(NCANDCAP + 1);
    const uint32_t Kvox = (uint32_t)(S.valid.size() / 3);

    std::vector<unsigned char> smem((size_t)SIM_LDS_BYTES + 64);
    Blk b = {0, 1, smem.data()};
    const int REQ_CAP = 4096;
    std::vector<BifRequest> reqs(REQ_CAP);
    std::vector<double> results((size_t)REQ_CAP * 6);
    auto serve = [&](int n_req) {
        for (int q = 0; q < n_req && q < REQ_CAP; q++)
            cb(reqs[q].pos, reqs[q].atts, reqs[q].n, reqs[q].r, reqs[q].kappa, reqs[q].d, &results[6 * (size_t)q]);
    };
    for (int it = 0; it < C.n_iter; it++) {
        const IterParams &P = tab[it];
        int req_count = 0;
        { int Nn = P.N; gen_candidates(S.np_state, S.valid.data(), Kvox, &Nn, 1, Nn, cand.data(), idx_scratch.data(), C.gs); }
        phase_sample(b, A, C, P, it);
        phase_assign(b, A, 0, A.oxy, sc.n_oxy, P.delta_art);
        phase_pre(b, A, C, P, 0, A.oxy, reqs.data(), &req_count, REQ_CAP, 0);
        serve(req_count);
        phase_seq(b, A, C, P, 0, A.oxy, results.data());
        phase_satisfy_art(b, A, C, P);
        req_count = 0;
        phase_assign(b, A, 1, A.co2, sc.n_co2, P.delta_ven);
        phase_pre(b, A, C, P, 1, A.co2, reqs.data(), &req_count, REQ_CAP, 0);
        serve(req_count);
        phase_seq(b, A, C, P, 1, A.co2, results.data());
        phase_satisfy_ven(b, A, P);
        if (trace_out) {
            trace_out[4 * it] = sc.n_nodes[0]; trace_out[4 * it + 1] = sc.n_oxy; trace_out[4 * it + 2] = sc.n_nodes[1]; trace_out[4 * it + 3] = sc.n_co2;
        }
    }
    const double *cp[2] = {A.npos[0], A.npos[1]}, *cr[2] = {A.nrad[0], A.nrad[1]};
    const int *cpar[2] = {A.npar[0], A.npar[1]}, *c0[2] = {A.nch0[0], A.nch0[1]}, *c1[2] = {A.nch1[0], A.nch1[1]};
    const unsigned char *cn[2] = {A.nnch[0], A.nnch[1]};
    long n_art = 0;
    long ne = export_edges(cp, cr, cpar, c0, c1, cn, sc.n_nodes, cfg.n_trees, edges_out, max_edges, &n_art);
    info_out[0] = ne; info_out[1] = n_art; info_out[2] = sc.err; info_out[3] = sc.py_pos; info_out[4] = sc.murray_steps;
    if (getenv("OCTA_SIMCORE_VERBOSE")) fprintf(stderr, "murray: steps %ld deferred %ld flush rounds %ld\n", sc.murray_steps, sc.murray_deferred, sc.flush_rounds);
    info_out[5] = sc.n_bif; info_out[6] = sc.respec; info_out[7] = C.n_iter;
    return sc.err ? -10 : 0;
}

// glibc restatements of csrc/glibc_trig.h: out[3 i] = gsin(x[i]), out[3 i + 1] = gcos(x[i]), out[3 i + 2] = gacos(c[i])
void octa_simcore_gtrig(const double *x, const double *c, int n, double *out) {
    for (int i = 0; i < n; i++) {
        out[3 * i] = octa_gtrig::gsin(x[i]);
        out[3 * i + 1] = octa_gtrig::gcos(x[i]);
        out[3 * i + 2] = octa_gtrig::gacos(c[i]);
    }
}

}  // extern "C"
