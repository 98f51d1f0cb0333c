// sim_api.cpp -- the public octa_sim_* entry points of include/octa_hip.h (round 3): csrc/sim.hip is compiled twice, once for the
// 3 x 3 mm^2 configurations (tables in LDS, 16-bit indices, two samples per CU: suffix S) and once with OCTA_SIM_LARGE for wide
// fields of view such as the reference's 12 x 12 mm^2 notebook run (32-bit indices, 64-bit kd elements, table area in HBM: suffix
// L). A simulator is bound to one build when it is created; every other call forwards to it.
//
// Choice at octa_sim_create: the large build when the configuration can outgrow the default capacities (14 336 nodes per forest,
// 13 312 live sinks): more than 8192 / 3 candidates per iteration on a field wider than 6 mm (param_scale > 6), and more than 40
// iterations; OCTA_SIM_BUILD=large / default overrides (tests run the short optic-nerve fixtures through BOTH).
#include <cstdlib>
#include <cstring>
#include <new>

#include "common.h"

extern "C" {
struct octa_simS_impl;
struct octa_simL_impl;
#define OCTA_DECL(SFX, T)                                                                                                                       \
    int octa_sim##SFX##_create(octa_ctx *, const octa_sim_config *, int, T **);                                                                 \
    void octa_sim##SFX##_destroy(T *);                                                                                                          \
    int octa_sim##SFX##_run(T *, const uint32_t *, const uint64_t *, octa_bif_fn, void *, void *);                                              \
    int octa_sim##SFX##_run_states(T *, const double *, const double *, const uint32_t *, const uint32_t *, octa_bif_fn, void *, void *);       \
    int octa_sim##SFX##_np_state(T *, int, uint32_t *);                                                                                         \
    int octa_sim##SFX##_edge_offsets(T *, int64_t *, int64_t *);                                                                                \
    int octa_sim##SFX##_export_edges(T *, double *);                                                                                            \
    int octa_sim##SFX##_export_edges_device(T *, double *, void *);                                                                             \
    int octa_sim##SFX##_trace(T *, int32_t *);                                                                                                  \
    int octa_sim##SFX##_stats(T *, int64_t *);                                                                                                  \
    int octa_sim##SFX##_spans(T *, int64_t *);                                                                                                  \
    int octa_sim##SFX##_timing(T *, double *);                                                                                                  \
    int octa_sim##SFX##_service_stats(T *, double *);                                                                                           \
    int octa_sim##SFX##_geometry(int, int *);                                                                                                   \
    int octa_sim##SFX##_fields(T *, int, double *, int64_t, int64_t *, double *, int64_t, int64_t *);
OCTA_DECL(S, octa_simS_impl)
OCTA_DECL(L, octa_simL_impl)
#undef OCTA_DECL
int octa_simS_kat_kd_order(octa_ctx *, const double *, int64_t, const uint8_t *, int32_t *);
}

// Persistent-kernel launches of this process so far, both builds (round 5). A generator that has just finished its own launch waits for the
// NEXT launch to be on the GPU before it enqueues its rasterisation (pipeline.py): the render workgroups (145 KB of LDS: a CU to themselves)
// and the next launch's 512 simulator workgroups otherwise race for the CUs the finished launch has just left -- when the rasteriser wins,
// the whole rasterisation (45 ms per 512 triples) runs before the launch can place its workgroups: one launch in three measured 462
// instead of 410 ms.
//
// Round 6: the order is kept ON THE DEVICE. Every launch takes a ticket (octa_sim_next_ticket) and its workgroups sign in on a small block
// of device memory (octa_sim_launch_flag: [0] = highest ticket whose first workgroup has started, [1 + t % 2] = workgroups of ticket t that
// have started, [3 + t % 2] = its grid size); a rasterisation that must come behind launch t is preceded, on its own stream, by a one-wave
// gate kernel that leaves when launch t is resident (csrc/order.hip: octa_order_wait_launch). Round 5 polled octa_sim_launch_count from
// Python and slept "to give the dispatcher time": two of two driver-command runs of that build sat in its slow mode.
#include <atomic>
#include <mutex>
static std::atomic<long long> g_sim_launches{0};
extern "C" long long octa_sim_next_ticket(void) { return g_sim_launches.fetch_add(1, std::memory_order_acq_rel) + 1; }
extern "C" long long octa_sim_launch_count(void) { return g_sim_launches.load(std::memory_order_acquire); }
extern "C" int *octa_sim_launch_flag(int device) {
    static std::mutex m;
    static int *flags[64] = {};
    if (device < 0 || device >= 64) return nullptr;
    std::lock_guard<std::mutex> g(m);
    if (!flags[device]) {
        int prev = 0;
        if (hipGetDevice(&prev) != hipSuccess || hipSetDevice(device) != hipSuccess) return nullptr;
        int *p = nullptr;
        if (hipMalloc(&p, 64) != hipSuccess || hipMemset(p, 0, 64) != hipSuccess || hipDeviceSynchronize() != hipSuccess) { (void)hipSetDevice(prev); return nullptr; }
        (void)hipSetDevice(prev);
        flags[device] = p;
    }
    return flags[device];
}

struct octa_sim {
    bool large = false;
    octa_simS_impl *s = nullptr;
    octa_simL_impl *l = nullptr;
};

static bool wants_large(const octa_sim_config *c) {
    if (const char *e = getenv("OCTA_SIM_BUILD")) {
        if (!strcmp(e, "large")) return true;
        if (!strcmp(e, "default")) return false;
    }
    if (c->param_scale <= 6.0) return false;
    int iters = 0, nmax = 0;
    for (int m = 0; m < c->n_modes && m < 8; m++) {
        const int I = (int)c->modes[m][0], N = (int)c->modes[m][1];
        if (I > 0) { iters += I; nmax = N > nmax ? N : nmax; }
    }
    return nmax > 8192 / 3 && iters > 40;       // a short run on a wide field (the nerve fixtures: 12 + 6 and 20 iterations) fits the default build
}

#define FWD(call_s, call_l) (sim->large ? (call_l) : (call_s))

extern "C" int octa_sim_create_ex(octa_ctx *ctx, const octa_sim_config *cfg, int B, int build, octa_sim **out) {
    if (!out || !cfg || build < 0 || build > 2) { octa::set_error("octa_sim_create: bad arguments"); return -2; }
    *out = nullptr;
    octa_sim *sim = new (std::nothrow) octa_sim();
    if (!sim) { octa::set_error("octa_sim_create: out of host memory"); return -1; }
    sim->large = build == 0 ? wants_large(cfg) : build == 2;
    const int rc = sim->large ? octa_simL_create(ctx, cfg, B, &sim->l) : octa_simS_create(ctx, cfg, B, &sim->s);
    if (rc) { delete sim; return rc; }
    *out = sim;
    return 0;
}
extern "C" int octa_sim_create(octa_ctx *ctx, const octa_sim_config *cfg, int B, octa_sim **out) { return octa_sim_create_ex(ctx, cfg, B, 0, out); }
extern "C" void octa_sim_destroy(octa_sim *sim) {
    if (!sim) return;
    if (sim->large) octa_simL_destroy(sim->l); else octa_simS_destroy(sim->s);
    delete sim;
}
#define NEED(sim, what) if (!(sim)) { octa::set_error(what ": null simulator"); return -2; }
extern "C" int octa_sim_run(octa_sim *sim, const uint32_t *a, const uint64_t *b, octa_bif_fn f, void *u, void *st) {
    NEED(sim, "octa_sim_run") return FWD(octa_simS_run(sim->s, a, b, f, u, st), octa_simL_run(sim->l, a, b, f, u, st));
}
extern "C" int octa_sim_run_states(octa_sim *sim, const double *fz, const double *stp, const uint32_t *a, const uint32_t *b, octa_bif_fn f, void *u, void *st) {
    NEED(sim, "octa_sim_run_states") return FWD(octa_simS_run_states(sim->s, fz, stp, a, b, f, u, st), octa_simL_run_states(sim->l, fz, stp, a, b, f, u, st));
}
extern "C" int octa_sim_np_state(octa_sim *sim, int k, uint32_t *o) { NEED(sim, "octa_sim_np_state") return FWD(octa_simS_np_state(sim->s, k, o), octa_simL_np_state(sim->l, k, o)); }
extern "C" int octa_sim_edge_offsets(octa_sim *sim, int64_t *a, int64_t *b) { NEED(sim, "octa_sim_edge_offsets") return FWD(octa_simS_edge_offsets(sim->s, a, b), octa_simL_edge_offsets(sim->l, a, b)); }
extern "C" int octa_sim_export_edges(octa_sim *sim, double *e) { NEED(sim, "octa_sim_export_edges") return FWD(octa_simS_export_edges(sim->s, e), octa_simL_export_edges(sim->l, e)); }
extern "C" int octa_sim_export_edges_device(octa_sim *sim, double *e, void *st) {
    NEED(sim, "octa_sim_export_edges_device") return FWD(octa_simS_export_edges_device(sim->s, e, st), octa_simL_export_edges_device(sim->l, e, st));
}
extern "C" int octa_sim_trace(octa_sim *sim, int32_t *t) { NEED(sim, "octa_sim_trace") return FWD(octa_simS_trace(sim->s, t), octa_simL_trace(sim->l, t)); }
extern "C" int octa_sim_stats(octa_sim *sim, int64_t *t) { NEED(sim, "octa_sim_stats") return FWD(octa_simS_stats(sim->s, t), octa_simL_stats(sim->l, t)); }
extern "C" int octa_sim_spans(octa_sim *sim, int64_t *t) { NEED(sim, "octa_sim_spans") return FWD(octa_simS_spans(sim->s, t), octa_simL_spans(sim->l, t)); }
extern "C" int octa_sim_timing(octa_sim *sim, double *t) { NEED(sim, "octa_sim_timing") return FWD(octa_simS_timing(sim->s, t), octa_simL_timing(sim->l, t)); }
extern "C" int octa_sim_service_stats(octa_sim *sim, double *t) { NEED(sim, "octa_sim_service_stats") return FWD(octa_simS_service_stats(sim->s, t), octa_simL_service_stats(sim->l, t)); }
extern "C" int octa_sim_fields(octa_sim *sim, int k, double *o, int64_t co, int64_t *no, double *c, int64_t cc, int64_t *nc) {
    NEED(sim, "octa_sim_fields") return FWD(octa_simS_fields(sim->s, k, o, co, no, c, cc, nc), octa_simL_fields(sim->l, k, o, co, no, c, cc, nc));
}
/* h_out4 as documented; a fifth query -- which build a configuration gets -- is octa_sim_is_large below. */
extern "C" int octa_sim_geometry(int num_cus, int *h_out4) { return octa_simS_geometry(num_cus, h_out4); }
extern "C" int octa_sim_kat_kd_order(octa_ctx *ctx, const double *p, int64_t n, const uint8_t *need, int32_t *o) { return octa_simS_kat_kd_order(ctx, p, n, need, o); }
extern "C" int octa_sim_is_large(const octa_sim *sim) { return sim && sim->large ? 1 : 0; }
