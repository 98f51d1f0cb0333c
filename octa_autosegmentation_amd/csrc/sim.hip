// sim.hip -- N1-N4: the vessel-graph simulator on gfx950. One 256-thread workgroup advances one
// sample through the block-cooperative phases of sim_core.h; B samples advance in lock-step with
// two launches per iteration (the host only serves the rare leaf-bifurcation LAPACK requests in
// between). No collective, no inter-workgroup communication: samples are independent.
//
//   launch A(it):  [venous ordered pass + CO2 removal of it-1]  O2 sampling, arterial nearest-node
//                  assignment (LDS-tiled), dict-order sort (LDS bitonic), per-node speculation
//   host:          bifurcation requests -> numpy/LAPACK callback -> results
//   launch B(it):  arterial ordered pass (RNG draws, node creation, Murray with bit-exact glibc pow),
//                  O2->CO2 conversion (cKDTree order + CPython set order emulated in LDS / by one
//                  lane), venous assignment + speculation
//   host:          bifurcation requests
// Per-sample state stays resident in HBM (about 16 MB per sample, dominated by the pre-generated
// candidate stream); LDS (155,648 B dynamic) holds tiles, sort keys and the kd key/index arrays.

#include <chrono>
#include <vector>

#include "common.h"
#include "sim_host.h"
#include <thread>
#include <atomic>
#include <cstdlib>
#include <cstring>

using namespace OCTA_SIMK;

// This file is compiled TWICE (build.py): the default build for the 3 x 3 mm^2 configurations (LDS-resident tables, two samples per
// CU) and -DOCTA_SIM_LARGE=1 for wide fields of view (sim_core.h). Each build exports its entry points under its own suffix;
// csrc/sim_api.cpp owns the public octa_sim_* names of include/octa_hip.h and picks the build per simulator (octa_sim_create).
#if OCTA_SIM_LARGE
#define OCTA_SIM_T octa_simL_impl
#define OCTA_SIM_FN(name) octa_simL_##name
#else
#define OCTA_SIM_T octa_simS_impl
#define OCTA_SIM_FN(name) octa_simS_##name
#endif
#define octa_sim_create OCTA_SIM_FN(create)
#define octa_sim_destroy OCTA_SIM_FN(destroy)
#define octa_sim_run OCTA_SIM_FN(run)
#define octa_sim_run_states OCTA_SIM_FN(run_states)
#define octa_sim_np_state OCTA_SIM_FN(np_state)
#define octa_sim_edge_offsets OCTA_SIM_FN(edge_offsets)
#define octa_sim_export_edges OCTA_SIM_FN(export_edges)
#define octa_sim_export_edges_device OCTA_SIM_FN(export_edges_device)
#define octa_sim_trace OCTA_SIM_FN(trace)
#define octa_sim_stats OCTA_SIM_FN(stats)
#define octa_sim_fields OCTA_SIM_FN(fields)
#define octa_sim_service_stats OCTA_SIM_FN(service_stats)
#define octa_sim_geometry OCTA_SIM_FN(geometry)
#define octa_sim_spans OCTA_SIM_FN(spans)
#define octa_sim_timing OCTA_SIM_FN(timing)
#define octa_sim_kat_kd_order OCTA_SIM_FN(kat_kd_order)
struct OCTA_SIM_T;
extern "C" void octa_sim_destroy(OCTA_SIM_T *S);
extern "C" long long octa_sim_next_ticket(void);
extern "C" int *octa_sim_launch_flag(int device);

namespace {

constexpr int SIM_THREADS = SIM_THREADS_PER_WG;   // 256: 4 waves, one per SIMD -- at 256 VGPRs per lane two such workgroups fill a CU's register files
constexpr int SIM_WG_PER_CU = 2;       // 2 x 80 KiB of LDS: while one sample is in a single-wave ordered pass the other's parallel phases use the CU
#if OCTA_SIM_LARGE
constexpr size_t SIM_LDS = 2048;            // collectives only: the table area of a workgroup is HBM scratch (BatchPtrs::wg_scratch)
#else
constexpr size_t SIM_LDS = SIM_LDS_BYTES;
static_assert((size_t)KD_MAILBOX_OFF + KD_MAILBOX_BYTES <= (size_t)SIM_USER_BYTES && SIM_WG_PER_CU * SIM_LDS <= 160 * 1024, "LDS budget");
#endif
static_assert((624 + 1248) * 4 <= SEQ_SIDE_LDS, "the Mersenne-Twister state and queue of the candidate stream fit the side-job LDS");
static_assert(SIM_THREADS == 64 * KD_WAVES, "kd mailboxes are sized for KD_WAVES waves");
constexpr int REQ_CAP = 8192;

static_assert(sizeof(BifRequest) == sizeof(octa_bif_request), "request layout must match the public header");
static_assert(MAXKEPT == OCTA_BIF_MAX_ATTS, "request capacity must match the public header");

struct BatchPtrs {
    double *npos[2], *nrad[2], *nkap[2];
    int *npar[2], *nch0[2], *nch1[2];
    unsigned char *nnch[2], *nact[2];
    double *oxy, *co2, *cand, *py_u;
    unsigned *py_state;   // [B][625] CPython generator states behind the stump draws (input of py_uniform_kernel)
    int *nn, *act_list;
    unsigned *sorted;
    int *gnode, *gstart, *gcount;
    Rec *rec;
    int *glist, *child_group;
    FlushRec *fl_rec;   // [B][2][MURRAY_FLUSH_LDS] deferred Murray flush records (sim_core.h)
    idx_t *kd_idx, *kd_rank;
    unsigned char *removed, *ven_near;
    unsigned long long *hashes;
    unsigned *pairs;
    unsigned long long *set_hash;
    int *set_key, *tmp_int;
    double *grid_pts;
    unsigned *cand_idx;          // [B][NCANDCAP] voxel picks of the overlapped candidate stream
    double *tmp_dbl;
    SampleScalars *sc;
    IterParams *iters;
    BifRequest *reqs;
    int *req_count;
    double *bif_results;
    unsigned *mt_state;          // [B][625] numpy stream after init (624 words + idx)
    unsigned short *valid;       // [B][valid_stride]: (i, j, k) per valid voxel; valid_stride = 0: one list for all samples (geometry file)
    size_t valid_stride;
    unsigned *valid_count;       // [B]
    int *n_per_iter;             // [n_iter]
    int *next_sample;            // [1] work queue of the persistent kernel
    unsigned char *wg_scratch;   // large build: [B][SIM_USER_BYTES] table area of the workgroup that runs sample s (Blk::umem); else null
#ifdef OCTA_SIM_DEBUG_SAT
    int *dbg;                    // diagnostic build: [B][n_iter][16] digests of phase_satisfy_art (dumped to $OCTA_SIM_DEBUG_DUMP after a run)
#endif
    int *trace;                  // [B][n_iter][4] arterial nodes, O2 sinks, venous nodes, CO2 sources at the end of every iteration (greenhouse.py:129-134)
    int n_samples;
    SimConst C;
};

__device__ __forceinline__ SimArrays sample_arrays(const BatchPtrs &B, int s) {
    SimArrays A;
    for (int f = 0; f < 2; f++) {
        A.npos[f] = B.npos[f] + (size_t)s * NCAP * 3;
        A.nrad[f] = B.nrad[f] + (size_t)s * NCAP;
        A.nkap[f] = B.nkap[f] + (size_t)s * NCAP;
        A.npar[f] = B.npar[f] + (size_t)s * NCAP;
        A.nch0[f] = B.nch0[f] + (size_t)s * NCAP;
        A.nch1[f] = B.nch1[f] + (size_t)s * NCAP;
        A.nnch[f] = B.nnch[f] + (size_t)s * NCAP;
        A.nact[f] = B.nact[f] + (size_t)s * NCAP;
    }
    A.oxy = B.oxy + (size_t)s * OCAP * 3;
    A.co2 = B.co2 + (size_t)s * CCAP * 3;
    A.cand = B.cand + (size_t)s * NCANDCAP * 3;
    A.py_u = B.py_u + (size_t)s * PYCAP;
    A.nn = B.nn + (size_t)s * OCAP;
    A.act_list = B.act_list + (size_t)s * NCAP;
    A.sorted = B.sorted + (size_t)s * SORTCAP;
    A.gnode = B.gnode + (size_t)s * GCAP;
    A.gstart = B.gstart + (size_t)s * GCAP;
    A.gcount = B.gcount + (size_t)s * GCAP;
    A.rec = B.rec + (size_t)s * GCAP;
    A.glist = B.glist + (size_t)s * GCAP;
    A.child_group = B.child_group + (size_t)s * NCAP;
    A.fl_rec = B.fl_rec + (size_t)s * 2 * MURRAY_FLUSH_LDS;
    A.kd_idx = B.kd_idx + (size_t)s * OCAP;
    A.kd_rank = B.kd_rank + (size_t)s * OCAP;
    A.removed = B.removed + (size_t)s * OCAP;
    A.ven_near = B.ven_near + (size_t)s * OCAP;
    A.hashes = B.hashes + (size_t)s * OCAP;
    A.pairs = B.pairs + (size_t)s * PCAP;
    A.set_hash = B.set_hash + (size_t)s * SETCAP;
    A.set_key = B.set_key + (size_t)s * SETCAP;
    A.tmp_int = B.tmp_int + (size_t)s * (OCAP + 2 * NCANDCAP);
    A.grid_pts = B.grid_pts + (size_t)s * GRID_N * 3;
    A.tmp_dbl = B.tmp_dbl + (size_t)s * OCAP * 3;
    A.sc = B.sc + s;
#ifdef OCTA_SIM_DEBUG_SAT
    A.dbg = B.dbg ? B.dbg + (size_t)s * B.C.n_iter * 16 : nullptr;
    A.dbg_it = 0;
#endif
    return A;
}

// ---- candidate stream (simulation_space.py:57-67): numpy's MT19937 run by ONE WAVE per sample.
// The 624-word state lives in LDS; a block of outputs is regenerated with lane-parallel sweeps (the
// recurrence only reaches 227 words back, further than one 64-lane chunk), tempered into an LDS
// output queue, and consumed lane-parallel: masked-rejection voxel picks are compacted in stream
// order with ballot/popcount, the N x 3 uniforms are formed from consecutive output pairs.
struct WaveMt {
    unsigned *st;   // [624] LDS
    unsigned *ob;   // [1248] LDS output queue
    int cur, avail;
    int lane;
    __device__ void regen_block() {
        for (int c = 0; c < 623; c += 64) {
            int k = c + lane;
            unsigned nv = 0;
            if (k < 623) {
                unsigned y = (st[k] & 0x80000000u) | (st[k + 1] & 0x7fffffffu);
                unsigned m = (k < 227) ? st[k + 397] : st[k - 227];
                nv = m ^ (y >> 1) ^ ((y & 1u) ? 0x9908b0dfu : 0u);
            }
            __builtin_amdgcn_wave_barrier();
            if (k < 623) st[k] = nv;
            __builtin_amdgcn_wave_barrier();
        }
        if (lane == 0) {
            unsigned y = (st[623] & 0x80000000u) | (st[0] & 0x7fffffffu);
            st[623] = st[396] ^ (y >> 1) ^ ((y & 1u) ? 0x9908b0dfu : 0u);
        }
        __builtin_amdgcn_wave_barrier();
    }
    __device__ static unsigned temper(unsigned y) {
        y ^= (y >> 11);
        y ^= (y << 7) & 0x9d2c5680u;
        y ^= (y << 15) & 0xefc60000u;
        y ^= (y >> 18);
        return y;
    }
    // make at least k (<= 624) outputs available at ob[cur..]
    __device__ void ensure(int k) {
        if (avail - cur >= k) return;
        const int rem = avail - cur;
        for (int c = 0; c < rem; c += 64) {
            int i = c + lane;
            unsigned v = (i < rem) ? ob[cur + i] : 0u;
            __builtin_amdgcn_wave_barrier();
            if (i < rem) ob[i] = v;
            __builtin_amdgcn_wave_barrier();
        }
        cur = 0;
        regen_block();
        for (int i = lane; i < 624; i += 64) ob[rem + i] = temper(st[i]);
        avail = rem + 624;
        __builtin_amdgcn_wave_barrier();
    }
};

// executed by wave 0 of the sample's workgroup; lds: 624 + 1248 + N words
__device__ void gen_candidates_wave(unsigned *g_state /*[625]*/, const unsigned short *valid, unsigned K, int N, double *out,
                                    unsigned *lds, unsigned *idx /* [N] scratch, LDS or HBM */, int lane, double gs) {
    WaveMt g;
    g.st = lds; g.ob = lds + 624; g.lane = lane;
    for (int i = lane; i < 624; i += 64) g.st[i] = g_state[i];
    int sidx = (int)g_state[624];
    __builtin_amdgcn_wave_barrier();
    for (int i = lane; i < 624 - sidx; i += 64) g.ob[i] = WaveMt::temper(g.st[sidx + i]);
    g.cur = 0; g.avail = 624 - sidx;
    __builtin_amdgcn_wave_barrier();
    const unsigned rng = K - 1;
    unsigned mask = rng;
    mask |= mask >> 1; mask |= mask >> 2; mask |= mask >> 4; mask |= mask >> 8; mask |= mask >> 16;
    if (rng == 0) {
        for (int i = lane; i < N; i += 64) idx[i] = 0;
    } else {
        int count = 0;
        const unsigned long long lt = (lane == 0) ? 0ull : (~0ull >> (64 - lane));
        while (count < N) {
            g.ensure(64);
            unsigned v = g.ob[g.cur + lane] & mask;
            bool acc = v <= rng;
            unsigned long long bal = __ballot(acc);
            int pre = __popcll(bal & lt);
            int pc = __popcll(bal);
            int need = N - count;
            if (pc >= need) {
                unsigned long long last = __ballot(acc && pre == need - 1);
                int consumed = __ffsll((long long)last);  // 1-based lane index = lanes consumed
                if (acc && pre < need) idx[count + pre] = v;
                g.cur += consumed;
                count = N;
            } else {
                if (acc) idx[count + pre] = v;
                g.cur += 64;
                count += pc;
            }
        }
    }
    __builtin_amdgcn_wave_barrier();
    const int total = 3 * N;
    for (int j0 = 0; j0 < total; j0 += 64) {
        int m = total - j0 < 64 ? total - j0 : 64;
        g.ensure(2 * m);
        if (lane < m) {
            unsigned a = g.ob[g.cur + 2 * lane] >> 5, bb = g.ob[g.cur + 2 * lane + 1] >> 6;
            double u = (a * 67108864.0 + bb) / 9007199254740992.0;
            int j = j0 + lane, i = j / 3, c = j - 3 * i;
            double vox = (double)valid[3 * idx[i] + c];
            out[j] = (vox + u) / gs;
        }
        g.cur += 2 * m;
    }
    __builtin_amdgcn_wave_barrier();
    for (int i = lane; i < 624; i += 64) g_state[i] = g.st[i];
    if (lane == 0) g_state[624] = (unsigned)(624 - (g.avail - g.cur));
}

// CPython's random.uniform(0, 1) stream of every sample (greenhouse.py:191,289 draw from it in node order): PYCAP doubles per sample
// from the generator state the host left behind the stump draws, one wave per sample (round 2 drew 4.2 M doubles per 128-sample batch
// on the host and copied 34 MB). random.random() = (a >> 5, b >> 6) of two consecutive outputs, as numpy's next_double.
__global__ void __launch_bounds__(64)
py_uniform_kernel(const unsigned *__restrict__ py_state /* [B][625] */, double *__restrict__ py_u /* [B][cap] */, int cap) {
    __shared__ unsigned lds[624 + 1248];
    const int s = blockIdx.x, lane = threadIdx.x;
    const unsigned *g_state = py_state + (size_t)s * 625;
    double *out = py_u + (size_t)s * cap;
    WaveMt g;
    g.st = lds; g.ob = lds + 624; g.lane = lane;
    for (int i = lane; i < 624; i += 64) g.st[i] = g_state[i];
    const int sidx = (int)g_state[624];
    __builtin_amdgcn_wave_barrier();
    for (int i = lane; i < 624 - sidx; i += 64) g.ob[i] = WaveMt::temper(g.st[sidx + i]);
    g.cur = 0; g.avail = 624 - sidx;
    __builtin_amdgcn_wave_barrier();
    for (int j0 = 0; j0 < cap; j0 += 64) {
        const int m = cap - j0 < 64 ? cap - j0 : 64;
        g.ensure(2 * m);
        if (lane < m) {
            const unsigned a = g.ob[g.cur + 2 * lane] >> 5, bb = g.ob[g.cur + 2 * lane + 1] >> 6;
            out[j0 + lane] = (a * 67108864.0 + bb) / 9007199254740992.0;
        }
        g.cur += 2 * m;
    }
}

// ---- iteration kernels
#define OCTA_PROF(slot, stmt)                                              \
    do {                                                                   \
        long _t0 = (long)wall_clock64();                                   \
        stmt;                                                              \
        if (threadIdx.x == 0) A.sc->prof[slot] += (long)wall_clock64() - _t0; \
    } while (0)
__global__ void __launch_bounds__(SIM_THREADS)
sim_iter_a_kernel(BatchPtrs B, int it, int finish_prev) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const int s = blockIdx.x;
    SimArrays A = sample_arrays(B, s);
    Blk b = {(int)threadIdx.x, (int)blockDim.x, smem};
    if (B.wg_scratch) b.umem = B.wg_scratch + (size_t)s * SIM_USER_BYTES;
    if (A.sc->err) return;
    if (finish_prev) {
        const IterParams Pp = B.iters[it - 1];
        OCTA_PROF(8, phase_seq(b, A, B.C, Pp, 1, A.co2, B.bif_results));
        OCTA_PROF(9, phase_satisfy_ven(b, A, Pp));
        if (threadIdx.x == 0) {
            int *tr = B.trace + ((size_t)s * B.C.n_iter + (it - 1)) * 4;
            tr[0] = A.sc->n_nodes[0]; tr[1] = A.sc->n_oxy; tr[2] = A.sc->n_nodes[1]; tr[3] = A.sc->n_co2;
        }
    }
    if (it >= B.C.n_iter) { murray_flush_pending(b, A); return; }      // the run is over: the radii the last passes left for "the other forest's next pass"
    const IterParams P = B.iters[it];
    {
        long _t0 = (long)wall_clock64();
        if (threadIdx.x < 64)
            gen_candidates_wave(B.mt_state + (size_t)s * 625, B.valid + (size_t)s * B.valid_stride, B.valid_count[s], P.N,
                                B.cand + (size_t)s * NCANDCAP * 3, reinterpret_cast<unsigned *>(b.user()),
                                reinterpret_cast<unsigned *>(b.user()) + 624 + 1248, (int)threadIdx.x, B.C.gs);
        octa_block_sync();
        if (threadIdx.x == 0) A.sc->prof[10] += (long)wall_clock64() - _t0;
    }
    OCTA_PROF(0, phase_sample(b, A, B.C, P, it));
    OCTA_PROF(1, phase_assign(b, A, 0, A.oxy, A.sc->n_oxy, P.delta_art));
    OCTA_PROF(2, phase_pre(b, A, B.C, P, 0, A.oxy, B.reqs, B.req_count, REQ_CAP, s));
}

__global__ void __launch_bounds__(SIM_THREADS)
sim_iter_b_kernel(BatchPtrs B, int it) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const int s = blockIdx.x;
    SimArrays A = sample_arrays(B, s);
    Blk b = {(int)threadIdx.x, (int)blockDim.x, smem};
    if (B.wg_scratch) b.umem = B.wg_scratch + (size_t)s * SIM_USER_BYTES;
    if (A.sc->err) return;
    const IterParams P = B.iters[it];
    OCTA_PROF(3, phase_seq(b, A, B.C, P, 0, A.oxy, B.bif_results));
    OCTA_PROF(4, phase_satisfy_art(b, A, B.C, P));
    OCTA_PROF(6, phase_assign(b, A, 1, A.co2, A.sc->n_co2, P.delta_ven));
    OCTA_PROF(7, phase_pre(b, A, B.C, P, 1, A.co2, B.reqs + REQ_CAP, B.req_count + 1, REQ_CAP, s));
}

#ifdef OCTA_SIM_CODE_SIZE_PROBE
// diagnostic (never launched): one kernel per phase, so that `hipcc -save-temps -DOCTA_SIM_CODE_SIZE_PROBE` lists every phase's code size
// (`; codeLenInByte` in the .s; tools/sim_code_size.sh) -- the persistent kernel inlines all of them, and its hot loops compete for an instruction
// cache of 64 KB per two CUs
#define OCTA_PROBE_KERNEL(name, call) \
    __global__ void __launch_bounds__(SIM_THREADS) name(BatchPtrs B, int it) { \
        extern __shared__ __attribute__((aligned(16))) unsigned char smem[]; \
        const int s = blockIdx.x; \
        SimArrays A = sample_arrays(B, s); \
        Blk b = {(int)threadIdx.x, (int)blockDim.x, smem}; \
        const IterParams P = B.iters[it]; \
        call; \
    }
OCTA_PROBE_KERNEL(probe_phase_sample, phase_sample(b, A, B.C, P, it))
OCTA_PROBE_KERNEL(probe_phase_assign_art, phase_assign(b, A, 0, A.oxy, A.sc->n_oxy, P.delta_art))
OCTA_PROBE_KERNEL(probe_phase_pre_art, phase_pre(b, A, B.C, P, 0, A.oxy, B.reqs, B.req_count, REQ_CAP, s))
OCTA_PROBE_KERNEL(probe_phase_seq_art, phase_seq(b, A, B.C, P, 0, A.oxy, B.bif_results))
OCTA_PROBE_KERNEL(probe_phase_seq_ven, phase_seq(b, A, B.C, P, 1, A.co2, B.bif_results))
OCTA_PROBE_KERNEL(probe_phase_satisfy_art, phase_satisfy_art(b, A, B.C, P))
OCTA_PROBE_KERNEL(probe_phase_satisfy_ven, phase_satisfy_ven(b, A, P))
OCTA_PROBE_KERNEL(probe_kd_build, kd_build(b, A.oxy, A.sc->n_oxy, A.kd_idx, A.kd_rank, reinterpret_cast<float *>(A.hashes), 0.0, B.C.sz, nullptr, A.removed, true))
#undef OCTA_PROBE_KERNEL
#endif

// ---- persistent form: one launch runs all iterations of every sample; a workgroup only waits for the host
// when ITS sample posted leaf-bifurcation requests (about one pass in ten), through a mailbox in pinned host
// memory. Samples are no longer in lock step, so a launch lasts as long as its slowest sample's SUM of phases
// instead of the sum of per-iteration maxima, and there are no per-iteration launches or stream syncs.
struct HostMail {
    BifRequest *reqs;    // pinned host [B][REQ_PER_SAMPLE]     device -> host
    double *results;     // pinned host [B][REQ_PER_SAMPLE][6]  host -> device
    int *req_n;          // pinned host [B]                     device -> host
    int *req_ticket;     // pinned host [B]                     device -> host (2*it+1: arterial pass, 2*it+2: venous)
    int *resp_ticket;    // pinned host [B]                     host -> device
    int *done;           // pinned host [1]                     device -> host: workgroups that have left the kernel
    long timeout_ticks;  // device-side bound on one wait for the host (100 MHz wall clock)
    long park_ticks;     // a workgroup that has waited this long parks (0: never, wait until timeout_ticks as round 1 did)
    int *launch_flag;    // device [8]: the launches' sign-in block (csrc/sim_api.cpp octa_sim_launch_flag), or NULL
    int ticket;          // this launch's ticket
    long *req_time;      // pinned host [B]: device clock at publication (-DOCTA_SIM_PROF_MAIL builds write it, OCTA_SIM_MAIL_DIAG=1 reads it)
};
constexpr int REQ_PER_SAMPLE = OCTA_SIM_LARGE ? 256 : 32;    // bifurcation requests of one sample per mailbox round trip (6.2 KB each, pinned host memory)
constexpr int ERR_HOST_TIMEOUT = 2048;

// One request / answer round trip with the host service thread (octa_sim_run). Returns 0 when the answer is in
// M.results, 1 when the workgroup must PARK (block-uniform).
//
// Why parking exists. Round 1 waited here for as long as it took (30 s bound, error bit 0x800) and the driver's run hit that
// bound. Measured in round 2 (tools/repro_mailbox_deadlock.py, profiles/r02_mailbox_repro.log): in about one launch in two
// hundred -- provoked by runtime activity of other host threads: device-wide waits, pageable copies -- the host, although it
// scans the mailbox all the time (longest pass < 0.1 ms), stops seeing ANY ticket of that launch from some moment on and sees
// them all when the kernel ends; meanwhile the workgroup reads its own ticket back correctly through a system-scope atomic,
// and neither publishing with an atomic exchange + system fences nor flushing the host's cache lines changes that. In other
// words: on this platform a running kernel's writes to pinned host memory are not guaranteed to become visible to the host
// before the kernel ends -- only kernel boundaries are. So liveness must not depend on it: a workgroup waits a few
// milliseconds (normal answers take tens of microseconds), then records its resume point in HBM and LEAVES the kernel; at
// the kernel boundary the host sees every parked request, serves it and launches the kernel again for the parked samples.
// A launch without an episode is unaffected; an episode costs one extra launch instead of the run.
__device__ inline int mail_roundtrip(const Blk &b, const SimArrays &A, const HostMail &M, int s, int n_req, int ticket, long t_kernel) {
    if (n_req == 0) return 0;  // block-uniform
    int *park = b.coll() + 98;
    if (b.tid == 0) {
        *park = 0;
        if (n_req > REQ_PER_SAMPLE) { atomicOr(&A.sc->err, ERR_REQ_CAP); n_req = REQ_PER_SAMPLE; }
#ifdef OCTA_SIM_PROF_MAIL            // diagnostic build: kd slots 0..2 = ticks spent publishing (fences + stores), polls, round trips
        const long t_pub = (long)wall_clock64();
        __hip_atomic_store(M.req_time + s, t_pub, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
#endif
        __threadfence_system();      // the request records (written by the whole block before the barrier)
        __hip_atomic_store(M.req_n + s, n_req, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
        __hip_atomic_exchange(M.req_ticket + s, ticket, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
        __threadfence_system();
        const long t0 = (long)wall_clock64();
        long polls = 0;
        while (__hip_atomic_load(M.resp_ticket + s, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_SYSTEM) != ticket) {
            __builtin_amdgcn_s_sleep(64);
            polls++;
            const long waited = (long)wall_clock64() - t0;
            if (OCTA_UNLIKELY(M.park_ticks > 0 && waited > M.park_ticks)) { *park = 1; break; }
            if (OCTA_UNLIKELY(waited > M.timeout_ticks)) {      // only without parking (OCTA_SIM_PARK_MS=0): the round-1 behaviour
                atomicOr(&A.sc->err, ERR_HOST_TIMEOUT);
                A.sc->prof[11] = __hip_atomic_load(M.resp_ticket + s, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_SYSTEM);
                A.sc->prof[12] = t0 - t_kernel;
                A.sc->prof[13] = ticket;
                A.sc->prof[14] = polls;
                A.sc->prof[15] = __hip_atomic_fetch_or(M.req_ticket + s, 0, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_SYSTEM);   // its own request, read back
                break;
            }
        }
        A.sc->prof[5] += (long)wall_clock64() - t0;
#ifdef OCTA_SIM_PROF_MAIL
        A.sc->kdprof[0] += t0 - t_pub; A.sc->kdprof[1] += polls; A.sc->kdprof[2] += 1;
#endif
        __threadfence_system();
    }
    b.sync();
    const int p = *park;
    b.sync();
    return p;
}

// block-uniform read of the sample's error bits
__device__ inline int uniform_err(const Blk &b, const SimArrays &A) {
    int *slot = b.coll() + 97;
    b.sync();
    if (b.tid == 0) *slot = atomicOr(&A.sc->err, 0);
    b.sync();
    int e = *slot;
    b.sync();
    return e;
}

// every iteration of ONE sample (or what is left of them after a park), by one workgroup
__device__ __forceinline__ void run_sample(const BatchPtrs &B, const HostMail &M, const int s_in, const Blk &b) {     // inlined: B stays kernel arguments (scalar registers)
    // The sample index is wave-uniform by construction; declaring it so keeps the sample's ~40 array base pointers in scalar registers
    // (per-sample device time 566 -> 509 ms). Round 3 could not ship this: 512-sample batches stopped being reproducible (about one
    // sample run in 2000 converted a few O2 sinks too few into CO2 sources). Round 4 found the cause -- not the addressing, but a barrier
    // in the pair sort's stage loop that the compiler left without its LDS wait (sim_core.h: octa_block_sync). The scalar
    // addressing only shortened the path between a stage's last LDS store and that barrier.
    const int s = __builtin_amdgcn_readfirstlane(s_in);
    SimArrays A = sample_arrays(B, s);
    int *req_n = b.coll() + 96;
    BifRequest *reqs = M.reqs + (size_t)s * REQ_PER_SAMPLE;
    const double *results = M.results + (size_t)s * REQ_PER_SAMPLE * 6;
    const int n_iter = B.C.n_iter;
    const long t_kernel = (long)wall_clock64();
    // where this sample stands: a fresh sample starts at (0, 0); a parked one resumes behind the mailbox it parked at (its
    // answer is in M.results, written by the host before this launch); finished or failed samples only sign off
    int *uni = b.coll() + 99;
    if (b.tid == 0) {
        uni[0] = A.sc->resume_it; uni[1] = A.sc->resume_stage; uni[2] = A.sc->finished | (A.sc->err != 0); A.sc->parked = 0;
        if (A.sc->t_begin == 0) A.sc->t_begin = t_kernel;
    }
    b.sync();
    const int it0 = OCTA_UNI(uni[0]);
    const int stage = OCTA_UNI(uni[1]);
    const int skip = OCTA_UNI(uni[2]);
    b.sync();
    int parked_at = -1, parked_stage = 0;
#if defined(OCTA_SIM_DEBUG_SAT) && defined(OCTA_SIM_ITER_PROF)
    long iter_prof_prev[16];
    for (int k = 0; k < 16; k++) iter_prof_prev[k] = A.sc->prof[k];
#endif
    // Half-steps (round 6): h = 2 * it + (1 behind the arterial mailbox of iteration it). A half-step is
    //   A: the ordered pass + satisfaction step of the forest whose bifurcation requests the previous mailbox answered -- arterial when h is odd,
    //      venous (of iteration it - 1) when h is even --, then
    //   B: [even: the iteration's O2 sampling,] assignment + speculation of the OTHER forest and its mailbox.
    // Written as ONE loop body with the forest a run-time value, the three per-forest phases are inlined ONCE instead of once per forest: the kernel is
    // 150 KB of code smaller (522 KB before; its hot loops compete for an instruction cache of 64 KB per two CUs) and the two workgroups
    // of a CU run the same instructions whichever forest they are at. Selecting a forest's arrays at run time costs nothing measurable.
    for (int h = 2 * it0 + stage; h <= 2 * n_iter && !skip; h++) {
        const int it = h >> 1;
        const int odd = __builtin_amdgcn_readfirstlane(h & 1);
        if (uniform_err(b, A)) break;
        // ---- A
        const int it_a = odd ? it : it - 1;
        if (it_a >= 0) {
            const IterParams Pa = B.iters[it_a];
            const int fa = odd ? 0 : 1;
            // the candidate stream of the NEXT iteration (numpy MT19937, one wave) runs beside the ordered ARTERIAL pass: it only depends on the
            // generator state, and the candidate buffer is free once phase_sample has consumed it
            const int n_next = (odd && it + 1 < n_iter) ? B.iters[it + 1].N : 0;
            auto next_candidates = [&](unsigned char *lds) {
                if (n_next <= 0) return;
                const long _t0 = (long)wall_clock64();
                gen_candidates_wave(B.mt_state + (size_t)s * 625, B.valid + (size_t)s * B.valid_stride, B.valid_count[s], n_next,
                                    B.cand + (size_t)s * NCANDCAP * 3, reinterpret_cast<unsigned *>(lds),
                                    B.cand_idx + (size_t)s * NCANDCAP, (int)(threadIdx.x & 63), B.C.gs);
                if ((threadIdx.x & 63) == 0) A.sc->prof[10] += (long)wall_clock64() - _t0;
            };
            OCTA_PROF(odd ? 3 : 8, phase_seq(b, A, B.C, Pa, fa, fa ? A.co2 : A.oxy, results, next_candidates));
            if (odd) {
#ifdef OCTA_SIM_DEBUG_SAT
                A.dbg_it = it;
#endif
                OCTA_PROF(4, phase_satisfy_art(b, A, B.C, Pa));
            } else {
                OCTA_PROF(9, phase_satisfy_ven(b, A, Pa));
                if (b.tid == 0) {      // iteration it - 1 is complete: the reference's per-step statistics
                    int *tr = B.trace + ((size_t)s * n_iter + (it - 1)) * 4;
                    tr[0] = A.sc->n_nodes[0]; tr[1] = A.sc->n_oxy; tr[2] = A.sc->n_nodes[1]; tr[3] = A.sc->n_co2;
                }
            }
        }
        if (!odd && it >= n_iter) break;
        // ---- B
        const IterParams P = B.iters[it];
        if (!odd) {
            if (it == 0) {   // later iterations get their candidates from the side job of the previous ordered arterial pass
                long _t0 = (long)wall_clock64();
                if (threadIdx.x < 64)
                    gen_candidates_wave(B.mt_state + (size_t)s * 625, B.valid + (size_t)s * B.valid_stride, B.valid_count[s], P.N,
                                        B.cand + (size_t)s * NCANDCAP * 3, reinterpret_cast<unsigned *>(b.user()),
                                        reinterpret_cast<unsigned *>(b.user()) + 624 + 1248, (int)threadIdx.x, B.C.gs);
                octa_block_sync();
                if (threadIdx.x == 0) A.sc->prof[10] += (long)wall_clock64() - _t0;
            }
            OCTA_PROF(0, phase_sample(b, A, B.C, P, it));
        }
        {
            const int g = odd;                                   // the forest that is assigned and speculated now
            const double *att = g ? A.co2 : A.oxy;
            const int n_att = g ? A.sc->n_co2 : A.sc->n_oxy;
            const double delta = g ? P.delta_ven : P.delta_art;
            OCTA_PROF(g ? 6 : 1, phase_assign(b, A, g, att, n_att, delta));
#if OCTA_SIM_DUP & 4
            phase_assign(b, A, g, att, n_att, delta);
#endif
            if (b.tid == 0) *req_n = 0;
            b.sync();
            OCTA_PROF(g ? 7 : 2, phase_pre(b, A, B.C, P, g, att, reqs, req_n, REQ_PER_SAMPLE, s));
#if OCTA_SIM_DUP & 32
            b.sync();
            if (b.tid == 0) *req_n = 0;
            b.sync();
            phase_pre(b, A, B.C, P, g, att, reqs, req_n, REQ_PER_SAMPLE, s);
#endif
            if (mail_roundtrip(b, A, M, s, *req_n, 2 * it + 1 + odd, t_kernel)) { parked_at = it; parked_stage = 1 + odd; break; }
        }
#if defined(OCTA_SIM_DEBUG_SAT) && defined(OCTA_SIM_ITER_PROF)
        // diagnostic build (tools/sim_iter_profile.py): the row of iteration `it` holds the 16 phase timers' growth during it
        if (odd && b.tid == 0 && A.dbg) {
            int *row = A.dbg + 16 * it;
            for (int k = 0; k < 16; k++) { row[k] = (int)(A.sc->prof[k] - iter_prof_prev[k]); iter_prof_prev[k] = A.sc->prof[k]; }
        }
#endif
    }
    if (parked_at < 0 && !skip) murray_flush_pending(b, A);      // the run is over (block-uniform): radii left for "the other forest's next pass"
    // sign-off: the host leaves its service loop when every SAMPLE has passed here (or when the launch has completed)
    b.sync();
    if (b.tid == 0) {
        if (parked_at >= 0) { A.sc->resume_it = parked_stage == 2 ? parked_at + 1 : parked_at; A.sc->resume_stage = parked_stage == 2 ? 0 : 1; A.sc->parked = 1; }
        else if (!skip) A.sc->finished = 1;
        if (!skip) A.sc->t_end = (long)wall_clock64();
        __threadfence_system();
        __hip_atomic_fetch_add(M.done, 1, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
    }
    b.sync();
}

// Work-queue form: the launch has at most SIM_WG_PER_CU workgroups per CU (80 KiB of LDS each); a workgroup takes the next sample of the
// batch from a device counter, runs all of its iterations, and comes back for another. Samples differ by +-10 % in run time
// and workgroups that park leave early: with one workgroup PER SAMPLE the CUs waited for the hardware dispatcher to place the
// next launch's workgroups (measured: 76 % of the CU time used with 2-12 launches in flight, whatever their size); resident
// workgroups that refill themselves only leave the CU idle at the very end of a launch.
__global__ void __launch_bounds__(SIM_THREADS, SIM_WG_PER_CU * SIM_THREADS / 256) __attribute__((amdgpu_waves_per_eu(SIM_WG_PER_CU * SIM_THREADS / 256, SIM_WG_PER_CU * SIM_THREADS / 256)))      // waves per SIMD: 2 at 256 threads (<= 256 registers per lane, accumulation registers included; with the explicit attribute the compiler WARNS when it cannot keep to that -- an experiment of round 6 took 32 accumulation registers on top of 256 and the kernel silently ran one workgroup per CU)
sim_persistent_kernel(BatchPtrs B, HostMail M) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    Blk b = {(int)threadIdx.x, SIM_THREADS, smem};      // the launch shape is fixed (sim_run_impl): as a constant it removes the one-thread (host build) paths and turns every stride into an immediate
    // sign in: a rasterisation ordered behind this launch waits (on its own stream, csrc/order.hip) until the launch's workgroups are resident
    if (threadIdx.x == 0 && M.launch_flag) {
        if (blockIdx.x == 0) {
            __hip_atomic_store(M.launch_flag + 3 + (M.ticket & 1), (int)gridDim.x, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            __hip_atomic_fetch_max(M.launch_flag, M.ticket, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
        }
        __hip_atomic_fetch_add(M.launch_flag + 1 + (M.ticket & 1), 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
    int *next = b.coll() + 104;
    while (true) {
        if (b.tid == 0) *next = atomicAdd(B.next_sample, 1);
        b.sync();
        const int s = *next;
        b.sync();
        if (s >= B.n_samples) break;
        Blk bs = b;
        if (B.wg_scratch) bs.umem = B.wg_scratch + (size_t)s * SIM_USER_BYTES;     // large build: this sample's table area in HBM
        run_sample(B, M, s, bs);
        // blk_scan alternates two LDS slots by a call counter that lives in the Blk: the per-sample COPY must hand its count back, or the
        // first scan of the next sample could write the slot the last scan of this one is still being read from (today two barriers at
        // the loop head separate them; this makes it hold by construction)
        b.scan_calls = bs.scan_calls;
    }
}

// Edge list of every sample ON THE DEVICE (round 3), in the reference's CSV row order (generate_vessel_graph.py:43-66: per forest,
// per tree, anytree's level order with the root excluded; rows = node xyz, parent xyz, radius). One workgroup per (sample, forest)
// walks its trees level by level: the next level = the children of the current one in parent order (child 0 before child 1), placed
// by an exclusive scan of the child counts; a level's rows are written in parallel. The frontiers (u16 ids) live in LDS. The host
// BFS of sim_host.h: export_edges produces the same rows (octa_sim_export_edges keeps it as the cross-check and for callers that
// want the list on the host); here the list never leaves HBM on its way to the rasteriser.
__global__ void __launch_bounds__(256)
sim_export_kernel(BatchPtrs B, const long *__restrict__ edge_off, int n_trees, double *__restrict__ edges) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const int s = blockIdx.x, f = blockIdx.y;
    SimArrays A = sample_arrays(B, s);
    Blk b = {(int)threadIdx.x, (int)blockDim.x, smem};
    if (B.wg_scratch) b.umem = B.wg_scratch + (size_t)s * SIM_USER_BYTES + (size_t)f * 2 * NCAP * sizeof(idx_t);   // the two forests' frontiers side by side
    idx_t *cur = reinterpret_cast<idx_t *>(b.user());
    idx_t *nxt = cur + NCAP;
    const int n_art_rows = A.sc->n_nodes[0] - n_trees;
    long row = edge_off[s] + (f ? n_art_rows : 0);
    const double *pos = A.npos[f], *rad = A.nrad[f];
    const int *par = A.npar[f], *c0 = A.nch0[f], *c1 = A.nch1[f];
    const unsigned char *nch = A.nnch[f];
    for (int t = 0; t < n_trees; t++) {
        b.sync();
        if (b.tid == 0) cur[0] = (idx_t)(2 * t);
        b.sync();
        int n_cur = 1;
        while (n_cur > 0) {
            int base = 0;
            for (int i0 = 0; i0 < n_cur; i0 += b.nth) {
                const int i = i0 + b.tid;
                int id = -1, nc = 0;
                if (i < n_cur) { id = cur[i]; nc = nch[id]; if (nc > 2) nc = 2; }
                int ex;
                const int tot = blk_scan(b, nc, &ex);
                if (nc >= 1 && base + ex < NCAP) nxt[base + ex] = (idx_t)c0[id];
                if (nc >= 2 && base + ex + 1 < NCAP) nxt[base + ex + 1] = (idx_t)c1[id];
                base += tot;
            }
            b.sync();
            const int n_next = base < NCAP ? base : NCAP;
            for (int j = b.tid; j < n_next; j += b.nth) {
                const int v = nxt[j], p = par[v];
                double *e = edges + 7 * (row + j);
                e[0] = pos[3 * v]; e[1] = pos[3 * v + 1]; e[2] = pos[3 * v + 2];
                e[3] = pos[3 * p]; e[4] = pos[3 * p + 1]; e[5] = pos[3 * p + 2];
                e[6] = rad[v];
            }
            row += n_next;
            idx_t *tmp = cur; cur = nxt; nxt = tmp;
            n_cur = n_next;
            b.sync();
        }
    }
}

}  // namespace

struct OCTA_SIM_T {
    octa_ctx *ctx = nullptr;
    int B = 0;
    SimConfig cfg;
    SimConst C;
    std::vector<IterParams> iters;
    BatchPtrs P;
    std::vector<void *> allocs;
    std::vector<unsigned short> fixed_valid;   // geometry file: the valid voxels (i, j, k), shared by all samples
    unsigned char *d_mask = nullptr;           // geometry file: the mask in HBM (SimConst::mask)
    long *d_edge_off = nullptr;     // [B + 1] row offsets for the device-side edge export
    BifRequest *h_reqs = nullptr;   // pinned [2*REQ_CAP]
    double *h_results = nullptr;    // pinned [2*REQ_CAP*6]
    int *h_req_count = nullptr;     // pinned [2]
    HostMail mail = {nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, 0, 0, nullptr, 0, nullptr};  // pinned mailbox of the persistent form
    bool lockstep = false;          // OCTA_SIM_LOCKSTEP=1: two launches per iteration (the round-1 form)
    double mail_timeout_ms = 30000; // OCTA_SIM_MAIL_TIMEOUT_MS: device-side bound on one wait for the host when parking is off
    double park_ms = 100.0;         // OCTA_SIM_PARK_MS: a workgroup that has waited this long for its answer parks (0: never). Normal answers
                                    // take tens of microseconds; 3 ms (until the end of round 2) also parked workgroups whenever a busy host
                                    // descheduled the service thread for a few milliseconds, and a park costs a drain + relaunch. Round 6:
                                    // 20 -> 100 ms. A parked sample is re-run ALONE at the launch's end (+400 ms for a 512-sample launch);
                                    // one headline run in ~16 lost a launch that way to a host stall of a few tens of milliseconds, which
                                    // costs its own length when it is waited out. The episodes parking exists for (a launch whose tickets
                                    // the host stops seeing) take 80 ms longer to resolve.
    int last_ticket = 0;            // ticket of the last persistent-kernel launch of this simulator
    int grid_cap = 512;             // workgroups per launch of the persistent kernel (SIM_WG_PER_CU per CU; OCTA_SIM_GRID overrides)
    int test_stall_ms = 0;          // OCTA_SIM_TEST_HOST_STALL_MS (test hook): the service thread sleeps once with a ticket pending
    long spin_scans = 4096;         // idle mailbox scans before the service thread starts sleeping 20 us between scans
    double diag_max_gap_ms = 0, diag_max_bif_ms = 0;
    long diag_tickets = 0, diag_relaunches = 0, diag_parked = 0, diag_passes = 0;
    bool ran = false;
    void *init_stage = nullptr;     // pinned staging of a run's per-sample set-up (valid-voxel lists, generator states)
    size_t init_stage_bytes = 0;
    // host copies for export
    std::vector<SampleScalars> h_sc;
    // kernel timing of the last run (HIP events on the launch stream)
    hipEvent_t ev[3] = {nullptr, nullptr, nullptr};
    double ms_a = 0, ms_b = 0, ms_total = 0, ms_host_bif = 0;
    long n_a = 0, n_b = 0, n_bif_req = 0;
    size_t bytes = 0;
    // pinned staging + stream of octa_sim_export_edges
    void *export_stage = nullptr;
    size_t export_cap = 0;
    hipStream_t export_stream = nullptr;
};

namespace {

template <class T>
int dev_alloc(OCTA_SIM_T *S, T **p, size_t count) {
    void *q = nullptr;
    size_t bytes = count * sizeof(T);
    // experiment knob (DESIGN.md 4.1 "Open"): OCTA_SIM_ALLOC=finegrained | uncached places the simulator's HBM state in fine-grained /
    // uncached device memory (other MTYPE: other caching rules of the vector L1 and the L2)
    static const int alloc_mode = [] { const char *e = getenv("OCTA_SIM_ALLOC"); return !e ? 0 : (!strcmp(e, "finegrained") ? 1 : (!strcmp(e, "uncached") ? 2 : 0)); }();
    hipError_t e = alloc_mode == 1 ? hipExtMallocWithFlags(&q, bytes ? bytes : 16, hipDeviceMallocFinegrained)
                 : alloc_mode == 2 ? hipExtMallocWithFlags(&q, bytes ? bytes : 16, hipDeviceMallocUncached)
                                   : hipMalloc(&q, bytes ? bytes : 16);
    if (e != hipSuccess) { octa::set_error("octa_sim: hipMalloc(%zu) failed: %s", bytes, hipGetErrorString(e)); return -1; }
    S->allocs.push_back(q);
    S->bytes += bytes;
    *p = reinterpret_cast<T *>(q);
    return 0;
}

}  // namespace

extern "C" int octa_sim_create(octa_ctx *ctx, const octa_sim_config *c, int B, OCTA_SIM_T **out) {
    if (!ctx || !c || !out || B <= 0) { octa::set_error("octa_sim_create: bad arguments"); return -2; }
    *out = nullptr;
    if (c->n_modes < 1 || c->n_modes > 8) { octa::set_error("octa_sim_create: n_modes must be 1..8"); return -2; }
    OCTA_HIP_CHECK(hipSetDevice(ctx->device));
    OCTA_SIM_T *S = new (std::nothrow) OCTA_SIM_T();
    if (!S) { octa::set_error("octa_sim_create: out of host memory"); return -1; }
    S->ctx = ctx;
    S->B = B;
    SimConfig &cfg = S->cfg;
    cfg.param_scale = c->param_scale; cfg.d = c->d; cfg.r = c->r; cfg.faz_mean = c->faz_radius_mean; cfg.faz_std = c->faz_radius_std;
    cfg.rotation_radius = c->rotation_radius; cfg.fc0 = c->faz_center[0]; cfg.fc1 = c->faz_center[1];
    cfg.sx = c->size[0]; cfg.sy = c->size[1]; cfg.sz = c->size[2]; cfg.n_trees = c->n_trees;
    int nw = 0;
    for (int w = 0; w < 4; w++) { cfg.walls[w] = c->walls[w]; nw += c->walls[w] ? 1 : 0; }
    if (c->n_source_walls < 0 || c->n_source_walls > 6) { octa::set_error("octa_sim_create: n_source_walls must be 0..6"); delete S; return -2; }
    if (c->n_source_walls > 0) {
        nw = cfg.n_wall_list = c->n_source_walls;
        for (int w = 0; w < nw; w++) {
            cfg.wall_list[w] = c->source_walls[w];
            if (cfg.wall_list[w] < 0 || cfg.wall_list[w] > 5) { octa::set_error("octa_sim_create: source_walls[%d] = %d is not a wall (0..5)", w, cfg.wall_list[w]); delete S; return -2; }
            if (cfg.wall_list[w] >= 4 && !c->geometry && c->forest_type == 0) {   // simulation_space.py:82-87 reads `self.valid_pixels`, which nothing sets
                octa::set_error("octa_sim_create: the z source walls need a sampling geometry file (the reference fails without one)");
                delete S; return -2;
            }
        }
    }
    cfg.forest_type = c->forest_type; cfg.nc0 = c->nerve_center[0]; cfg.nc1 = c->nerve_center[1]; cfg.nr = c->nerve_radius;
    if (c->geometry) {   // simulation_space.py:29-34
        const int *g = c->geometry_shape;
        if (g[0] < 1 || g[1] < 1 || g[2] < 1 || g[0] > 65535 || g[1] > 65535 || g[2] > 65535 || (size_t)g[0] * g[1] * g[2] > ((size_t)1 << 26)) {
            octa::set_error("octa_sim_create: sampling geometry [%d][%d][%d]: every dimension must be 1..65535 and the mask at most 2^26 voxels", g[0], g[1], g[2]);
            delete S; return -2;
        }
        for (int k = 0; k < 3; k++) cfg.gshape[k] = g[k];
        cfg.geometry.assign(c->geometry, c->geometry + (size_t)g[0] * g[1] * g[2]);
        const double gs = (double)cfg.gs();
        cfg.sx = g[0] / gs; cfg.sy = g[1] / gs; cfg.sz = g[2] / gs;       // shape = geometry.shape / max(geometry.shape)
        if (cfg.forest_type != 0) { octa::set_error("octa_sim_create: a sampling geometry with nerve forests is not supported"); delete S; return -2; }
        // a wall whose face 0 has no valid voxel: the reference's random.choice raises IndexError on the empty list
        SampleInit probe;
        std::vector<int> face[3];
        init_mask(cfg, 0.0, &probe, face);
        if (probe.valid.empty()) { octa::set_error("octa_sim_create: the sampling geometry has no valid voxel"); delete S; return -2; }
        S->fixed_valid = probe.valid;
        for (int w = 0; w < 6; w++) {
            bool used = false;
            if (cfg.n_wall_list > 0) { for (int k = 0; k < cfg.n_wall_list; k++) used |= cfg.wall_list[k] == w; }
            else used = w < 4 && cfg.walls[w];
            if (used && face[w >> 1].empty()) { octa::set_error("octa_sim_create: source wall %d: face 0 of the sampling geometry along axis %d has no valid voxel", w, w >> 1); delete S; return -2; }
        }
    }
    if (cfg.forest_type != 0 && cfg.forest_type != 1) { octa::set_error("octa_sim_create: forest_type must be 0 (stumps) or 1 (nerve)"); delete S; return -2; }
    for (int m = 0; m < c->n_modes; m++) {
        const double *q = c->modes[m];
        cfg.modes.push_back(ModeCfg{(int)q[0], (int)q[1], q[2], q[3], q[4], q[5], q[6], q[7], q[8], q[9], q[10], q[11], q[12]});
        if ((int)q[1] > NCANDCAP || (int)q[1] < 0) { octa::set_error("octa_sim_create: N=%d exceeds %d", (int)q[1], NCANDCAP); delete S; return -2; }
    }
    if ((nw == 0 && cfg.forest_type == 0) || cfg.n_trees < 1 || 2 * cfg.n_trees > NCAP) { octa::set_error("octa_sim_create: bad forest config"); delete S; return -2; }
    if (std::ceil(cfg.sx * 76) > 76 || std::ceil(cfg.sy * 76) > 76) { octa::set_error("octa_sim_create: simulation space larger than the unit square"); delete S; return -2; }
    S->iters = build_iter_table(cfg, &S->C);
    BatchPtrs &P = S->P;
    memset(&P, 0, sizeof(P));
    P.C = S->C;
    const size_t nb = (size_t)B;
    int rc = 0;
    for (int f = 0; f < 2 && !rc; f++) {
        rc |= dev_alloc(S, &P.npos[f], nb * NCAP * 3); rc |= dev_alloc(S, &P.nrad[f], nb * NCAP); rc |= dev_alloc(S, &P.nkap[f], nb * NCAP);
        rc |= dev_alloc(S, &P.npar[f], nb * NCAP); rc |= dev_alloc(S, &P.nch0[f], nb * NCAP); rc |= dev_alloc(S, &P.nch1[f], nb * NCAP);
        rc |= dev_alloc(S, &P.nnch[f], nb * NCAP); rc |= dev_alloc(S, &P.nact[f], nb * NCAP);
    }
    rc |= dev_alloc(S, &P.oxy, nb * OCAP * 3); rc |= dev_alloc(S, &P.co2, nb * CCAP * 3);
    rc |= dev_alloc(S, &P.cand, nb * NCANDCAP * 3); rc |= dev_alloc(S, &P.py_u, nb * PYCAP); rc |= dev_alloc(S, &P.py_state, nb * 625);
    rc |= dev_alloc(S, &P.nn, nb * OCAP); rc |= dev_alloc(S, &P.act_list, nb * NCAP);
    rc |= dev_alloc(S, &P.sorted, nb * SORTCAP); rc |= dev_alloc(S, &P.gnode, nb * GCAP); rc |= dev_alloc(S, &P.gstart, nb * GCAP);
    rc |= dev_alloc(S, &P.gcount, nb * GCAP); rc |= dev_alloc(S, &P.rec, nb * GCAP);
    rc |= dev_alloc(S, &P.glist, nb * GCAP); rc |= dev_alloc(S, &P.child_group, nb * NCAP); rc |= dev_alloc(S, &P.fl_rec, nb * 2 * MURRAY_FLUSH_LDS);
    rc |= dev_alloc(S, &P.kd_idx, nb * OCAP); rc |= dev_alloc(S, &P.kd_rank, nb * OCAP);
    rc |= dev_alloc(S, &P.removed, nb * OCAP); rc |= dev_alloc(S, &P.ven_near, nb * OCAP); rc |= dev_alloc(S, &P.hashes, nb * OCAP);
    rc |= dev_alloc(S, &P.pairs, nb * PCAP); rc |= dev_alloc(S, &P.set_hash, nb * SETCAP); rc |= dev_alloc(S, &P.set_key, nb * SETCAP);
    rc |= dev_alloc(S, &P.grid_pts, nb * GRID_N * 3); rc |= dev_alloc(S, &P.cand_idx, nb * NCANDCAP);
    rc |= dev_alloc(S, &P.tmp_int, nb * (OCAP + 2 * NCANDCAP)); rc |= dev_alloc(S, &P.tmp_dbl, nb * OCAP * 3);
    rc |= dev_alloc(S, &P.sc, nb); rc |= dev_alloc(S, &P.iters, S->iters.size() + 1);
    rc |= dev_alloc(S, &P.reqs, (size_t)2 * REQ_CAP); rc |= dev_alloc(S, &P.req_count, 4); rc |= dev_alloc(S, &P.bif_results, (size_t)2 * REQ_CAP * 6);
    rc |= dev_alloc(S, &P.mt_state, nb * 625); rc |= dev_alloc(S, &P.valid_count, nb);
    if (cfg.fixed()) {       // one voxel list and the mask itself, shared by every sample
        P.valid_stride = 0;
        rc |= dev_alloc(S, &P.valid, S->fixed_valid.size());
        rc |= dev_alloc(S, &S->d_mask, cfg.geometry.size());
        if (!rc) {
            OCTA_HIP_CHECK(hipMemcpy(P.valid, S->fixed_valid.data(), S->fixed_valid.size() * 2, hipMemcpyHostToDevice));
            OCTA_HIP_CHECK(hipMemcpy(S->d_mask, cfg.geometry.data(), cfg.geometry.size(), hipMemcpyHostToDevice));
            P.C.mask = S->d_mask;
        }
    } else {
        P.valid_stride = (size_t)76 * 76 * 3;
        rc |= dev_alloc(S, &P.valid, nb * P.valid_stride);
    }
    rc |= dev_alloc(S, &P.n_per_iter, S->iters.size() + 1);  // kept for diagnostics
    rc |= dev_alloc(S, &P.next_sample, 4);
    rc |= dev_alloc(S, &P.trace, nb * (S->iters.size() + 1) * 4);
#ifdef OCTA_SIM_DEBUG_SAT
    rc |= dev_alloc(S, &P.dbg, nb * (S->iters.size() + 1) * 16);
#endif
    P.wg_scratch = nullptr;
#if OCTA_SIM_LARGE
    rc |= dev_alloc(S, &P.wg_scratch, nb * (size_t)SIM_USER_BYTES);
#endif
    P.n_samples = B;
    if (!rc) {       // pinned staging of a run's per-sample set-up (sim_run_impl): allocated here, not in the first run
        const size_t n_mt = (size_t)B * 625, n_valid = (size_t)B * P.valid_stride;
        const size_t stage_need = ((n_valid * 2 + 255) & ~(size_t)255) + 2 * n_mt * 4;
        if (hipHostMalloc(&S->init_stage, stage_need) != hipSuccess) { octa::set_error("octa_sim_create: hipHostMalloc (set-up staging) failed"); S->init_stage = nullptr; rc = -1; }
        else { S->init_stage_bytes = stage_need; memset(S->init_stage, 0, stage_need); }
    }
    if (!rc) {
        hipError_t e1 = hipHostMalloc((void **)&S->h_reqs, sizeof(BifRequest) * 2 * REQ_CAP);
        hipError_t e2 = hipHostMalloc((void **)&S->h_results, sizeof(double) * 2 * REQ_CAP * 6);
        hipError_t e3 = hipHostMalloc((void **)&S->h_req_count, sizeof(int) * 4);
        if (e1 != hipSuccess || e2 != hipSuccess || e3 != hipSuccess) { octa::set_error("octa_sim_create: hipHostMalloc failed"); rc = -1; }
    }
    if (!rc) {
        const unsigned fl = hipHostMallocCoherent | hipHostMallocMapped;
        hipError_t e1 = hipHostMalloc((void **)&S->mail.reqs, sizeof(BifRequest) * nb * REQ_PER_SAMPLE, fl);
        hipError_t e2 = hipHostMalloc((void **)&S->mail.results, sizeof(double) * nb * REQ_PER_SAMPLE * 6, fl);
        hipError_t e3 = hipHostMalloc((void **)&S->mail.req_n, sizeof(int) * (nb * 5 + 18), fl);      // + [B] longs: publication stamps of the diagnostic build
        if (e1 != hipSuccess || e2 != hipSuccess || e3 != hipSuccess) { octa::set_error("octa_sim_create: hipHostMalloc (mailbox) failed"); rc = -1; }
        else { S->mail.req_ticket = S->mail.req_n + nb; S->mail.resp_ticket = S->mail.req_n + 2 * nb; S->mail.done = S->mail.req_n + 3 * nb; S->mail.req_time = reinterpret_cast<long *>(S->mail.req_n + ((3 * nb + 16 + 1) & ~(size_t)1)); }
        const char *ls = getenv("OCTA_SIM_LOCKSTEP");
        S->lockstep = ls && ls[0] == '1';
        if (const char *e = getenv("OCTA_SIM_MAIL_TIMEOUT_MS")) { double v = atof(e); if (v >= 1.0) S->mail_timeout_ms = v; }
        if (const char *e = getenv("OCTA_SIM_TEST_HOST_STALL_MS")) S->test_stall_ms = atoi(e);
        if (const char *e = getenv("OCTA_SIM_PARK_MS")) { double v = atof(e); if (v >= 0.0) S->park_ms = v; }
        S->grid_cap = SIM_WG_PER_CU * (ctx->num_cus > 0 ? ctx->num_cus : 256);
        if (const char *e = getenv("OCTA_SIM_GRID")) { int v = atoi(e); if (v >= 1) S->grid_cap = v; }
        // several ranks per host (one service thread per step in flight and rank): give the cores back sooner
        if (const char *e = getenv("WORLD_SIZE")) { if (atoi(e) > 1) S->spin_scans = 256; }
        if (const char *e = getenv("OCTA_SIM_SPIN_SCANS")) S->spin_scans = atol(e);
    }
    if (!rc) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void *>(sim_persistent_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, (int)SIM_LDS);
        if (e != hipSuccess) { octa::set_error("octa_sim_create: cannot reserve %zu B of LDS: %s", SIM_LDS, hipGetErrorString(e)); rc = -1; }
    }
    if (!rc) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void *>(sim_iter_a_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, (int)SIM_LDS);
        if (e == hipSuccess) e = hipFuncSetAttribute(reinterpret_cast<const void *>(sim_iter_b_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, (int)SIM_LDS);
        if (e != hipSuccess) { octa::set_error("octa_sim_create: cannot reserve %zu B of LDS: %s", SIM_LDS, hipGetErrorString(e)); rc = -1; }
    }
    if (!rc) for (int k = 0; k < 3; k++) if (hipEventCreate(&S->ev[k]) != hipSuccess) { octa::set_error("octa_sim_create: hipEventCreate failed"); rc = -1; }
    if (rc) { octa_sim_destroy(S); return -1; }
    *out = S;
    return 0;
}

extern "C" void octa_sim_destroy(OCTA_SIM_T *S) {
    if (!S) return;
    hipError_t e = hipSetDevice(S->ctx->device);
    for (void *p : S->allocs) e = hipFree(p);
    if (S->h_reqs) e = hipHostFree(S->h_reqs);
    if (S->h_results) e = hipHostFree(S->h_results);
    if (S->h_req_count) e = hipHostFree(S->h_req_count);
    if (S->mail.reqs) e = hipHostFree(S->mail.reqs);
    if (S->mail.results) e = hipHostFree(S->mail.results);
    if (S->mail.req_n) e = hipHostFree(S->mail.req_n);
    for (int k = 0; k < 3; k++) if (S->ev[k]) e = hipEventDestroy(S->ev[k]);
    if (S->export_stage) e = hipHostFree(S->export_stage);
    if (S->init_stage) e = hipHostFree(S->init_stage);
    if (S->export_stream) e = hipStreamDestroy(S->export_stream);
    if (S->d_edge_off) e = hipFree(S->d_edge_off);
    (void)e;
    delete S;
}

namespace {
// what differs between the entry points: how sample s gets its FAZ radius, stump nodes and generator states
struct SampleSource {
    const uint32_t *np_seeds = nullptr;
    const uint64_t *py_seeds = nullptr;
    const double *faz = nullptr, *stumps = nullptr;        // [B], [B][2][2 * n_trees][3]
    const uint32_t *np_states = nullptr, *py_states = nullptr;   // [B][625] each: 624 words + position
    void fill(const SimConfig &cfg, int s, SampleInit *I) const {
        if (np_seeds) { init_sample(cfg, np_seeds[s], py_seeds[s], I); return; }
        Mt19937 np, py;
        memcpy(np.mt, np_states + (size_t)s * 625, 624 * 4); np.idx = (int)np_states[(size_t)s * 625 + 624];
        memcpy(py.mt, py_states + (size_t)s * 625, 624 * 4); py.idx = (int)py_states[(size_t)s * 625 + 624];
        const size_t n = (size_t)2 * cfg.n_trees * 3;
        init_sample_given(cfg, faz[s], stumps + (size_t)s * 2 * n, stumps + (size_t)s * 2 * n + n, np, py, I);
    }
};
int sim_run_impl(OCTA_SIM_T *S, const SampleSource &src, octa_bif_fn bif, void *user, void *stream_);
}  // namespace

extern "C" int octa_sim_run(OCTA_SIM_T *S, const uint32_t *h_np_seeds, const uint64_t *h_py_seeds, octa_bif_fn bif, void *user,
                            void *stream_) {
    if (!S || !h_np_seeds || !h_py_seeds || !bif) { octa::set_error("octa_sim_run: bad arguments"); return -2; }
    SampleSource src;
    src.np_seeds = h_np_seeds; src.py_seeds = h_py_seeds;
    return sim_run_impl(S, src, bif, user, stream_);
}

extern "C" int octa_sim_run_states(OCTA_SIM_T *S, const double *h_faz_radius, const double *h_stumps, const uint32_t *h_np_states,
                                   const uint32_t *h_py_states, octa_bif_fn bif, void *user, void *stream_) {
    if (!S || !h_faz_radius || !h_stumps || !h_np_states || !h_py_states || !bif) { octa::set_error("octa_sim_run_states: bad arguments"); return -2; }
    for (int s = 0; s < S->B; s++)
        if (h_np_states[(size_t)s * 625 + 624] > 624 || h_py_states[(size_t)s * 625 + 624] > 624) { octa::set_error("octa_sim_run_states: corrupt generator state"); return -2; }
    SampleSource src;
    src.faz = h_faz_radius; src.stumps = h_stumps; src.np_states = h_np_states; src.py_states = h_py_states;
    return sim_run_impl(S, src, bif, user, stream_);
}

// numpy's generator of sample `sample` after the run: 624 words + position (the candidate stream is its only consumer)
extern "C" int octa_sim_np_state(OCTA_SIM_T *S, int sample, uint32_t *h_state625) {
    if (!S || !S->ran || !h_state625 || sample < 0 || sample >= S->B) { octa::set_error("octa_sim_np_state: bad arguments"); return -2; }
    OCTA_HIP_CHECK(hipSetDevice(S->ctx->device));
    OCTA_HIP_CHECK(hipMemcpy(h_state625, S->P.mt_state + (size_t)sample * 625, 625 * sizeof(uint32_t), hipMemcpyDeviceToHost));
    return 0;
}

namespace {
int sim_run_impl(OCTA_SIM_T *S, const SampleSource &src, octa_bif_fn bif, void *user, void *stream_) {
    hipStream_t stream = (hipStream_t)stream_;
    OCTA_HIP_CHECK(hipSetDevice(S->ctx->device));
    const int B = S->B;
    BatchPtrs &P = S->P;
    const SimConst &C = S->C;
    // ---- host init of every sample (FAZ radius, validity mask, stumps, RNG streams)
    {
        std::vector<SampleScalars> sc(B);
        // the large staging arrays (18 MB of valid-voxel lists, 2.6 MB of generator states for 512 samples) live in ONE pinned block that a
        // simulator keeps: allocated and zero-filled per run they cost ~3 ms of page faults and a pageable copy, between two kernels
        const size_t n_mt = (size_t)B * 625, n_valid = (size_t)B * P.valid_stride;
        const size_t stage_need = ((n_valid * 2 + 255) & ~(size_t)255) + 2 * n_mt * 4;
        if (S->init_stage_bytes < stage_need) {
            if (S->init_stage) OCTA_HIP_CHECK(hipHostFree(S->init_stage));
            S->init_stage = nullptr; S->init_stage_bytes = 0;
            OCTA_HIP_CHECK(hipHostMalloc(&S->init_stage, stage_need));
            S->init_stage_bytes = stage_need;
            memset(S->init_stage, 0, stage_need);
        }
        unsigned short *valid = static_cast<unsigned short *>(S->init_stage);       // entries behind a sample's count are never read
        unsigned *mt = reinterpret_cast<unsigned *>(static_cast<char *>(S->init_stage) + ((n_valid * 2 + 255) & ~(size_t)255));
        unsigned *py_mt = mt + n_mt;
        std::vector<unsigned> vcount(B);
        const int n0 = 2 * S->cfg.n_trees;
        std::vector<double> npos((size_t)B * 2 * n0 * 3);
        constexpr bool time_init = false;      // development aid: host-init timing on stderr
        const auto t_init0 = std::chrono::steady_clock::now();
        // the samples are independent: a few host threads share them (round 5: 5.6 of the 6.9 ms this block takes for 512 samples were this
        // loop on one core, and the block sits between two persistent kernels of the pipeline). OCTA_SIM_INIT_THREADS overrides.
        static const int init_threads = [] {
            if (const char *e = getenv("OCTA_SIM_INIT_THREADS")) { const int v = atoi(e); if (v >= 1) return v < 64 ? v : 64; }
            int world = 1;
            if (const char *e = getenv("WORLD_SIZE")) { const int v = atoi(e); if (v > 1) world = v; }
            const int hw = (int)std::thread::hardware_concurrency();
            const int v = hw / (2 * world);                 // half of this rank's share of the host: the service threads and the trainer need the rest
            return v < 1 ? 1 : (v > 8 ? 8 : v);
        }();
        std::atomic<int> bad_sample{-1};
        auto fill_range = [&](int s_begin, int s_end) {
            SampleInit I;
            I.want_py_u = false;             // the uniforms are drawn on the device from the generator state (py_uniform_kernel)
            for (int s = s_begin; s < s_end; s++) {
                src.fill(S->cfg, s, &I);
                memset(&sc[s], 0, sizeof(SampleScalars));
                sc[s].faz_radius = I.faz_radius;
                sc[s].py_cap = PYCAP;
                sc[s].n_nodes[0] = sc[s].n_nodes[1] = n0;
                memcpy(&mt[(size_t)s * 625], I.np_state.mt, 624 * 4);
                mt[(size_t)s * 625 + 624] = (unsigned)I.np_state.idx;
                vcount[s] = (unsigned)(I.valid.size() / 3);
                if (vcount[s] == 0) { int none = -1; bad_sample.compare_exchange_strong(none, s); continue; }
                if (P.valid_stride) memcpy(&valid[(size_t)s * P.valid_stride], I.valid.data(), I.valid.size() * 2);
                memcpy(&py_mt[(size_t)s * 625], I.py_state.mt, 624 * 4);
                py_mt[(size_t)s * 625 + 624] = (unsigned)I.py_state.idx;
                for (int f = 0; f < 2; f++) memcpy(&npos[((size_t)s * 2 + f) * n0 * 3], I.pos[f].data(), sizeof(double) * n0 * 3);
            }
        };
        {
            const int nt = B >= 64 ? (init_threads < B / 32 ? init_threads : B / 32) : 1;
            std::vector<std::thread> pool;
            const int per = (B + nt - 1) / nt;
            for (int t = 1; t < nt; t++) pool.emplace_back(fill_range, t * per < B ? t * per : B, (t + 1) * per < B ? (t + 1) * per : B);
            fill_range(0, per < B ? per : B);
            for (auto &th : pool) th.join();
        }
        if (bad_sample.load() >= 0) { octa::set_error("octa_sim_run: sample %d has no valid voxel", bad_sample.load()); return -2; }
        if (time_init) fprintf(stderr, "[octa_sim] host init: per-sample fill %.2f ms for %d samples\n", std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t_init0).count(), B);
        OCTA_HIP_CHECK(hipMemcpyAsync(P.sc, sc.data(), sizeof(SampleScalars) * B, hipMemcpyHostToDevice, stream));
        OCTA_HIP_CHECK(hipMemcpyAsync(P.mt_state, mt, n_mt * 4, hipMemcpyHostToDevice, stream));
        if (P.valid_stride) OCTA_HIP_CHECK(hipMemcpyAsync(P.valid, valid, n_valid * 2, hipMemcpyHostToDevice, stream));
        OCTA_HIP_CHECK(hipMemcpyAsync(P.valid_count, vcount.data(), vcount.size() * 4, hipMemcpyHostToDevice, stream));
        OCTA_HIP_CHECK(hipMemcpyAsync(P.py_state, py_mt, n_mt * 4, hipMemcpyHostToDevice, stream));
        hipLaunchKernelGGL(py_uniform_kernel, dim3((unsigned)B), dim3(64), 0, stream, P.py_state, P.py_u, PYCAP);
        OCTA_HIP_CHECK(hipGetLastError());
        OCTA_HIP_CHECK(hipMemcpyAsync(P.iters, S->iters.data(), sizeof(IterParams) * S->iters.size(), hipMemcpyHostToDevice, stream));
        std::vector<int> Ns(S->iters.size() + 1, 0);
        for (size_t i = 0; i < S->iters.size(); i++) Ns[i] = S->iters[i].N;
        OCTA_HIP_CHECK(hipMemcpyAsync(P.n_per_iter, Ns.data(), sizeof(int) * Ns.size(), hipMemcpyHostToDevice, stream));
        // stump nodes: root (kappa 4, no parent) + one child per tree; one strided copy per array puts the
        // n0 leading entries of every sample's slice in place
        std::vector<double> rad((size_t)B * n0, C.r), kap((size_t)B * n0, 4.0);
        std::vector<int> par((size_t)B * n0), c0((size_t)B * n0), c1((size_t)B * n0, -1);
        std::vector<unsigned char> nch((size_t)B * n0), act((size_t)B * n0, 1);
        for (size_t k = 0; k < (size_t)B * n0; k++) {
            const int i = (int)(k % n0);
            par[k] = (i & 1) ? i - 1 : -1; c0[k] = (i & 1) ? -1 : i + 1; nch[k] = (i & 1) ? 0 : 1;
        }
        std::vector<double> pos_f((size_t)B * n0 * 3);
        auto put = [&](void *dst, size_t elem, const void *src, size_t per_sample_cap, size_t count) -> hipError_t {
            return hipMemcpy2DAsync(dst, per_sample_cap * elem, src, count * elem, count * elem, (size_t)B, hipMemcpyHostToDevice, stream);
        };
        for (int f = 0; f < 2; f++) {
            OCTA_HIP_CHECK(hipMemsetAsync(P.nact[f], 0, (size_t)B * NCAP, stream));
            for (int s = 0; s < B; s++) memcpy(&pos_f[(size_t)s * n0 * 3], &npos[((size_t)s * 2 + f) * n0 * 3], sizeof(double) * n0 * 3);
            OCTA_HIP_CHECK(put(P.npos[f], sizeof(double), pos_f.data(), (size_t)NCAP * 3, (size_t)n0 * 3));
            OCTA_HIP_CHECK(hipStreamSynchronize(stream));  // pos_f is reused for the second forest
            OCTA_HIP_CHECK(put(P.nrad[f], sizeof(double), rad.data(), NCAP, n0));
            OCTA_HIP_CHECK(put(P.nkap[f], sizeof(double), kap.data(), NCAP, n0));
            OCTA_HIP_CHECK(put(P.npar[f], sizeof(int), par.data(), NCAP, n0));
            OCTA_HIP_CHECK(put(P.nch0[f], sizeof(int), c0.data(), NCAP, n0));
            OCTA_HIP_CHECK(put(P.nch1[f], sizeof(int), c1.data(), NCAP, n0));
            OCTA_HIP_CHECK(put(P.nnch[f], 1, nch.data(), NCAP, n0));
            OCTA_HIP_CHECK(put(P.nact[f], 1, act.data(), NCAP, n0));
        }
        OCTA_HIP_CHECK(hipMemsetAsync(P.req_count, 0, sizeof(int) * 4, stream));
#ifdef OCTA_SIM_DEBUG_SAT
        OCTA_HIP_CHECK(hipMemsetAsync(P.dbg, 0xff, sizeof(int) * (size_t)B * (S->iters.size() + 1) * 16, stream));
#endif
        OCTA_HIP_CHECK(hipMemsetAsync(P.child_group, 0, sizeof(int) * (size_t)B * NCAP, stream));
        OCTA_HIP_CHECK(hipStreamSynchronize(stream));  // host vectors go out of scope
        if (time_init) fprintf(stderr, "[octa_sim] host init: %.2f ms in all (fill + copies + fills)\n", std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t_init0).count());
    }
    // ---- iterations
    auto serve = [&](int slot) -> int {
        OCTA_HIP_CHECK(hipMemcpyAsync(S->h_req_count, P.req_count, sizeof(int) * 2, hipMemcpyDeviceToHost, stream));
        OCTA_HIP_CHECK(hipStreamSynchronize(stream));
        int n = S->h_req_count[slot];
        if (n > REQ_CAP) n = REQ_CAP;
        if (n > 0) {
            OCTA_HIP_CHECK(hipMemcpyAsync(S->h_reqs, P.reqs + (size_t)slot * REQ_CAP, sizeof(BifRequest) * n, hipMemcpyDeviceToHost, stream));
            OCTA_HIP_CHECK(hipStreamSynchronize(stream));
            auto t0 = std::chrono::steady_clock::now();
            bif(n, reinterpret_cast<const octa_bif_request *>(S->h_reqs), S->h_results, user);
            S->ms_host_bif += std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count();
            S->n_bif_req += n;
            OCTA_HIP_CHECK(hipMemcpyAsync(P.bif_results, S->h_results, sizeof(double) * 6 * n, hipMemcpyHostToDevice, stream));
            OCTA_HIP_CHECK(hipMemsetAsync(P.req_count + slot, 0, sizeof(int), stream));
        }
        return 0;
    };
    S->ms_a = S->ms_b = S->ms_total = S->ms_host_bif = 0; S->n_a = S->n_b = S->n_bif_req = 0;
    auto wall0 = std::chrono::steady_clock::now();
    if (!S->lockstep) {
        // ---- persistent form: ONE launch runs every iteration of every sample; this thread answers mailbox tickets until all
        // workgroups have signed off (a counter in the pinned block, incremented by the kernel with a system-scope atomic) or the
        // launch has completed. Workgroups whose answer did not reach them in time have PARKED (mail_roundtrip): their requests
        // are served here, at the kernel boundary, and the kernel is launched again for them. While a launch runs this thread's
        // only job is the mailbox; it asks the runtime about the launch only after a millisecond of silence.
        HostMail &M = S->mail;
        using clk = std::chrono::steady_clock;
        M.timeout_ticks = (long)(S->mail_timeout_ms * 1e5);
        M.park_ticks = (long)(S->park_ms * 1e5);
        S->diag_max_gap_ms = 0; S->diag_tickets = 0; S->diag_max_bif_ms = 0; S->diag_relaunches = 0; S->diag_parked = 0; S->diag_passes = 0;
        S->h_sc.resize(B);
        bool stalled_once = false;
        double ms_kernels = 0;
        // OCTA_SIM_MAIL_DIAG=1 (tools/sim_mailbox_profile.py): what a ticket waits for, from the host's side -- passes of this loop, and, with a
        // -DOCTA_SIM_PROF_MAIL library, the device's publication stamp against the host's clock
        const bool mail_diag = getenv("OCTA_SIM_MAIL_DIAG") != nullptr;
        double diag_min_d = 1e300, diag_sum_serv = 0;
        std::vector<double> diag_lat;
        long pass_hist[6] = {0, 0, 0, 0, 0, 0};
        int launches = 0;
        auto answer = [&](int s, int n) {
            if (n > REQ_PER_SAMPLE) n = REQ_PER_SAMPLE;
            if (n <= 0) return;
            auto t0 = clk::now();
            bif(n, reinterpret_cast<const octa_bif_request *>(M.reqs + (size_t)s * REQ_PER_SAMPLE), M.results + (size_t)s * REQ_PER_SAMPLE * 6, user);
            const double bms = std::chrono::duration<double, std::milli>(clk::now() - t0).count();
            S->ms_host_bif += bms;
            if (bms > S->diag_max_bif_ms) S->diag_max_bif_ms = bms;
            S->n_bif_req += n;
        };
        while (true) {
            for (int s = 0; s < 3 * B + 1; s++) M.req_n[s] = 0;
            memset(M.req_time, 0, sizeof(long) * (size_t)B);
            std::vector<int> seen(B, 0);
            OCTA_HIP_CHECK(hipMemsetAsync(P.next_sample, 0, sizeof(int), stream));
            // this launch's ticket and its sign-in counter (csrc/sim_api.cpp): the counter of the launch before the previous one is reused
            M.launch_flag = octa_sim_launch_flag(S->ctx->device);
            M.ticket = (int)(octa_sim_next_ticket() & 0x3fffffff);
            if (M.launch_flag) {
                OCTA_HIP_CHECK(hipMemsetAsync(M.launch_flag + 1 + (M.ticket & 1), 0, sizeof(int), stream));
                OCTA_HIP_CHECK(hipMemsetAsync(M.launch_flag + 3 + (M.ticket & 1), 0, sizeof(int), stream));
            }
            S->last_ticket = M.ticket;
            OCTA_HIP_CHECK(hipEventRecord(S->ev[0], stream));
            const int grid = B < S->grid_cap ? B : S->grid_cap;
            hipLaunchKernelGGL(sim_persistent_kernel, dim3((unsigned)grid), dim3(SIM_THREADS), SIM_LDS, stream, P, M);
            OCTA_HIP_CHECK(hipGetLastError());
            OCTA_HIP_CHECK(hipEventRecord(S->ev[1], stream));
            launches++;
            long idle = 0;
            auto last_progress = clk::now(), iter_start = last_progress;
            int last_done = 0;
            while (true) {
                {   // longest single pass of this loop, whatever it was spent in (callback, descheduling, ...)
                    const auto t = clk::now();
                    const double it_ms = std::chrono::duration<double, std::milli>(t - iter_start).count();
                    if (it_ms > S->diag_max_gap_ms) S->diag_max_gap_ms = it_ms;
                    if (mail_diag) pass_hist[it_ms < 0.01 ? 0 : it_ms < 0.03 ? 1 : it_ms < 0.1 ? 2 : it_ms < 0.3 ? 3 : it_ms < 1.0 ? 4 : 5]++;
                    iter_start = t;
                }
                bool any = false;
                S->diag_passes++;
                for (int s = 0; s < B; s++) {
                    const int t = __atomic_load_n(M.req_ticket + s, __ATOMIC_ACQUIRE);
                    if (t == seen[s]) continue;
                    any = true;
                    if (S->test_stall_ms > 0 && !stalled_once) {   // test hook: the host goes away once, with a ticket pending
                        stalled_once = true;
                        std::this_thread::sleep_for(std::chrono::milliseconds(S->test_stall_ms));
                    }
                    const auto h_seen = clk::now();
                    answer(s, __atomic_load_n(M.req_n + s, __ATOMIC_RELAXED));
                    __atomic_store_n(M.resp_ticket + s, t, __ATOMIC_RELEASE);
                    if (mail_diag) {      // host clock at sighting / answer against the device's publication stamp (clock offset = the smallest difference seen)
                        const double d = std::chrono::duration<double, std::micro>(h_seen - wall0).count() - (double)__atomic_load_n(M.req_time + s, __ATOMIC_RELAXED) * 1e-2;
                        if (d < diag_min_d) diag_min_d = d;
                        diag_sum_serv += std::chrono::duration<double, std::micro>(clk::now() - h_seen).count();
                        diag_lat.push_back(d);
                    }
                    seen[s] = t;
                    S->diag_tickets++;
                }
                const int done = __atomic_load_n(M.done, __ATOMIC_ACQUIRE);
                if (done >= B) break;
                if (any || done != last_done) { idle = 0; last_done = done; last_progress = clk::now(); continue; }
                if ((++idle & 63) == 0) {
                    const double silent = std::chrono::duration<double, std::milli>(clk::now() - last_progress).count();
                    if (silent > 1.0) {
                        hipError_t q = hipEventQuery(S->ev[1]);
                        if (q == hipSuccess) break;                       // launch over although not every sign-off was seen
                        if (q != hipErrorNotReady) { octa::set_error("octa_sim_run: kernel failed: %s", hipGetErrorString(q)); return -1; }
                        if (silent > 1000.0 * 120.0) { octa::set_error("octa_sim_run: the simulation kernel made no progress for 120 s"); return -1; }
                    }
                    if (idle > S->spin_scans) std::this_thread::sleep_for(std::chrono::microseconds(20));
                }
            }
            float ms = 0;
            OCTA_HIP_CHECK(hipEventSynchronize(S->ev[1]));
            OCTA_HIP_CHECK(hipEventElapsedTime(&ms, S->ev[0], S->ev[1]));
            ms_kernels += ms;
            // kernel boundary: everything the launch wrote is visible now
            OCTA_HIP_CHECK(hipMemcpyAsync(S->h_sc.data(), P.sc, sizeof(SampleScalars) * B, hipMemcpyDeviceToHost, stream));
            OCTA_HIP_CHECK(hipStreamSynchronize(stream));
            int n_parked = 0;
            for (int s = 0; s < B; s++) {
                if (!S->h_sc[s].parked || S->h_sc[s].err) continue;
                n_parked++;
                answer(s, M.req_n[s]);
            }
            if (n_parked == 0) break;
            S->diag_relaunches++;
            S->diag_parked += n_parked;
            if (launches > 4 * (int)S->iters.size() + 16) { octa::set_error("octa_sim_run: %d launches without finishing (parking loop)", launches); return -1; }
        }
        const float ms = (float)ms_kernels;
        S->ms_b = ms; S->n_b = launches;
        if (mail_diag) {
            fprintf(stderr, "[octa] mailbox service: %ld passes over %d tickets in %.1f ms of kernel (%.2f us per pass; <10 us %ld, <30 %ld, <100 %ld, <300 %ld, <1000 %ld, longer %ld; longest %.3f ms), "
                            "%ld tickets, %.1f ms in the callback\n", S->diag_passes, B, ms_kernels, 1e3 * ms_kernels / (double)(S->diag_passes ? S->diag_passes : 1),
                    pass_hist[0], pass_hist[1], pass_hist[2], pass_hist[3], pass_hist[4], pass_hist[5], S->diag_max_gap_ms, S->diag_tickets, S->ms_host_bif);
            if (!diag_lat.empty() && S->mail.req_time[0] != 0) {
                long h[7] = {0, 0, 0, 0, 0, 0, 0};
                double sum = 0;
                for (double d : diag_lat) { const double x = d - diag_min_d; sum += x; h[x < 10 ? 0 : x < 30 ? 1 : x < 100 ? 2 : x < 300 ? 3 : x < 1000 ? 4 : x < 3000 ? 5 : 6]++; }
                fprintf(stderr, "[octa]   publication (device clock) -> seen by the host, above the fastest ticket: %.1f us on average (<10 us %ld, <30 %ld, <100 %ld, <300 %ld, <1000 %ld, <3000 %ld, "
                                "longer %ld); seen -> answered %.1f us\n", sum / (double)diag_lat.size(), h[0], h[1], h[2], h[3], h[4], h[5], h[6], diag_sum_serv / (double)diag_lat.size());
            }
        }
    } else
    for (int it = 0; it <= C.n_iter; it++) {
        float ms = 0;
        OCTA_HIP_CHECK(hipEventRecord(S->ev[0], stream));
        hipLaunchKernelGGL(sim_iter_a_kernel, dim3((unsigned)B), dim3(SIM_THREADS), SIM_LDS, stream, P, it, it > 0 ? 1 : 0);
        OCTA_HIP_CHECK(hipGetLastError());
        OCTA_HIP_CHECK(hipEventRecord(S->ev[1], stream));
        if (it == C.n_iter) {
            OCTA_HIP_CHECK(hipEventSynchronize(S->ev[1]));
            OCTA_HIP_CHECK(hipEventElapsedTime(&ms, S->ev[0], S->ev[1]));
            S->ms_a += ms; S->n_a++;
            break;
        }
        if (serve(0)) return -1;
        OCTA_HIP_CHECK(hipEventElapsedTime(&ms, S->ev[0], S->ev[1]));
        S->ms_a += ms; S->n_a++;
        OCTA_HIP_CHECK(hipEventRecord(S->ev[0], stream));
        hipLaunchKernelGGL(sim_iter_b_kernel, dim3((unsigned)B), dim3(SIM_THREADS), SIM_LDS, stream, P, it);
        OCTA_HIP_CHECK(hipGetLastError());
        OCTA_HIP_CHECK(hipEventRecord(S->ev[1], stream));
        if (serve(1)) return -1;
        OCTA_HIP_CHECK(hipEventElapsedTime(&ms, S->ev[0], S->ev[1]));
        S->ms_b += ms; S->n_b++;
    }
    S->ms_total = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - wall0).count();
    if (S->lockstep) {                   // (the persistent form has fetched the samples' scalars at its last kernel boundary)
        S->h_sc.resize(B);
        OCTA_HIP_CHECK(hipMemcpyAsync(S->h_sc.data(), P.sc, sizeof(SampleScalars) * B, hipMemcpyDeviceToHost, stream));
        OCTA_HIP_CHECK(hipStreamSynchronize(stream));
    }
    S->ran = true;
#ifdef OCTA_SIM_DEBUG_SAT
    if (const char *path = getenv("OCTA_SIM_DEBUG_DUMP")) {
        std::vector<int> h((size_t)B * S->iters.size() * 16);
        OCTA_HIP_CHECK(hipMemcpy(h.data(), P.dbg, h.size() * sizeof(int), hipMemcpyDeviceToHost));
        if (FILE *fp = fopen(path, "wb")) { fwrite(h.data(), sizeof(int), h.size(), fp); fclose(fp); }
    }
#endif
    for (int s = 0; s < B; s++)
        if (S->h_sc[s].err) {
            if ((S->h_sc[s].err & ERR_HOST_TIMEOUT) && !S->lockstep)
                octa::set_error("octa_sim_run: sample %d failed with capacity/error bits 0x%x: with parking disabled (OCTA_SIM_PARK_MS=0) a workgroup waited "
                                "more than %.0f ms for the host's mailbox answer (request ticket %d, answered %d, %ld tickets served this run, longest pass of the "
                                "service loop %.1f ms, longest bifurcation callback %.1f ms; the workgroup began waiting for ticket %ld at %.1f ms into its "
                                "launch, polled %ld times, at its deadline read answer %ld and its own request word back as %ld)", s, S->h_sc[s].err,
                                S->mail_timeout_ms, S->mail.req_ticket[s], S->mail.resp_ticket[s], S->diag_tickets, S->diag_max_gap_ms, S->diag_max_bif_ms,
                                S->h_sc[s].prof[13], S->h_sc[s].prof[12] * 1e-5, S->h_sc[s].prof[14], S->h_sc[s].prof[11], S->h_sc[s].prof[15]);
            else
                octa::set_error("octa_sim_run: sample %d failed with capacity/error bits 0x%x", s, S->h_sc[s].err);
            return -3;
        }
    return 0;
}
}  // namespace

extern "C" int octa_sim_edge_offsets(OCTA_SIM_T *S, int64_t *h_edge_off, int64_t *h_n_art) {
    if (!S || !S->ran || !h_edge_off) { octa::set_error("octa_sim_edge_offsets: run the simulation first"); return -2; }
    h_edge_off[0] = 0;
    for (int s = 0; s < S->B; s++) {
        const SampleScalars &sc = S->h_sc[s];
        int64_t na = sc.n_nodes[0] - S->cfg.n_trees, nv = sc.n_nodes[1] - S->cfg.n_trees;
        h_edge_off[s + 1] = h_edge_off[s] + na + nv;
        if (h_n_art) h_n_art[s] = na;
    }
    return 0;
}

extern "C" int octa_sim_export_edges(OCTA_SIM_T *S, double *h_edges) {
    if (!S || !S->ran || !h_edges) { octa::set_error("octa_sim_export_edges: run the simulation first"); return -2; }
    OCTA_HIP_CHECK(hipSetDevice(S->ctx->device));
    const BatchPtrs &P = S->P;
    const int B = S->B;
    // Twelve strided copies (one per node array and forest: B rows of the longest forest's node count, pitch = the per-sample
    // capacity) into pinned staging + ONE wait, instead of a synchronous hipMemcpy per sample, forest and array (1 536 round
    // trips per 128-sample batch in round 1).
    int nmax[2] = {0, 0};
    for (int s = 0; s < B; s++) for (int f = 0; f < 2; f++) nmax[f] = S->h_sc[s].n_nodes[f] > nmax[f] ? S->h_sc[s].n_nodes[f] : nmax[f];
    size_t need = 0;
    for (int f = 0; f < 2; f++) need += (size_t)B * nmax[f] * (24 + 8 + 4 + 4 + 4 + 1) + 64 * 6;
    if (S->export_cap < need) {
        if (S->export_stage) { hipError_t e = hipHostFree(S->export_stage); (void)e; S->export_stage = nullptr; S->export_cap = 0; }
        OCTA_HIP_CHECK(hipHostMalloc(&S->export_stage, need + need / 4));
        S->export_cap = need + need / 4;
    }
    hipStream_t stream = S->export_stream;
    if (!stream) { OCTA_HIP_CHECK(hipStreamCreateWithFlags(&S->export_stream, hipStreamNonBlocking)); stream = S->export_stream; }
    char *base = static_cast<char *>(S->export_stage);
    double *pos[2], *rad[2];
    int *par[2], *c0[2], *c1[2];
    unsigned char *nch[2];
    size_t off = 0;
    auto take = [&](size_t bytes) { char *p = base + off; off += (bytes + 63) & ~(size_t)63; return p; };
    for (int f = 0; f < 2; f++) {
        const size_t n = (size_t)nmax[f];
        pos[f] = reinterpret_cast<double *>(take((size_t)B * n * 24)); rad[f] = reinterpret_cast<double *>(take((size_t)B * n * 8));
        par[f] = reinterpret_cast<int *>(take((size_t)B * n * 4)); c0[f] = reinterpret_cast<int *>(take((size_t)B * n * 4));
        c1[f] = reinterpret_cast<int *>(take((size_t)B * n * 4)); nch[f] = reinterpret_cast<unsigned char *>(take((size_t)B * n));
        if (n == 0) continue;
        auto get = [&](void *dst, const void *src, size_t elem, size_t cap_elems) -> hipError_t {
            return hipMemcpy2DAsync(dst, n * elem, src, cap_elems * elem, n * elem, (size_t)B, hipMemcpyDeviceToHost, stream);
        };
        OCTA_HIP_CHECK(get(pos[f], P.npos[f], 24, NCAP)); OCTA_HIP_CHECK(get(rad[f], P.nrad[f], 8, NCAP));
        OCTA_HIP_CHECK(get(par[f], P.npar[f], 4, NCAP)); OCTA_HIP_CHECK(get(c0[f], P.nch0[f], 4, NCAP));
        OCTA_HIP_CHECK(get(c1[f], P.nch1[f], 4, NCAP)); OCTA_HIP_CHECK(get(nch[f], P.nnch[f], 1, NCAP));
    }
    OCTA_HIP_CHECK(hipStreamSynchronize(stream));
    long total = 0;
    for (int s = 0; s < B; s++) {
        const SampleScalars &sc = S->h_sc[s];
        const double *cp[2], *cr[2];
        const int *cpar[2], *cc0[2], *cc1[2];
        const unsigned char *cn[2];
        for (int f = 0; f < 2; f++) {
            const size_t n = (size_t)nmax[f];
            cp[f] = pos[f] + (size_t)s * n * 3; cr[f] = rad[f] + (size_t)s * n; cpar[f] = par[f] + (size_t)s * n;
            cc0[f] = c0[f] + (size_t)s * n; cc1[f] = c1[f] + (size_t)s * n; cn[f] = nch[f] + (size_t)s * n;
        }
        long n_art = 0;
        long ne = export_edges(cp, cr, cpar, cc0, cc1, cn, sc.n_nodes, S->cfg.n_trees, h_edges + 7 * total, 1L << 40, &n_art);
        if (ne < 0) { octa::set_error("octa_sim_export_edges: export failed"); return -1; }
        total += ne;
    }
    return 0;
}

extern "C" int octa_sim_export_edges_device(OCTA_SIM_T *S, double *d_edges, void *stream_) {
    if (!S || !S->ran || !d_edges) { octa::set_error("octa_sim_export_edges_device: run the simulation first"); return -2; }
    OCTA_HIP_CHECK(hipSetDevice(S->ctx->device));
    hipStream_t stream = (hipStream_t)stream_;
    const int B = S->B;
    std::vector<long> off(B + 1, 0);
    for (int s = 0; s < B; s++) off[s + 1] = off[s] + (S->h_sc[s].n_nodes[0] - S->cfg.n_trees) + (S->h_sc[s].n_nodes[1] - S->cfg.n_trees);
    if (!S->d_edge_off) OCTA_HIP_CHECK(hipMalloc(&S->d_edge_off, sizeof(long) * (B + 1)));
    OCTA_HIP_CHECK(hipMemcpyAsync(S->d_edge_off, off.data(), sizeof(long) * (B + 1), hipMemcpyHostToDevice, stream));
    OCTA_HIP_CHECK(hipStreamSynchronize(stream));     // `off` is a local; the copy is a few hundred bytes
    const size_t lds = OCTA_SIM_LARGE ? 2048 : 2048 + (size_t)NCAP * 2 * sizeof(idx_t);
    hipLaunchKernelGGL(sim_export_kernel, dim3((unsigned)B, 2), dim3(256), lds, stream, S->P, S->d_edge_off, S->cfg.n_trees, d_edges);
    OCTA_HIP_CHECK(hipGetLastError());
    return 0;
}

extern "C" int octa_sim_trace(OCTA_SIM_T *S, int32_t *h_trace) {
    if (!S || !S->ran || !h_trace) { octa::set_error("octa_sim_trace: run the simulation first"); return -2; }
    OCTA_HIP_CHECK(hipSetDevice(S->ctx->device));
    OCTA_HIP_CHECK(hipMemcpy(h_trace, S->P.trace, sizeof(int32_t) * (size_t)S->B * S->iters.size() * 4, hipMemcpyDeviceToHost));
    return 0;
}

extern "C" int octa_sim_stats(OCTA_SIM_T *S, int64_t *h_stats) {
    if (!S || !S->ran || !h_stats) { octa::set_error("octa_sim_stats: run the simulation first"); return -2; }
    for (int s = 0; s < S->B; s++) {
        const SampleScalars &sc = S->h_sc[s];
        int64_t *o = h_stats + 32 * s;
        o[0] = sc.err; o[1] = sc.py_pos; o[2] = sc.murray_steps; o[3] = sc.n_bif; o[4] = sc.respec;
        o[5] = sc.n_nodes[0]; o[6] = sc.n_nodes[1];
        memcpy(&o[7], &sc.faz_radius, 8);
        for (int k = 0; k < 16; k++) o[8 + k] = sc.prof[k];
        for (int k = 0; k < 8; k++) o[24 + k] = sc.kdprof[k];
    }
    return 0;
}

extern "C" int octa_sim_fields(OCTA_SIM_T *S, int sample, double *h_oxy, int64_t cap_oxy, int64_t *n_oxy, double *h_co2,
                               int64_t cap_co2, int64_t *n_co2) {
    if (!S || !S->ran || sample < 0 || sample >= S->B) { octa::set_error("octa_sim_fields: bad arguments"); return -2; }
    OCTA_HIP_CHECK(hipSetDevice(S->ctx->device));
    const SampleScalars &sc = S->h_sc[sample];
    if (n_oxy) *n_oxy = sc.n_oxy;
    if (n_co2) *n_co2 = sc.n_co2;
    if (h_oxy) {
        int64_t n = sc.n_oxy < cap_oxy ? sc.n_oxy : cap_oxy;
        OCTA_HIP_CHECK(hipMemcpy(h_oxy, S->P.oxy + (size_t)sample * OCAP * 3, sizeof(double) * 3 * n, hipMemcpyDeviceToHost));
    }
    if (h_co2) {
        int64_t n = sc.n_co2 < cap_co2 ? sc.n_co2 : cap_co2;
        OCTA_HIP_CHECK(hipMemcpy(h_co2, S->P.co2 + (size_t)sample * CCAP * 3, sizeof(double) * 3 * n, hipMemcpyDeviceToHost));
    }
    return 0;
}

#if !OCTA_SIM_LARGE
namespace {
__global__ void __launch_bounds__(SIM_THREADS)
sim_kat_kd_kernel(const double *pts, int n, const unsigned char *need, idx_t *out_idx, idx_t *out_rank, float *xy, double zlo, double zhi) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    Blk b = {(int)threadIdx.x, (int)blockDim.x, smem};
    kd_build(b, pts, n, out_idx, out_rank, xy, zlo, zhi, nullptr, need);
}
}  // namespace

extern "C" int octa_sim_kat_kd_order(octa_ctx *ctx, const double *h_pts, int64_t n, const uint8_t *h_need, int32_t *h_indices) {
    if (!ctx || !h_pts || !h_indices || n < 0 || n > OCAP) { octa::set_error("octa_sim_kat_kd_order: bad arguments"); return -2; }
    if (n == 0) return 0;
    OCTA_HIP_CHECK(hipSetDevice(ctx->device));
    double *d_pts = nullptr;
    unsigned char *d_need = nullptr;
    idx_t *d_idx = nullptr;
    float *d_xy = nullptr;
    double zlo = h_pts[2], zhi = h_pts[2];
    for (int64_t i = 1; i < n; i++) { zlo = h_pts[3 * i + 2] < zlo ? h_pts[3 * i + 2] : zlo; zhi = h_pts[3 * i + 2] > zhi ? h_pts[3 * i + 2] : zhi; }
    OCTA_HIP_CHECK(hipMalloc(&d_xy, sizeof(float) * 2 * n));
    OCTA_HIP_CHECK(hipMalloc(&d_pts, sizeof(double) * 3 * n));
    OCTA_HIP_CHECK(hipMalloc(&d_idx, sizeof(idx_t) * 2 * n));
    OCTA_HIP_CHECK(hipMemcpy(d_pts, h_pts, sizeof(double) * 3 * n, hipMemcpyHostToDevice));
    if (h_need) {
        OCTA_HIP_CHECK(hipMalloc(&d_need, n));
        OCTA_HIP_CHECK(hipMemcpy(d_need, h_need, n, hipMemcpyHostToDevice));
    }
    OCTA_HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void *>(sim_kat_kd_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, (int)SIM_LDS));
    hipLaunchKernelGGL(sim_kat_kd_kernel, dim3(1), dim3(SIM_THREADS), SIM_LDS, 0, d_pts, (int)n, d_need, d_idx, d_idx + n, d_xy, zlo, zhi);
    OCTA_HIP_CHECK(hipGetLastError());
    std::vector<idx_t> h(n);
    OCTA_HIP_CHECK(hipMemcpy(h.data(), d_idx, sizeof(idx_t) * n, hipMemcpyDeviceToHost));
    for (int64_t i = 0; i < n; i++) h_indices[i] = (int32_t)h[i];
    (void)hipFree(d_pts); (void)hipFree(d_idx); (void)hipFree(d_xy); if (d_need) (void)hipFree(d_need);
    return 0;
}

#endif

extern "C" int octa_sim_service_stats(OCTA_SIM_T *S, double *h_out4) {
    if (!S || !S->ran || !h_out4) { octa::set_error("octa_sim_service_stats: run the simulation first"); return -2; }
    h_out4[0] = (double)S->diag_tickets; h_out4[1] = S->diag_max_gap_ms; h_out4[2] = (double)S->diag_relaunches; h_out4[3] = (double)S->diag_parked;
    h_out4[4] = S->diag_max_bif_ms;
    return 0;
}

extern "C" int octa_sim_geometry(int num_cus, int *h_out4) {
    if (!h_out4) { octa::set_error("octa_sim_geometry: null output"); return -2; }
    h_out4[0] = SIM_THREADS; h_out4[1] = SIM_WG_PER_CU; h_out4[2] = (int)SIM_LDS; h_out4[3] = SIM_WG_PER_CU * (num_cus > 0 ? num_cus : 256);
    return 0;
}

extern "C" int octa_sim_spans(OCTA_SIM_T *S, int64_t *h_spans) {
    if (!S || !S->ran || !h_spans) { octa::set_error("octa_sim_spans: run the simulation first"); return -2; }
    for (int s = 0; s < S->B; s++) { h_spans[2 * s] = S->h_sc[s].t_begin; h_spans[2 * s + 1] = S->h_sc[s].t_end; }
    return 0;
}

extern "C" int octa_sim_timing(OCTA_SIM_T *S, double *h_out8) {
    if (!S || !S->ran || !h_out8) { octa::set_error("octa_sim_timing: run the simulation first"); return -2; }
    h_out8[0] = S->ms_a; h_out8[1] = (double)S->n_a; h_out8[2] = S->ms_b; h_out8[3] = (double)S->n_b;
    h_out8[4] = S->ms_total; h_out8[5] = S->ms_host_bif; h_out8[6] = (double)S->n_bif_req; h_out8[7] = (double)S->bytes;
    return 0;
}
