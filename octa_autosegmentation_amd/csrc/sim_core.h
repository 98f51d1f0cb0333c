// sim_core.h -- the space-colonisation vessel-graph simulator as per-sample, block-cooperative
// phases for gfx950 (one 256-thread workgroup advances one sample; B samples advance in lock-step,
// one launch per phase group). The same source compiles as plain host C++ (one "thread") for
// tests/native/sim_core_host.cpp so that the phase logic can be checked against the oracle on a
// machine without a GPU -- test infrastructure, never a product fallback.
//
// Reference semantics restated here (file:line under the reference tree):
//   greenhouse.py:319-341  sample_oxygen_sinks      -> phase_sample
//   greenhouse.py:343-366  assign_attraction_points -> phase_assign (nearest ACTIVE node within delta,
//                                                      dict order = order of first hit)
//   greenhouse.py:157-307  grow_vessels             -> phase_pre (per-node geometry, parallel) +
//                                                      phase_seq (ordered pass: RNG draws, node creation,
//                                                      Murray propagation arterial_tree.py:174-184)
//   greenhouse.py:98-112   O2 -> CO2 conversion     -> phase_satisfy_art (cKDTree result order + CPython
//                                                      set iteration order are part of the result)
//   greenhouse.py:120-123  CO2 removal              -> phase_satisfy_ven
// Parameters per iteration (greenhouse.py:139-155 expansion, :34-51 mode switch) do not depend on the
// data, so the host tabulates them (IterParams).
//
// Data layout in HBM (per sample, struct-of-arrays, fixed capacities; see SimArrays): node positions
// xyz-interleaved doubles, radii, kappa, parent/child ids, active flags; ordered O2 / CO2 lists; the
// pre-generated candidate stream and Python-uniform stream; scratch for assignment / grouping.
// All arithmetic is IEEE double, -ffp-contract=off, fma() explicit where the reference goes through
// OpenBLAS ddot; pow through gpow (bit-exact glibc pow).
#pragma once
#include <cmath>
#include <cstdint>
#include <cstring>
#include <algorithm>
#include "gpow.h"
#include "glibc_trig.h"

// Two builds of this header (round 3): the default one for the 3 x 3 mm^2 configurations -- every per-sample table of a phase fits
// 78 KiB of LDS, indices are 16-bit, packed words carry 14-bit indices -- and OCTA_SIM_LARGE for wide fields of view (the reference's
// notebook: 12 x 12 mm^2, N = 8000, 400 + 500 iterations, ~16 x the nodes and sinks): the same phases with 32-bit indices, 18-bit
// packed indices, 64-bit kd elements, and the "LDS" user area of a workgroup placed in HBM / L2 scratch (Blk::umem). The large build
// trades speed for capacity; it exists so that the path is COMPLETE for the reference's configurations.
#ifndef OCTA_SIM_LARGE
#define OCTA_SIM_LARGE 0
#endif
#if OCTA_SIM_LARGE
#define OCTA_SIMK octa_simk_large
#else
#define OCTA_SIMK octa_simk
#endif
namespace OCTA_SIMK {

#if OCTA_SIM_LARGE
typedef unsigned int idx_t;            // point / node / position index of the per-phase tables
typedef unsigned long long kdw_t;      // kd element: quantised key << IDX_BITS | point index
constexpr int IDX_BITS = 18;           // index bits of packed words (kd elements, hit pairs, sorted attractor keys)
constexpr int GROUP_BITS = 17;         // group index bits of child_group tags
constexpr int NCAP = 1 << 17;          // nodes per forest
constexpr int OCAP = 1 << 18;          // live O2 sinks
constexpr int CCAP = OCAP;
constexpr int GCAP = 1 << 17;          // nodes with attractors per growth pass
constexpr int SORTCAP = OCAP;          // attractor keys of one assignment
constexpr int PCAP = 1 << 21;          // (new node, sink) hit pairs per iteration: the FIRST iteration of a mode runs with the mode's raw radii
                                       // (not yet divided by param_scale, greenhouse.py:84-85), so every new node of that iteration "reaches"
                                       // a disc of eps_k = 0.0675 of the whole field -- 2 x 10^5 pairs at the notebook's mode switch
constexpr int SETCAP = 1 << 20;        // slots of the emulated CPython set (a table never exceeds 8 x its entries; half is resize staging)
constexpr int PYCAP = 1 << 20;         // pre-generated random.uniform draws per sample
constexpr int ACCCAP = 8192;           // accepted sinks per iteration
constexpr int GRID_MAX = 448;          // uniform-grid cells per axis
constexpr int SIM_USER_BYTES = 12 << 20;  // per-workgroup table area (HBM scratch in this build): the largest tenant is the pair sort (4 B x PCAP)
constexpr int LSET_PAIRS = 1 << 14, LSET_CAP = 1 << 18;   // the "LDS" set replay (wave-parallel resizes) takes every iteration of this build
#else
typedef unsigned short idx_t;
typedef unsigned int kdw_t;
constexpr int IDX_BITS = 14;
constexpr int GROUP_BITS = 13;
constexpr int NCAP = 14336;    // nodes per forest
constexpr int OCAP = 13312;    // live O2 sinks (the LDS-resident kd elements bound this)
constexpr int CCAP = OCAP;     // live CO2 sources (8192 until a full-length seed -- 953121 -- peaked at 8353; every scratch array the CO2 list
                               // passes through is sized for OCAP points)
constexpr int GCAP = 8192;     // nodes with attractors per growth pass
constexpr int SORTCAP = 16384; // keys per block sort
constexpr int PCAP = 16384;    // (new node, sink) hit pairs per iteration
constexpr int SETCAP = 16384;  // slots of the emulated CPython set
constexpr int PYCAP = 32768;   // pre-generated random.uniform draws per sample
constexpr int ACCCAP = 2048;   // accepted sinks per iteration
constexpr int GRID_MAX = 112;   // uniform-grid cells per axis (x, y); the thin z extent is not binned (cell ends int[112^2 + 1] + ids u16[GRID_N] = 77 KiB)
constexpr int SIM_USER_BYTES = 78 * 1024;   // per-workgroup table area: LDS
constexpr int LSET_PAIRS = 512, LSET_CAP = 4096;     // <= 512 keys: the table never exceeds 2048 slots (+ as many for the resize staging)
#endif
static_assert(OCAP <= (1 << IDX_BITS) && GCAP <= (1 << GROUP_BITS) && NCAP <= (1 << 30), "index widths");
constexpr unsigned IDX_MASK = (1u << IDX_BITS) - 1u;
constexpr idx_t IDX_NONE = (idx_t)~(idx_t)0;
constexpr int MAXKEPT = 256;   // attractors shipped with one bifurcation request
constexpr int NCANDCAP = 8192; // candidates per iteration
constexpr int TILE = 1024;     // points per LDS tile in brute-force queries
constexpr int KD_RANGES = OCAP / 17 + 1;  // ranges per kd level (a range that is split further holds at least 17 points)
constexpr int KD_TAB_BYTES_PER_RANGE = 4 * (int)sizeof(idx_t) + 1;   // range start / end of two levels + split dimension
constexpr int KD_MAILBOX_OFF = OCAP * (int)sizeof(kdw_t) + ((KD_RANGES * KD_TAB_BYTES_PER_RANGE + 15) / 16) * 16;   // offset (after user()) of the swap mailbox / box table
constexpr int KD_TEAM_MIN = 192;   // ranges at least this long get a whole wave, shorter ones 16 lanes
#ifndef OCTA_SIM_THREADS
#define OCTA_SIM_THREADS 256
#endif
constexpr int SIM_THREADS_PER_WG = OCTA_SIM_THREADS;           // threads per simulator workgroup (build-time choice, see sim.hip)
constexpr int KD_WAVES = SIM_THREADS_PER_WG / 64;              // waves per workgroup
constexpr int KD_MAILBOX_BYTES = (OCAP / 2 + 64) * (int)sizeof(idx_t);   // one slot per possible swap: ranges are disjoint
constexpr int SIM_LDS_BYTES = SIM_USER_BYTES + 2048;   // default build: dynamic LDS of the simulator kernels = half a CU's 160 KiB, so TWO workgroups (two samples) share a CU

enum ErrBits { ERR_NODE_CAP = 1, ERR_OXY_CAP = 2, ERR_CO2_CAP = 4, ERR_GROUP_CAP = 8, ERR_PAIR_CAP = 16, ERR_SET_CAP = 32,
               ERR_PY_CAP = 64, ERR_KEPT_CAP = 128, ERR_REQ_CAP = 256, ERR_ACC_CAP = 512, ERR_MISSING_BIF = 1024 };

struct IterParams {
    int t, first_mode, N, pad;
    double eps_n, eps_s, eps_k, delta_art, delta_ven, d, gamma_art, gamma_ven, phi, omega, kappa;
};

struct SimConst {
    double ps, r, rotation_radius, fc0, fc1, sx, sy, sz;
    double gs;          // SimulationSpace.geometry_size: 76, or the largest dimension of the geometry file's mask
    int n_iter, n_max;  // iterations, max candidates per iteration (stride of the candidate stream)
    int fixed;          // a geometry file gives the mask (simulation_space.py:29-34): `mask` [gshape[0]][gshape[1]][gshape[2]] bytes
    int gshape[3];
    const unsigned char *mask;
};

// growth record of one node with attractors (one per group, dict order)
struct __attribute__((aligned(8))) Rec {
    double newpos[3];
    double thr;       // (dist_to_center / (2 FAZ_radius)) ** 5
    double r1_used;   // inter nodes: child radius the speculation used
    int node;
    int req;          // bifurcation request id or -1
    unsigned char type;     // 0 none, 1 leaf that grows, 3 inter node (grow says whether it sprouts)
    unsigned char draw;     // consumes one random.uniform
    unsigned char ang_gt90; // angle(vector_to_center, avg_xy) > 90
    unsigned char grow;     // inter nodes: reaches the draw with the radius used
    unsigned int child;     // inter nodes: the (only) child
};

// one marked node of a deferred Murray flush (round 6: murray_flush_prepare_wave -> murray_flush_rounds_wave)
struct __attribute__((aligned(8))) FlushRec { double kk, acc; int node, pend, s0, s1; };

struct __attribute__((aligned(8))) BifRequest {
    int sample, n;
    double pos[3];
    double r, kappa, d;
    double atts[MAXKEPT * 3];
};

struct SampleScalars {
    double faz_radius;
    int n_nodes[2];
    int n_oxy, n_co2;
    int py_pos, py_cap;
    int err;
    int new_begin[2], new_end[2];
    int n_groups[2];
    int n_sorted[2];
    int n_grow[2];
    int pass_tag[2];
    int pass_counter;
    long murray_steps;
    long flush_rounds;     // rounds of the deferred Murray evaluation (sum over passes)
    long murray_deferred;  // radii recomputed by it (the rest of murray_steps are eager steps)
    long n_bif;
    long respec;
    long prof[16];  // accumulated 100 MHz ticks per phase (thread 0), see sim.hip
    long kdprof[8]; // kd_build breakdown: bbox, dim, gather, nth(wave), nth(thread), next-level, finalize
    // persistent form (sim.hip): a workgroup whose host answer does not arrive in time PARKS -- it records where to resume and
    // leaves the kernel; the host serves it at the kernel boundary and launches again
    int resume_it, resume_stage;   // stage 0: top of iteration resume_it; 1: behind its arterial mailbox
    int parked, finished;
    long t_begin, t_end;           // 100 MHz wall clock when a workgroup first took the sample / last left it
    int fl_pending[2];             // marked nodes of forest f whose Murray radii are still to be evaluated from A.fl_rec (deferred flush, round 6)
};

// pointers to ONE sample's slices
struct SimArrays {
    double *npos[2];   // [NCAP*3]
    double *nrad[2];   // [NCAP]
    double *nkap[2];   // [NCAP]
    int *npar[2], *nch0[2], *nch1[2];  // [NCAP]
    unsigned char *nnch[2], *nact[2];  // [NCAP]
    // forest f's arrays by SELECTION, not by indexing: a run-time index into these pointer pairs made the compiler keep the whole
    // struct in private (scratch) memory -- every use of a base pointer in a phase was a scratch load (716 of them in the
    // persistent kernel) -- while a select keeps the struct in registers
    OCTA_HD double *npos_of(int f) const { return f ? npos[1] : npos[0]; }
    OCTA_HD double *nrad_of(int f) const { return f ? nrad[1] : nrad[0]; }
    OCTA_HD double *nkap_of(int f) const { return f ? nkap[1] : nkap[0]; }
    OCTA_HD int *npar_of(int f) const { return f ? npar[1] : npar[0]; }
    OCTA_HD int *nch0_of(int f) const { return f ? nch0[1] : nch0[0]; }
    OCTA_HD int *nch1_of(int f) const { return f ? nch1[1] : nch1[0]; }
    OCTA_HD unsigned char *nnch_of(int f) const { return f ? nnch[1] : nnch[0]; }
    OCTA_HD unsigned char *nact_of(int f) const { return f ? nact[1] : nact[0]; }
    double *oxy;       // [OCAP*3]
    double *co2;       // [CCAP*3]
    const double *cand;  // [n_max][3] candidate sinks of the CURRENT iteration
    const double *py_u;  // [PYCAP]
    // scratch
    int *nn;             // [OCAP] nearest active node per attractor
    int *act_list;       // [NCAP]
    unsigned *sorted;    // [SORTCAP] attractors grouped by their nearest node (groups in dict order, members ascending)
    int *gnode, *gstart, *gcount;  // [GCAP]
    Rec *rec;            // [GCAP]
    int *glist;          // [GCAP] groups that grow under the speculation, ascending (dict order)
    int *child_group;    // [NCAP] tag<<14 | grow<<13 | group of the inter-node whose first child this node is
    FlushRec *fl_rec;    // [2][MURRAY_FLUSH_LDS] records of the forests' deferred Murray flushes
    idx_t *kd_idx, *kd_rank;  // [OCAP]
    unsigned char *removed;  // [OCAP]
    unsigned char *ven_near; // [OCAP]
    unsigned long long *hashes;  // [OCAP]
    unsigned *pairs;     // [PCAP]
    unsigned long long *set_hash;  // [SETCAP]
    int *set_key;        // [SETCAP]
    double *grid_pts;    // [GRID_N*3] coordinates of the current uniform grid's points, in cell order
    int *tmp_int;        // [OCAP + 2*NCANDCAP] general scratch
    double *tmp_dbl;     // [OCAP*3] general scratch (stable compaction staging)
    SampleScalars *sc;
#ifdef OCTA_SIM_DEBUG_SAT
    int *dbg;            // diagnostic build (tools/repro_sim_race.py): [n_iter][16] digest rows of phase_satisfy_art for this sample
    int dbg_it;
#endif
};

// ------------------------------------------------------------------ execution abstraction
// A value every lane of the wave holds alike (read from one address after a barrier, a block total ...): telling the compiler keeps it
// -- and everything derived from it: loop bounds, base pointers, per-iteration parameters -- in scalar registers. Round 3 measured
// -10 % per-sample device time from the sample index alone (its ~40 array base pointers were 80 vector registers and the main source
// of register spills). Used for values that come out of the LDS or out of registers. (Round 3 blamed the irreproducible 512-sample
// batches it saw with uniform annotations on scalar loads of the sample's counters; round 4 found no scalar load of mutable data in the
// ISA and traced the events to a barrier without its LDS wait, see octa_block_sync below.)
#if defined(__HIP_DEVICE_COMPILE__)
#define OCTA_UNI(x) __builtin_amdgcn_readfirstlane(x)
#else
#define OCTA_UNI(x) (x)
#endif

// Workgroup barrier of the simulator kernels: __syncthreads() PRECEDED BY an explicit s_waitcnt vmcnt(0) lgkmcnt(0).
//
// Why (round 4; DESIGN.md 4.1 "The barrier the compiler left without its LDS wait"): hipcc's __syncthreads() is
//     fence release (workgroup); s_barrier; fence acquire (workgroup)
// and the release should become s_waitcnt lgkmcnt(0) in front of the barrier. hipcc 7.2 DROPS that wait when the barrier stands at a
// loop header and the pending LDS stores arrive over the back edge only -- blk_sort_u32's stage loop: a wave's last exchange of one
// stage could still be in the LDS queue when another wave read the key in the next stage; with two workgroups per CU about one
// 512-sample launch in six lost a sink from the sorted pair list. Measured with tools/repro_sim_race.py on the shipped sources
// (30 208 sample runs each): lgkmcnt(0) only: 0 events; vmcnt(0) only: 127; neither (plain __syncthreads(), FLAT-addressed tables):
// 26-38 per 50-76 k. The vmcnt half is insurance for the hand-overs through HBM scratch (LLVM leaves global stores pending at
// workgroup scope by design: "the L1 keeps all memory operations in order for wavefronts in the same work-group"; nothing measured
// here contradicts that) and costs nothing measurable. While the tables were FLAT-addressed, vmcnt(0) alone cured the events --
// FLAT stores to the LDS count in vmcnt too -- which round 4 first misread as a global-memory ordering problem.
// tools/isa/check_barrier_cfg.py finds barriers reachable with LDS stores pending in a listing. OCTA_SIM_SYNC_DRAIN is the experiment
// knob (0: plain __syncthreads(); bit 0: the wait; bit 1: buffer_inv sc0 behind the barrier; bit 2: agent-scope fences instead;
// bit 3: buffer_inv sc1 behind the barrier); OCTA_SIM_SYNC_NO_LGKM / OCTA_SIM_SYNC_NO_VM keep one half of the wait only.
#ifndef OCTA_SIM_SYNC_DRAIN
#define OCTA_SIM_SYNC_DRAIN 1
#endif
#if defined(__HIPCC__)
__device__ __forceinline__ void octa_block_sync() {
#if defined(__HIP_DEVICE_COMPILE__)
#if OCTA_SIM_SYNC_DRAIN & 4
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
    __builtin_amdgcn_s_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
    return;
#endif
#if OCTA_SIM_SYNC_DRAIN & 1
#if defined(OCTA_SIM_SYNC_NO_LGKM)
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#elif defined(OCTA_SIM_SYNC_NO_VM)
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
#else
    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
#endif
#endif
    __syncthreads();
#if OCTA_SIM_SYNC_DRAIN & 2
    asm volatile("buffer_inv sc0" ::: "memory");
#endif
#if OCTA_SIM_SYNC_DRAIN & 8
    asm volatile("buffer_inv sc1" ::: "memory");
#endif
#endif
}
#endif

#ifndef OCTA_SIM_DS_MASK
#define OCTA_SIM_DS_MASK 0
#endif
// Measurement knob (tools/sim_traffic.py): run an idempotent part of an iteration TWICE, so that the difference of the kernel's
// FETCH_SIZE / WRITE_SIZE counters against the plain build is that part's share of the L2-miss traffic. Bits: 1 kd order, 2 every
// uniform grid's build, 4 assignment (both forests), 8 the candidate tests' queries, 32 speculation (both forests).
#ifndef OCTA_SIM_DUP
#define OCTA_SIM_DUP 0
#endif
struct Blk {
    int tid, nth;
    unsigned char *smem;  // LDS (device) or heap (host emulation); first 2 KiB reserved for collectives
    unsigned char *umem = nullptr;   // the per-phase table area when it is NOT the LDS behind the collectives (large build: HBM scratch)
    mutable unsigned scan_calls = 0; // blk_scan's slot toggle (per thread; all threads of a workgroup make the same calls)
    OCTA_HD inline void sync() const {
#if defined(__HIP_DEVICE_COMPILE__)
        octa_block_sync();
#endif
    }
    OCTA_HD inline int *coll() const { return reinterpret_cast<int *>(smem); }
    // The table area of a phase: the LDS behind the collectives in the default build (ds_* instructions), the workgroup's HBM scratch in
    // the wide-field build. Until round 4 the default build chose between the two at run time (`umem ? umem : smem + 2048`), so the
    // compiler could not prove the pointer to be LDS and every table access was a FLAT instruction (1027 of them: 64-bit addresses,
    // both wait counters): 513 -> 492 ms per sample. -DOCTA_SIM_FLAT_USER restores that form.
#if !OCTA_SIM_LARGE && !defined(OCTA_SIM_FLAT_USER)
    OCTA_HD inline unsigned char *user() const { return smem + 2048; }
#else
    OCTA_HD inline unsigned char *user() const { return umem ? umem : smem + 2048; }
#endif
    // The same area for ONE tenant (TENANT: 1 kd order, 2 uniform grid, 4 greedy acceptance, 8 ordered pass, 16 compaction tile, 32 pair
    // sort, 64 set replay): LDS at compile time for the tenants named in OCTA_SIM_DS_MASK (only meaningful with -DOCTA_SIM_FLAT_USER).
    // Round 4 used it to find WHICH tenant's ds_* addressing brought the irreproducibility back: 32, the pair sort.
    template <int TENANT>
    OCTA_HD inline unsigned char *user_of() const {
#if !OCTA_SIM_LARGE
        if (OCTA_SIM_DS_MASK & TENANT) return smem + 2048;
#endif
        return user();
    }
};

// Inclusive prefix sum over the 64 lanes of a wave with DPP adds: four shifts inside the rows of 16 lanes, then lane 15 of rows 0 / 2 onto
// rows 1 / 3 and lane 31 onto rows 2 and 3. Six VALU instructions; `__shfl_up` is six `ds_bpermute` round trips through the LDS unit.
#if defined(__HIP_DEVICE_COMPILE__)
__device__ __forceinline__ int wave_scan_incl(int v) {
    v += __builtin_amdgcn_update_dpp(0, v, 0x111, 0xf, 0xf, true);     // row_shr:1
    v += __builtin_amdgcn_update_dpp(0, v, 0x112, 0xf, 0xf, true);     // row_shr:2
    v += __builtin_amdgcn_update_dpp(0, v, 0x114, 0xf, 0xf, true);     // row_shr:4
    v += __builtin_amdgcn_update_dpp(0, v, 0x118, 0xf, 0xf, true);     // row_shr:8
    v += __builtin_amdgcn_update_dpp(0, v, 0x142, 0xa, 0xf, false);    // row_bcast:15 -> rows 1 and 3
    v += __builtin_amdgcn_update_dpp(0, v, 0x143, 0xc, 0xf, false);    // row_bcast:31 -> rows 2 and 3
    return v;
}
#endif

// exclusive scan of one int per thread; returns block total. Contains ONE block sync (round 4; three until then): the wave totals go
// into one of two alternating slots of the collectives area (every thread makes the same calls, so the slots alternate alike), and
// every thread sums the totals of the waves in front of it itself. A wave can only reach the call after next -- which writes the
// same slot again -- through the next call's barrier, i.e. after every wave has read this call's totals.
OCTA_HD inline int blk_scan(const Blk &b, int v, int *excl) {
#if defined(__HIP_DEVICE_COMPILE__)
    const int lane = b.tid & 63, wv = b.tid >> 6, nw = (b.nth + 63) >> 6;
    const int inc = wave_scan_incl(v);
    int *sh = b.coll() + ((b.scan_calls++ & 1u) ? 16 : 0);
    if (lane == 63 || b.tid == b.nth - 1) sh[wv] = inc;
    b.sync();
    int base = 0, tot = 0;
    for (int k = 0; k < nw; k++) { const int t = sh[k]; base += k < wv ? t : 0; tot += t; }
    *excl = base + inc - v;
    return OCTA_UNI(tot);
#else
    (void)b;
    *excl = 0;
    return v;
#endif
}

// Ordered compaction of the indices 0 .. n-1 that satisfy pred, with coalesced access: every wave takes a contiguous segment, its lanes 64
// consecutive indices per step (ballot + population counts give the positions), ONE block scan of the waves' counts in between. pred(i)
// is evaluated twice (count, emit); emit(i, position). Returns the number of kept indices. The chunk-per-thread form this replaces
// (thread t walks indices t * chunk ...) made every lane of a load touch its own cache line. Host build: one thread, a plain loop.
template <class Pred, class Emit>
OCTA_HD inline int ordered_compact(const Blk &b, int n, Pred pred, Emit emit) {
#if defined(__HIP_DEVICE_COMPILE__)
    const int lane = b.tid & 63, wv = b.tid >> 6, nw = (b.nth + 63) >> 6;
    const int seg = ((n + nw * 64 - 1) / (nw * 64)) * 64;
    const int s0 = wv * seg < n ? wv * seg : n, s1 = s0 + seg < n ? s0 + seg : n;
    int c = 0;
    for (int i0 = s0; i0 < s1; i0 += 64) c += (int)__popcll(__ballot(i0 + lane < s1 && pred(i0 + lane)));
    int ex;
    const int tot = blk_scan(b, lane == 0 ? c : 0, &ex);
    int at = __builtin_amdgcn_readfirstlane(ex);
    for (int i0 = s0; i0 < s1; i0 += 64) {
        const bool keep = i0 + lane < s1 && pred(i0 + lane);
        const unsigned long long m = __ballot(keep);
        if (keep) emit(i0 + lane, at + (int)__popcll(m & ((1ull << lane) - 1ull)));
        at += (int)__popcll(m);
    }
    return tot;
#else
    (void)b;
    int at = 0;
    for (int i = 0; i < n; i++) if (pred(i)) { emit(i, at); at++; }
    return at;
#endif
}

OCTA_HD inline void atomic_min_int(int *p, int v) {
#if defined(__HIP_DEVICE_COMPILE__)
    atomicMin(p, v);
#else
    if (v < *p) *p = v;
#endif
}
OCTA_HD inline int atomic_min_ret_int(int *p, int v) {
#if defined(__HIP_DEVICE_COMPILE__)
    return atomicMin(p, v);
#else
    int o = *p; if (v < o) *p = v; return o;
#endif
}
OCTA_HD inline int atomic_add_int(int *p, int v) {
#if defined(__HIP_DEVICE_COMPILE__)
    return atomicAdd(p, v);
#else
    int o = *p; *p = o + v; return o;
#endif
}
OCTA_HD inline void atomic_or_int(int *p, int v) {
#if defined(__HIP_DEVICE_COMPILE__)
    atomicOr(p, v);
#else
    *p |= v;
#endif
}
OCTA_HD inline void atomic_and_int(int *p, int v) {
#if defined(__HIP_DEVICE_COMPILE__)
    atomicAnd(p, v);
#else
    *p &= v;
#endif
}


OCTA_HD inline unsigned long long dbl_sortable(double v) {
    unsigned long long u = octa_gpow::asu64(v);
    return (u >> 63) ? ~u : (u | 0x8000000000000000ULL);
}
OCTA_HD inline double dbl_unsortable(unsigned long long e) {
    unsigned long long u = (e >> 63) ? (e & 0x7fffffffffffffffULL) : ~e;
    return octa_gpow::asf64(u);
}
OCTA_HD inline void atomic_max_u64(unsigned long long *p, unsigned long long v) {
#if defined(__HIP_DEVICE_COMPILE__)
    atomicMax(p, v);
#else
    if (v > *p) *p = v;
#endif
}
OCTA_HD inline void atomic_min_u64(unsigned long long *p, unsigned long long v) {
#if defined(__HIP_DEVICE_COMPILE__)
    atomicMin(p, v);
#else
    if (v < *p) *p = v;
#endif
}

// ascending sort of n_pow2 32-bit keys that live in LDS
OCTA_HD inline void blk_sort_u32(const Blk &b, unsigned *keys, int n_pow2) {
#if defined(__HIP_DEVICE_COMPILE__)
    for (int k = 2; k <= n_pow2; k <<= 1) {
        for (int j = k >> 1; j > 0; j >>= 1) {
            b.sync();
            for (int i = b.tid; i < n_pow2; i += b.nth) {
                int l = i ^ j;
                if (l > i) {
                    unsigned a = keys[i], c = keys[l];
                    bool up = (i & k) == 0;
                    if ((a > c) == up) { keys[i] = c; keys[l] = a; }
                }
            }
        }
    }
    b.sync();
#else
    (void)b;
    std::sort(keys, keys + n_pow2);
#endif
}

// ------------------------------------------------------------------ vector helpers (as in the oracle)
struct V3 { double x, y, z; };
OCTA_HD inline V3 v3(double x, double y, double z) { V3 v = {x, y, z}; return v; }
OCTA_HD inline V3 ld3(const double *p) { return v3(p[0], p[1], p[2]); }
OCTA_HD inline void st3(double *p, V3 v) { p[0] = v.x; p[1] = v.y; p[2] = v.z; }
OCTA_HD inline V3 sub(V3 a, V3 b) { return v3(a.x - b.x, a.y - b.y, a.z - b.z); }
OCTA_HD inline V3 add(V3 a, V3 b) { return v3(a.x + b.x, a.y + b.y, a.z + b.z); }
OCTA_HD inline V3 mul(V3 a, double s) { return v3(a.x * s, a.y * s, a.z * s); }
OCTA_HD inline V3 divs(V3 a, double s) { return v3(a.x / s, a.y / s, a.z / s); }
OCTA_HD inline double norm3(V3 v) { return sqrt(fma(v.z, v.z, fma(v.y, v.y, v.x * v.x))); }  // OpenBLAS ddot form
OCTA_HD inline double norm2(double a, double c) { return sqrt(fma(c, c, a * a)); }
OCTA_HD inline double dot3_blas(V3 a, V3 c) { return fma(a.z, c.z, fma(a.y, c.y, a.x * c.x)); }
OCTA_HD inline double dot2_blas(double a0, double a1, double b0, double b1) { return fma(a1, b1, a0 * b0); }
OCTA_HD inline V3 unit(V3 v) { return divs(v, norm3(v)); }
OCTA_HD inline double rownorm(V3 v) { return sqrt((v.x * v.x + v.y * v.y) + v.z * v.z); }
OCTA_HD inline V3 cross(V3 a, V3 c) { return v3(a.y * c.z - a.z * c.y, a.z * c.x - a.x * c.z, a.x * c.y - a.y * c.x); }
OCTA_HD inline double sqdist(V3 a, V3 c) {
    double dx = a.x - c.x, dy = a.y - c.y, dz = a.z - c.z;
    return (dx * dx + dy * dy) + dz * dz;
}
OCTA_HD inline double clip1(double c) { return fmin(fmax(c, -1.0), 1.0); }
OCTA_HD inline double rad2deg() { return 180.0 / 3.14159265358979323846; }
OCTA_HD inline double deg2rad() { return 3.14159265358979323846 / 180.0; }
OCTA_HD inline double angle_uv(V3 u, double nu, V3 v) {
    double c = ((u.x * v.x + u.y * v.y) + u.z * v.z) / nu / rownorm(v);
    return acos(clip1(c)) * rad2deg();
}
OCTA_HD inline double angle2(double u0, double u1, double v0, double v1) {
    double c = dot2_blas(u0, u1, v0, v1) / norm2(u0, u1) / norm2(v0, v1);
    return acos(clip1(c)) * rad2deg();
}
OCTA_HD inline double oxygen_distance(double rad, double ps) {  // greenhouse.py:309-317
    const double c_oxygen = 203.9e-3, kap = 0.02 * c_oxygen, r0 = 3.5e-3;
    double q = rad * ps / r0;
    double c1 = kap * q * exp(1 - q);
    return c1 * 6 / ps;
}

// ------------------------------------------------------------------ CPython hashing (float, 3-tuple)
OCTA_HD inline unsigned long long py_hash_double(double v) {
    const unsigned long long MOD = (1ULL << 61) - 1;
    int e;
    double m = frexp(v, &e);
    int sign = 1;
    if (m < 0) { sign = -1; m = -m; }
    unsigned long long x = 0;
    while (m != 0.0) {
        x = ((x << 28) & MOD) | x >> (61 - 28);
        m *= 268435456.0;
        e -= 28;
        unsigned long long y = (unsigned long long)m;
        m -= (double)y;
        x += y;
        if (x >= MOD) x -= MOD;
    }
    e = e >= 0 ? e % 61 : 61 - 1 - ((-1 - e) % 61);
    x = ((x << e) & MOD) | x >> (61 - e);
    if (sign < 0) x = 0ULL - x;
    if (x == ~0ULL) x = ~0ULL - 1;
    return x;
}
OCTA_HD inline unsigned long long py_hash_tuple3(V3 t) {
    const unsigned long long P1 = 11400714785074694791ULL, P2 = 14029467366897019727ULL, P5 = 2870177450012600261ULL;
    unsigned long long acc = P5;
    double c[3] = {t.x, t.y, t.z};
    for (int i = 0; i < 3; i++) {
        acc += py_hash_double(c[i]) * P2;
        acc = (acc << 31) | (acc >> 33);
        acc *= P1;
    }
    acc += 3ULL ^ (P5 ^ 3527539ULL);
    if (acc == ~0ULL) return 1546275796ULL;
    return acc;
}

// CPython 3.10 set (add only): table arrays in LDS or global scratch. Run by ONE thread (lane 0, nl 1) or by the nl = 64 lanes of
// a wave that all execute the same insertions (same addresses, same values) and share the bulk loops of a resize -- clearing the
// new table, scanning the old one for its entries, moving the new table down -- which dominate the replay for a few hundred keys.
struct PySetView {
    unsigned long long *hash;
    int *key;
    int mask, fill, used;
    int *err;
    int cap;  // slots available in hash[] / key[]; the upper half is the resize staging area
    int lane = 0, nl = 1;
};
OCTA_HD inline void pyset_team_sync(const PySetView &s) {
#if defined(__HIP_DEVICE_COMPILE__)
    if (s.nl > 1) __builtin_amdgcn_wave_barrier();
#else
    (void)s;
#endif
}
OCTA_HD inline void pyset_init(PySetView &s) {
    for (int i = 0; i < 8; i++) s.key[i] = -1;
    s.mask = 7; s.fill = 0; s.used = 0;
}
OCTA_HD inline void pyset_insert_clean(unsigned long long *h, int *k, int mask, int key, unsigned long long hash) {
    unsigned long long perturb = hash;
    unsigned long long i = hash & (unsigned long long)mask;
    while (true) {
        unsigned long long e = i;
        int probes = (i + 9 <= (unsigned long long)mask) ? 9 : 0;
        do {
            if (k[e] < 0) { k[e] = key; h[e] = hash; return; }
            e++;
        } while (probes--);
        perturb >>= 5;
        i = (i * 5 + 1 + perturb) & (unsigned long long)mask;
    }
}
// resize into the upper half of the scratch arrays, then move down (cap bounds the table)
OCTA_HD inline void pyset_resize(PySetView &s, int minused) {
    int newsize = 8;
    while (newsize <= minused) newsize <<= 1;
    if (newsize > s.cap / 2) { atomic_or_int(s.err, ERR_SET_CAP); return; }
    unsigned long long *nh = s.hash + s.cap / 2;
    int *nk = s.key + s.cap / 2;
    for (int i = s.lane; i < newsize; i += s.nl) nk[i] = -1;
    pyset_team_sync(s);
    for (int base = 0; base <= s.mask; base += s.nl) {          // entries of the old table in slot order
        const int slot = base + s.lane;
        const bool used = slot <= s.mask && s.key[slot] >= 0;
#if defined(__HIP_DEVICE_COMPILE__)
        unsigned long long m = s.nl > 1 ? __ballot(used) : (used ? 1ull : 0ull);
        while (m) {
            const int j = (int)__ffsll((long long)m) - 1;
            m &= m - 1ull;
            pyset_insert_clean(nh, nk, newsize - 1, s.key[base + j], s.hash[base + j]);
        }
#else
        if (used) pyset_insert_clean(nh, nk, newsize - 1, s.key[slot], s.hash[slot]);
#endif
    }
    pyset_team_sync(s);
    for (int i = s.lane; i < newsize; i += s.nl) { s.key[i] = nk[i]; s.hash[i] = nh[i]; }
    pyset_team_sync(s);
    s.mask = newsize - 1;
    s.fill = s.used;
}
OCTA_HD inline void pyset_add(PySetView &s, int key, unsigned long long hash) {
    unsigned long long perturb = hash;
    unsigned long long i = hash & (unsigned long long)s.mask;
    while (true) {
        unsigned long long e = i;
        int probes = (i + 9 <= (unsigned long long)s.mask) ? 9 : 0;
        do {
            if (s.key[e] < 0) {
                s.fill++; s.used++;
                s.key[e] = key; s.hash[e] = hash;
                if (!OCTA_UNLIKELY(!((long)s.fill * 5 < (long)s.mask * 3))) return;
                pyset_resize(s, s.used > 50000 ? s.used * 2 : s.used * 4);
                return;
            }
            if (s.hash[e] == hash && s.key[e] == key) return;
            e++;
        } while (probes--);
        perturb >>= 5;
        i = (i * 5 + 1 + perturb) & (unsigned long long)s.mask;
    }
}

// ------------------------------------------------------------------ libstdc++ std::nth_element restated
// Elements are (key, idx) pairs ordered by (key, idx) -- scipy's index_compare for one split dimension.
// LDS form (round 3): ONE 32-bit word per element, q << 14 | idx. idx names the point (OCAP <= 2^14); q is the split-dimension
// coordinate quantised to 18 bits over its range's [min, max] -- a monotone map (rounding of x - min, of the product with the
// scale and the truncation are all monotone), so two words whose q differ compare exactly like the coordinates; words with the
// same q (about n / 2^18 of the comparisons) are decided on the doubles themselves, fetched from the point list. Every
// comparison has the exact (key, idx) outcome with 4 B instead of 10 B of LDS per element, and a swap moves one word.
constexpr int KD_IDX_BITS = IDX_BITS;
constexpr kdw_t KD_IDX_MASK = ((kdw_t)1 << KD_IDX_BITS) - 1u;
constexpr kdw_t KD_QMAX = ((kdw_t)1 << (8 * (int)sizeof(kdw_t) - KD_IDX_BITS)) - 1u;     // 18 key bits (default build) / 46 (large build)
static_assert(OCAP <= (1 << KD_IDX_BITS), "kd index bits");
struct KdPair {
    kdw_t *kv;           // packed elements (LDS on the device)
    const double *pts;   // [n][3] coordinates
    int d;               // split dimension of the range being partitioned
};
OCTA_HD inline bool kd_less_w(const KdPair &a, kdw_t x, kdw_t y) {
    if (!OCTA_UNLIKELY(!((x ^ y) >> KD_IDX_BITS))) return x < y;
    const unsigned xi = (unsigned)(x & KD_IDX_MASK), yi = (unsigned)(y & KD_IDX_MASK);
    const double fx = a.pts[3 * xi + a.d], fy = a.pts[3 * yi + a.d];
    if (fx == fy) return xi < yi;
    return fx < fy;
}
OCTA_HD inline bool kd_less(const KdPair &a, int i, int j) { return kd_less_w(a, a.kv[i], a.kv[j]); }
OCTA_HD inline void kd_swap(const KdPair &a, int i, int j) { kdw_t t = a.kv[i]; a.kv[i] = a.kv[j]; a.kv[j] = t; }
OCTA_HD inline kdw_t kd_quant(double x, double mn, double scale) {
    const double t = (x - mn) * scale;
    if (!(t > 0.0)) return 0;
    if (t >= (double)KD_QMAX) return KD_QMAX;
    return (kdw_t)t;
}
OCTA_HD inline void kd_push_heap(const KdPair &a, int first, int hole, int top, kdw_t v) {
    int parent = (hole - 1) / 2;
    while (hole > top && kd_less_w(a, a.kv[first + parent], v)) {
        a.kv[first + hole] = a.kv[first + parent];
        hole = parent;
        parent = (hole - 1) / 2;
    }
    a.kv[first + hole] = v;
}
OCTA_HD inline void kd_adjust_heap(const KdPair &a, int first, int hole, int len, kdw_t v) {
    const int top = hole;
    int second = hole;
    while (second < (len - 1) / 2) {
        second = 2 * (second + 1);
        if (kd_less(a, first + second, first + (second - 1))) second--;
        a.kv[first + hole] = a.kv[first + second];
        hole = second;
    }
    if ((len & 1) == 0 && second == (len - 2) / 2) {
        second = 2 * (second + 1);
        a.kv[first + hole] = a.kv[first + (second - 1)];
        hole = second - 1;
    }
    kd_push_heap(a, first, hole, top, v);
}
OCTA_HD inline void kd_heap_select(const KdPair &a, int first, int middle, int last) {
    int len = middle - first;
    if (len >= 2) {
        int parent = (len - 2) / 2;
        while (true) {
            kdw_t v = a.kv[first + parent];
            kd_adjust_heap(a, first, parent, len, v);
            if (parent == 0) break;
            parent--;
        }
    }
    for (int i = middle; i < last; ++i)
        if (kd_less(a, i, first)) {
            kdw_t v = a.kv[i];
            a.kv[i] = a.kv[first];
            kd_adjust_heap(a, first, 0, middle - first, v);
        }
}
// __insertion_sort of [first, last)
OCTA_HD inline void kd_insertion_sort(const KdPair &a, int first, int last) {
    if (first == last) return;
    for (int i = first + 1; i < last; ++i) {
        const kdw_t v = a.kv[i];
        if (kd_less_w(a, v, a.kv[first])) {
            for (int k = i; k > first; --k) a.kv[k] = a.kv[k - 1];
            a.kv[first] = v;
        } else {
            int lastp = i, next = i - 1;
            while (kd_less_w(a, v, a.kv[next])) {
                a.kv[lastp] = a.kv[next];
                lastp = next; --next;
            }
            a.kv[lastp] = v;
        }
    }
}
// __move_median_to_first(first, first + 1, mid, last - 1)
OCTA_HD inline void kd_median_to_first(const KdPair &a, int first, int last) {
    const int mid = first + (last - first) / 2;
    const int A = first + 1, B = mid, C = last - 1;
    if (kd_less(a, A, B)) {
        if (kd_less(a, B, C)) kd_swap(a, first, B);
        else if (kd_less(a, A, C)) kd_swap(a, first, C);
        else kd_swap(a, first, A);
    } else if (kd_less(a, A, C)) kd_swap(a, first, A);
    else if (kd_less(a, B, C)) kd_swap(a, first, C);
    else kd_swap(a, first, B);
}
OCTA_HD inline void kd_nth_element(const KdPair &a, int first, int nth, int last) {
    if (first == last || nth == last) return;
    int n = last - first, lg = 0;
    while ((n >> (lg + 1)) > 0) lg++;
    int depth = lg * 2;
    while (last - first > 3) {
        if (depth == 0) {
            kd_heap_select(a, first, nth + 1, last);
            kd_swap(a, first, nth);
            return;
        }
        --depth;
        kd_median_to_first(a, first, last);   // __unguarded_partition_pivot
        int lo = first + 1, hi = last;
        while (true) {
            while (kd_less(a, lo, first)) ++lo;
            --hi;
            while (kd_less(a, first, hi)) --hi;
            if (!(lo < hi)) break;
            kd_swap(a, lo, hi);
            ++lo;
        }
        int cut = lo;
        if (cut <= nth) first = cut; else last = cut;
    }
    kd_insertion_sort(a, first, last);
}


#if defined(__HIP_DEVICE_COMPILE__)
// The same algorithm as kd_nth_element executed by a team of TW lanes (a whole wave or a 16-lane
// quarter). Elements are distinct under (key, idx), so one Hoare pass with pivot p is a pure function
// of the array: with S = #{x < p} the cut is first+1+S, and the sequential cursors swap the k-th
// element > p that lies left of the cut (counted from the left) with the k-th element < p right of
// the cut (counted from the right). Every lane takes one contiguous chunk, remembers "x < p" in a bit
// mask, a team scan of the chunk counts gives the cut and every rank, the left side posts its
// positions to an LDS mailbox and the right side performs the swaps. Median-of-three, the <= 3
// element insertion sort and the heap fallback stay with lane 0 of the team.
// mb: u16 mailbox; a range [s, e) owns the slots from (s+1)/2 (ranges of one level are disjoint).
template <int TW, int NW>
__device__ inline void kd_nth_element_team(const KdPair &a, int first, int nth, int last, bool live,
                                           idx_t *mb, int depth_left = -1) {
    const int tl = (int)(threadIdx.x & (TW - 1));
    if (!live || first == last || nth == last) return;
    idx_t *box = mb + ((first + 1) >> 1);
    int n = last - first, lg = 0;
    while ((n >> (lg + 1)) > 0) lg++;
    int depth = depth_left >= 0 ? depth_left : lg * 2;     // a range handed over by kd_nth_element_wave_long keeps its introselect budget
    while (last - first > 3) {
        if (OCTA_UNLIKELY(depth == 0)) {
            if (tl == 0) { kd_heap_select(a, first, nth + 1, last); kd_swap(a, first, nth); }
            __builtin_amdgcn_wave_barrier();
            return;
        }
        --depth;
        // __move_median_to_first(first, first + 1, mid, last - 1): the four words are fetched together (every lane the same addresses: LDS
        // broadcasts), the median is chosen on registers by all lanes alike, lane 0 stores the swap and the pivot never is read back
        // (round 4: lane 0 walking kd_median_to_first was ~five dependent LDS round trips per pass)
        kdw_t pk;
        {
            const int iA = first + 1, iB = first + (last - first) / 2, iC = last - 1;
            const kdw_t x0 = a.kv[first], xa = a.kv[iA], xb = a.kv[iB], xc = a.kv[iC];
            const bool ab = kd_less_w(a, xa, xb), bc = kd_less_w(a, xb, xc), ac = kd_less_w(a, xa, xc);
            const int im = ab ? (bc ? iB : (ac ? iC : iA)) : (ac ? iA : (bc ? iC : iB));
            pk = im == iA ? xa : (im == iB ? xb : xc);
            __builtin_amdgcn_wave_barrier();
            if (tl == 0) { a.kv[first] = pk; a.kv[im] = x0; }
        }
        __builtin_amdgcn_wave_barrier();
        const int base = first + 1, m = last - base;
        const int c = ((m + TW - 1) / TW) | 1;  // odd chunk length: lanes hit distinct LDS banks
        int p0 = base + tl * c;
        if (p0 > last) p0 = last;
        const int p1 = (p0 + c < last) ? p0 + c : last;
        unsigned long long ms[NW];
        int nS = 0;
#pragma unroll
        for (int w = 0; w < NW; w++) {
            const int q0 = p0 + 64 * w;
            int lim = p1 - q0;
            lim = lim > 64 ? 64 : lim;
            unsigned long long bits = 0;
#pragma unroll 4
            for (int i = 0; i < lim; i++) {
                const kdw_t x = a.kv[q0 + i];
                bool lt = x < pk;
                if (OCTA_UNLIKELY(!((x ^ pk) >> KD_IDX_BITS))) lt = kd_less_w(a, x, pk);     // same bucket as the pivot: exact (rare)
                if (lt) bits |= 1ull << i;
            }
            ms[w] = bits;
            nS += __popcll(bits);
        }
        // prefix sum of the chunk counts over the team: DPP adds (a 16-lane team is one DPP row) instead of ds_bpermute steps
        int inc, totS;
        if (TW == 64) {
            inc = wave_scan_incl(nS);
            totS = __builtin_amdgcn_readlane(inc, 63);
        } else {
            static_assert(TW == 64 || TW == 16, "team width");
            inc = nS;
            inc += __builtin_amdgcn_update_dpp(0, inc, 0x111, 0xf, 0xf, true);
            inc += __builtin_amdgcn_update_dpp(0, inc, 0x112, 0xf, 0xf, true);
            inc += __builtin_amdgcn_update_dpp(0, inc, 0x114, 0xf, 0xf, true);
            inc += __builtin_amdgcn_update_dpp(0, inc, 0x118, 0xf, 0xf, true);
            totS = __shfl(inc, TW - 1, TW);
        }
        const int preS = inc - nS;
        const int cut = base + totS;
        // left of the cut: post the positions of the elements > pivot, ranked from the left
        {
            int g = (p0 - base) - preS;
#pragma unroll
            for (int w = 0; w < NW; w++) {
                const int q0 = p0 + 64 * w;
                int lim = ((p1 < cut) ? p1 : cut) - q0;
                if (lim <= 0) continue;
                unsigned long long inv = ~ms[w];
                if (lim < 64) inv &= (1ull << lim) - 1ull;
                while (inv) {
                    int i = (int)__ffsll((long long)inv) - 1;
                    inv &= inv - 1ull;
                    box[g++] = (idx_t)(q0 + i);
                }
            }
        }
        __builtin_amdgcn_wave_barrier();
        // right of the cut: the elements < pivot, ranked from the right, swap with their partners
        {
            int r = totS - preS - nS;
#pragma unroll
            for (int w = NW - 1; w >= 0; w--) {
                const int q0 = p0 + 64 * w;
                if (q0 >= p1) continue;
                unsigned long long sm = ms[w];
                if (cut > q0) { int sh = cut - q0; sm = (sh >= 64) ? 0ull : (sm >> sh) << sh; }
                while (sm) {
                    int i = 63 - __clzll((long long)sm);
                    sm &= ~(1ull << i);
                    kd_swap(a, q0 + i, (int)box[r++]);
                }
            }
        }
        __builtin_amdgcn_wave_barrier();
        if (cut <= nth) first = cut; else last = cut;
    }
    if (tl == 0) kd_insertion_sort(a, first, last);
    __builtin_amdgcn_wave_barrier();
}
// Ranges too long for the bit masks of kd_nth_element_team (more than 64 lanes x 4 words x 64 elements; only the large build has
// them): the same Hoare passes by one wave with the comparisons RECOMPUTED in the posting and the swapping sweep instead of
// remembered; once the active range fits the masks the rest is handed to kd_nth_element_team with the remaining depth budget.
constexpr int KD_MASK_MAX = 16000;      // 64 lanes x (4 x 64 mask bits) = 16384, minus the rounding of the chunk length up to the next odd number
__device__ inline void kd_nth_element_wave_long(const KdPair &a, int first, int nth, int last, idx_t *mb) {
    const int tl = (int)(threadIdx.x & 63);
    if (first == last || nth == last) return;
    idx_t *box = mb + ((first + 1) >> 1);
    int n = last - first, lg = 0;
    while ((n >> (lg + 1)) > 0) lg++;
    int depth = lg * 2;
    while (last - first > KD_MASK_MAX) {
        if (depth == 0) {
            if (tl == 0) { kd_heap_select(a, first, nth + 1, last); kd_swap(a, first, nth); }
            __builtin_amdgcn_wave_barrier();
            return;
        }
        --depth;
        if (tl == 0) kd_median_to_first(a, first, last);
        __builtin_amdgcn_wave_barrier();
        const kdw_t pk = a.kv[first];
        const int base = first + 1, m = last - base;
        const int c = ((m + 63) / 64) | 1;
        int p0 = base + tl * c;
        if (p0 > last) p0 = last;
        const int p1 = (p0 + c < last) ? p0 + c : last;
        int nS = 0;
        for (int i = p0; i < p1; i++) nS += kd_less_w(a, a.kv[i], pk) ? 1 : 0;
        int inc = nS;
#pragma unroll
        for (int d = 1; d < 64; d <<= 1) {
            int u = __shfl_up(inc, d, 64);
            if (tl >= d) inc += u;
        }
        const int totS = __shfl(inc, 63, 64);
        const int preS = inc - nS;
        const int cut = base + totS;
        {   // left of the cut: the elements > pivot, ranked from the left
            int g = (p0 - base) - preS;
            const int lim = p1 < cut ? p1 : cut;
            for (int i = p0; i < lim; i++)
                if (!kd_less_w(a, a.kv[i], pk)) box[g++] = (idx_t)i;
        }
        __builtin_amdgcn_wave_barrier();
        {   // right of the cut: the elements < pivot, ranked from the right, swap with their partners
            int r = totS - preS - nS;
            const int lo = p0 > cut ? p0 : cut;
            for (int i = p1 - 1; i >= lo; i--)
                if (kd_less_w(a, a.kv[i], pk)) kd_swap(a, i, (int)box[r++]);
        }
        __builtin_amdgcn_wave_barrier();
        if (cut <= nth) first = cut; else last = cut;
    }
    kd_nth_element_team<64, 4>(a, first, nth, last, true, mb, depth);
}
#endif

// Per-range bounding boxes as SINGLE-precision outer bounds (max rounded up, min rounded down; 24 B per range in LDS, folded with
// 32-bit LDS atomics on order-preserving encodings). They only have to answer "which dimension has the largest spread" (scipy: the
// first one, strict >): with s_hi = max_up - min_dn >= spread >= s_lo = s_hi - the two roundings, dimension k is the answer when
// s_lo[k] > s_hi[j] for both others; a range whose intervals overlap (two spreads equal to 1e-7 relative: about once per thousand
// builds) is measured exactly by its thread. The quantisation of the keys needs no exact bounds at all (kd_quant is monotone for any
// offset / scale; values outside clamp).
OCTA_HD inline unsigned f32_bits(float v) { unsigned u; memcpy(&u, &v, 4); return u; }
OCTA_HD inline unsigned f32_sortable(float v) {
    unsigned u;
    memcpy(&u, &v, 4);
    return (u >> 31) ? ~u : (u | 0x80000000u);
}
OCTA_HD inline float f32_unsortable(unsigned e) {
    unsigned u = (e >> 31) ? (e & 0x7fffffffu) : ~e;
    float v;
    memcpy(&v, &u, 4);
    return v;
}
OCTA_HD inline float f32_round_up(double x) {
#if defined(__HIP_DEVICE_COMPILE__)
    return __double2float_ru(x);
#else
    float f = (float)x;
    return (double)f < x ? nextafterf(f, INFINITY) : f;
#endif
}
OCTA_HD inline float f32_round_down(double x) {
#if defined(__HIP_DEVICE_COMPILE__)
    return __double2float_rd(x);
#else
    float f = (float)x;
    return (double)f > x ? nextafterf(f, -INFINITY) : f;
#endif
}
OCTA_HD inline double f32_ulp(float v) {   // spacing of single-precision numbers at |v| (>= the rounding error of one directed conversion)
    const float a = fabsf(v);
    return (double)nextafterf(a, INFINITY) - (double)a;
}
OCTA_HD inline void atomic_max_u32(unsigned *p, unsigned v) {
#if defined(__HIP_DEVICE_COMPILE__)
    atomicMax(p, v);
#else
    if (v > *p) *p = v;
#endif
}
OCTA_HD inline void atomic_min_u32(unsigned *p, unsigned v) {
#if defined(__HIP_DEVICE_COMPILE__)
    atomicMin(p, v);
#else
    if (v < *p) *p = v;
#endif
}

// scipy cKDTree build order (leafsize 16, compact, median): fills kd_idx (tree.indices) and kd_rank.
// Level-synchronous: range boundaries depend only on n; a range that became a leaf is marked done.
// `need` (optional) flags the points whose rank will be read: a range without any such point is not
// partitioned further (its internal order is never observed), which prunes most of the deep levels.
// LDS (78 KiB, kd_lds_layout below): packed elements u32[OCAP], per-level range tables, and the swap mailbox, whose area (plus the
// tail) hosts the box table while no partition is running.
constexpr int KD_TAB_OFF = OCAP * (int)sizeof(kdw_t);       // rs, re, rs2, re2 (idx_t each), rd (s8) per range
constexpr int KD_BOX_BYTES = KD_RANGES * 24;
static_assert(KD_RANGES >= OCAP / 17 + 1, "ranges per level");
static_assert(KD_MAILBOX_OFF % 16 == 0 && KD_MAILBOX_OFF >= KD_TAB_OFF + KD_RANGES * KD_TAB_BYTES_PER_RANGE, "kd tables overlap the mailbox");
static_assert(KD_MAILBOX_OFF + (KD_BOX_BYTES > KD_MAILBOX_BYTES ? KD_BOX_BYTES : KD_MAILBOX_BYTES) <= SIM_USER_BYTES, "kd table layout (box table and mailbox share their area)");
// xy: [n][2] floats of global scratch -- the x and y coordinates rounded to single precision, written here and read by the box and
// key passes of every level (8 B instead of 24 + 8 B of gathered doubles per element and level: the gathers were the kernel's
// largest HBM/L2 read stream); zlo / zhi: bounds of every point's z (the slab is thin: z wins the "largest spread" only for ranges
// whose x and y boxes are thinner than the slab, and those are measured exactly).
OCTA_HD inline void kd_build(const Blk &b, const double *pts, int n, idx_t *out_idx, idx_t *out_rank,
                              float *xy, double zlo, double zhi, long *kdprof = nullptr, const unsigned char *need = nullptr,
                              const bool flag_in_sign = false) {
#if defined(__HIP_DEVICE_COMPILE__) && !defined(OCTA_SIM_PROF_ASSIGN) && !defined(OCTA_SIM_PROF_SAMPLE) && !defined(OCTA_SIM_PROF_SET) && !defined(OCTA_SIM_PROF_MAIL) && !defined(OCTA_SIM_PROF_SEQ) && !defined(OCTA_SIM_PROF_SEQ2)
#define KDP(slot) do { if (kdprof && b.tid == 0) { long _t = (long)wall_clock64(); kdprof[slot] += _t - _kt; _kt = _t; } } while (0)
    long _kt = (long)wall_clock64();
#else
#define KDP(slot) do { } while (0)
#endif
    kdw_t *kv = reinterpret_cast<kdw_t *>(b.user_of<1>());
    idx_t *tab = reinterpret_cast<idx_t *>(b.user_of<1>() + KD_TAB_OFF);
    idx_t *rs = tab, *re = tab + KD_RANGES;                        // range start / end
    idx_t *rs2 = tab + 2 * KD_RANGES, *re2 = tab + 3 * KD_RANGES;  // next level
    signed char *rd = reinterpret_cast<signed char *>(tab + 4 * KD_RANGES); // bbox pass: 1 = holds a needed point; then split dim (-1 = leaf / not needed)
    unsigned *bbf = reinterpret_cast<unsigned *>(b.user_of<1>() + KD_MAILBOX_OFF); // [nr][6]: max xyz (rounded up), min xyz (rounded down)
#if defined(__HIP_DEVICE_COMPILE__) && !OCTA_SIM_LARGE
    // Round 6: the element array is sized for OCAP sinks but most builds have far fewer -- when 12 bytes per sink fit it, the single-precision (x, y)
    // copies live in the LDS BEHIND the n element words instead of in HBM scratch: the box and key passes of every level gather them (63 % of the builds,
    // 22 % of the gathered elements of a run). One code path: the pointer is generic, the loads are FLAT.
    if ((size_t)n * (sizeof(kdw_t) + 8) + 8 <= (size_t)KD_TAB_OFF)
        xy = reinterpret_cast<float *>(b.user_of<1>() + (((size_t)n * sizeof(kdw_t) + 7) & ~(size_t)7));
#endif
    for (int i = b.tid; i < n; i += b.nth) {
        kv[i] = (kdw_t)i;
        // round to nearest: x lies within one float spacing of it. flag_in_sign (the caller vouches for x >= 0: the sinks are valid
        // positions): the "rank is needed" flag rides in the sign of x -- one gather less per element and level of the box pass
        const float fx = (float)pts[3 * i];
        xy[2 * i] = (flag_in_sign && need[i]) ? -fx : fx; xy[2 * i + 1] = (float)pts[3 * i + 1];
    }
    if (b.tid == 0) { rs[0] = 0; re[0] = (idx_t)n; }
    const float z_up = f32_round_up(zhi), z_dn = f32_round_down(zlo);
    b.sync();
    int nr = (n > 16) ? 1 : 0;  // a range of <= leafsize points is a leaf: left in input order
    while (nr > 0) {
        // 1. bounding box per range, element-parallel: every thread walks a contiguous chunk of the element
        //    array (ranges are sorted, disjoint slices of it), gathers the three coordinates, keeps exact
        //    running extrema and folds their single-precision outer bounds into the per-range table.
        for (int q = b.tid; q < nr; q += b.nth) {
            for (int k = 0; k < 2; k++) { bbf[6 * q + k] = 0u; bbf[6 * q + 3 + k] = ~0u; }
            bbf[6 * q + 2] = f32_sortable(z_up); bbf[6 * q + 5] = f32_sortable(z_dn);      // z: the slab's bounds for every range
            rd[q] = 0;
        }
        b.sync();
        {
            const int chunk = ((n + b.nth - 1) / b.nth) | 1;      // odd: the lanes' element words lie in distinct LDS banks
            const int i0 = b.tid * chunk, i1 = (i0 + chunk < n) ? i0 + chunk : n;
            int q = 0;
            if (i0 < i1) {  // last range starting at or before i0
                int lo_ = 0, hi_ = nr - 1;
                while (lo_ < hi_) { int mid = (lo_ + hi_ + 1) >> 1; if (rs[mid] <= i0) lo_ = mid; else hi_ = mid - 1; }
                q = lo_;
            }
            bool have = false, hit = false;
            float mx[2] = {0, 0}, mn[2] = {0, 0};
            // the extrema of the ROUNDED coordinates, widened by one float spacing each way, bound the true extrema
            auto flush = [&](int qq) {
                for (int k = 0; k < 2; k++) {
                    atomic_max_u32(&bbf[6 * qq + k], f32_sortable(nextafterf(mx[k], INFINITY)));
                    atomic_min_u32(&bbf[6 * qq + 3 + k], f32_sortable(nextafterf(mn[k], -INFINITY)));
                }
            };
            // KB elements per round: the range walk and the element words come out of the LDS first, then the KB coordinate pairs (and
            // need flags) of the live ones are fetched TOGETHER, then folded in element order (round 4: 30.9 -> 27.1 ms per sample; the
            // same batching in the key pass below measured slower, 18.9 -> 21.7, and is not used there). The walk keeps the bounds of
            // its range and the start of the next one in registers: one table read per range crossed instead of three per element.
            constexpr int KB = 8;
            int qw = q, cs = 0, ce = 0, ns = 0x7fffffff;
            if (i0 < i1) { cs = rs[qw]; ce = re[qw]; ns = qw + 1 < nr ? (int)rs[qw + 1] : 0x7fffffff; }
            for (int ib = i0; ib < i1; ib += KB) {
                int eq[KB], eid[KB];
                kdw_t ew[KB];
#pragma unroll
                for (int k = 0; k < KB; k++) ew[k] = kv[ib + k < i1 ? ib + k : i1 - 1];
#pragma unroll
                for (int k = 0; k < KB; k++) {
                    const int i = ib + k;
                    eq[k] = -1; eid[k] = 0;
                    if (i < i1) {
                        while (i >= ns) { qw++; cs = ns; ce = re[qw]; ns = qw + 1 < nr ? (int)rs[qw + 1] : 0x7fffffff; }
                        if (i >= cs && i < ce) { eq[k] = qw; eid[k] = (int)(ew[k] & KD_IDX_MASK); }
                    }
                }
                float ex[KB], ey[KB];
                unsigned char nd[KB];
#pragma unroll
                for (int k = 0; k < KB; k++) {
                    const float sx = xy[2 * eid[k]];
                    ex[k] = flag_in_sign ? fabsf(sx) : sx; ey[k] = xy[2 * eid[k] + 1];
                    nd[k] = flag_in_sign ? (unsigned char)(f32_bits(sx) >> 31) : (need ? need[eid[k]] : (unsigned char)1);
                }
#pragma unroll
                for (int k = 0; k < KB; k++) {
                    if (eq[k] < 0) continue;    // element of a finished leaf / beyond the chunk
                    if (eq[k] != q) {
                        if (have) { flush(q); have = false; }
                        if (hit) { rd[q] = 1; hit = false; }
                        q = eq[k];
                    }
                    if (nd[k]) hit = true;
                    if (!have) { mx[0] = mn[0] = ex[k]; mx[1] = mn[1] = ey[k]; }
                    else {
                        mx[0] = mx[0] > ex[k] ? mx[0] : ex[k]; mn[0] = mn[0] < ex[k] ? mn[0] : ex[k];
                        mx[1] = mx[1] > ey[k] ? mx[1] : ey[k]; mn[1] = mn[1] < ey[k] ? mn[1] : ey[k];
                    }
                    have = true;
                }
            }
            if (have) flush(q);
            if (hit) rd[q] = 1;
        }
        int *pend_n = b.coll() + 396, *pend = b.coll() + 420;        // ranges whose exact extrema the whole workgroup measures (below)
        if (b.tid == 0) *pend_n = 0;
        b.sync();
        KDP(0);
        // split dimension = the first one with the largest spread
        for (int q = b.tid; q < nr; q += b.nth) {
            if (rd[q] == 0) { rd[q] = -1; continue; }     // no needed point below this range
            double shi[3], slo[3];
            for (int k = 0; k < 3; k++) {
                const float up = f32_unsortable(bbf[6 * q + k]), dn = f32_unsortable(bbf[6 * q + 3 + k]);
                shi[k] = (double)up - (double)dn;
                slo[k] = shi[k] - 2.0 * (f32_ulp(up) + f32_ulp(dn));
            }
            slo[2] = 0.0;      // z: the table holds the slab's bounds, not the range's -- an upper bound of the spread only
            int d = 0;
            if (shi[1] > shi[d]) d = 1;
            if (shi[2] > shi[d]) d = 2;
            bool sure = true;
            for (int k = 0; k < 3; k++) if (k != d && !(slo[d] > shi[k])) sure = false;
            if (!OCTA_UNLIKELY(!sure)) { rd[q] = (signed char)d; continue; }
            // overlapping intervals (or a degenerate box): the exact extrema of the range decide, as scipy's do. A long range is left
            // to the whole workgroup: the extreme sinks of the list persist over many iterations, so a near-tie of the ROOT range's
            // x and y spreads (about one sample in 300) used to cost one thread a scan of all ~10^4 sinks in every iteration it lasted
            // (measured: 128 ms in the slowest sample of a 512-sample launch, i.e. +20 % on the launch)
            if ((int)re[q] - (int)rs[q] > 256) {
                const int slot = atomic_add_int(pend_n, 1);
                if (slot < 64) { pend[slot] = q; rd[q] = (signed char)-2; continue; }
            }
            double hi3[3] = {0, 0, 0}, lo3[3] = {0, 0, 0};
            for (int i = rs[q]; i < re[q]; i++) {
                const double *p = pts + 3 * (int)(kv[i] & KD_IDX_MASK);
                for (int k = 0; k < 3; k++) {
                    if (i == rs[q]) { hi3[k] = lo3[k] = p[k]; }
                    else { hi3[k] = hi3[k] > p[k] ? hi3[k] : p[k]; lo3[k] = lo3[k] < p[k] ? lo3[k] : p[k]; }
                }
            }
            d = 0;
            double size = 0;
            for (int k = 0; k < 3; k++) if (hi3[k] - lo3[k] > size) { d = k; size = hi3[k] - lo3[k]; }
            rd[q] = (hi3[d] == lo3[d]) ? (signed char)-1 : (signed char)d;
        }
        b.sync();
        {
            const int n_pend = OCTA_UNI(*pend_n) < 64 ? OCTA_UNI(*pend_n) : 64;
            unsigned long long *ex = reinterpret_cast<unsigned long long *>(b.coll() + 400);      // max xyz, min xyz as sortable words
            for (int t = 0; t < n_pend; t++) {
                const int q = OCTA_UNI(pend[t]);
                for (int k = b.tid; k < 6; k += b.nth) ex[k] = k < 3 ? 0ull : ~0ull;
                b.sync();
                double hi3[3] = {0, 0, 0}, lo3[3] = {0, 0, 0};
                bool any = false;
                for (int i = (int)rs[q] + b.tid; i < (int)re[q]; i += b.nth) {
                    const double *p = pts + 3 * (int)(kv[i] & KD_IDX_MASK);
                    for (int k = 0; k < 3; k++) {
                        if (!any) { hi3[k] = lo3[k] = p[k]; }
                        else { hi3[k] = hi3[k] > p[k] ? hi3[k] : p[k]; lo3[k] = lo3[k] < p[k] ? lo3[k] : p[k]; }
                    }
                    any = true;
                }
                if (any)
                    for (int k = 0; k < 3; k++) { atomic_max_u64(&ex[k], dbl_sortable(hi3[k])); atomic_min_u64(&ex[3 + k], dbl_sortable(lo3[k])); }
                b.sync();
                if (b.tid == 0) {
                    int d = 0;
                    double size = 0, lo_d = 0, hi_d = 0;
                    for (int k = 0; k < 3; k++) {
                        const double hk = dbl_unsortable(ex[k]), lk = dbl_unsortable(ex[3 + k]);
                        if (k == 0) { hi_d = hk; lo_d = lk; }
                        if (hk - lk > size) { d = k; size = hk - lk; hi_d = hk; lo_d = lk; }
                    }
                    rd[q] = (hi_d == lo_d) ? (signed char)-1 : (signed char)d;
                }
                b.sync();
            }
        }
        KDP(1);
        // 2. quantised split-dimension keys, element-parallel with the same chunking
        {
            const int chunk = ((n + b.nth - 1) / b.nth) | 1;      // odd: the lanes' element words lie in distinct LDS banks
            const int i0 = b.tid * chunk, i1 = (i0 + chunk < n) ? i0 + chunk : n;
            int q = 0;
            if (i0 < i1) {
                int lo_ = 0, hi_ = nr - 1;
                while (lo_ < hi_) { int mid = (lo_ + hi_ + 1) >> 1; if (rs[mid] <= i0) lo_ = mid; else hi_ = mid - 1; }
                q = lo_;
            }
            int qc = -1, d = -1;
            double mnd = 0, scale = 0;
            int cs = 0, ce = 0, ns = 0x7fffffff;
            if (i0 < i1) { cs = rs[q]; ce = re[q]; ns = q + 1 < nr ? (int)rs[q + 1] : 0x7fffffff; }
            for (int i = i0; i < i1; i++) {
                while (i >= ns) { q++; cs = ns; ce = re[q]; ns = q + 1 < nr ? (int)rs[q + 1] : 0x7fffffff; }
                if (i < cs || i >= ce) continue;
                if (q != qc) {
                    qc = q; d = rd[q];
                    if (d >= 0) {
                        mnd = (double)f32_unsortable(bbf[6 * q + 3 + d]);
                        const double w = (double)f32_unsortable(bbf[6 * q + d]) - mnd;
                        scale = w > 0.0 ? (double)KD_QMAX / w : 0.0;
                    }
                }
                if (d >= 0) {
                    const unsigned id = (unsigned)(kv[i] & KD_IDX_MASK);
                    // x / y: the rounded coordinate (rounding is monotone, so the key still is); z: the double itself
                    const float fv = d < 2 ? xy[2 * id + d] : 0.f;
                    const double cv = d < 2 ? (double)((flag_in_sign && d == 0) ? fabsf(fv) : fv) : pts[3 * id + 2];
                    kv[i] = (kd_quant(cv, mnd, scale) << KD_IDX_BITS) | (kdw_t)id;
                }
            }
        }
        b.sync();    // the box table is dead from here on: its area is the mailbox of the partitions
        KDP(2);
        // 3. nth_element per range: long ranges by one wave each, short ranges by a quarter wave each
#if defined(__HIP_DEVICE_COMPILE__)
        {
            idx_t *mb = reinterpret_cast<idx_t *>(b.user_of<1>() + KD_MAILBOX_OFF);
            const int wv = b.tid >> 6, nw = (b.nth + 63) >> 6;
            for (int q = wv; q < nr; q += nw) {
                int d = rd[q];
                int s = rs[q], e = re[q];
                if (d < 0 || e - s < KD_TEAM_MIN) continue;
                const KdPair kp = {kv, pts, d};
                if (e - s > KD_MASK_MAX) kd_nth_element_wave_long(kp, s, s + (e - s) / 2, e, mb);
                else if (e - s > 4000) kd_nth_element_team<64, 4>(kp, s, s + (e - s) / 2, e, true, mb);
                else kd_nth_element_team<64, 1>(kp, s, s + (e - s) / 2, e, true, mb);
            }
        }
        // (no barrier here: the quarter-wave partitions below work on OTHER ranges -- their own elements, their own mailbox slots -- so a wave
        // that is through with its long ranges goes on with the short ones)
        KDP(3);
        {
            idx_t *mb = reinterpret_cast<idx_t *>(b.user_of<1>() + KD_MAILBOX_OFF);
            const int team = b.tid >> 4, nteam = b.nth >> 4;
            for (int q0 = 0; q0 < nr; q0 += nteam) {
                int q = q0 + team;
                bool live = q < nr;
                int s = 0, e = 0, d = 0;
                if (live) {
                    s = rs[q]; e = re[q]; d = rd[q];
                    live = d >= 0 && e - s < KD_TEAM_MIN;
                }
                const KdPair kp = {kv, pts, d < 0 ? 0 : d};
                kd_nth_element_team<16, 1>(kp, s, s + (e - s) / 2, e, live, mb);
            }
        }
#else
        for (int q = b.tid; q < nr; q += b.nth) {
            int d = rd[q];
            if (d < 0) continue;
            int s = rs[q], e = re[q];
            const KdPair kp = {kv, pts, d};
            kd_nth_element(kp, s, s + (e - s) / 2, e);
        }
#endif
        b.sync();
        KDP(4);
        // 4. children that are still longer than a leaf form the next level (ordered compaction)
        {
            int base = 0;
            for (int q0 = 0; q0 < nr; q0 += b.nth) {
                const int q = q0 + b.tid;
                int s = 0, e = 0, m = 0, c = 0;
                if (q < nr && rd[q] >= 0) {
                    s = rs[q]; e = re[q]; m = s + (e - s) / 2;
                    c = (m - s > 16 ? 1 : 0) + (e - m > 16 ? 1 : 0);
                }
                int ex;
                const int tot = blk_scan(b, c, &ex);
                int pos = base + ex;
                if (m - s > 16) { if (pos < KD_RANGES) { rs2[pos] = (idx_t)s; re2[pos] = (idx_t)m; } pos++; }
                if (e - m > 16) { if (pos < KD_RANGES) { rs2[pos] = (idx_t)m; re2[pos] = (idx_t)e; } }
                base += tot;
            }
            nr = base < KD_RANGES ? base : KD_RANGES;
        }
        // (no barrier here: the next level's first step writes the box table and the range flags -- neither is read after the scan's own barriers
        // above -- and its barrier comes before anything reads the new ranges)
        { idx_t *t = rs; rs = rs2; rs2 = t; t = re; re = re2; re2 = t; }
        KDP(5);
    }
    for (int i = b.tid; i < n; i += b.nth) { const idx_t id = (idx_t)(kv[i] & KD_IDX_MASK); out_idx[i] = id; out_rank[id] = (idx_t)i; }
    b.sync();
    KDP(6);
#undef KDP
}


// ------------------------------------------------------------------ uniform grid (LDS-binned counting sort)
// Points are binned on (x, y) with a cell edge >= the query radius, so a radius query visits the
// 3x3 cell neighbourhood. Cell offsets and the id list stay in LDS (histogram -> scan -> scatter); the
// coordinates are written to HBM scratch IN CELL ORDER, so a query reads three contiguous runs (one per
// cell row) with independent loads instead of chasing start -> item -> point. Item order inside a cell
// is not defined; every consumer is order-free (existence tests, arg-min with an explicit id tie-break,
// pair lists that are sorted afterwards). The grid is valid until the next use of the LDS user area.
constexpr int GRID_N = NCAP > OCAP ? NCAP : OCAP;  // points per grid
struct Grid {
    int nx, ny;
    double x0, y0, inv, cell;
    const int *cell_end;           // LDS [nx*ny]: end offset of each cell (its start is the end of the previous cell)
    const idx_t *items;   // LDS [n]: point ids in cell order
    const double *spts;            // HBM [n][3]: coordinates in cell order
    const float *sptf;             // ... or, for a grid built `as_float`, the coordinates ROUNDED to single precision (12 B per point)
    const double *src;             // the point list the grid was built over (item id -> exact coordinates)
};
OCTA_HD inline int grid_clampi(int v, int hi) { return v < 0 ? 0 : (v > hi ? hi : v); }
OCTA_HD inline int grid_cx(const Grid &G, double x) { return grid_clampi((int)floor((x - G.x0) * G.inv), G.nx - 1); }
OCTA_HD inline int grid_cy(const Grid &G, double y) { return grid_clampi((int)floor((y - G.y0) * G.inv), G.ny - 1); }

// ids: optional list of point ids (n entries) -- point i is pts[3*ids[i]]; items then hold ids[i]
OCTA_HD inline Grid grid_build_once(const Blk &b, const SimArrays &A, const double *pts, const int *ids, int n, double radius, int div, bool as_float,
                                    const unsigned char *mask) {
    Grid G;
    const double span = 1.2;
    double cell = fmax(radius / div, span / GRID_MAX);
    int nc = (int)ceil(span / cell);
    if (nc < 1) nc = 1;
    if (nc > GRID_MAX) nc = GRID_MAX;
    G.nx = G.ny = nc; G.x0 = G.y0 = -0.1; G.inv = 1.0 / cell; G.cell = cell;
    const int ncell = nc * nc;
    int *hist = reinterpret_cast<int *>(b.user_of<2>());  // [ncell + 1]
    idx_t *items = reinterpret_cast<idx_t *>(b.user_of<2>() + (size_t)(GRID_MAX * GRID_MAX + 1) * 4);  // [GRID_N]
    static_assert((size_t)(GRID_MAX * GRID_MAX + 1) * 4 + (size_t)GRID_N * sizeof(idx_t) <= (size_t)SIM_USER_BYTES, "grid table layout");
    G.cell_end = hist; G.items = items; G.spts = A.grid_pts; G.sptf = reinterpret_cast<const float *>(A.grid_pts); G.src = pts;
    if (n > GRID_N) n = GRID_N;
    b.sync();
    for (int c = b.tid; c <= ncell; c += b.nth) hist[c] = 0;
    b.sync();
    // both passes fetch four points per thread and round: the loads (id, then coordinates) of the four are independent
    constexpr int GB = 4;
    for (int i0 = b.tid; i0 < n; i0 += GB * b.nth) {
        double px[GB], py[GB];
        bool live[GB];
#pragma unroll
        for (int k = 0; k < GB; k++) {
            const int i = i0 + k * b.nth;
            const int ic = i < n ? i : n - 1;
            const double *p = pts + 3 * (ids ? ids[ic] : ic);
            px[k] = p[0]; py[k] = p[1];
            live[k] = i < n && (!mask || mask[ic]);
        }
#pragma unroll
        for (int k = 0; k < GB; k++)
            if (live[k]) atomic_add_int(&hist[grid_cy(G, py[k]) * nc + grid_cx(G, px[k])], 1);
    }
    b.sync();
    {   // exclusive scan over the cells: contiguous chunk per thread + ONE block scan
        const int total = ncell + 1;
        const int chunk = ((total + b.nth - 1) / b.nth) | 1;      // odd: distinct LDS banks across the lanes
        const int c0 = b.tid * chunk, c1 = (c0 + chunk < total) ? c0 + chunk : total;
        int local = 0;
        for (int c = c0; c < c1; c++) local += hist[c];
        int ex;
        blk_scan(b, local, &ex);
        int run = ex;
        for (int c = c0; c < c1; c++) { int v = hist[c]; hist[c] = run; run += v; }
    }
    b.sync();
    for (int i0 = b.tid; i0 < n; i0 += GB * b.nth) {
        V3 p[GB];
        int id[GB];
        bool live[GB];
#pragma unroll
        for (int k = 0; k < GB; k++) {
            const int i = i0 + k * b.nth;
            const int ic = i < n ? i : n - 1;
            id[k] = ids ? ids[ic] : ic;
            p[k] = ld3(pts + 3 * id[k]);
            live[k] = i < n && (!mask || mask[ic]);
        }
#pragma unroll
        for (int k = 0; k < GB; k++)
            if (live[k]) {
                int pos = atomic_add_int(&hist[grid_cy(G, p[k].y) * nc + grid_cx(G, p[k].x)], 1);  // hist[c] ends as the END of cell c
                items[pos] = (idx_t)id[k];
                if (as_float) { float *f = reinterpret_cast<float *>(A.grid_pts) + 3 * pos; f[0] = (float)p[k].x; f[1] = (float)p[k].y; f[2] = (float)p[k].z; }
                else st3(A.grid_pts + 3 * pos, p[k]);
            }
    }
    b.sync();
    return G;
}
// div: cells per query radius along an axis (1: a query visits 3 x 3 cells). as_float: the cell-ordered copy holds the coordinates rounded
// to single precision (grid_visit_f; the caller decides exactly with Grid::src where the rounding could matter). mask (optional, without
// ids): only the points i with mask[i] != 0 enter the grid -- the list is streamed once, coalesced, instead of gathered through an id list
OCTA_HD inline Grid grid_build(const Blk &b, const SimArrays &A, const double *pts, const int *ids, int n, double radius, int div = 1,
                               bool as_float = false, const unsigned char *mask = nullptr) {
#if OCTA_SIM_DUP & 2
    grid_build_once(b, A, pts, ids, n, radius, div, as_float, mask);
#endif
    return grid_build_once(b, A, pts, ids, n, radius, div, as_float, mask);
}
// visits every point of the cells overlapping [px-radius, px+radius] x [py-radius, py+radius]: body(item = point id, pt = coordinates).
// Three cell rows per round (a query radius never exceeds the cell edge, so one round is the rule): the bounds of the rows' runs come
// out of the LDS together, then the first GRID_VB points of EVERY run are fetched together -- their loads are independent -- and only
// then handed to the body; runs longer than that continue GRID_VB points at a time. Round 4: the macro this replaces fetched one point
// per loop trip, i.e. one dependent L2 / HBM round trip per visited point (~10 per query, 43 queries per thread and assignment:
// that chain, not bandwidth, was the 57 ms of phase_assign). The order of the visits is row by row as before; every consumer is
// order-free anyway (existence tests, arg-min with an explicit id tie-break, pair lists that are sorted afterwards).
constexpr int GRID_VB = 4;
// single-precision copy: |rounded - exact| <= 2^-24 |x| per coordinate. The simulator's point lists live in the unit square plus a
// margin of a few growth steps; for |x| < 4 a distance measured in single precision from the ROUNDED query to a rounded point is off
// by less than 2 x sqrt(3) x 2.4e-7 (both roundings) + ~1e-8 (the arithmetic) = 8.4e-7 < GRID_F_TAU
constexpr double GRID_F_TAU = 1e-6;
struct Pt3f { float x, y, z; };
OCTA_HD inline float sqdist_f(Pt3f a, float cx, float cy, float cz) { const float dx = a.x - cx, dy = a.y - cy, dz = a.z - cz; return fmaf(dz, dz, fmaf(dy, dy, dx * dx)); }
template <bool FLT> struct GridPt;
template <> struct GridPt<false> { typedef V3 T; static OCTA_HD inline V3 load(const Grid &G, int k) { return ld3(G.spts + 3 * k); } static OCTA_HD inline V3 zero() { return v3(0, 0, 0); } };
template <> struct GridPt<true> {
    typedef Pt3f T;
    static OCTA_HD inline Pt3f load(const Grid &G, int k) { const float *f = G.sptf + 3 * k; Pt3f r = {f[0], f[1], f[2]}; return r; }
    static OCTA_HD inline Pt3f zero() { Pt3f r = {0.f, 0.f, 0.f}; return r; }
};
template <bool FLT, class F>
OCTA_HD inline void grid_visit_impl(const Grid &G, double px, double py, double radius, F &&body) {
    const int cy0 = grid_cy(G, py - radius), cy1 = grid_cy(G, py + radius);
    const int cx0 = grid_cx(G, px - radius), cx1 = grid_cx(G, px + radius);
    for (int cyb = cy0; cyb <= cy1; cyb += 3) {
        int k0[3], k1[3];
#pragma unroll
        for (int r = 0; r < 3; r++) {
            const int row = cyb + r;
            k0[r] = k1[r] = 0;
            if (row <= cy1) {
                const int c0 = row * G.nx + cx0;
                k0[r] = c0 ? G.cell_end[c0 - 1] : 0;
                k1[r] = G.cell_end[row * G.nx + cx1];
            }
        }
        typename GridPt<FLT>::T pt[3][GRID_VB];
        int it[3][GRID_VB];
#pragma unroll
        for (int r = 0; r < 3; r++)
#pragma unroll
            for (int u = 0; u < GRID_VB; u++) {
                const int k = k0[r] + u < k1[r] ? k0[r] + u : k0[r];      // k0[r] is a valid slot whenever the grid holds a point; an empty grid has no run
                pt[r][u] = k0[r] < k1[r] ? GridPt<FLT>::load(G, k) : GridPt<FLT>::zero();
                it[r][u] = k0[r] < k1[r] ? (int)G.items[k] : 0;
            }
#pragma unroll
        for (int r = 0; r < 3; r++) {
#pragma unroll
            for (int u = 0; u < GRID_VB; u++)
                if (k0[r] + u < k1[r]) body(it[r][u], pt[r][u]);
            for (int kb = k0[r] + GRID_VB; kb < k1[r]; kb += GRID_VB) {
                typename GridPt<FLT>::T q[GRID_VB];
                int iq[GRID_VB];
#pragma unroll
                for (int u = 0; u < GRID_VB; u++) {
                    const int k = kb + u < k1[r] ? kb + u : kb;
                    q[u] = GridPt<FLT>::load(G, k);
                    iq[u] = (int)G.items[k];
                }
#pragma unroll
                for (int u = 0; u < GRID_VB; u++)
                    if (kb + u < k1[r]) body(iq[u], q[u]);
            }
        }
    }
}
template <class F>
OCTA_HD inline void grid_visit(const Grid &G, double px, double py, double radius, F &&body) { grid_visit_impl<false>(G, px, py, radius, body); }
// over a grid built `as_float`: body(item, Pt3f = the point ROUNDED to single precision)
template <class F>
OCTA_HD inline void grid_visit_f(const Grid &G, double px, double py, double radius, F &&body) { grid_visit_impl<true>(G, px, py, radius, body); }

// ------------------------------------------------------------------ Murray propagation (one thread)
// dirty list of the ordered pass: inter-node groups that did not sprout under the speculation but whose
// child radius was changed by an earlier node of the same pass (they must be re-evaluated at their turn)
struct DirtyList {
    int *v;      // ascending, unique
    int n, cap;
    bool overflow;
};
OCTA_HD inline void dirty_insert(DirtyList &D, int g) {
    int i = 0;
    while (i < D.n && D.v[i] < g) i++;
    if (i < D.n && D.v[i] == g) return;
    if (OCTA_UNLIKELY(D.n >= D.cap)) { D.overflow = true; return; }
    for (int k = D.n; k > i; k--) D.v[k] = D.v[k - 1];
    D.v[i] = g;
    D.n++;
}

// State of the ordered pass that lives in LDS / registers instead of HBM.
struct SeqLds {
    double *rad;               // [NCAP] radii of the forest: the HBM array itself (L2-resident during the pass; round 2 kept an LDS copy)
    idx_t *par;       // [NCAP] node word (LDS): parent id (PAR_MASK = none) | PAR_DEF (marked for the flush) | PAR_TAG (carries this pass's tag):
                      // ONE LDS read per ancestor of a walk's chain enumeration (round 6: parent, deferred bit and tag were three)
    const double *log_tab;     // glibc pow tables (gpow.h) (LDS)
    const uint64_t *exp_tab;
    int *deferred;             // [NCAP / 32] bitmap: radius to be recomputed when the pass ends (2 KiB, LDS)
    int *changed;              // [GCAP / 32] bitmap over this pass's groups: a walk rewrote the radius of the group's child (1 KiB, LDS)
    double *fl_val;            // [MURRAY_FLUSH_LDS] radii the flush has finished, by list slot (LDS, round 6)
    int *fl_done;              // [MURRAY_FLUSH_LDS / 32] bitmap: slot finished in an earlier round
    int *slot_of;              // [NCAP] (HBM scratch) list slot of a deferred node, written by the walk that marks it
};
#ifndef OCTA_MURRAY_EPT
#define OCTA_MURRAY_EPT 4      // (-DOCTA_MURRAY_EPT=1: 64-slot waves -- a test build in which nearly every pass takes the three-wave and the whole-workgroup forms)
#endif
constexpr int MURRAY_EPT = OCTA_MURRAY_EPT;
constexpr int MURRAY_FLUSH_WAVE = MURRAY_EPT * 64;  // deferred nodes one wave of the flush keeps in registers (MURRAY_EPT per lane)
constexpr int MURRAY_FLUSH_LDS = 3 * MURRAY_FLUSH_WAVE;   // deferred nodes of a pass whose finished radii / done bits live in the LDS: up to three waves share the rounds
// PAR_TAG: the node carries THIS pass's tag in child_group, i.e. it is the child of an inter-node group of the pass. A walk that meets
// no such node has no eager part and needs no topology record from HBM at all (round 6).
constexpr int PAR_BITS = (int)sizeof(idx_t) * 8 - 2;
constexpr unsigned PAR_MASK = (1u << PAR_BITS) - 1u, PAR_DEF = 1u << PAR_BITS, PAR_TAG = 2u << PAR_BITS;
static_assert((unsigned)NCAP <= PAR_MASK, "node ids fit the parent field of a node word");
OCTA_HD inline bool changed_get(const SeqLds &L, int g) { return ((unsigned)L.changed[g >> 5] >> (g & 31)) & 1u; }
OCTA_HD inline void changed_set(const SeqLds &L, int g) { L.changed[g >> 5] |= (int)(1u << (g & 31)); }   // the ordered pass's wave only, all lanes alike
OCTA_HD inline bool deferred_get(const SeqLds &L, int id) { return ((unsigned)L.deferred[id >> 5] >> (id & 31)) & 1u; }
struct WalkRec { int nch, c0, c1, cg; double k; };
OCTA_HD inline WalkRec walk_load(const SimArrays &A, int f, int id, bool want_cg) {
    WalkRec r;
    const int j = id < 0 ? 0 : id;
    r.nch = A.nnch_of(f)[j]; r.c0 = A.nch0_of(f)[j]; r.c1 = A.nch1_of(f)[j];
    r.cg = want_cg ? A.child_group[j] : 0;
    r.k = A.nkap_of(f)[j];
    return r;
}
OCTA_HD inline int walk_parent(const SeqLds &L, int id) {
    if (id < 0) return -1;
    const unsigned p = (unsigned)L.par[id] & PAR_MASK;
    return p == PAR_MASK ? -1 : (int)p;
}

// Murray's law from node id towards the root (arterial_tree.py:174-184): the reference recomputes (r_c0^k + r_c1^k)^(1/k) for
// every node on the way, one walk per sprouting node, until a radius does not change or the root (never updated) is reached.
// A radius is READ during the pass only as "the child radius of an inter-node whose turn is still to come" (eval_inter; those
// children carry this pass's tag in child_group). So a walk is split:
//   * eager part -- from `id` up to the highest node on the path whose tag names a group still to be visited: computed now, in
//     the reference's order and with its early exit; everything at or below a pending node therefore always holds exact radii;
//   * deferred part -- the nodes above: only marked (bitmap in LDS, list in HBM) up to the first node an earlier walk of this
//     pass already marked. Their radii are pure functions of their children's, so recomputing each marked node ONCE when the
//     pass ends (murray_flush, children before parents, all ready nodes of the workgroup in parallel) leaves the radii the
//     walks would have left -- the reference's early exit cannot trigger above a sprouting node (each ancestor's k-th-power sum
//     grows by at least one leaf's share, orders of magnitude above an ulp), and if it could, an unchanged child gives its
//     parent the value it already holds.
// Radii and the parent chain are read from LDS; topology records come from HBM/L2 lane-parallel, one round trip per 64 ancestors.
#if defined(__HIP_DEVICE_COMPILE__)
__device__ inline double readlane_f64(double v, int j /* wave-uniform */) {
    int lo = __builtin_amdgcn_readlane(__double2loint(v), j), hi = __builtin_amdgcn_readlane(__double2hiint(v), j);
    return __hiloint2double(hi, lo);
}
// any pending tag on the path from `start` up to the first deferred node / the root (paths longer than one chunk only)
__device__ inline bool murray_pending_above(const SimArrays &A, int start, int cur_g, int pass_tag, const SeqLds &L) {
    const int lane = (int)(threadIdx.x & 63);
    int cur = start;
    while (true) {
        int nn = 0, mine = -1;
        bool ended = false;
        for (; nn < 64; nn++) {
            const unsigned w = (unsigned)L.par[cur], p = w & PAR_MASK;
            if (p == PAR_MASK || (w & PAR_DEF)) { ended = true; break; }
            if (lane == nn) mine = cur;
            cur = (int)p;
        }
        const int cg = lane < nn ? A.child_group[mine] : 0;
        if (__ballot(lane < nn && (cg >> (GROUP_BITS + 1)) == pass_tag && (cg & ((1 << GROUP_BITS) - 1)) > cur_g)) return true;
        if (ended) return false;
    }
}
// Wave form; all 64 lanes call it with identical arguments. Per chunk of 64 ancestors: (1) the chain is followed through the LDS
// parent array, lane j keeps the j-th node; (2) lane-parallel, every lane fetches its node's topology and tag from HBM/L2 and the
// eager length is voted; the eager lanes raise the children that are NOT on the path to the node's kappa; (3) the sequential
// chain -- two pow evaluations per node, operands fetched with readlane -- is executed uniformly; (4) the lanes store the new radii
// / mark the deferred nodes. Floating-point addition is commutative, so "on-path power + other power" is bit-identical to the
// reference's c0-then-c1 order. Returns the eager steps; n_def counts the nodes appended to the deferred list (A.act_list).
// the chain's two evaluations per node have wave-uniform operands: table entries through the scalar cache (round 6; -DOCTA_WALK_POW_LDS: the LDS tables)
#ifdef OCTA_WALK_POW_LDS
#define OCTA_WALK_POW(x, y) octa_gpow::gpow_t((x), (y), L.log_tab, L.exp_tab)
#else
#define OCTA_WALK_POW(x, y) octa_gpow::gpow_u((x), (y))
#endif
__device__ inline long murray_to_root(const SimArrays &A, int f, int id, int cur_g, int pass_tag, DirtyList *D, const SeqLds &L, int &n_def
#ifdef OCTA_SIM_PROF_SEQ2
                                      , long *seq2_dbg = nullptr
#endif
                                      ) {
    if (id < 0) return 0;
    const int lane = (int)(threadIdx.x & 63);
    double *rad = L.rad;
    long steps = 0;
    int first = id, below = -1;  // first node of the chunk and its on-path child (-1 at the start of the walk)
    double rp_prev = 0, k_last = 0, inv_k = 0;
    bool defer_rest = false;     // an earlier chunk ended its eager part: nothing pending above
    while (true) {
        int cur = first, nn = 0, mine = -1;
        unsigned minew = 0, tags = 0;
        bool ended = false;
#ifdef OCTA_SIM_PROF_SEQ2
        const long _c0 = (long)wall_clock64();
#endif
        for (; nn < 64; nn++) {
            const unsigned w = (unsigned)L.par[cur], p = w & PAR_MASK;
            if (p == PAR_MASK || (w & PAR_DEF)) { ended = true; break; }
            if (lane == nn) { mine = cur; minew = w; }
            tags |= w;
            cur = (int)p;
        }
        bool any_tag = (tags & PAR_TAG) != 0;
        if (nn == 0) break;
        int eager_n = 0;
        WalkRec r;
        r.nch = 0; r.c0 = r.c1 = -1; r.cg = 0; r.k = 0;
        // round 6: only a node that carries this pass's tag can be pending, and the tags are an LDS bitmap: three walks in four meet none
        // on their way to the first marked node / the root and are over after the chain walk -- no topology records, no tags from HBM
        if (!defer_rest && !any_tag && !ended) {
            int c2 = cur;
            while (true) {
                const unsigned w2 = (unsigned)L.par[c2], p2 = w2 & PAR_MASK;
                if (p2 == PAR_MASK || (w2 & PAR_DEF)) break;
                if (w2 & PAR_TAG) { any_tag = true; break; }
                c2 = (int)p2;
            }
        }
        if (!defer_rest && any_tag) {
            if (lane < nn) r = walk_load(A, f, mine, true);
            const unsigned long long pend = __ballot(lane < nn && (r.cg >> (GROUP_BITS + 1)) == pass_tag && (r.cg & ((1 << GROUP_BITS) - 1)) > cur_g);
            if (!ended && murray_pending_above(A, cur, cur_g, pass_tag, L)) eager_n = nn;
            else eager_n = pend ? 64 - __builtin_clzll(pend) : 0;
        }
#ifdef OCTA_SIM_PROF_SEQ2
        const long _c1 = (long)wall_clock64();
        if (seq2_dbg) { seq2_dbg[5] += eager_n > 0; seq2_dbg[6] += nn; seq2_dbg[0] += _c1 - _c0; }
#endif
        bool stop = false;
        if (eager_n > 0) {
            int onpath = __builtin_amdgcn_update_dpp(0, mine, 0x138, 0xf, 0xf, false);     // wave_shr:1: the node of the lane in front (lane 0: set below)
            if (lane == 0) onpath = below;
            double v_pw = 0, v_old = 0;  // lane j: sum of the off-path child powers, radius before the walk
            if (lane < eager_n) {
                double pw = 0;
                if (onpath < 0) {
                    if (r.nch >= 1) {
                        pw = octa_gpow::gpow_t(rad[r.c0], r.k, L.log_tab, L.exp_tab);
                        if (r.nch >= 2) pw = pw + octa_gpow::gpow_t(rad[r.c1], r.k, L.log_tab, L.exp_tab);
                    }
                } else if (r.nch >= 2) {
                    pw = octa_gpow::gpow_t(rad[r.c0 == onpath ? r.c1 : r.c0], r.k, L.log_tab, L.exp_tab);
                }
                v_pw = pw; v_old = rad[mine];
            }
            int done = 0;
            double my_rp = 0;
            for (int jj = 0; jj < eager_n; jj++) {
                const int j = __builtin_amdgcn_readfirstlane(jj);
                const int nch = __builtin_amdgcn_readlane(r.nch, j);
                if (nch == 0) { stop = true; break; }
                const double k = readlane_f64(r.k, j), pw_j = readlane_f64(v_pw, j), old_j = readlane_f64(v_old, j);
                if (k != k_last) { k_last = k; inv_k = 1.0 / k; }
                double s;
                if (j == 0 && below < 0) {
                    s = pw_j;
                } else {
                    s = OCTA_WALK_POW(rp_prev, k);
                    if (nch >= 2) s = s + pw_j;
                }
                const double rp = OCTA_WALK_POW(s, inv_k);
                steps++;
                if (old_j == rp) { stop = true; break; }
                if (lane == j) my_rp = rp;
                done = j + 1;
                rp_prev = rp;
                const int cg = __builtin_amdgcn_readlane(r.cg, j);
                if ((cg >> (GROUP_BITS + 1)) == pass_tag) {
                    const int g2 = cg & ((1 << GROUP_BITS) - 1);
                    if (g2 > cur_g) {
                        changed_set(L, g2);
                        if (D && !((cg >> GROUP_BITS) & 1)) dirty_insert(*D, g2);
                    }
                }
            }
            if (lane < done) rad[mine] = my_rp;
#ifdef OCTA_SIM_PROF_SEQ2
            if (seq2_dbg) seq2_dbg[4] += (long)wall_clock64() - _c1;
#endif
        }
        if (stop) break;          // the reference's walk ends here: nothing above changes
        if (eager_n < nn) {
            if (lane >= eager_n && lane < nn) {
                atomic_or_int(&L.deferred[mine >> 5], (int)(1u << (mine & 31)));
                L.par[mine] = (idx_t)(minew | PAR_DEF);      // a plain store: the lanes' nodes are distinct words
                A.act_list[n_def + lane - eager_n] = mine;
                L.slot_of[mine] = n_def + lane - eager_n;
            }
            n_def += nn - eager_n;
            defer_rest = true;
        }
        __builtin_amdgcn_wave_barrier();
        if (ended) break;
        below = __builtin_amdgcn_readlane(mine, 63);
        first = cur;
    }
    return steps;
}
#else
OCTA_HD inline long murray_to_root(const SimArrays &A, int f, int id, int cur_g, int pass_tag, DirtyList *D, const SeqLds &L, int &n_def) {
    long steps = 0;
    double *rad = L.rad;
    int path_len = 0, eager_n = 0;
    for (int cur = id; cur >= 0; path_len++) {
        const int par = walk_parent(L, cur);
        if (par < 0 || deferred_get(L, cur)) break;
        const int cg = A.child_group[cur];
        if ((cg >> (GROUP_BITS + 1)) == pass_tag && (cg & ((1 << GROUP_BITS) - 1)) > cur_g) eager_n = path_len + 1;
        cur = par;
    }
    int k = 0;
    for (; k < eager_n; k++) {
        const int par = walk_parent(L, id);
        const WalkRec r = walk_load(A, f, id, true);
        if (r.nch == 0) return steps;
        double s = octa_gpow::gpow_t(rad[r.c0], r.k, L.log_tab, L.exp_tab);
        if (r.nch >= 2) s = s + octa_gpow::gpow_t(rad[r.c1], r.k, L.log_tab, L.exp_tab);
        double rp = octa_gpow::gpow_t(s, 1.0 / r.k, L.log_tab, L.exp_tab);
        steps++;
        if (rad[id] == rp) return steps;
        rad[id] = rp;
        if ((r.cg >> (GROUP_BITS + 1)) == pass_tag) {
            const int g2 = r.cg & ((1 << GROUP_BITS) - 1);
            if (g2 > cur_g) {
                changed_set(L, g2);
                if (D && !((r.cg >> GROUP_BITS) & 1)) dirty_insert(*D, g2);
            }
        }
        id = par;
    }
    for (; k < path_len; k++) {
        atomic_or_int(&L.deferred[id >> 5], (int)(1u << (id & 31)));
        L.par[id] = (idx_t)((unsigned)L.par[id] | PAR_DEF);
        L.slot_of[id] = n_def;
        A.act_list[n_def++] = id;
        id = walk_parent(L, id);
    }
    return steps;
}
#endif

// The deferred radii of one pass (A.act_list[0..n_def), bitmap L.deferred), children before parents: in every round each thread
// looks at its marked nodes, adds the k-th power of every child that is final by now and, once both are in, writes the node's
// radius; the bits are cleared between two syncs, so a round only sees children finished in earlier rounds. Rounds = height of
// the marked forest. Up to MURRAY_EPT nodes per thread keep their operands in registers; a longer list re-reads them per round.
#if defined(__HIP_DEVICE_COMPILE__)
// The one-wave flush in two halves (round 6). PREPARE (the pass's own wave, right behind its last visit; 64 lanes, four list slots each):
// topology records of the marked nodes, the list slots of their marked children, the radii of their unmarked children (final already) raised
// to the node's kappa -- everything that needs the pass's LDS state (marked bitmap) and three parallel round trips to HBM -- written as one
// 32-byte record per slot. ROUNDS (any ONE wave, later): children before parents through fl_val / fl_done in the LDS, one pow evaluation per
// step for the whole wave. The rounds are a dependency chain of ~30 levels x 2 evaluations (27 ms per sample) that nothing reads before the
// OTHER forest's ordered pass is over (arterial radii: phase_sample / phase_pre of the next iteration; venous radii: the next phase_pre of
// the venous side), so they run as a side job of that pass on a wave that idles there (phase_seq).
__device__ inline void murray_flush_prepare_wave(const SimArrays &A, int f, const SeqLds &L, int n_def, FlushRec *recs, int base = 0) {
    // entries [base, base + MURRAY_FLUSH_WAVE) of the list (a longer list is prepared chunk by chunk: the entries are independent here)
    const int lane = (int)(threadIdx.x & 63) + base;
    int node[MURRAY_EPT], s0[MURRAY_EPT], s1[MURRAY_EPT];
    double kk[MURRAY_EPT];
    int pend[MURRAY_EPT];     // 1: marked child 0 not in yet (s0 = its list slot), 2: marked child 1 likewise, 4: radius not written yet,
                              // 16 / 32: child 0 / 1 is NOT marked -- its radius is final already and s0 / s1 is its node: the rounds fetch it
    int c0[MURRAY_EPT], c1[MURRAY_EPT], nch[MURRAY_EPT];
    for (int e = 0; e < MURRAY_EPT; e++) {
        const int idx = lane + e * 64;
        pend[e] = 0; node[e] = 0; c0[e] = c1[e] = 0; nch[e] = 0; kk[e] = 1; s0[e] = s1[e] = 0;
        if (idx < n_def) node[e] = A.act_list[idx];
    }
    for (int e = 0; e < MURRAY_EPT; e++) {
        if (lane + e * 64 < n_def) {
            const WalkRec r = walk_load(A, f, node[e], false);
            c0[e] = r.c0; c1[e] = r.c1; kk[e] = r.k; nch[e] = r.nch;
        }
    }
    // third round trip, all entries at once: the list slot of a marked child. (The powers of the unmarked children -- final radii -- were
    // evaluated here until the list was dealt out to three waves: they are the first thing the rounds' waves do now, off the pass's wave.)
    for (int e = 0; e < MURRAY_EPT; e++) {
        if (lane + e * 64 >= n_def) continue;
        pend[e] = 4;
        if (nch[e] >= 1) { if (deferred_get(L, c0[e])) { pend[e] |= 1; s0[e] = L.slot_of[c0[e]]; } else { pend[e] |= 16; s0[e] = c0[e]; } }
        if (nch[e] >= 2) { if (deferred_get(L, c1[e])) { pend[e] |= 2; s1[e] = L.slot_of[c1[e]]; } else { pend[e] |= 32; s1[e] = c1[e]; } }
    }
    for (int e = 0; e < MURRAY_EPT; e++) {
        const int idx = lane + e * 64;
        if (idx < n_def) { FlushRec r; r.kk = kk[e]; r.acc = 0.0; r.node = node[e]; r.pend = pend[e]; r.s0 = s0[e]; r.s1 = s1[e]; recs[idx] = r; }
    }
}

// rad: the forest's radii; fl_val [MURRAY_FLUSH_LDS] doubles and fl_done [MURRAY_FLUSH_LDS / 32] ints of LDS that belong to the calling wave
// (w, nw): wave w of nw (1 ... 3) that share the list -- the list is dealt out in blocks of 64 slots, block b to wave b % nw (entry e of
// this wave: slots (e * nw + w) * 64 ...), so that a list of up to 64 * nw nodes has ONE node per lane: a round then costs one child
// power and one root per lane instead of up to MURRAY_EPT of each in turn. The waves run side by side WITHOUT barriers: a slot's radius
// is stored before its done bit (one wave's LDS operations execute in order), a wave that finds nothing ready looks again. With at most
// two children per node the sum of a node is the same in whatever order the children finish.
// cleared: the caller zeroed fl_done in front of a barrier (several waves); otherwise this wave clears it (one wave alone).
__device__ inline int murray_flush_rounds_wave(double *rad, const double *log_tab, const uint64_t *exp_tab, double *fl_val, int *fl_done,
                                               const FlushRec *recs, int n_def, int w = 0, int nw = 1, bool cleared = false) {
    const int lane = (int)(threadIdx.x & 63);
    int node[MURRAY_EPT], s0[MURRAY_EPT], s1[MURRAY_EPT], pend[MURRAY_EPT];
    double kk[MURRAY_EPT], acc[MURRAY_EPT], cv0[MURRAY_EPT], cv1[MURRAY_EPT];
    if (!cleared) for (int i = lane; i < MURRAY_FLUSH_LDS / 32; i += 64) fl_done[i] = 0;
    int n_own = 0;
#pragma unroll
    for (int e = 0; e < MURRAY_EPT; e++) { const int left = n_def - (e * nw + w) * 64; n_own += left < 0 ? 0 : (left > 64 ? 64 : left); }
    for (int e = 0; e < MURRAY_EPT; e++) {
        const int idx = (e * nw + w) * 64 + lane;
        node[e] = 0; s0[e] = s1[e] = 0; pend[e] = 0; kk[e] = 1; acc[e] = 0; cv0[e] = cv1[e] = 0.5;
        if (idx < n_def) { const FlushRec r = recs[idx]; node[e] = r.node; s0[e] = r.s0; s1[e] = r.s1; pend[e] = r.pend; kk[e] = r.kk; acc[e] = r.acc; }
    }
    unsigned ready0 = 0;          // the unmarked children: radii final since the pass ended, fetched here and raised in the first round
    for (int e = 0; e < MURRAY_EPT; e++) {
        if (pend[e] & 16) { cv0[e] = rad[s0[e]]; ready0 |= 1u << (2 * e); }
        if (pend[e] & 32) { cv1[e] = rad[s1[e]]; ready0 |= 2u << (2 * e); }
        pend[e] &= ~48;
    }
    __builtin_amdgcn_wave_barrier();
    int rounds = 0, done = 0;
    auto done_word = [&](int slot) { return (unsigned)__hip_atomic_load(fl_done + (slot >> 5), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP); };
    while (done < n_own) {
        rounds++;
        unsigned ready = 0;
        {   // the done words of all waiting children first (independent LDS reads, one wait), then the finished children's radii (likewise)
            unsigned dw0[MURRAY_EPT], dw1[MURRAY_EPT];
#pragma unroll
            for (int e = 0; e < MURRAY_EPT; e++) {
                dw0[e] = (pend[e] & 1) ? done_word(s0[e]) : 0u;
                dw1[e] = (pend[e] & 2) ? done_word(s1[e]) : 0u;
            }
#pragma unroll
            for (int e = 0; e < MURRAY_EPT; e++) {
                if ((pend[e] & 1) && ((dw0[e] >> (s0[e] & 31)) & 1u)) { ready |= 1u << (2 * e); pend[e] &= ~1; }
                if ((pend[e] & 2) && ((dw1[e] >> (s1[e] & 31)) & 1u)) { ready |= 2u << (2 * e); pend[e] &= ~2; }
            }
#pragma unroll
            for (int e = 0; e < MURRAY_EPT; e++) {
                if (ready & (1u << (2 * e))) cv0[e] = *(volatile double *)(fl_val + s0[e]);
                if (ready & (2u << (2 * e))) cv1[e] = *(volatile double *)(fl_val + s1[e]);
            }
        }
        ready |= ready0;          // (first round: the unmarked children fetched above)
        ready0 = 0;
        while (__ballot(ready != 0)) {
            const bool on = ready != 0;
            const int t = on ? (int)__ffs((int)ready) - 1 : 0;
            if (on) ready &= ready - 1u;
            double x = 0.5, k = 2.0;
#pragma unroll
            for (int e = 0; e < MURRAY_EPT; e++) if ((t >> 1) == e) { x = (t & 1) ? cv1[e] : cv0[e]; k = kk[e]; }
            if (!on) { x = 0.5; k = 2.0; }
            const double pw = octa_gpow::gpow_t(x, k, log_tab, exp_tab);
#pragma unroll
            for (int e = 0; e < MURRAY_EPT; e++) if (on && (t >> 1) == e) acc[e] = acc[e] + pw;
        }
        unsigned fin = 0;
        for (int e = 0; e < MURRAY_EPT; e++) if ((pend[e] & 4) && !(pend[e] & 3)) fin |= 1u << e;
        while (__ballot(fin != 0)) {
            const bool on = fin != 0;
            const int t = on ? (int)__ffs((int)fin) - 1 : 0;
            if (on) fin &= fin - 1u;
            double a = 0.5, k = 2.0;
#pragma unroll
            for (int e = 0; e < MURRAY_EPT; e++) if (t == e) { a = acc[e]; k = kk[e]; }
            if (!on) { a = 0.5; k = 2.0; }
            const double rp = octa_gpow::gpow_t(a, 1.0 / k, log_tab, exp_tab);
#pragma unroll
            for (int e = 0; e < MURRAY_EPT; e++)
                if (on && t == e) { *(volatile double *)(fl_val + (e * nw + w) * 64 + lane) = rp; rad[node[e]] = rp; pend[e] = 8; }
        }
        __builtin_amdgcn_wave_barrier();
        bool any = false;
        for (int e = 0; e < MURRAY_EPT; e++) {
            const unsigned long long fb = __ballot(pend[e] == 8);       // slots (e * nw + w) * 64 + lane
            if (pend[e] == 8) pend[e] = 0;
            if (fb) {
                if (lane == 0) {
                    int *dw = fl_done + 2 * (e * nw + w);
                    __hip_atomic_fetch_or(dw, (int)(unsigned)fb, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
                    __hip_atomic_fetch_or(dw + 1, (int)(unsigned)(fb >> 32), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
                }
                done += (int)__popcll(fb);
                any = true;
            }
        }
        __builtin_amdgcn_wave_barrier();
        if (!any) __builtin_amdgcn_s_sleep(2);                    // waiting for another wave's slots: leave the issue port to the SIMD's other wave
    }
    return rounds;
}
#endif

OCTA_HD inline int murray_flush(const Blk &b, const SimArrays &A, int f, const SeqLds &L, int n_def) {
    if (n_def <= 0) return 0;
    int *ctl = b.coll() + 90;
    if (b.tid == 0) ctl[0] = 0;
    double *rad = L.rad;
    int rounds = 0;
#if defined(__HIP_DEVICE_COMPILE__)
    if (n_def <= MURRAY_EPT * 64) {
        // the usual case (about 135 marked nodes per pass): ONE wave runs the rounds -- its LDS accesses are ordered, so the two block
        // barriers per round are not needed and the finished count is a ballot. (phase_seq normally defers the rounds to the other
        // forest's pass; this synchronous form serves the end of a run.)
        static_assert(MURRAY_EPT * 64 <= MURRAY_FLUSH_LDS, "flush slots");
        if (b.tid < 64) {
            FlushRec *recs = A.fl_rec + (size_t)f * MURRAY_FLUSH_LDS;
            murray_flush_prepare_wave(A, f, L, n_def, recs);
            __builtin_amdgcn_wave_barrier();
            __threadfence_block();
            rounds = murray_flush_rounds_wave(rad, L.log_tab, L.exp_tab, L.fl_val, L.fl_done, recs, n_def);
        }
        b.sync();
        return rounds;
    }
#endif
    if (n_def <= MURRAY_EPT * b.nth) {
        int node[MURRAY_EPT], c0[MURRAY_EPT], c1[MURRAY_EPT];
        double kk[MURRAY_EPT], acc[MURRAY_EPT];
        int pend[MURRAY_EPT];     // 1: c0 not added yet, 2: c1 not added yet, 4: radius not written yet, 8: written in this round
                                  // (int on purpose: as an unsigned char array this state was miscomputed by hipcc 7.2 -O3 on gfx950)
        for (int e = 0; e < MURRAY_EPT; e++) {
            const int idx = b.tid + e * b.nth;
            pend[e] = 0; node[e] = 0; c0[e] = c1[e] = 0; kk[e] = 1; acc[e] = 0;
            if (idx < n_def) {
                node[e] = A.act_list[idx];
                const WalkRec r = walk_load(A, f, node[e], false);
                c0[e] = r.c0; c1[e] = r.c1; kk[e] = r.k;
                pend[e] = 4 | (r.nch >= 1 ? 1 : 0) | (r.nch >= 2 ? 2 : 0);
            }
        }
        b.sync();
        while (true) {
            rounds++;
            int fin = 0;
            for (int e = 0; e < MURRAY_EPT; e++) {
                if (!(pend[e] & 4)) continue;
                if ((pend[e] & 1) && !deferred_get(L, c0[e])) { acc[e] = acc[e] + octa_gpow::gpow_t(rad[c0[e]], kk[e], L.log_tab, L.exp_tab); pend[e] &= ~1; }
                if ((pend[e] & 2) && !deferred_get(L, c1[e])) { acc[e] = acc[e] + octa_gpow::gpow_t(rad[c1[e]], kk[e], L.log_tab, L.exp_tab); pend[e] &= ~2; }
                if (!(pend[e] & 3)) {
                    rad[node[e]] = octa_gpow::gpow_t(acc[e], 1.0 / kk[e], L.log_tab, L.exp_tab);
                    pend[e] = 8;
                    fin++;
                }
            }
            b.sync();
            for (int e = 0; e < MURRAY_EPT; e++)
                if (pend[e] == 8) { atomic_and_int(&L.deferred[node[e] >> 5], (int)~(1u << (node[e] & 31))); pend[e] = 0; }
            if (fin) atomic_add_int(&ctl[0], fin);
            b.sync();
            if (ctl[0] >= n_def) break;
        }
    } else {
        b.sync();
        while (true) {
            rounds++;
            int fin = 0;
            for (int idx = b.tid; idx < n_def; idx += b.nth) {
                const int id = A.act_list[idx];
                if (id < 0 || !deferred_get(L, id)) continue;
                const WalkRec r = walk_load(A, f, id, false);
                if ((r.nch >= 1 && deferred_get(L, r.c0)) || (r.nch >= 2 && deferred_get(L, r.c1))) continue;
                double s = 0;
                if (r.nch >= 1) s = octa_gpow::gpow_t(rad[r.c0], r.k, L.log_tab, L.exp_tab);
                if (r.nch >= 2) s = s + octa_gpow::gpow_t(rad[r.c1], r.k, L.log_tab, L.exp_tab);
                rad[id] = octa_gpow::gpow_t(s, 1.0 / r.k, L.log_tab, L.exp_tab);
                A.act_list[idx] = -1 - id;      // finished in this round: bit cleared after the sync
                fin++;
            }
            b.sync();
            for (int idx = b.tid; idx < n_def; idx += b.nth) {
                const int v = A.act_list[idx];
                if (v < 0 && v != (int)0x80000000) {
                    const int id = -1 - v;
                    atomic_and_int(&L.deferred[id >> 5], (int)~(1u << (id & 31)));
                    A.act_list[idx] = (int)0x80000000;
                }
            }
            if (fin) atomic_add_int(&ctl[0], fin);
            b.sync();
            if (ctl[0] >= n_def) break;
        }
    }
    b.sync();
    return rounds;
}

// node creation of the ordered pass: the counter, the parent's child count and the LDS mirrors are kept by
// the caller (parent_nch = children the parent has before this call)
OCTA_HD inline int seq_add_node(const SimArrays &A, int f, int &n_nodes, V3 p, double r, int parent, int parent_nch,
                                double kappa, const SeqLds &L) {
    int id = n_nodes;
    if (OCTA_UNLIKELY(id >= NCAP)) { atomic_or_int(&A.sc->err, ERR_NODE_CAP); return -1; }
    n_nodes = id + 1;
    st3(A.npos_of(f) + 3 * id, p);
    A.nrad_of(f)[id] = r; A.nkap_of(f)[id] = kappa; A.npar_of(f)[id] = parent;   // L.rad IS A.nrad_of(f)
    L.par[id] = (idx_t)parent;
    A.nch0_of(f)[id] = -1; A.nch1_of(f)[id] = -1; A.nnch_of(f)[id] = 0; A.nact_of(f)[id] = 1;
    if (parent_nch == 0) A.nch0_of(f)[parent] = id; else if (parent_nch == 1) A.nch1_of(f)[parent] = id;
    A.nnch_of(f)[parent] = (unsigned char)(parent_nch + 1);
    return id;
}

OCTA_HD inline int add_node(const SimArrays &A, int f, V3 p, double r, int parent, double kappa, double *rad_mirror = nullptr) {
    int id = A.sc->n_nodes[f];
    if (OCTA_UNLIKELY(id >= NCAP)) { atomic_or_int(&A.sc->err, ERR_NODE_CAP); return -1; }
    A.sc->n_nodes[f] = id + 1;
    st3(A.npos_of(f) + 3 * id, p);
    A.nrad_of(f)[id] = r; A.nkap_of(f)[id] = kappa; A.npar_of(f)[id] = parent;
    if (rad_mirror) rad_mirror[id] = r;
    A.nch0_of(f)[id] = -1; A.nch1_of(f)[id] = -1; A.nnch_of(f)[id] = 0; A.nact_of(f)[id] = 1;
    if (parent >= 0) {
        int c = A.nnch_of(f)[parent];
        if (c == 0) A.nch0_of(f)[parent] = id; else if (c == 1) A.nch1_of(f)[parent] = id;
        A.nnch_of(f)[parent] = (unsigned char)(c + 1);
    }
    return id;
}

// ------------------------------------------------------------------ phase: sample oxygen sinks
OCTA_HD inline void phase_sample(const Blk &b, const SimArrays &A, const SimConst &C, const IterParams &P, int iter) {
    SampleScalars *sc = A.sc;
    const int N = P.N;
    const double *cand = A.cand;
    (void)iter;
    const double en = fmax(P.eps_n, P.eps_k), es = P.eps_s;
    const double en2 = en * en;
    // thresholds of the single-precision tests, rounded AWAY from the sure side (one float spacing is ~1e-7 relative: far inside the margin)
    const float en_hi2f = nextafterf((float)((en + GRID_F_TAU) * (en + GRID_F_TAU)), INFINITY);
    const float es_lo2f = es > GRID_F_TAU ? nextafterf((float)((es - GRID_F_TAU) * (es - GRID_F_TAU)), -INFINITY) : 0.f;
    const float es_hi2f = nextafterf((float)((es + GRID_F_TAU) * (es + GRID_F_TAU)), INFINITY);
    const double GSd = C.gs;
    const double fcx = C.fc0 * GSd, fcy = C.fc1 * GSd, fr = sc->faz_radius * GSd * 0.5;
    int *vlist = A.tmp_int;               // valid candidate indices, in order
    int *plist = A.tmp_int + NCANDCAP;    // passing candidate indices, in order
#if defined(OCTA_SIM_PROF_SAMPLE) && defined(__HIP_DEVICE_COMPILE__)
    // diagnostic build: the kd slots of the phase profile hold this phase's steps: validity, grid over the arterial nodes, radius test,
    // grid over the sinks, nearest-sink test, compaction of the passers, greedy acceptance + append
    long _st = (long)wall_clock64();
#define SSP(slot) do { if (b.tid == 0) { long _t = (long)wall_clock64(); sc->kdprof[slot] += _t - _st; _st = _t; } } while (0)
#else
#define SSP(slot) do { } while (0)
#endif
    // 1. is_valid_position (simulation_space.py:89-98), ordered compaction
    for (int i = b.tid; i < N; i += b.nth) {
        V3 p = ld3(cand + 3 * i);
        int ok = !(p.x >= C.sx || p.y >= C.sy || p.z >= C.sz || p.x < 0 || p.y < 0 || p.z < 0);
        if (ok && C.fixed) {
            // geometry[(pos * geometry_size).astype(np.uint16)] > 0: the candidate came from a valid voxel, but (v + u) / gs * gs may
            // land one voxel below v
            const int vi = (int)(unsigned short)(int)(p.x * GSd), vj = (int)(unsigned short)(int)(p.y * GSd), vk = (int)(unsigned short)(int)(p.z * GSd);
            ok = vi < C.gshape[0] && vj < C.gshape[1] && vk < C.gshape[2] && C.mask[((size_t)vi * C.gshape[1] + vj) * C.gshape[2] + vk] != 0;
        } else if (ok) {
            double dd = sqrt((p.x - fcx) * (p.x - fcx) + (p.y - fcy) * (p.y - fcy));
            ok = dd > fr;
        }
        A.removed[i] = (unsigned char)ok;
    }
    b.sync();
    const int n_valid = ordered_compact(b, N, [&](int i) { return A.removed[i] != 0; }, [&](int i, int pos) { vlist[pos] = i; });
    b.sync();
    SSP(0);
    // 2. tests against all arterial nodes (ball en, oxygen distance) and existing sinks (NN <= es). The candidates are the queries:
    //    a candidate stops at the first node / sink that rejects it, and most are rejected by one of the first few. (Round 3 measured
    //    the inverted form -- grid over the <= N candidates, every node and sink visiting the cells around itself -- which writes no
    //    cell-ordered coordinates of the big point sets, but evaluates every live candidate against every node in range: 23 + 14 ms per
    //    sample against 14 + 9 ms; the inverted form is kept where the queries are few AND every hit is needed: phase_satisfy_art step 3.)
    const int n_art = sc->n_nodes[0];
    const int n_oxy = sc->n_oxy;
    unsigned char *okf = A.removed;  // per valid candidate
    double *oxd = A.tmp_dbl;         // oxygen distance per arterial node
    for (int i = b.tid; i < n_art; i += b.nth) oxd[i] = oxygen_distance(A.nrad[0][i], C.ps);
    {
        // both grids of this phase hold single-precision copies of their points (12 instead of 24 bytes per visited point, one load
        // instruction instead of two): the distance to the rounded point decides unless it lies within GRID_F_TAU of the threshold --
        // then the exact point is fetched and the reference's own expression evaluated (about one test in 10^5)
        Grid G = grid_build(b, A, A.npos[0], nullptr, n_art, en, 1, true);
        SSP(1);
        for (int rep = 0; rep < ((OCTA_SIM_DUP & 8) ? 2 : 1); rep++)
        for (int vi = b.tid; vi < n_valid; vi += b.nth) {
            const V3 c = ld3(cand + 3 * vlist[vi]);
            bool ok = true;
            const float cxf = (float)c.x, cyf = (float)c.y, czf = (float)c.z;
            grid_visit_f(G, c.x, c.y, en, [&](int j, const Pt3f &qf) {
                if (ok) {
                    const float d2f = sqdist_f(qf, cxf, cyf, czf);
                    if (d2f <= en_hi2f) {
                        const double lim = fmin(en, oxd[j]), lo = lim - GRID_F_TAU, hi = lim + GRID_F_TAU;
                        if (lo > 0.0 && (double)d2f < lo * lo) ok = false;
                        else if (OCTA_UNLIKELY((double)d2f <= hi * hi)) {
                            const double d2 = sqdist(ld3(G.src + 3 * j), c);
                            if (d2 <= en2 && !(sqrt(d2) > oxd[j])) ok = false;
                        }
                    }
                }
            });
            okf[vi] = ok ? 1 : 0;
        }
        b.sync();
        SSP(2);
    }
    {
        Grid G = grid_build(b, A, A.oxy, nullptr, n_oxy, es, 1, true);
        SSP(3);
        for (int rep = 0; rep < ((OCTA_SIM_DUP & 8) ? 2 : 1); rep++)
        for (int vi = b.tid; vi < n_valid; vi += b.nth) {
            if (!okf[vi]) continue;
            V3 c = ld3(cand + 3 * vlist[vi]);
            bool ok = true;
            const float cxf = (float)c.x, cyf = (float)c.y, czf = (float)c.z;
            grid_visit_f(G, c.x, c.y, es, [&](int j, const Pt3f &qf) {
                if (ok) {
                    const float d2f = sqdist_f(qf, cxf, cyf, czf);
                    if (d2f < es_lo2f) ok = false;
                    else if (OCTA_UNLIKELY(d2f <= es_hi2f) && sqrt(sqdist(ld3(G.src + 3 * j), c)) <= es) ok = false;
                }
            });
            okf[vi] = ok ? 1 : 0;
        }
        b.sync();
        SSP(4);
    }
    const int n_pass = ordered_compact(b, n_valid, [&](int vi) { return okf[vi] != 0; }, [&](int vi, int pos) { plist[pos] = vlist[vi]; });
    b.sync();
    SSP(5);
#if defined(OCTA_SIM_PROF_SAMPLE) && defined(__HIP_DEVICE_COMPILE__)
    if (b.tid == 0) sc->kdprof[7] += n_pass;          // (diagnostic build: passers per sample in the last slot)
#endif
    // 3. ordered greedy acceptance against the sinks accepted earlier in this call (strict >)
    double *acc = reinterpret_cast<double *>(b.user_of<4>());  // [ACCCAP][3]
    int *ctl = b.coll() + 100;
    if (b.tid == 0) { ctl[0] = 0; ctl[1] = 0; ctl[2] = 0; }
    b.sync();
#if defined(__HIP_DEVICE_COMPILE__)
    // Round 6: the greedy rule -- candidate j is accepted iff no ACCEPTED candidate i < j lies within eps_s -- only chains candidates that
    // conflict with each other, and those are few (a few dozen pairs among ~300 passers). So: every pair (i < j) of passers is tested by
    // the whole workgroup (a thread takes the rows j and n - 1 - j: n - 1 tests each, the lanes of a wave read the same earlier candidate
    // from the LDS), the conflicting pairs are listed per row, one wave settles the rows that have any in order, and an ordered compaction
    // appends the accepted ones. The one-wave loop this replaces tested every passer against all accepted sinks, one passer after the
    // other: 13 of the phase's 48 ms per sample with three waves idle. The outcome is the same set in the same order: a row without
    // conflicts is accepted whatever the others do, and a row's conflicts are settled after all rows before it.
    constexpr int GP_MAX = 1024, GPAIR_CAP = 2048;
    static_assert((size_t)GP_MAX * 24 + (size_t)GP_MAX * 4 + (size_t)GPAIR_CAP * 2 + GP_MAX / 8 <= (size_t)SIM_USER_BYTES, "greedy acceptance tables");
    bool settled = false;
    if (n_pass <= GP_MAX) {
        double *pp = acc;                                                            // [n_pass][3] the passers' positions
        unsigned short *rcnt = reinterpret_cast<unsigned short *>(pp + 3 * GP_MAX);  // [GP_MAX] conflicts of row j with rows before it
        unsigned short *rstart = rcnt + GP_MAX;                                      // [GP_MAX] where they are listed
        unsigned short *pl = rstart + GP_MAX;                                        // [GPAIR_CAP] the earlier rows, row by row
        unsigned *okb = reinterpret_cast<unsigned *>(pl + GPAIR_CAP);                // [GP_MAX / 32] accepted
        for (int q = b.tid; q < n_pass; q += b.nth) { st3(pp + 3 * q, ld3(cand + 3 * plist[q])); rcnt[q] = 0; }
        for (int w = b.tid; w < (n_pass + 31) / 32; w += b.nth) okb[w] = ~0u;
        b.sync();
        // sqrt is monotone and correctly rounded: outside a relative band of 1e-14 around eps_s^2 the squared distance decides
        const double es2 = es * es, es2_lo = es2 * (1.0 - 1e-14), es2_hi = es2 * (1.0 + 1e-14);
        auto conflicts = [&](const V3 &c, int i) {
            const V3 d = sub(c, ld3(pp + 3 * i));
            const double s2 = (d.x * d.x + d.y * d.y) + d.z * d.z;
            if (s2 > es2_hi) return false;
            if (s2 < es2_lo) return true;
            return !(sqrt(s2) > es);
        };
        for (int t = b.tid; 2 * t < n_pass; t += b.nth) {
            for (int side = 0; side < 2; side++) {
                const int j = side == 0 ? t : n_pass - 1 - t;
                if (side == 1 && j == t) break;
                const V3 c = ld3(pp + 3 * j);
                int cnt = 0;
                {   // eight earlier candidates per step, their positions fetched together: the loop is a chain of LDS round trips otherwise
                    int i = 0;
                    for (; i + 8 <= j; i += 8) {
                        V3 a[8];
#pragma unroll
                        for (int u = 0; u < 8; u++) a[u] = ld3(pp + 3 * (i + u));
#pragma unroll
                        for (int u = 0; u < 8; u++) {
                            const V3 d = sub(c, a[u]);
                            const double s2 = (d.x * d.x + d.y * d.y) + d.z * d.z;
                            if (!(s2 > es2_hi)) cnt += (s2 < es2_lo || !(sqrt(s2) > es)) ? 1 : 0;
                        }
                    }
                    for (; i < j; i++) cnt += conflicts(c, i) ? 1 : 0;
                }
                if (cnt > 0) {
                    const int at = atomic_add_int(&ctl[1], cnt);
                    if (at + cnt <= GPAIR_CAP) {
                        int w = at;
                        for (int i = 0; i < j; i++) if (conflicts(c, i)) pl[w++] = (unsigned short)i;
                        rstart[j] = (unsigned short)at; rcnt[j] = (unsigned short)cnt;
                    } else {
                        ctl[2] = 1;
                    }
                }
            }
        }
        b.sync();
        if (OCTA_UNI(ctl[2]) == 0) {
            settled = true;
            if (b.tid < 64) {
                const int lane = b.tid;
                for (int j0 = 0; j0 < n_pass; j0 += 64) {
                    unsigned long long m = __ballot(j0 + lane < n_pass && rcnt[j0 + lane] != 0);
                    while (m) {
                        const int j = j0 + (int)__ffsll((long long)m) - 1;
                        m &= m - 1ull;
                        const int s0 = rstart[j], c0 = rcnt[j];
                        bool hit = false;
                        for (int k = lane; k < c0; k += 64) { const int i = pl[s0 + k]; hit |= ((okb[i >> 5] >> (i & 31)) & 1u) != 0; }
                        if (__any(hit) && lane == 0) okb[j >> 5] &= ~(1u << (j & 31));
                        __builtin_amdgcn_wave_barrier();
                    }
                }
            }
            b.sync();
            const int room = OCAP - n_oxy < ACCCAP ? OCAP - n_oxy : ACCCAP;
            const int acc_all = ordered_compact(b, n_pass, [&](int q) { return ((okb[q >> 5] >> (q & 31)) & 1u) != 0; },
                                                [&](int q, int pos) { if (pos < room) st3(A.oxy + 3 * (size_t)(n_oxy + pos), ld3(pp + 3 * q)); });
            if (b.tid == 0) {
                if (acc_all > ACCCAP) atomic_or_int(&sc->err, ERR_ACC_CAP);
                ctl[0] = acc_all < ACCCAP ? acc_all : ACCCAP;
            }
        }
    }
    b.sync();
    if (!settled && b.tid < 64) {
        const int lane = b.tid;
        int acc_n = 0;
        for (int q0 = 0; q0 < n_pass; q0 += 64) {
            // the next 64 candidates, one per lane: two memory round trips per 64 candidates instead of two per candidate
            V3 mine = v3(0, 0, 0);
            if (q0 + lane < n_pass) mine = ld3(cand + 3 * plist[q0 + lane]);
            const int cnt = n_pass - q0 < 64 ? n_pass - q0 : 64;
            for (int jj = 0; jj < cnt; jj++) {
                const int jl = __builtin_amdgcn_readfirstlane(jj);
                const V3 c = v3(readlane_f64(mine.x, jl), readlane_f64(mine.y, jl), readlane_f64(mine.z, jl));
                bool conflict = false;
                for (int j = lane; j < acc_n; j += 64)
                    if (!(rownorm(sub(c, ld3(acc + 3 * j))) > es)) conflict = true;
                if (!__any(conflict)) {
                    if (acc_n < ACCCAP) { if (lane == 0) st3(acc + 3 * acc_n, c); }
                    else if (lane == 0) atomic_or_int(&sc->err, ERR_ACC_CAP);
                    acc_n++;
                    __builtin_amdgcn_wave_barrier();
                }
            }
        }
        if (lane == 0) ctl[0] = acc_n < ACCCAP ? acc_n : ACCCAP;
    }
#else
    {
        int acc_n = 0;
        for (int q = 0; q < n_pass; q++) {
            V3 c = ld3(cand + 3 * plist[q]);
            bool conflict = false;
            for (int j = 0; j < acc_n; j++)
                if (!(rownorm(sub(c, ld3(acc + 3 * j))) > es)) { conflict = true; break; }
            if (!conflict) {
                if (acc_n < ACCCAP) st3(acc + 3 * acc_n, c); else atomic_or_int(&sc->err, ERR_ACC_CAP);
                acc_n++;
            }
        }
        ctl[0] = acc_n < ACCCAP ? acc_n : ACCCAP;
    }
#endif
    b.sync();
    int acc_n = ctl[0];
    if (n_oxy + acc_n > OCAP) { if (b.tid == 0) atomic_or_int(&sc->err, ERR_OXY_CAP); acc_n = OCAP - n_oxy; }
#if defined(__HIP_DEVICE_COMPILE__)
    if (!settled)
#endif
    for (int j = b.tid; j < acc_n * 3; j += b.nth) A.oxy[3 * n_oxy + j] = acc[j];
    b.sync();
    if (b.tid == 0) sc->n_oxy = n_oxy + acc_n;
    b.sync();
    SSP(6);
#undef SSP
}

// ------------------------------------------------------------------ phase: nearest active node + dict order
OCTA_HD inline void phase_assign(const Blk &b, const SimArrays &A, int f, const double *att, int n_att, double delta) {
    SampleScalars *sc = A.sc;
    const int n_nodes = sc->n_nodes[f];
#if defined(OCTA_SIM_PROF_ASSIGN) && defined(__HIP_DEVICE_COMPILE__)
    // diagnostic build: the kd slots of the phase profile hold this phase's steps (both forests): active list, grid, queries, heads,
    // counts + starts, member scatter, member order
    long _at = (long)wall_clock64();
#define ASP(slot) do { if (b.tid == 0) { long _t = (long)wall_clock64(); sc->kdprof[slot] += _t - _at; _at = _t; } } while (0)
#else
#define ASP(slot) do { } while (0)
#endif
    // active node list (ascending id)
    // the grid over the ACTIVE nodes is built straight from the activity flags (no list of the active nodes: the node positions are
    // streamed once per pass, coalesced, instead of gathered through the list)
    ASP(0);
    {
        Grid G = grid_build(b, A, A.npos_of(f), nullptr, n_nodes, delta, 1, true, A.nact_of(f));
        ASP(1);
        // (Round 4 measured a pruned nearest-neighbour search on a grid of four cells per radius -- rows outwards from the query's own, a
        // row dropped or cut as soon as the best distance allows: fewer points read per query, results identical -- and it was SLOWER,
        // 48 -> 53 ms per sample: the pruning makes every row pair a dependent round trip. Fetching the next query's point while the
        // current query runs changes nothing.)
        // The nearest node is chosen on single-precision copies of the nodes: the smallest and the second smallest rounded distance are
        // kept; if they are farther apart than the rounding can move them the winner is certain, and so is "within delta" unless the
        // winner's distance lies in the band around delta (then its exact coordinates decide). Otherwise -- a near tie, including the
        // exact ties the index breaks -- the query is repeated on the exact coordinates (the nodes' own array, by item id).
        const float big = 3.0e38f;
        for (int a = b.tid; a < n_att; a += b.nth) {
            const V3 p = ld3(att + 3 * a);
            const float pxf = (float)p.x, pyf = (float)p.y, pzf = (float)p.z;
            float m1 = big, m2 = big;
            int j1 = -1;
            grid_visit_f(G, p.x, p.y, delta, [&](int j, const Pt3f &qf) {
                const float d2 = sqdist_f(qf, pxf, pyf, pzf);
                if (d2 < m1) { m2 = m1; m1 = d2; j1 = j; }
                else if (d2 < m2) m2 = d2;
            });
            int r = -1;
            if (j1 >= 0) {
                const double d1 = sqrt((double)m1);
                const double lim2 = (d1 + 2.0 * GRID_F_TAU) * (d1 + 2.0 * GRID_F_TAU);
                if (!OCTA_UNLIKELY(!((double)m2 > lim2))) {                              // the winner is certain
                    if (d1 < delta - GRID_F_TAU) r = j1;
                    else if (OCTA_UNLIKELY(d1 <= delta + GRID_F_TAU)) r = sqrt(sqdist(ld3(G.src + 3 * j1), p)) <= delta ? j1 : -1;
                } else if (OCTA_UNLIKELY(d1 <= delta + 3.0 * GRID_F_TAU)) {          // near tie of candidates that may be in range: exact arg-min (distance, then id)
                    double bd = INFINITY;
                    int best = -1;
                    grid_visit_f(G, p.x, p.y, delta, [&](int j, const Pt3f &) {
                        const double d2 = sqdist(ld3(G.src + 3 * j), p);
                        if (d2 < bd || (d2 == bd && j < best)) { bd = d2; best = j; }
                    });
                    r = (best >= 0 && sqrt(bd) <= delta) ? best : -1;
                }
            }
            A.nn[a] = r;
        }
        b.sync();
        ASP(2);
    }
    // Group bookkeeping -- dict order without a sort: attractor a heads a group iff it is the first hit of its node, so an ordered
    // compaction of the heads IS the dict order; members are scattered with a per-group cursor and each (short) member list is put
    // back into attractor order by one thread. The grid is dead, so the tables of these steps live in the table area (LDS in the
    // default build): every step is a scatter or an atomic on a few-KB table, which in HBM scratch was a 128-byte line in and out of
    // the L2 per 4-byte update (round 4: this phase was 30 % of the kernel's L2-miss traffic, its write side most of all).
    //   fa[n_nodes]: first attractor of a node, then -(group + 1);  gc / cur[n_groups]: member count / scatter cursor;
    //   srt[<= n_att]: members grouped.  fa always fits; the rest falls back to the HBM arrays when the counts are too large.
    int *fa = reinterpret_cast<int *>(b.user());
    static_assert((size_t)NCAP * 4 <= (size_t)SIM_USER_BYTES, "first-attractor table");
    for (int i = b.tid; i < n_nodes; i += b.nth) fa[i] = 0x7fffffff;
    b.sync();
    for (int a = b.tid; a < n_att; a += b.nth) { const int r = A.nn[a]; if (r >= 0) atomic_min_int(&fa[r], a); }
    b.sync();
    int n_groups = 0;
#if defined(__HIP_DEVICE_COMPILE__)
    {   // ordered compaction of the heads, a contiguous segment of the attractors per wave, 64 consecutive ones per step
        const int lane = b.tid & 63, wv = b.tid >> 6, nw = (b.nth + 63) >> 6;
        const int seg = ((n_att + nw * 64 - 1) / (nw * 64)) * 64;
        const int s0 = wv * seg < n_att ? wv * seg : n_att, s1 = s0 + seg < n_att ? s0 + seg : n_att;
        int cnt = 0;
        for (int a0 = s0; a0 < s1; a0 += 64) {
            const int a = a0 + lane;
            bool h = false;
            if (a < s1) { const int r = A.nn[a]; h = r >= 0 && fa[r] == a; }
            cnt += __popcll(__ballot(h));
        }
        int ex;
        n_groups = blk_scan(b, lane == 0 ? cnt : 0, &ex);
        int base = __builtin_amdgcn_readfirstlane(ex);
        for (int a0 = s0; a0 < s1; a0 += 64) {
            const int a = a0 + lane;
            bool h = false;
            int r = -1;
            if (a < s1) { r = A.nn[a]; h = r >= 0 && fa[r] == a; }
            const unsigned long long m = __ballot(h);
            if (h) {
                const int g = base + __popcll(m & ((1ull << lane) - 1ull));
                if (g < GCAP) { A.gnode[g] = r; fa[r] = -(g + 1); }      // (a reader of fa[r] on another lane holds a different a: no match either way)
            }
            base += __popcll(m);
        }
    }
#else
    {
        const int chunk = (n_att + b.nth - 1) / b.nth;
        const int i0 = b.tid * chunk, i1 = (i0 + chunk < n_att) ? i0 + chunk : n_att;
        int local = 0;
        for (int a = i0; a < i1; a++) { int r = A.nn[a]; local += (r >= 0 && fa[r] == a) ? 1 : 0; }
        int ex;
        n_groups = blk_scan(b, local, &ex);
        int g = ex;
        for (int a = i0; a < i1; a++) {
            int r = A.nn[a];
            if (r >= 0 && fa[r] == a) {
                if (g < GCAP) { A.gnode[g] = r; fa[r] = -(g + 1); }
                g++;
            }
        }
    }
#endif
    b.sync();
    ASP(3);
    if (n_groups > GCAP) { if (b.tid == 0) atomic_or_int(&sc->err, ERR_GROUP_CAP); n_groups = GCAP; }
    int n_sorted = 0;
    // staged: the tables are in the table area and the results are copied out; otherwise the HBM arrays themselves are the tables
    auto book = [&](int *gc, int *cur, unsigned *srt, const bool staged) __attribute__((always_inline)) {
        for (int g = b.tid; g < n_groups; g += b.nth) gc[g] = 0;
        b.sync();
        for (int a = b.tid; a < n_att; a += b.nth) {
            const int r = A.nn[a];
            if (r >= 0) { const int v = fa[r]; if (v < 0) atomic_add_int(&gc[-v - 1], 1); }
        }
        b.sync();
        {
            const int chunk = ((n_groups + b.nth - 1) / b.nth) | 1;      // odd: distinct LDS banks across the lanes
            const int g0 = b.tid * chunk, g1 = (g0 + chunk < n_groups) ? g0 + chunk : n_groups;
            int local = 0;
            for (int g = g0; g < g1; g++) local += gc[g];
            int ex;
            n_sorted = blk_scan(b, local, &ex);
            int run = ex;
            for (int g = g0; g < g1; g++) { const int c = gc[g]; A.gstart[g] = run; cur[g] = run; if (staged) A.gcount[g] = c; run += c; }
        }
        b.sync();
        ASP(4);
        if (n_sorted > SORTCAP) { if (b.tid == 0) atomic_or_int(&sc->err, ERR_OXY_CAP); }
        for (int a = b.tid; a < n_att; a += b.nth) {
            const int r = A.nn[a];
            if (r >= 0) {
                const int v = fa[r];
                if (v < 0) { const int pos = atomic_add_int(&cur[-v - 1], 1); if (pos < SORTCAP) srt[pos] = (unsigned)a; }
            }
        }
        b.sync();
        ASP(5);
        for (int g = b.tid; g < n_groups; g += b.nth) {
            const int cnt = gc[g];
            unsigned *v = srt + (cur[g] - cnt);      // the cursor ended at the group's end
            for (int i = 1; i < cnt; i++) {
                unsigned x = v[i];
                int j = i - 1;
                while (j >= 0 && v[j] > x) { v[j + 1] = v[j]; j--; }
                v[j + 1] = x;
            }
        }
        if (staged) {
            b.sync();
            const int n_out = n_sorted < SORTCAP ? n_sorted : SORTCAP;
            for (int k = b.tid; k < n_out; k += b.nth) A.sorted[k] = srt[k];
        }
    };
#ifdef OCTA_SIM_ASSIGN_FORCE_HBM
    const bool fits = false;
#else
    const bool fits = (size_t)4 * ((size_t)n_nodes + 2 * (size_t)n_groups + (size_t)n_att) <= (size_t)SIM_USER_BYTES;
#endif
    if (fits) book(fa + n_nodes, fa + n_nodes + n_groups, reinterpret_cast<unsigned *>(fa + n_nodes + 2 * n_groups), true);
    else book(A.gcount, A.tmp_int, A.sorted, false);
    if (b.tid == 0) { sc->n_groups[f] = n_groups; sc->n_sorted[f] = n_sorted; }
    b.sync();
    ASP(6);
#undef ASP
}

// ------------------------------------------------------------------ per-node growth geometry
struct GrowCtx {
    const SimArrays *A;
    const SimConst *C;
    const IterParams *P;
    int f;
    const double *att;
    double gamma;
    const double *rad;  // radii of forest f: HBM array, or the LDS copy during the ordered pass
    const double *log_tab = octa_gpow::LOG_TAB;      // glibc pow tables: the constant arrays, or the ordered pass's LDS copies (round 6: a
    const uint64_t *exp_tab = octa_gpow::EXP_TAB;    // re-speculation's eight pow evaluations each made two dependent trips to the global tables)
    // the sprout's radius r_2 = r and kappa are constants of a pass (greenhouse.py:262): its three powers are evaluated once per pass, not per group
    double r2k = 0, r24 = 0, r22 = 0;
    OCTA_HD void init_powers() {
        r2k = octa_gpow::gpow_t(C->r, P->kappa, log_tab, exp_tab);
        r24 = octa_gpow::gpow_t(C->r, 4.0, log_tab, exp_tab);
        r22 = octa_gpow::gpow_t(C->r, 2.0, log_tab, exp_tab);
    }
};

// acos / cos / sin whose results reach a node position: glibc's, bit for bit (glibc_trig.h), inside the restated domain
OCTA_HD inline double pos_acos(double c) { if (OCTA_UNLIKELY(!(c > 0.0 && c <= 1.0))) return acos(c); return octa_gtrig::gacos(c); }      // (the library fallbacks are cold code: out of the hot path's cache lines)
OCTA_HD inline double pos_cos(double x) { if (OCTA_UNLIKELY(!(x >= 0.0 && x < 2.4))) return cos(x); return octa_gtrig::gcos(x); }
OCTA_HD inline double pos_sin(double x) { if (OCTA_UNLIKELY(!(x >= 0.0 && x < 2.4))) return sin(x); return octa_gtrig::gsin(x); }
// The attractors of one group, in attractor order: body(position). A visit is two dependent loads (the sorted key, then the
// attractor it names) in front of a few hundred cycles of arithmetic (angles, acos); one at a time the speculation was a chain of
// ~100-250 such round trips per thread and iteration (round 4: pre_art + pre_ven 61 ms per sample). Here the keys run two batches ahead
// of the arithmetic and the positions one batch ahead, ATT_B attractors per batch; the order of the visits is unchanged.
constexpr int ATT_B = 4;
template <class F>
OCTA_HD inline void for_each_attractor(const SimArrays &A, const double *att, int s, int cnt, F &&body) {
    if (cnt <= 0) return;
    int a1[ATT_B] = {}, a2[ATT_B] = {};
    V3 p0[ATT_B] = {}, p1[ATT_B] = {};
    auto keys = [&](int k0, int *a) {
#pragma unroll
        for (int u = 0; u < ATT_B; u++) a[u] = (int)(A.sorted[s + (k0 + u < cnt ? k0 + u : cnt - 1)] & IDX_MASK);
    };
    auto points = [&](const int *a, V3 *p) {
#pragma unroll
        for (int u = 0; u < ATT_B; u++) p[u] = ld3(att + 3 * a[u]);
    };
    keys(0, a1);
    points(a1, p0);
    keys(ATT_B, a1);
    for (int k0 = 0; k0 < cnt; k0 += ATT_B) {
        if (k0 + ATT_B < cnt) points(a1, p1);           // batch k0 + ATT_B: its keys arrived one batch ago
        if (k0 + 2 * ATT_B < cnt) keys(k0 + 2 * ATT_B, a2);
#pragma unroll
        for (int u = 0; u < ATT_B; u++)
            if (k0 + u < cnt) body(p0[u]);
#pragma unroll
        for (int u = 0; u < ATT_B; u++) { p0[u] = p1[u]; a1[u] = a2[u]; }
    }
}

// inter-node sprouting (greenhouse.py:259-306) for group g with the CURRENT child radius
// WAVE_COOP: the caller is a whole wave executing uniformly (the ordered pass on the device): the attractor loop is shared by its lanes.
// A compile-time choice (round 4): as a run-time flag both loops were compiled into the ordered pass.
template <bool WAVE_COOP>
OCTA_HD inline void eval_inter(const GrowCtx &G, int g, Rec &R) {
    const SimArrays &A = *G.A;
    // the ordered pass re-speculates with the group's record in hand: node and child come from it, not from two more dependent loads
    const int f = G.f, id = WAVE_COOP ? R.node : A.gnode[g];
    const double kappa = G.P->kappa, r = G.C->r, gamma = G.gamma, omega = G.P->omega, d = G.P->d;
    const int ch = WAVE_COOP ? (int)R.child : A.nch0_of(f)[id];
    R.type = 3; R.grow = 0; R.draw = 0; R.req = -1; R.node = id;
    const V3 pos = ld3(A.npos_of(f) + 3 * id);
    const double r1 = G.rad[ch], r2 = r;
    R.r1_used = r1;
    R.child = (idx_t)ch;
    const double *lt = G.log_tab;
    const uint64_t *et = G.exp_tab;
    auto gpow = [lt, et](double x, double y) { return octa_gpow::gpow_t(x, y, lt, et); };
    (void)r2;
    double rp = gpow(gpow(r1, kappa) + G.r2k, 1 / kappa);
    double rp4, rp2, r14, r12;
#if defined(__HIP_DEVICE_COMPILE__)
    if (WAVE_COOP) {
        // the four remaining powers are independent: four lanes evaluate one each (one pow latency instead of four in a row; the wave has
        // nothing else to do -- every lane of the ordered pass computes the same values)
        const int lane = (int)(threadIdx.x & 63);
        const double v = gpow((lane & 2) ? r1 : rp, (lane & 1) ? 2.0 : 4.0);
        rp4 = readlane_f64(v, 0); rp2 = readlane_f64(v, 1); r14 = readlane_f64(v, 2); r12 = readlane_f64(v, 3);
    } else
#endif
    { rp4 = gpow(rp, 4.0); rp2 = gpow(rp, 2.0); r14 = gpow(r1, 4.0); r12 = gpow(r1, 2.0); }
    double phi1 = acos((rp4 + r14 - G.r24) / (2 * rp2 * r12)) * rad2deg();
    // phi_2 and the rotation built from it reach the node position (phi_1 only enters angle windows): glibc's values (glibc_trig.h)
    double phi2 = pos_acos((rp4 + G.r24 - r14) / (2 * rp2 * G.r22)) * rad2deg();
    V3 dist_seg = sub(ld3(A.npos_of(f) + 3 * ch), pos);
    V3 prox_seg = sub(pos, ld3(A.npos_of(f) + 3 * A.npar_of(f)[id]));
    double nd = norm3(dist_seg), npx = norm3(prox_seg);
    double lo = phi1 + phi2 - gamma / 2, hi = phi1 + phi2 + gamma / 2, pl = phi2 + gamma / 2;
    V3 avg = v3(0, 0, 0);
    int kept = 0;
    const int s = A.gstart[g], cnt = A.gcount[g];
#if defined(__HIP_DEVICE_COMPILE__)
    if (WAVE_COOP) {
        // re-speculation inside the ordered pass: the 64 lanes take one attractor each (angles, unit vector), the kept unit vectors
        // are then summed in attractor order with lane reads -- the same operations in the same order as the loop below
        const int lane = (int)(threadIdx.x & 63);
        for (int k0 = 0; k0 < cnt; k0 += 64) {
            const int k = k0 + lane;
            bool keep = false;
            V3 u = v3(0, 0, 0);
            if (k < cnt) {
                const int a = (int)(A.sorted[s + k] & IDX_MASK);
                const V3 w = sub(ld3(G.att + 3 * a), pos);
                const double ad = angle_uv(dist_seg, nd, w), ap = angle_uv(prox_seg, npx, w);
                keep = lo <= ad && ad <= hi && ap <= pl;
                if (keep) u = unit(w);
            }
            unsigned long long m = __ballot(keep);
            while (m) {
                const int j = (int)__ffsll((long long)m) - 1;
                m &= m - 1ull;
                const V3 uj = v3(readlane_f64(u.x, j), readlane_f64(u.y, j), readlane_f64(u.z, j));
                avg = kept == 0 ? uj : add(avg, uj);
                kept++;
            }
        }
    } else
#endif
    for_each_attractor(A, G.att, s, cnt, [&](const V3 &ap_) {
        V3 w = sub(ap_, pos);
        double ad = angle_uv(dist_seg, nd, w), ap = angle_uv(prox_seg, npx, w);
        if (lo <= ad && ad <= hi && ap <= pl) {
            V3 u = unit(w);
            avg = kept == 0 ? u : add(avg, u);
            kept++;
        }
    });
    if (kept == 0) return;
    V3 dv = unit(dist_seg);
    V3 cr = cross(dv, avg);
    if (cr.x == 0 && cr.y == 0 && cr.z == 0) return;
    const double vc0 = G.C->fc0 - pos.x, vc1 = G.C->fc1 - pos.y;
    R.grow = 1; R.draw = 1;
    R.thr = gpow(norm2(vc0, vc1) / (2 * A.sc->faz_radius), 5.0);
    R.ang_gt90 = angle2(vc0, vc1, avg.x, avg.y) > 90 ? 1 : 0;
    V3 k = unit(cr);
    double th = phi2 * deg2rad();
    double ct = pos_cos(th), st = pos_sin(th);
    V3 kxd = cross(k, dv);
    V3 vrot = add(add(mul(dv, ct), mul(kxd, st)), mul(mul(k, dot3_blas(k, dv)), 1 - ct));
    V3 gg = add(mul(unit(vrot), omega), mul(unit(avg), 1 - omega));
    st3(R.newpos, add(pos, mul(unit(gg), d)));
}

// leaf growth (greenhouse.py:177-258); bifurcation candidates ship a request to the host
OCTA_HD inline void eval_leaf(const GrowCtx &G, int g, Rec &R, BifRequest *reqs, int *req_count, int req_cap, int sample) {
    const SimArrays &A = *G.A;
    const int f = G.f, id = A.gnode[g];
    const double r = G.C->r, gamma = G.gamma, omega = G.P->omega, d = G.P->d, kappa = G.P->kappa;
    R.type = 0; R.grow = 0; R.draw = 0; R.req = -1; R.node = id; R.r1_used = 0;
    const V3 pos = ld3(A.npos_of(f) + 3 * id);
    V3 v = sub(pos, ld3(A.npos_of(f) + 3 * A.npar_of(f)[id]));
    double nv = norm3(v);
    double lim = fmax(gamma / 2, 0.0);
    V3 avg = v3(0, 0, 0);
    int kept = 0;
    double sum = 0;
    const int s = A.gstart[g], cnt = A.gcount[g];
    for_each_attractor(A, G.att, s, cnt, [&](const V3 &ap_) {
        V3 w = sub(ap_, pos);
        double an = angle_uv(v, nv, w);
        if (an <= lim) {
            V3 u = unit(w);
            avg = kept == 0 ? u : add(avg, u);
            sum += an;
            kept++;
        }
    });
    if (kept == 0) return;
    double mean = sum / (double)kept, var = 0;
    for_each_attractor(A, G.att, s, cnt, [&](const V3 &ap_) {
        double an = angle_uv(v, nv, sub(ap_, pos));
        if (an <= lim) var += (an - mean) * (an - mean);
    });
    double sd = sqrt(var / (double)kept);
    const double vc0 = G.C->fc0 - pos.x, vc1 = G.C->fc1 - pos.y;
    R.type = 1;
    if (sd > G.P->phi) {
        R.draw = 1;
        R.thr = octa_gpow::gpow(norm2(vc0, vc1) / (2 * A.sc->faz_radius), 5.0);
        R.ang_gt90 = angle2(vc0, vc1, avg.x, avg.y) > 90 ? 1 : 0;
        if (R.ang_gt90) {
            int q = atomic_add_int(req_count, 1);
            if (q < req_cap && kept <= MAXKEPT) {
                BifRequest &Q = reqs[q];
                Q.sample = sample; Q.n = kept;
                st3(Q.pos, pos);
                Q.r = r; Q.kappa = kappa; Q.d = d;
                int w = 0;
                for_each_attractor(A, G.att, s, cnt, [&](const V3 &p) {
                    if (angle_uv(v, nv, sub(p, pos)) <= lim) { st3(Q.atts + 3 * w, p); w++; }
                });
                R.req = q;
            } else {
                atomic_or_int(&A.sc->err, kept > MAXKEPT ? ERR_KEPT_CAP : ERR_REQ_CAP);
            }
        }
    }
    // elongation (used unless the node bifurcates)
    V3 gg = add(mul(unit(v), omega), mul(unit(avg), 1 - omega));
    if (G.C->rotation_radius > 0 && G.P->t > 15) {
        gg = unit(gg);
        double cn = norm2(vc0, vc1);
        double cv0 = vc0 / cn, cv1 = vc1 / cn;
        V3 np_ = add(pos, mul(gg, d));
        double dist_new = norm2(G.C->fc0 - np_.x, G.C->fc1 - np_.y);
        double weight = fmax(G.P->first_mode ? 0.0 : 0.01, G.C->rotation_radius - dist_new);
        weight = sqrt(weight);
        V3 ort = v3(-cv1, cv0, 0);
        if (angle2(gg.x, gg.y, ort.x, ort.y) > 90) ort = mul(ort, -1.0);
        V3 outv = v3(-cv0, -cv1, 0);
        gg = add(add(mul(gg, 1 - weight), mul(ort, 0.7 * weight)), mul(outv, 0.3 * weight));
    }
    st3(R.newpos, add(pos, mul(unit(gg), d)));
}

// parallel speculation over all groups of forest f
OCTA_HD inline void phase_pre(const Blk &b, const SimArrays &A, const SimConst &C, const IterParams &P, int f,
                              const double *att, BifRequest *reqs, int *req_count, int req_cap, int sample) {
    GrowCtx G = {&A, &C, &P, f, att, f == 0 ? P.gamma_art : P.gamma_ven, A.nrad_of(f)};
    G.init_powers();
    const int ng = A.sc->n_groups[f];
    if (b.tid == 0) { A.sc->pass_counter++; A.sc->pass_tag[f] = A.sc->pass_counter; }
    b.sync();
    const int tag = A.sc->pass_tag[f];
    int n_grow = 0;
    // Leaves and inter-nodes are evaluated by different code of a few thousand instructions each, and the groups come in dict order,
    // i.e. mixed: taken 64 consecutive groups at a time EVERY wave ran both evaluations one after the other (a full-length sample:
    // ~3000 leaf and ~600 inter-node groups per pass towards the end, 57 of 57 waves mixed). The groups are therefore split by kind
    // first (two id lists in the table area, order irrelevant), each list is evaluated by whole waves, and the list of the groups
    // that grow -- which has to be in group order -- comes from one ordered compaction of a flag per group afterwards.
    // ... and inside a kind by the number of attractors (16 buckets; a counting sort in the table area): the attractor loops of a
    // wave run as long as its longest group, and the counts range from 1 to ~10 around a mean of 2.
    idx_t *leaf = reinterpret_cast<idx_t *>(b.user());            // [n_leaf] leaves, then [n_inter] inter-nodes behind them
    idx_t *posv = leaf + GCAP;                                    // [ng] rank of a group inside its bucket
    unsigned char *keyv = reinterpret_cast<unsigned char *>(posv + GCAP);      // [ng] bucket: kind * 16 + min(count, 15); 32 = neither
    unsigned char *growf = keyv + GCAP;                           // [ng] the group grows
    static_assert((size_t)2 * GCAP * sizeof(idx_t) + (size_t)2 * GCAP <= (size_t)SIM_USER_BYTES, "speculation lists");
    int *cnt = b.coll() + 110;       // [32] bucket sizes, then bucket starts
    static_assert(110 + 32 <= 396, "bucket counters in the collectives area");
    for (int k = b.tid; k < 32; k += b.nth) cnt[k] = 0;
    b.sync();
    for (int g = b.tid; g < ng; g += b.nth) {
        const int id = A.gnode[g];
        const int nch = A.nnch_of(f)[id], par = A.npar_of(f)[id];
        const int cls = nch == 0 ? 0 : ((par >= 0 && nch == 1) ? 1 : 2);
        int key = 32;
        if (cls == 2) {
            Rec R;
            memset(&R, 0, sizeof(R));
            R.type = 0; R.node = id; R.req = -1;
            A.rec[g] = R;
        } else {
            const int c = A.gcount[g];
            key = cls * 16 + (c < 15 ? c : 15);
            posv[g] = (idx_t)atomic_add_int(&cnt[key], 1);
        }
        keyv[g] = (unsigned char)key;
        growf[g] = 0;
    }
    b.sync();
    const int n_leaf = [&] { int t = 0; for (int k = 0; k < 16; k++) t += cnt[k]; return t; }();
    const int n_inter = [&] { int t = 0; for (int k = 16; k < 32; k++) t += cnt[k]; return t; }();
    b.sync();
    if (b.tid == 0) { int run = 0; for (int k = 0; k < 32; k++) { const int c = cnt[k]; cnt[k] = run; run += c; } }
    b.sync();
    for (int g = b.tid; g < ng; g += b.nth) {
        const int key = keyv[g];
        if (key < 32) leaf[cnt[key] + (int)posv[g]] = (idx_t)g;
    }
    b.sync();
    const idx_t *inter = leaf + n_leaf;
    for (int k = b.tid; k < n_leaf; k += b.nth) {
        const int g = (int)leaf[k];
        Rec R;
        memset(&R, 0, sizeof(R));
        eval_leaf(G, g, R, reqs, req_count, req_cap, sample);
        A.rec[g] = R;
        growf[g] = R.type == 1 ? 1 : 0;
    }
    for (int k = b.tid; k < n_inter; k += b.nth) {
        const int g = (int)inter[k];
        Rec R;
        memset(&R, 0, sizeof(R));
        eval_inter<false>(G, g, R);
        A.rec[g] = R;
        growf[g] = (R.type == 3 && R.grow) ? 1 : 0;
        if (R.type == 3) A.child_group[A.nch0_of(f)[A.gnode[g]]] = (tag << (GROUP_BITS + 1)) | ((int)R.grow << GROUP_BITS) | g;
    }
    b.sync();
#if defined(__HIP_DEVICE_COMPILE__)
    {   // ordered compaction of the flags: a contiguous segment of the groups per wave, 64 consecutive ones per step
        const int lane = b.tid & 63, wv = b.tid >> 6, nw = (b.nth + 63) >> 6;
        const int seg = ((ng + nw * 64 - 1) / (nw * 64)) * 64;
        const int s0 = wv * seg < ng ? wv * seg : ng, s1 = s0 + seg < ng ? s0 + seg : ng;
        int c = 0;
        for (int g0 = s0; g0 < s1; g0 += 64) c += (int)__popcll(__ballot(g0 + lane < s1 && growf[g0 + lane]));
        int ex;
        n_grow = blk_scan(b, lane == 0 ? c : 0, &ex);
        int at = __builtin_amdgcn_readfirstlane(ex);
        for (int g0 = s0; g0 < s1; g0 += 64) {
            const bool gr = g0 + lane < s1 && growf[g0 + lane];
            const unsigned long long m = __ballot(gr);
            if (gr) A.glist[at + (int)__popcll(m & ((1ull << lane) - 1ull))] = g0 + lane;
            at += (int)__popcll(m);
        }
    }
#else
    for (int g = 0; g < ng; g++) if (growf[g]) A.glist[n_grow++] = g;
#endif
    if (b.tid == 0) A.sc->n_grow[f] = n_grow;
    b.sync();
}

// ordered pass (one thread): RNG draws, node creation, Murray propagation, deactivation.
// Only the groups that grow under the speculation are visited, plus the (rare) inter-nodes whose child
// radius an earlier node of this pass changed (dirty list fed by murray_to_root); this visits exactly the
// groups for which the reference's sequential loop does anything.
// The ordered pass consumes two streams whose order is known when it starts: the records of the groups that grow (A.glist /
// A.rec, read-only during the pass) and the pre-generated random.uniform draws. One wave runs the pass, so both are fetched 64
// entries at a time, one per lane (two memory round trips per 64 visits instead of one dependent round trip per visit), and
// handed out with v_readlane. Host build: plain loads.
struct GrowWindow {
    const SimArrays *A;
    int n_grow, base, g_mine;
    Rec mine;                  // record of group g_mine (typed loads and typed lane reads: no aliasing through int*)
    OCTA_HD inline void init(const SimArrays *A_, int n) { A = A_; n_grow = n; base = -64; g_mine = 0x7fffffff; }
#if defined(__HIP_DEVICE_COMPILE__)
    __device__ inline void refill(int from) {
        base = from;
        const int k = from + (int)(threadIdx.x & 63);
        g_mine = 0x7fffffff;
        if (k < n_grow) {
            g_mine = A->glist[k];
            mine = A->rec[g_mine];
        }
    }
    __device__ inline int group(int gi) {
        if (gi >= n_grow) return 0x7fffffff;
        if (gi >= base + 64) refill(gi);
        return __builtin_amdgcn_readlane(g_mine, __builtin_amdgcn_readfirstlane(gi - base));
    }
    static __device__ inline double lane_f64(double v, int j) {
        return __hiloint2double(__builtin_amdgcn_readlane(__double2hiint(v), j), __builtin_amdgcn_readlane(__double2loint(v), j));
    }
    __device__ inline Rec record(int gi) const {     // after group(gi)
        const int j = __builtin_amdgcn_readfirstlane(gi - base);
        Rec R;
        R.newpos[0] = lane_f64(mine.newpos[0], j); R.newpos[1] = lane_f64(mine.newpos[1], j); R.newpos[2] = lane_f64(mine.newpos[2], j);
        R.thr = lane_f64(mine.thr, j);
        R.r1_used = lane_f64(mine.r1_used, j);
        R.node = __builtin_amdgcn_readlane(mine.node, j);
        R.req = __builtin_amdgcn_readlane(mine.req, j);
        const int flags = __builtin_amdgcn_readlane((int)mine.type | ((int)mine.draw << 8) | ((int)mine.ang_gt90 << 16) | ((int)mine.grow << 24), j);
        R.type = (unsigned char)(flags & 255); R.draw = (unsigned char)((flags >> 8) & 255);
        R.ang_gt90 = (unsigned char)((flags >> 16) & 255); R.grow = (unsigned char)((flags >> 24) & 255);
        R.child = (unsigned)__builtin_amdgcn_readlane((int)mine.child, j);
        return R;
    }
#else
    inline int group(int gi) { return gi < n_grow ? A->glist[gi] : 0x7fffffff; }
    inline Rec record(int gi) const { return A->rec[A->glist[gi]]; }
#endif
};
struct UniformWindow {
    const double *u;
    int cap, base;
    double mine;
    OCTA_HD inline void init(const double *u_, int cap_) { u = u_; cap = cap_; base = -64; mine = 0; }
    OCTA_HD inline double at(int pos) {            // pos < cap
#if defined(__HIP_DEVICE_COMPILE__)
        if (pos >= base + 64 || pos < base) {
            base = pos;
            const int k = pos + (int)(threadIdx.x & 63);
            mine = k < cap ? u[k] : 0.0;
        }
        const int j = __builtin_amdgcn_readfirstlane(pos - base);
        return __hiloint2double(__builtin_amdgcn_readlane(__double2hiint(mine), j), __builtin_amdgcn_readlane(__double2loint(mine), j));
#else
        return u[pos];
#endif
    }
};

#if defined(__HIP_DEVICE_COMPILE__)
#define OCTA_FLUSH_T0() ((long)wall_clock64())
#else
#define OCTA_FLUSH_T0() 0L
#endif
struct NoSideJob { OCTA_HD void operator()(unsigned char *) const {} };
constexpr int SEQ_SIDE_LDS = 7680;   // bytes of LDS handed to the side job of the ordered pass

// side: work that does not depend on the ordered pass, run by the SECOND wave while the first one is busy with it
// (device only; e.g. the candidate stream of the next iteration). It gets SEQ_SIDE_LDS bytes of LDS.
template <class Side = NoSideJob>
OCTA_HD inline void phase_seq(const Blk &b, const SimArrays &A, const SimConst &C, const IterParams &P, int f,
                              const double *att, const double *bif_results /* [req][6] */, Side side = Side()) {
    SampleScalars *sc = A.sc;
    // LDS for the duration of the pass (44 KiB): parents (NCAP u16) of this forest, the pow tables, two bitmaps, the side job's area.
    // The radii stay where they are, in HBM / L2 (round 3; an LDS copy of NCAP doubles alone is 112 KiB): the pass's wave reads a
    // radius only inside a Murray walk (lane-parallel, one more round trip per 64 ancestors) and when an inter-node is re-speculated;
    // "has this group's child radius changed since the speculation" is answered by the `changed` bitmap the walks keep.
#if defined(OCTA_SIM_PROF_SEQ) && defined(__HIP_DEVICE_COMPILE__)
    // diagnostic build: kd slots 0..6 = set-up, the wave's loop, walks, flush (ticks), visits, walks, deferred nodes (counts)
    const long _q0 = (long)wall_clock64();
#endif
    SeqLds L;
    L.rad = A.nrad_of(f);
    constexpr int DEF_WORDS = (NCAP + 31) / 32, CHG_WORDS = (GCAP + 31) / 32;     // the two bitmaps
    L.par = reinterpret_cast<idx_t *>(b.user_of<8>());
    double *ltab = reinterpret_cast<double *>(b.user_of<8>() + (((size_t)NCAP * sizeof(idx_t) + 15) & ~(size_t)15));
    uint64_t *etab = reinterpret_cast<uint64_t *>(ltab + 384);
    constexpr int FLD_WORDS = MURRAY_FLUSH_LDS / 32;
    static_assert((size_t)NCAP * sizeof(idx_t) + 16 + 384 * 8 + 256 * 8 + (DEF_WORDS + CHG_WORDS + FLD_WORDS) * 4 + MURRAY_FLUSH_LDS * 8 + SEQ_SIDE_LDS <= (size_t)SIM_USER_BYTES, "ordered-pass table layout");
    static_assert((DEF_WORDS + CHG_WORDS + FLD_WORDS) % 2 == 0, "fl_val is 8-byte aligned");
    L.log_tab = ltab; L.exp_tab = etab;
    L.deferred = reinterpret_cast<int *>(etab + 256);
    L.changed = L.deferred + DEF_WORDS;
    L.fl_done = L.changed + CHG_WORDS;
    L.fl_val = reinterpret_cast<double *>(L.fl_done + FLD_WORDS);
    L.slot_of = A.tmp_int;                 // [NCAP] of the general scratch: free between the assignment and the satisfaction steps
    static_assert(NCAP <= OCAP + 2 * NCANDCAP, "slot_of fits the scratch");
    unsigned char *side_lds = reinterpret_cast<unsigned char *>(L.fl_val + MURRAY_FLUSH_LDS);
    const int n_before = sc->n_nodes[f];
    const int tag_now = sc->pass_tag[f];
    // node words into the LDS: the parent, and whether the node carries this pass's tag (phase_pre tagged the child of every inter-node group)
    for (int i = b.tid; i < n_before; i += b.nth) {
        const int p = A.npar_of(f)[i];
        const bool tg = (A.child_group[i] >> (GROUP_BITS + 1)) == tag_now;
        L.par[i] = (idx_t)((p < 0 ? PAR_MASK : (unsigned)p) | (tg ? PAR_TAG : 0u));
    }
    for (int i = b.tid; i < 384; i += b.nth) ltab[i] = octa_gpow::LOG_TAB[i];
    for (int i = b.tid; i < 256; i += b.nth) etab[i] = octa_gpow::EXP_TAB[i];
    for (int i = b.tid; i < DEF_WORDS + CHG_WORDS + FLD_WORDS; i += b.nth) L.deferred[i] = 0;      // both bitmaps and the flush's done bits (adjacent)
    if (b.tid == 0) b.coll()[91] = 0;
    b.sync();
#if defined(OCTA_SIM_PROF_SEQ) && defined(__HIP_DEVICE_COMPILE__)
    const long _q1 = (long)wall_clock64();
    if (b.tid == 0) sc->kdprof[0] += _q1 - _q0;
#endif
#if defined(__HIP_DEVICE_COMPILE__)
#define SEQT(slot, stmt) do { long _t0 = (long)wall_clock64(); stmt; t_acc[slot] += (long)wall_clock64() - _t0; } while (0)
#else
#define SEQT(slot, stmt) do { stmt; } while (0)
#endif
    // deferred Murray flush (round 6): the rounds of the OTHER forest's last pass run here as a side job (third wave); this pass's own marked
    // nodes are only prepared at its end and evaluated beside the other forest's next pass. Nothing reads a forest's upper radii in between:
    // arterial radii are read by phase_sample / phase_pre of the next iteration (behind the venous pass), venous radii by the next venous
    // phase_pre (behind the arterial pass).
#if defined(__HIP_DEVICE_COMPILE__)
    const int pend_other = OCTA_UNI(sc->fl_pending[1 - f]), pend_own = OCTA_UNI(sc->fl_pending[f]);
#endif
    long t_acc[4] = {0, 0, 0, 0};  // murray, re-speculation, visits, -
#if defined(__HIP_DEVICE_COMPILE__)
    if (pend_own > 0) {       // (the passes alternate between the forests, so this forest's last flush has run beside the other's pass; kept for callers that break the order)
        const int wv = b.tid >> 6;
        if (wv < 3 && wv * 64 < pend_own)
            murray_flush_rounds_wave(L.rad, L.log_tab, L.exp_tab, L.fl_val, L.fl_done, A.fl_rec + (size_t)f * MURRAY_FLUSH_LDS, pend_own, wv, 3, true);
        b.sync();
        if (b.tid == 0) sc->fl_pending[f] = 0;
        for (int i = b.tid; i < FLD_WORDS; i += b.nth) L.fl_done[i] = 0;
        b.sync();
    }
#endif
#if defined(OCTA_SIM_PROF_SEQ2) && defined(__HIP_DEVICE_COMPILE__)
    // second diagnostic build of the pass: kd slots 0..6 = ticks in the walks' chain enumeration (tag test included), in leaf bodies, in inter-node
    // bodies (walks excluded), in re-speculations, in the walks' eager parts; walks with an eager part, ancestors enumerated (counts)
    long q2[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    long _v0 = 0, _v1 = 0;
#define SEQ2_TOP() do { _v0 = (long)wall_clock64(); } while (0)
#define SEQ2_SEL() do { _v1 = (long)wall_clock64(); (void)_v0; } while (0)
#define SEQ2_END(slot, cnt, walk_before) do { q2[slot] += (long)wall_clock64() - _v1 - (t_acc[0] - (walk_before)); (void)(cnt); } while (0)
#define SEQ2_ARG , q2
#else
#define SEQ2_TOP() do { } while (0)
#define SEQ2_SEL() do { } while (0)
#define SEQ2_END(slot, cnt, walk_before) do { } while (0)
#define SEQ2_ARG
#endif
    // The first wave runs the pass with all lanes doing the same thing (same values, same stores); the lanes
    // only differ inside murray_to_root. One thread alone on the host build.
    if (b.tid < (b.nth >= 64 ? 64 : 1)) {
        GrowCtx G = {&A, &C, &P, f, att, f == 0 ? P.gamma_art : P.gamma_ven, L.rad, L.log_tab, L.exp_tab};
        G.init_powers();
        const int ng = sc->n_groups[f];
        const int n_grow = sc->n_grow[f];
        const int tag = sc->pass_tag[f];
        DirtyList D;
        D.v = b.coll() + 128; D.n = 0; D.cap = 256; D.overflow = false;
        // scalars of the sample stay in registers during the pass
        int n_nodes = n_before, py_pos = sc->py_pos, err = 0;
        const int py_cap = sc->py_cap;
        long steps = 0, n_bif = 0, respec = 0;
        int n_def = 0;           // nodes on the deferred list (A.act_list) of this pass
        UniformWindow U;
        U.init(A.py_u, py_cap);
        GrowWindow W;
        W.init(&A, n_grow);
        const int INF = 0x7fffffff;
        int gi = 0;
        int last_g = -1;
        int d_head = INF;
        bool scan_all = false;  // fallback when the dirty list overflows: visit every remaining group
        while (true) {
            int g;
            Rec R;
            SEQ2_TOP();
            if (OCTA_UNLIKELY(scan_all)) {
                g = last_g + 1;
                if (g >= ng) break;
                R = A.rec[g];
            } else {
                const int g1 = W.group(gi);
                const int g2 = d_head;            // head of the dirty list, kept in a register: it only changes in a walk or when it is taken
                g = g1 < g2 ? g1 : g2;
                if (g == INF) break;
                if (g == g2) { for (int k = 1; k < D.n; k++) D.v[k - 1] = D.v[k]; D.n--; d_head = D.n > 0 ? D.v[0] : INF; }
                if (g == g1) {
                    R = W.record(gi);
                    gi++;
                } else {
                    R = A.rec[g];
                }
            }
            last_g = g;
            t_acc[2]++;
            SEQ2_SEL();
            const long _wb = t_acc[0]; (void)_wb;
            if (R.type == 0) continue;
            const int id = R.node;
            // ONE inlined copy of the walk serves both kinds of growth (round 6: the pass's loop is tens of KB of code; a second copy of
            // murray_to_root in it is a few KB more for the instruction cache to hold)
            bool walk = false;
            if (R.type == 1) {
                bool bif = false;
                if (R.draw) {
                    if (OCTA_UNLIKELY(py_pos >= py_cap)) { err |= ERR_PY_CAP; break; }
                    const double u = U.at(py_pos);
                    py_pos++;
                    bif = (R.thr > u) && R.ang_gt90;
                }
                if (OCTA_UNLIKELY(bif)) {
                    if (OCTA_UNLIKELY(R.req < 0)) { err |= ERR_MISSING_BIF; continue; }
                    const double *o = bif_results + 6 * (size_t)R.req;
                    seq_add_node(A, f, n_nodes, v3(o[0], o[1], o[2]), C.r, id, 0, P.kappa, L);
                    seq_add_node(A, f, n_nodes, v3(o[3], o[4], o[5]), C.r, id, 1, P.kappa, L);
                    walk = true;
                    n_bif++;
                } else {
                    seq_add_node(A, f, n_nodes, ld3(R.newpos), C.r, id, 0, P.kappa, L);
                }
            } else {
                if (OCTA_UNLIKELY(changed_get(L, g))) {      // (one visit in eleven) a walk of this pass rewrote the child's radius (every rewrite changes it: the walk stops at old == new)
#if defined(__HIP_DEVICE_COMPILE__)
                    SEQT(1, eval_inter<true>(G, g, R));      // the device's workgroups have at least one whole wave (SIM_THREADS_PER_WG >= 64)
#else
                    SEQT(1, eval_inter<false>(G, g, R));
#endif
                    respec++;
                }
                if (!R.grow) { SEQ2_END(2, 4, _wb); continue; }
                if (OCTA_UNLIKELY(py_pos >= py_cap)) { err |= ERR_PY_CAP; break; }
                const double u = U.at(py_pos);
                py_pos++;
                if (R.thr <= u && !R.ang_gt90) { SEQ2_END(2, 4, _wb); continue; }
                seq_add_node(A, f, n_nodes, ld3(R.newpos), C.r, id, 1, P.kappa, L);
                walk = true;
            }
            if (walk) {
                SEQT(0, steps += murray_to_root(A, f, id, g, tag, &D, L, n_def SEQ2_ARG));
                d_head = D.n > 0 ? D.v[0] : INF;
                t_acc[3]++;
                A.nact_of(f)[id] = 0;
            }
            SEQ2_END(R.type == 1 ? 1 : 2, 3, _wb);
            if (OCTA_UNLIKELY(D.overflow && !scan_all)) { scan_all = true; D.n = 0; }
        }
        sc->new_begin[f] = n_before;
        sc->new_end[f] = n_nodes;
        sc->n_nodes[f] = n_nodes;
        sc->py_pos = py_pos;
        if (b.tid == 0) {        // one lane: the read-modify-write statistics must not be repeated by the other 63
            if (err) sc->err |= err;
            sc->murray_steps += steps + n_def;
            sc->murray_deferred += n_def;
#if defined(__HIP_DEVICE_COMPILE__)
            b.coll()[91] = n_def <= MURRAY_FLUSH_LDS ? 0 : n_def;        // > 0: evaluated at once by the whole workgroup, below
#else
            b.coll()[91] = n_def;
#endif
            sc->n_bif += n_bif;
            sc->respec += respec;
            sc->kdprof[7] += t_acc[0]; (void)t_acc[1]; (void)t_acc[2];
#if defined(OCTA_SIM_PROF_SEQ2) && defined(__HIP_DEVICE_COMPILE__)
            q2[3] = t_acc[1];
            for (int k = 0; k < 7; k++) sc->kdprof[k] += q2[k];
#endif
#if defined(OCTA_SIM_PROF_SEQ) && defined(__HIP_DEVICE_COMPILE__)
            sc->kdprof[1] += (long)wall_clock64() - _q1; sc->kdprof[2] += t_acc[0]; sc->kdprof[4] += t_acc[2]; sc->kdprof[5] += t_acc[3]; sc->kdprof[6] += n_def;
#endif
        }
#if defined(__HIP_DEVICE_COMPILE__)
        if (n_def > 0 && n_def <= MURRAY_FLUSH_LDS) {
            // this pass's marked nodes: records now (they need the marked bitmap, which dies with the pass), rounds beside the other forest's pass
            const long tp0 = (long)wall_clock64();
            for (int base = 0; base < n_def; base += MURRAY_FLUSH_WAVE)
                murray_flush_prepare_wave(A, f, L, n_def, A.fl_rec + (size_t)f * MURRAY_FLUSH_LDS, base);
            if (b.tid == 0) { sc->fl_pending[f] = n_def; sc->kdprof[7] += (long)wall_clock64() - tp0; }
        }
#endif
    }
#if defined(__HIP_DEVICE_COMPILE__)
    else {
        // the other forest's deferred rounds, its list dealt out in blocks of 64 slots: beside the venous pass to the third, fourth and second
        // wave (all idle there); beside the arterial pass to the third and fourth only -- the second carries the candidate stream for most of
        // that pass, and slots it took would hold up the other waves' parents -- unless the (venous) list is longer than the two can hold.
        // The waves of one list run side by side without barriers.
        static_assert(SIM_THREADS_PER_WG == 256, "the ordered pass hands its side work to waves 1 - 3");
        const int wv = b.tid >> 6;
        if (wv == 1) side(side_lds);
        const int nw = (f == 1 || pend_other > 2 * MURRAY_FLUSH_WAVE) ? 3 : 2;
        const int w = wv == 2 ? 0 : (wv == 3 ? 1 : 2);
        if (w < nw && w * 64 < pend_other) {
            const int rounds = murray_flush_rounds_wave(A.nrad_of(1 - f), L.log_tab, L.exp_tab, L.fl_val, L.fl_done, A.fl_rec + (size_t)(1 - f) * MURRAY_FLUSH_LDS,
                                                        pend_other, w, nw, true);
            if (b.tid == 128) sc->flush_rounds += rounds;
        }
    }
#else
    (void)side; (void)side_lds;
#endif
#undef SEQT
#undef SEQ2_TOP
#undef SEQ2_SEL
#undef SEQ2_END
#undef SEQ2_ARG
    b.sync();
#if defined(__HIP_DEVICE_COMPILE__)
    if (b.tid == 0 && pend_other > 0) sc->fl_pending[1 - f] = 0;
#endif
    {
        long t0 = OCTA_FLUSH_T0();
        const int rounds = murray_flush(b, A, f, L, b.coll()[91]);
        if (b.tid == 0) {
            sc->kdprof[7] += OCTA_FLUSH_T0() - t0; sc->flush_rounds += rounds;
#if defined(OCTA_SIM_PROF_SEQ) && defined(__HIP_DEVICE_COMPILE__)
            sc->kdprof[3] += OCTA_FLUSH_T0() - t0;
#endif
        }
    }
    b.sync();
}

// The run is over (or a caller wants final radii): evaluate whatever a forest's last ordered pass left for "the other forest's next pass".
// One wave, the constant pow tables, fl_val / fl_done at the start of the table area.
OCTA_HD inline void murray_flush_pending(const Blk &b, const SimArrays &A) {
#if defined(__HIP_DEVICE_COMPILE__)
    b.sync();
    double *fl_val = reinterpret_cast<double *>(b.user_of<8>());
    int *fl_done = reinterpret_cast<int *>(fl_val + MURRAY_FLUSH_LDS);
    for (int f = 0; f < 2; f++) {
        const int n = OCTA_UNI(A.sc->fl_pending[f]);
        if (n <= 0) continue;              // block-uniform
        for (int i = b.tid; i < MURRAY_FLUSH_LDS / 32; i += b.nth) fl_done[i] = 0;
        b.sync();
        const int wv = b.tid >> 6;
        if (wv < 3 && wv * 64 < n) {
            const int rounds = murray_flush_rounds_wave(A.nrad_of(f), octa_gpow::LOG_TAB, octa_gpow::EXP_TAB, fl_val, fl_done, A.fl_rec + (size_t)f * MURRAY_FLUSH_LDS, n,
                                                        wv, 3, true);
            if (b.tid == 0) { A.sc->fl_pending[f] = 0; A.sc->flush_rounds += rounds; }
        }
        b.sync();
    }
#else
    (void)b; (void)A;
#endif
}

// stable removal of flagged points from an ordered list (element_mesh.py:196-211 delete_all), IN PLACE: a point only moves towards
// the front, so the list is rewritten tile by tile -- a tile = the contiguous chunks of a group of threads; its kept points are
// packed into LDS (each thread at its scanned offset), then copied to their final place with coalesced stores. One read of the
// list, one write of the kept points that moved; no staging copy in HBM. `stage` is unused (kept for the callers' signature).
constexpr int COMPACT_TILE = 3072;     // points per LDS tile (72 KiB of the user area)
OCTA_HD inline int compact_points(const Blk &b, double *pts, int n, const unsigned char *removed, double *stage) {
    (void)stage;
    if (b.nth == 1) {                                                 // host build: one thread, a forward copy is in place already
        int w = 0;
        for (int i = 0; i < n; i++)
            if (!removed[i]) { if (w != i) { pts[3 * w] = pts[3 * i]; pts[3 * w + 1] = pts[3 * i + 1]; pts[3 * w + 2] = pts[3 * i + 2]; } w++; }
        return w;
    }
    double *tile = reinterpret_cast<double *>(b.user_of<16>());
    static_assert((size_t)COMPACT_TILE * 24 <= (size_t)SIM_USER_BYTES, "compaction tile");
#if defined(__HIP_DEVICE_COMPILE__)
    // a tile = COMPACT_TILE consecutive points; every wave takes a contiguous quarter of it, its lanes 64 consecutive points per step
    // (coalesced flags and coordinates; round 4: a contiguous chunk per THREAD made every lane of a load touch its own cache line)
    const int lane = b.tid & 63, wv = b.tid >> 6, nw = (b.nth + 63) >> 6;
    int base = 0;                                                     // kept points in front of the tile = its first destination
    for (int t0 = 0; t0 < n; t0 += COMPACT_TILE) {
        const int tn = n - t0 < COMPACT_TILE ? n - t0 : COMPACT_TILE;
        const int seg = ((tn + nw * 64 - 1) / (nw * 64)) * 64;
        const int s0 = t0 + (wv * seg < tn ? wv * seg : tn), s1 = s0 + seg < t0 + tn ? s0 + seg : t0 + tn;
        int c = 0;
        for (int i0 = s0; i0 < s1; i0 += 64) c += (int)__popcll(__ballot(i0 + lane < s1 && !removed[i0 + lane]));
        int ex;
        const int kept = blk_scan(b, lane == 0 ? c : 0, &ex);
        if (base != t0 || kept != tn) {                               // (uniform) points in front of the first removal keep their place
            int at = __builtin_amdgcn_readfirstlane(ex);
            for (int i0 = s0; i0 < s1; i0 += 64) {
                const bool keep = i0 + lane < s1 && !removed[i0 + lane];
                const unsigned long long m = __ballot(keep);
                if (keep) {
                    const int w = at + (int)__popcll(m & ((1ull << lane) - 1ull));
                    const V3 v = ld3(pts + 3 * (size_t)(i0 + lane));
                    tile[3 * w] = v.x; tile[3 * w + 1] = v.y; tile[3 * w + 2] = v.z;
                }
                at += (int)__popcll(m);
            }
            b.sync();
            for (int j = b.tid; j < kept * 3; j += b.nth) pts[(size_t)3 * base + j] = tile[j];
            b.sync();                                                 // the tile is reused; the next tile's sources lie behind this one's destinations
        }
        base += kept;
    }
    return base;
#else
    (void)tile;
    return n;    // (not reached: the host build has one thread)
#endif
}

#if defined(__HIP_DEVICE_COMPILE__)
#define OCTA_SUBPROF(sc, slot, t0) do { if (b.tid == 0) { long _t1 = (long)wall_clock64(); (sc)->prof[slot] += _t1 - (t0); (t0) = _t1; } } while (0)
#define OCTA_SUBPROF_T0() ((long)wall_clock64())
#else
#define OCTA_SUBPROF(sc, slot, t0) do { (void)(t0); } while (0)
#define OCTA_SUBPROF_T0() 0L
#endif

// ------------------------------------------------------------------ phase: satisfied O2 sinks -> CO2
OCTA_HD inline void phase_satisfy_art(const Blk &b, const SimArrays &A, const SimConst &C, const IterParams &P) {
    const double zext = C.sz;      // every sink passed is_valid_position: 0 <= z < size_z
    SampleScalars *sc = A.sc;
    const int nb = sc->new_begin[0], ne = sc->new_end[0];
    const int n_new = ne - nb, n_oxy = sc->n_oxy;
    if (n_new <= 0 || n_oxy <= 0) return;
    const double ek = P.eps_k, ek2 = ek * ek;
    long t0 = OCTA_SUBPROF_T0();
    for (int i = b.tid; i < n_oxy; i += b.nth) A.removed[i] = 0;
    int *ctl = b.coll() + 100;
    if (b.tid == 0) ctl[0] = 0;
    b.sync();
    // 1. (new node, sink) hits from a grid over the new nodes: raw pairs node_local << 14 | sink
    {
        int *new_ids = A.tmp_int;
        for (int j = b.tid; j < n_new; j += b.nth) new_ids[j] = nb + j;
        b.sync();
        Grid G = grid_build(b, A, A.npos[0], new_ids, n_new, ek);
        for (int o = b.tid; o < n_oxy; o += b.nth) {
            const V3 p = ld3(A.oxy + 3 * o);
            grid_visit(G, p.x, p.y, ek, [&](int j, const V3 &q) {
                if (sqdist(p, q) <= ek2) {
                    int slot = atomic_add_int(&ctl[0], 1);
                    if (slot < PCAP) A.pairs[slot] = ((unsigned)(j - nb) << IDX_BITS) | (unsigned)o;
                    A.removed[o] = 1;
                }
            });
        }
        b.sync();
    }
    OCTA_SUBPROF(sc, 12, t0);
    int n_pairs = ctl[0];
    if (n_pairs > PCAP) { if (b.tid == 0) atomic_or_int(&sc->err, ERR_PAIR_CAP); n_pairs = PCAP; }
    if (n_new > (1 << (32 - IDX_BITS))) { if (b.tid == 0) atomic_or_int(&sc->err, ERR_PAIR_CAP); }
    b.sync();
    if (n_pairs == 0) return;  // nothing satisfied: no conversion, no deletion (uniform across the block)
    // 2. cKDTree order of the O2 list, only as deep as the hit sinks need it; pairs get kd ranks
    kd_build(b, A.oxy, n_oxy, A.kd_idx, A.kd_rank, reinterpret_cast<float *>(A.hashes) /* free until step 3 */, 0.0, zext, sc->kdprof, A.removed, true);
#if OCTA_SIM_DUP & 1
    kd_build(b, A.oxy, n_oxy, A.kd_idx, A.kd_rank, reinterpret_cast<float *>(A.hashes), 0.0, zext, nullptr, A.removed, true);
#endif
    for (int i = b.tid; i < n_pairs; i += b.nth) {
        unsigned pr = A.pairs[i];
        A.pairs[i] = (pr & ~IDX_MASK) | (unsigned)A.kd_rank[pr & IDX_MASK];
    }
    b.sync();
    OCTA_SUBPROF(sc, 11, t0);
    // 3. venous proximity + tuple hash for every removed sink. Inverted like the candidate tests (round 3): the grid holds the few
    //    hundred removed sinks, every venous node visits the cells around itself and flags the sinks within eps_k (same expression,
    //    same operand order; an existence test). Round 2 binned all ~13 k venous nodes per iteration for these few hundred queries.
    {
        const int n_ven = sc->n_nodes[1];
        int *rem = A.tmp_int;                 // removed sinks, ascending
        int n_rem = 0;
        {
            const int chunk = (n_oxy + b.nth - 1) / b.nth;
            const int i0 = b.tid * chunk < n_oxy ? b.tid * chunk : n_oxy, i1 = (i0 + chunk < n_oxy) ? i0 + chunk : n_oxy;
            int local = 0;
            for (int o = i0; o < i1; o++) local += A.removed[o] ? 1 : 0;
            int ex;
            n_rem = blk_scan(b, local, &ex);
            int run = ex;
            for (int o = i0; o < i1; o++) if (A.removed[o]) rem[run++] = o;
        }
        b.sync();
        for (int k = b.tid; k < n_rem; k += b.nth) {
            const int o = rem[k];
            A.ven_near[o] = 0;
            A.hashes[o] = py_hash_tuple3(ld3(A.oxy + 3 * o));
        }
        Grid G = grid_build(b, A, A.oxy, rem, n_rem, ek);      // (its leading barrier orders the flag resets before the visits)
        constexpr int VB = 4;
        for (int j0 = b.tid; j0 < n_ven; j0 += VB * b.nth) {
            V3 qv[VB];
#pragma unroll
            for (int u = 0; u < VB; u++) { const int j = j0 + u * b.nth; qv[u] = ld3(A.npos[1] + 3 * (j < n_ven ? j : n_ven - 1)); }
#pragma unroll
            for (int u = 0; u < VB; u++) {
                if (j0 + u * b.nth >= n_ven) break;
                const V3 q = qv[u];
                grid_visit(G, q.x, q.y, ek, [&](int o, const V3 &p) {
                    if (sqrt(sqdist(q, p)) <= ek) A.ven_near[o] = 1;
                });
            }
        }
        b.sync();
    }
    OCTA_SUBPROF(sc, 12, t0);
    // 4. sort the pairs: new nodes in order, hits in cKDTree order
    unsigned *keys = reinterpret_cast<unsigned *>(b.user_of<32>());
    int n_pow2 = 1;
    while (n_pow2 < n_pairs) n_pow2 <<= 1;
    for (int i = b.tid; i < n_pow2; i += b.nth) keys[i] = i < n_pairs ? A.pairs[i] : 0xffffffffu;
    b.sync();
    if (n_pairs > 0) blk_sort_u32(b, keys, n_pow2);
    for (int i = b.tid; i < n_pairs; i += b.nth) A.pairs[i] = keys[i];
    b.sync();
    OCTA_SUBPROF(sc, 13, t0);
    // 5. CPython set insertion order -> CO2 append order
    bool set_in_lds = false;
#ifdef OCTA_SIM_DEBUG_SAT
    int dbg_n_ins = -1, dbg_mask = -1, dbg_base = -1;
    const int dbg_n_co2_in = sc->n_co2;
#endif
#if defined(__HIP_DEVICE_COMPILE__)
    // Usual case (<= LSET_PAIRS hits): the insert stream (sink, hash) is compacted into the table area in parallel and the set is
    // replayed there by the WHOLE workgroup (round 4; one wave inserting key after key until then: 21 ms per sample, the other three
    // waves idle). What makes a parallel replay possible: a CPython set that only grows never moves an entry inside one table, so
    // the slot of the k-th distinct key is the first slot of ITS probe sequence that no earlier key of the same table holds -- a
    // fixed point that insertion with priorities computes in any order (a key that finds a slot held by a later key takes it and
    // carries the displaced key on along that key's own sequence). A resize re-inserts the entries in the old table's slot order,
    // which is the same operation with the old slot as the priority. The resizes happen at fixed counts of distinct keys
    // (5, 19, 77, 307, ... -> 32, 128, 512, 2048 slots), so the replay is one priority insertion per table generation:
    //   fp[OCAP]: first arrival of a sink (duplicates of a key never touch the table), later own[slot] = priority of the slot's entry,
    //   at the end the slot's key (-1: empty) -- the table the read-out below walks;
    //   in_key / in_hash: arrivals; dk / dh: distinct keys in arrival order; ord0 / ord1: entry of a priority (this / next generation).
#if defined(OCTA_SIM_PROF_SET)
    // diagnostic build: the kd slots of the phase profile hold the steps of the set replay: insert stream, distinct keys, generations,
    // key table, read-out
    long _pt = (long)wall_clock64();
#define PSP(slot) do { if (b.tid == 0) { long _t = (long)wall_clock64(); sc->kdprof[slot] += _t - _pt; _pt = _t; } } while (0)
#else
#define PSP(slot) do { } while (0)
#endif
    constexpr int EMPTY = 0x7fffffff;
    // own: [max(OCAP, s_max)] ints; in_key / in_hash: [n_pairs]; dk / dh / ord0 / ord1: [number of distinct sinks]. Returns the table's mask;
    // own[slot] then holds the slot's key (-1: empty).
    auto replay = [&](int *own, int *in_key, unsigned long long *in_hash, int *dk, unsigned long long *dh, int *ord0, int *ord1,
                      const int s_max) __attribute__((always_inline)) -> int {
        int n_ins = 0;
        for (int p0 = 0; p0 < n_pairs; p0 += b.nth) {          // ordered compaction of the insert stream, b.nth pairs per round
            const int i = p0 + b.tid;
            int o = -1, take = 0;
            unsigned long long hsh = 0;
            if (i < n_pairs) {
                o = (int)A.kd_idx[keys[i] & IDX_MASK];         // the sorted pairs are still in the LDS
                take = A.ven_near[o] ? 0 : 1;
                hsh = A.hashes[o];                             // fetched beside the flag, not behind it
            }
            int ex;
            const int tot = blk_scan(b, take, &ex);
            if (take) { in_key[n_ins + ex] = o; in_hash[n_ins + ex] = hsh; }
            n_ins += tot;
        }
        b.sync();
        PSP(0);
        // distinct keys in arrival order
        for (int i = b.tid; i < n_ins; i += b.nth) own[in_key[i]] = EMPTY;
        b.sync();
        for (int i = b.tid; i < n_ins; i += b.nth) atomic_min_int(&own[in_key[i]], i);
        b.sync();
        int D = 0;
        for (int p0 = 0; p0 < n_ins; p0 += b.nth) {
            const int i = p0 + b.tid;
            const bool first = i < n_ins && own[in_key[i]] == i;
            int ex;
            const int tot = blk_scan(b, first ? 1 : 0, &ex);
            if (first) { dk[D + ex] = in_key[i]; dh[D + ex] = in_hash[i]; }
            D += tot;
        }
        b.sync();
        PSP(1);
        // one priority insertion per table generation
        int S = 8, n_prev = 0;
        int *ord = ord0, *ord_next = ord1;
        while (true) {
            const int thr = (3 * (S - 1) + 4) / 5;            // the insertion that makes fill * 5 >= mask * 3 resizes the table
            const int upto = D < thr ? D : thr;               // distinct keys of this generation's table
            const unsigned long long mask = (unsigned long long)(S - 1);
            for (int p = n_prev + b.tid; p < upto; p += b.nth) ord[p] = p;      // behind the re-inserted entries: the arrivals, in order
            for (int e = b.tid; e < S; e += b.nth) own[e] = EMPTY;
            b.sync();
            for (int p = b.tid; p < upto; p += b.nth) {
                int cur = p;
                // CPython's probe sequence: slot i, then up to nine slots behind it (if they fit), then i = 5 i + 1 + (perturb >>= 5)
                unsigned long long hsh = dh[ord[cur]], i = hsh & mask, perturb = hsh;
                int lin = 0, nlin = (i + 9 <= mask) ? 9 : 0;
                auto next = [&] {
                    if (lin < nlin) { lin++; return; }
                    perturb >>= 5;
                    i = (i * 5 + 1 + perturb) & mask;
                    nlin = (i + 9 <= mask) ? 9 : 0; lin = 0;
                };
                while (true) {
                    const int slot = (int)i + lin;
                    const int old = atomic_min_ret_int(&own[slot], cur);
                    if (old == EMPTY) break;                  // an empty slot: placed
                    if (old > cur) {                          // a later entry held it: it moves on along ITS sequence, from behind this slot
                        cur = old;
                        hsh = dh[ord[cur]]; i = hsh & mask; perturb = hsh; lin = 0; nlin = (i + 9 <= mask) ? 9 : 0;
                        while ((int)i + lin != slot) next();  // (its first visit of the slot is the one it was placed by)
                    }
                    next();
                }
            }
            b.sync();
            if (D < thr) break;                               // no resize behind this generation: its table is the set
            const int minused = upto > 50000 ? upto * 2 : upto * 4;
            int newS = 8;
            while (newS <= minused) newS <<= 1;
            if (newS > s_max) { if (b.tid == 0) atomic_or_int(&sc->err, ERR_SET_CAP); break; }
            // the entries in slot order = the priorities of the re-insertion
            {
                const int lane = b.tid & 63, wv = b.tid >> 6, nw = (b.nth + 63) >> 6;
                const int seg = ((S + nw * 64 - 1) / (nw * 64)) * 64;
                const int s0 = wv * seg < S ? wv * seg : S, s1 = s0 + seg < S ? s0 + seg : S;
                int c = 0;
                for (int e0 = s0; e0 < s1; e0 += 64) c += (int)__popcll(__ballot(e0 + lane < s1 && own[e0 + lane] != EMPTY));
                int ex;
                blk_scan(b, lane == 0 ? c : 0, &ex);
                int at = __builtin_amdgcn_readfirstlane(ex);
                for (int e0 = s0; e0 < s1; e0 += 64) {
                    const int pr = e0 + lane < s1 ? own[e0 + lane] : EMPTY;
                    const unsigned long long m = __ballot(pr != EMPTY);
                    if (pr != EMPTY) ord_next[at + (int)__popcll(m & ((1ull << lane) - 1ull))] = ord[pr];
                    at += (int)__popcll(m);
                }
            }
            b.sync();
            { int *t = ord; ord = ord_next; ord_next = t; }
            n_prev = upto;
            S = newS;
        }
        PSP(2);
        for (int e = b.tid; e < S; e += b.nth) { const int pr = own[e]; own[e] = pr == EMPTY ? -1 : dk[ord[pr]]; }
        b.sync();
        PSP(3);
#ifdef OCTA_SIM_DEBUG_SAT
        dbg_n_ins = n_ins;
#endif
        return S - 1;
    };
    // the tables of the replay: in the table area for the usual <= LSET_PAIRS hits; in the sample's HBM scratch for the few iterations
    // with more (the first iteration of a mode runs with the mode's raw radii: thousands of hits -- replayed by ONE thread until
    // round 4, 16 of the 21 ms per sample this step took)
    constexpr bool HBM_REPLAY = (size_t)PCAP + OCAP <= (size_t)OCAP + 2 * (size_t)NCANDCAP && (size_t)OCAP * 2 <= (size_t)OCAP * 3
                                && PCAP <= SETCAP && OCAP <= SETCAP;
    const int *t_key = nullptr;
    int mask = -1;
    if (n_pairs <= LSET_PAIRS) {
        constexpr int S_MAX = LSET_CAP / 2;
        int *own = reinterpret_cast<int *>(b.user_of<64>());                              // [max(OCAP, S_MAX)]
        constexpr int OWN_N = OCAP > S_MAX ? OCAP : S_MAX;
        int *in_key = own + OWN_N;                                                        // [LSET_PAIRS]
        int *dk = in_key + LSET_PAIRS, *ord0 = dk + LSET_PAIRS, *ord1 = ord0 + LSET_PAIRS;
        unsigned long long *in_hash = reinterpret_cast<unsigned long long *>(ord1 + LSET_PAIRS + ((OWN_N + 4 * LSET_PAIRS) & 1));
        unsigned long long *dh = in_hash + LSET_PAIRS;
        static_assert((size_t)(OWN_N + 4 * LSET_PAIRS + 1) * 4 + (size_t)2 * LSET_PAIRS * 8 <= (size_t)SIM_USER_BYTES, "set replay layout");
        static_assert((size_t)LSET_PAIRS * 4 <= (size_t)OWN_N * 4, "the sorted pairs (start of the table area) end before the insert stream");
        mask = replay(own, in_key, in_hash, dk, dh, ord0, ord1, S_MAX);
        t_key = own;
        set_in_lds = true;
    } else if (HBM_REPLAY && n_pairs <= PCAP) {
        // own: set_key [SETCAP]; in_key: tmp_int [PCAP]; ord0: tmp_int behind it [OCAP]; in_hash: set_hash [PCAP]; dh, dk, ord1: tmp_dbl
        int *own = A.set_key;
        int *in_key = A.tmp_int, *ord0 = A.tmp_int + PCAP;
        unsigned long long *in_hash = A.set_hash;
        unsigned long long *dh = reinterpret_cast<unsigned long long *>(A.tmp_dbl);
        int *dk = reinterpret_cast<int *>(A.tmp_dbl + OCAP), *ord1 = dk + OCAP;
        mask = replay(own, in_key, in_hash, dk, dh, ord0, ord1, SETCAP / 2);
        t_key = own;
        set_in_lds = true;
    }
    if (set_in_lds) {
        const int n_co2_0 = sc->n_co2;
        // read-out in slot order: a contiguous run of slots per thread, ONE block scan, then the converted sinks' coordinates fetched
        // eight at a time (one scan and one dependent fetch per 256 slots until round 4)
        int base = 0;
        {
            const int per = (mask + 1 + b.nth - 1) / b.nth;
            const int e0 = b.tid * per < mask + 1 ? b.tid * per : mask + 1, e1 = e0 + per < mask + 1 ? e0 + per : mask + 1;
            int cnt = 0;
            for (int e = e0; e < e1; e++) cnt += t_key[e] >= 0 ? 1 : 0;
            int ex2;
            base = blk_scan(b, cnt, &ex2);
            int dst = n_co2_0 + ex2;
            constexpr int RB = 8;
            for (int eb = e0; eb < e1; eb += RB) {
                int k[RB];
                V3 v[RB];
#pragma unroll
                for (int u = 0; u < RB; u++) { k[u] = eb + u < e1 ? t_key[eb + u] : -1; }
#pragma unroll
                for (int u = 0; u < RB; u++) v[u] = ld3(A.oxy + 3 * (k[u] >= 0 ? k[u] : 0));
#pragma unroll
                for (int u = 0; u < RB; u++)
                    if (k[u] >= 0) { if (dst < CCAP) st3(A.co2 + 3 * dst, v[u]); dst++; }
            }
        }
        b.sync();
        PSP(4);
        if (b.tid == 0) {
            int n_co2 = n_co2_0 + base;
            if (n_co2 > CCAP) { sc->err |= ERR_CO2_CAP; n_co2 = CCAP; }
            sc->n_co2 = n_co2;
        }
#ifdef OCTA_SIM_DEBUG_SAT
        dbg_mask = mask; dbg_base = base;
#endif
    }
#undef PSP
#endif
    if (!set_in_lds && b.tid == 0) {
        PySetView S;
        S.hash = A.set_hash; S.key = A.set_key; S.err = &sc->err; S.cap = SETCAP;
        pyset_init(S);
        for (int i = 0; i < n_pairs; i++) {
            int o = (int)A.kd_idx[A.pairs[i] & IDX_MASK];
            if (!A.ven_near[o]) pyset_add(S, o, A.hashes[o]);
        }
        int n_co2 = sc->n_co2;
        for (int e = 0; e <= S.mask; e++)
            if (S.key[e] >= 0) {
                if (n_co2 >= CCAP) { sc->err |= ERR_CO2_CAP; break; }
                int o = S.key[e];
                A.co2[3 * n_co2] = A.oxy[3 * o]; A.co2[3 * n_co2 + 1] = A.oxy[3 * o + 1]; A.co2[3 * n_co2 + 2] = A.oxy[3 * o + 2];
                n_co2++;
            }
        sc->n_co2 = n_co2;
    }
    b.sync();
    OCTA_SUBPROF(sc, 14, t0);
#if defined(OCTA_SIM_DEBUG_SAT) && !defined(OCTA_SIM_ITER_PROF)
    {   // digest of this call (compared between repeated runs on the host) + in-kernel recounts of every stage of step 5
        int ex, loc;
        loc = 0; for (int o = b.tid; o < n_oxy; o += b.nth) loc += A.removed[o] ? 1 : 0;
        const int c_rem = blk_scan(b, loc, &ex);
        loc = 0; for (int o = b.tid; o < n_oxy; o += b.nth) loc += (A.removed[o] && A.ven_near[o]) ? 1 : 0;
        const int c_ven = blk_scan(b, loc, &ex);
        // distinct sinks named by the sorted pairs (all, and those not near a venous node): byte marks in tmp_dbl
        unsigned char *mark = reinterpret_cast<unsigned char *>(A.tmp_dbl);
        for (int o = b.tid; o < 2 * n_oxy; o += b.nth) mark[o] = 0;
        b.sync();
        for (int i = b.tid; i < n_pairs; i += b.nth) {
            const int o = (int)A.kd_idx[A.pairs[i] & IDX_MASK];
            mark[o] = 1;
            if (!A.ven_near[o]) mark[n_oxy + o] = 1;
        }
        b.sync();
        loc = 0; for (int o = b.tid; o < n_oxy; o += b.nth) loc += mark[o];
        const int c_pair_all = blk_scan(b, loc, &ex);
        loc = 0; for (int o = b.tid; o < n_oxy; o += b.nth) loc += mark[n_oxy + o];
        const int c_pair_take = blk_scan(b, loc, &ex);
        loc = 0; for (int o = b.tid; o < n_oxy; o += b.nth) loc += (mark[o] != (A.removed[o] ? 1 : 0)) ? 1 : 0;
        const int c_pair_vs_removed = blk_scan(b, loc, &ex);
        int c_slots = -1, c_in_distinct = -1, c_slot_bad = -1;
        if (set_in_lds) {     // the LDS table and insert stream are still intact (only the collectives area was used since)
            const int *t_key = reinterpret_cast<const int *>(b.user());
            const int *in_key = t_key + (OCAP > LSET_CAP / 2 ? OCAP : LSET_CAP / 2);
            loc = 0; for (int e = b.tid; e <= dbg_mask; e += b.nth) loc += t_key[e] >= 0 ? 1 : 0;
            c_slots = blk_scan(b, loc, &ex);
            loc = 0; for (int e = b.tid; e <= dbg_mask; e += b.nth) { const int k = t_key[e]; if (k >= 0 && (k >= n_oxy || !A.removed[k] || A.ven_near[k])) loc++; }
            c_slot_bad = blk_scan(b, loc, &ex);
            for (int o = b.tid; o < n_oxy; o += b.nth) mark[o] = 0;
            b.sync();
            for (int i = b.tid; i < dbg_n_ins; i += b.nth) { const int k = in_key[i]; if (k >= 0 && k < n_oxy) mark[k] = 1; }
            b.sync();
            loc = 0; for (int o = b.tid; o < n_oxy; o += b.nth) loc += mark[o];
            c_in_distinct = blk_scan(b, loc, &ex);
        }
        b.sync();
        if (b.tid == 0 && A.dbg) {
            int *row = A.dbg + 16 * A.dbg_it;
            row[0] = n_new; row[1] = n_oxy; row[2] = n_pairs; row[3] = c_rem; row[4] = c_ven; row[5] = c_pair_all; row[6] = c_pair_take;
            row[7] = c_pair_vs_removed; row[8] = dbg_n_ins; row[9] = dbg_mask; row[10] = dbg_base; row[11] = c_slots; row[12] = c_in_distinct;
            row[13] = c_slot_bad; row[14] = dbg_n_co2_in; row[15] = sc->n_co2;
        }
        b.sync();
    }
#endif
    // 6. delete the satisfied sinks (order-preserving)
    int keep = compact_points(b, A.oxy, n_oxy, A.removed, A.tmp_dbl);
    if (b.tid == 0) sc->n_oxy = keep;
    b.sync();
    OCTA_SUBPROF(sc, 15, t0);
}

// ------------------------------------------------------------------ phase: CO2 near new venous nodes removed
OCTA_HD inline void phase_satisfy_ven(const Blk &b, const SimArrays &A, const IterParams &P) {
    SampleScalars *sc = A.sc;
    const int nb = sc->new_begin[1], ne = sc->new_end[1];
    const int n_new = ne - nb, n_co2 = sc->n_co2;
    if (n_new <= 0 || n_co2 <= 0) return;
    const double ek2 = P.eps_k * P.eps_k;
    const double ek = P.eps_k;
    {
        int *new_ids = A.tmp_int;
        for (int j = b.tid; j < n_new; j += b.nth) new_ids[j] = nb + j;
        b.sync();
        Grid G = grid_build(b, A, A.npos[1], new_ids, n_new, ek);
        for (int o = b.tid; o < n_co2; o += b.nth) {
            const V3 p = ld3(A.co2 + 3 * o);
            bool hit = false;
            grid_visit(G, p.x, p.y, ek, [&](int, const V3 &q) {
                if (!hit && sqdist(p, q) <= ek2) hit = true;
            });
            A.removed[o] = hit ? 1 : 0;
        }
        b.sync();
    }
    int keep = compact_points(b, A.co2, n_co2, A.removed, A.tmp_dbl);
    if (b.tid == 0) sc->n_co2 = keep;
    b.sync();
}

}  // namespace OCTA_SIMK
