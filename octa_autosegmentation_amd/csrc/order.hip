// order.hip -- device-side ordering between a launch of the persistent simulator kernel and work of ANOTHER stream (round 6).
//
// Nothing co-resides with the simulator's workgroups (two of them fill a CU's registers and LDS), so which of two things enqueued at the
// same moment gets the CUs decides a whole launch: a rasterisation dispatched while the next launch's 512 workgroups are being placed
// takes the half-filled CUs and keeps them -- the launch then lasts two samples instead of one ("the headline's slow mode").
// The harmless order is: the launch resident FIRST, the rasterisation behind it (its workgroups then get what finished samples leave).
// Round 5 approximated that from the host: poll the process's launch counter, sleep 0.5 ms "for the dispatcher". Here the rasteriser's
// stream itself waits: a one-wave gate kernel in front of the render kernels spins on the launch's sign-in block (csrc/sim_api.cpp
// octa_sim_launch_flag; written by the persistent kernel's workgroups as they start, csrc/sim.hip) until the launch with the given
// ticket has all but `slack` of its workgroups on the GPU, waits `settle_us` more, and leaves. No host thread polls or sleeps.
// The gate wave itself holds registers of one SIMD, i.e. at most one simulator workgroup cannot be placed while it spins: hence `slack`.
// A launch that never comes (its thread failed) is bounded by `timeout_us`.
#include "common.h"

extern "C" int *octa_sim_launch_flag(int device);

namespace {

__global__ void __launch_bounds__(64) order_gate_kernel(const int *flag, int ticket, int slack, long timeout_ticks, long settle_ticks, int *out3) {
    if (threadIdx.x != 0) return;
    const long t0 = (long)wall_clock64();
    int state = 0;                     // 1: the launch was resident, 2: timed out, 3: resident but for a few workgroups this wave itself keeps out
    int last_n = -1;
    long t_last = t0;
    while (true) {
        const int t = __hip_atomic_load(flag, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_AGENT);
        if (t > ticket) { state = 1; break; }                                  // a later launch has started: this one is long past
        const long now = (long)wall_clock64();
        if (t == ticket) {
            const int g = __hip_atomic_load(flag + 3 + (ticket & 1), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            const int n = __hip_atomic_load(flag + 1 + (ticket & 1), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            if (g > 0 && n >= g - slack) { state = 1; break; }
            // Measured: depending on which hardware queues the streams got, a launch sometimes stops at 508 of 512 workgroups while this
            // wave spins, and the last four start the moment it leaves (their samples then begin 44 ms late when the gate sits out its
            // 50 ms). A sign-in count that is within eight of the grid and has not moved for 0.2 ms is "resident".
            if (n != last_n) { last_n = n; t_last = now; }
            else if (g > 0 && n >= g - 8 && now - t_last > 20000) { state = 3; break; }
        }
        if (now - t0 > timeout_ticks) { state = 2; break; }
        __builtin_amdgcn_s_sleep(32);
    }
    const long t1 = (long)wall_clock64();
    if (state == 1) while ((long)wall_clock64() - t1 < settle_ticks) __builtin_amdgcn_s_sleep(32);
    if (out3) { out3[0] = state; out3[1] = (int)(t1 - t0); out3[2] = __hip_atomic_load(flag + 1 + (ticket & 1), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
}

}  // namespace

// Everything enqueued on `stream` after this call runs when the persistent-kernel launch with `ticket` (octa_sim_launch_count() after that
// launch was made; ticket + 1 = "the next launch") is resident, or after timeout_us. d_out3 (optional, device int[3]): 1 = resident /
// 2 = timed out, 100 MHz ticks waited, workgroups signed in.
extern "C" int octa_order_wait_launch(octa_ctx *ctx, long long ticket, int timeout_us, int settle_us, int *d_out3, void *stream_) {
    if (!ctx || ticket <= 0 || timeout_us < 0 || settle_us < 0) { octa::set_error("octa_order_wait_launch: bad arguments"); return -2; }
    OCTA_HIP_CHECK(hipSetDevice(ctx->device));
    const int *flag = octa_sim_launch_flag(ctx->device);
    if (!flag) { octa::set_error("octa_order_wait_launch: no sign-in block for device %d", ctx->device); return -1; }
    hipLaunchKernelGGL(order_gate_kernel, dim3(1), dim3(64), 0, (hipStream_t)stream_, flag, (int)(ticket & 0x3fffffff), 2, (long)timeout_us * 100L,
                       (long)settle_us * 100L, d_out3);
    OCTA_HIP_CHECK(hipGetLastError());
    return 0;
}
