// common.cpp -- context lifecycle and error reporting of liboctahip.so.
#include "common.h"

namespace octa {
static thread_local char g_err[1024] = "";
void set_error(const char *fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
}
}  // namespace octa

extern "C" int octa_abi_version(void) { return 1; }

extern "C" const char *octa_last_error(void) { return octa::g_err; }

extern "C" int octa_ctx_create(int device, octa_ctx **out) {
    if (!out) { octa::set_error("octa_ctx_create: null out"); return -2; }
    *out = nullptr;
    int count = 0;
    OCTA_HIP_CHECK(hipGetDeviceCount(&count));
    if (device < 0 || device >= count) { octa::set_error("octa_ctx_create: device %d out of range (have %d)", device, count); return -2; }
    OCTA_HIP_CHECK(hipSetDevice(device));
    hipDeviceProp_t prop;
    OCTA_HIP_CHECK(hipGetDeviceProperties(&prop, device));
    octa_ctx *c = new (std::nothrow) octa_ctx();
    if (!c) { octa::set_error("octa_ctx_create: out of host memory"); return -1; }
    c->device = device;
    c->num_cus = prop.multiProcessorCount;
    *out = c;
    return 0;
}

extern "C" void octa_ctx_destroy(octa_ctx *ctx) {
    if (!ctx) return;
    hipError_t e = hipSetDevice(ctx->device);
    (void)e;
    ctx->r_edge_off.release();
    ctx->r_ucount.release();
    ctx->r_seg_total.release();
    ctx->r_sides.release();
    ctx->r_edge_meta.release();
    ctx->r_tile_count.release();
    ctx->r_tile_fill.release();
    ctx->r_tile_total.release();
    ctx->r_tile_list.release();
    ctx->r_counters.release();
    ctx->zero_page.release();
    ctx->wgrad_ws.release();
    ctx->zero_ring.release();
    delete ctx;
}

extern "C" size_t octa_ctx_scratch_bytes(const octa_ctx *ctx) { return ctx ? ctx->scratch_bytes() : 0; }
