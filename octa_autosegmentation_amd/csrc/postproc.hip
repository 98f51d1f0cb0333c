// postproc.hip -- connected-component size filter for the inference post-processing.
//
// Replaces `RemoveSmallObjects(min_size=160)` of the configs' post_processing lists (configs/config_ves_seg-S.yml:103-113;
// MONAI -> skimage.morphology.remove_small_objects: label the foreground with scipy.ndimage.label, connectivity 1 = the
// 4-neighbourhood in 2-D, drop every component with fewer than min_size pixels) in test.py / validate.py, for a batch of
// masks resident in HBM.
// Union-find over the pixel grid (label equivalence): every foreground pixel starts as its own root, is united with its
// right / lower neighbours (and the two lower diagonals for connectivity 2) by linking the larger root under the smaller
// one with atomicMin, then every pixel looks up its root, the roots count their members, and a pixel survives if its
// component is large enough. Integer work, exact; HBM-bound (a few passes over 4 bytes per pixel).

#include "common.h"

namespace {

__device__ __forceinline__ int cc_find(int *L, int x) {
    // path halving; L[x] <= x always, roots have L[r] == r
    while (true) {
        const int p = L[x];
        if (p == x) return x;
        const int g = L[p];
        if (g != p) L[x] = g;   // benign race: only ever moves x closer to its root
        x = p;
    }
}

__device__ __forceinline__ void cc_union(int *L, int a, int b) {
    while (true) {
        a = cc_find(L, a);
        b = cc_find(L, b);
        if (a == b) return;
        if (a > b) { const int t = a; a = b; b = t; }
        const int old = atomicMin(&L[b], a);   // link root b under a (a < b)
        if (old == b) return;                  // b was still a root: linked
        b = old;                               // someone linked b elsewhere meanwhile: unite with that instead
    }
}

// read-only walk to the root (after the merge kernel the forest is final)
__device__ __forceinline__ int cc_root(const int *L, int x) {
    while (true) {
        const int p = L[x];
        if (p == x) return x;
        x = p;
    }
}

__global__ void __launch_bounds__(256)
cc_init_kernel(const unsigned char *__restrict__ in, int *__restrict__ L, int *__restrict__ cnt, long n_total) {
    const long i = (long)blockIdx.x * 256 + threadIdx.x;
    if (i >= n_total) return;
    L[i] = in[i] ? (int)i : -1;
    cnt[i] = 0;
}

__global__ void __launch_bounds__(256)
cc_merge_kernel(int *__restrict__ L, int H, int W, int B, int conn8) {
    const long i = (long)blockIdx.x * 256 + threadIdx.x;
    const long n = (long)H * W;
    if (i >= n * B) return;
    if (L[i] < 0) return;
    const int p = (int)(i % n), y = p / W, x = p % W;
    if (x + 1 < W && L[i + 1] >= 0) cc_union(L, (int)i, (int)i + 1);
    if (y + 1 < H) {
        if (L[i + W] >= 0) cc_union(L, (int)i, (int)i + W);
        if (conn8) {
            if (x + 1 < W && L[i + W + 1] >= 0) cc_union(L, (int)i, (int)i + W + 1);
            if (x > 0 && L[i + W - 1] >= 0) cc_union(L, (int)i, (int)i + W - 1);
        }
    }
}

__global__ void __launch_bounds__(256)
cc_count_kernel(const int *__restrict__ L, int *__restrict__ root, int *__restrict__ cnt, long n_total) {
    const long i0 = (long)blockIdx.x * 256 + threadIdx.x;
    const long i = i0 < n_total ? i0 : n_total - 1;      // every lane stays for the ballots; the tail lanes hold no pixel
    int r = -1;
    if (i0 < n_total && L[i] >= 0) r = cc_root(L, (int)i);
    if (i0 < n_total) root[i] = r;
    // members are counted per wave and root: the lanes that share the first pending lane's root add their number with ONE atomic
    // (a vessel mask is a few large components: one atomic per pixel serialised ~0.6 M adds on a handful of addresses, 1 ms at 1216^2)
    unsigned long long pending = __ballot(r >= 0);
    while (pending) {
        const int leader = (int)__ffsll((long long)pending) - 1;
        const int r0 = __shfl(r, leader, 64);
        const unsigned long long same = __ballot(r == r0) & pending;
        if ((int)(threadIdx.x & 63) == leader) atomicAdd(&cnt[r0], (int)__popcll(same));
        pending &= ~same;
    }
}

__global__ void __launch_bounds__(256)
cc_filter_kernel(const int *__restrict__ root, const int *__restrict__ cnt, unsigned char *__restrict__ out, long n_total, int min_size,
                 unsigned char on_value) {
    const long i = (long)blockIdx.x * 256 + threadIdx.x;
    if (i >= n_total) return;
    const int r = root[i];
    out[i] = (r >= 0 && cnt[r] >= min_size) ? on_value : (unsigned char)0;
}

}  // namespace

extern "C" int octa_remove_small_objects(octa_ctx *ctx, const uint8_t *d_in, int B, int H, int W, int min_size, int connectivity,
                                         uint8_t on_value, uint8_t *d_out, void *stream_) {
    if (!ctx || !d_in || !d_out || B <= 0 || H <= 0 || W <= 0) { octa::set_error("octa_remove_small_objects: bad arguments"); return -2; }
    if (connectivity != 1 && connectivity != 2) { octa::set_error("octa_remove_small_objects: connectivity must be 1 (4-neighbourhood) or 2 (8-neighbourhood)"); return -2; }
    const long n_total = (long)B * H * W;
    if (n_total > 0x7fffffffL) { octa::set_error("octa_remove_small_objects: more than 2^31 pixels in one batch"); return -2; }
    hipStream_t stream = (hipStream_t)stream_;
    OCTA_HIP_CHECK(hipSetDevice(ctx->device));
    if (ctx->r_tile_fill.reserve(sizeof(int) * (size_t)n_total)) return -1;
    if (ctx->r_tile_list.reserve(sizeof(int) * 2 * (size_t)n_total)) return -1;
    int *L = ctx->r_tile_fill.as<int>(), *cnt = ctx->r_tile_list.as<int>(), *root = cnt + n_total;
    const unsigned blocks = (unsigned)((n_total + 255) / 256);
    hipLaunchKernelGGL(cc_init_kernel, dim3(blocks), dim3(256), 0, stream, d_in, L, cnt, n_total);
    // labels are global pixel indices, so images of a batch can never be united: neighbours are tested inside one image
    hipLaunchKernelGGL(cc_merge_kernel, dim3(blocks), dim3(256), 0, stream, L, H, W, B, connectivity == 2 ? 1 : 0);
    hipLaunchKernelGGL(cc_count_kernel, dim3(blocks), dim3(256), 0, stream, L, root, cnt, n_total);
    hipLaunchKernelGGL(cc_filter_kernel, dim3(blocks), dim3(256), 0, stream, root, cnt, d_out, n_total, min_size, on_value);
    OCTA_HIP_CHECK(hipGetLastError());
    return 0;
}
