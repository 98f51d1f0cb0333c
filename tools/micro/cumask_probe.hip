// Which CUs does a stream created with hipExtStreamCreateWithCUMask really use, and how do mask bits map to (XCC, SE, CU)?
// For each test mask: 2048 single-wave workgroups record HW_ID / XCC_ID and spin 0.2 ms; the distinct CUs are counted per XCC.
// Build: hipcc --offload-arch=gfx950 -O2 tools/micro/cumask_probe.hip -o tools/micro/cumask_probe ; run on the GPU box.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#include <set>
#include <map>
#include <tuple>
__global__ void __launch_bounds__(64) probe(unsigned *out) {
    const unsigned hw = __builtin_amdgcn_s_getreg((31 << 11) | (0 << 6) | 4);
    const unsigned xcc = __builtin_amdgcn_s_getreg((3 << 11) | (0 << 6) | 20);
    if (threadIdx.x == 0) { out[blockIdx.x * 2] = hw; out[blockIdx.x * 2 + 1] = xcc; }
    const long t0 = wall_clock64();
    while (wall_clock64() - t0 < 20000) __builtin_amdgcn_s_sleep(32);
}
static void run(const char *name, const std::vector<uint32_t> &mask, unsigned *d, int nb, bool list) {
    hipStream_t st;
    hipError_t e = hipExtStreamCreateWithCUMask(&st, (uint32_t)mask.size(), mask.data());
    if (e != hipSuccess) { printf("%s: create failed: %s\n", name, hipGetErrorString(e)); return; }
    hipMemsetAsync(d, 0xff, nb * 8, st);
    hipLaunchKernelGGL(probe, dim3(nb), dim3(64), 0, st, d);
    hipStreamSynchronize(st);
    std::vector<unsigned> h(nb * 2); hipMemcpy(h.data(), d, nb * 8, hipMemcpyDeviceToHost);
    std::map<unsigned, std::set<std::tuple<unsigned, unsigned, unsigned>>> per;
    for (int b = 0; b < nb; b++) {
        const unsigned hw = h[b * 2], xcc = h[b * 2 + 1] & 15;
        per[xcc].insert({(hw >> 13) & 7, (hw >> 12) & 1, (hw >> 8) & 15});
    }
    size_t tot = 0; for (auto &kv : per) tot += kv.second.size();
    printf("%-28s CUs used %3zu |", name, tot);
    for (auto &kv : per) printf(" xcc%u:%zu", kv.first, kv.second.size());
    printf("\n");
    if (list) for (auto &kv : per) { printf("    xcc%u:", kv.first); for (auto &t : kv.second) printf(" se%u.sh%u.cu%u", std::get<0>(t), std::get<1>(t), std::get<2>(t)); printf("\n"); }
    hipStreamDestroy(st);
}
int main() {
    const int nb = 8192;
    unsigned *d; hipMalloc(&d, nb * 8);
    hipDeviceProp_t p; hipGetDeviceProperties(&p, 0);
    printf("device CUs %d\n", p.multiProcessorCount);
    std::vector<uint32_t> all(8, 0xffffffffu);
    run("all 256", all, d, nb, false);
    for (int k : {0, 1, 2, 7, 8, 9, 31, 32, 33, 64, 255}) { std::vector<uint32_t> m(8, 0); m[k / 32] |= 1u << (k % 32); char nm[64]; snprintf(nm, 64, "bit %d", k); run(nm, m, d, nb, true); }
    { std::vector<uint32_t> m(8, 0); m[0] = 0xffffffffu; run("bits 0-31", m, d, nb, true); }
    { std::vector<uint32_t> m(8, 0); m[7] = 0xffffffffu; run("bits 224-255", m, d, nb, true); }
    { std::vector<uint32_t> m(8, 0xffffffffu); m[7] = 0; run("bits 0-223", m, d, nb, false); }
    { std::vector<uint32_t> m(8, 0); for (int i = 0; i < 256; i += 8) m[i / 32] |= 1u << (i % 32); run("every 8th bit", m, d, nb, true); }
    return 0;
}
