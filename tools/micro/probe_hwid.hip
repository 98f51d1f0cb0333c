// Where do the waves of co-resident workgroups land? 512 workgroups of 256 threads with 80 KiB of LDS each (the simulator's launch
// shape); lane 0 of every wave records HW_REG_HW_ID and XCC_ID, then the workgroup spins ~2 ms so that all 512 are resident together.
// Build: hipcc --offload-arch=gfx950 -O2 tools/micro/probe_hwid.hip -o tools/micro/probe_hwid ; run on the GPU box.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#include <map>
#include <tuple>
__global__ void __launch_bounds__(256, 2) probe(unsigned *out) {
    extern __shared__ unsigned char smem[];
    smem[threadIdx.x] = 1;
    const unsigned hw = __builtin_amdgcn_s_getreg((31 << 11) | (0 << 6) | 4);     // HW_REG_HW_ID, all 32 bits
    const unsigned xcc = __builtin_amdgcn_s_getreg((3 << 11) | (0 << 6) | 20);    // HW_REG_XCC_ID bits [3:0]
    if ((threadIdx.x & 63) == 0) { out[(blockIdx.x * 4 + (threadIdx.x >> 6)) * 2] = hw; out[(blockIdx.x * 4 + (threadIdx.x >> 6)) * 2 + 1] = xcc; }
    const long t0 = wall_clock64();
    while (wall_clock64() - t0 < 200000) __builtin_amdgcn_s_sleep(32);
    if (smem[threadIdx.x] == 0) out[0] = 0;
}
int main() {
    const int nb = 512;
    unsigned *d; hipMalloc(&d, nb * 8 * 4);
    hipFuncSetAttribute((const void *)probe, hipFuncAttributeMaxDynamicSharedMemorySize, 81920);
    hipLaunchKernelGGL(probe, dim3(nb), dim3(256), 81920, 0, d);
    hipDeviceSynchronize();
    std::vector<unsigned> h(nb * 8); hipMemcpy(h.data(), d, nb * 8 * 4, hipMemcpyDeviceToHost);
    std::map<std::tuple<unsigned, unsigned, unsigned, unsigned>, std::vector<std::tuple<int, int, unsigned, unsigned>>> cu;
    int simd_eq_wave = 0;
    for (int b = 0; b < nb; b++) for (int w = 0; w < 4; w++) {
        unsigned hw = h[(b * 4 + w) * 2], xcc = h[(b * 4 + w) * 2 + 1] & 15;
        unsigned wave_id = hw & 15, simd = (hw >> 4) & 3, cuid = (hw >> 8) & 15, sh = (hw >> 12) & 1, se = (hw >> 13) & 7;
        cu[{xcc, se, sh, cuid}].push_back({b, w, simd, wave_id});
        simd_eq_wave += simd == (unsigned)w;
    }
    printf("CUs seen: %zu; waves whose SIMD == wave index: %d of %d\n", cu.size(), simd_eq_wave, nb * 4);
    int shown = 0;
    for (auto &kv : cu) {
        if (shown++ >= 6) break;
        printf("xcc %u se %u sh %u cu %u:", std::get<0>(kv.first), std::get<1>(kv.first), std::get<2>(kv.first), std::get<3>(kv.first));
        for (auto &t : kv.second) printf(" [wg %d wave %d -> simd %u slot %u]", std::get<0>(t), std::get<1>(t), std::get<2>(t), std::get<3>(t));
        printf("\n");
    }
    // histogram: for workgroup pairs sharing a CU, do their wave 0s share a SIMD?
    int share = 0, pairs = 0;
    for (auto &kv : cu) {
        std::map<int, unsigned> w0;
        for (auto &t : kv.second) if (std::get<1>(t) == 0) w0[std::get<0>(t)] = std::get<2>(t);
        if (w0.size() == 2) { pairs++; auto it = w0.begin(); unsigned a = it->second; ++it; share += a == it->second; }
    }
    printf("CUs with two workgroups: %d; wave 0 of both on the same SIMD: %d\n", pairs, share);
    return 0;
}
