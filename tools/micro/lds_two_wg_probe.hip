// Do all DS instruction forms address the whole 80 KiB of LDS of a workgroup when TWO such workgroups share a CU (the second one's
// allocation starts at 80 KiB and ends at 160 KiB -- beyond the 64 KiB / 128 KiB marks older LDS sizes ended at)?
// Every workgroup writes a pattern derived from its id over its whole allocation with one store width, all workgroups spin until
// everyone has written (so that co-resident workgroups overlap in time), then each reads its allocation back with every load width and
// with LDS atomics. Any mismatch = a DS form that wraps or clips. build: hipcc --offload-arch=gfx950 -O2 ... ; run on the GPU box.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
constexpr int LDS_BYTES = 81920;
__device__ __forceinline__ unsigned pat(unsigned wg, unsigned i) { return (wg * 2654435761u) ^ (i * 40503u + 0x9e3779b9u); }

template <int MODE>
__global__ void __launch_bounds__(256, 2) probe(unsigned long long *bad, unsigned *firstbad, int rounds) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const unsigned wg = blockIdx.x, tid = threadIdx.x;
    unsigned *w32 = reinterpret_cast<unsigned *>(smem);
    const int n32 = LDS_BYTES / 4;
    unsigned long long nbad = 0;
    for (int r = 0; r < rounds; r++) {
        const unsigned salt = wg + 7919u * r;
        // ---- write with one width
        if (MODE == 0) { for (int i = tid; i < LDS_BYTES; i += 256) smem[i] = (unsigned char)(pat(salt, i >> 2) >> (8 * (i & 3))); }
        if (MODE == 1) { unsigned short *p = reinterpret_cast<unsigned short *>(smem); for (int i = tid; i < LDS_BYTES / 2; i += 256) p[i] = (unsigned short)(pat(salt, i >> 1) >> (16 * (i & 1))); }
        if (MODE == 2) { for (int i = tid; i < n32; i += 256) w32[i] = pat(salt, i); }
        if (MODE == 3) { unsigned long long *p = reinterpret_cast<unsigned long long *>(smem); for (int i = tid; i < n32 / 2; i += 256) p[i] = (unsigned long long)pat(salt, 2 * i) | ((unsigned long long)pat(salt, 2 * i + 1) << 32); }
        if (MODE == 4) { uint4 *p = reinterpret_cast<uint4 *>(smem); for (int i = tid; i < n32 / 4; i += 256) p[i] = make_uint4(pat(salt, 4 * i), pat(salt, 4 * i + 1), pat(salt, 4 * i + 2), pat(salt, 4 * i + 3)); }
        if (MODE == 5) { for (int i = tid; i < n32; i += 256) w32[i] = 0; __syncthreads(); for (int i = tid; i < n32; i += 256) atomicAdd(&w32[i], pat(salt, i)); }
        if (MODE == 6) { for (int i = tid; i < n32; i += 256) w32[i] = 0; __syncthreads(); for (int i = tid; i < n32; i += 256) atomicMax(&w32[i], pat(salt, i)); }
        if (MODE == 7) { unsigned long long *p = reinterpret_cast<unsigned long long *>(smem); for (int i = tid; i < n32 / 2; i += 256) p[i] = 0; __syncthreads();
                         for (int i = tid; i < n32 / 2; i += 256) atomicMax(&p[i], (unsigned long long)pat(salt, 2 * i) | ((unsigned long long)pat(salt, 2 * i + 1) << 32)); }
        if (MODE == 8) { for (int i = tid; i < n32; i += 256) w32[i] = ~0u; __syncthreads(); for (int i = tid; i < n32; i += 256) { unsigned o = atomicMin(&w32[i], pat(salt, i)); if (o != ~0u) nbad++; } }
        if (MODE == 9) { unsigned short *p = reinterpret_cast<unsigned short *>(smem); for (int i = tid; i < LDS_BYTES / 2; i += 256) p[(i * 37) % (LDS_BYTES / 2)] = (unsigned short)(pat(salt, ((i * 37) % (LDS_BYTES / 2)) >> 1) >> (16 * (((i * 37) % (LDS_BYTES / 2)) & 1))); }
        __syncthreads();
        // some time for the co-resident workgroup to run its own round over the same physical LDS
        for (int k = 0; k < 4; k++) __builtin_amdgcn_s_sleep(32);
        __syncthreads();
        // ---- read back with every width
        for (int i = tid; i < n32; i += 256) if (w32[i] != pat(salt, i)) { nbad++; if (atomicAdd(firstbad, 1u) < 6) printf("mode %d wg %u round %d: u32[%d] = %08x want %08x\n", MODE, wg, r, i, w32[i], pat(salt, i)); }
        for (int i = tid; i < LDS_BYTES; i += 256) if (smem[i] != (unsigned char)(pat(salt, i >> 2) >> (8 * (i & 3)))) nbad++;
        { const unsigned short *p = reinterpret_cast<const unsigned short *>(smem); for (int i = tid; i < LDS_BYTES / 2; i += 256) if (p[i] != (unsigned short)(pat(salt, i >> 1) >> (16 * (i & 1)))) nbad++; }
        { const unsigned long long *p = reinterpret_cast<const unsigned long long *>(smem); for (int i = tid; i < n32 / 2; i += 256) if (p[i] != ((unsigned long long)pat(salt, 2 * i) | ((unsigned long long)pat(salt, 2 * i + 1) << 32))) nbad++; }
        { const uint4 *p = reinterpret_cast<const uint4 *>(smem); for (int i = tid; i < n32 / 4; i += 256) { uint4 v = p[i]; if (v.x != pat(salt, 4 * i) || v.y != pat(salt, 4 * i + 1) || v.z != pat(salt, 4 * i + 2) || v.w != pat(salt, 4 * i + 3)) nbad++; } }
        __syncthreads();
    }
    if (nbad) atomicAdd(bad, nbad);
}
template <int MODE> void run(int grid, int rounds) {
    unsigned long long *bad; unsigned *fb;
    (void)hipMalloc(&bad, 8); (void)hipMalloc(&fb, 4); (void)hipMemset(bad, 0, 8); (void)hipMemset(fb, 0, 4);
    (void)hipFuncSetAttribute((const void *)probe<MODE>, hipFuncAttributeMaxDynamicSharedMemorySize, LDS_BYTES);
    hipLaunchKernelGGL(probe<MODE>, dim3(grid), dim3(256), LDS_BYTES, 0, bad, fb, rounds);
    (void)hipDeviceSynchronize();
    unsigned long long h; (void)hipMemcpy(&h, bad, 8, hipMemcpyDeviceToHost);
    static const char *names[] = {"store u8", "store u16", "store u32", "store u64", "store u128", "atomicAdd u32", "atomicMax u32", "atomicMax u64", "atomicMin u32 (rtn)", "scattered u16"};
    printf("%-20s grid %d: %llu mismatches\n", names[MODE], grid, h);
    (void)hipFree(bad); (void)hipFree(fb);
}
int main() {
    for (int grid : {512, 256}) { run<0>(grid, 40); run<1>(grid, 40); run<2>(grid, 40); run<3>(grid, 40); run<4>(grid, 40); run<5>(grid, 40); run<6>(grid, 40); run<7>(grid, 40); run<8>(grid, 40); run<9>(grid, 40); }
    return 0;
}
