// Does a workgroup barrier order GLOBAL memory between the waves of a workgroup on MI355X?
// 512 workgroups x 256 threads, 80 KiB of LDS each (two per CU, the simulator's shape). Per round: (A) all threads clear a byte array,
// barrier; (B) the first M threads set M scattered flags (so the second wave stores with a sparse exec mask), barrier; (C) all threads
// count the flags; a count != M is a store one wave made in front of the barrier that another wave did not see behind it.
// modes: 0 __syncthreads(); 1 s_waitcnt vmcnt(0) + __syncthreads(); 2 as 1 + buffer_inv sc1; 3 workgroup-scope atomic stores (sc0) + 1;
//        4 agent-scope atomic stores (sc1) + __syncthreads()
// build: hipcc --offload-arch=gfx950 -O3 tools/micro/barrier_visibility.hip -o tools/micro/barrier_visibility
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
template <int MODE> __device__ __forceinline__ void sync() {
    if (MODE == 1 || MODE == 2 || MODE == 3) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    if (MODE == 2) asm volatile("buffer_inv sc1" ::: "memory");
}
template <int MODE> __device__ __forceinline__ void put(unsigned char *p, unsigned char v) {
    if (MODE == 3) __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
    else if (MODE == 4) __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    else *p = v;
}
template <int MODE>
__global__ void __launch_bounds__(256, 2) probe(unsigned char *flags, int N, int rounds, int lds_work, unsigned long long *bad, unsigned *first) {
    extern __shared__ unsigned smem[];
    unsigned char *f = flags + (size_t)blockIdx.x * N;
    const int tid = threadIdx.x;
    unsigned long long nbad = 0;
    for (int r = 0; r < rounds; r++) {
        const int M = 65 + (r * 7 + blockIdx.x) % 60;                  // 65..124: the second wave stores with 1..60 lanes
        for (int o = tid; o < N; o += 256) put<MODE>(f + o, 0);
        if (tid == 0) smem[0] = 0;
        sync<MODE>();
        if (tid < M) {
            const unsigned o = ((unsigned)tid * 2654435761u + (unsigned)r * 40503u + blockIdx.x * 977u) % (unsigned)(N / 128) * 128u + (unsigned)tid;   // one flag per 128-byte line, distinct
            put<MODE>(f + o, 1);
        }
        // LDS traffic in between, as the simulator's phases have
        unsigned acc = 0;
        for (int k = 0; k < lds_work; k++) { smem[64 + ((tid * 33 + k * 257) & 16383)] = acc + k; acc += smem[64 + ((tid * 65 + k * 129) & 16383)]; }
        sync<MODE>();
        int cnt = 0;
        for (int o = tid; o < N; o += 256) cnt += f[o];
        atomicAdd(&smem[0], (unsigned)cnt);
        if (acc == 0x12345678u) smem[1] = acc;
        sync<MODE>();
        if (tid == 0 && (int)smem[0] != M) { nbad++; if (atomicAdd(first, 1u) < 8) printf("mode %d: wg %d round %d: counted %u of %d flags\n", MODE, blockIdx.x, r, smem[0], M); }
        sync<MODE>();
    }
    if (tid == 0 && nbad) atomicAdd(bad, nbad);
}
template <int MODE> void run(int grid, int rounds, int lds_work) {
    const int N = 16384;
    unsigned char *flags; unsigned long long *bad; unsigned *first;
    hipMalloc(&flags, (size_t)grid * N); hipMalloc(&bad, 8); hipMalloc(&first, 4);
    hipMemset(bad, 0, 8); hipMemset(first, 0, 4);
    hipFuncSetAttribute((const void *)probe<MODE>, hipFuncAttributeMaxDynamicSharedMemorySize, 81920);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    hipEventRecord(e0);
    hipLaunchKernelGGL(probe<MODE>, dim3(grid), dim3(256), 81920, 0, flags, N, rounds, lds_work, bad, first);
    hipEventRecord(e1); hipDeviceSynchronize();
    float ms; hipEventElapsedTime(&ms, e0, e1);
    unsigned long long h; hipMemcpy(&h, bad, 8, hipMemcpyDeviceToHost);
    printf("mode %d grid %d lds_work %d: %llu bad rounds of %lld (%.1f ms)\n", MODE, grid, lds_work, h, (long long)grid * rounds, ms);
    hipFree(flags); hipFree(bad); hipFree(first);
}
int main(int argc, char **argv) {
    const int rounds = argc > 1 ? atoi(argv[1]) : 2000;
    for (int lw : {0, 16}) {
        run<0>(512, rounds, lw); run<1>(512, rounds, lw); run<2>(512, rounds, lw); run<3>(512, rounds, lw); run<4>(512, rounds, lw);
        run<0>(256, rounds, lw); run<1>(256, rounds, lw);
    }
    return 0;
}
