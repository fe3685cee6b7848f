// micro-benchmarks: LDS atomic / read / write rates on gfx950 for the access shapes the histogram kernels use
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <vector>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)

__device__ __forceinline__ uint32_t mix(uint32_t x) { x ^= x >> 16; x *= 0x7feb352du; x ^= x >> 15; x *= 0x846ca68bu; x ^= x >> 16; return x; }

template <int MODE, int THREADS>
__global__ void __launch_bounds__(THREADS) k(uint32_t *out, int iters, uint32_t nbins_mask)
{
    extern __shared__ uint32_t lds[];
    const int tid = threadIdx.x;
    for (uint32_t i = tid; i <= nbins_mask; i += THREADS) lds[i] = 0;
    __syncthreads();
    uint32_t s = mix(blockIdx.x * THREADS + tid + 1), acc = 0;
    for (int it = 0; it < iters; it++) {
#pragma unroll
        for (int u = 0; u < 8; u++) {
            s = s * 1664525u + 1013904223u;
            uint32_t a;
            if (MODE == 0 || MODE == 10) a = (tid + u * THREADS) & nbins_mask;         // consecutive, conflict-free
            else a = (s >> 8) & nbins_mask;                                            // random
            if (MODE == 0 || MODE == 1) atomicAdd(&lds[a], 1u);                        // no-return atomic
            else if (MODE == 2 || MODE == 10) acc += atomicAdd(&lds[a], 1u);           // returning atomic
            else if (MODE == 3) acc += lds[a];                                         // random read
            else if (MODE == 4) ((uint16_t *)lds)[a] = (uint16_t)s;                    // random 2-byte write
            else if (MODE == 5) { if (a != (uint32_t)tid) atomicAdd(&lds[a & ~31u | (tid & 31)], 1u); }   // random row, own bank
            else if (MODE == 6) atomicAdd((unsigned long long *)&lds[(a & ~1u)], 1ull);      // 64-bit atomic random
            else if (MODE == 11) acc += atomicAdd(&lds[a & ~31u | (tid & 31)], 1u);          // returning atomic: random row, own bank
        }
    }
    __syncthreads();
    uint32_t t = acc;
    for (uint32_t i = tid; i <= nbins_mask; i += THREADS) t += lds[i];
    if (t == 0xdeadbeef) out[0] = t;
}

template <int MODE, int THREADS>
int run(const char *name, int nbins, int wgs_per_cu)
{
    uint32_t *d; CK(hipMalloc(&d, 4));
    const int iters = 512, grid = 256 * wgs_per_cu;
    size_t lds = (size_t)nbins * 4;
    CK(hipFuncSetAttribute((const void *)k<MODE, THREADS>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    hipEvent_t a, b; CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
    hipLaunchKernelGGL((k<MODE, THREADS>), dim3(grid), dim3(THREADS), lds, 0, d, 8, (uint32_t)(nbins - 1));
    CK(hipDeviceSynchronize());
    CK(hipEventRecord(a));
    hipLaunchKernelGGL((k<MODE, THREADS>), dim3(grid), dim3(THREADS), lds, 0, d, iters, (uint32_t)(nbins - 1));
    CK(hipEventRecord(b)); CK(hipEventSynchronize(b));
    float ms; CK(hipEventElapsedTime(&ms, a, b));
    double ops = (double)grid * THREADS * iters * 8;
    printf("%-44s bins=%6d thr=%4d wg/cu=%d : %8.1f Gops/s  (%.2f ops/clk/CU @2.4GHz)  %.3f ms\n", name, nbins, THREADS, wgs_per_cu,
           ops / ms / 1e6, ops / ms / 1e6 / 256 / 2.4, ms);
    CK(hipFree(d));
    return 0;
}

int main()
{
    run<0, 256>("atomic noret consecutive", 32768, 1);
    run<0, 1024>("atomic noret consecutive", 32768, 1);
    run<1, 256>("atomic noret random", 32768, 1);
    run<1, 1024>("atomic noret random", 32768, 1);
    run<1, 256>("atomic noret random", 512, 4);
    run<1, 256>("atomic noret random", 512, 8);
    run<1, 256>("atomic noret random", 8192, 4);
    run<2, 256>("atomic ret random", 512, 4);
    run<2, 256>("atomic ret random", 512, 8);
    run<2, 1024>("atomic ret random", 32768, 1);
    run<10, 256>("atomic ret consecutive", 512, 8);
    run<2, 1024>("atomic ret random (the ring words of a 1024-thread scatter workgroup)", 512, 1);
    run<2, 1024>("atomic ret random", 128, 1);
    run<11, 1024>("atomic ret random row, own bank", 512, 1);
    run<11, 1024>("atomic ret random row, own bank", 32768, 1);
    run<4, 1024>("write16 random", 65536, 1);
    run<3, 256>("read random", 512, 8);
    run<3, 1024>("read random", 32768, 1);
    run<4, 256>("write16 random", 8192, 8);
    run<5, 1024>("atomic noret random row, own bank", 32768, 1);
    run<6, 1024>("atomic64 noret random", 32768, 1);
    return 0;
}
