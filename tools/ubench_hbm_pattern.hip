// What can the HBM of an MI355X deliver for the ACCESS PATTERNS of the paged-scatter kernels -- with no LDS work, no ids, no barriers?
//     hipcc --offload-arch=gfx950 -O3 -o tools/ubench_hbm_pattern tools/ubench_hbm_pattern.hip && tools/ubench_hbm_pattern
// A wave's step = NR read instructions + NW write instructions:
//   read   STREAM: 1 KiB of consecutive bytes per instruction (16 B per lane), the waves' blocks interleaved
//          PAGES:  one whole page (1 KiB, or 1.5 KiB as 1 KiB + 512 B like the u24 pages) at a pseudo-random page number
//   write  STREAM: 1 KiB of consecutive bytes per instruction
//          LINES:  sixteen 64-byte lines per instruction (four lanes each), every line at a pseudo-random line number of the region --
//                  what a ring flush stores (rings_flush_wave), plain or write-through (sc1)
// The rows printed: bytes read + written / time.  These are the ceilings the fractions of DESIGN.md section 4 should be read against.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstdint>

enum { R_NONE = 0, R_STREAM = 1, R_PAGES = 2, R_PAGES15 = 3 };
enum { W_NONE = 0, W_STREAM = 1, W_LINES = 2, W_LINES_SC1 = 3 };

__device__ __forceinline__ uint32_t mix(uint32_t x)
{
    x ^= x >> 16; x *= 0x7FEB352Du; x ^= x >> 15; x *= 0x846CA68Bu; x ^= x >> 16;
    return x;
}

__device__ __forceinline__ void store16(uint8_t *p, uint4 x, bool sc1)
{
    typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
    const u32x4 v = {x.x, x.y, x.z, x.w};
    if (sc1) asm volatile("global_store_dwordx4 %0, %1, off sc1" : : "v"(p), "v"(v) : "memory");
    else *reinterpret_cast<uint4 *>(p) = x;
}

template <int RMODE, int WMODE, int NR, int NW>
__global__ void __launch_bounds__(512)
pattern(const uint8_t *__restrict__ src, uint64_t src_bytes, uint8_t *__restrict__ dst, uint64_t dst_bytes, uint32_t steps, uint32_t *sink)
{
    const uint32_t lane = threadIdx.x & 63u, wave = (blockIdx.x * blockDim.x + threadIdx.x) >> 6, nwaves = (gridDim.x * blockDim.x) >> 6;
    const uint64_t page_bytes = RMODE == R_PAGES15 ? 1536 : 1024;
    const uint32_t npages = (uint32_t)(src_bytes / page_bytes), nlines = (uint32_t)(dst_bytes / 64);
    uint32_t acc = 0;
    for (uint32_t s = 0; s < steps; s++) {
        uint4 x[NR > 0 ? NR : 1];
        uint2 y[NR > 0 ? NR : 1];
#pragma unroll
        for (int r = 0; r < NR; r++) {
            const uint64_t n = ((uint64_t)s * NR + r) * nwaves + wave;                 // the n-th KiB / page of the run
            if (RMODE == R_STREAM) x[r] = *reinterpret_cast<const uint4 *>(src + (n * 1024ull) % src_bytes + lane * 16u);
            if (RMODE == R_PAGES || RMODE == R_PAGES15) {
                const uint8_t *pg = src + (uint64_t)(mix((uint32_t)n * 2654435761u + 12345u) % npages) * page_bytes;
                x[r] = reinterpret_cast<const uint4 *>(pg)[lane];
                if (RMODE == R_PAGES15) y[r] = reinterpret_cast<const uint2 *>(pg + 1024)[lane];
            }
        }
#pragma unroll
        for (int r = 0; r < NR; r++) { acc ^= x[r].x ^ x[r].w; if (RMODE == R_PAGES15) acc ^= y[r].y; }
#pragma unroll
        for (int w = 0; w < NW; w++) {
            const uint64_t n = ((uint64_t)s * NW + w) * nwaves + wave;
            const uint4 v = make_uint4(acc, lane, s, w);
            if (WMODE == W_STREAM) *reinterpret_cast<uint4 *>(dst + (n * 1024ull) % dst_bytes + lane * 16u) = v;
            if (WMODE == W_LINES || WMODE == W_LINES_SC1) {
                const uint32_t line = mix(((uint32_t)n * 16u + (lane >> 2)) * 2246822519u + 777u) % nlines;
                store16(dst + (uint64_t)line * 64ull + (lane & 3u) * 16u, v, WMODE == W_LINES_SC1);
            }
        }
    }
    if (acc == 0x12345678u && sink) sink[0] = acc;
}

template <int RMODE, int WMODE, int NR, int NW>
static void run(const char *name, const uint8_t *src, uint64_t src_bytes, uint8_t *dst, uint64_t dst_bytes, uint32_t *sink, int grid, int threads)
{
    const uint64_t per_step = (uint64_t)grid * (threads / 64) * ((uint64_t)NR * (RMODE == R_PAGES15 ? 1536 : RMODE ? 1024 : 0) + (uint64_t)NW * (WMODE ? 1024 : 0));
    const uint32_t steps = (uint32_t)(8.0e9 / (double)per_step) + 1;
    hipEvent_t a, b;
    hipEventCreate(&a); hipEventCreate(&b);
    float best = 1e30f;
    for (int rep = 0; rep < 4; rep++) {
        hipEventRecord(a);
        hipLaunchKernelGGL((pattern<RMODE, WMODE, NR, NW>), dim3(grid), dim3(threads), 0, 0, src, src_bytes, dst, dst_bytes, steps, sink);
        hipEventRecord(b);
        hipEventSynchronize(b);
        float ms; hipEventElapsedTime(&ms, a, b);
        if (rep && ms < best) best = ms;
    }
    const double rb = (double)steps * grid * (threads / 64) * NR * (RMODE == R_PAGES15 ? 1536 : RMODE ? 1024 : 0), wb = (double)steps * grid * (threads / 64) * NW * (WMODE ? 1024 : 0);
    printf("%-86s grid %4d x %4d  %7.3f ms  read %6.2f GB write %6.2f GB  -> %6.0f GB/s (%.2f of 8 TB/s)\n", name, grid, threads, best, rb / 1e9, wb / 1e9,
           (rb + wb) / 1e6 / best, (rb + wb) / 1e6 / best / 8000.0);
    fflush(stdout);
    hipEventDestroy(a); hipEventDestroy(b);
}

int main()
{
    const uint64_t src_bytes = 6ull << 30, dst_bytes = 6ull << 30;
    uint8_t *src, *dst; uint32_t *sink;
    if (hipMalloc(&src, src_bytes) != hipSuccess || hipMalloc(&dst, dst_bytes) != hipSuccess || hipMalloc(&sink, 64) != hipSuccess) { fprintf(stderr, "hipMalloc failed\n"); return 1; }
    hipMemset(src, 1, src_bytes); hipMemset(dst, 0, dst_bytes);
    hipDeviceSynchronize();
    for (int cfg = 0; cfg < 2; cfg++) {
        const int grid = cfg == 0 ? 512 : 2048, threads = cfg == 0 ? 512 : 256;          // 16 waves per CU either way; persistent-style vs many small workgroups
        printf("-- %d workgroups of %d threads\n", grid, threads);
        run<R_STREAM, W_NONE, 4, 0>("stream read", src, src_bytes, dst, dst_bytes, sink, grid, threads);
        run<R_NONE, W_STREAM, 0, 4>("stream write", src, src_bytes, dst, dst_bytes, sink, grid, threads);
        run<R_STREAM, W_STREAM, 2, 2>("stream copy", src, src_bytes, dst, dst_bytes, sink, grid, threads);
        run<R_PAGES, W_NONE, 4, 0>("random 1 KiB pages read, four in flight per wave (histogram pass)", src, src_bytes, dst, dst_bytes, sink, grid, threads);
        run<R_PAGES15, W_NONE, 2, 0>("random 1.5 KiB pages read, two in flight per wave", src, src_bytes, dst, dst_bytes, sink, grid, threads);
        run<R_NONE, W_LINES, 0, 2>("random 64-byte lines written (plain stores)", src, src_bytes, dst, dst_bytes, sink, grid, threads);
        run<R_NONE, W_LINES_SC1, 0, 2>("random 64-byte lines written (sc1: write-through)", src, src_bytes, dst, dst_bytes, sink, grid, threads);
        run<R_STREAM, W_LINES_SC1, 1, 2>("k = 12 scatter: 1 KiB streamed in per 2 KiB of random lines out (sc1)", src, src_bytes, dst, dst_bytes, sink, grid, threads);
        run<R_STREAM, W_LINES_SC1, 1, 3>("level 1 (u24): 1 KiB streamed in per 3 KiB of random lines out (sc1)", src, src_bytes, dst, dst_bytes, sink, grid, threads);
        run<R_PAGES15, W_LINES_SC1, 2, 2>("level 2: two random 1.5 KiB pages in per 2 KiB of random lines out (sc1)", src, src_bytes, dst, dst_bytes, sink, grid, threads);
        run<R_PAGES15, W_LINES, 2, 2>("level 2, plain stores", src, src_bytes, dst, dst_bytes, sink, grid, threads);
    }
    // the same patterns inside regions small enough for the memory-side cache (256 MB "Infinity Cache"): does a working set that fits change the ceilings?
    printf("-- small regions (512 workgroups of 512 threads)\n");
    for (uint64_t mb : {32ull, 64ull, 128ull, 192ull, 512ull, 2048ull}) {
        char name[128];
        snprintf(name, sizeof name, "region %4llu MB: random 1 KiB pages read", (unsigned long long)mb);
        run<R_PAGES, W_NONE, 4, 0>(name, src, mb << 20, dst, mb << 20, sink, 512, 512);
        snprintf(name, sizeof name, "region %4llu MB: random 64-byte lines written (plain)", (unsigned long long)mb);
        run<R_NONE, W_LINES, 0, 2>(name, src, mb << 20, dst, mb << 20, sink, 512, 512);
        snprintf(name, sizeof name, "region %4llu MB: random 64-byte lines written (sc1)", (unsigned long long)mb);
        run<R_NONE, W_LINES_SC1, 0, 2>(name, src, mb << 20, dst, mb << 20, sink, 512, 512);
        snprintf(name, sizeof name, "region %4llu MB + %4llu MB: pages in, lines out (plain)", (unsigned long long)mb, (unsigned long long)mb);
        run<R_PAGES, W_LINES, 2, 2>(name, src, mb << 20, dst, mb << 20, sink, 512, 512);
    }
    return 0;
}
