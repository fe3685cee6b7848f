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
#include "../kmerdb_amd/csrc/kdb_probe.hip.h"
using namespace kdbprobe;

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
        run<R_NONE, W_RINGS_SC1, 0, 4>("lines written the way the rings place them (sc1)", src, src_bytes, dst, dst_bytes, sink, grid, threads);
        run<R_NONE, W_RINGS, 0, 4>("lines written the way the rings place them (plain: the L2 may put the halves of a 128-byte line together)", src, src_bytes, dst, dst_bytes, sink, grid, threads);
        run<R_NONE, W_CHUNK128_SC1, 0, 2>("random 128-byte pieces written (sc1)", src, src_bytes, dst, dst_bytes, sink, grid, threads);
        run<R_NONE, W_CHUNK256_SC1, 0, 2>("random 256-byte pieces written (sc1)", src, src_bytes, dst, dst_bytes, sink, grid, threads);
        run<R_NONE, W_CHUNK512_SC1, 0, 2>("random 512-byte pieces written (sc1)", src, src_bytes, dst, dst_bytes, sink, grid, threads);
        run<R_STREAM, W_CHUNK128_SC1, 1, 2>("1 KiB streamed in per 2 KiB of random 128-byte pieces out (sc1)", src, src_bytes, dst, dst_bytes, sink, grid, threads);
        run<R_STREAM, W_RINGS_SC1, 2, 4>("k = 12 scatter, ring placement: 1 KiB streamed in per 2 KiB of lines out (sc1)", src, src_bytes, dst, dst_bytes, sink, grid, threads);
        run<R_PAGES15, W_RINGS_SC1, 2, 2>("level 2, ring placement: two random 1.5 KiB pages in per 2 KiB of lines out (sc1)", src, src_bytes, dst, dst_bytes, sink, grid, threads);
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
