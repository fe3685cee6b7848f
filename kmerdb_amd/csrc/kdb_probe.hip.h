// kdb_probe.hip.h -- what the memory system of THIS device delivers for the access patterns of the paged-scatter kernels with no compute at all
// (no LDS work, no ids, no barriers): the ceilings the fractions of DESIGN.md section 4 are read against.  Used by kdb_hbm_pattern_probe
// (bench.py prints the ceilings next to the kernels' own rates, same box, same run) and by tools/ubench_hbm_pattern.hip (the full table).
// A wave's step = NR read instructions + NW write instructions:
//   read   STREAM: 1 KiB of consecutive bytes per instruction (16 B per lane), the waves' blocks interleaved
//          PAGES:  one whole page (1 KiB, or 1.5 KiB as 1 KiB + 512 B like the u24 pages) at a pseudo-random page number
//   write  STREAM: 1 KiB of consecutive bytes per instruction
//          LINES:  sixteen 64-byte lines per instruction (four lanes each), every line at a pseudo-random line number of the region --
//                  what a ring flush of rounds 2-4 stored (rings_flush_wave), plain or write-through (sc1)
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace kdbprobe {

enum { R_NONE = 0, R_STREAM = 1, R_PAGES = 2, R_PAGES15 = 3 };
enum { W_NONE = 0, W_STREAM = 1, W_LINES = 2, W_LINES_SC1 = 3, W_RINGS_SC1 = 4, W_RINGS = 5, W_CHUNK128_SC1 = 6, W_CHUNK256_SC1 = 7, W_CHUNK512_SC1 = 8 };
//          CHUNKn: like LINES, in pieces of n bytes (8, 4 or 2 pieces per instruction)
//          RINGS:  the lines go where the scatter kernels' rings put them: a workgroup owns a contiguous range of 1 KiB pages, each of its 512 rings
//                  (64 per wave) fills its page line by line and then takes the next page of the range; one instruction writes the next line of
//                  sixteen of the wave's rings -- every line to another page, the sixteen lines of a page sixteen instructions apart

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
            if (WMODE == W_RINGS_SC1 || WMODE == W_RINGS) {
                // ring = 64 x (wave of the workgroup) + 16 x (w-th quarter) + lane / 4; its c-th line: page (c / 16) x 512 + ring of the workgroup's range, line c % 16
                const uint32_t wg_waves = blockDim.x >> 6, wwave = threadIdx.x >> 6;
                const uint64_t c = ((uint64_t)s * NW + w) / 4u;
                const uint32_t ring = wwave * 64u + (uint32_t)(((uint64_t)s * NW + w) & 3u) * 16u + (lane >> 2);
                const uint64_t wg_bytes = dst_bytes / gridDim.x / 1024ull * 1024ull;
                const uint64_t page = (c / 16u) * (uint64_t)(wg_waves * 64u) + ring;
                const uint64_t off = (page * 1024ull + (c & 15u) * 64ull) % wg_bytes;
                store16(dst + (uint64_t)blockIdx.x * wg_bytes + off + (lane & 3u) * 16u, v, WMODE == W_RINGS_SC1);
            }
            if (WMODE == W_CHUNK128_SC1 || WMODE == W_CHUNK256_SC1 || WMODE == W_CHUNK512_SC1) {
                const uint32_t lanes = WMODE == W_CHUNK128_SC1 ? 8u : WMODE == W_CHUNK256_SC1 ? 16u : 32u, per = 64u / lanes;     // lanes per piece, pieces per instruction
                const uint32_t piece = mix(((uint32_t)n * per + lane / lanes) * 2246822519u + 777u) % (uint32_t)(dst_bytes / (lanes * 16u));
                store16(dst + (uint64_t)piece * (lanes * 16u) + (lane % lanes) * 16u, v, true);
            }
            if (WMODE == W_LINES || WMODE == W_LINES_SC1) {
                const uint32_t line = mix(((uint32_t)n * 16u + (lane >> 2)) * 2246822519u + 777u) % nlines;
                store16(dst + (uint64_t)line * 64ull + (lane & 3u) * 16u, v, WMODE == W_LINES_SC1);
            }
        }
    }
    if (acc == 0x12345678u && sink) sink[0] = acc;
}


}  // namespace kdbprobe
