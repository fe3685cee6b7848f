// kdb_hist.hip.h -- what the LDS-histogram paths of the engine share (gfx950).
//
// Direct 64-bit global atomics on uniformly random ids are bound by the memory-side atomic rate (device-scope
// RMWs execute at the memory side on gfx950: ~23 G random atomics/s measured, MI355X_MICROARCH.md "Global float
// atomics").  The LDS-histogram paths replace the per-k-mer global atomic by per-k-mer LDS atomics:
//
//   k <= 7          count_lds_kernel (here): the whole 4^k vector lives in LDS (<= 64 KiB of u32), persistent
//                   workgroups, one global flush at the end.
//   8 <= k <= 17    kdb_scatter.hip.h: ids are scattered into buckets of 32768 bins through LDS write-combining rings
//                   and pages in HBM (one level for k <= 12, two for k >= 13); one 32768-bin LDS histogram per bucket.
//
// Same counting semantics as count_direct_kernel (kmer.py:234-317, :526-565; parse.py:133-136).  Windows containing
// N in EXPAND mode go to the vector through expand_n_window (in place or via the work list).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "../../include/kdbhip.h"
#include "kdb_kernels.hip.h"

namespace kdb {

constexpr int BIN_BITS = 15;                      // 32768 u32 bins = 128 KiB of LDS per bucket histogram
constexpr int BUCKET_BINS = 1 << BIN_BITS;
constexpr int P2_THREADS = 1024;
constexpr int SMALLK_MAX = 7;

struct ProfHook {
    virtual void begin(int kernel) = 0;
    virtual void begin_on(int kernel, hipStream_t) { begin(kernel); }     // the span's events on another stream than the engine's compute stream
    virtual void end() = 0;
    virtual ~ProfHook() {}
};

inline const char *&partition_error_ref() { static thread_local const char *msg = ""; return msg; }
inline const char *partition_error() { return partition_error_ref(); }

// visit every counted window of the staged tile owned by this lane:
//   f(id)                 for a clean window
//   g(F, i, nwin)         for a window whose only defects are N's (EXPAND mode)
template <bool EXPAND, int THREADS = TPB, typename FClean, typename FN>
__device__ __forceinline__ void for_each_window(const TileLds<EXPAND> &L, int k, int canonical, FClean f, FN g)
{
    const int j = threadIdx.x;
    const IdParams<uint32_t> idp(k, canonical);
    const uint32_t kmask = (1u << k) - 1u;
    const uint32_t k1mask = kmask >> 1;
#pragma unroll 1
    for (int q = 0; q < TILE_CHUNKS / THREADS; q++) {
        const int c = j + q * THREADS;
        const Hood h = load_hood(L, c);
        uint32_t N32 = 0;
        if (EXPAND) N32 = (L.nn[c] & 0xFFFFu) | (L.nn[c + 1] << 16);
        // Degenerate stretch?  (poly-A/G reads, microsatellites: the 64 lanes of the wave, 16 bases apart, see the same
        // k-mer.)  One wave-uniform test per 16 windows; only then do the per-window same-key shortcuts run.
        uint64_t same; uint32_t id0;
        const bool degenerate = wave_dominant(idp.id(h, 0), &same, &id0);
        if (!degenerate) {                     // the hot loop: straight-line, no calls
#pragma unroll
            for (int i = 0; i < 16; i++) {
                const bool crosses = window_crosses(h, i, k1mask);
                const uint32_t vwin = (h.V >> i) & kmask;
                if (vwin == 0 && !crosses) {
                    f(idp.id(h, i), false);
                } else if (EXPAND && !crosses) {
                    const uint32_t nwin = (N32 >> i) & kmask;
                    if (nwin == vwin) g(h.F(), i, nwin);
                }
            }
        } else {
#pragma unroll 1
            for (int i = 0; i < 16; i++) {
                const bool crosses = window_crosses(h, i, k1mask);
                const uint32_t vwin = (h.V >> i) & kmask;
                if (vwin == 0 && !crosses) {
                    f(idp.id_dyn(h, i), true);
                } else if (EXPAND && !crosses) {
                    const uint32_t nwin = (N32 >> i) & kmask;
                    if (nwin == vwin) g(h.F(), i, nwin);
                }
            }
        }
    }
}

// ---------------------------------------------------------------------------------
// k <= 7: whole vector in LDS
// ---------------------------------------------------------------------------------
template <bool EXPAND>
__global__ void __launch_bounds__(TPB)
count_lds_kernel(const uint8_t *__restrict__ bases, uint64_t nbytes, uint32_t ntiles, int k, int canonical,
                 unsigned long long *__restrict__ table, DevCounters *ctr)
{
    __shared__ TileLds<EXPAND> L;
    __shared__ uint32_t hist[1 << (2 * SMALLK_MAX)];
    __shared__ unsigned long long s_tot[3];
    const int j = threadIdx.x;
    const uint32_t nbins = 1u << (2 * k);
    for (uint32_t i = j; i < nbins; i += TPB) hist[i] = 0;
    if (j < 3) s_tot[j] = 0;
    unsigned long long emitted = 0, nbad_tot = 0, nmark_tot = 0;
    const uint64_t idmask = (1ull << (2 * k)) - 1ull;
    const UniformStarts ulen(batch_uniform_len(ctr), TPB);
    for (uint32_t t = blockIdx.x; t < ntiles; t += gridDim.x) {
        uint32_t nbad;
        __syncthreads();                                   // previous tile fully consumed (and hist zeroed)
        stage_tile(L, bases, nbytes, t, &nbad, ulen, ctr);
        nbad_tot += nbad & 0xFFFFu; nmark_tot += nbad >> 16;
        __syncthreads();
        for_each_window(L, k, canonical,
            [&](uint32_t id, bool deg) { if (deg) lds_hist_add(hist, id); else atomicAdd(&hist[id], 1u); emitted++; },
            [&](uint64_t F, int i, uint32_t nwin) { expand_n_window(table, F, i, k, canonical, idmask, nwin, &emitted, ctr); });
    }
    __syncthreads();
    for (uint32_t i = j; i < nbins; i += TPB) {
        uint32_t c = hist[i];
        if (c) __hip_atomic_fetch_add(&table[i], (unsigned long long)c, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
    unsigned long long wt = wave_sum(emitted), wb = wave_sum(nbad_tot), wm = wave_sum(nmark_tot);
    if ((j & 63) == 0) { if (wt) atomicAdd(&s_tot[0], wt); if (wb) atomicAdd(&s_tot[1], wb); if (wm) atomicAdd(&s_tot[2], wm); }
    __syncthreads();
    if (j == 0) {
        if (s_tot[0]) __hip_atomic_fetch_add(&ctr->total_kmers, s_tot[0], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if (s_tot[1]) __hip_atomic_fetch_add(&ctr->n_bad, s_tot[1], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if (s_tot[2]) __hip_atomic_fetch_add(&ctr->marks_seen, s_tot[2], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
}

// block-wide exclusive scan helper: returns the exclusive prefix of v, total in *tot
template <int THREADS>
__device__ __forceinline__ uint32_t block_excl_scan(uint32_t v, uint32_t *wsum /* LDS [THREADS/64] */, uint32_t *tot)
{
    const int j = threadIdx.x, lane = j & 63, wave = j >> 6;
    uint32_t s = v;
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) { uint32_t t = __shfl_up(s, o, 64); if (lane >= o) s += t; }
    __syncthreads();                       // wsum may still be read by a previous call
    if (lane == 63) wsum[wave] = s;
    __syncthreads();
    uint32_t woff = 0, total = 0;
#pragma unroll
    for (int w = 0; w < THREADS / 64; w++) { uint32_t x = wsum[w]; if (w < wave) woff += x; total += x; }
    *tot = total;
    return woff + s - v;
}

// ---------------------------------------------------------------------------------
// P2: one LDS histogram per (bucket, slice); flush with contiguous 64-bit atomics
// ---------------------------------------------------------------------------------
__device__ __noinline__ void hist_add8_degenerate(uint32_t *hist, uint32_t a, uint32_t b, uint32_t c, uint32_t d)
{
    lds_hist_add(hist, a & 0xFFFFu); lds_hist_add(hist, a >> 16);
    lds_hist_add(hist, b & 0xFFFFu); lds_hist_add(hist, b >> 16);
    lds_hist_add(hist, c & 0xFFFFu); lds_hist_add(hist, c >> 16);
    lds_hist_add(hist, d & 0xFFFFu); lds_hist_add(hist, d >> 16);
}

__device__ __forceinline__ void hist_add8(uint32_t *hist, const uint4 &x)
{
    // eight remainders per 16-byte load; the same-key shortcut only runs when the first one looks degenerate
    uint64_t same; uint32_t k0;
    if (wave_dominant(x.x & 0xFFFFu, &same, &k0)) {
        hist_add8_degenerate(hist, x.x, x.y, x.z, x.w);
    } else {
        atomicAdd(&hist[x.x & 0xFFFFu], 1u); atomicAdd(&hist[x.x >> 16], 1u);
        atomicAdd(&hist[x.y & 0xFFFFu], 1u); atomicAdd(&hist[x.y >> 16], 1u);
        atomicAdd(&hist[x.z & 0xFFFFu], 1u); atomicAdd(&hist[x.z >> 16], 1u);
        atomicAdd(&hist[x.w & 0xFFFFu], 1u); atomicAdd(&hist[x.w >> 16], 1u);
    }
}

// workgroup index -> (bucket, slice, number of slices of that bucket): bucket = largest b with slice_base[b] <= wg
__device__ __forceinline__ bool p2_locate(const uint32_t *__restrict__ slice_base, uint32_t nbuckets, uint32_t wg,
                                          uint32_t *b, uint32_t *s, uint32_t *nslices)
{
    if (wg >= slice_base[nbuckets]) return false;          // the grid is an upper bound on the number of slices
    uint32_t lo = 0, hi = nbuckets - 1;
    while (lo < hi) { const uint32_t mid = (lo + hi + 1) >> 1; if (slice_base[mid] <= wg) lo = mid; else hi = mid - 1; }
    *b = lo; *s = wg - slice_base[lo]; *nslices = slice_base[lo + 1] - slice_base[lo];
    return true;
}

// ---------------------------------------------------------------------------------
// host: k <= 7
// ---------------------------------------------------------------------------------
inline int smallk_count(hipStream_t stream, const uint8_t *d_bases, size_t nbytes, int k, int canonical, int n_expand,
                        unsigned long long *d_table, DevCounters *d_ctr, ProfHook &prof)
{
    const uint64_t ntiles_all = (nbytes + TILE_BYTES - 1) / TILE_BYTES;
    if (ntiles_all > 0xFFFFFFFFull) { partition_error_ref() = "batch too large"; return 1; }
    const uint32_t grid = (uint32_t)(ntiles_all < 512 ? ntiles_all : 512);
    prof.begin(KDB_KERNEL_COUNT);
    if (n_expand)
        hipLaunchKernelGGL(count_lds_kernel<true>, dim3(grid), dim3(TPB), 0, stream, d_bases, (uint64_t)nbytes, (uint32_t)ntiles_all, k, canonical, d_table, d_ctr);
    else
        hipLaunchKernelGGL(count_lds_kernel<false>, dim3(grid), dim3(TPB), 0, stream, d_bases, (uint64_t)nbytes, (uint32_t)ntiles_all, k, canonical, d_table, d_ctr);
    prof.end();
    if (hipGetLastError() != hipSuccess) { partition_error_ref() = "count_lds_kernel failed to launch"; return 1; }
    return 0;
}

}  // namespace kdb
