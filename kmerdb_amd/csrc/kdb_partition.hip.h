// kdb_partition.hip.h -- the LDS-histogram paths of the engine (gfx950).
//
// Direct 64-bit global atomics on uniformly random ids are bound by the
// memory-side atomic rate (MI355X_MICROARCH.md, "Global float atomics": 64
// lanes in 64 different lines run ~17x below the streaming rate).  These paths
// replace the per-k-mer global atomic by per-k-mer LDS atomics:
//
//   k <= 7          count_lds_kernel: the whole 4^k vector lives in LDS (<= 64 KiB
//                   of u32), persistent workgroups, one global flush at the end.
//   8 <= k <= 12    radix partition on the id's high bits into B = 4^k / 32768
//                   buckets, then one LDS histogram of 32768 u32 bins per bucket:
//                     P0 bucket_count_kernel   exact bucket sizes (ids recomputed, not stored)
//                     P0b bucket_scan_kernel   exclusive scan -> bucket bases / cursors
//                     P1 partition_kernel      ids -> 15-bit remainders, multisplit in LDS,
//                                              coalesced runs appended to the bucket arrays
//                     P2 bucket_hist_kernel    LDS histogram per bucket slice, flushed into
//                                              the uint64 vector with contiguous atomics
//
// Same counting semantics as count_direct_kernel (kmer.py:234-317, :526-565;
// parse.py:133-136); windows containing N in EXPAND mode are rare and go
// straight to the vector through expand_n_window.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "../../include/kdbhip.h"
#include "kdb_kernels.hip.h"

namespace kdb {

constexpr int BIN_BITS = 15;                      // 32768 u32 bins = 128 KiB of LDS per bucket histogram
constexpr int BUCKET_BINS = 1 << BIN_BITS;
constexpr int MAXB = 512;                         // buckets at k = 12
constexpr int P2_THREADS = 1024;
constexpr int PERSIST_GRID = 2048;
constexpr int SMALLK_MAX = 7;

struct ProfHook {
    virtual void begin(int kernel) = 0;
    virtual void end() = 0;
    virtual ~ProfHook() {}
};

struct PartitionState {
    uint16_t *d_elems = nullptr;          // bucketed 15-bit remainders
    size_t elems_cap = 0;                 // in elements
    unsigned long long *d_bucket_total = nullptr;   // [MAXB]
    uint32_t *d_bucket_base = nullptr;              // [MAXB + 1]
    uint32_t *d_bucket_cursor = nullptr;            // [MAXB]
};

inline const char *&partition_error_ref() { static thread_local const char *msg = ""; return msg; }
inline const char *partition_error() { return partition_error_ref(); }

inline bool partition_supported(int k, int /*n_mode*/) { return k >= 1 && k <= 12; }

inline void partition_free(PartitionState &st)
{
    if (st.d_elems) (void)hipFree(st.d_elems);
    if (st.d_bucket_total) (void)hipFree(st.d_bucket_total);
    if (st.d_bucket_base) (void)hipFree(st.d_bucket_base);
    if (st.d_bucket_cursor) (void)hipFree(st.d_bucket_cursor);
    st = PartitionState();
}

// visit every counted window of the staged tile owned by this lane:
//   f(id)                 for a clean window
//   g(F, i, nwin)         for a window whose only defects are N's (EXPAND mode)
template <bool EXPAND, typename FClean, typename FN>
__device__ __forceinline__ void for_each_window(const TileLds<EXPAND> &L, int k, int canonical, FClean f, FN g)
{
    const int j = threadIdx.x;
    const uint64_t idmask = (1ull << (2 * k)) - 1ull;
    const uint32_t kmask = (1u << k) - 1u;
    const uint32_t k1mask = kmask >> 1;
#pragma unroll 1
    for (int q = 0; q < CHUNKS_PER_THREAD; q++) {
        const int c = j + q * TPB;
        const Hood h = load_hood(L, c);
        uint32_t N32 = 0;
        if (EXPAND) N32 = (L.nn[c] & 0xFFFFu) | (L.nn[c + 1] << 16);
#pragma unroll
        for (int i = 0; i < 16; i++) {
            const bool crosses = (((h.S >> 1) >> i) & k1mask) != 0;
            const uint32_t vwin = (h.V >> i) & kmask;
            if (vwin == 0 && !crosses) {
                f(window_id<uint32_t>(h, i, k, canonical, idmask));
            } else if (EXPAND && !crosses) {
                const uint32_t nwin = (N32 >> i) & kmask;
                if (nwin == vwin) g(h.F, i, nwin);
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
    __shared__ unsigned long long s_tot[2];
    const int j = threadIdx.x;
    const uint32_t nbins = 1u << (2 * k);
    for (uint32_t i = j; i < nbins; i += TPB) hist[i] = 0;
    if (j < 2) s_tot[j] = 0;
    unsigned long long emitted = 0, nbad_tot = 0;
    const uint64_t idmask = (1ull << (2 * k)) - 1ull;
    for (uint32_t t = blockIdx.x; t < ntiles; t += gridDim.x) {
        uint32_t nbad;
        __syncthreads();                                   // previous tile fully consumed (and hist zeroed)
        stage_tile(L, bases, nbytes, t, &nbad);
        nbad_tot += nbad;
        __syncthreads();
        for_each_window(L, k, canonical,
            [&](uint32_t id) { atomicAdd(&hist[id], 1u); emitted++; },
            [&](uint64_t F, int i, uint32_t nwin) { expand_n_window(table, F, i, k, canonical, idmask, nwin, &emitted); });
    }
    __syncthreads();
    for (uint32_t i = j; i < nbins; i += TPB) {
        uint32_t c = hist[i];
        if (c) __hip_atomic_fetch_add(&table[i], (unsigned long long)c, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
    unsigned long long wt = wave_sum(emitted), wb = wave_sum(nbad_tot);
    if ((j & 63) == 0) { if (wt) atomicAdd(&s_tot[0], wt); if (wb) atomicAdd(&s_tot[1], wb); }
    __syncthreads();
    if (j == 0) {
        if (s_tot[0]) __hip_atomic_fetch_add(&ctr->total_kmers, s_tot[0], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if (s_tot[1]) __hip_atomic_fetch_add(&ctr->n_bad, s_tot[1], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
}

// ---------------------------------------------------------------------------------
// P0: exact bucket sizes (persistent workgroups; also counts bad residues)
// ---------------------------------------------------------------------------------
__global__ void __launch_bounds__(TPB)
bucket_count_kernel(const uint8_t *__restrict__ bases, uint64_t nbytes, uint32_t tile0, uint32_t ntiles, int k, int canonical,
                    unsigned long long *__restrict__ bucket_total, DevCounters *ctr)
{
    __shared__ TileLds<false> L;
    __shared__ uint32_t cnt[MAXB];
    __shared__ unsigned long long s_bad;
    const int j = threadIdx.x;
    for (int i = j; i < MAXB; i += TPB) cnt[i] = 0;
    if (j == 0) s_bad = 0;
    unsigned long long nbad_tot = 0;
    for (uint32_t t = blockIdx.x; t < ntiles; t += gridDim.x) {
        uint32_t nbad;
        __syncthreads();
        stage_tile(L, bases, nbytes, (uint64_t)tile0 + t, &nbad);
        nbad_tot += nbad;
        __syncthreads();
        for_each_window(L, k, canonical,
            [&](uint32_t id) { atomicAdd(&cnt[id >> BIN_BITS], 1u); },
            [&](uint64_t, int, uint32_t) {});
    }
    __syncthreads();
    for (int i = j; i < MAXB; i += TPB) {
        uint32_t c = cnt[i];
        if (c) __hip_atomic_fetch_add(&bucket_total[i], (unsigned long long)c, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
    unsigned long long wb = wave_sum(nbad_tot);
    if ((j & 63) == 0 && wb) atomicAdd(&s_bad, wb);
    __syncthreads();
    if (j == 0 && s_bad) __hip_atomic_fetch_add(&ctr->n_bad, s_bad, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

// P0b: exclusive scan of the MAXB totals (one workgroup of MAXB threads); adds Sum to total_kmers
__global__ void __launch_bounds__(MAXB)
bucket_scan_kernel(const unsigned long long *__restrict__ bucket_total, uint32_t *__restrict__ bucket_base,
                   uint32_t *__restrict__ bucket_cursor, DevCounters *ctr)
{
    __shared__ uint32_t wsum[MAXB / 64];
    const int j = threadIdx.x;
    const uint32_t v = (uint32_t)bucket_total[j];
    uint32_t s = v;
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) { uint32_t t = __shfl_up(s, o, 64); if ((j & 63) >= o) s += t; }
    if ((j & 63) == 63) wsum[j >> 6] = s;
    __syncthreads();
    uint32_t woff = 0;
    for (int w = 0; w < (j >> 6); w++) woff += wsum[w];
    const uint32_t excl = woff + s - v;
    bucket_base[j] = excl;
    bucket_cursor[j] = excl;
    if (j == MAXB - 1) {
        bucket_base[MAXB] = excl + v;
        if (excl + v) __hip_atomic_fetch_add(&ctr->total_kmers, (unsigned long long)(excl + v), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
}

// ---------------------------------------------------------------------------------
// P1: ids -> buckets.  One tile per workgroup: count per bucket in LDS, scan,
// reserve one contiguous run per (tile, bucket) with a returning global atomic,
// multisplit the 15-bit remainders into bucket order in LDS, copy runs out.
// ---------------------------------------------------------------------------------
template <bool EXPAND>
struct PartLds {
    TileLds<EXPAND> tile;
    uint16_t stage[TILE_CHUNKS * 16];
    uint32_t cnt[MAXB];        // per-bucket count, then the running local cursor
    uint32_t offs[MAXB];       // exclusive scan of cnt within the tile
    uint32_t gbase[MAXB];      // where this tile's run of bucket b starts in d_elems
    uint32_t wsum[TPB / 64];
};

template <bool EXPAND>
__global__ void __launch_bounds__(TPB)
partition_kernel(const uint8_t *__restrict__ bases, uint64_t nbytes, uint32_t tile0, int k, int canonical,
                 uint16_t *__restrict__ elems, uint32_t *__restrict__ bucket_cursor,
                 unsigned long long *__restrict__ table, DevCounters *ctr)
{
    static_assert(MAXB == 2 * TPB, "scan handles two buckets per thread");
    __shared__ PartLds<EXPAND> P;
    const int j = threadIdx.x;
    const int lane = j & 63, wave = j >> 6;
    P.cnt[2 * j] = 0; P.cnt[2 * j + 1] = 0;
    uint32_t nbad;
    stage_tile(P.tile, bases, nbytes, (uint64_t)tile0 + blockIdx.x, &nbad);   // bad residues were counted by P0
    __syncthreads();

    // (b) per-bucket counts; N windows (EXPAND) go straight to the vector
    const uint64_t idmask = (1ull << (2 * k)) - 1ull;
    unsigned long long expanded = 0;
    for_each_window(P.tile, k, canonical,
        [&](uint32_t id) { atomicAdd(&P.cnt[id >> BIN_BITS], 1u); },
        [&](uint64_t F, int i, uint32_t nwin) { expand_n_window(table, F, i, k, canonical, idmask, nwin, &expanded); });
    __syncthreads();

    // (c) exclusive scan over MAXB buckets (2 per thread) + global reservation
    {
        const uint32_t a = P.cnt[2 * j], b = P.cnt[2 * j + 1];
        uint32_t s = a + b;
#pragma unroll
        for (int o = 1; o < 64; o <<= 1) { uint32_t t = __shfl_up(s, o, 64); if (lane >= o) s += t; }
        if (lane == 63) P.wsum[wave] = s;
        __syncthreads();
        uint32_t woff = 0;
        for (int w = 0; w < wave; w++) woff += P.wsum[w];
        const uint32_t excl = woff + s - (a + b);
        P.offs[2 * j] = excl;
        P.offs[2 * j + 1] = excl + a;
        P.gbase[2 * j] = a ? __hip_atomic_fetch_add(&bucket_cursor[2 * j], a, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : 0u;
        P.gbase[2 * j + 1] = b ? __hip_atomic_fetch_add(&bucket_cursor[2 * j + 1], b, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : 0u;
        P.cnt[2 * j] = excl;              // becomes the local cursor
        P.cnt[2 * j + 1] = excl + a;
    }
    __syncthreads();

    // (d) multisplit into bucket order (N windows were handled in (b))
    for_each_window(P.tile, k, canonical,
        [&](uint32_t id) {
            const uint32_t slot = atomicAdd(&P.cnt[id >> BIN_BITS], 1u);
            P.stage[slot] = (uint16_t)(id & (BUCKET_BINS - 1));
        },
        [&](uint64_t, int, uint32_t) {});
    __syncthreads();

    // (e) copy the runs out: wave w takes buckets w, w+4, ...
    for (int b = wave; b < MAXB; b += TPB / 64) {
        const uint32_t o = P.offs[b];
        const uint32_t n = P.cnt[b] - o;
        const uint32_t g = P.gbase[b];
        for (uint32_t l = lane; l < n; l += 64) elems[(uint64_t)g + l] = P.stage[o + l];
    }

    if (EXPAND) {
        unsigned long long we = wave_sum(expanded);
        if (lane == 0 && we) __hip_atomic_fetch_add(&ctr->total_kmers, we, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
}

// ---------------------------------------------------------------------------------
// P2: one LDS histogram per (bucket, slice); flush with contiguous 64-bit atomics
// ---------------------------------------------------------------------------------
__global__ void __launch_bounds__(P2_THREADS)
bucket_hist_kernel(const uint16_t *__restrict__ elems, const uint32_t *__restrict__ bucket_base, int nslices,
                   unsigned long long *__restrict__ table)
{
    __shared__ uint32_t hist[BUCKET_BINS];
    const int tid = threadIdx.x;
    const int b = blockIdx.x / nslices, s = blockIdx.x % nslices;
    const uint64_t base = bucket_base[b], n = (uint64_t)bucket_base[b + 1] - base;
    const uint64_t g0 = base + n * (uint64_t)s / (uint64_t)nslices;
    const uint64_t g1 = base + n * (uint64_t)(s + 1) / (uint64_t)nslices;
    if (g1 == g0) return;
    for (int i = tid; i < BUCKET_BINS; i += P2_THREADS) hist[i] = 0;
    __syncthreads();
    uint64_t a0 = (g0 + 7ull) & ~7ull; if (a0 > g1) a0 = g1;
    uint64_t a1 = g1 & ~7ull; if (a1 < a0) a1 = a0;
    for (uint64_t g = g0 + tid; g < a0; g += P2_THREADS) atomicAdd(&hist[elems[g]], 1u);
    const uint4 *v4 = reinterpret_cast<const uint4 *>(elems);
    for (uint64_t v = a0 / 8 + tid; v < a1 / 8; v += P2_THREADS) {
        const uint4 x = v4[v];
        atomicAdd(&hist[x.x & 0xFFFFu], 1u); atomicAdd(&hist[x.x >> 16], 1u);
        atomicAdd(&hist[x.y & 0xFFFFu], 1u); atomicAdd(&hist[x.y >> 16], 1u);
        atomicAdd(&hist[x.z & 0xFFFFu], 1u); atomicAdd(&hist[x.z >> 16], 1u);
        atomicAdd(&hist[x.w & 0xFFFFu], 1u); atomicAdd(&hist[x.w >> 16], 1u);
    }
    for (uint64_t g = a1 + tid; g < g1; g += P2_THREADS) atomicAdd(&hist[elems[g]], 1u);
    __syncthreads();
    unsigned long long *dst = table + ((uint64_t)b << BIN_BITS);
    for (int i = tid; i < BUCKET_BINS; i += P2_THREADS) {
        const uint32_t c = hist[i];
        if (c) __hip_atomic_fetch_add(&dst[i], (unsigned long long)c, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
}

// ---------------------------------------------------------------------------------
// host: run the LDS-histogram path over one device-resident batch
// ---------------------------------------------------------------------------------
inline int partition_count(PartitionState &st, hipStream_t stream, const uint8_t *d_bases, size_t nbytes, int k, int canonical,
                           int n_expand, unsigned long long *d_table, DevCounters *d_ctr, ProfHook &prof)
{
#define KDB_P_TRY(expr)                                                             \
    do {                                                                            \
        hipError_t _e = (expr);                                                     \
        if (_e != hipSuccess) { partition_error_ref() = hipGetErrorString(_e); return 1; } \
    } while (0)
    const uint64_t ntiles_all = (nbytes + TILE_BYTES - 1) / TILE_BYTES;
    if (k <= SMALLK_MAX) {
        const uint32_t grid = (uint32_t)(ntiles_all < (uint64_t)PERSIST_GRID / 4 ? ntiles_all : (uint64_t)PERSIST_GRID / 4);
        prof.begin(KDB_KERNEL_COUNT);
        if (n_expand)
            hipLaunchKernelGGL(count_lds_kernel<true>, dim3(grid), dim3(TPB), 0, stream, d_bases, (uint64_t)nbytes,
                               (uint32_t)ntiles_all, k, canonical, d_table, d_ctr);
        else
            hipLaunchKernelGGL(count_lds_kernel<false>, dim3(grid), dim3(TPB), 0, stream, d_bases, (uint64_t)nbytes,
                               (uint32_t)ntiles_all, k, canonical, d_table, d_ctr);
        prof.end();
        KDB_P_TRY(hipGetLastError());
        return 0;
    }
    if (!st.d_bucket_total) {
        KDB_P_TRY(hipMalloc((void **)&st.d_bucket_total, MAXB * sizeof(unsigned long long)));
        KDB_P_TRY(hipMalloc((void **)&st.d_bucket_base, (MAXB + 1) * sizeof(uint32_t)));
        KDB_P_TRY(hipMalloc((void **)&st.d_bucket_cursor, MAXB * sizeof(uint32_t)));
    }
    // sub-batches keep element indices within 32 bits
    const uint64_t max_tiles = (1ull << 31) / TILE_BYTES;       // 2 Gi positions per sub-batch
    const size_t need = (size_t)((ntiles_all < max_tiles ? ntiles_all : max_tiles) * (uint64_t)TILE_BYTES);
    if (st.elems_cap < need) {
        if (st.d_elems) { KDB_P_TRY(hipStreamSynchronize(stream)); (void)hipFree(st.d_elems); st.d_elems = nullptr; st.elems_cap = 0; }
        KDB_P_TRY(hipMalloc((void **)&st.d_elems, need * sizeof(uint16_t) + 64));
        st.elems_cap = need;
    }
    const int nbuckets = 1 << (2 * k - BIN_BITS);
    for (uint64_t t0 = 0; t0 < ntiles_all; t0 += max_tiles) {
        const uint32_t nt = (uint32_t)((ntiles_all - t0) < max_tiles ? (ntiles_all - t0) : max_tiles);
        KDB_P_TRY(hipMemsetAsync(st.d_bucket_total, 0, MAXB * sizeof(unsigned long long), stream));
        prof.begin(KDB_KERNEL_BUCKET_COUNT);
        hipLaunchKernelGGL(bucket_count_kernel, dim3(nt < (uint32_t)PERSIST_GRID ? nt : (uint32_t)PERSIST_GRID), dim3(TPB), 0, stream,
                           d_bases, (uint64_t)nbytes, (uint32_t)t0, nt, k, canonical, st.d_bucket_total, d_ctr);
        prof.end();
        prof.begin(KDB_KERNEL_BUCKET_SCAN);
        hipLaunchKernelGGL(bucket_scan_kernel, dim3(1), dim3(MAXB), 0, stream, st.d_bucket_total, st.d_bucket_base,
                           st.d_bucket_cursor, d_ctr);
        prof.end();
        prof.begin(KDB_KERNEL_PARTITION);
        if (n_expand)
            hipLaunchKernelGGL(partition_kernel<true>, dim3(nt), dim3(TPB), 0, stream, d_bases, (uint64_t)nbytes, (uint32_t)t0, k,
                               canonical, st.d_elems, st.d_bucket_cursor, d_table, d_ctr);
        else
            hipLaunchKernelGGL(partition_kernel<false>, dim3(nt), dim3(TPB), 0, stream, d_bases, (uint64_t)nbytes, (uint32_t)t0, k,
                               canonical, st.d_elems, st.d_bucket_cursor, d_table, d_ctr);
        prof.end();
        // slices per bucket: aim at >= PERSIST_GRID workgroups, <= ~256 Ki elements each
        uint64_t per_bucket = ((uint64_t)nt * TILE_BYTES) / (uint64_t)nbuckets;
        int nslices = (int)((per_bucket + (256u << 10) - 1) / (256u << 10));
        if (nslices * nbuckets < PERSIST_GRID) nslices = (PERSIST_GRID + nbuckets - 1) / nbuckets;
        if (nslices < 1) nslices = 1;
        prof.begin(KDB_KERNEL_BUCKET_HIST);
        hipLaunchKernelGGL(bucket_hist_kernel, dim3((unsigned)(nbuckets * nslices)), dim3(P2_THREADS), 0, stream, st.d_elems,
                           st.d_bucket_base, nslices, d_table);
        prof.end();
        KDB_P_TRY(hipGetLastError());
    }
    return 0;
#undef KDB_P_TRY
}

}  // namespace kdb
