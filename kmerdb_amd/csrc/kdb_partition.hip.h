// kdb_partition.hip.h -- the LDS-histogram paths of the engine (gfx950).
//
// Direct 64-bit global atomics on uniformly random ids are bound by the memory-side atomic rate (device-scope
// RMWs execute at the memory side on gfx950: ~23 G random atomics/s measured, MI355X_MICROARCH.md "Global float
// atomics").  These paths replace the per-k-mer global atomic by per-k-mer LDS atomics:
//
//   k <= 7          count_lds_kernel: the whole 4^k vector lives in LDS (<= 64 KiB of u32), persistent
//                   workgroups, one global flush at the end.
//   8 <= k <= 12    radix partition on id bits 15..23 into B = 4^k / 32768 buckets, then one LDS histogram of
//                   32768 u32 bins per bucket.  Persistent workgroups; workgroup w owns tiles w, w+G, w+2G, ...
//                   in BOTH P0 and P1, which is what makes the scatter free of global atomics:
//                     P0  bucket_count_kernel   per-(tile, bucket) counts and per-(bucket, workgroup) totals
//                                               (ids are recomputed in P1, never stored)
//                     P0b wg_scan_kernel        per bucket: exclusive scan over the workgroups -> private slices
//                     P0c bucket_scan_kernel    bucket bases; P2 slice table (slices ~ bucket size); total k-mers
//                     P1  partition_kernel      chunk word pairs -> registers; ids computed and placed eight at a time
//                                               (slots from returning LDS atomics); bucket-ordered staging in LDS
//                                               (reusing the dead tile image); flat copy-out of the 15-bit
//                                               remainders into the workgroup's private slices
//                     P2  bucket_hist_kernel    LDS histogram per (bucket, slice); plain or atomic 64-bit flush
//   k = 13          the same three phases over 4 x 512 = 2048 buckets: bucket_count_allpass_kernel, partition_wide_kernel
//                   (run-by-run copy-out, no bucket stored per slot), bucket_hist_kernel.  (Option wide=0, and k = 14 by
//                   option: P1 + P2 once per 4^12-bin id range, "pass"; ids outside the pass are skipped.)
//   k = 14..17      kdb_twolevel.hip.h; its deferred histogram pass (pending_hist_kernel) lives here with P2.
//
// Same counting semantics as count_direct_kernel (kmer.py:234-317, :526-565; parse.py:133-136).  Windows containing
// N in EXPAND mode go to the vector through expand_n_window (in place or via the work list).  Degenerate stretches
// (poly-A/G, microsatellites) are detected per 16 windows and served by wave-aggregated atomics.
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
constexpr int PERSIST_GRID = 2048;                // upper bound on the persistent grid of P0/P1
constexpr int PART_GRID_DEFAULT = 2048;           // persistent workgroups of P0 (8 per CU) and P1 (2 per CU)
constexpr int SMALLK_MAX = 7;

struct ProfHook {
    virtual void begin(int kernel) = 0;
    virtual void end() = 0;
    virtual ~ProfHook() {}
};

struct PartitionState {
    uint16_t *d_elems = nullptr;          // bucketed 15-bit remainders
    size_t elems_cap = 0;                 // in elements
    uint32_t *d_bucket_total = nullptr;   // [MAXB]
    uint32_t *d_bucket_base = nullptr;    // [MAXB + 1]
    uint32_t *d_slice_base = nullptr;     // [MAXB + 1]: P2 workgroup index -> (bucket, slice)
    uint32_t *d_wide = nullptr;           // k = 13: totals [2048] | bases [2049] | slice bases [2049]
    uint32_t *d_wg_cnt = nullptr;         // [MAXB][G]: per-(bucket, workgroup) counts, then offsets
    uint16_t *d_tile_cnt = nullptr;       // [tiles][MAXB]: per-(tile, bucket) counts
    size_t tile_cnt_cap = 0;              // in tiles
    int slices = 0;                       // P2 workgroups per bucket (0 = auto)
    int grid = 0;                         // persistent workgroups of P0/P1 (0 = default)
    int wide = 1;                         // k = 13: one scatter pass over 2048 buckets (0: one P1 + P2 pass per id range)
    int reuse_image = 1;                  // k <= 12: P0 stores the encoded tiles, P1 reads them instead of encoding again
    uint32_t *d_img = nullptr;            // forward words [cap + 1] | masks [cap + 1]
    size_t img_cap = 0;                   // in chunks
};

inline const char *&partition_error_ref() { static thread_local const char *msg = ""; return msg; }
inline const char *partition_error() { return partition_error_ref(); }

constexpr int PASS_SHIFT = BIN_BITS + 9;           // ids are split as  pass | 9-bit bucket | 15-bit bin
constexpr int MAX_LDS_K = 14;                      // k = 13, 14: 4 / 16 passes over the input, one id range (4^12 bins) per pass
inline bool partition_supported(int k, int /*n_mode*/) { return k >= 1 && k <= 17; }   // 13..17: kdb_twolevel.hip.h (or multi-pass for 13, 14)
constexpr int WG_CNT_ROWS = 2048;                  // rows of d_wg_cnt: 4 x 512 buckets here (k = 13), up to 1024 L1 digits in kdb_twolevel.hip.h

inline void partition_free(PartitionState &st)
{
    if (st.d_elems) (void)hipFree(st.d_elems);
    if (st.d_bucket_total) (void)hipFree(st.d_bucket_total);
    if (st.d_bucket_base) (void)hipFree(st.d_bucket_base);
    if (st.d_slice_base) (void)hipFree(st.d_slice_base);
    if (st.d_img) (void)hipFree(st.d_img);
    if (st.d_wide) (void)hipFree(st.d_wide);
    if (st.d_wg_cnt) (void)hipFree(st.d_wg_cnt);
    if (st.d_tile_cnt) (void)hipFree(st.d_tile_cnt);
    st = PartitionState();
}

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
    __shared__ unsigned long long s_tot[2];
    const int j = threadIdx.x;
    const uint32_t nbins = 1u << (2 * k);
    for (uint32_t i = j; i < nbins; i += TPB) hist[i] = 0;
    if (j < 2) s_tot[j] = 0;
    unsigned long long emitted = 0, nbad_tot = 0;
    const uint64_t idmask = (1ull << (2 * k)) - 1ull;
    const UniformStarts ulen(batch_uniform_len(ctr), TPB);
    for (uint32_t t = blockIdx.x; t < ntiles; t += gridDim.x) {
        uint32_t nbad;
        __syncthreads();                                   // previous tile fully consumed (and hist zeroed)
        stage_tile(L, bases, nbytes, t, &nbad, ulen);
        nbad_tot += nbad;
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
    unsigned long long wt = wave_sum(emitted), wb = wave_sum(nbad_tot);
    if ((j & 63) == 0) { if (wt) atomicAdd(&s_tot[0], wt); if (wb) atomicAdd(&s_tot[1], wb); }
    __syncthreads();
    if (j == 0) {
        if (s_tot[0]) __hip_atomic_fetch_add(&ctr->total_kmers, s_tot[0], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if (s_tot[1]) __hip_atomic_fetch_add(&ctr->n_bad, s_tot[1], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
}

// ---------------------------------------------------------------------------------
// P0: exact sizes.  Persistent workgroups; workgroup w owns tiles w, w+G, w+2G, ... (the SAME
// ownership partition_kernel uses).  Writes the per-(tile, bucket) counts (u16, 1 KiB per tile)
// and the per-(bucket, workgroup) totals, so that the scans below can hand every workgroup a
// private, exactly sized slice of every bucket: the scatter path has no global atomics at all
// and the bucket contents come out in a deterministic order.
// ---------------------------------------------------------------------------------
// The sizing pass leaves the encoded tiles in HBM (forward 2-bit words and masks, 8 bytes per 16 positions), so that the
// scatter pass does not encode the residues a second time.  The reverse-strand word is the forward word with its
// 2-bit groups in reverse order, complemented: v_bfrev_b32, swap the bits of every pair, not.
template <int THREADS>
__device__ __forceinline__ void image_store(const TileLds<false> &L, uint32_t *__restrict__ img_fwd, uint32_t *__restrict__ img_msk,
                                            uint32_t t, uint32_t ntiles)
{
    const int j = threadIdx.x;
#pragma unroll
    for (int q = 0; q < TILE_CHUNKS / THREADS; q++) {
        const int c = j + q * THREADS;
        img_fwd[(size_t)t * TILE_CHUNKS + c] = L.fwd[c];
        img_msk[(size_t)t * TILE_CHUNKS + c] = L.msk[c];
    }
    if (t + 1 == ntiles && j == 0) { img_fwd[(size_t)ntiles * TILE_CHUNKS] = L.fwd[TILE_CHUNKS]; img_msk[(size_t)ntiles * TILE_CHUNKS] = L.msk[TILE_CHUNKS]; }
}

template <int THREADS>
__device__ __forceinline__ void image_load(TileLds<false> &L, const uint32_t *__restrict__ img_fwd, const uint32_t *__restrict__ img_msk, uint32_t t)
{
    const int j = threadIdx.x;
    const uint32_t *ff = img_fwd + (size_t)t * TILE_CHUNKS, *mm = img_msk + (size_t)t * TILE_CHUNKS;
#pragma unroll
    for (int q = 0; q <= TILE_CHUNKS / THREADS; q++) {
        const int cc = q < TILE_CHUNKS / THREADS ? j + q * THREADS : TILE_CHUNKS;            // (+ the halo chunk, thread 0)
        if (q < TILE_CHUNKS / THREADS || j == 0) {
            const uint32_t f = ff[cc], y = __builtin_bitreverse32(f);
            L.fwd[cc] = f;
            L.rc[cc] = ~(((y >> 1) & 0x55555555u) | ((y & 0x55555555u) << 1));
            L.msk[cc] = mm[cc];
        }
    }
}

template <bool CANON, bool MULTIPASS>
__global__ void __launch_bounds__(TPB)
bucket_count_kernel(const uint8_t *__restrict__ bases, uint64_t nbytes, uint32_t tile0, uint32_t ntiles, int k,
                    uint32_t pass /* MULTIPASS: only ids with (id >> PASS_SHIFT) == pass are counted */,
                    uint32_t *__restrict__ tile_cnt /* [ntiles][MAXB/2]: two u16 counts per word */,
                    uint32_t *__restrict__ wg_cnt /* [MAXB][gridDim.x] */, DevCounters *ctr,
                    uint32_t *__restrict__ img_fwd = nullptr /* [ntiles * TILE_CHUNKS + 1]: the encoded tiles, for P1 */,
                    uint32_t *__restrict__ img_msk = nullptr)
{
    static_assert(MAXB == 2 * TPB, "two buckets per thread");
    __shared__ TileLds<false> L;
    __shared__ uint32_t cnt[MAXB + 32];                 // + 32 dump slots: windows that are not counted add there (no branch)
    __shared__ unsigned long long s_bad;
    const int j = threadIdx.x;
    cnt[2 * j] = 0; cnt[2 * j + 1] = 0;
    if (j == 0) s_bad = 0;
    unsigned long long nbad_tot = 0;
    uint32_t tot0 = 0, tot1 = 0;
    const UniformStarts ulen(batch_uniform_len(ctr), TPB);
    const IdParams<uint32_t> idp(k, CANON ? 1 : 0);
    const uint32_t dump = MAXB + (j & 31);
    for (uint32_t t = blockIdx.x; t < ntiles; t += gridDim.x) {
        uint32_t nbad;
        stage_tile(L, bases, nbytes, (uint64_t)tile0 + t, &nbad, ulen);
        nbad_tot += nbad;
        __syncthreads();
        if (img_fwd) image_store<TPB>(L, img_fwd, img_msk, t, ntiles);
#pragma unroll 1
        for (int q = 0; q < CHUNKS_PER_THREAD; q++) {
            const Hood h = load_hood(L, j + q * TPB);
            const uint32_t bad16 = windows_bad16(h, k);
            uint64_t same; uint32_t id0;
            if (!wave_dominant(idp.id(h, 0), &same, &id0)) {
                // the hot loop: straight-line, ~8 VALU + one LDS atomic per window
#pragma unroll
                for (int i = 0; i < 16; i++) {
                    const uint32_t id = idp.id(h, i);
                    uint32_t skip = bad_fill(bad16, i);
                    if (MULTIPASS) skip |= ((id >> PASS_SHIFT) == pass) ? 0u : ~0u;
                    // byte offset of the bucket's counter (or of this lane's dump slot): shift, and, bfi
                    const uint32_t off = bfi(skip, dump * 4u, (id >> (BIN_BITS - 2)) & ((MAXB - 1) * 4u));
                    atomicAdd(reinterpret_cast<uint32_t *>(reinterpret_cast<char *>(cnt) + off), 1u);
                }
            } else {
                // degenerate stretch (poly-A/G, microsatellites): same-key lanes are served by one atomic
#pragma unroll 1
                for (int i = 0; i < 16; i++) {
                    const uint32_t id = idp.id_dyn(h, i);
                    bool take = !((bad16 >> i) & 1u);
                    if (MULTIPASS) take = take && (id >> PASS_SHIFT) == pass;
                    if (take) lds_hist_add(cnt, (id >> BIN_BITS) & (MAXB - 1));
                }
            }
        }
        __syncthreads();
        const uint32_t c0 = cnt[2 * j], c1 = cnt[2 * j + 1];      // <= 16384 each
        cnt[2 * j] = 0; cnt[2 * j + 1] = 0;                       // own entries: next tile's atomics come after the next barrier
        tile_cnt[(size_t)t * (MAXB / 2) + j] = c0 | (c1 << 16);
        tot0 += c0; tot1 += c1;
    }
    wg_cnt[(size_t)(2 * j) * gridDim.x + blockIdx.x] = tot0;
    wg_cnt[(size_t)(2 * j + 1) * gridDim.x + blockIdx.x] = tot1;
    unsigned long long wb = wave_sum(nbad_tot);
    if ((j & 63) == 0 && wb) atomicAdd(&s_bad, wb);
    __syncthreads();
    if (j == 0 && s_bad && pass == 0) __hip_atomic_fetch_add(&ctr->n_bad, s_bad, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

// P0 for all four id ranges of k = 13 in ONE pass over the input: 4 x 512 counters, one count matrix and one
// per-(bucket, workgroup) matrix per range (`range_tiles` / `range_wg` entries apart).  The four P1 / P2 passes
// that follow still re-scan the input, but the sizing pass is no longer repeated with them.
constexpr int ALLPASS = 4;
template <bool CANON>
__global__ void __launch_bounds__(TPB)
bucket_count_allpass_kernel(const uint8_t *__restrict__ bases, uint64_t nbytes, uint32_t tile0, uint32_t ntiles, int k,
                            uint32_t *__restrict__ tile_cnt /* [ALLPASS][ntiles][MAXB/2] */, size_t range_tiles /* words */,
                            uint32_t *__restrict__ wg_cnt /* [ALLPASS][MAXB][gridDim.x] */, size_t range_wg, DevCounters *ctr,
                            uint32_t *__restrict__ img_fwd = nullptr, uint32_t *__restrict__ img_msk = nullptr)
{
    static_assert(MAXB == 2 * TPB, "two buckets per thread and range");
    constexpr int NC = ALLPASS * MAXB;
    __shared__ TileLds<false> L;
    __shared__ uint32_t cnt[NC + 32];                   // + 32 dump slots
    __shared__ unsigned long long s_bad;
    const int j = threadIdx.x;
    for (int c = j; c < NC; c += TPB) cnt[c] = 0;
    if (j == 0) s_bad = 0;
    unsigned long long nbad_tot = 0;
    uint32_t tot[ALLPASS][2] = {};
    const UniformStarts ulen(batch_uniform_len(ctr), TPB);
    const IdParams<uint32_t> idp(k, CANON ? 1 : 0);
    const uint32_t dump = NC + (j & 31);
    for (uint32_t t = blockIdx.x; t < ntiles; t += gridDim.x) {
        uint32_t nbad;
        stage_tile(L, bases, nbytes, (uint64_t)tile0 + t, &nbad, ulen);
        nbad_tot += nbad;
        __syncthreads();
        if (img_fwd) image_store<TPB>(L, img_fwd, img_msk, t, ntiles);
#pragma unroll 1
        for (int q = 0; q < CHUNKS_PER_THREAD; q++) {
            const Hood h = load_hood(L, j + q * TPB);
            const uint32_t bad16 = windows_bad16(h, k);
            uint64_t same; uint32_t id0;
            if (!wave_dominant(idp.id(h, 0), &same, &id0)) {
#pragma unroll
                for (int i = 0; i < 16; i++) {
                    const uint32_t id = idp.id(h, i);
                    const uint32_t off = bfi(bad_fill(bad16, i), dump * 4u, (id >> (BIN_BITS - 2)) & ((NC - 1) * 4u));    // range | bucket
                    atomicAdd(reinterpret_cast<uint32_t *>(reinterpret_cast<char *>(cnt) + off), 1u);
                }
            } else {
#pragma unroll 1
                for (int i = 0; i < 16; i++) {
                    const uint32_t id = idp.id_dyn(h, i);
                    if (!((bad16 >> i) & 1u)) lds_hist_add(cnt, (id >> BIN_BITS) & (NC - 1));
                }
            }
        }
        __syncthreads();
#pragma unroll
        for (int p = 0; p < ALLPASS; p++) {
            const uint32_t c0 = cnt[p * MAXB + 2 * j], c1 = cnt[p * MAXB + 2 * j + 1];
            cnt[p * MAXB + 2 * j] = 0; cnt[p * MAXB + 2 * j + 1] = 0;
            tile_cnt[p * range_tiles + (size_t)t * (MAXB / 2) + j] = c0 | (c1 << 16);
            tot[p][0] += c0; tot[p][1] += c1;
        }
    }
#pragma unroll
    for (int p = 0; p < ALLPASS; p++) {
        wg_cnt[p * range_wg + (size_t)(2 * j) * gridDim.x + blockIdx.x] = tot[p][0];
        wg_cnt[p * range_wg + (size_t)(2 * j + 1) * gridDim.x + blockIdx.x] = tot[p][1];
    }
    unsigned long long wb = wave_sum(nbad_tot);
    if ((j & 63) == 0 && wb) atomicAdd(&s_bad, wb);
    __syncthreads();
    if (j == 0 && s_bad) __hip_atomic_fetch_add(&ctr->n_bad, s_bad, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
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

// P0b: one workgroup per bucket: exclusive scan of that bucket's per-workgroup counts (in place), bucket total out
constexpr int SCAN_PER_THREAD = PERSIST_GRID / TPB;
__global__ void __launch_bounds__(TPB)
wg_scan_kernel(uint32_t *__restrict__ wg_cnt /* [MAXB][G] in: counts, out: offsets within the bucket */, uint32_t G,
               uint32_t *__restrict__ bucket_total)
{
    __shared__ uint32_t wsum[TPB / 64];
    const int j = threadIdx.x;
    uint32_t *row = wg_cnt + (size_t)blockIdx.x * G;
    uint32_t v[SCAN_PER_THREAD], sum = 0;
#pragma unroll
    for (int i = 0; i < SCAN_PER_THREAD; i++) {
        const uint32_t w = (uint32_t)j * SCAN_PER_THREAD + i;
        v[i] = (w < G) ? row[w] : 0u;
        sum += v[i];
    }
    uint32_t tot;
    uint32_t run = block_excl_scan<TPB>(sum, wsum, &tot);
#pragma unroll
    for (int i = 0; i < SCAN_PER_THREAD; i++) {
        const uint32_t w = (uint32_t)j * SCAN_PER_THREAD + i;
        if (w < G) row[w] = run;
        run += v[i];
    }
    if (j == 0) bucket_total[blockIdx.x] = tot;
}

// P0c: exclusive scan of the MAXB bucket totals (one workgroup of MAXB threads); adds Sum to total_kmers.
// Also cuts every bucket into ceil(n_b / slice_elems) slices for P2 (so that big buckets -- canonical ids are
// far from uniform over the id space, and real data is skewed -- get proportionally more workgroups).
__global__ void __launch_bounds__(MAXB)
bucket_scan_kernel(const uint32_t *__restrict__ bucket_total, uint32_t *__restrict__ bucket_base,
                   uint32_t *__restrict__ slice_base /* [MAXB + 1] */, uint32_t slice_elems, int add_total, DevCounters *ctr)
{
    __shared__ uint32_t wsum[MAXB / 64];
    uint32_t tot;
    const uint32_t v = bucket_total[threadIdx.x];
    const uint32_t excl = block_excl_scan<MAXB>(v, wsum, &tot);
    bucket_base[threadIdx.x] = excl;
    const uint32_t nsl = v ? (v + slice_elems - 1) / slice_elems : 0u;
    uint32_t stot;
    const uint32_t sexcl = block_excl_scan<MAXB>(nsl, wsum, &stot);
    slice_base[threadIdx.x] = sexcl;
    if (threadIdx.x == MAXB - 1) {
        bucket_base[MAXB] = tot;
        slice_base[MAXB] = stot;
        if (tot && add_total) __hip_atomic_fetch_add(&ctr->total_kmers, (unsigned long long)tot, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
}

// ---------------------------------------------------------------------------------
// P1: ids -> buckets.  Persistent 512-thread workgroups (same tile ownership as P0).  Per tile:
//   (a) stage + encode into LDS; load the tile's 512 bucket counts (from P0), exclusive scan:
//       the run of bucket b starts at slot offs[b] in LDS and at cur[b] in d_elems, where cur[]
//       is the workgroup's private running cursor (LDS; no global atomics)
//   (b) one pass over the windows: slot = returning LDS atomic on the bucket's cursor; the 15-bit
//       remainder and the 9-bit bucket go to LDS in bucket order (u16 + u8 per slot)
//   (c) flat copy-out: consecutive lanes -> consecutive slots -> runs of consecutive addresses
// ---------------------------------------------------------------------------------
constexpr int P1_THREADS = 512;
constexpr int TILE_POS = TILE_CHUNKS * 16;

// The 2-bit image of the tile is dead once every lane has pulled its ids into registers, so the bucket-ordered
// staging array lives in the same LDS bytes: one 32-bit word per id (the id itself: bucket in bits 15..23,
// bin in bits 0..14) = one LDS write per id and one LDS read per id in the copy-out.
template <bool EXPAND>
struct PartLds {
    union {
        TileLds<EXPAND> tile;
        struct { uint32_t stage[TILE_POS]; } o;     // the id itself (bucket in bits 15..23, bin in bits 0..14): one write, one read per id
    } u;
    uint32_t lcur[MAXB];         // local cursor: next free slot of bucket b in `stage`
    uint32_t delta[MAXB];        // (position in d_elems of the run of bucket b) - (its first slot)
    uint32_t wsum[P1_THREADS / 64];
    uint32_t nids;
};

template <bool EXPAND, bool CANON, bool MULTIPASS>
__global__ void __launch_bounds__(P1_THREADS, 4)
partition_kernel(const uint8_t *__restrict__ bases, uint64_t nbytes, uint32_t tile0, uint32_t ntiles, int k,
                 uint32_t pass, uint16_t *__restrict__ elems, const uint32_t *__restrict__ bucket_base,
                 const uint32_t *__restrict__ wg_off /* [MAXB][gridDim.x] */,
                 const uint16_t *__restrict__ tile_cnt /* [ntiles][MAXB] */,
                 unsigned long long *__restrict__ table, DevCounters *ctr,
                 const uint32_t *__restrict__ img_fwd = nullptr /* encoded tiles written by P0 (not with EXPAND) */,
                 const uint32_t *__restrict__ img_msk = nullptr)
{
    static_assert(MAXB == P1_THREADS, "one bucket per thread");
    constexpr int CPT = TILE_CHUNKS / P1_THREADS;            // chunks per thread
    constexpr uint32_t NO_ID = 0xFFFFFFFFu;
    __shared__ PartLds<EXPAND> P;
    const int j = threadIdx.x;
    uint32_t cur = bucket_base[j] + wg_off[(size_t)j * gridDim.x + blockIdx.x];      // thread b owns bucket b's running cursor
    const uint64_t idmask = (1ull << (2 * k)) - 1ull;
    const int canonical = CANON ? 1 : 0;
    const IdParams<uint32_t> idp(k, canonical);
    const uint32_t kmask = (1u << k) - 1u, k1mask = kmask >> 1;
    const UniformStarts ulen(batch_uniform_len(ctr), P1_THREADS);
    unsigned long long expanded = 0;

    for (uint32_t t = blockIdx.x; t < ntiles; t += gridDim.x) {
        // (a) 2-bit image of the tile; this tile's 512 bucket counts -> slots
        const uint32_t c = tile_cnt[(size_t)t * MAXB + j];
        uint32_t nbad;
        if constexpr (!EXPAND) {
            if (img_fwd) image_load<P1_THREADS>(P.u.tile, img_fwd, img_msk, t);
            else stage_tile<EXPAND, P1_THREADS, false>(P.u.tile, bases, nbytes, (uint64_t)tile0 + t, &nbad, ulen);
        } else {
            stage_tile<EXPAND, P1_THREADS, false>(P.u.tile, bases, nbytes, (uint64_t)tile0 + t, &nbad, ulen);   // bad residues were counted by P0
        }
        uint32_t tot;
        const uint32_t excl = block_excl_scan<P1_THREADS>(c, P.wsum, &tot);                   // (two barriers inside)
        P.lcur[j] = excl;
        P.delta[j] = cur - excl;
        cur += c;
        if (j == 0) P.nids = tot;

        // (b) every lane pulls the two word pairs ("hoods") of its chunks into registers: 12 registers instead of 32 ids
        Hood hs[CPT];
        uint32_t bad[CPT];
        bool degenerate = false;
#pragma unroll
        for (int q = 0; q < CPT; q++) {
            const int cc = j + q * P1_THREADS;
            hs[q] = load_hood(P.u.tile, cc);
            bad[q] = windows_bad16(hs[q], k);
            uint64_t same; uint32_t id0;
            degenerate |= wave_dominant(idp.id(hs[q], 0), &same, &id0);
            if (EXPAND && pass == 0 && bad[q]) {
                const uint32_t N32 = (P.u.tile.nn[cc] & 0xFFFFu) | (P.u.tile.nn[cc + 1] << 16);
#pragma unroll 1
                for (int i = 0; i < 16; i++) {
                    if (((bad[q] >> i) & 1u) && !window_crosses(hs[q], i, k1mask)) {
                        const uint32_t vwin = (hs[q].V >> i) & kmask, nwin = (N32 >> i) & kmask;
                        if (nwin == vwin) expand_n_window(table, hs[q].F(), i, k, canonical, idmask, nwin, &expanded, ctr);
                    }
                }
            }
        }
        __syncthreads();          // the tile image is dead from here on: `stage` reuses its bytes

        // (c) ids are computed eight at a time and placed at once: slot = returning LDS atomic on the bucket's cursor.
        //     VALU work (ids) and LDS work (atomics, staging writes) alternate instead of coming in two bursts.
        if (!degenerate) {
#pragma unroll
            for (int q = 0; q < CPT; q++) {
#pragma unroll
                for (int g = 0; g < 16; g += 8) {
                    uint32_t id8[8], slot[8];
#pragma unroll
                    for (int u = 0; u < 8; u++) {
                        uint32_t id = idp.id(hs[q], g + u);
                        uint32_t skip = bad_fill(bad[q], g + u);
                        if (MULTIPASS) { skip |= ((id >> PASS_SHIFT) == pass) ? 0u : ~0u; id &= (1u << PASS_SHIFT) - 1u; }
                        id8[u] = id | skip;                                    // NO_ID where the window is not counted in this pass
                    }
#pragma unroll
                    for (int u = 0; u < 8; u++) slot[u] = (id8[u] != NO_ID) ? atomicAdd(&P.lcur[id8[u] >> BIN_BITS], 1u) : 0u;
#pragma unroll
                    for (int u = 0; u < 8; u++)
                        if (id8[u] != NO_ID) P.u.o.stage[slot[u]] = id8[u];
                }
            }
        } else {
#pragma unroll 1
            for (int q = 0; q < CPT * 16; q++) {
                const Hood &h = (CPT > 1 && q >= 16) ? hs[CPT - 1] : hs[0];
                const uint32_t b16 = (CPT > 1 && q >= 16) ? bad[CPT - 1] : bad[0];
                const int i = q & 15;
                uint32_t id = idp.id_dyn(h, i);
                bool take = !((b16 >> i) & 1u);
                if (MULTIPASS) { take = take && (id >> PASS_SHIFT) == pass; id &= (1u << PASS_SHIFT) - 1u; }
                if (take) {
                    const uint32_t slot = lds_cursor_take(P.lcur, id >> BIN_BITS);
                    P.u.o.stage[slot] = id;
                }
            }
        }
        __syncthreads();

        // (d) flat copy-out: consecutive lanes -> consecutive slots -> runs of consecutive addresses
        const uint32_t nids = P.nids;
        for (uint32_t sl0 = j; sl0 < nids; sl0 += 8 * P1_THREADS) {
            uint32_t v[8], d[8];
#pragma unroll
            for (int u = 0; u < 8; u++) {
                const uint32_t sl = sl0 + u * P1_THREADS;
                v[u] = (sl < nids) ? P.u.o.stage[sl] : 0u;
            }
#pragma unroll
            for (int u = 0; u < 8; u++) d[u] = P.delta[(v[u] >> 15) & (MAXB - 1)];
#pragma unroll
            for (int u = 0; u < 8; u++)
                if (sl0 + u * P1_THREADS < nids) elems[(uint64_t)d[u] + sl0 + u * P1_THREADS] = (uint16_t)(v[u] & (BUCKET_BINS - 1));
        }
        __syncthreads();          // stage / lcur / delta are rewritten by the next tile
    }

    if (EXPAND) {
        unsigned long long we = wave_sum(expanded);
        if ((j & 63) == 0 && we) __hip_atomic_fetch_add(&ctr->total_kmers, we, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
}

// ---------------------------------------------------------------------------------
// k = 13 in ONE scatter pass: WIDE_B = 4 x 512 = 2048 buckets (bucket = id >> 15, 11 bits: id range | bucket).
// The staging array holds the 15-bit bins only; the copy-out goes run by run (eight lanes per run), so no bucket has
// to be remembered per slot and the 2048 cursors + deltas fit beside the 32 KiB staging array (3 workgroups per CU).
// Thread j owns the four consecutive buckets 4j .. 4j+3 (their counts are one 8-byte load from range j / 128's matrix).
// ---------------------------------------------------------------------------------
constexpr int WIDE_B = ALLPASS * MAXB;
constexpr int WIDE_OWN = WIDE_B / P1_THREADS;        // buckets per thread
constexpr uint32_t WIDE_LONG_RUN = 24;               // elements of a run copied by its eight lanes; tails by the whole workgroup

// bucket totals [WIDE_B] -> bases [WIDE_B + 1], P2 slice table [WIDE_B + 1], Sum -> total_kmers
__global__ void __launch_bounds__(P1_THREADS)
wide_scan_kernel(const uint32_t *__restrict__ bucket_total, uint32_t *__restrict__ bucket_base, uint32_t *__restrict__ slice_base,
                 uint32_t slice_elems, DevCounters *ctr)
{
    __shared__ uint32_t wsum[P1_THREADS / 64];
    const uint32_t j = threadIdx.x;
    uint32_t v[WIDE_OWN], nsl[WIDE_OWN], sum = 0, ssum = 0;
#pragma unroll
    for (int u = 0; u < WIDE_OWN; u++) {
        v[u] = bucket_total[j * WIDE_OWN + u];
        nsl[u] = v[u] ? (v[u] + slice_elems - 1) / slice_elems : 0u;
        sum += v[u]; ssum += nsl[u];
    }
    uint32_t tot, stot;
    uint32_t run = block_excl_scan<P1_THREADS>(sum, wsum, &tot);
    uint32_t srun = block_excl_scan<P1_THREADS>(ssum, wsum, &stot);
#pragma unroll
    for (int u = 0; u < WIDE_OWN; u++) {
        bucket_base[j * WIDE_OWN + u] = run; slice_base[j * WIDE_OWN + u] = srun;
        run += v[u]; srun += nsl[u];
    }
    if (j == P1_THREADS - 1) {
        bucket_base[WIDE_B] = tot; slice_base[WIDE_B] = stot;
        if (tot) __hip_atomic_fetch_add(&ctr->total_kmers, (unsigned long long)tot, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
}

template <bool EXPAND>
struct WideLds {
    union {
        TileLds<EXPAND> tile;
        uint16_t stage[TILE_POS];            // the tile's 15-bit bins in bucket order (reuses the dead image's bytes)
    } u;
    uint32_t lcur[WIDE_B];                   // cursor of each bucket's run; after the placement: the end of the run
    uint32_t delta[WIDE_B];                  // (position of the run in d_elems) - (its first slot)
    uint32_t wsum[P1_THREADS / 64];
    uint32_t nlong;
    uint16_t longb[TILE_POS / WIDE_LONG_RUN + 1];
};

template <bool EXPAND, bool CANON>
__global__ void __launch_bounds__(P1_THREADS, 6)
partition_wide_kernel(const uint8_t *__restrict__ bases, uint64_t nbytes, uint32_t tile0, uint32_t ntiles, int k,
                      uint16_t *__restrict__ elems, const uint32_t *__restrict__ bucket_base /* [WIDE_B + 1] */,
                      const uint32_t *__restrict__ wg_off /* [WIDE_B][gridDim.x] */,
                      const uint16_t *__restrict__ tile_cnt /* [ALLPASS][ntiles][MAXB] */, size_t range_elems /* u16 per matrix */,
                      unsigned long long *__restrict__ table, DevCounters *ctr,
                      const uint32_t *__restrict__ img_fwd = nullptr, const uint32_t *__restrict__ img_msk = nullptr)
{
    constexpr int CPT = TILE_CHUNKS / P1_THREADS;
    constexpr uint32_t NO_ID = 0xFFFFFFFFu;
    __shared__ WideLds<EXPAND> P;
    const int j = threadIdx.x;
    const uint32_t o0 = (uint32_t)j * WIDE_OWN;                  // first owned bucket (in id order)
    uint32_t cur[WIDE_OWN];
#pragma unroll
    for (int u = 0; u < WIDE_OWN; u++) cur[u] = bucket_base[o0 + u] + wg_off[(size_t)(o0 + u) * gridDim.x + blockIdx.x];
    const uint16_t *my_cnt = tile_cnt + (size_t)(o0 / MAXB) * range_elems + (o0 % MAXB);
    const uint64_t idmask = (1ull << (2 * k)) - 1ull;
    const int canonical = CANON ? 1 : 0;
    const IdParams<uint32_t> idp(k, canonical);
    const uint32_t kmask = (1u << k) - 1u, k1mask = kmask >> 1;
    const UniformStarts ulen(batch_uniform_len(ctr), P1_THREADS);
    unsigned long long expanded = 0;

    for (uint32_t t = blockIdx.x; t < ntiles; t += gridDim.x) {
        // (a) the tile's image; its 2048 bucket counts -> slots
        const uint2 cw = *reinterpret_cast<const uint2 *>(my_cnt + (size_t)t * MAXB);      // four u16 counts
        const uint32_t c[WIDE_OWN] = {cw.x & 0xFFFFu, cw.x >> 16, cw.y & 0xFFFFu, cw.y >> 16};
        if (j == 0) P.nlong = 0;
        uint32_t nbad;
        if constexpr (!EXPAND) {
            if (img_fwd) image_load<P1_THREADS>(P.u.tile, img_fwd, img_msk, t);
            else stage_tile<EXPAND, P1_THREADS, false>(P.u.tile, bases, nbytes, (uint64_t)tile0 + t, &nbad, ulen);
        } else {
            stage_tile<EXPAND, P1_THREADS, false>(P.u.tile, bases, nbytes, (uint64_t)tile0 + t, &nbad, ulen);   // bad residues were counted by P0
        }
        uint32_t tot;
        uint32_t run = block_excl_scan<P1_THREADS>(c[0] + c[1] + c[2] + c[3], P.wsum, &tot);    // (two barriers inside)
#pragma unroll
        for (int u = 0; u < WIDE_OWN; u++) {
            P.lcur[o0 + u] = run;
            P.delta[o0 + u] = cur[u] - run;
            cur[u] += c[u];
            run += c[u];
            if (c[u] > WIDE_LONG_RUN) P.longb[atomicAdd(&P.nlong, 1u)] = (uint16_t)(o0 + u);
        }

        // (b) word pairs of the two chunks into registers; windows with N go to the vector at once (EXPAND)
        Hood hs[CPT];
        uint32_t bad[CPT];
        bool degenerate = false;
#pragma unroll
        for (int q = 0; q < CPT; q++) {
            const int cc = j + q * P1_THREADS;
            hs[q] = load_hood(P.u.tile, cc);
            bad[q] = windows_bad16(hs[q], k);
            uint64_t same; uint32_t id0;
            degenerate |= wave_dominant(idp.id(hs[q], 0), &same, &id0);
            if (EXPAND && bad[q]) {
                const uint32_t N32 = (P.u.tile.nn[cc] & 0xFFFFu) | (P.u.tile.nn[cc + 1] << 16);
#pragma unroll 1
                for (int i = 0; i < 16; i++) {
                    if (((bad[q] >> i) & 1u) && !window_crosses(hs[q], i, k1mask)) {
                        const uint32_t vwin = (hs[q].V >> i) & kmask, nwin = (N32 >> i) & kmask;
                        if (nwin == vwin) expand_n_window(table, hs[q].F(), i, k, canonical, idmask, nwin, &expanded, ctr);
                    }
                }
            }
        }
        __syncthreads();          // the tile image is dead from here on: `stage` reuses its bytes

        // (c) ids eight at a time: slot = returning LDS atomic on the bucket's cursor; the bin goes to its slot
        if (!degenerate) {
#pragma unroll
            for (int q = 0; q < CPT; q++) {
#pragma unroll
                for (int g = 0; g < 16; g += 8) {
                    uint32_t id8[8], slot[8];
#pragma unroll
                    for (int u = 0; u < 8; u++) id8[u] = idp.id(hs[q], g + u) | bad_fill(bad[q], g + u);     // NO_ID where not counted
#pragma unroll
                    for (int u = 0; u < 8; u++) slot[u] = (id8[u] != NO_ID) ? atomicAdd(&P.lcur[id8[u] >> BIN_BITS], 1u) : 0u;
#pragma unroll
                    for (int u = 0; u < 8; u++)
                        if (id8[u] != NO_ID) P.u.stage[slot[u]] = (uint16_t)(id8[u] & (BUCKET_BINS - 1));
                }
            }
        } else {
#pragma unroll 1
            for (int q = 0; q < CPT * 16; q++) {
                const Hood &h = (CPT > 1 && q >= 16) ? hs[CPT - 1] : hs[0];
                const uint32_t b16 = (CPT > 1 && q >= 16) ? bad[CPT - 1] : bad[0];
                const int i = q & 15;
                const uint32_t id = idp.id_dyn(h, i);
                if (!((b16 >> i) & 1u)) P.u.stage[lds_cursor_take(P.lcur, id >> BIN_BITS)] = (uint16_t)(id & (BUCKET_BINS - 1));
            }
        }
        __syncthreads();

        // (d) copy-out run by run, eight lanes per run, 64 runs per step: run o = stage[lcur[o-1] .. lcur[o]) -> elems + delta[o]
        {
            const uint32_t sub = j & 7u;
            for (uint32_t o = j >> 3; o < (uint32_t)WIDE_B; o += P1_THREADS / 8) {
                const uint32_t start = o ? P.lcur[o - 1] : 0u, d = P.delta[o];
                uint32_t end = P.lcur[o];
                end = end < start + WIDE_LONG_RUN ? end : start + WIDE_LONG_RUN;
                uint16_t v[WIDE_LONG_RUN / 8];
#pragma unroll
                for (uint32_t u = 0; u < WIDE_LONG_RUN / 8; u++) { const uint32_t sl = start + sub + 8u * u; v[u] = sl < end ? P.u.stage[sl] : (uint16_t)0; }
#pragma unroll
                for (uint32_t u = 0; u < WIDE_LONG_RUN / 8; u++) { const uint32_t sl = start + sub + 8u * u; if (sl < end) elems[(uint64_t)d + sl] = v[u]; }
            }
            const uint32_t nlong = P.nlong;                           // the tails of long runs (skew): all threads per run
            for (uint32_t i = 0; i < nlong; i++) {
                const uint32_t o = P.longb[i];
                const uint32_t start = (o ? P.lcur[o - 1] : 0u) + WIDE_LONG_RUN, end = P.lcur[o], d = P.delta[o];
                for (uint32_t sl = start + j; sl < end; sl += P1_THREADS) elems[(uint64_t)d + sl] = P.u.stage[sl];
            }
        }
        __syncthreads();          // stage / lcur / delta are rewritten by the next tile
    }
    if (EXPAND) {
        unsigned long long we = wave_sum(expanded);
        if ((j & 63) == 0 && we) __hip_atomic_fetch_add(&ctr->total_kmers, we, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
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

// add elems[g0 .. g1) to the workgroup's LDS histogram: 16-byte loads on the aligned middle, four in flight per lane
__device__ __forceinline__ void hist_accumulate(uint32_t *hist, const uint16_t *__restrict__ elems, uint64_t g0, uint64_t g1, int tid)
{
    uint64_t a0 = (g0 + 7ull) & ~7ull; if (a0 > g1) a0 = g1;
    uint64_t a1 = g1 & ~7ull; if (a1 < a0) a1 = a0;
    for (uint64_t g = g0 + tid; g < a0; g += P2_THREADS) lds_hist_add(hist, elems[g]);
    const uint4 *v4 = reinterpret_cast<const uint4 *>(elems);
    const uint64_t v1 = a1 / 8;
    uint64_t v = a0 / 8 + tid;
    for (; v + 3ull * P2_THREADS < v1; v += 4ull * P2_THREADS) {
        uint4 x[4];
#pragma unroll
        for (int u = 0; u < 4; u++) x[u] = v4[v + (uint64_t)u * P2_THREADS];
#pragma unroll
        for (int u = 0; u < 4; u++) hist_add8(hist, x[u]);
    }
    for (; v < v1; v += P2_THREADS) hist_add8(hist, v4[v]);
    for (uint64_t g = a1 + tid; g < g1; g += P2_THREADS) lds_hist_add(hist, elems[g]);
}

// add the LDS histogram to 32768 consecutive bins of the vector
__device__ __forceinline__ void hist_flush(const uint32_t *hist, unsigned long long *__restrict__ dst, bool only_writer, int tid,
                                           bool dst_is_zero = false)
{
    if (only_writer && dst_is_zero) {
        // nothing has been added to the vector since it was cleared: the counts ARE the new values.  Every bin is stored,
        // zeros included: whole lines go out as a pure write stream (a masked store of the non-zero bins alone would
        // make the memory side read the rest of each line back)
        for (int base = 0; base < BUCKET_BINS; base += 8 * P2_THREADS) {
#pragma unroll
            for (int u = 0; u < 8; u++) dst[base + u * P2_THREADS + tid] = (unsigned long long)hist[base + u * P2_THREADS + tid];
        }
    } else if (only_writer) {
        // this workgroup is the only writer of these 32768 bins during this launch: plain read-modify-write, eight
        // loads in flight per lane (a load-add-store chain per bin would expose the HBM latency 32 times over)
        for (int base = 0; base < BUCKET_BINS; base += 8 * P2_THREADS) {
            uint32_t c[8];
            unsigned long long v[8];
#pragma unroll
            for (int u = 0; u < 8; u++) c[u] = hist[base + u * P2_THREADS + tid];
#pragma unroll
            for (int u = 0; u < 8; u++) v[u] = c[u] ? dst[base + u * P2_THREADS + tid] : 0ull;
#pragma unroll
            for (int u = 0; u < 8; u++) if (c[u]) dst[base + u * P2_THREADS + tid] = v[u] + c[u];
        }
    } else {
        for (int i = tid; i < BUCKET_BINS; i += P2_THREADS) {
            const uint32_t c = hist[i];
            if (c) __hip_atomic_fetch_add(&dst[i], (unsigned long long)c, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
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

__global__ void __launch_bounds__(P2_THREADS)
bucket_hist_kernel(const uint16_t *__restrict__ elems, const uint32_t *__restrict__ bucket_base,
                   const uint32_t *__restrict__ slice_base, uint32_t nbuckets, unsigned long long *__restrict__ table)
{
    __shared__ uint32_t hist[BUCKET_BINS];
    const int tid = threadIdx.x;
    uint32_t b, s, nslices;
    if (!p2_locate(slice_base, nbuckets, blockIdx.x, &b, &s, &nslices)) return;
    const uint64_t base = bucket_base[b], n = (uint64_t)bucket_base[b + 1] - base;
    const uint64_t g0 = base + n * (uint64_t)s / (uint64_t)nslices;
    const uint64_t g1 = base + n * (uint64_t)(s + 1) / (uint64_t)nslices;
    if (g1 == g0) return;
    for (int i = tid; i < BUCKET_BINS; i += P2_THREADS) hist[i] = 0;
    __syncthreads();
    hist_accumulate(hist, elems, g0, g1, tid);
    __syncthreads();
    hist_flush(hist, table + ((uint64_t)b << BIN_BITS), nslices == 1, tid);
}

// ---------------------------------------------------------------------------------
// P2 over several partitioned batches at once (kdb_twolevel.hip.h defers the flush: for k >= 15 the pass over the
// vector costs more than the histogramming, so it is paid once per PENDING_MAX batches, not once per batch).
// Every pending batch has its own element array and bucket bases; slice s of n takes the s-th n-th of each.
// ---------------------------------------------------------------------------------
constexpr int PENDING_MAX = 16;
struct PendingSet {
    const uint16_t *elems[PENDING_MAX];
    const uint32_t *base[PENDING_MAX];       // [nbuckets + 1] each
    int n;
};

__global__ void __launch_bounds__(1024)
pending_slice_kernel(PendingSet set, uint32_t R, uint32_t slice_elems, uint32_t *__restrict__ slice_base /* [R + 1] */)
{
    __shared__ uint32_t wsum[1024 / 64];
    const uint32_t j = threadIdx.x;
    const uint32_t per = (R + 1023u) / 1024u;
    const uint32_t lo = j * per < R ? j * per : R, hi = (lo + per < R) ? lo + per : R;
    auto nslices_of = [&](uint32_t i) {
        uint64_t v = 0;
        for (int p = 0; p < set.n; p++) v += (uint64_t)(set.base[p][i + 1] - set.base[p][i]);
        return (uint32_t)((v + slice_elems - 1) / slice_elems);
    };
    uint32_t ssum = 0;
    for (uint32_t i = lo; i < hi; i++) ssum += nslices_of(i);
    uint32_t stot;
    uint32_t srun = block_excl_scan<1024>(ssum, wsum, &stot);
    for (uint32_t i = lo; i < hi; i++) { slice_base[i] = srun; srun += nslices_of(i); }
    if (j == 0) slice_base[R] = stot;
}

__global__ void __launch_bounds__(P2_THREADS)
pending_hist_kernel(PendingSet set, const uint32_t *__restrict__ slice_base, uint32_t nbuckets, unsigned long long *__restrict__ table,
                    int table_is_zero)
{
    __shared__ uint32_t hist[BUCKET_BINS];
    const int tid = threadIdx.x;
    uint32_t b, s, nslices;
    if (!p2_locate(slice_base, nbuckets, blockIdx.x, &b, &s, &nslices)) return;
    for (int i = tid; i < BUCKET_BINS; i += P2_THREADS) hist[i] = 0;
    __syncthreads();
    for (int p = 0; p < set.n; p++) {
        const uint64_t base = set.base[p][b], n = (uint64_t)set.base[p][b + 1] - base;
        const uint64_t g0 = base + n * (uint64_t)s / (uint64_t)nslices;
        const uint64_t g1 = base + n * (uint64_t)(s + 1) / (uint64_t)nslices;
        if (g1 > g0) hist_accumulate(hist, set.elems[p], g0, g1, tid);
    }
    __syncthreads();
    hist_flush(hist, table + ((uint64_t)b << BIN_BITS), nslices == 1, tid, table_is_zero != 0);
}

// ---------------------------------------------------------------------------------
// host: run the LDS-histogram path over one device-resident batch
// ---------------------------------------------------------------------------------
inline int partition_count(PartitionState &st, hipStream_t stream, const uint8_t *d_bases, size_t nbytes, int k, int canonical,
                           int n_expand, unsigned long long *d_table, DevCounters *d_ctr, ProfHook &prof)
{
#define KDB_P_ALLOC(expr)                                                           \
    do {                                                                            \
        hipError_t _e = (expr);                                                     \
        if (_e != hipSuccess) { (void)hipGetLastError(); partition_error_ref() = "scratch allocation failed"; return 2; } \
    } while (0)
#define KDB_P_TRY(expr)                                                             \
    do {                                                                            \
        hipError_t _e = (expr);                                                     \
        if (_e != hipSuccess) { partition_error_ref() = hipGetErrorString(_e); return 1; } \
    } while (0)
    const uint64_t ntiles_all = (nbytes + TILE_BYTES - 1) / TILE_BYTES;
    if (k <= SMALLK_MAX) {
        const uint32_t grid = (uint32_t)(ntiles_all < 512 ? ntiles_all : 512);
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
        KDB_P_ALLOC(hipMalloc((void **)&st.d_bucket_total, MAXB * sizeof(uint32_t)));
        KDB_P_ALLOC(hipMalloc((void **)&st.d_bucket_base, (MAXB + 1) * sizeof(uint32_t)));
        KDB_P_ALLOC(hipMalloc((void **)&st.d_slice_base, (MAXB + 1) * sizeof(uint32_t)));
        KDB_P_ALLOC(hipMalloc((void **)&st.d_wg_cnt, (size_t)WG_CNT_ROWS * PERSIST_GRID * sizeof(uint32_t)));
    }
    {
        size_t need_tiles = (size_t)(ntiles_all < (1ull << 31) / TILE_BYTES ? ntiles_all : (1ull << 31) / TILE_BYTES);
        if (k == 13) need_tiles *= ALLPASS;                     // one count matrix per id range
        if (st.tile_cnt_cap < need_tiles) {
            if (st.d_tile_cnt) { KDB_P_TRY(hipStreamSynchronize(stream)); (void)hipFree(st.d_tile_cnt); st.d_tile_cnt = nullptr; st.tile_cnt_cap = 0; }
            KDB_P_ALLOC(hipMalloc((void **)&st.d_tile_cnt, need_tiles * MAXB * sizeof(uint16_t)));
            st.tile_cnt_cap = need_tiles;
        }
    }
    // sub-batches keep element indices within 32 bits
    const uint64_t max_tiles = (1ull << 31) / TILE_BYTES;       // 2 Gi positions per sub-batch
    const size_t need = (size_t)((ntiles_all < max_tiles ? ntiles_all : max_tiles) * (uint64_t)TILE_BYTES);
    if (st.elems_cap < need) {
        if (st.d_elems) { KDB_P_TRY(hipStreamSynchronize(stream)); (void)hipFree(st.d_elems); st.d_elems = nullptr; st.elems_cap = 0; }
        KDB_P_ALLOC(hipMalloc((void **)&st.d_elems, need * sizeof(uint16_t) + 64));
        st.elems_cap = need;
    }
    const bool use_img = st.reuse_image && !n_expand && (k <= 12 || (k == 13 && st.wide));
    if (use_img && st.img_cap < need / 16) {
        if (st.d_img) { KDB_P_TRY(hipStreamSynchronize(stream)); (void)hipFree(st.d_img); st.d_img = nullptr; st.img_cap = 0; }
        KDB_P_ALLOC(hipMalloc((void **)&st.d_img, 2 * (need / 16 + 1) * sizeof(uint32_t)));
        st.img_cap = need / 16;
    }
    uint32_t *const img_fwd = use_img ? st.d_img : nullptr, *const img_msk = use_img ? st.d_img + st.img_cap + 1 : nullptr;
    const int kk = k > 12 ? 12 : k;                             // bits below PASS_SHIFT describe a k=12-sized id range
    const int nbuckets = 1 << (2 * kk - BIN_BITS);
    const uint32_t npass = 1u << (2 * (k - kk));                // 1, 4 (k=13), 16 (k=14)
    const uint32_t Gmax = st.grid > 0 ? (uint32_t)st.grid : (uint32_t)PART_GRID_DEFAULT;
    for (uint64_t t0 = 0; t0 < ntiles_all; t0 += max_tiles) {
        const uint32_t nt = (uint32_t)((ntiles_all - t0) < max_tiles ? (ntiles_all - t0) : max_tiles);
        const uint32_t G = nt < Gmax ? nt : Gmax;
        const bool allpass = npass == (uint32_t)ALLPASS;
        const size_t range_tiles = (size_t)nt * (MAXB / 2), range_wg = (size_t)MAXB * G;
        if (allpass && !st.d_wide) KDB_P_ALLOC(hipMalloc((void **)&st.d_wide, (3 * (size_t)WIDE_B + 2) * sizeof(uint32_t)));
        if (allpass) {
            prof.begin(KDB_KERNEL_BUCKET_COUNT);
            if (canonical)
                hipLaunchKernelGGL((bucket_count_allpass_kernel<true>), dim3(G), dim3(TPB), 0, stream, d_bases, (uint64_t)nbytes, (uint32_t)t0, nt, k,
                                   (uint32_t *)st.d_tile_cnt, range_tiles, st.d_wg_cnt, range_wg, d_ctr, img_fwd, img_msk);
            else
                hipLaunchKernelGGL((bucket_count_allpass_kernel<false>), dim3(G), dim3(TPB), 0, stream, d_bases, (uint64_t)nbytes, (uint32_t)t0, nt, k,
                                   (uint32_t *)st.d_tile_cnt, range_tiles, st.d_wg_cnt, range_wg, d_ctr, img_fwd, img_msk);
            prof.end();
        }
        if (allpass && st.wide) {
            // one scatter pass over 2048 buckets (partition_wide_kernel), one P2 launch
            uint32_t *const tot = st.d_wide, *const base = st.d_wide + WIDE_B, *const slice = st.d_wide + 2 * WIDE_B + 1;
            const uint64_t positions = (uint64_t)nt * TILE_BYTES;
            const uint32_t target = st.slices > 0 ? (uint32_t)st.slices : 2048u;
            uint64_t se = (positions + target - 1) / target;
            if (se < 65536) se = 65536;
            const uint32_t slice_elems = (uint32_t)se;
            const uint32_t p2_grid = (uint32_t)(positions / slice_elems) + (uint32_t)WIDE_B + 1u;
            prof.begin(KDB_KERNEL_BUCKET_SCAN);
            hipLaunchKernelGGL(wg_scan_kernel, dim3(WIDE_B), dim3(TPB), 0, stream, st.d_wg_cnt, G, tot);
            hipLaunchKernelGGL(wide_scan_kernel, dim3(1), dim3(P1_THREADS), 0, stream, tot, base, slice, slice_elems, d_ctr);
            prof.end();
            prof.begin(KDB_KERNEL_PARTITION);
#define KDB_LAUNCH_PW(E, C)                                                                                                         \
    hipLaunchKernelGGL((partition_wide_kernel<E, C>), dim3(G), dim3(P1_THREADS), 0, stream, d_bases, (uint64_t)nbytes, (uint32_t)t0, nt, \
                       k, st.d_elems, base, st.d_wg_cnt, st.d_tile_cnt, range_tiles * 2, d_table, d_ctr, img_fwd, img_msk)
            if (n_expand) { if (canonical) KDB_LAUNCH_PW(true, true); else KDB_LAUNCH_PW(true, false); }
            else          { if (canonical) KDB_LAUNCH_PW(false, true); else KDB_LAUNCH_PW(false, false); }
#undef KDB_LAUNCH_PW
            prof.end();
            prof.begin(KDB_KERNEL_BUCKET_HIST);
            hipLaunchKernelGGL(bucket_hist_kernel, dim3(p2_grid), dim3(P2_THREADS), 0, stream, st.d_elems, base, slice, (uint32_t)WIDE_B, d_table);
            prof.end();
            KDB_P_TRY(hipGetLastError());
            continue;
        }
        for (uint32_t pass = 0; pass < npass; pass++) {
            uint32_t *const wg_cnt = st.d_wg_cnt + (allpass ? pass * range_wg : 0);
            const uint16_t *const tile_cnt = st.d_tile_cnt + (allpass ? pass * range_tiles * 2 : 0);
            if (!allpass) {
            prof.begin(KDB_KERNEL_BUCKET_COUNT);
#define KDB_LAUNCH_P0(C, M)                                                                                                  \
    hipLaunchKernelGGL((bucket_count_kernel<C, M>), dim3(G), dim3(TPB), 0, stream, d_bases, (uint64_t)nbytes, (uint32_t)t0, nt, k, \
                       pass, (uint32_t *)st.d_tile_cnt, st.d_wg_cnt, d_ctr, img_fwd, img_msk)
            if (npass > 1) { if (canonical) KDB_LAUNCH_P0(true, true); else KDB_LAUNCH_P0(false, true); }
            else           { if (canonical) KDB_LAUNCH_P0(true, false); else KDB_LAUNCH_P0(false, false); }
#undef KDB_LAUNCH_P0
            prof.end();
            }
            prof.begin(KDB_KERNEL_BUCKET_SCAN);
            hipLaunchKernelGGL(wg_scan_kernel, dim3(MAXB), dim3(TPB), 0, stream, wg_cnt, G, st.d_bucket_total);
            // P2 slices: about `target` workgroups in total, each bucket cut in proportion to its size (fewer, larger
            // slices win: each slice zeroes and flushes a 128 KiB histogram, and single-slice buckets flush without atomics)
            const uint64_t positions = (uint64_t)nt * TILE_BYTES;
            const uint32_t target = st.slices > 0 ? (uint32_t)st.slices : 512u;      // option p2_slices = target workgroup count
            uint64_t se = (positions / npass + target - 1) / target;
            if (se < 65536) se = 65536;
            const uint32_t slice_elems = (uint32_t)se;
            const uint32_t p2_grid = (uint32_t)(positions / slice_elems) + (uint32_t)nbuckets + 1u;
            hipLaunchKernelGGL(bucket_scan_kernel, dim3(1), dim3(MAXB), 0, stream, st.d_bucket_total, st.d_bucket_base,
                               st.d_slice_base, slice_elems, 1, d_ctr);
            prof.end();
            prof.begin(KDB_KERNEL_PARTITION);
#define KDB_LAUNCH_P1(E, C, M)                                                                                                   \
    hipLaunchKernelGGL((partition_kernel<E, C, M>), dim3(G), dim3(P1_THREADS), 0, stream, d_bases, (uint64_t)nbytes, (uint32_t)t0, nt, \
                       k, pass, st.d_elems, st.d_bucket_base, wg_cnt, tile_cnt, d_table, d_ctr, img_fwd, img_msk)
            if (n_expand) {
                if (npass > 1) { if (canonical) KDB_LAUNCH_P1(true, true, true); else KDB_LAUNCH_P1(true, false, true); }
                else           { if (canonical) KDB_LAUNCH_P1(true, true, false); else KDB_LAUNCH_P1(true, false, false); }
            } else {
                if (npass > 1) { if (canonical) KDB_LAUNCH_P1(false, true, true); else KDB_LAUNCH_P1(false, false, true); }
                else           { if (canonical) KDB_LAUNCH_P1(false, true, false); else KDB_LAUNCH_P1(false, false, false); }
            }
#undef KDB_LAUNCH_P1
            prof.end();
            prof.begin(KDB_KERNEL_BUCKET_HIST);
            hipLaunchKernelGGL(bucket_hist_kernel, dim3(p2_grid), dim3(P2_THREADS), 0, stream, st.d_elems,
                               st.d_bucket_base, st.d_slice_base, (uint32_t)MAXB, d_table + ((uint64_t)pass << PASS_SHIFT));
            prof.end();
        }
        KDB_P_TRY(hipGetLastError());
    }
    return 0;
#undef KDB_P_TRY
#undef KDB_P_ALLOC
}

}  // namespace kdb
