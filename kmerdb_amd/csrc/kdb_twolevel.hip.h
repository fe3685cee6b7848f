// kdb_twolevel.hip.h -- two-level radix partition for 13 <= k <= 17 (ids of 26..34 bits).
//
//   id = [ L1 digit : 2k-24 bits ][ bucket : 9 bits ][ bin : 15 bits ]
//
// Level 1 (from the residues, once):   ids are scattered by their L1 digit (4 / 16 / 64 / 256 / 1024 buckets) into a
//   u32 array of 24-bit remainders.          l1_count_kernel -> scans -> l1_partition_kernel
// Level 2 (per L1 bucket, on id arrays): exactly the k = 12 pipeline of kdb_partition.hip.h with the front end
//   replaced by a coalesced load of ids.     ids_count_kernel -> scans -> ids_partition_kernel -> bucket_hist_kernel
//
// The histogram pass is deferred: a partitioned batch is kept (PendingPart) and P2 runs over up to PENDING_MAX batches
// at once (twolevel_flush: at kdb_sync / kdb_finish, after PENDING_MAX batches, or over budget), because for k >= 15
// one pass over the 4^k vector costs more than everything else in a batch.
//
// Every id is read from HBM as a residue once, written/read as u32 once and as u16 once: 13 B/k-mer of traffic
// instead of the >= 64 B a random 64-bit RMW costs, and no global atomics on any scatter path.
// Same counting semantics as everywhere else (kmer.py:234-317, :526-565; parse.py:133-136).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include <vector>

#include "kdb_partition.hip.h"

namespace kdb {

constexpr int L1_SHIFT = 24;                       // bits below the L1 digit
constexpr int L1_THREADS = 512;
constexpr int L1_HALF_CHUNKS = TILE_CHUNKS / 2;    // the tile is scattered in two halves of 8192 positions
constexpr int L1_HALF_POS = L1_HALF_CHUNKS * 16;
constexpr int MAXD1 = 1024;                        // L1 digits at k = 17 (256 at k = 16)
static_assert(L1_HALF_CHUNKS == L1_THREADS, "one chunk per thread per half");

// count `digit` into cnt[] from all active lanes.  Few digits => heavy same-address conflicts => match by ballot.
__device__ __forceinline__ void digit_count(uint32_t *cnt, uint32_t digit, bool valid, int few_digits)
{
    if (few_digits) {
        uint64_t todo = __ballot(valid);
        while (todo) {                                            // one round per distinct digit present in the wave
            const int first = __ffsll((unsigned long long)todo) - 1;
            const uint32_t d0 = (uint32_t)__builtin_amdgcn_readlane((int)digit, first);
            const uint64_t same = __ballot(valid && digit == d0);
            if (valid && digit == d0 && lane_rank_in(same) == 0) atomicAdd(&cnt[d0], (uint32_t)__popcll(same));
            todo &= ~same;
        }
    } else if (valid) {
        atomicAdd(&cnt[digit], 1u);
    }
}

// slot = cursor[digit]++ for all active valid lanes (same aggregation)
__device__ __forceinline__ uint32_t digit_take(uint32_t *cur, uint32_t digit, bool valid, int few_digits)
{
    uint32_t slot = 0;
    if (few_digits) {
        uint64_t todo = __ballot(valid);
        while (todo) {
            const int first = __ffsll((unsigned long long)todo) - 1;
            const uint32_t d0 = (uint32_t)__builtin_amdgcn_readlane((int)digit, first);
            const bool mine = valid && digit == d0;
            const uint64_t same = __ballot(mine);
            const uint32_t r = lane_rank_in(same);
            uint32_t base = 0;
            if (mine && r == 0) base = atomicAdd(&cur[d0], (uint32_t)__popcll(same));
            base = (uint32_t)__builtin_amdgcn_readlane((int)base, __ffsll((unsigned long long)same) - 1);
            if (mine) slot = base + r;
            todo &= ~same;
        }
    } else if (valid) {
        slot = atomicAdd(&cur[digit], 1u);
    }
    return slot;
}

// ---------------------------------------------------------------------------------
// L1 P0: per-(digit, workgroup) sizes; also counts bad residues.  Persistent, tile ownership w, w+G, ...
// ---------------------------------------------------------------------------------
template <typename ID, int D1>
__global__ void __launch_bounds__(TPB)
l1_count_kernel(const uint8_t *__restrict__ bases, uint64_t nbytes, uint32_t tile0, uint32_t ntiles, int k, int canonical,
                int few_digits, uint32_t *__restrict__ wg_cnt /* [D1][gridDim.x] */, DevCounters *ctr)
{
    __shared__ TileLds<false> L;
    __shared__ uint32_t cnt[D1];
    __shared__ unsigned long long s_bad;
    const int j = threadIdx.x;
    for (int d = j; d < D1; d += TPB) cnt[d] = 0;
    if (j == 0) s_bad = 0;
    unsigned long long nbad_tot = 0;
    const UniformStarts ulen(batch_uniform_len(ctr), TPB);
    const IdParams<ID> idp(k, canonical);
    for (uint32_t t = blockIdx.x; t < ntiles; t += gridDim.x) {
        uint32_t nbad;
        __syncthreads();
        stage_tile(L, bases, nbytes, (uint64_t)tile0 + t, &nbad, ulen);
        nbad_tot += nbad;
        __syncthreads();
#pragma unroll 1
        for (int q = 0; q < CHUNKS_PER_THREAD; q++) {
            const Hood h = load_hood(L, j + q * TPB);
            const uint32_t bad16 = windows_bad16(h, k);
#pragma unroll
            for (int i = 0; i < 16; i++)
                digit_count(cnt, (uint32_t)(idp.id(h, i) >> L1_SHIFT), !((bad16 >> i) & 1u), few_digits);
        }
    }
    __syncthreads();
    for (int d = j; d < D1; d += TPB) wg_cnt[(size_t)d * gridDim.x + blockIdx.x] = cnt[d];
    unsigned long long wb = wave_sum(nbad_tot);
    if ((j & 63) == 0 && wb) atomicAdd(&s_bad, wb);
    __syncthreads();
    if (j == 0 && s_bad) __hip_atomic_fetch_add(&ctr->n_bad, s_bad, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

// ---------------------------------------------------------------------------------
// L1 P1: residues -> u32 remainders grouped by L1 digit.  Each 16 KiB tile is scattered in two halves of 8192
// positions (one 16-base chunk per thread per half): count, scan, place (ids stay in registers), copy out.
// ---------------------------------------------------------------------------------
// D1 digits; ids wider than 32 bits (k = 17: 10-bit digit + 24-bit remainder) keep digit bits 8.. in a byte array
template <bool EXPAND, int D1, bool WIDE>
struct L1Lds {
    TileLds<EXPAND> tile;
    uint32_t stage[L1_HALF_POS];        // low 32 bits of the id (remainder + low 8 digit bits)
    uint8_t stage_hi[WIDE ? L1_HALF_POS : 4];
    uint32_t cnt[D1];
    uint32_t lcur[D1];
    uint32_t delta[D1];
    uint32_t wsum[L1_THREADS / 64];
    uint32_t nids;
};

template <typename ID, int D1, bool EXPAND>
__global__ void __launch_bounds__(L1_THREADS)
l1_partition_kernel(const uint8_t *__restrict__ bases, uint64_t nbytes, uint32_t tile0, uint32_t ntiles, int k, int canonical,
                    int few_digits, uint32_t *__restrict__ elems32, const uint32_t *__restrict__ l1_base,
                    const uint32_t *__restrict__ wg_off /* [D1][gridDim.x] */,
                    unsigned long long *__restrict__ table, DevCounters *ctr)
{
    constexpr bool WIDE = sizeof(ID) > 4;
    constexpr int DPT = D1 > L1_THREADS ? D1 / L1_THREADS : 1;      // digits whose running cursor a thread owns: j*DPT .. j*DPT+DPT-1
    __shared__ L1Lds<EXPAND, D1, WIDE> P;
    const int j = threadIdx.x;
    uint32_t cur[DPT];
#pragma unroll
    for (int u = 0; u < DPT; u++) {
        const int d = j * DPT + u;
        cur[u] = d < D1 ? l1_base[d] + wg_off[(size_t)d * gridDim.x + blockIdx.x] : 0u;
    }
    const IdParams<ID> idp(k, canonical);
    const uint32_t kmask = (1u << k) - 1u, k1mask = kmask >> 1;
    const uint64_t idmask = (1ull << (2 * k)) - 1ull;
    const UniformStarts ulen(batch_uniform_len(ctr), L1_THREADS);
    unsigned long long expanded = 0;

    for (uint32_t t = blockIdx.x; t < ntiles; t += gridDim.x) {
        uint32_t nbad;
        stage_tile<EXPAND, L1_THREADS, false>(P.tile, bases, nbytes, (uint64_t)tile0 + t, &nbad, ulen);
        for (int half = 0; half < 2; half++) {
#pragma unroll
            for (int u = 0; u < DPT; u++) if (j * DPT + u < D1) P.cnt[j * DPT + u] = 0;
            __syncthreads();                                        // tile staged (half 0) / previous copy-out done; cnt zeroed
            const int c = half * L1_HALF_CHUNKS + j;
            const Hood h = load_hood(P.tile, c);
            uint32_t N32 = 0;
            if (EXPAND) N32 = (P.tile.nn[c] & 0xFFFFu) | (P.tile.nn[c + 1] << 16);
            uint32_t ids[16];                                       // low 32 bits
            uint32_t hi = 0;                                        // WIDE: bits 32, 33 of the 16 ids
            const uint32_t bad16 = windows_bad16(h, k);           // one sliding-window OR instead of two extracts + compare per window
            const uint32_t vmask = ~bad16 & 0xFFFFu;
#pragma unroll
            for (int i = 0; i < 16; i++) {
                const bool valid = (vmask >> i) & 1u;
                const ID id = idp.id(h, i);
                ids[i] = (uint32_t)id;
                if (WIDE) hi |= ((uint32_t)((uint64_t)id >> 32) & 3u) << (2 * i);
                digit_count(P.cnt, (uint32_t)(id >> L1_SHIFT), valid, few_digits);
                if (EXPAND && !valid && !window_crosses(h, i, k1mask)) {
                    const uint32_t vwin = (h.V >> i) & kmask, nwin = (N32 >> i) & kmask;
                    if (nwin == vwin) expand_n_window(table, h.F(), i, k, canonical, idmask, nwin, &expanded, ctr);
                }
            }
            __syncthreads();
            {
                uint32_t cc[DPT], sum = 0;
#pragma unroll
                for (int u = 0; u < DPT; u++) { cc[u] = (j * DPT + u < D1) ? P.cnt[j * DPT + u] : 0u; sum += cc[u]; }
                uint32_t tot;
                uint32_t run = block_excl_scan<L1_THREADS>(sum, P.wsum, &tot);
#pragma unroll
                for (int u = 0; u < DPT; u++) {
                    if (j * DPT + u < D1) {
                        P.lcur[j * DPT + u] = run;
                        P.delta[j * DPT + u] = cur[u] - run;
                        cur[u] += cc[u];
                        run += cc[u];
                    }
                }
                if (j == 0) P.nids = tot;
            }
            __syncthreads();
#pragma unroll
            for (int i = 0; i < 16; i++) {
                const bool valid = (vmask >> i) & 1u;
                const uint32_t h2 = WIDE ? (hi >> (2 * i)) & 3u : 0u;
                const uint32_t slot = digit_take(P.lcur, (ids[i] >> L1_SHIFT) | (h2 << 8), valid, few_digits);
                if (valid) { P.stage[slot] = ids[i]; if (WIDE) P.stage_hi[slot] = (uint8_t)h2; }
            }
            __syncthreads();
            const uint32_t nids = P.nids;
#pragma unroll 4
            for (uint32_t sl = j; sl < nids; sl += L1_THREADS) {
                const uint32_t v = P.stage[sl];
                const uint32_t d = (v >> L1_SHIFT) | (WIDE ? (uint32_t)P.stage_hi[sl] << 8 : 0u);
                elems32[(uint64_t)P.delta[d] + sl] = v & ((1u << L1_SHIFT) - 1u);
            }
            // the next half's first barrier orders this copy-out before stage/cnt are rewritten
        }
        __syncthreads();                                            // tile LDS is restaged by the next iteration
    }
    if (EXPAND) {
        unsigned long long we = wave_sum(expanded);
        if ((j & 63) == 0 && we) __hip_atomic_fetch_add(&ctr->total_kmers, we, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
}

// ---------------------------------------------------------------------------------
// L2 setup: tiles of 16384 ids per L1 bucket -> exclusive scan of the tile counts (one tiny workgroup)
// ---------------------------------------------------------------------------------
__global__ void __launch_bounds__(MAXD1)
l2_setup_kernel(const uint32_t *__restrict__ l1_base, uint32_t nb1, uint32_t *__restrict__ tile_base /* [nb1 + 1] */)
{
    __shared__ uint32_t wsum[MAXD1 / 64];
    const uint32_t j = threadIdx.x;
    const uint32_t nt = (j < nb1) ? (l1_base[j + 1] - l1_base[j] + TILE_POS - 1) / TILE_POS : 0u;
    uint32_t tot;
    const uint32_t excl = block_excl_scan<MAXD1>(nt, wsum, &tot);
    if (j < nb1) tile_base[j] = excl;
    if (j == 0) tile_base[nb1] = tot;
}

// four consecutive ids in one 16-byte load; L1 bucket bases are only 4-byte aligned
typedef uint32_t ids4_t __attribute__((ext_vector_type(4), aligned(4)));

// L1 bucket of global L2 tile T: largest b1 with tile_base[b1] <= T (wave-uniform binary search)
__device__ __forceinline__ uint32_t l2_bucket_of_tile(const uint32_t *__restrict__ tile_base, uint32_t nb1, uint32_t T)
{
    uint32_t lo = 0, hi = nb1 - 1;
    while (lo < hi) { const uint32_t mid = (lo + hi + 1) >> 1; if (tile_base[mid] <= T) lo = mid; else hi = mid - 1; }
    return lo;
}

// ---------------------------------------------------------------------------------
// L2 P0 on id arrays: per-(tile, bucket) counts.  L2 tiles are numbered globally (tile_base[b1] + t), and the
// persistent workgroups take tiles round-robin regardless of the L1 bucket: canonical ids are far from uniform
// over the L1 digits (and real data is skewed), so a per-bucket share of workgroups would leave most of them idle.
// ids[l1_base[b1] .. l1_base[b1+1]) are the 24-bit remainders of L1 bucket b1.
// ---------------------------------------------------------------------------------
__global__ void __launch_bounds__(TPB)
ids_count_kernel(const uint32_t *__restrict__ ids, const uint32_t *__restrict__ l1_base, const uint32_t *__restrict__ tile_base,
                 uint32_t nb1, uint32_t *__restrict__ tile_cnt /* [tiles][MAXB/2] */)
{
    __shared__ uint32_t cnt[MAXB];
    const int j = threadIdx.x;
    const uint32_t ntiles = tile_base[nb1];
    cnt[2 * j] = 0; cnt[2 * j + 1] = 0;
    for (uint32_t T = blockIdx.x; T < ntiles; T += gridDim.x) {
        const uint32_t b1 = l2_bucket_of_tile(tile_base, nb1, T);
        const uint32_t r0 = l1_base[b1], n = l1_base[b1 + 1] - r0;
        const uint32_t base = (T - tile_base[b1]) * (uint32_t)TILE_POS;
        __syncthreads();
        const uint32_t nhere = n - base < (uint32_t)TILE_POS ? n - base : (uint32_t)TILE_POS;      // ids of this tile
        const uint32_t *src = ids + r0 + base;
        for (uint32_t o = 4u * j; o < nhere; o += 16 * TPB) {
            ids4_t v[4];
#pragma unroll
            for (int u = 0; u < 4; u++)                             // four 16-byte loads in flight (the array is padded past its end)
                if (o + u * 4 * TPB < nhere) v[u] = *reinterpret_cast<const ids4_t *>(src + o + u * 4 * TPB);
            uint64_t same; uint32_t b0;
            const bool deg = wave_dominant(v[0][0] >> BIN_BITS, &same, &b0);
#pragma unroll
            for (int u = 0; u < 4; u++) {
#pragma unroll
                for (int e = 0; e < 4; e++) {
                    if (o + u * 4 * TPB + e < nhere) {
                        const uint32_t b = v[u][e] >> BIN_BITS;
                        if (deg) lds_hist_add(cnt, b); else atomicAdd(&cnt[b], 1u);
                    }
                }
            }
        }
        __syncthreads();
        const uint32_t c0 = cnt[2 * j], c1 = cnt[2 * j + 1];
        cnt[2 * j] = 0; cnt[2 * j + 1] = 0;
        tile_cnt[(size_t)T * (MAXB / 2) + j] = c0 | (c1 << 16);
    }
}

// ---------------------------------------------------------------------------------
// per (L1 bucket, bucket): exclusive scan of the tile counts over the tiles of that L1 bucket -> where each tile's
// run starts inside bucket (b1, b), and the bucket's total.  blockIdx = (column group of 64 buckets, L1 bucket);
// 256 threads = 4 tile-quarters x 64 columns, sixteen rows of loads in flight.
// ---------------------------------------------------------------------------------
constexpr int TSCAN_COLS = 64;
__global__ void __launch_bounds__(256)
tile_scan_kernel(const uint16_t *__restrict__ tile_cnt /* [tiles][MAXB] */, const uint32_t *__restrict__ tile_base,
                 uint32_t *__restrict__ tile_off /* [tiles][MAXB] */, uint32_t *__restrict__ bucket_total /* [nb1*MAXB] */)
{
    __shared__ uint32_t part[4][TSCAN_COLS];
    const uint32_t b1 = blockIdx.y, col = blockIdx.x * TSCAN_COLS + (threadIdx.x & (TSCAN_COLS - 1)), q = threadIdx.x / TSCAN_COLS;
    const uint32_t t0 = tile_base[b1], nt = tile_base[b1 + 1] - t0;
    const uint32_t per = (nt + 3) / 4, lo = q * per < nt ? q * per : nt, hi = (lo + per) < nt ? (lo + per) : nt;
    uint32_t sum = 0;
    for (uint32_t t = lo; t < hi; t += 16) {
        uint32_t v[16];
#pragma unroll
        for (int u = 0; u < 16; u++) v[u] = (t + u < hi) ? tile_cnt[(size_t)(t0 + t + u) * MAXB + col] : 0u;
#pragma unroll
        for (int u = 0; u < 16; u++) sum += v[u];
    }
    part[q][threadIdx.x & (TSCAN_COLS - 1)] = sum;
    __syncthreads();
    uint32_t run = 0, tot = 0;
#pragma unroll
    for (int qq = 0; qq < 4; qq++) { const uint32_t x = part[qq][threadIdx.x & (TSCAN_COLS - 1)]; if (qq < (int)q) run += x; tot += x; }
    if (q == 0) bucket_total[(size_t)b1 * MAXB + col] = tot;
    for (uint32_t t = lo; t < hi; t += 16) {
        uint32_t v[16];
#pragma unroll
        for (int u = 0; u < 16; u++) v[u] = (t + u < hi) ? tile_cnt[(size_t)(t0 + t + u) * MAXB + col] : 0u;
#pragma unroll
        for (int u = 0; u < 16; u++)
            if (t + u < hi) { tile_off[(size_t)(t0 + t + u) * MAXB + col] = run; run += v[u]; }
    }
}

// exclusive scan of R = nb1 * 512 bucket totals -> global bucket bases (== positions in the id / element arrays,
// because buckets are laid out in (L1 bucket, bucket) order) and the P2 slice table.  One workgroup.
constexpr int BIGSCAN_THREADS = 1024;
__global__ void __launch_bounds__(BIGSCAN_THREADS)
big_bucket_scan_kernel(const uint32_t *__restrict__ bucket_total, uint32_t R, uint32_t *__restrict__ bucket_base /* [R+1] */,
                       uint32_t *__restrict__ slice_base /* [R+1] */, uint32_t slice_elems, DevCounters *ctr /* total += Sum, or null */)
{
    __shared__ uint32_t wsum[BIGSCAN_THREADS / 64];
    const uint32_t j = threadIdx.x;
    const uint32_t per = (R + BIGSCAN_THREADS - 1) / BIGSCAN_THREADS;
    const uint32_t lo = j * per, hi = (lo + per < R) ? lo + per : R;
    uint32_t sum = 0, ssum = 0;
    for (uint32_t i = lo; i < hi; i++) { const uint32_t v = bucket_total[i]; sum += v; ssum += v ? (v + slice_elems - 1) / slice_elems : 0u; }
    uint32_t tot, stot;
    uint32_t run = block_excl_scan<BIGSCAN_THREADS>(sum, wsum, &tot);
    uint32_t srun = block_excl_scan<BIGSCAN_THREADS>(ssum, wsum, &stot);
    for (uint32_t i = lo; i < hi; i++) {
        const uint32_t v = bucket_total[i];
        bucket_base[i] = run; slice_base[i] = srun;
        run += v; srun += v ? (v + slice_elems - 1) / slice_elems : 0u;
    }
    if (j == 0) {
        bucket_base[R] = tot; slice_base[R] = stot;
        if (ctr && tot) __hip_atomic_fetch_add(&ctr->total_kmers, (unsigned long long)tot, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
}

// The same exclusive scan (bases only) spread over the chip, for R up to 1024 x 512 buckets: one workgroup over
// 524288 totals takes 0.5 ms; three small launches take ~20 us.  Used when the histogram pass is deferred (the slice
// table is then built at flush time from all pending batches).
constexpr int BSCAN_WG = 4096;          // entries per workgroup: 1024 threads x 4
__global__ void __launch_bounds__(BIGSCAN_THREADS)
bscan_local_kernel(const uint32_t *__restrict__ bucket_total, uint32_t R, uint32_t *__restrict__ bucket_base, uint32_t *__restrict__ blk_sum)
{
    __shared__ uint32_t wsum[BIGSCAN_THREADS / 64];
    const uint32_t i0 = blockIdx.x * BSCAN_WG + threadIdx.x * 4u;
    uint32_t v[4];
#pragma unroll
    for (int u = 0; u < 4; u++) v[u] = i0 + u < R ? bucket_total[i0 + u] : 0u;
    uint32_t tot;
    uint32_t run = block_excl_scan<BIGSCAN_THREADS>(v[0] + v[1] + v[2] + v[3], wsum, &tot);
#pragma unroll
    for (int u = 0; u < 4; u++) { if (i0 + u < R) bucket_base[i0 + u] = run; run += v[u]; }
    if (threadIdx.x == 0) blk_sum[blockIdx.x] = tot;
}

__global__ void __launch_bounds__(BIGSCAN_THREADS)
bscan_top_kernel(uint32_t *__restrict__ blk_sum /* in: sums, out: offsets */, uint32_t nblk, uint32_t *__restrict__ bucket_base, uint32_t R)
{
    __shared__ uint32_t wsum[BIGSCAN_THREADS / 64];
    const uint32_t v = threadIdx.x < nblk ? blk_sum[threadIdx.x] : 0u;
    uint32_t tot;
    const uint32_t excl = block_excl_scan<BIGSCAN_THREADS>(v, wsum, &tot);
    if (threadIdx.x < nblk) blk_sum[threadIdx.x] = excl;
    if (threadIdx.x == 0) bucket_base[R] = tot;
}

__global__ void __launch_bounds__(BIGSCAN_THREADS)
bscan_add_kernel(uint32_t *__restrict__ bucket_base, const uint32_t *__restrict__ blk_off, uint32_t R)
{
    const uint32_t off = blk_off[blockIdx.x], i0 = blockIdx.x * BSCAN_WG + threadIdx.x * 4u;
    if (off == 0) return;
#pragma unroll
    for (int u = 0; u < 4; u++) if (i0 + u < R) bucket_base[i0 + u] += off;
}

// ---------------------------------------------------------------------------------
// L2 P1 on id arrays -> 15-bit remainders grouped by (L1 bucket, bucket): the layout P2 expects
// ---------------------------------------------------------------------------------
struct IdsPartLds {
    uint32_t stage[TILE_POS];          // the 24-bit remainder itself: bucket in bits 15..23, bin in bits 0..14
    uint32_t lcur[MAXB];
    uint32_t delta[MAXB];
    uint32_t wsum[P1_THREADS / 64];
    uint32_t nids;
};

__global__ void __launch_bounds__(P1_THREADS)
ids_partition_kernel(const uint32_t *__restrict__ ids, const uint32_t *__restrict__ l1_base, const uint32_t *__restrict__ tile_base,
                     uint32_t nb1, uint16_t *__restrict__ elems, const uint32_t *__restrict__ bucket_base /* [nb1*MAXB + 1] */,
                     const uint32_t *__restrict__ tile_off /* [tiles][MAXB] */, const uint16_t *__restrict__ tile_cnt /* [tiles][MAXB] */)
{
    static_assert(MAXB == P1_THREADS, "one bucket per thread");
    __shared__ IdsPartLds P;
    const int j = threadIdx.x;
    const uint32_t ntiles = tile_base[nb1];
    for (uint32_t T = blockIdx.x; T < ntiles; T += gridDim.x) {
        const uint32_t b1 = l2_bucket_of_tile(tile_base, nb1, T);
        const uint32_t r0 = l1_base[b1], n = l1_base[b1 + 1] - r0;
        const uint32_t base = (T - tile_base[b1]) * (uint32_t)TILE_POS;
        const uint32_t c = tile_cnt[(size_t)T * MAXB + j];
        const uint32_t g = bucket_base[(size_t)b1 * MAXB + j] + tile_off[(size_t)T * MAXB + j];   // where this tile's run of bucket j goes
        uint32_t tot;
        const uint32_t excl = block_excl_scan<P1_THREADS>(c, P.wsum, &tot);     // first barrier inside also fences the previous copy-out
        P.lcur[j] = excl;
        P.delta[j] = g - excl;
        if (j == 0) P.nids = tot;
        __syncthreads();
        for (uint32_t o = j; o < (uint32_t)TILE_POS; o += 4 * P1_THREADS) {
            uint32_t id[4];
            bool ok[4];
#pragma unroll
            for (int u = 0; u < 4; u++) {
                ok[u] = base + o + u * P1_THREADS < n;
                id[u] = ok[u] ? ids[r0 + base + o + u * P1_THREADS] : 0u;
            }
            uint64_t same; uint32_t b0;
            const bool deg = wave_dominant(id[0] >> BIN_BITS, &same, &b0);
#pragma unroll
            for (int u = 0; u < 4; u++) {
                if (ok[u]) {
                    const uint32_t b = id[u] >> BIN_BITS;
                    const uint32_t slot = deg ? lds_cursor_take(P.lcur, b) : atomicAdd(&P.lcur[b], 1u);
                    P.stage[slot] = id[u];
                }
            }
        }
        __syncthreads();
        const uint32_t nids = P.nids;
#pragma unroll 8
        for (uint32_t sl = j; sl < nids; sl += P1_THREADS) {
            const uint32_t v = P.stage[sl];
            elems[(uint64_t)P.delta[v >> BIN_BITS] + sl] = (uint16_t)(v & (BUCKET_BINS - 1));
        }
    }
}

// ---------------------------------------------------------------------------------
// host: two-level count of one device-resident batch (13 <= k <= 16)
// ---------------------------------------------------------------------------------
// one partitioned batch whose histogram pass (P2) has not run yet
struct PendingPart {
    uint16_t *elems = nullptr;         // 15-bit bins grouped by (L1 bucket, bucket)
    size_t cap = 0;                    // in elements
    uint32_t *base2 = nullptr;         // [R + 1] bucket bases inside `elems`
    size_t base_cap = 0;               // in entries
    uint64_t positions = 0;
};

struct TwoLevelState {
    // deferred flush: batches are partitioned as they come, and P2 runs over all pending batches at once (at kdb_sync /
    // kdb_finish, after PENDING_MAX batches, or when they hold more than budget_bytes): one pass over the 4^k vector
    // instead of one per batch
    std::vector<PendingPart> pending, pool;
    size_t pending_bytes = 0, budget_bytes = 0;      // budget 0 = decide at first use (a third of the free memory, <= 64 GiB)
    uint32_t pending_R = 0;
    int defer = 1;
    bool table_is_zero = false;        // the engine cleared the vector and nothing has been added since: the first flush stores instead of adding
    uint32_t *d_elems32 = nullptr;
    size_t cap32 = 0;                  // in elements
    uint32_t *d_l1_total = nullptr;    // [MAXD1]
    uint32_t *d_l1_base = nullptr;     // [MAXD1 + 1]
    uint32_t *d_l1_slice = nullptr;    // [MAXD1 + 1] (unused output of the shared scan kernel)
    uint32_t *d_tile_base = nullptr;   // [MAXD1 + 1]
    uint32_t *d_total2 = nullptr;      // [MAXD1 * MAXB]
    uint32_t *d_base2 = nullptr;       // [MAXD1 * MAXB + 1]
    uint32_t *d_slice2 = nullptr;      // [MAXD1 * MAXB + 1]
    uint32_t *d_tile_off = nullptr;    // [tiles][MAXB]: where each L2 tile's run starts inside its bucket
    size_t tile_off_cap = 0;           // in tiles
};
constexpr int L2_WGS = 4096;           // persistent level-2 workgroups (tiles are dealt round-robin, any L1 bucket)

inline void twolevel_release(PendingPart &pp)
{
    if (pp.elems) (void)hipFree(pp.elems);
    if (pp.base2) (void)hipFree(pp.base2);
    pp = PendingPart();
}

// pending batches are dropped uncounted (kdb_reset)
inline void twolevel_drop_pending(TwoLevelState &tl)
{
    for (auto &pp : tl.pending) tl.pool.push_back(pp);
    tl.pending.clear();
    tl.pending_bytes = 0;
}

// P2 over everything pending; the buffers go back to the pool (stream order makes their reuse safe)
inline int twolevel_flush(TwoLevelState &tl, hipStream_t stream, unsigned long long *d_table, ProfHook &prof)
{
    if (tl.pending.empty()) return 0;
    const int table_is_zero = tl.table_is_zero ? 1 : 0;
    tl.table_is_zero = false;
    PendingSet set;
    uint64_t positions = 0;
    set.n = (int)tl.pending.size();
    for (int p = 0; p < set.n; p++) { set.elems[p] = tl.pending[p].elems; set.base[p] = tl.pending[p].base2; positions += tl.pending[p].positions; }
    for (int p = set.n; p < PENDING_MAX; p++) { set.elems[p] = nullptr; set.base[p] = nullptr; }
    const uint32_t R = tl.pending_R;
    uint64_t se = (positions + 2047) / 2048;
    if (se < 65536) se = 65536;
    if (se > 0x40000000ull) se = 0x40000000ull;
    const uint32_t slice_elems = (uint32_t)se;
    const uint32_t p2_grid = (uint32_t)(positions / slice_elems) + R + 1u;
    prof.begin(KDB_KERNEL_BUCKET_HIST);
    hipLaunchKernelGGL(pending_slice_kernel, dim3(1), dim3(1024), 0, stream, set, R, slice_elems, tl.d_slice2);
    hipLaunchKernelGGL(pending_hist_kernel, dim3(p2_grid), dim3(P2_THREADS), 0, stream, set, tl.d_slice2, R, d_table, table_is_zero);
    prof.end();
    twolevel_drop_pending(tl);
    if (hipGetLastError() != hipSuccess) { partition_error_ref() = "deferred histogram pass failed to launch"; return 1; }
    return 0;
}

inline void twolevel_free(TwoLevelState &tl)
{
    twolevel_drop_pending(tl);
    for (auto &pp : tl.pool) twolevel_release(pp);
    tl.pool.clear();
    if (tl.d_elems32) (void)hipFree(tl.d_elems32);
    if (tl.d_l1_total) (void)hipFree(tl.d_l1_total);
    if (tl.d_l1_base) (void)hipFree(tl.d_l1_base);
    if (tl.d_l1_slice) (void)hipFree(tl.d_l1_slice);
    if (tl.d_tile_base) (void)hipFree(tl.d_tile_base);
    if (tl.d_total2) (void)hipFree(tl.d_total2);
    if (tl.d_base2) (void)hipFree(tl.d_base2);
    if (tl.d_slice2) (void)hipFree(tl.d_slice2);
    if (tl.d_tile_off) (void)hipFree(tl.d_tile_off);
    const int defer = tl.defer;
    tl = TwoLevelState();
    tl.defer = defer;
}

// a buffer set for one more pending batch: from the pool if one fits, else fresh memory
inline bool twolevel_acquire(TwoLevelState &tl, size_t need_elems, size_t need_base, PendingPart *out)
{
    int best = -1;
    for (int i = 0; i < (int)tl.pool.size(); i++)
        if (tl.pool[i].cap >= need_elems && tl.pool[i].base_cap >= need_base && (best < 0 || tl.pool[i].cap < tl.pool[best].cap)) best = i;
    if (best >= 0) { *out = tl.pool[best]; tl.pool.erase(tl.pool.begin() + best); return true; }
    for (int attempt = 0; attempt < 2; attempt++) {
        PendingPart pp;
        if (hipMalloc((void **)&pp.elems, need_elems * sizeof(uint16_t) + 64) == hipSuccess &&
            hipMalloc((void **)&pp.base2, need_base * sizeof(uint32_t)) == hipSuccess) {
            pp.cap = need_elems; pp.base_cap = need_base;
            *out = pp;
            return true;
        }
        (void)hipGetLastError();
        twolevel_release(pp);
        if (attempt == 0) {                                        // give the pooled (too small) buffers back and try once more
            if (tl.pool.empty()) return false;
            (void)hipDeviceSynchronize();
            for (auto &q : tl.pool) twolevel_release(q);
            tl.pool.clear();
        }
    }
    return false;
}

inline bool twolevel_supported(int k) { return k >= 13 && k <= 17; }

inline int twolevel_count(PartitionState &st, TwoLevelState &tl, hipStream_t stream, const uint8_t *d_bases, size_t nbytes, int k,
                          int canonical, int n_expand, unsigned long long *d_table, DevCounters *d_ctr, ProfHook &prof)
{
#define KDB_T_ALLOC(expr)                                                           \
    do {                                                                            \
        hipError_t _e = (expr);                                                     \
        if (_e != hipSuccess) { (void)hipGetLastError(); partition_error_ref() = "scratch allocation failed"; return 2; } \
    } while (0)
#define KDB_T_TRY(expr)                                                             \
    do {                                                                            \
        hipError_t _e = (expr);                                                     \
        if (_e != hipSuccess) { partition_error_ref() = hipGetErrorString(_e); return 1; } \
    } while (0)
    const uint64_t ntiles_all = (nbytes + TILE_BYTES - 1) / TILE_BYTES;
    const uint64_t max_tiles = (1ull << 31) / TILE_BYTES;
    const size_t need_tiles = (size_t)(ntiles_all < max_tiles ? ntiles_all : max_tiles);
    const size_t need = need_tiles * (size_t)TILE_BYTES;
    if (!st.d_bucket_total) {
        KDB_T_ALLOC(hipMalloc((void **)&st.d_bucket_total, MAXB * sizeof(uint32_t)));
        KDB_T_ALLOC(hipMalloc((void **)&st.d_bucket_base, (MAXB + 1) * sizeof(uint32_t)));
        KDB_T_ALLOC(hipMalloc((void **)&st.d_slice_base, (MAXB + 1) * sizeof(uint32_t)));
        KDB_T_ALLOC(hipMalloc((void **)&st.d_wg_cnt, (size_t)WG_CNT_ROWS * PERSIST_GRID * sizeof(uint32_t)));
    }
    if (!tl.d_l1_total) {
        KDB_T_ALLOC(hipMalloc((void **)&tl.d_l1_total, MAXD1 * sizeof(uint32_t)));
        KDB_T_ALLOC(hipMalloc((void **)&tl.d_l1_base, (MAXD1 + 1) * sizeof(uint32_t)));
        KDB_T_ALLOC(hipMalloc((void **)&tl.d_l1_slice, (MAXD1 + 1) * sizeof(uint32_t)));
        KDB_T_ALLOC(hipMalloc((void **)&tl.d_tile_base, (MAXD1 + 1) * sizeof(uint32_t)));
        KDB_T_ALLOC(hipMalloc((void **)&tl.d_total2, (size_t)MAXD1 * MAXB * sizeof(uint32_t)));
        KDB_T_ALLOC(hipMalloc((void **)&tl.d_base2, ((size_t)MAXD1 * MAXB + 1) * sizeof(uint32_t)));
        KDB_T_ALLOC(hipMalloc((void **)&tl.d_slice2, ((size_t)MAXD1 * MAXB + 1) * sizeof(uint32_t)));
    }
    if (tl.tile_off_cap < need_tiles + MAXD1 + 1) {
        if (tl.d_tile_off) { KDB_T_TRY(hipStreamSynchronize(stream)); (void)hipFree(tl.d_tile_off); tl.d_tile_off = nullptr; tl.tile_off_cap = 0; }
        KDB_T_ALLOC(hipMalloc((void **)&tl.d_tile_off, (need_tiles + MAXD1 + 1) * MAXB * sizeof(uint32_t)));
        tl.tile_off_cap = need_tiles + MAXD1 + 1;
    }
    if (st.tile_cnt_cap < need_tiles + MAXD1 + 1) {
        if (st.d_tile_cnt) { KDB_T_TRY(hipStreamSynchronize(stream)); (void)hipFree(st.d_tile_cnt); st.d_tile_cnt = nullptr; st.tile_cnt_cap = 0; }
        KDB_T_ALLOC(hipMalloc((void **)&st.d_tile_cnt, (need_tiles + MAXD1 + 1) * MAXB * sizeof(uint16_t)));
        st.tile_cnt_cap = need_tiles + MAXD1 + 1;
    }
    if (!tl.defer && st.elems_cap < need) {
        if (st.d_elems) { KDB_T_TRY(hipStreamSynchronize(stream)); (void)hipFree(st.d_elems); st.d_elems = nullptr; st.elems_cap = 0; }
        KDB_T_ALLOC(hipMalloc((void **)&st.d_elems, need * sizeof(uint16_t) + 64));
        st.elems_cap = need;
    }
    if (tl.cap32 < need) {
        if (tl.d_elems32) { KDB_T_TRY(hipStreamSynchronize(stream)); (void)hipFree(tl.d_elems32); tl.d_elems32 = nullptr; tl.cap32 = 0; }
        KDB_T_ALLOC(hipMalloc((void **)&tl.d_elems32, need * sizeof(uint32_t) + 64));
        tl.cap32 = need;
    }
    if (tl.defer && tl.budget_bytes == 0) {
        size_t free_b = 0, total_b = 0;
        (void)hipMemGetInfo(&free_b, &total_b);
        tl.budget_bytes = free_b / 3;
        if (tl.budget_bytes > (64ull << 30)) tl.budget_bytes = 64ull << 30;
        if (tl.budget_bytes < (1ull << 30)) tl.budget_bytes = 1ull << 30;
    }
    const int nb1 = 1 << (2 * k - L1_SHIFT);                          // 4, 16, 64, 256, 1024
    const int few = nb1 <= 4 ? 1 : 0;        // 4 digits: 16 lanes per address -> match by ballot; 16+ digits: plain LDS atomics
    const uint32_t Gmax = st.grid > 0 ? (uint32_t)st.grid : (uint32_t)PART_GRID_DEFAULT;
    for (uint64_t t0 = 0; t0 < ntiles_all; t0 += max_tiles) {
        const uint32_t nt = (uint32_t)((ntiles_all - t0) < max_tiles ? (ntiles_all - t0) : max_tiles);
        const uint32_t G = nt < Gmax ? nt : Gmax;
        const uint64_t positions = (uint64_t)nt * TILE_BYTES;
        // where this sub-batch's elements will live.  Acquired before any kernel of the sub-batch is launched: "no room"
        // (return 2) makes the engine count the whole batch with direct atomics, so nothing of it may be counted yet.
        const uint32_t R = (uint32_t)(1 << (2 * k - L1_SHIFT)) * (uint32_t)MAXB;
        PendingPart pp;
        if (tl.defer) {
            if (!tl.pending.empty() && tl.pending_R != R) { if (twolevel_flush(tl, stream, d_table, prof)) return 1; }
            if (!twolevel_acquire(tl, (size_t)positions, (size_t)R + 1, &pp)) {
                if (twolevel_flush(tl, stream, d_table, prof)) return 1;           // pending buffers return to the pool
                if (!twolevel_acquire(tl, (size_t)positions, (size_t)R + 1, &pp)) {
                    partition_error_ref() = "scratch allocation failed";
                    return t0 == 0 ? 2 : 1;                                        // (after the first sub-batch the pool holds a buffer that fits)
                }
            }
        } else {
            pp.elems = st.d_elems; pp.base2 = tl.d_base2;
        }
        // ---- level 1
        prof.begin(KDB_KERNEL_BUCKET_COUNT);
        if (k <= 16)
            hipLaunchKernelGGL((l1_count_kernel<uint32_t, 256>), dim3(G), dim3(TPB), 0, stream, d_bases, (uint64_t)nbytes, (uint32_t)t0, nt, k,
                               canonical, few, st.d_wg_cnt, d_ctr);
        else
            hipLaunchKernelGGL((l1_count_kernel<uint64_t, 1024>), dim3(G), dim3(TPB), 0, stream, d_bases, (uint64_t)nbytes, (uint32_t)t0, nt, k,
                               canonical, few, st.d_wg_cnt, d_ctr);
        prof.end();
        prof.begin(KDB_KERNEL_BUCKET_SCAN);
        hipLaunchKernelGGL(wg_scan_kernel, dim3(nb1), dim3(TPB), 0, stream, st.d_wg_cnt, G, tl.d_l1_total);
        hipLaunchKernelGGL(big_bucket_scan_kernel, dim3(1), dim3(BIGSCAN_THREADS), 0, stream, tl.d_l1_total, (uint32_t)nb1, tl.d_l1_base, tl.d_l1_slice,
                           1u << 30, d_ctr);
        prof.end();
        prof.begin(KDB_KERNEL_PARTITION);
#define KDB_LAUNCH_L1(ID, D, EX)                                                                                                        \
    hipLaunchKernelGGL((l1_partition_kernel<ID, D, EX>), dim3(G), dim3(L1_THREADS), 0, stream, d_bases, (uint64_t)nbytes, (uint32_t)t0, nt, \
                       k, canonical, few, tl.d_elems32, tl.d_l1_base, st.d_wg_cnt, d_table, d_ctr)
        if (k <= 16) { if (n_expand) KDB_LAUNCH_L1(uint32_t, 256, true); else KDB_LAUNCH_L1(uint32_t, 256, false); }
        else         { if (n_expand) KDB_LAUNCH_L1(uint64_t, 1024, true); else KDB_LAUNCH_L1(uint64_t, 1024, false); }
#undef KDB_LAUNCH_L1
        prof.end();
        // ---- level 2: the k = 12 pipeline on every L1 bucket's id array, all buckets per launch (ranges stay on the device)
        uint64_t se = (positions + 2047) / 2048;
        if (se < 65536) se = 65536;
        const uint32_t slice_elems = (uint32_t)se;
        const uint32_t p2_grid = (uint32_t)(positions / slice_elems) + R + 1u;
        prof.begin(KDB_KERNEL_BUCKET_COUNT);
        hipLaunchKernelGGL(l2_setup_kernel, dim3(1), dim3(MAXD1), 0, stream, tl.d_l1_base, (uint32_t)nb1, tl.d_tile_base);
        hipLaunchKernelGGL(ids_count_kernel, dim3(L2_WGS), dim3(TPB), 0, stream, tl.d_elems32, tl.d_l1_base, tl.d_tile_base, (uint32_t)nb1,
                           (uint32_t *)st.d_tile_cnt);
        prof.end();
        prof.begin(KDB_KERNEL_BUCKET_SCAN);
        hipLaunchKernelGGL(tile_scan_kernel, dim3(MAXB / TSCAN_COLS, (unsigned)nb1), dim3(256), 0, stream, st.d_tile_cnt, tl.d_tile_base,
                           tl.d_tile_off, tl.d_total2);
        if (tl.defer && R >= 2u * BSCAN_WG) {
            const uint32_t nblk = (R + BSCAN_WG - 1) / BSCAN_WG;                    // <= 128 (R <= 1024 x 512)
            hipLaunchKernelGGL(bscan_local_kernel, dim3(nblk), dim3(BIGSCAN_THREADS), 0, stream, tl.d_total2, R, pp.base2, tl.d_slice2);
            hipLaunchKernelGGL(bscan_top_kernel, dim3(1), dim3(BIGSCAN_THREADS), 0, stream, tl.d_slice2, nblk, pp.base2, R);
            hipLaunchKernelGGL(bscan_add_kernel, dim3(nblk), dim3(BIGSCAN_THREADS), 0, stream, pp.base2, tl.d_slice2, R);
        } else {
            hipLaunchKernelGGL(big_bucket_scan_kernel, dim3(1), dim3(BIGSCAN_THREADS), 0, stream, tl.d_total2, R, pp.base2, tl.d_slice2, slice_elems, nullptr);
        }
        prof.end();
        prof.begin(KDB_KERNEL_PARTITION);
        hipLaunchKernelGGL(ids_partition_kernel, dim3(L2_WGS / 2), dim3(P1_THREADS), 0, stream, tl.d_elems32, tl.d_l1_base, tl.d_tile_base,
                           (uint32_t)nb1, pp.elems, pp.base2, tl.d_tile_off, st.d_tile_cnt);
        prof.end();
        if (tl.defer) {
            pp.positions = positions;
            tl.pending.push_back(pp);
            tl.pending_R = R;
            tl.pending_bytes += (size_t)positions * sizeof(uint16_t);
            if ((int)tl.pending.size() == PENDING_MAX || tl.pending_bytes >= tl.budget_bytes) { if (twolevel_flush(tl, stream, d_table, prof)) return 1; }
        } else {
            prof.begin(KDB_KERNEL_BUCKET_HIST);
            hipLaunchKernelGGL(bucket_hist_kernel, dim3(p2_grid), dim3(P2_THREADS), 0, stream, st.d_elems, tl.d_base2, tl.d_slice2, R, d_table);
            prof.end();
        }
        KDB_T_TRY(hipGetLastError());
    }
    return 0;
#undef KDB_T_TRY
#undef KDB_T_ALLOC
}

}  // namespace kdb
