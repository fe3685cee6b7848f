// kdb_smallk.hip.h -- k <= 8: the whole 4^k count vector of a workgroup lives in LDS (gfx950).
//
// 4^8 = 65536 bins.  As 16-bit counters they are 128 KiB -- one CU's LDS holds them next to the two tile images of the
// scatter kernels' front end (kdb_scatter.hip.h: sc_stage_chunk, 16 windows per lane from one hood), so the residues are read
// once, every id costs ONE LDS atomic, and nothing but the finished histogram goes to HBM: no rings, no pages, no second pass.
// (Through the paged scatter k = 8 took 2.35 ms per 10 M reads: two buckets, every histogram slice flushed with atomics.)
//
//   k = 8    bin v counts in half v >> 15 of word v & 0x7FFF (the layout page_hist_kernel<true> uses at k = 17).  A half that
//            wraps stays exact: the atomics on a word are serialised, the one that carries a half over 0xFFFF sees it in the
//            value it gets back and notes "+65536 for this bin" in a short list (a low half's carry into the high half is taken
//            back out, see smallk_after16); the notes are added to the vector after the histogram.  A full list adds its notes to
//            the vector directly.
//   k <= 7   4^k <= 16384 bins: plain 32-bit counters, non-returning atomics (a workgroup sees < 2^32 windows per launch).  The 32768
//            words hold 2^r copies of the histogram (r = min(5, 15 - 2k)), copy = lane mod 2^r in the LOW address bits: at k <= 5
//            the 32 lanes of an LDS access group then sit in 32 different banks whatever their ids are (4^k bins would otherwise
//            mean 64 lanes queueing on a handful of addresses); the copies are summed in the flush.
//
// One persistent workgroup of 1024 threads per CU (16 waves, as the scatter kernels have); tiles of 1023 chunks dealt round robin.
// Same counting semantics as everywhere else (kmer.py:234-317, :526-565; parse.py:133-136).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "kdb_scatter.hip.h"

namespace kdb {

constexpr int SMALLK_THREADS = 1024, SMALLK_GRID = 256;
constexpr int SMALLK_TILE_STRIDE = SMALLK_THREADS - 1, SMALLK_TILE_POS = SMALLK_TILE_STRIDE * 16;
constexpr int SMALLK_WORDS = 32768;                 // 128 KiB
constexpr uint32_t SMALLK_NOTES = 1024;
constexpr int SMALLK_LDS_MAX_K = 8;

struct SmallkNotes { uint32_t n; uint32_t e[SMALLK_NOTES]; };      // entry: bin | (1 << 16 if -65536 instead of +65536)

__device__ __forceinline__ void smallk_note(SmallkNotes &wl, unsigned long long *__restrict__ table, uint32_t bin, uint32_t negative)
{
    const uint32_t s = atomicAdd(&wl.n, 1u);
    if (s < SMALLK_NOTES) wl.e[s] = bin | (negative << 16);
    else __hip_atomic_fetch_add(&table[bin], negative ? 0ull - 65536ull : 65536ull, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);   // (the list is full: straight to the vector)
}

// what the returned word of an add of n (1 <= n <= 64) to bin v says: did the 16-bit field wrap?
__device__ __forceinline__ void smallk_after16(uint32_t *hist, SmallkNotes &wl, unsigned long long *__restrict__ table, uint32_t v, uint32_t n, uint32_t old)
{
    if (v >> 15) {
        if ((old >> 16) + n > 0xFFFFu) smallk_note(wl, table, v, 0u);
    } else if ((old & 0xFFFFu) + n > 0xFFFFu) {
        smallk_note(wl, table, v, 0u);
        if ((old >> 16) == 0xFFFFu) smallk_note(wl, table, v | 0x8000u, 0u);        // the carry wrapped the other half
        const uint32_t old2 = atomicSub(&hist[v & 0x7FFFu], 0x10000u);               // the carry does not belong there
        if ((old2 >> 16) == 0u) smallk_note(wl, table, v | 0x8000u, 1u);             // ... and taking it out un-wrapped it
    }
}

__device__ __noinline__ void smallk_after16_rare(uint32_t *hist, SmallkNotes &wl, unsigned long long *__restrict__ table, uint32_t v, uint32_t old)
{
    smallk_after16(hist, wl, table, v, 1u, old);
}

template <bool EXPAND, bool CANON, bool HALVES /* k = 8 */, bool RAGGED /* see scatter_bases_kernel */>
__global__ void __launch_bounds__(SMALLK_THREADS, 4)
count_smallk_kernel(const uint8_t *__restrict__ bases, uint64_t nbytes, uint32_t tile0, uint32_t ntiles, int k,
                    unsigned long long *__restrict__ table, DevCounters *ctr, RecStarts rs)
{
    constexpr int NID = 16;
    if ((batch_uniform_len(ctr) == 0u) != RAGGED) return;        // (the other variant counts this batch)
    // offsets that do not tile the buffer (lens_kernel ran before this kernel on the same stream): the job fails at the sync, and the walk
    // through such offsets for a tile's record starts need not end -- nothing is counted
    if (RAGGED && ctr->bad_layout) return;
    using Tile = ScTile<EXPAND, SMALLK_THREADS>;
    __shared__ Tile T[2];
    __shared__ uint32_t hist[SMALLK_WORDS];
    __shared__ SmallkNotes wl;
    const int j = threadIdx.x;
    const uint32_t G = sc_pin(gridDim.x);
    ntiles = sc_pin(ntiles);
    for (int i = j; i < SMALLK_WORDS; i += SMALLK_THREADS) hist[i] = 0;
    if (j == 0) { wl.n = 0; T[0].has_n[0] = 0; T[1].has_n[0] = 0; }
    if (EXPAND) __syncthreads();                     // (has_n is cleared before the first image is staged)
    const int canonical = CANON ? 1 : 0;
    const IdParams<uint32_t> idp(k, canonical);
    const WinOr winor(k);
    const uint64_t idmask = (1ull << (2 * k)) - 1ull;
    const uint32_t kmask = (1u << k) - 1u;
    const bool owner_of_windows = j < SMALLK_TILE_STRIDE;
    const int rlog = HALVES ? 0 : (15 - 2 * k < 5 ? 15 - 2 * k : 5);                 // log2(copies of the histogram), k <= 7
    const uint32_t copy = (uint32_t)j & ((1u << rlog) - 1u);
    const uint32_t ulen = batch_uniform_len(ctr);
    uint32_t x = 0, xstep = 0;
    if (ulen) {
        x = (uint32_t)((((uint64_t)tile0 + blockIdx.x) * (uint64_t)SMALLK_TILE_POS + 16ull * j) % ulen);
        xstep = (uint32_t)(((uint64_t)G * SMALLK_TILE_POS) % ulen);
    }
    unsigned long long emitted = 0;
    uint32_t stat_tot = 0;                           // bad residues | record-start marks met << 16 (< 4096 tiles per workgroup: scatter_max_tiles)
    int buf = sc_pin(0);

    const uint32_t my_byte = 16u * (uint32_t)j;
    auto fetch_tile = [&](uint64_t tile_no) -> ScChunk {
        const uint64_t first = tile_no * (uint64_t)(SMALLK_TILE_STRIDE * 16);
        if (first + (uint64_t)(SMALLK_THREADS * 16) <= nbytes) {
            ScChunk c;
            c.v = *reinterpret_cast<const uint4 *>(bases + first + my_byte);
            c.nexist = 0u;
            return c;
        }
        return sc_fetch(bases, nbytes, tile_no * SMALLK_TILE_STRIDE + (uint64_t)j);
    };
    ScChunk mine;
    mine.v = make_uint4(0, 0, 0, 0); mine.nexist = 0xFFFFu;
    if (blockIdx.x < ntiles) {
        mine = fetch_tile((uint64_t)tile0 + blockIdx.x);
        const uint32_t nb_ = sc_stage_chunk<EXPAND>(T[0], mine, j, true, ulen ? uniform_starts(x, ulen) : 0u, blockIdx.x + 1u,
                                                    (((uint64_t)tile0 + blockIdx.x) * SMALLK_TILE_STRIDE + (uint64_t)j) * 16ull, ctr, owner_of_windows);
        if (owner_of_windows) stat_tot += nb_;
        if (ulen) { x += xstep; if (x >= ulen) x -= ulen; }
        if (blockIdx.x + G < ntiles) mine = fetch_tile((uint64_t)tile0 + blockIdx.x + G);
    }
    __syncthreads();
    // a ragged batch: record starts come from the offsets (kdb_scatter.hip.h, RecStarts), ORed into an image after the barrier that ends its staging
    constexpr bool ragged = RAGGED;
    // (as in scatter_bases_kernel, round 5: first_rec is read per lane -- through an index the compiler cannot see through, or it scalarises the word behind an
    //  `s_waitcnt vmcnt(0)` in the middle of the staging --, and the offsets are requested a tile earlier than they are applied, behind the staging's own wait)
    auto first_rec_of = [&](uint64_t tile_no) -> uint32_t {
        uint32_t idx = (uint32_t)((tile_no * (uint64_t)SMALLK_TILE_POS) >> FIRST_REC_SHIFT);
        asm volatile("" : "+v"(idx));
        return rs.first_rec[idx];
    };
    uint32_t first_next = 0;
    StartProbe probe, probe2;
    probe.r = 0; probe.off = ~0ull; probe.beyond = ~0ull;
    probe2 = probe;
    if (ragged && blockIdx.x < ntiles) {
        const uint64_t P0 = ((uint64_t)tile0 + blockIdx.x) * (uint64_t)SMALLK_TILE_POS;
        starts_apply<SMALLK_THREADS>(T[0], rs, P0, starts_fetch<SMALLK_THREADS>(rs, rs.first_rec[P0 >> FIRST_REC_SHIFT], j));
        if (blockIdx.x + G < ntiles) probe2 = starts_fetch<SMALLK_THREADS>(rs, rs.first_rec[(((uint64_t)tile0 + blockIdx.x + G) * (uint64_t)SMALLK_TILE_POS) >> FIRST_REC_SHIFT], j);
        if (blockIdx.x + 2 * G < ntiles) first_next = first_rec_of((uint64_t)tile0 + blockIdx.x + 2ull * G);
        __syncthreads();
    }

    for (uint32_t t = blockIdx.x; t < ntiles; t += G) {
        const uint64_t tile = (uint64_t)tile0 + t;
        const Hood h = sc_load_hood<CANON>(T[buf], j < SMALLK_TILE_STRIDE ? j : 0);
        // (k = 1: a window is its one base -- nothing lies strictly inside it, and WinOr is made for k >= 2)
        const uint32_t bad16 = !owner_of_windows ? 0xFFFFu : k == 1 ? (h.V & 0xFFFFu) : windows_bad16(h, winor);
        uint32_t N32 = 0;
        if (EXPAND && owner_of_windows) N32 = (T[buf].nn[j] & 0xFFFFu) | (T[buf].nn[j + 1] << 16);
        if (EXPAND && T[buf].has_n[0] == t + 1u) {                       // (workgroup-uniform; a tile without an N: nothing of this runs)
            // the N-windows of this wave, decoded where they are found, queued in the wave's own lanes' slots of the idle image and dealt
            // out evenly: their 4 or 16 fills are LDS atomics like every other id (kdb_scatter.hip.h, "N expansion"); no other wave is involved
            const uint32_t lane = (uint32_t)j & 63u;
            const int wbase = j & ~63;
            NQueue Q{&T[buf ^ 1].fwd[wbase], &T[buf ^ 1].msk[wbase], 0u};
            auto count_fills = [&](const NWindow &w) {
                emitted += w.nfill;
#pragma unroll 1
                for (uint32_t f = 0; f < w.nfill; f++) {
                    const uint32_t id = (uint32_t)nwindow_fill<CANON>(w, f, k, idmask);
                    if (HALVES) {
                        const uint32_t old = atomicAdd(&hist[id & 0x7FFFu], 1u << ((id >> 11) & 16u));
                        smallk_after16(hist, wl, table, id, 1u, old);
                    } else {
                        atomicAdd(&hist[(id << rlog) | copy], 1u);
                    }
                }
            };
            const uint32_t nonly = (N32 && bad16) ? (k == 1 ? N32 & bad16 & 0xFFFFu : windows_nonly16(h, N32, bad16, winor)) : 0u;
            if (__ballot(nonly != 0u)) {                                 // (wave-uniform)
                uint32_t few = 0;                                        // windows with one or two N's (more: the work list)
#pragma unroll 1
                for (uint32_t m = nonly; m; m &= m - 1u) {
                    const int i = __builtin_ctz(m);
                    const uint32_t nwin = (N32 >> i) & kmask;
                    if (__builtin_popcount(nwin) <= 2) few |= 1u << i;
                    else expand_n_window(table, h.F(), i, k, canonical, idmask, nwin, &emitted, ctr);
                }
                // one entry per window here (lane | window << 6): the lane that takes it rebuilds the window from the tile image
                const uint32_t mine_n = (uint32_t)__builtin_popcount(few);
                const uint32_t incl = wave_incl_scan(mine_n), tot = (uint32_t)__builtin_amdgcn_readlane((int)incl, 63);
                uint32_t sl = incl - mine_n;
                Q.n = tot < NQ_ENTRIES ? tot : NQ_ENTRIES;
#pragma unroll 1
                for (uint32_t m = few; m; m &= m - 1u) {
                    const uint32_t i = (uint32_t)__builtin_ctz(m);
                    if (sl < NQ_ENTRIES) Q.at(sl) = lane | (i << 6); else count_fills(nwindow_decode(h.F(), (int)i, k, idmask, (N32 >> i) & kmask));
                    sl++;
                }
                __builtin_amdgcn_wave_barrier();
#pragma unroll 1
                for (uint32_t e = lane; e < Q.n; e += 64u) {
                    const uint32_t en = Q.at(e);
                    const int c = wbase + (int)(en & 63u), i = (int)((en >> 6) & 15u);
                    const uint64_t F = ((uint64_t)T[buf].fwd[c] << 32) | T[buf].fwd[c + 1];
                    count_fills(nwindow_decode(F, i, k, idmask, (((T[buf].nn[c] & 0xFFFFu) | (T[buf].nn[c + 1] << 16)) >> i) & kmask));
                }
                __builtin_amdgcn_wave_barrier();                         // (the queue is read before this wave stages its chunks over it)
            }
        }
        uint32_t pend = ~bad16 & 0xFFFFu;
        emitted += (unsigned long long)__builtin_popcount(pend);
        uint64_t same; uint32_t id0;
        if (wave_dominant(h.f0 >> (32 - 2 * k), &same, &id0)) {
            // degenerate stretch (poly-A/G, short-period repeats): lanes that share an id with >= 15 others add it once, together
#pragma unroll 1
            for (int u = 0; u < NID; u++) {
                const uint32_t idu = idp.id_any(h, u);
                const bool live = (pend >> u) & 1u;
                const uint64_t act = __ballot(live);
                if (!act) continue;
                const uint32_t lead = (uint32_t)__builtin_amdgcn_readlane((int)idu, __ffsll((unsigned long long)act) - 1);
                const bool same_id = live && idu == lead;
                const uint64_t grp = __ballot(same_id);
                if (__popcll(grp) >= 16 && same_id) {
                    pend &= ~(1u << u);
                    if (lane_rank_in(grp) == 0) {
                        const uint32_t n = (uint32_t)__popcll(grp);
                        if (HALVES) {
                            const uint32_t old = atomicAdd(&hist[idu & 0x7FFFu], (idu >> 15) ? n << 16 : n);
                            smallk_after16(hist, wl, table, idu, n, old);
                        } else {
                            atomicAdd(&hist[(idu << rlog) | copy], n);
                        }
                    }
                }
            }
        }
        // the ids, one LDS atomic each; k = 8: the returned words are looked at once all sixteen are on their way
        uint32_t ids[NID], got[NID], shs[NID];
#pragma unroll
        for (int u = 0; u < NID; u++) {
            if (HALVES) {
                // k = 8: the id TIMES FOUR, so that the byte address of its word is one AND.  Canonical: the forward word shifted two bits
                // short (two bits of the next base stay below the id: they can only decide between equal ids) against the reverse word
                // taken two bits to the left, min -- the same k-mer wins; those two bits never reach the address or the half's shift
                uint32_t id4;
                if (CANON) {
                    const uint32_t wf = u == 0 ? h.f0 : __builtin_amdgcn_alignbit(h.f0, h.f1, 32 - 2 * u);
                    const uint32_t wr4 = u == 0 ? h.r0 << 2 : __builtin_amdgcn_alignbit(h.r1, h.r0, 2 * u - 2);
                    const uint32_t f4 = wf >> 14, r4 = wr4 & 0x3FFFCu;
                    id4 = f4 < r4 ? f4 : r4;
                } else {
                    id4 = idp.id(h, u) << 2;
                }
                ids[u] = id4;
                shs[u] = (id4 >> 13) & 16u;                  // half id >> 15 of word id & 0x7FFF
                // no exec-mask region per id: a window that does not count adds 0 (its returned word can only raise a false alarm below)
                got[u] = atomicAdd(reinterpret_cast<uint32_t *>(reinterpret_cast<char *>(hist) + (id4 & 0x1FFFCu)), ((pend >> u) & 1u) << shs[u]);
            } else {
                uint32_t id;
                if (CANON) {
                    const uint32_t wf = u == 0 ? h.f0 : __builtin_amdgcn_alignbit(h.f0, h.f1, 32 - 2 * u);
                    const uint32_t wr = u == 0 ? h.r0 : __builtin_amdgcn_alignbit(h.r1, h.r0, 2 * u);
                    const uint32_t f = wf >> idp.p.fshift, r = wr & idp.p.mask;
                    id = f < r ? f : r;
                } else {
                    id = idp.id(h, u);
                }
                ids[u] = id; shs[u] = 0u;
                if ((pend >> u) & 1u) atomicAdd(&hist[(id << rlog) | copy], 1u);
            }
        }
        // while they fly: encode the next tile's chunk into the other image, request the chunk after it
        if (t + G < ntiles) {
            const uint32_t nb_ = sc_stage_chunk<EXPAND>(T[buf ^ 1], mine, j, true, ulen ? uniform_starts(x, ulen) : 0u, t + G + 1u,
                                                        ((tile + G) * SMALLK_TILE_STRIDE + (uint64_t)j) * 16ull, ctr, owner_of_windows);
            if (owner_of_windows) stat_tot += nb_;
            if (ulen) { x += xstep; if (x >= ulen) x -= ulen; }
            if (ragged) {
                probe = probe2;                                          // the staged tile's offsets (requested a tile ago)
                if (t + 2 * G < ntiles) {
                    probe2 = starts_fetch<SMALLK_THREADS>(rs, first_next, j);
                    if (t + 3 * G < ntiles) first_next = first_rec_of(tile + 3ull * G);
                }
            }
            if (t + 2 * G < ntiles) mine = fetch_tile(tile + 2ull * G);
        }
        if (HALVES) {
            // a field wrapped iff it stood at 0xFFFF -- rare; one test for all sixteen first: the word rotated so that the bin's field is
            // taken out (v_bfe_u32 at the half's shift), max3: 1.5 instructions per id (a window that was not counted can raise the alarm; the rare path looks at pend)
            uint32_t m = 0u;                           // max over the sixteen fields as they stood: 0xFFFF iff one of them wrapped
#pragma unroll
            for (int u = 0; u < NID; u++) {
                const uint32_t fld = __builtin_amdgcn_ubfe(got[u], shs[u], 16u);
                m = fld > m ? fld : m;
            }
            if (m == 0xFFFFu) {
#pragma unroll
                for (int u = 0; u < NID; u++)                  // (compile-time indices: ids[] and got[] stay plain registers)
                    if ((pend >> u) & 1u) smallk_after16_rare(hist, wl, table, ids[u] >> 2, got[u]);
            }
        }
        __syncthreads();                             // the next image is complete; this one may be overwritten in the next round
        if (ragged) {                                // (kernel-uniform) its record starts, and a second barrier before anybody reads them
            if (t + G < ntiles) starts_apply<SMALLK_THREADS>(T[buf ^ 1], rs, (tile + G) * (uint64_t)SMALLK_TILE_POS, probe);
            __syncthreads();
        }
        buf = sc_pin(buf ^ 1);
    }
    __syncthreads();

    // the histogram goes to the vector: contiguous 64-bit atomics, every workgroup starting somewhere else
    const uint32_t nbins = 1u << (2 * k);
    if (HALVES) {
        const uint32_t rot = (blockIdx.x * 4099u) & (SMALLK_WORDS - 1);
        for (uint32_t i = (uint32_t)j; i < (uint32_t)SMALLK_WORDS; i += SMALLK_THREADS) {
            const uint32_t w = (i + rot) & (SMALLK_WORDS - 1), c = hist[w];
            if (c & 0xFFFFu) __hip_atomic_fetch_add(&table[w], (unsigned long long)(c & 0xFFFFu), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            if (c >> 16) __hip_atomic_fetch_add(&table[w + SMALLK_WORDS], (unsigned long long)(c >> 16), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
        const uint32_t nw = wl.n < SMALLK_NOTES ? wl.n : SMALLK_NOTES;
        for (uint32_t q = (uint32_t)j; q < nw; q += SMALLK_THREADS) {
            const uint32_t e = wl.e[q];
            __hip_atomic_fetch_add(&table[e & 0xFFFFu], (e >> 16) ? 0ull - 65536ull : 65536ull, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
    } else {
        const uint32_t rot = (blockIdx.x * 67u) & (nbins - 1u);
        for (uint32_t i = (uint32_t)j; i < nbins; i += SMALLK_THREADS) {
            const uint32_t w = (i + rot) & (nbins - 1u);
            unsigned long long c = 0;
            for (uint32_t r = 0; r < (1u << rlog); r++) c += hist[(w << rlog) | r];
            if (c) __hip_atomic_fetch_add(&table[w], c, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
    }
    const unsigned long long we = wave_sum(emitted), wb = wave_sum((unsigned long long)(stat_tot & 0xFFFFu)), wm = wave_sum((unsigned long long)(stat_tot >> 16));
    if ((j & 63) == 0) {
        if (we) __hip_atomic_fetch_add(&ctr->total_kmers, we, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if (wb) __hip_atomic_fetch_add(&ctr->n_bad, wb, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if (wm) __hip_atomic_fetch_add(&ctr->marks_seen, wm, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
}

// host: k <= 8.  returns 0 ok, 1 error (partition_error())
inline int smallk_lds_count(hipStream_t stream, const uint8_t *d_bases, size_t nbytes, const RecStarts &rs, int k, int canonical, int n_expand, int grid_opt,
                            unsigned long long *d_table, DevCounters *d_ctr, ProfHook &prof)
{
    const uint64_t ntiles_all = ((nbytes + 15) / 16 + SMALLK_TILE_STRIDE - 1) / SMALLK_TILE_STRIDE;
    const uint32_t Gmax = grid_opt > 0 ? (uint32_t)grid_opt : (uint32_t)SMALLK_GRID;
    const uint64_t max_tiles = scatter_max_tiles(Gmax, SMALLK_TILE_POS);
    prof.begin(KDB_KERNEL_COUNT);
    for (uint64_t t0 = 0; t0 < ntiles_all; t0 += max_tiles) {
        const uint32_t nt = (uint32_t)((ntiles_all - t0) < max_tiles ? (ntiles_all - t0) : max_tiles);
        const uint32_t G = nt < Gmax ? nt : Gmax;
#define KDB_LAUNCH_SMALLK1(E, CN, HV, RAG)                                                                                       \
    hipLaunchKernelGGL((count_smallk_kernel<E, CN, HV, RAG>), dim3(G), dim3(SMALLK_THREADS), 0, stream, d_bases, (uint64_t)nbytes, \
                       (uint32_t)t0, nt, k, d_table, d_ctr, rs)
#define KDB_LAUNCH_SMALLK(E, CN, HV) do { KDB_LAUNCH_SMALLK1(E, CN, HV, false); KDB_LAUNCH_SMALLK1(E, CN, HV, true); } while (0)
#define KDB_LAUNCH_SMALLK_MODES(HV)                                                                                              \
    do {                                                                                                                         \
        if (n_expand) { if (canonical) KDB_LAUNCH_SMALLK(true, true, HV); else KDB_LAUNCH_SMALLK(true, false, HV); }             \
        else          { if (canonical) KDB_LAUNCH_SMALLK(false, true, HV); else KDB_LAUNCH_SMALLK(false, false, HV); }           \
    } while (0)
        if (k == SMALLK_LDS_MAX_K) KDB_LAUNCH_SMALLK_MODES(true); else KDB_LAUNCH_SMALLK_MODES(false);
#undef KDB_LAUNCH_SMALLK_MODES
#undef KDB_LAUNCH_SMALLK
#undef KDB_LAUNCH_SMALLK1
    }
    prof.end();
    if (hipGetLastError() != hipSuccess) { partition_error_ref() = "count_smallk_kernel failed to launch"; return 1; }
    return 0;
}

}  // namespace kdb
