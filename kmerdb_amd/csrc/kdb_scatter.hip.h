// kdb_scatter.hip.h -- the "paged scatter" paths of the engine (gfx950): one pass over the input per radix level,
// no sizing pass, no per-tile counts, no staging order.
//
// A radix partition with exact slices (round 1 of this engine) needs to know, before it scatters, how many ids of
// every tile go to every bucket (a sizing pass + scans), and it pays four LDS operations per id (cursor atomic, staged
// write, staged read, delta lookup).  Here every bucket has a small ring in LDS that persists across the tiles of a
// persistent workgroup ("software write-combining"):
//
//   place   slot = returning LDS atomic on the ring's word (base << 16 | count); the element goes to its slot: two LDS
//           operations per id.  A ring that is full refuses the element (the lane keeps it and tries again after the
//           flush: skewed data costs extra rounds, never correctness).
//   flush   after a barrier, the thread that owns a ring writes its complete lines to HBM -- always whole, aligned
//           lines: 128-byte pieces, eight lanes each, since round 5 (the memory system takes random 64-byte writes at
//           3.4-4.6 TB/s and 128-byte ones at 5.3: ElemFmt<u16w / u24w / u32w>, one workgroup of 1024 threads per CU);
//           64-byte lines, four lanes each, in the forms of rounds 2-4 (two workgroups of 512 threads per CU; k = 13)
//           -- into PAGES of 1 KiB that belong to that ring alone.  A workgroup takes page
//           numbers from a private arithmetic sequence (w, w + G, w + 2G, ...), whose length is bounded by the
//           number of ids the workgroup can emit: no global atomics, no over-provisioning guess, no overflow path.
//   tags    when a page is closed its tag (bucket << 12 | elements) is written; pages_sort_* turn the tags into one
//           page list per bucket (a counting sort over ~10^6 pages: microseconds), and page_hist_kernel builds the
//           32768-bin LDS histogram of a bucket from its pages.
//
// Same counting semantics as everywhere else (kmer.py:234-317, :526-565; parse.py:133-136): windows with N in EXPAND
// mode go to the vector through expand_n_window; degenerate stretches (poly-A/G, microsatellites: >= 16 lanes of a
// wave with one id) are collected in a small (id, count) table per workgroup and added to the vector at the end.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "kdb_kernels.hip.h"
#include "kdb_hist.hip.h"

namespace kdb {

constexpr int SC_THREADS = 512;
constexpr int SC_TILE_CHUNKS = 512;                      // one 16-base chunk per thread
constexpr int SC_LINE_BYTES = 64;
constexpr int SC_PAGE_LINES = 16;
constexpr int SC_PAGE_BYTES = SC_LINE_BYTES * SC_PAGE_LINES;   // 1 KiB: 512 u16 / 256 u32 elements
constexpr int SC_TAG_SHIFT = 12;                         // tag = bucket << 12 | elements in the page
constexpr uint32_t SC_NO_PAGE = 0xFFFFFFFFu;
constexpr int SC_GRID = 512;                             // persistent workgroups: two per CU
constexpr int SC_HOT = 64;                               // entries of a workgroup's table of degenerate ids
// id bits below the bucket field (see below).  One level (k <= 12, 512 buckets at most): 9 -- 512 consecutive bins = 4 KiB runs,
// and the bucket field is still far enough from the leading bits that canonical ids fill the buckets evenly (+-4 %; lo = 12:
// +-14 %, so some buckets need two histogram slices and flush with atomics: page_hist 0.62 -> 0.73 ms at k = 12).  Two levels
// (k >= 13): 12 -- 32 KiB runs; measured k = 15: 5.05 (lo = 6) / 4.99 / 4.94 (12) / 4.96 ms, k = 17: 8.15 / 7.94 / 7.91 / 7.87 ms
// (profiles/r03/lo_bits_sweep.md).
constexpr int SC_LO_BITS_ONE_LEVEL = 9, SC_LO_BITS_TWO_LEVEL = 12;
constexpr int SC_LO_BITS_MAX = 15;                       // all bin bits below the bucket field = buckets from the leading id bits

// Which id bits select the bucket.  Canonical ids (min of the two strands) crowd the LOW end of the id space, so the
// leading bits make buckets of very different sizes (2 : 1 : ... : 0); a ring must absorb the arrivals of one round,
// so uneven buckets mean refused elements and extra rounds.  The bucket is therefore taken from the bits just above
// lowest `lo` (engine option sc_lo_bits) -- as good as uniform in either strand mode once a few leading bits are left
// above them -- and the 15 histogram bits of a bucket are the LEADING bits plus the lowest `lo`:
//     id = [ hi : 15 - lo ][ bucket(s) : 2k - 15 ][ lo ],   bin = hi << lo | low bits.
// A bucket's bins are then 2^hi runs of 2^lo consecutive counters in the vector.  Round 2 ran with lo = 6 (512 runs of
// 512 bytes, 2^(2k - 9) counters apart): at k = 17 every run of a bucket lies in another 128 MiB of the vector, each
// needs its own address translation, and the histogram pass ran at 0.28 of the HBM peak.  lo = 12: 8 runs of 32 KiB.

// Tile image: the forward 2-bit word and the masks of every 16-base chunk (the reverse-strand word is derived from
// the forward one when the hood is loaded: v_bfrev_b32, swap the bits of each pair, not).  A tile is SC_TILE_CHUNKS
// chunks of which the first SC_TILE_STRIDE own windows; the last one is only the right-hand neighbour of chunk 510,
// so no thread stages a second ("halo") chunk and tiles advance by 511 chunks.
constexpr int SC_TILE_STRIDE = SC_TILE_CHUNKS - 1;
constexpr int SC_TILE_POS = SC_TILE_STRIDE * 16;         // window start positions per tile (8176)
// (CHUNKS = threads of the workgroup: 512, or 1024 for the one-level kernel of k = 13 -- 1023 chunks of windows, 16368 positions)
constexpr uint32_t SC_NPOS_MAX = 256;                    // (what the LDS has room for next to two workgroups' rings and images: 3 % N)
template <bool EXPAND, int CHUNKS = SC_TILE_CHUNKS>
struct ScTile {
    uint32_t fwd[CHUNKS];
    uint32_t msk[CHUNKS];                                // inv | st << 16
    uint32_t nn[EXPAND ? CHUNKS : 1];
    uint32_t has_n[1];                                   // EXPAND: == the image's generation (tile number + 1) iff a chunk of it holds an N
    // EXPAND, scatter kernels: the positions of the image's N's (in the tile's coordinates), listed while the image is staged; ncnt
    // counts them all, the list holds the first SC_NPOS_MAX (a tile with more is "dense": every wave looks through its own windows).
    // Cleared by thread 0 once the tile's N-windows have been dealt with (after the first barrier of the tile's placement).
    uint32_t ncnt;
    uint16_t npos[EXPAND ? SC_NPOS_MAX : 2];
};

__device__ __forceinline__ uint32_t rc_word(uint32_t f)   // forward word of a chunk -> its reverse-strand word
{
    const uint32_t y = __builtin_bitreverse32(f);
    return ~bfi(0x55555555u, y >> 1, y << 1);
}

template <bool CANON, typename TILE>
__device__ __forceinline__ Hood sc_load_hood(const TILE &L, int c)
{
    Hood h;
    h.f0 = L.fwd[c]; h.f1 = L.fwd[c + 1];
    if (CANON) { h.r0 = rc_word(h.f0); h.r1 = rc_word(h.f1); } else { h.r0 = 0; h.r1 = 0; }
    const uint32_t m0 = L.msk[c], m1 = L.msk[c + 1];
    h.V = (m0 & 0xFFFFu) | (m1 << 16);
    h.S = (m0 >> 16) | (m1 & 0xFFFF0000u);
    return h;
}

// 16 bytes that are read exactly once (residues, pages): -DKDB_NT_LOADS reads them non-temporal (experiment: tools/experiments/exp_r04_nt.sh)
__device__ __forceinline__ uint4 load_once16(const void *p)
{
#ifdef KDB_NT_LOADS
    typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
    const u32x4 v = __builtin_nontemporal_load(reinterpret_cast<const u32x4 *>(p));
    return make_uint4(v.x, v.y, v.z, v.w);
#else
    return *reinterpret_cast<const uint4 *>(p);
#endif
}

// a 16-byte chunk on its way from HBM to the tile image (loaded one tile ahead); nexist: bit b = byte b lies past the end of the buffer
struct ScChunk { uint4 v; uint32_t nexist; };

__device__ __forceinline__ ScChunk sc_fetch(const uint8_t *__restrict__ bases, uint64_t nbytes, uint64_t g)
{
    ScChunk c;
    if ((g + 1) * 16ull <= nbytes) {
        c.v = *reinterpret_cast<const uint4 *>(bases + g * 16ull);
        c.nexist = 0u;
    } else {
        uint32_t w[4];
        const int nv = load_chunk(bases, nbytes, g, w);
        c.v = make_uint4(w[0], w[1], w[2], w[3]);
        c.nexist = 0xFFFFu & ~((1u << nv) - 1u);
    }
    return c;
}

// Record starts of a ragged batch, straight from the offsets (lens_kernel: first_rec).  A tile of THREADS chunks: thread j looks at
// record first_rec[tile start >> 12] + j -- the loads go out while the tile is staged, the start bits are ORed into the staged
// image's masks after the barrier that ends the staging (the residues themselves are never written).
struct RecStarts { const uint64_t *offs; const uint32_t *first_rec; uint32_t nreads; uint32_t skip_first /* record 0 continues a tiled long record: no start */; };
struct StartProbe { uint32_t r; uint64_t off; uint64_t beyond /* offset of the record behind this round's last (uniform) */; };

template <int THREADS>
__device__ __forceinline__ StartProbe starts_fetch(const RecStarts &rs, uint32_t first, int j)
{
    StartProbe p;
    p.r = first + (uint32_t)j;
    p.off = p.r < rs.nreads ? rs.offs[p.r] : ~0ull;
    const uint32_t nx = first + (uint32_t)THREADS;
    p.beyond = nx < rs.nreads ? rs.offs[nx] : ~0ull;
    return p;
}

template <int THREADS, typename TILE>
__device__ __forceinline__ void starts_apply(TILE &img, const RecStarts &rs, uint64_t P0 /* byte position of the tile */, StartProbe p)
{
    constexpr uint64_t SPAN = (uint64_t)THREADS * 16ull;
    auto mark = [&](const StartProbe &q) {
        const uint64_t d = q.off - P0;                                   // (wraps for a record that starts before the tile)
        if (d < SPAN && !(q.r == 0u && rs.skip_first)) atomicOr(&img.msk[d >> 4], 0x10000u << (d & 15u));
    };
    // The first round stands OUTSIDE the loop (round 5): inside it, the loop header's `s_waitcnt vmcnt(0)` -- there for the offsets a further round fetches -- also made
    // the first round wait for every load the wave had in flight, the residues of the tile after next among them.
    mark(p);
    // (uniform) the usual case: no record behind this round starts inside the tile.  NOT `beyond - P0 >= SPAN`: the walk begins at the
    // record that holds the 4 KiB boundary below the tile, and with reads of a few bases a whole round can end before the tile begins --
    // the difference wrapped and the tile got no record starts at all (k = 2, reads of 2..6 bases: found by tests/fuzz_gpu.py, round 4)
    while (p.beyond < P0 + SPAN) {
        p = starts_fetch<THREADS>(rs, p.r - (uint32_t)threadIdx.x + (uint32_t)THREADS, (int)threadIdx.x);     // reads shorter than ~24 bases
        mark(p);
    }
}

// encode a fetched chunk into slot c of the tile image; returns the number of bad residues in it (low half) and, in a
// ragged batch, the number of record-start marks it carries (high half).
// What encode16 does, ordered for the common case: the is-N / neither-ACGT-nor-N masks are only worked out for a chunk
// that holds a residue outside ACGT at all, and start marks are only gathered when the batch has them.
// pos0 = byte position of the chunk in the batch; DROP mode: residues that are neither ACGT nor N go to the batch's list of suspects
// (kdb_kernels.hip.h, defer_suspects16: an IUPAC code that every window of its record shields with an N is no error)
template <bool EXPAND, bool NLIST = false /* list the N's in L.npos (scatter kernels) */, typename TILE>
__device__ __forceinline__ uint32_t sc_stage_chunk(TILE &L, const ScChunk &ch, int c, bool uniform, uint32_t ustarts, uint32_t gen, uint64_t pos0, DevCounters *ctr,
                                                   bool owner = true /* false: the halo chunk, staged again as chunk 0 of the next tile -- its suspects are listed there, once */)
{
    const uint32_t w[4] = {ch.v.x, ch.v.y, ch.v.z, ch.v.w};
    uint32_t fwd = 0, back[4];
#pragma unroll
    for (int q = 0; q < 4; q++) {
        const uint32_t t = ((w[q] ^ (w[q] >> 1)) >> 1) & 0x03030303u;          // bits 1^2 and 2^3 of every byte: A0 C1 G2 T3 (kmer.py:44-49)
        fwd |= __builtin_amdgcn_udot4(t, 0x01041040u, 0u, false) << (24 - 8 * q);
        back[q] = w[q] ^ __builtin_amdgcn_perm(0u, 0x54474341u /* "ACGT" */, t);  // zero byte: the byte is the letter its code stands for (and bit 7 is clear)
    }
    // the usual chunk of a batch without start marks: sixteen letters out of ACGT, all inside the buffer -- nothing to gather
    if (uniform && ((back[0] | back[1] | back[2] | back[3]) | ch.nexist) == 0u) {
        L.fwd[c] = fwd; L.msk[c] = ustarts << 16;
        if (EXPAND) L.nn[c] = 0u;
        return 0u;
    }
    uint32_t notacgt[4];
#pragma unroll
    for (int q = 0; q < 4; q++) notacgt[q] = nonzero_bytes(back[q] & 0x7F7F7F7Fu);
    const uint32_t exist = 0xFFFFu & ~ch.nexist;
    const uint32_t inv = gather16(notacgt[0], notacgt[1], notacgt[2], notacgt[3]) | ch.nexist;
    uint32_t st = ustarts;
    uint32_t nbad = 0, nn = 0, nmark = 0, hi16 = 0;
    const uint32_t hib = (w[0] | w[1] | w[2] | w[3]) & 0x80808080u;
    if (!uniform) {                                                        // (wave-uniform)
        st = gather16(w[0] & 0x80808080u, w[1] & 0x80808080u, w[2] & 0x80808080u, w[3] & 0x80808080u) & exist;
        nmark = (uint32_t)__builtin_popcount(st);
    } else if (hib) {
        // no marks in a uniform batch: a byte with bit 7 set is no residue (0xC1 is not 'A'; kmer.py:170 raises)
        hi16 = gather16(w[0] & 0x80808080u, w[1] & 0x80808080u, w[2] & 0x80808080u, w[3] & 0x80808080u) & exist;
        nbad = (uint32_t)__builtin_popcount(hi16);
    }
    if (inv & exist) {                                                     // some residue is not ACGT: N, or an error
        uint32_t b4[4];
#pragma unroll
        for (int q = 0; q < 4; q++) b4[q] = notacgt[q] & nonzero_bytes((w[q] & 0x7F7F7F7Fu) ^ 0x4E4E4E4Eu);
        const uint32_t bad = gather16(b4[0], b4[1], b4[2], b4[3]) & exist;
        nn = inv & ~bad & exist;
        uint32_t errs = bad & ~hi16;                                       // (a byte counts once: at most 16 per chunk, what stat_tot's 16-bit halves rely on)
        nbad += (!EXPAND && errs) ? (owner ? defer_suspects16(errs, pos0, ctr) : 0u) : (uint32_t)__builtin_popcount(errs);
    }
    L.fwd[c] = fwd; L.msk[c] = inv | ((st & exist) << 16);
    if (EXPAND) { L.nn[c] = nn; if (nn) L.has_n[0] = gen; }             // (every lane that writes it writes the same value)
    if (EXPAND && NLIST && nn) {
        uint32_t sl = atomicAdd(&L.ncnt, (uint32_t)__builtin_popcount(nn));
#pragma unroll 1
        for (uint32_t m = nn; m; m &= m - 1u) { if (sl < SC_NPOS_MAX) L.npos[sl] = (uint16_t)(16u * (uint32_t)c + (uint32_t)__builtin_ctz(m)); sl++; }
    }
    return nbad | (nmark << 16);
}

// Diagnostic build only (-DKDB_SC_PROF; tools/sc_phases.sh): per-phase shader cycles of scatter_bases_kernel, summed
// over all waves.  In the real build no stamp executes.
#ifdef KDB_SC_PROF
__device__ unsigned long long g_sc_prof[32];       // [0..8]: scatter_bases_kernel, [16..24]: scatter_ids_kernel
__device__ int g_sc_ablate;          // 1: no HBM line stores of the flush; 2: no high-byte half-lines (u24); 4: no page tags; 8: no drain at the end of the kernel
#define SC_ABLATE(bit) (g_sc_ablate & (bit))
#define SC_STAMP_INIT unsigned long long sc_last, sc_acc[8] = {0, 0, 0, 0, 0, 0, 0, 0}; asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(sc_last) :: "memory")
#define SC_STAMP(i) do { unsigned long long sc_t; asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(sc_t) :: "memory"); sc_acc[i] += sc_t - sc_last; sc_last = sc_t; } while (0)
#define SC_STAMP_FN [&](int sc_i) { SC_STAMP(sc_i); }
#define SC_STAMP_END_AT(o) do { if ((threadIdx.x & 63) == 0) for (int q = 0; q < 8; q++) atomicAdd(&g_sc_prof[(o) + q], sc_acc[q]); if (threadIdx.x == 0) atomicAdd(&g_sc_prof[(o) + 8], 1ull); } while (0)
#define SC_STAMP_END SC_STAMP_END_AT(0)
// when every workgroup of the last launch started and ended (s_memrealtime: one 100 MHz clock for the whole device; s_memtime is per XCD)
__device__ unsigned long long g_sc_wg[2][2][1024];
#define SC_WG_CLOCK(kern, what) do { if (threadIdx.x == 0 && blockIdx.x < 1024) g_sc_wg[kern][what][blockIdx.x] = wall_clock64(); } while (0)
#else
#define SC_WG_CLOCK(kern, what)
#define SC_ABLATE(bit) 0
#define SC_STAMP_INIT
#define SC_STAMP(i)
#define SC_STAMP_FN NoStamp()
#define SC_STAMP_END_AT(o)
#define SC_STAMP_END
#endif

// ---------------------------------------------------------------------------------
// rings
// ---------------------------------------------------------------------------------
// Element formats.  A page is SC_PAGE_LINES lines; a line holds LINE_ELEMS elements:
//   uint16_t  bins (k <= 12, level 2):            32 per 64-byte line, 1 KiB pages of 512
//   uint32_t  remainders of level 1 at k = 17:      16 per line, 1 KiB pages of 256 (25 bits used)
//   u24       remainders of level 1 at k <= 16 (24 bits): kept as a u16 array (bits 0..15) and a u8 array (bits 16..23), in LDS
//             and in the page alike: 16 lines of 64 B, then 16 half-lines of 32 B = 1.5 KiB pages of 512 -- 3 bytes per
//             k-mer written by level 1 and read by level 2 instead of 4.  The half-lines leave the CU two at a time, as
//             whole 64-byte lines (a 32-byte write costs the HBM a whole sector): the u8 array of a ring holds twice as many
//             elements as its u16 array, so the high bytes of an even line are still there when the odd line after it goes out
//   u16w      the same 1 KiB pages of 512 uint16_t bins, written in 128-BYTE pieces (8 "lines" of 64 elements per page; rings of >= 128 elements).
//             The memory system takes random 64-byte writes at 3.4-4.6 TB/s and random 128-byte writes at 5.3 (no compute at all:
//             tools/ubench_hbm_pattern.hip, profiles/r05/ubench_hbm_pattern.txt) -- the scatter kernels ran within 4-15 % of the former.
struct u24 {};
struct u16w {};
template <typename ELEM> struct ElemFmt;
template <> struct ElemFmt<uint16_t> { using lo_t = uint16_t; static constexpr bool HI = false; static constexpr int LINE_ELEMS = 32, PAGE_BYTES = 1024, LINE_BYTES = 64, PAGE_LINES = 16; };
template <> struct ElemFmt<uint32_t> { using lo_t = uint32_t; static constexpr bool HI = false; static constexpr int LINE_ELEMS = 16, PAGE_BYTES = 1024, LINE_BYTES = 64, PAGE_LINES = 16; };
template <> struct ElemFmt<u24>      { using lo_t = uint16_t; static constexpr bool HI = true;  static constexpr int LINE_ELEMS = 32, PAGE_BYTES = 1536, LINE_BYTES = 64, PAGE_LINES = 16; };
template <> struct ElemFmt<u16w>     { using lo_t = uint16_t; static constexpr bool HI = false; static constexpr int LINE_ELEMS = 64, PAGE_BYTES = 1024, LINE_BYTES = 128, PAGE_LINES = 8; };
// (the same for level 1: u24w = u24 pages whose u16 halves leave in 128-byte pieces and whose high bytes leave 128 at a time -- two pieces' worth;
//  u32w = the 25-bit remainders of k = 17, 32 to a piece)
struct u24w {};
struct u32w {};
template <> struct ElemFmt<u24w>     { using lo_t = uint16_t; static constexpr bool HI = true;  static constexpr int LINE_ELEMS = 64, PAGE_BYTES = 1536, LINE_BYTES = 128, PAGE_LINES = 8; };
template <> struct ElemFmt<u32w>     { using lo_t = uint32_t; static constexpr bool HI = false; static constexpr int LINE_ELEMS = 32, PAGE_BYTES = 1024, LINE_BYTES = 128, PAGE_LINES = 8; };
constexpr int SC_HI_OFFSET = SC_PAGE_LINES * SC_LINE_BYTES;          // u24 pages: where the high bytes start

template <typename ELEM, int RINGS, int C>
struct alignas(16) RingLds {
    using F = ElemFmt<ELEM>;
    using lo_t = typename F::lo_t;
    static_assert((C & (C - 1)) == 0 && C >= 2 * F::LINE_ELEMS, "ring = at least two lines, power of two");
    // A ring's word: fill << 16 | tail.  fill = elements in the ring (plus the requests it refused since the last flush);
    // tail = where the next element goes, as a BYTE offset into the ring's lo_t array that wraps by masking (it keeps
    // counting past the ring's size; the owner brings it back below C * SZ whenever it flushes a line).  One request =
    // one returning atomic add of INC: the slot is `old & POS_MASK` -- one instruction -- and the ring was full iff
    // `old & FULL_MASK`.  The oldest element sits at (tail / SZ - fill) mod C: refused requests advance both alike.
    static constexpr uint32_t SZ = (uint32_t)sizeof(lo_t);
    static constexpr uint32_t INC = (1u << 16) | SZ;
    static constexpr uint32_t POS_MASK = (uint32_t)C * SZ - 1u;
    static constexpr uint32_t FULL_MASK = 0xFFFF0000u & ~((uint32_t)(C - 1) << 16);
    // positions are kept modulo WRAP elements: the size of the u8 array of a u24 ring (2 C), else of the ring itself
    static constexpr uint32_t WRAP = F::HI ? 2u * (uint32_t)C : (uint32_t)C;
    uint32_t word[RINGS];
    lo_t ring[RINGS * C];
    uint8_t hi[F::HI ? RINGS * WRAP : 4]; // (u24: bits 16..23 of the element at position p mod 2 C; the low half is at p mod C in ring[])
    uint32_t pg_count;                    // pages this workgroup has taken so far
    uint32_t retry[2];                    // "some lane still holds an element" flags of alternating rounds
    // ids that >= 16 lanes of a wave share (poly-A/G reads, microsatellites) never enter a ring: a small direct-mapped
    // table of (id, count) per workgroup absorbs them, and goes to the vector once, at the end of the kernel
    unsigned long long hot_tag[SC_HOT];   // 0 = free, else 1 << 40 | id
    uint32_t hot_cnt[SC_HOT];

    __device__ __forceinline__ void put(uint32_t woff /* ring * 4 */, uint32_t got /* the ring's word before the request */, uint32_t el)
    {
        const uint32_t off = (woff * (uint32_t)(C * sizeof(lo_t) / 4)) | (got & POS_MASK);          // byte offset into ring[]
        *reinterpret_cast<lo_t *>(reinterpret_cast<char *>(ring) + off) = (lo_t)el;
        if (F::HI) *(reinterpret_cast<uint8_t *>(hi) + ((woff * (WRAP / 4u)) | ((got & (WRAP * SZ - 1u)) / SZ))) = (uint8_t)(el >> 16);
    }
    // what a ring's word says: elements to take (clamped: refused requests counted too), position of the oldest one (mod WRAP)
    static __device__ __forceinline__ void decode(uint32_t wd, uint32_t *r, uint32_t *head)
    {
        const uint32_t fill = wd >> 16;
        *head = ((wd & 0xFFFFu) / SZ - fill) & (WRAP - 1u);
        *r = fill > (uint32_t)C ? (uint32_t)C : fill;
    }
    static __device__ __forceinline__ uint32_t encode(uint32_t head, uint32_t fill)
    {
        return (fill << 16) | (((head + fill) & (WRAP - 1u)) * SZ);
    }
    // a line that starts at position p (mod WRAP) of ring b: index of its first element in ring[]; u24: and, in the upper
    // half, the index in hi[] of the PAIR of lines it belongs to (lines are 32 elements, pages begin at multiples of 2 C)
    static __device__ __forceinline__ uint32_t line_elem(uint32_t b, uint32_t p)
    {
        const uint32_t lo = b * (uint32_t)C + (p & (uint32_t)(C - 1));
        return F::HI ? lo | ((b * WRAP + (p & (WRAP - 1u) & ~(2u * (uint32_t)F::LINE_ELEMS - 1u))) << 16) : lo;
    }
};

// where line `ln` of page `pg` lives (and, for u24, its half-line of high bytes)
template <typename ELEM>
__device__ __forceinline__ uint8_t *page_line(uint8_t *pages, uint32_t line /* pg * SC_PAGE_LINES + ln */)
{
    using F = ElemFmt<ELEM>;
    if (F::PAGE_BYTES == F::PAGE_LINES * F::LINE_BYTES) return pages + (size_t)line * F::LINE_BYTES;      // pages of lines only: line n at n * 64 (u16w: n * 128)
    return pages + (size_t)(line / F::PAGE_LINES) * F::PAGE_BYTES + (size_t)(line % F::PAGE_LINES) * F::LINE_BYTES;
}
template <typename ELEM>
__device__ __forceinline__ uint8_t *page_line_hi(uint8_t *pages, uint32_t line)
{
    using F = ElemFmt<ELEM>;
    return pages + (size_t)(line / F::PAGE_LINES) * F::PAGE_BYTES + SC_HI_OFFSET + (size_t)(line % F::PAGE_LINES) * (F::LINE_BYTES / 2);
}

struct ScOut {
    uint8_t *pages;                       // page p at pages + p * ElemFmt<ELEM>::PAGE_BYTES
    uint32_t *tag;                        // one per page, preset to SC_NO_PAGE
    uint32_t wg_pages;                    // page numbers of workgroup w: w + p * gridDim.x, p < wg_pages ...
    const uint32_t *wg_range;             // ... or, if not null, wg_range[w] + p, p < wg_range[w + 1] - wg_range[w]  (level 2: needs differ per workgroup)
    uint32_t contig;                      // (wg_range null) 1: page numbers w * wg_pages + p: a workgroup's pages lie together
    uint32_t wg_base;                     // filled in by the kernel at its start: the workgroup's first page number (wg_range[w] / w * wg_pages / w); with wg_range, wg_pages = the length of the range
    uint32_t grid;                        // filled in by the kernel at its start: gridDim.x
    uint32_t extra_elems;                 // EXPAND: fills of N-windows the host sized the workgroups' page sequences for, on top of one id per window
                                          // position (scatter_wg_pages); a placement round of fills that would leave too few pages for the ids still
                                          // to come adds its fills to the vector directly
};

// A value that the compiler must keep in a scalar register: kernel arguments and gridDim.x are loads from the kernarg / dispatch
// segments, which it otherwise re-issues wherever it runs short of SGPRs -- and the s_waitcnt lgkmcnt(0) behind such a load also
// waits for every LDS operation of the wave.  One of them sat between the slot requests and the work meant to overlap them.
__device__ __forceinline__ uint32_t sc_pin(uint32_t v) { asm volatile("" : "+s"(v)); return v; }

// a workgroup's page range and the grid size, read once (ring_next_line runs inside the flush: no loads there)
__device__ __forceinline__ ScOut sc_out_of_workgroup(ScOut o)
{
    if (o.wg_range) {
        o.wg_base = o.wg_range[blockIdx.x];
        o.wg_pages = o.wg_range[blockIdx.x + 1] - o.wg_base;
    } else {
        o.wg_base = o.contig ? blockIdx.x * o.wg_pages : blockIdx.x;     // (once: the product was a 16-cycle v_mul_lo_u32 in every page turn)
    }
    o.wg_base = sc_pin(o.wg_base); o.wg_pages = sc_pin(o.wg_pages); o.contig = sc_pin(o.contig);
    o.grid = sc_pin(gridDim.x);
    return o;
}

// the thread that owns ring `b`: its current page and the lines written into it
struct RingOwner {
    uint32_t pg = SC_NO_PAGE, ln = 0;
};

// the next 64-byte line of ring b's page sequence, as a line number (line n lives at pages + n * 64)
template <typename ELEM, int RINGS, int C>
__device__ __forceinline__ uint32_t ring_next_line(RingLds<ELEM, RINGS, C> &R, const ScOut &o, RingOwner &w, uint32_t bucket, DevCounters *ctr)
{
    constexpr uint32_t LINE_ELEMS = ElemFmt<ELEM>::LINE_ELEMS, PAGE_LINES = ElemFmt<ELEM>::PAGE_LINES;
    if (w.pg == SC_NO_PAGE || w.ln == PAGE_LINES) {
        const bool tag_when_taken = o.wg_range == nullptr;               // (scatter_bases_kernel; see below)
        if (w.pg != SC_NO_PAGE && !tag_when_taken && !SC_ABLATE(4)) o.tag[w.pg] = (bucket << SC_TAG_SHIFT) | (PAGE_LINES * LINE_ELEMS);
        uint32_t p = atomicAdd(&R.pg_count, 1u);
        const uint32_t cap = o.wg_pages;
        if (p >= cap) {                   // cannot happen (the sequence is sized for every id the workgroup can emit); never write out of bounds
            __hip_atomic_fetch_add(&ctr->internal_err, 1ull, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            p = cap - 1;
        }
        w.pg = (o.wg_range || o.contig) ? o.wg_base + p : p * o.grid + o.wg_base;
        w.ln = 0;
        // scatter_bases_kernel writes the tag when the page is TAKEN, as if it were going to be filled (a ring leaves a page only
        // when it is full; ring_drain corrects the last one): the rings that turn a page in the same flush take consecutive page
        // numbers, so their 4-byte tags fall into the same lines and leave the L2 together -- written on completion, the tags of
        // one moment belong to pages taken at different times, all over the array.  k = 12 scatter -1 %, level 1 at k = 15 / 17
        // -0.5 %; level 2 (scatter_ids_kernel: page ranges planned per workgroup) was 2 % slower with it at k = 17 and keeps
        // writing the tag of the page it leaves (tools/experiments/exp_r03w.sh).
        if (tag_when_taken && !SC_ABLATE(4)) o.tag[w.pg] = (bucket << SC_TAG_SHIFT) | (PAGE_LINES * LINE_ELEMS);
    }
    return w.pg * PAGE_LINES + w.ln++;
}

// A 16-byte store of a page line: write-through (`sc1`: the line leaves the L2 with the store and is dropped there).  With plain
// stores the L2 wrote 17 % more bytes to memory than the kernel stores (rocprofv3, k = 12: 51.0 M 64-byte write requests leave
// the L2 for 45.2 M that enter it, WRITE_SIZE 3.28 GB against 2.80 GB of lines and tags): a page's two 64-byte lines that share
// a 128-byte L2 line arrive two flushes apart; where the first has been written back and the line is still resident, the second
// dirties it again and both halves go out once more.  Write-through: 43.8 M requests, 2.80 GB -- what the engine counts -- at the
// same speed (k = 12 -0.6 %, k = 13 +0.5 %, k = 17 unchanged; non-temporal stores do the same for the bytes, make this kernel 3 %
// slower and the histogram pass 8 % faster: profiles/r04/write_traffic.md).  -DKDB_LINE_STORE_PLAIN / _NT build the other forms.
__device__ __forceinline__ void store_line16(uint8_t *p, const uint4 &x)
{
#if defined(KDB_LINE_STORE_PLAIN)
    *reinterpret_cast<uint4 *>(p) = x;
#else
    typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
    const u32x4 v = {x.x, x.y, x.z, x.w};
#if defined(KDB_LINE_STORE_NT)
    asm volatile("global_store_dwordx4 %0, %1, off nt" : : "v"(p), "v"(v) : "memory");
#else
    asm volatile("global_store_dwordx4 %0, %1, off sc1" : : "v"(p), "v"(v) : "memory");
#endif
#endif
}

// Flush of the 64 rings a wave owns (lane = owner of ring b): complete lines go to HBM.  The owners only DESCRIBE their
// lines (where in LDS, which line of which page) in a small per-wave list; then four lanes copy each line, 16 bytes
// each, so that one store instruction writes sixteen whole 64-byte lines.  (One lane writing its own line with four
// 16-byte stores costs four partial-line requests at the L2 per line: 0.5 ms of a 2 ms kernel, measured.)
struct LineDesc { uint32_t elem, line; };          // first element of the line in the ring arrays; line number in the pages

template <typename ELEM, int RINGS, int C, typename Stamp>
__device__ __forceinline__ void rings_flush_wave(RingLds<ELEM, RINGS, C> &R, const ScOut &o, RingOwner &w, uint32_t b, uint32_t bucket, DevCounters *ctr,
                                                 LineDesc *desc /* LDS, 64 entries of this wave */, Stamp stamp)
{
    using F = ElemFmt<ELEM>;
    using lo_t = typename F::lo_t;
    constexpr uint32_t LINE_ELEMS = F::LINE_ELEMS;
    const uint32_t lane = threadIdx.x & 63u;
    const uint32_t wd = b < (uint32_t)RINGS ? R.word[b] : 0u;             // (workgroups with fewer rings than threads)
    uint32_t r, base;
    R.decode(wd, &r, &base);
    const uint32_t nfull = r / LINE_ELEMS;
#pragma unroll 1
    for (uint32_t l = 0; l < (uint32_t)C / LINE_ELEMS; l++) {
        const bool has = nfull > l;
        const uint64_t m = __ballot(has);
        if (!m) break;                                                   // (wave-uniform)
        const uint32_t n = (uint32_t)__popcll(m);
        if (has) {
            LineDesc d;
            d.elem = R.line_elem(b, base + l * LINE_ELEMS);
            d.line = ring_next_line(R, o, w, bucket, ctr);
            desc[lane_rank_in(m)] = d;
        }
        __builtin_amdgcn_wave_barrier();
        stamp(6);
        // two groups of sixteen lines per step: both list entries are read together, then both lines, then the stores
        // (each dependent LDS read queues behind the slot requests of the CU's other workgroup: fewer trips, not fewer bytes)
        const char *const rb = reinterpret_cast<const char *>(R.ring), *const hb = reinterpret_cast<const char *>(R.hi);
        constexpr uint32_t LPL = (uint32_t)F::LINE_BYTES / 16u, GROUP = 64u / LPL;      // lanes per line (4; u16w: 8), lines per store instruction (16; 8)
        const uint32_t q16 = (lane & (LPL - 1u)) * 16u;
        for (uint32_t g = 0; g < n; g += 2u * GROUP) {
            const uint32_t e0 = g + lane / LPL, e1 = e0 + GROUP;
            const LineDesc d0 = desc[e0 < n ? e0 : 0u], d1 = desc[e1 < n ? e1 : 0u];      // (entry 0 exists: n >= 1)
            const uint32_t lo0 = F::HI ? d0.elem & 0xFFFFu : d0.elem, lo1 = F::HI ? d1.elem & 0xFFFFu : d1.elem;
            const uint4 x0 = *reinterpret_cast<const uint4 *>(rb + lo0 * (uint32_t)sizeof(lo_t) + q16);
            const uint4 x1 = *reinterpret_cast<const uint4 *>(rb + lo1 * (uint32_t)sizeof(lo_t) + q16);
            uint4 y0 = make_uint4(0, 0, 0, 0), y1 = y0;
            // u24: an odd line takes the 64 high bytes of its pair along (its own and those of the even line before it)
            const bool pair0 = F::HI && (lo0 & LINE_ELEMS) != 0u, pair1 = F::HI && (lo1 & LINE_ELEMS) != 0u;
            if (F::HI) {
                y0 = *reinterpret_cast<const uint4 *>(hb + (d0.elem >> 16) + q16);
                y1 = *reinterpret_cast<const uint4 *>(hb + (d1.elem >> 16) + q16);
            }
            if (e0 < n) {
                if (!SC_ABLATE(1)) store_line16(page_line<ELEM>(o.pages, d0.line) + q16, x0);
                if (pair0 && !SC_ABLATE(2)) store_line16(page_line_hi<ELEM>(o.pages, d0.line - 1u) + q16, y0);
            }
            if (e1 < n) {
                if (!SC_ABLATE(1)) store_line16(page_line<ELEM>(o.pages, d1.line) + q16, x1);
                if (pair1 && !SC_ABLATE(2)) store_line16(page_line_hi<ELEM>(o.pages, d1.line - 1u) + q16, y1);
            }
        }
        __builtin_amdgcn_wave_barrier();
        stamp(7);
    }
    if (nfull) R.word[b] = R.encode(base + nfull * LINE_ELEMS, r - nfull * LINE_ELEMS);
}

// one lane copies a (possibly incomplete) line of its ring to the next line of its page sequence
template <typename ELEM, int RINGS, int C>
__device__ __forceinline__ void ring_copy_line(RingLds<ELEM, RINGS, C> &R, const ScOut &o, RingOwner &w, uint32_t elem, uint32_t bucket, DevCounters *ctr)
{
    using F = ElemFmt<ELEM>;
    const uint32_t line = ring_next_line(R, o, w, bucket, ctr);
    const uint4 *src = reinterpret_cast<const uint4 *>(&R.ring[F::HI ? elem & 0xFFFFu : elem]);
    uint4 *dst = reinterpret_cast<uint4 *>(page_line<ELEM>(o.pages, line));
    constexpr int Q = F::LINE_BYTES / 16;
    uint4 x[Q];
#pragma unroll
    for (int q = 0; q < Q; q++) x[q] = src[q];
#pragma unroll
    for (int q = 0; q < Q; q++) dst[q] = x[q];
    if (F::HI) {
        // the whole pair of half-lines this line belongs to (an even line's half went nowhere yet; what lies behind the last
        // element is whatever the ring held: the tag says how many count)
        const uint4 *sh = reinterpret_cast<const uint4 *>(&R.hi[elem >> 16]);
        uint4 *dh = reinterpret_cast<uint4 *>(page_line_hi<ELEM>(o.pages, line & ~1u));
        uint4 y[Q];
#pragma unroll
        for (int q = 0; q < Q; q++) y[q] = sh[q];
#pragma unroll
        for (int q = 0; q < Q; q++) dh[q] = y[q];
    }
}

// end of the kernel (or of an input segment): the owner writes what its ring holds -- complete lines and the incomplete
// one (elements past the count are whatever the ring held: the tag says how many count) -- and the open page gets its tag
template <typename ELEM, int RINGS, int C>
__device__ __forceinline__ void ring_drain(RingLds<ELEM, RINGS, C> &R, const ScOut &o, RingOwner &w, uint32_t b, uint32_t bucket, DevCounters *ctr)
{
    constexpr uint32_t LINE_ELEMS = ElemFmt<ELEM>::LINE_ELEMS;
    uint32_t r, base;
    R.decode(R.word[b], &r, &base);
    const uint32_t nfull = r / LINE_ELEMS, rem = r - nfull * LINE_ELEMS;
    if (ElemFmt<ELEM>::HI && r == 0 && w.pg != SC_NO_PAGE && (w.ln & 1u)) {
        // u24: the last line that went out was an even one and nothing follows it: its half-line of high bytes is still here
        constexpr int Q = ElemFmt<ELEM>::LINE_BYTES / 16;
        const uint4 *sh = reinterpret_cast<const uint4 *>(&R.hi[R.line_elem(b, base - LINE_ELEMS) >> 16]);
        uint4 *dh = reinterpret_cast<uint4 *>(page_line_hi<ELEM>(o.pages, w.pg * (uint32_t)ElemFmt<ELEM>::PAGE_LINES + w.ln - 1u));
        uint4 y[Q];
#pragma unroll
        for (int q = 0; q < Q; q++) y[q] = sh[q];
#pragma unroll
        for (int q = 0; q < Q; q++) dh[q] = y[q];
    }
    for (uint32_t l = 0; l < nfull + (rem ? 1u : 0u); l++)
        ring_copy_line(R, o, w, R.line_elem(b, base + l * LINE_ELEMS), bucket, ctr);
    if (w.pg != SC_NO_PAGE) o.tag[w.pg] = (bucket << SC_TAG_SHIFT) | ((w.ln - (rem ? 1u : 0u)) * LINE_ELEMS + rem);
    R.word[b] = 0;
    w = RingOwner();
}

// Place NID elements per thread (ring word offsets woff[], elements el[], `pend` = which of them exist) in rounds of
// ROUND: slot requests (ROUND returning LDS atomics in flight), element writes, barrier, flush of the complete lines,
// barrier.  A ring that is full refuses (skew): the round is repeated for the refused elements after the flush.
// `overlap()` runs once, between the first requests and their use (work that hides the atomics' latency).
struct NoStamp { __device__ __forceinline__ void operator()(int) const {} };
struct NoHook { __device__ __forceinline__ void operator()() const {} };
template <bool V> struct BoolTag { static constexpr bool value = V; };
// `make(i, woff, el)` computes element i's ring-word offset and value.  It is called inside the first pass's request loop, element
// by element, so that the instructions that make id u + 1 issue while the atomic of id u is on its way (the ids are two thirds of
// a tile's VALU work; computed up front they left the LDS pipe idle and the sixteen round trips exposed).
// `between()` runs once, right after the first barrier: what `overlap()` staged is complete, nobody reads it before the second barrier.
template <typename ELEM, int RINGS, int C, int NID, int ROUND, typename Make, typename Overlap, typename Stamp = NoStamp, typename Between = NoHook>
__device__ __forceinline__ void rings_place(RingLds<ELEM, RINGS, C> &R, const ScOut &out, RingOwner &own, uint32_t my_ring /* RINGS if none */,
                                            uint32_t my_bucket, DevCounters *ctr,
                                            LineDesc *desc, Make make, uint32_t pend, uint32_t &round,
                                            Overlap overlap, Stamp stamp = Stamp() /* diagnostic build: phase clock */, Between between = Between())
{
    static_assert(NID % ROUND == 0, "whole rounds");
    using RL = RingLds<ELEM, RINGS, C>;
    const int j = threadIdx.x;
    bool overlapped = false, hooked = false;
    uint32_t woff[NID], el[NID];
#pragma unroll
    for (int g = 0; g < NID; g += ROUND) {
        uint32_t retry_mask = (pend >> g) & ((1u << ROUND) - 1u);
        auto pass = [&](auto first_tag) -> bool {
            constexpr bool FIRST = decltype(first_tag)::value;
            // which of the ROUND elements this lane places: bit u of retry_mask, shifted out at the top one by one -- the carry
            // of `sh + sh` is the predicate (one v_add_co_u32 per element instead of an AND and a compare).
            // (Measured and dropped: got[] kept across tiles so that no instruction clears it -- 16 registers that live through
            //  the whole tile; k = 12 scatter 1.437 -> 1.455 ms, and the 3-byte level-1 kernel spills.)
            auto requests = [&]() {
            uint32_t sh = retry_mask << (32 - ROUND);
            bool act[ROUND];
            uint32_t got[ROUND];
#pragma unroll
            for (int u = ROUND - 1; u >= 0; u--) {
                if (FIRST) make(g + u, woff[g + u], el[g + u]);
                uint32_t nsh;
                act[u] = __builtin_uadd_overflow(sh, sh, &nsh);
                sh = nsh;
                got[u] = 0u;
                if (act[u]) got[u] = atomicAdd(reinterpret_cast<uint32_t *>(reinterpret_cast<char *>(R.word) + woff[g + u]), RL::INC);
            }
            if (!overlapped) { overlapped = true; overlap(); }
            uint32_t ovf = 0;
#pragma unroll
            for (int u = 0; u < ROUND; u++) ovf |= got[u];
            if (FIRST && __ballot((ovf & RL::FULL_MASK) != 0) == 0) {
                // the usual case, wave-uniform: every request of this wave got a slot
#pragma unroll
                for (int u = 0; u < ROUND; u++) {
                    if (act[u])
                        R.put(woff[g + u], got[u], el[g + u]);
                }
                retry_mask = 0;
            } else {
                uint32_t still = 0;
#pragma unroll
                for (int u = 0; u < ROUND; u++) {
                    if (act[u]) {
                        if ((got[u] & RL::FULL_MASK) == 0) R.put(woff[g + u], got[u], el[g + u]);
                        else still |= 1u << u;
                    }
                }
                retry_mask = still;
            }
            };
            requests();
            if (__ballot(retry_mask != 0) && (j & 63) == 0) R.retry[round & 1u] = 1u;     // (any lane of this wave)
            stamp(1);
            __syncthreads();
            stamp(2);
            if (!hooked) { hooked = true; between(); }
            const uint32_t again = R.retry[round & 1u];
            if (j == 0) R.retry[(round + 1u) & 1u] = 0u;
            rings_flush_wave(R, out, own, my_ring, my_bucket, ctr, desc + (j & ~63), stamp);
            round++;
            stamp(3);
            __syncthreads();
            stamp(4);
            return again != 0u;
        };
        bool again = pass(BoolTag<true>());
        while (again) again = pass(BoolTag<false>());
    }
}

// A degenerate id's count leaves the kernel: added to the vector with one atomic -- or, when the histogram pass of the batch before
// runs beside this kernel (the overlapped one-level path: its plain read-modify-writes must not meet an atomic on the same bin),
// appended to the batch's side list, which apply_hot_kernel adds to the vector behind that pass.  Rare: a few per workgroup and launch.
__device__ __forceinline__ void hot_add(unsigned long long *__restrict__ table, DevCounters *ctr, unsigned long long id, unsigned long long n)
{
    unsigned long long *side = ctr->hot_side;
    if (side) {
        const unsigned long long slot = __hip_atomic_fetch_add(&side[0], 1ull, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if (slot < side[1]) { side[2 + 2 * slot] = id; side[3 + 2 * slot] = n; }
        else __hip_atomic_fetch_add(&ctr->internal_err, 1ull, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        return;
    }
    __hip_atomic_fetch_add(&table[id], n, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    ctr->table_dirty = 1;
}

// before a scatter kernel of the overlapped path: an empty side list, and the counters point at it; behind it: they point nowhere again
__global__ void hot_side_kernel(DevCounters *ctr, unsigned long long *side, unsigned long long cap)
{
    if (side) { side[0] = 0ull; side[1] = cap; }
    ctr->hot_side = side;
}

__global__ void __launch_bounds__(256)
apply_hot_kernel(const unsigned long long *__restrict__ side, unsigned long long *__restrict__ table)
{
    const unsigned long long n = side[0] < side[1] ? side[0] : side[1];
    for (unsigned long long i = (unsigned long long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (unsigned long long)gridDim.x * blockDim.x)
        __hip_atomic_fetch_add(&table[side[2 + 2 * i]], side[3 + 2 * i], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

// ---------------------------------------------------------------------------------
// N expansion (EXPAND mode: replace_with_none=False, the reference CLI's default; kmer.py:545-565, :586-621): a window whose
// only defects are m N's counts once for each of its 4^m fills.  Round 3 expanded such a window where it was found: one lane
// looping over the fills with global atomics while the other 63 waited, sixteen times per tile -- 21 x the time of the same
// reads without N's at 0.5 % N (25.4 ms against 1.2 ms per 10 M ragged reads).  Now their fills take the same way into the rings (or
// the LDS histogram, k <= 8) as every other id: a tile image lists its N's while it is staged, and at the top of the tile every
// (N, window) pair is one lane's work, dealt out over the whole workgroup (scatter_bases_kernel).  Only a tile
// dense with N's (more than SC_NPOS_MAX) falls back to a queue per wave (below), and only windows with more than two N's (all-N
// reads: 4^k fills each) go to the work list of expand_worklist_kernel.
// ---------------------------------------------------------------------------------
// which of the sixteen windows of a chunk hold N's and nothing else that disqualifies them (every non-ACGT base is an N and
// exists, no record start inside); bad16 = windows_bad16 of the same hood
__device__ __forceinline__ uint32_t windows_nonly16(const Hood &h, uint32_t N32, uint32_t bad16, const WinOr &o)
{
    Hood a = h, b = h;
    a.V = h.V & ~N32; a.S = 0u;                      // a defect that is not an N
    b.V = 0u;                                        // a record start strictly inside
    return bad16 & ~windows_bad16(a, o) & ~windows_bad16(b, o) & 0xFFFFu;
}

struct NWindow { uint64_t base; uint32_t sh0, sh1, nfill; };          // forward id with the N's zeroed; bit positions of the (first two) N's; 4, 16, or 0 = not expanded here

__device__ __forceinline__ NWindow nwindow_decode(uint64_t F, int i, int k, uint64_t idmask, uint32_t nwin /* bit j: base j of the window is N */)
{
    NWindow w;
    const uint32_t m = (uint32_t)__builtin_popcount(nwin);
    const uint32_t j0 = (uint32_t)__builtin_ctz(nwin | 0x80000000u), rest = nwin & (nwin - 1u);
    const uint32_t j1 = rest ? (uint32_t)__builtin_ctz(rest) : j0;
    w.sh0 = 2u * ((uint32_t)k - 1u - j0); w.sh1 = 2u * ((uint32_t)k - 1u - j1);
    w.base = ((F >> (64 - 2 * k - 2 * i)) & idmask) & ~(3ull << w.sh0) & ~(3ull << w.sh1);
    w.nfill = m == 1u ? 4u : m == 2u ? 16u : 0u;
    return w;
}

// fill f of the window (f < nfill; with one N f < 4, so the second field adds nothing): kmer.py:559-565 -> kmer_to_id of the filled k-mer
template <bool CANON>
__device__ __forceinline__ uint64_t nwindow_fill(const NWindow &w, uint32_t f, int k, uint64_t idmask)
{
    uint64_t id = w.base | ((uint64_t)(f & 3u) << w.sh0) | ((uint64_t)(f >> 2) << w.sh1);
    if (CANON) {
        uint64_t y = __builtin_bitreverse64(id);
        y = ((y >> 1) & 0x5555555555555555ull) | ((y & 0x5555555555555555ull) << 1);
        const uint64_t r = (~y >> (64 - 2 * k)) & idmask;
        id = id < r ? id : r;
    }
    return id;
}

// Dense tiles (and the k <= 8 kernel): the N-windows of one wave are queued in LDS as small entries -- lane, window, where its N's sit:
// the lane that holds the hood notes them at slots it gets from a prefix sum over the wave -- and then dealt out, two entries to a lane.
// An entry stands for FOUR fills: a window with one N, or a window with two N's and one of the four letters for its second N (four
// entries).  A lane builds its eight ids from the two forward words it reads back out of the tile image and asks the rings for their
// slots in the tile's own request phase.  What finds no slot among the 128 entries is added to the vector directly by the lane that found it.
constexpr uint32_t NQ_ENTRIES = 128;
// entry: lane | window << 6 | first N's position in the window << 10 | second N's << 15 | letter of the second N << 20 | two N's << 22
struct NQueue {
    uint32_t *q0, *q1;       // entries 0..63 and 64..127: the wave's own slots of two arrays of the idle image
    uint32_t n;
    __device__ __forceinline__ uint32_t &at(uint32_t e) { return (e < 64u ? q0 : q1)[e & 63u]; }
};

// inclusive prefix sum over the lanes of a wave of the number of bits each lane has set in `mask`
__device__ __forceinline__ uint32_t wave_incl_scan(uint32_t v);

// inclusive prefix sum over the lanes of a wave
// (DPP: six VALU instructions.  __shfl_up is ds_bpermute_b32 -- six dependent trips through the LDS, which the slot requests of both
//  workgroups of the CU keep busy: the two scans of a wave that holds an N cost more than its whole tile otherwise does.)
__device__ __forceinline__ uint32_t wave_incl_scan(uint32_t v)
{
    // row_shr:1,2,4,8 inside the rows of sixteen lanes (a lane without a source adds 0) ...
    v += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x111, 0xF, 0xF, false);
    v += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x112, 0xF, 0xF, false);
    v += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x114, 0xF, 0xF, false);
    v += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x118, 0xF, 0xF, false);
    // ... then row_bcast:15 into rows 1 and 3 (the total of the row before), row_bcast:31 into rows 2 and 3 (the total of the first half)
    v += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x142, 0xA, 0xF, false);
    v += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x143, 0xC, 0xF, false);
    return v;
}

// ---------------------------------------------------------------------------------
// scatter from residues: ids -> (ring, element).  8 <= k <= 12: ring = bucket (id bits 15..) spread over `sub` rings,
// element = the 15-bit bin; larger k (level 1 of the two-level path): ring = leading digit, element = the rest.
// ---------------------------------------------------------------------------------
// K != 0: compiled for k = K with the one-level defaults (bucket field = id bits 9..17, one ring per bucket): shifts and masks are
// immediates and two dozen scalar registers stay free (the kernel spills scalars into vector lanes: v_readlane / v_writelane are VALU work)
// RAGGED: compiled for batches whose records differ in length (record starts from the offsets) or for batches of equal-length
// records (record starts computed).  The host does not know which a device-resident batch is -- lens_kernel finds out on the
// device -- so it launches both and the one that does not apply returns at once (a few microseconds; keeping both in one kernel
// cost the equal-length batches of the headline 3 % through scalar-register pressure, same device, profiles/r04).
template <typename ID, typename ELEM, int RINGS, int C, int ROUND, bool EXPAND, bool CANON, int K = 0, int THREADS = SC_THREADS, bool RAGGED = false>
__global__ void __launch_bounds__(THREADS, 4)
scatter_bases_kernel(const uint8_t *__restrict__ bases, uint64_t nbytes, uint32_t tile0, uint32_t ntiles, int k,
                     int ring_shift /* id bits below this level's bucket field (they stay in the element) */,
                     int ring_bits /* width of the bucket field */, int sub_log2 /* rings per bucket = 1 << sub_log2 */,
                     ScOut out_arg, unsigned long long *__restrict__ table, DevCounters *ctr, RecStarts rs)
{
    constexpr int NID = 16;                                              // ids per thread per tile (one chunk), placed in one round
    if ((batch_uniform_len(ctr) == 0u) != RAGGED) return;                // (the other variant counts this batch)
    // offsets that do not tile the buffer (lens_kernel ran before this kernel on the same stream): the job fails at the sync, and the walk
    // through such offsets for a tile's record starts need not end -- nothing is counted
    if (RAGGED && ctr->bad_layout) return;
    // tile = one chunk per thread; the last chunk is only the right-hand neighbour of the one before it
    constexpr int TILE_CHUNKS = THREADS, TILE_STRIDE = THREADS - 1, TILE_POS = TILE_STRIDE * 16;
    using Tile = ScTile<EXPAND, TILE_CHUNKS>;
    // (K = 17: level 1 of the two-level path with four-byte remainders -- 512 leading digits, one ring each)
    if (K == 17 && !ElemFmt<ELEM>::HI) { k = 17; ring_shift = SC_LO_BITS_TWO_LEVEL + 9; ring_bits = 9; sub_log2 = __builtin_ctz((unsigned)RINGS) - 9; }
    else
    if (K && !ElemFmt<ELEM>::HI) { k = K; ring_shift = SC_LO_BITS_ONE_LEVEL; ring_bits = 2 * K - (int)(8 * sizeof(typename ElemFmt<ELEM>::lo_t) - (K <= 12 ? 1 : 0)); sub_log2 = __builtin_ctz((unsigned)RINGS) - ring_bits; }
    // level 1 of the two-level path with its defaults (24-bit remainders: the digit = id bits 21 .. 2K - 4, RINGS >> digit bits rings per digit)
    if (K && ElemFmt<ELEM>::HI) { k = K; ring_shift = SC_LO_BITS_TWO_LEVEL + 9; ring_bits = 2 * K - 24; sub_log2 = __builtin_ctz((unsigned)RINGS) - ring_bits; }
    const ScOut out = sc_out_of_workgroup(out_arg);
    const uint32_t G = out.grid;
    ntiles = sc_pin(ntiles);
    __shared__ Tile T[2];                                                // this tile's image and the next one's (staged while the atomics fly)
    static_assert(sizeof(Tile) >= THREADS * sizeof(LineDesc), "a dead tile image holds the waves' line lists");
    __shared__ RingLds<ELEM, RINGS, C> R;
    const int j = threadIdx.x;
    for (int b = j; b < RINGS; b += THREADS) R.word[b] = 0;
    if (j == 0) { R.pg_count = 0; R.retry[0] = 0; R.retry[1] = 0; T[0].has_n[0] = 0; T[1].has_n[0] = 0; T[0].ncnt = 0; T[1].ncnt = 0; }
    if (j < SC_HOT) { R.hot_tag[j] = 0ull; R.hot_cnt[j] = 0; }
    if (EXPAND) __syncthreads();                                         // (has_n is cleared before the first image is staged)
    RingOwner own;
    // ring r is owned (flushed, drained) by thread r * (THREADS / RINGS): the owners are spread over all the waves
    static_assert(THREADS % RINGS == 0, "whole threads per ring");
    constexpr int OWN_STEP = THREADS / RINGS;
    const uint32_t my_ring = (j % OWN_STEP) == 0 ? (uint32_t)(j / OWN_STEP) : (uint32_t)RINGS;
    const uint32_t my_bucket = my_ring >> sub_log2;                      // the bucket of that ring
    const uint32_t sub4 = ((uint32_t)j & ((1u << sub_log2) - 1u)) * 4u;  // which of the bucket's rings this thread places into (as a byte offset into word[])
    const int ring_word_sh = sub_log2 + 2;
    const int canonical = CANON ? 1 : 0;
    const IdParams<ID> idp(k, canonical);
    const WinOr winor(k);
    const uint64_t idmask = (1ull << (2 * k)) - 1ull;
    const uint32_t kmask = (1u << k) - 1u;
    const ID keep = (ID)(((ID)1 << ring_shift) - 1);                     // element = id with the bucket field cut out
    const bool owner_of_windows = j < TILE_STRIDE;                       // the last thread's chunk is only the neighbour of the chunk before it
    const uint32_t ulen = batch_uniform_len(ctr);
    // record starts of a uniform-length batch: residue class of this thread's chunk start, advanced tile by tile
    uint32_t x = 0, xstep = 0;
    if (ulen) {
        x = (uint32_t)((((uint64_t)tile0 + blockIdx.x) * (uint64_t)TILE_POS + 16ull * j) % ulen);
        xstep = (uint32_t)(((uint64_t)G * TILE_POS) % ulen);
    }
    unsigned long long extra = 0;                                        // k-mers added to the vector directly
    uint32_t stat_tot = 0;                                               // bad residues | record-start marks met << 16 (a workgroup takes < 4096 tiles: scatter_max_tiles)
    uint32_t round = 0;
    int buf = sc_pin(0);
    SC_STAMP_INIT;
    SC_WG_CLOCK(0, 0);

    // this thread's chunk of a tile (a uniform number).  A tile whose 512 chunks all lie inside the buffer -- all but the last one or
    // two of a batch -- is read at a scalar base plus this thread's fixed offset: no 64-bit address arithmetic, no bounds test per lane
    const uint32_t my_byte = 16u * (uint32_t)j;
    auto fetch_tile = [&](uint64_t tile_no) -> ScChunk {
        const uint64_t first = tile_no * (uint64_t)(TILE_STRIDE * 16);
        if (first + (uint64_t)(TILE_CHUNKS * 16) <= nbytes) {
            ScChunk c;
            c.v = load_once16(bases + first + my_byte);
            c.nexist = 0u;
            return c;
        }
        return sc_fetch(bases, nbytes, tile_no * TILE_STRIDE + (uint64_t)j);
    };
    // prologue: the first tile's image; the second tile's chunk is requested
    ScChunk mine;
    mine.v = make_uint4(0, 0, 0, 0); mine.nexist = 0xFFFFu;
    if (blockIdx.x < ntiles) {
        mine = fetch_tile((uint64_t)tile0 + blockIdx.x);
        const uint32_t nb_ = sc_stage_chunk<EXPAND, true>(T[0], mine, j, true, ulen ? uniform_starts(x, ulen) : 0u, blockIdx.x + 1u,
                                                          (((uint64_t)tile0 + blockIdx.x) * TILE_STRIDE + (uint64_t)j) * 16ull, ctr, owner_of_windows);
        if (owner_of_windows) stat_tot += nb_;
        if (ulen) { x += xstep; if (x >= ulen) x -= ulen; }
        if (blockIdx.x + G < ntiles) mine = fetch_tile((uint64_t)tile0 + blockIdx.x + G);
    }
    __syncthreads();
    // a ragged batch: the record starts of the first tile; the walk through the offsets starts at first_rec[tile start >> 12], fetched a tile ahead
    constexpr bool ragged = RAGGED;
    // The offsets run TWO tiles ahead of the image they are applied to, and are asked for BEHIND the staging of a tile (round 5).  Fetched and applied in the same
    // iteration, the wait for them (right behind the first barrier) was an `s_waitcnt vmcnt(0)`: the residues of the tile after next are requested behind a branch,
    // so the compiler cannot count them as younger -- every wave waited for its own HBM load of a moment ago, on every tile: the same residues with ONE record a base
    // shorter took 1.55 ms instead of 1.22 (tools/experiments/exp_r05_ragged_kernel_on_uniform.py).  Now the wait that the staging needs anyway (for the residues
    // requested a tile ago) also covers the offsets requested before them, and nothing waits behind the barrier.
    // Three things the ISA showed (a wave stalled twice per tile): first_rec[...] is one word at a uniform address, so the compiler read it with a vector load and
    // put a v_readfirstlane -- and an `s_waitcnt vmcnt(0)` -- right behind it, in the middle of the staging; the index now goes through a register it cannot see
    // through (first_rec_of), the word stays per lane and is waited for a tile later.  And whatever is waited for between the request of the next residues (`mine`:
    // behind a branch, so never counted as "younger") and their use waits for those residues too: the offsets are therefore requested a tile EARLIER than they are
    // applied and copied (probe2 -> probe) right behind the staging's own wait, which covers them.
    auto first_rec_of = [&](uint64_t tile_no) -> uint32_t {
        uint32_t idx = (uint32_t)((tile_no * (uint64_t)TILE_POS) >> FIRST_REC_SHIFT);
        asm volatile("" : "+v"(idx));
        return rs.first_rec[idx];
    };
    uint32_t first_next = 0;                                             // first_rec of the tile whose offsets are fetched next (two tiles after the one being placed)
    StartProbe probe, probe2;                                            // offsets: of the tile being staged (applied behind the barrier), of the tile after it (on their way)
    probe.r = 0; probe.off = ~0ull; probe.beyond = ~0ull;
    probe2 = probe;
    if (ragged && blockIdx.x < ntiles) {
        const uint64_t P0 = ((uint64_t)tile0 + blockIdx.x) * (uint64_t)TILE_POS;
        starts_apply<THREADS>(T[0], rs, P0, starts_fetch<THREADS>(rs, rs.first_rec[P0 >> FIRST_REC_SHIFT], j));
        if (blockIdx.x + G < ntiles) probe2 = starts_fetch<THREADS>(rs, rs.first_rec[(((uint64_t)tile0 + blockIdx.x + G) * (uint64_t)TILE_POS) >> FIRST_REC_SHIFT], j);
        if (blockIdx.x + 2 * G < ntiles) first_next = first_rec_of((uint64_t)tile0 + blockIdx.x + 2ull * G);
        __syncthreads();
    }

    // an id -> byte offset of its ring's word, and the element (the bucket field cut out)
    auto ring_and_element = [&](ID id, uint32_t &woff_u, uint32_t &el_u) {
        // (34-bit ids: the bucket field may reach past bit 31)
        const uint32_t ring = sizeof(ID) > 4 ? (uint32_t)((uint64_t)id >> ring_shift) & ((1u << ring_bits) - 1u)
                                             : __builtin_amdgcn_ubfe((uint32_t)id, (uint32_t)ring_shift, (uint32_t)ring_bits);
        woff_u = (ring << ring_word_sh) | sub4;
        el_u = bfi((uint32_t)keep, (uint32_t)id, (uint32_t)(id >> ring_bits));                  // (< 2^32)
    };

    for (uint32_t t = blockIdx.x; t < ntiles; t += G) {
        const uint64_t tile = (uint64_t)tile0 + t;
        const Hood h = sc_load_hood<CANON>(T[buf], j < TILE_STRIDE ? j : 0);
        const uint32_t bad16 = owner_of_windows ? windows_bad16(h, winor) : 0xFFFFu;
        uint32_t N32 = 0;
        if (EXPAND && owner_of_windows) N32 = (T[buf].nn[j] & 0xFFFFu) | (T[buf].nn[j + 1] << 16);
        // The tile's N-windows (EXPAND).  The image lists its N's (sc_stage_chunk); every (N, window) pair -- k per N -- is one lane's work,
        // dealt out over the WHOLE workgroup here, at the top of the tile: the lane tests its pair's window in the image (every defect an
        // N, no record start inside, this N its first), and asks the rings for slots for the window's 4 or 16 fills like for any other id,
        // in the rings' request phase (the last placement ended with a barrier, the tile's own begins below and its flush writes the
        // lines out).  No queue, no prefix sums, no placement round of its own.  (Before: the wave that held an N did all of this for its
        // 1024 positions alone -- two wave scans, a queue, five dependent trips through the LDS -- while the other waves waited at the
        // barrier: a third of a tile's time at 0.05 % N, profiles/r04/experiments.md.)  A fill whose ring is full goes to the vector.
        const uint32_t ncnt = EXPAND ? T[buf].ncnt : 0u;                 // (workgroup-uniform: complete since the barrier that ended the staging)
        // (the position of this lane's first pair's N is read along with the count, whatever the count will say: one trip through the LDS less)
        const uint32_t kinv = (1u << 20) / (uint32_t)k + 1u;             // pr / k = pr x kinv >> 20, exact for pr < 2^20 / k
        uint32_t np_first = 0;
        if (EXPAND) np_first = T[buf].npos[(((uint32_t)j * kinv) >> 20) & (SC_NPOS_MAX - 1u)];
        if (EXPAND && ncnt != 0u) {
            SC_STAMP(0);                                                 // (diagnostic build: the N block is clocked under "drain")
            // room in this workgroup's page sequence for the tile's fills (at most 8 per pair: a window with two N's is found through
            // its first N only) on top of every id its remaining tiles can still emit?
            constexpr uint32_t PAGE_ELEMS = (uint32_t)ElemFmt<ELEM>::LINE_ELEMS * (uint32_t)ElemFmt<ELEM>::PAGE_LINES;
            using RL = RingLds<ELEM, RINGS, C>;
            const uint32_t tiles_left = (ntiles - t + G - 1u) / G;
            const uint32_t fills_max = ncnt <= SC_NPOS_MAX ? 8u * ncnt * (uint32_t)k : (uint32_t)THREADS * 8u;
            const uint32_t need = (uint32_t)(((uint64_t)tiles_left * TILE_POS + (uint64_t)fills_max + (uint64_t)RINGS * C + PAGE_ELEMS - 1u) / PAGE_ELEMS) + (uint32_t)RINGS + 2u;
            const bool room = R.pg_count + need <= out.wg_pages;         // (else: the ids still to come need the pages.  pg_count only moves between a placement's barriers)
            // four fills: slots requested together, then written; a refusal (or no room) goes to the vector
            auto place4 = [&](const ID (&fid)[4], uint32_t live) {
                uint32_t fw[4], fe[4], fg[4];
#pragma unroll
                for (int u = 0; u < 4; u++) {
                    ring_and_element(fid[u], fw[u], fe[u]);
                    fg[u] = RL::FULL_MASK;
                    if (room && ((live >> u) & 1u)) fg[u] = atomicAdd(reinterpret_cast<uint32_t *>(reinterpret_cast<char *>(R.word) + fw[u]), RL::INC);
                }
#pragma unroll
                for (int u = 0; u < 4; u++) {
                    if (!((live >> u) & 1u)) continue;
                    if ((fg[u] & RL::FULL_MASK) == 0u) { R.put(fw[u], fg[u], fe[u]); continue; }
                    __hip_atomic_fetch_add(&table[(uint64_t)fid[u]], 1ull, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    extra += 1ull; ctr->table_dirty = 1;
                }
            };
            if (ncnt <= SC_NPOS_MAX) {
                const uint32_t npairs = ncnt * (uint32_t)k;              // (<= 256 x 17: pr x kinv < 2^32)
#pragma unroll 1
                for (uint32_t pr = (uint32_t)j; pr < npairs; pr += (uint32_t)THREADS) {
                    const uint32_t n = (pr * kinv) >> 20, w = pr - n * (uint32_t)k;              // the pair: N number n, window that has it at position w
                    const int W = (int)(pr == (uint32_t)j ? np_first : (uint32_t)T[buf].npos[n]) - (int)w;      // where that window starts
                    if (W < 0 || W >= TILE_POS) continue;                                          // (not a window of this tile)
                    const int wc = W >> 4;
                    const uint32_t wi = (uint32_t)W & 15u;
                    const uint32_t m0 = T[buf].msk[wc], m1 = T[buf].msk[wc + 1];
                    const uint32_t f0w = T[buf].fwd[wc], f1w = T[buf].fwd[wc + 1];                 // (read with the masks: one trip)
                    const uint32_t Vw = (m0 & 0xFFFFu) | (m1 << 16), Sw = (m0 >> 16) | (m1 & 0xFFFF0000u);
                    const uint32_t Nw = (T[buf].nn[wc] & 0xFFFFu) | (T[buf].nn[wc + 1] << 16);
                    const uint32_t nwin = (Nw >> wi) & kmask;
                    const bool ok = ((Vw >> wi) & kmask) == nwin && (((Sw >> 1) >> wi) & (kmask >> 1)) == 0u && (nwin & ((1u << w) - 1u)) == 0u;
                    if (!ok) continue;
                    const uint64_t F = ((uint64_t)f0w << 32) | f1w;
                    if (__builtin_popcount(nwin) > 2) { expand_n_window(table, F, (int)wi, k, canonical, idmask, nwin, &extra, ctr); continue; }      // the work list
                    if (sizeof(ID) == 4) {
                        // 32-bit ids (k <= 16): the window with its N fields zeroed, forward (bf) and reverse-complemented (br, those fields
                        // cleared too); a fill is two shifted ORs and a min.  Two N's: the second one's four letters one after the other.
                        const uint32_t km2 = 2u * (uint32_t)(k - 1), rest = nwin & (nwin - 1u);
                        const uint32_t s0 = km2 - 2u * (uint32_t)__builtin_ctz(nwin), s1 = rest ? km2 - 2u * (uint32_t)__builtin_ctz(rest) : s0;
                        const uint32_t top = wi ? __builtin_amdgcn_alignbit(f0w, f1w, 32u - 2u * wi) : f0w;
                        const uint32_t bf = (top >> (32 - 2 * k)) & ~(3u << s0) & ~(3u << s1);
                        const uint32_t br = CANON ? (rc_word(bf) >> (32 - 2 * k)) & ~(3u << (km2 - s0)) & ~(3u << (km2 - s1)) : 0u;
#pragma unroll 1
                        for (uint32_t g = 0; g < (rest ? 4u : 1u); g++) {
                            const uint32_t bfg = rest ? bf | (g << s1) : bf, brg = rest ? br | ((3u - g) << (km2 - s1)) : br;
                            ID fid[4];
#pragma unroll
                            for (uint32_t u = 0; u < 4u; u++) {
                                const uint32_t a = bfg | (u << s0), r = brg | ((3u - u) << (km2 - s0));
                                fid[u] = (ID)(CANON ? (a < r ? a : r) : a);
                            }
                            place4(fid, 0xFu);
                        }
                        continue;
                    }
                    const NWindow nw = nwindow_decode(F, (int)wi, k, idmask, nwin);
#pragma unroll 1
                    for (uint32_t f0 = 0; f0 < nw.nfill; f0 += 4u) {
                        ID fid[4];
#pragma unroll
                        for (int u = 0; u < 4; u++) fid[u] = (ID)nwindow_fill<CANON>(nw, f0 + (uint32_t)u, k, idmask);
                        place4(fid, 0xFu);
                    }
                }
            } else {
            // A tile dense with N's (more than the list holds): every wave looks through its own windows; the N-only ones are queued in the
            // wave's own lanes' slots of the other image (idle until the tile's own placement stages the next tile into it), four fills to
            // an entry, and dealt out two entries to a lane.
            const uint32_t lane = (uint32_t)j & 63u;
            const int wbase = j & ~63;
            NQueue Q{&T[buf ^ 1].fwd[wbase], &T[buf ^ 1].msk[wbase], 0u};
            auto to_vector = [&](const NWindow &w) {                     // (no slot in the queue: straight to the vector)
#pragma unroll 1
                for (uint32_t f = 0; f < w.nfill; f++)
                    __hip_atomic_fetch_add(&table[nwindow_fill<CANON>(w, f, k, idmask)], 1ull, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                if (w.nfill) { extra += w.nfill; ctr->table_dirty = 1; }
            };
            const uint32_t nonly = (N32 && bad16) ? windows_nonly16(h, N32, bad16, winor) : 0u;
            const uint32_t km2 = 2u * (uint32_t)(k - 1);
            ID nbf[2] = {0, 0}, nbr[2] = {0, 0};
            uint32_t nsh[2] = {0, 0}, pend2 = 0;
            if (__ballot(nonly != 0u)) {                                 // (wave-uniform)
                // the N-only windows of this lane: with one N (one entry), with two (four entries); more: the work list
                uint32_t one_n = 0, two_n = 0;
#pragma unroll 1
                for (uint32_t m = nonly; m; m &= m - 1u) {
                    const int i = __builtin_ctz(m);
                    const uint32_t nwin = (N32 >> i) & kmask, cnt = (uint32_t)__builtin_popcount(nwin);
                    if (cnt == 1u) one_n |= 1u << i;
                    else if (cnt == 2u) two_n |= 1u << i;
                    else expand_n_window(table, h.F(), i, k, canonical, idmask, nwin, &extra, ctr);
                }
                // slots: this lane's entries behind those of the lanes before it
                const uint32_t mine_n = (uint32_t)__builtin_popcount(one_n) + 4u * (uint32_t)__builtin_popcount(two_n);
                const uint32_t incl = wave_incl_scan(mine_n), tot = (uint32_t)__builtin_amdgcn_readlane((int)incl, 63);
                uint32_t sl = incl - mine_n;
                Q.n = tot < NQ_ENTRIES ? tot : NQ_ENTRIES;
#pragma unroll 1
                for (uint32_t m = one_n | two_n; m; m &= m - 1u) {
                    const uint32_t i = (uint32_t)__builtin_ctz(m), nwin = (N32 >> i) & kmask;
                    const uint32_t j0 = (uint32_t)__builtin_ctz(nwin), rest = nwin & (nwin - 1u);
                    const uint32_t e0 = lane | (i << 6) | (j0 << 10);
                    if (!rest) {
                        if (sl < NQ_ENTRIES) Q.at(sl) = e0; else to_vector(nwindow_decode(h.F(), (int)i, k, idmask, nwin));
                        sl++;
                    } else if (sl + 4u <= NQ_ENTRIES) {
                        const uint32_t e1 = e0 | ((uint32_t)__builtin_ctz(rest) << 15) | (1u << 22);
                        Q.at(sl) = e1; Q.at(sl + 1u) = e1 | (1u << 20); Q.at(sl + 2u) = e1 | (2u << 20); Q.at(sl + 3u) = e1 | (3u << 20);
                        sl += 4u;
                    } else {
                        to_vector(nwindow_decode(h.F(), (int)i, k, idmask, nwin));
                        for (uint32_t q = 0; q < 4u; q++) if (sl + q < NQ_ENTRIES) Q.at(sl + q) = 0xFFFFFFFFu;      // (slots of the queue that stay empty)
                        sl += 4u;
                    }
                }
                __builtin_amdgcn_wave_barrier();
            }
            if (Q.n) {                                                   // (wave-uniform)
                // this lane's share: up to two entries.  Kept per entry: the forward id with the first N zeroed (the second N's letter filled in),
                // its reverse-complement id with the first N's field cleared, where that N sits -- a fill is then two shifted ORs and a min
#pragma unroll
                for (uint32_t q = 0; q < 2u; q++) {
                    const uint32_t ei = lane * 2u + q;
                    const uint32_t e = ei < Q.n ? Q.at(ei) : 0xFFFFFFFFu;
                    if (e != 0xFFFFFFFFu) {
                        const int c = wbase + (int)(e & 63u);
                        const uint32_t i = (e >> 6) & 15u, s0 = km2 - 2u * ((e >> 10) & 31u);
                        const uint64_t F = ((uint64_t)T[buf].fwd[c] << 32) | T[buf].fwd[c + 1];
                        ID bf = (ID)((F >> (64 - 2 * k - 2 * (int)i)) & idmask) & ~((ID)3 << s0);
                        if ((e >> 22) & 1u) { const uint32_t s1 = km2 - 2u * ((e >> 15) & 31u); bf = (bf & ~((ID)3 << s1)) | ((ID)((e >> 20) & 3u) << s1); }
                        nbf[q] = bf; nsh[q] = s0; pend2 |= 0xFu << (4u * q);
                        if (CANON) {
                            uint64_t y = __builtin_bitreverse64((uint64_t)bf);
                            y = ((y >> 1) & 0x5555555555555555ull) | ((y & 0x5555555555555555ull) << 1);
                            nbr[q] = (ID)((~y >> (64 - 2 * k)) & idmask) & ~((ID)3 << (km2 - s0));
                        }
                    }
                }
            }
            auto fill_id = [&](int u) -> ID {                           // (u is a compile-time number wherever this is called)
                const ID f0 = (ID)(u & 3);
                ID id = nbf[u >> 2] | (f0 << nsh[u >> 2]);
                if (CANON) {
                    const ID r = nbr[u >> 2] | (((ID)3 - f0) << (km2 - nsh[u >> 2]));
                    id = id < r ? id : r;
                }
                return id;
            };
            if (pend2 & 0xFu) { const ID fid[4] = {fill_id(0), fill_id(1), fill_id(2), fill_id(3)}; place4(fid, pend2 & 0xFu); }
            if (pend2 >> 4)   { const ID fid[4] = {fill_id(4), fill_id(5), fill_id(6), fill_id(7)}; place4(fid, pend2 >> 4); }
            }
            SC_STAMP(5);
        }
        uint64_t same; uint32_t id0;
        // (the test only has to fire on degenerate stretches -- poly-A/G, short-period repeats: there the lanes' chunks begin with the
        //  same k bases; the loop below compares the ids themselves)
        const bool degenerate = wave_dominant(k < 16 ? h.f0 >> (32 - 2 * k) : h.f0, &same, &id0);
        uint32_t pend = ~bad16 & 0xFFFFu;                                // bit u: window u is counted and still has to be placed
        if (degenerate) {
            // lanes that share an id with >= 15 others add it to the vector at once (one atomic per wave and id)
#pragma unroll 1
            for (int u = 0; u < NID; u++) {
                const ID idu = idp.id_any(h, u);
                const bool live = (pend >> u) & 1u;
                const uint64_t act = __ballot(live);
                if (!act) continue;
                const int first = __ffsll((unsigned long long)act) - 1;
                const uint32_t lo0 = (uint32_t)__builtin_amdgcn_readlane((int)(uint32_t)idu, first);
                const uint32_t hi0 = sizeof(ID) > 4 ? (uint32_t)__builtin_amdgcn_readlane((int)(uint32_t)((uint64_t)idu >> 32), first) : 0u;
                const bool same_id = live && (uint32_t)idu == lo0 && (sizeof(ID) > 4 ? (uint32_t)((uint64_t)idu >> 32) == hi0 : true);
                const uint64_t grp = __ballot(same_id);
                if (__popcll(grp) >= 16 && same_id) {
                    pend &= ~(1u << u);
                    if (lane_rank_in(grp) == 0) {
                        const uint32_t n = (uint32_t)__popcll(grp);
                        const unsigned long long want = (1ull << 40) | (unsigned long long)idu;
                        const uint32_t hs = ((uint32_t)idu * 2654435761u) >> (32 - 6);                 // SC_HOT = 64 slots
                        const unsigned long long old = atomicCAS(&R.hot_tag[hs], 0ull, want);
                        if (old == 0ull || old == want) atomicAdd(&R.hot_cnt[hs], n);
                        else hot_add(table, ctr, (unsigned long long)idu, (unsigned long long)n);
                        extra += (unsigned long long)n;
                    }
                }
            }
        }
        // 32-bit canonical ids: both strands' windows come TOP-aligned out of one v_alignbit_b32 each (the forward one from
        // f0:f1 at 32 - 2u; the reverse one from r1:r0 moved up by 16 - k bases once per chunk, at 2u), min() compares the
        // k-mers in their leading 2k bits -- the bits below only break ties between equal k-mers -- and one shift drops them
        // (kmer.py:307-315: min(fwd id, reverse-complement id))
        uint32_t r2lo = 0, r2hi = 0;
        if (CANON && sizeof(ID) == 4) {
            const uint64_t r2 = h.R() << (2 * (16 - k));
            r2lo = (uint32_t)r2; r2hi = (uint32_t)(r2 >> 32);
        }
        // element u of this lane: byte offset of its ring's word, and the element (called from the request loop of rings_place)
        auto make = [&](int u, uint32_t &woff_u, uint32_t &el_u) {
            ID id;
            if (CANON && sizeof(ID) == 4) {
                const uint32_t wf = u == 0 ? h.f0 : __builtin_amdgcn_alignbit(h.f0, h.f1, 32 - 2 * u);
                const uint32_t wr = u == 0 ? r2lo : __builtin_amdgcn_alignbit(r2hi, r2lo, 2 * u);
                id = (ID)((wf < wr ? wf : wr) >> (32 - 2 * k));
            } else {
                id = idp.id(h, u);
            }
            ring_and_element(id, woff_u, el_u);
        };
        SC_STAMP(0);                                                     // hood, window masks, ids
        // place; while the first slot requests fly: encode the next tile's chunk into the other image, request the chunk after it
        // (this tile's image is dead since the hoods were loaded: its first 4 KiB serve as the waves' line lists)
        rings_place<ELEM, RINGS, C, NID, ROUND>(R, out, own, my_ring, my_bucket, ctr, reinterpret_cast<LineDesc *>(&T[buf]), make, pend, round, [&]() {
            if (t + G < ntiles) {
                const uint32_t nb_ = sc_stage_chunk<EXPAND, true>(T[buf ^ 1], mine, j, true, ulen ? uniform_starts(x, ulen) : 0u, t + G + 1u,
                                                                  ((tile + G) * TILE_STRIDE + (uint64_t)j) * 16ull, ctr, owner_of_windows);
                if (owner_of_windows) stat_tot += nb_;
                if (ulen) { x += xstep; if (x >= ulen) x -= ulen; }
                if (ragged) {                                            // (kernel-uniform)
                    probe = probe2;                                      // the staged tile's offsets (requested a tile ago: here behind the wait for its residues)
                    if (t + 2 * G < ntiles) {                            // the record starts of the tile after it: offsets on their way
                        probe2 = starts_fetch<THREADS>(rs, first_next, j);
                        if (t + 3 * G < ntiles) first_next = first_rec_of(tile + 3ull * G);
                    }
                }
                if (t + 2 * G < ntiles) mine = fetch_tile(tile + 2ull * G);
            }
        }, SC_STAMP_FN, [&]() {
            if (ragged && t + G < ntiles) starts_apply<THREADS>(T[buf ^ 1], rs, (tile + G) * (uint64_t)TILE_POS, probe);
            if (EXPAND && j == 0) T[buf].ncnt = 0u;                      // (every wave has dealt with this image's N's; it is staged again after the next barrier)
        });
        SC_STAMP(1);                                                     // placement, staging of the next tile, flush
        buf = sc_pin(buf ^ 1);                                           // (uniform: the image's address is scalar arithmetic, not a 16-cycle v_mul_lo_u32 per lane)
    }

    if (my_ring < (uint32_t)RINGS && !SC_ABLATE(8)) ring_drain(R, out, own, my_ring, my_bucket, ctr);
    if (j < SC_HOT && R.hot_tag[j])          // (every wave passed the last round's barriers after its last insertion)
        hot_add(table, ctr, R.hot_tag[j] & ((1ull << 40) - 1ull), (unsigned long long)R.hot_cnt[j]);
    SC_STAMP(5);
    SC_STAMP_END;
    SC_WG_CLOCK(0, 1);
    const unsigned long long we = wave_sum(extra), wb = wave_sum((unsigned long long)(stat_tot & 0xFFFFu)), wm = wave_sum((unsigned long long)(stat_tot >> 16));
    if ((j & 63) == 0) {
        if (we) __hip_atomic_fetch_add(&ctr->total_kmers, we, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if (wb) __hip_atomic_fetch_add(&ctr->n_bad, wb, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if (wm) __hip_atomic_fetch_add(&ctr->marks_seen, wm, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
}

// ---------------------------------------------------------------------------------
// level 2 of the two-level path (13 <= k <= 17): the pages level 1 wrote (u32 remainders  [ hi ][ bucket2 : 9 ][ lo : 6 ],
// grouped by leading digit b1 through their page list) -> pages of 15-bit (k = 17: 16-bit) bins, tagged with the final
// bucket b1 << 9 | bucket2.  A workgroup takes a contiguous span of level 1's page list, so its rings hold elements
// of one b1 at a time; where the span passes into the next b1 the rings are drained (partial pages).
// ---------------------------------------------------------------------------------
struct PageEntry { uint32_t page, nelems; };

// digit whose page range contains list position p: largest b with page_base[b] <= p (wave-uniform)
__device__ __forceinline__ uint32_t l2_digit_of(const uint32_t *__restrict__ page_base, uint32_t nb1, uint32_t p)
{
    uint32_t lo = 0, hi = nb1 - 1;
    while (lo < hi) { const uint32_t mid = (lo + hi + 1) >> 1; if (page_base[mid] <= p) lo = mid; else hi = mid - 1; }
    return lo;
}

// spans and page ranges of the level-2 workgroups: workgroup w scatters list positions [P w / G, P (w + 1) / G); it can
// need one page per 512 elements, one partial page per ring and digit it touches, and one spare (in_page_elems: per level-1 page)
__global__ void __launch_bounds__(1024)
l2_plan_kernel(const uint32_t *__restrict__ page_base1 /* [nb1 + 1] */, uint32_t nb1, uint32_t G, uint32_t rings, uint32_t in_page_elems,
               uint32_t *__restrict__ wg_range /* [G + 1] */, uint32_t *__restrict__ cursor /* arena pages handed out so far: advanced by what this batch really needs */,
               uint32_t cap /* pages in the arena */, DevCounters *ctr)
{
    __shared__ uint32_t wsum[1024 / 64];
    const uint32_t w = threadIdx.x, P = page_base1[nb1];
    uint32_t need = 0;
    if (w < G) {
        const uint32_t s0 = (uint32_t)((uint64_t)P * w / G), s1 = (uint32_t)((uint64_t)P * (w + 1) / G);
        if (s1 > s0) {
            const uint32_t d0 = l2_digit_of(page_base1, nb1, s0), d1 = l2_digit_of(page_base1, nb1, s1 - 1);
            need = ((s1 - s0) * in_page_elems + 511u) / 512u + rings * (d1 - d0 + 1u) + 1u;
        }
    }
    uint32_t tot;
    const uint32_t excl = block_excl_scan<1024>(need, wsum, &tot);
    // (the host only submits a batch whose worst case fits behind the worst case of everything before it; if that ever failed,
    //  no workgroup gets a page and the job fails loudly)
    const uint32_t range0 = *cursor;
    const bool fits = tot <= cap - (range0 < cap ? range0 : cap);
    __syncthreads();                                  // (every thread has read the cursor)
    if (w < G) wg_range[w] = range0 + (fits ? excl : 0u);
    if (w == 0) { wg_range[G] = range0 + (fits ? tot : 0u); *cursor = range0 + (fits ? tot : 0u); if (!fits) atomicAdd(&ctr->internal_err, 1ull); }
}

// THREADS = 512 (RINGS = 512 rings of 64 elements, 64-byte lines, two workgroups per CU) or 1024 (ELEM = u16w: 512 rings of 128 elements, 128-byte
// pieces, one workgroup per CU: ring r is owned by thread 2 r)
template <typename IN /* uint32_t or u24: level 1's element format */, typename ELEM /* u16 / u16w */, int RINGS, int C, bool FIXED = false /* ring_shift = 12, ring_bits = 9: the defaults, compiled in */,
          int THREADS = SC_THREADS>
__global__ void __launch_bounds__(THREADS, 4)
scatter_ids_kernel(const uint8_t *__restrict__ pages1, const PageEntry *__restrict__ list1, const uint32_t *__restrict__ page_base1, uint32_t nb1,
                   int ring_shift, int ring_bits, ScOut out_arg, DevCounters *ctr)
{
    if (FIXED) { ring_shift = SC_LO_BITS_TWO_LEVEL; ring_bits = 9; }
    const ScOut out = sc_out_of_workgroup(out_arg);
    constexpr int NID = 16;
    constexpr int EP = ElemFmt<IN>::LINE_ELEMS * SC_PAGE_LINES / 64;     // elements of a page per lane: 4 (u32) or 8 (u24)
    constexpr int PPT = NID / EP;                                        // pages per thread and tile: 4 or 2
    constexpr uint32_t NWAVES = THREADS / 64;
    constexpr uint32_t L2_TILE_PAGES = NWAVES * PPT;                     // wave w reads pages w, NWAVES + w, ... of a tile: a whole page per load instruction
    constexpr size_t IN_PAGE_BYTES = ElemFmt<IN>::PAGE_BYTES;
    static_assert(THREADS % RINGS == 0, "whole threads per ring");
    constexpr int OWN_STEP = THREADS / RINGS;
    __shared__ RingLds<ELEM, RINGS, C> R;
    __shared__ LineDesc desc[THREADS];
    const int j = threadIdx.x, wave = j >> 6, lane = j & 63;
    for (int b = j; b < RINGS; b += THREADS) R.word[b] = 0;
    if (j == 0) { R.pg_count = 0; R.retry[0] = 0; R.retry[1] = 0; }
    RingOwner own;
    const uint32_t my_ring = (j % OWN_STEP) == 0 ? (uint32_t)(j / OWN_STEP) : (uint32_t)RINGS;      // the ring this thread flushes and drains (RINGS: none)
    const uint32_t keep = (1u << ring_shift) - 1u;
    const uint32_t P = page_base1[nb1];
    const uint32_t s0 = (uint32_t)((uint64_t)P * blockIdx.x / out.grid), s1 = (uint32_t)((uint64_t)P * (blockIdx.x + 1) / out.grid);
    uint32_t round = 0;
    __syncthreads();
    if (s1 == s0) return;

    // tile = up to L2_TILE_PAGES consecutive pages of one digit.  The page addresses come out of the page list, so a tile's
    // pages are two dependent trips to HBM.  u24 pages (k <= 16): three tiles are in flight -- A (its elements are in registers,
    // being placed), B (its pages are being read: their list entries arrived a tile ago) and C (its list entries are being
    // read): level 2 1.92 -> 1.86 ms.  u32 pages (k = 17, four pages per lane and tile): holding the entries for another
    // tile spills registers (3 % slower, measured); there all entries of B are read, then all its pages (two trips instead of
    // one entry-then-page pair after the other: 2.24 -> 2.04 ms).
    struct TileIt { uint32_t pos, npg, b1, end_b1; bool valid; };
    auto tile_after = [&](const TileIt &t) {
        TileIt n = t;
        n.pos = t.pos + t.npg;
        n.valid = t.valid && n.pos < s1;
        if (n.valid && n.pos == t.end_b1) {
            n.b1 = l2_digit_of(page_base1, nb1, n.pos);
            n.end_b1 = page_base1[n.b1 + 1] < s1 ? page_base1[n.b1 + 1] : s1;
        }
        n.npg = n.valid ? (n.end_b1 - n.pos < L2_TILE_PAGES ? n.end_b1 - n.pos : L2_TILE_PAGES) : 0u;
        return n;
    };
    uint4 nx[PPT];                       // low parts: 4 x u32 or 8 x u16 per page
    uint2 nh[ElemFmt<IN>::HI ? PPT : 1]; // u24: 8 high bytes per page
    uint32_t nxvalid[PPT];
    PageEntry ent[PPT];                  // list entries of the tile whose pages are read next
    auto load_entries = [&](const TileIt &t) {
#pragma unroll
        for (int q = 0; q < PPT; q++) {
            const uint32_t pi = (uint32_t)q * NWAVES + (uint32_t)wave;  // wave w reads pages w, NWAVES + w, ... of a tile: a whole page per load instruction
            ent[q].page = 0; ent[q].nelems = 0;
            if (pi < t.npg) ent[q] = list1[t.pos + pi];
        }
    };
    auto load_pages = [&]() {
#pragma unroll
        for (int q = 0; q < PPT; q++) {
            const uint32_t first = (uint32_t)lane * (uint32_t)EP, ne = ent[q].nelems;
            nxvalid[q] = ne > first ? (ne - first < (uint32_t)EP ? ne - first : (uint32_t)EP) : 0u;
            nx[q] = make_uint4(0, 0, 0, 0);
            if (ElemFmt<IN>::HI) nh[q] = make_uint2(0, 0);
            if (ne) {                                                    // (a page of the list holds at least one element)
                const uint8_t *pg = pages1 + (size_t)ent[q].page * IN_PAGE_BYTES;
                nx[q] = reinterpret_cast<const uint4 *>(pg)[lane];
                if (ElemFmt<IN>::HI) nh[q] = reinterpret_cast<const uint2 *>(pg + SC_HI_OFFSET)[lane];
            }
        }
    };
    constexpr bool ENTRIES_AHEAD = ElemFmt<IN>::HI;
    TileIt A;
    A.pos = s0; A.b1 = l2_digit_of(page_base1, nb1, s0); A.valid = true;
    A.end_b1 = page_base1[A.b1 + 1] < s1 ? page_base1[A.b1 + 1] : s1;
    A.npg = A.end_b1 - A.pos < L2_TILE_PAGES ? A.end_b1 - A.pos : L2_TILE_PAGES;
    uint32_t cur_b1 = A.b1;
    load_entries(A);
    load_pages();
    TileIt B = tile_after(A);
    if (ENTRIES_AHEAD) load_entries(B);
    TileIt Cn = tile_after(B);
    SC_STAMP_INIT;
    SC_WG_CLOCK(1, 0);
    while (true) {
        // this tile's elements -> ring word offsets and 15/16-bit bins, one by one from the request loop of rings_place (the page
        // data stays in nx / nh until overlap() loads the next tile's over it: every element has been made by then)
        uint32_t pend = 0;
#pragma unroll
        for (int q = 0; q < PPT; q++) pend |= ((1u << nxvalid[q]) - 1u) << (q * EP);
        auto make = [&](int idx, uint32_t &woff_u, uint32_t &el_u) {
            const int q = idx / EP, i = idx % EP;
            const uint32_t lo[4] = {nx[q].x, nx[q].y, nx[q].z, nx[q].w};
            uint32_t e;
            if (ElemFmt<IN>::HI) {
                const uint32_t hb[2] = {nh[q].x, nh[q].y};
                e = ((lo[i >> 1] >> (16 * (i & 1))) & 0xFFFFu) | (((hb[i >> 2] >> (8 * (i & 3))) & 0xFFu) << 16);
            } else {
                e = lo[i & 3];
            }
            woff_u = __builtin_amdgcn_ubfe(e, (uint32_t)ring_shift, (uint32_t)ring_bits) << 2;
            el_u = bfi(keep, e, e >> ring_bits);
        };
        SC_STAMP(0);                                                     // (diagnostic build) wait for the tile's pages, ring words and bins
        if (A.b1 != cur_b1) {
            // the span passed into another digit: what the rings hold belongs to the old one
            if (my_ring < (uint32_t)RINGS) ring_drain(R, out, own, my_ring, (cur_b1 << ring_bits) | my_ring, ctr);
            cur_b1 = A.b1;
            __syncthreads();
        }
        const TileIt Dn = tile_after(Cn);
        rings_place<ELEM, RINGS, C, NID, NID>(R, out, own, my_ring, (cur_b1 << ring_bits) | my_ring, ctr, desc, make, pend, round, [&]() {
            if (!B.valid) return;
            if (ENTRIES_AHEAD) {
                load_pages();                                            // B's pages (entries in hand) ...
                // ... and B's entries are dead from here on: nxvalid[] is final before C's entries are read.  Without this the compiler
                // kept B's nelems alive past the loads below, read C's entries into other registers and copied them over right behind
                // the loads -- an `s_waitcnt vmcnt(0)` for the entries AND the pages in the middle of the request phase: every tile waited
                // for its successor's pages (round 5, found in the ISA; worth 0.7 % of the kernel: other waves filled the wait).
#pragma unroll
                for (int q = 0; q < PPT; q++) asm volatile("" : "+v"(nxvalid[q]));
                asm volatile("" ::: "memory");
                load_entries(Cn);                                        // C's entries
            }
            else { load_entries(B); load_pages(); }
        }, SC_STAMP_FN);
        if (!B.valid) break;
        A = B;
        B = Cn;
        Cn = Dn;
    }
    if (my_ring < (uint32_t)RINGS) ring_drain(R, out, own, my_ring, (cur_b1 << ring_bits) | my_ring, ctr);
    SC_STAMP(5);
    SC_STAMP_END_AT(16);
    SC_WG_CLOCK(1, 1);
}

// ---------------------------------------------------------------------------------
// page tags -> one page list per bucket (counting sort), bucket sizes, P2 slice table
// ---------------------------------------------------------------------------------
// pages per bucket and elements per bucket.  Every workgroup takes a contiguous chunk of page numbers and counts it
// in LDS first: global atomics on a few hundred addresses would serialise at the memory side (2.3 ms for 3 M pages,
// measured; 0.04 ms this way).  The LDS counters cover a WINDOW of PAGES_LDS_NB buckets that starts at the smallest
// bucket of the chunk: with up to 2^18 final buckets (two-level path) a chunk of the arena still holds the pages of a
// few workgroups, i.e. of a few leading digits (x 512 buckets each); a tag outside the window takes a global atomic.
constexpr int PAGES_LDS_NB = 4096;
constexpr int PAGES_THREADS = 1024;

// smallest bucket among the valid tags of [p0, p1) (block-wide; 0xFFFFFFFF if none)
__device__ __forceinline__ uint32_t pages_window_start(const uint32_t *__restrict__ tag, uint32_t p0, uint32_t p1, uint32_t *s_min)
{
    if (threadIdx.x == 0) *s_min = 0xFFFFFFFFu;
    __syncthreads();
    uint32_t m = 0xFFFFFFFFu;
    for (uint32_t p = p0 + threadIdx.x; p < p1; p += PAGES_THREADS) {
        const uint32_t t = tag[p];
        if (t != SC_NO_PAGE && (t & ((1u << SC_TAG_SHIFT) - 1u))) { const uint32_t b = t >> SC_TAG_SHIFT; m = b < m ? b : m; }
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) { const uint32_t x = (uint32_t)__shfl_xor((int)m, o, 64); m = x < m ? x : m; }
    if ((threadIdx.x & 63) == 0 && m != 0xFFFFFFFFu) atomicMin(s_min, m);
    __syncthreads();
    return *s_min;
}

__global__ void __launch_bounds__(PAGES_THREADS)
pages_count_kernel(const uint32_t *__restrict__ tag, uint32_t npages, uint32_t nb, uint32_t *__restrict__ bkt_pages, uint32_t *__restrict__ bkt_elems,
                   unsigned long long *__restrict__ stat /* {pages with elements, lines written into them} += ; may be null */, uint32_t line_elems)
{
    __shared__ uint32_t cp[PAGES_LDS_NB], ce[PAGES_LDS_NB];
    __shared__ uint32_t s_min, s_pages, s_lines;
    const uint32_t chunk = (npages + gridDim.x - 1) / gridDim.x, p0 = blockIdx.x * chunk, p1 = p0 + chunk < npages ? p0 + chunk : npages;
    const uint32_t w0 = pages_window_start(tag, p0, p1, &s_min);
    if (w0 == 0xFFFFFFFFu) return;
    for (uint32_t b = threadIdx.x; b < (uint32_t)PAGES_LDS_NB; b += PAGES_THREADS) { cp[b] = 0; ce[b] = 0; }
    if (threadIdx.x == 0) { s_pages = 0; s_lines = 0; }
    __syncthreads();
    uint32_t my_pages = 0, my_lines = 0;
    for (uint32_t p = p0 + threadIdx.x; p < p1; p += PAGES_THREADS) {
        const uint32_t t = tag[p];
        if (t == SC_NO_PAGE) continue;
        const uint32_t b = t >> SC_TAG_SHIFT, n = t & ((1u << SC_TAG_SHIFT) - 1u);
        if (n == 0) continue;
        my_pages++; my_lines += (n + line_elems - 1u) / line_elems;
        if (b - w0 < (uint32_t)PAGES_LDS_NB) { atomicAdd(&cp[b - w0], 1u); atomicAdd(&ce[b - w0], n); }
        else { atomicAdd(&bkt_pages[b], 1u); atomicAdd(&bkt_elems[b], n); }
    }
    if (stat) {
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) { my_pages += (uint32_t)__shfl_xor((int)my_pages, o, 64); my_lines += (uint32_t)__shfl_xor((int)my_lines, o, 64); }
        if ((threadIdx.x & 63) == 0 && my_pages) { atomicAdd(&s_pages, my_pages); atomicAdd(&s_lines, my_lines); }
    }
    __syncthreads();
    for (uint32_t b = threadIdx.x; b < (uint32_t)PAGES_LDS_NB && w0 + b < nb; b += PAGES_THREADS)
        if (cp[b]) { atomicAdd(&bkt_pages[w0 + b], cp[b]); atomicAdd(&bkt_elems[w0 + b], ce[b]); }
    if (stat && threadIdx.x == 0 && s_pages) {
        __hip_atomic_fetch_add(&stat[0], (unsigned long long)s_pages, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        __hip_atomic_fetch_add(&stat[1], (unsigned long long)s_lines, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
}

// exclusive scan of the page counts (one workgroup; nb <= 1024 * per-thread loop) -> page_base[nb + 1], P2 slice table
// (slices ~ pages), Sum(elements) -> total_kmers
__global__ void __launch_bounds__(1024)
pages_scan_kernel(const uint32_t *__restrict__ bkt_pages, const uint32_t *__restrict__ bkt_elems, uint32_t nb,
                  uint32_t *__restrict__ page_base, uint32_t *__restrict__ slice_base, uint32_t slice_pages, DevCounters *ctr)
{
    __shared__ uint32_t wsum[1024 / 64];
    __shared__ unsigned long long s_tot;
    const uint32_t j = threadIdx.x;
    if (j == 0) s_tot = 0;
    const uint32_t per = (nb + 1023u) / 1024u;
    const uint32_t lo = j * per < nb ? j * per : nb, hi = lo + per < nb ? lo + per : nb;
    uint32_t sum = 0, ssum = 0;
    unsigned long long el = 0;
    for (uint32_t i = lo; i < hi; i++) { const uint32_t v = bkt_pages[i]; sum += v; ssum += (v + slice_pages - 1) / slice_pages; el += bkt_elems[i]; }
    uint32_t tot, stot;
    uint32_t run = block_excl_scan<1024>(sum, wsum, &tot);
    uint32_t srun = block_excl_scan<1024>(ssum, wsum, &stot);
    for (uint32_t i = lo; i < hi; i++) {
        const uint32_t v = bkt_pages[i];
        page_base[i] = run; slice_base[i] = srun;
        run += v; srun += (v + slice_pages - 1) / slice_pages;
    }
    el = wave_sum(el);
    if ((j & 63) == 0 && el) atomicAdd(&s_tot, el);
    __syncthreads();
    if (j == 0) {
        page_base[nb] = tot; slice_base[nb] = stot;
        if (ctr && s_tot) __hip_atomic_fetch_add(&ctr->total_kmers, s_tot, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
}

// page p -> its place in the list of its bucket (bkt_pages counts down; the order inside a bucket does not matter).
// Same chunking and window: a workgroup reserves, per bucket, one range for all its pages of that bucket, and fills it
// from LDS cursors.
__global__ void __launch_bounds__(PAGES_THREADS)
pages_place_kernel(const uint32_t *__restrict__ tag, uint32_t npages, uint32_t nb, uint32_t *__restrict__ bkt_pages, const uint32_t *__restrict__ page_base,
                   PageEntry *__restrict__ list)
{
    __shared__ uint32_t cur[PAGES_LDS_NB];
    __shared__ uint32_t s_min;
    const uint32_t chunk = (npages + gridDim.x - 1) / gridDim.x, p0 = blockIdx.x * chunk, p1 = p0 + chunk < npages ? p0 + chunk : npages;
    const uint32_t w0 = pages_window_start(tag, p0, p1, &s_min);
    if (w0 == 0xFFFFFFFFu) return;
    for (uint32_t b = threadIdx.x; b < (uint32_t)PAGES_LDS_NB; b += PAGES_THREADS) cur[b] = 0;
    __syncthreads();
    for (uint32_t p = p0 + threadIdx.x; p < p1; p += PAGES_THREADS) {
        const uint32_t t = tag[p];
        if (t != SC_NO_PAGE && (t & ((1u << SC_TAG_SHIFT) - 1u)) && (t >> SC_TAG_SHIFT) - w0 < (uint32_t)PAGES_LDS_NB) atomicAdd(&cur[(t >> SC_TAG_SHIFT) - w0], 1u);
    }
    __syncthreads();
    for (uint32_t b = threadIdx.x; b < (uint32_t)PAGES_LDS_NB && w0 + b < nb; b += PAGES_THREADS) {
        const uint32_t c = cur[b];
        if (c) cur[b] = page_base[w0 + b] + atomicSub(&bkt_pages[w0 + b], c) - c;
    }
    __syncthreads();
    for (uint32_t p = p0 + threadIdx.x; p < p1; p += PAGES_THREADS) {
        const uint32_t t = tag[p];
        if (t == SC_NO_PAGE) continue;
        const uint32_t b = t >> SC_TAG_SHIFT, n = t & ((1u << SC_TAG_SHIFT) - 1u);
        if (n == 0) continue;
        const uint32_t pos = b - w0 < (uint32_t)PAGES_LDS_NB ? atomicAdd(&cur[b - w0], 1u) : page_base[b] + atomicSub(&bkt_pages[b], 1u) - 1u;
        list[pos].page = p;
        list[pos].nelems = n;
    }
}

// ---------------------------------------------------------------------------------
// P2 over page lists: one 32768-bin LDS histogram per (bucket, slice of its pages)
// ---------------------------------------------------------------------------------
// k = 17: 16-bit bins.  The 65536 counters of a bucket share the 32768 histogram words: bin v counts in half v >> 15 of
// word v & 0x7FFF, sixteen bits each, so the pages are read once.  A half that wraps (a bin seen 65536 times in one slice)
// stays exact: all atomics on a word are serialised, so exactly one adder sees the half at 0xFFFF; it notes "+65536 for
// this bin" in a short list that is added to the vector after the histogram.  A wrapping low half carries into the high
// half: the adder takes the carry back out (and notes the high half's own wrap / un-wrap if the carry or its removal
// crossed zero), so each half is the bin's count mod 65536 whenever no atomic is in flight.
constexpr uint32_t WRAP_MAX = 4096;      // >= 4 x elements of the largest slice / 65536 (pages_scan_kernel's slices hold < 2^26 elements)
template <uint32_t CAP> struct WrapListT { uint32_t n; uint32_t e[CAP]; };      // entry: bin | (1 << 16 if -65536 instead of +65536)
using WrapList = WrapListT<WRAP_MAX>;

__device__ __forceinline__ void wrap_note(WrapList &wl, uint32_t bin, uint32_t negative)
{
    const uint32_t s = atomicAdd(&wl.n, 1u);
    if (s < WRAP_MAX) wl.e[s] = bin | (negative << 16);
}

// what the word an add of 1 to bin v got back says (the rare part: the field stood at 0xFFFF)
__device__ __noinline__ void hist_wrapped16(uint32_t *hist, uint32_t v, uint32_t old, WrapList &wl)
{
    if (v >> 15) {
        if ((old >> 16) == 0xFFFFu) wrap_note(wl, v, 0u);
    } else if ((old & 0xFFFFu) == 0xFFFFu) {
        wrap_note(wl, v, 0u);
        if ((old >> 16) == 0xFFFFu) wrap_note(wl, v | 0x8000u, 0u);            // the carry wrapped the other half
        const uint32_t old2 = atomicSub(&hist[v & 0x7FFFu], 0x10000u);          // the carry does not belong there
        if ((old2 >> 16) == 0u) wrap_note(wl, v | 0x8000u, 1u);                 // ... and taking it out un-wrapped it
    }
}

// Eight 16-bit bins of a page chunk.  All eight returning atomics are issued before any returned word is looked at (one
// LDS round trip per chunk, not eight one after the other), each is ONE instruction whatever half the bin lives in
// (increment 1 << 16 * half -- a branch on the half made two instructions of it, each with half the lanes), and the
// test "did a field stand at 0xFFFF" is four operations per element: the word rotated so that the bin's field is the
// low one, + 1, XOR: bit 16 flips iff the field was all ones.  Round 3's form cost 1.5 ms per 1.39 G elements at k = 13
// against 0.6 ms for the 15-bit bins of k = 12.
// Round 4, second look at the code the compiler made of it: the per-element `e < nvalid` predicates and the run-time index into
// old[] of the rare path had put an exec-mask region, a dozen register copies and an `s_waitcnt lgkmcnt(0)` behind EVERY atomic
// (31 VALU instructions per element, the pass VALU-bound at 85 %).  Now: a lane whose eight elements are all there (all but one
// lane of a page's last line) runs straight-line code with compile-time indices only (the wrap test: bfe, max3 -- 1.5 instructions per element);
// the others take the loop below it.
__device__ __forceinline__ void hist_add_page_chunk16(uint32_t *hist, const uint4 &x, uint32_t nvalid, WrapList &wl)
{
    if (nvalid >= 8u) {
        const uint32_t w[4] = {x.x, x.y, x.z, x.w};
        uint32_t sh[8], old[8];
#pragma unroll
        for (uint32_t e = 0; e < 8; e++) {
            // byte address of the word (v & 0x7FFF) * 4 and the half's shift 16 * (v >> 15), two instructions each, straight from the pair
            const uint32_t byte = (e & 1u) ? (w[e >> 1] >> 14) & 0x1FFFCu : (w[e >> 1] << 2) & 0x1FFFCu;
            sh[e] = (e & 1u) ? (w[e >> 1] >> 27) & 16u : (w[e >> 1] >> 11) & 16u;
            old[e] = atomicAdd(reinterpret_cast<uint32_t *>(reinterpret_cast<uint8_t *>(hist) + byte), 1u << sh[e]);
        }
        uint32_t m = 0u;
#pragma unroll
        for (uint32_t e = 0; e < 8; e++) {
            const uint32_t fld = __builtin_amdgcn_ubfe(old[e], sh[e], 16u);                       // the bin's field as it stood (v_bfe_u32 at the half's shift)
            m = fld > m ? fld : m;
        }
        if (m == 0xFFFFu) {
#pragma unroll
            for (uint32_t e = 0; e < 8; e++) hist_wrapped16(hist, (e & 1u) ? w[e >> 1] >> 16 : w[e >> 1] & 0xFFFFu, old[e], wl);
        }
        return;
    }
    unsigned long long lo = ((unsigned long long)x.y << 32) | x.x, hi = ((unsigned long long)x.w << 32) | x.z;
#pragma unroll 1
    for (uint32_t e = 0; e < nvalid; e++) {
        const uint32_t v = (uint32_t)((e < 4u ? lo : hi) >> (16u * (e & 3u))) & 0xFFFFu;
        const uint32_t old = atomicAdd(&hist[v & 0x7FFFu], 1u << ((v >> 11) & 16u));
        hist_wrapped16(hist, v, old, wl);
    }
}

__device__ __forceinline__ void hist_add_page_chunk(uint32_t *hist, const uint4 &x, uint32_t nvalid)
{
    if (nvalid >= 8) { hist_add8(hist, x); return; }
    unsigned long long lo = ((unsigned long long)x.y << 32) | x.x, hi = ((unsigned long long)x.w << 32) | x.z;
    for (uint32_t e = 0; e < nvalid; e++) {
        const unsigned long long w = e < 4 ? lo : hi;
        atomicAdd(&hist[(uint32_t)(w >> (16 * (e & 3))) & 0xFFFFu], 1u);
    }
}

// add the LDS histogram of a bucket to the vector: histogram bin i = hi << lo_bits | low lives at  hi << hi_shift | bucket << lo_bits | low
// (dst already points at the bucket's first run): 2^(15 - lo_bits) runs of 2^lo_bits counters.  A lane takes two adjacent
// bins (16 bytes of the vector), so a wave-instruction moves 1 KiB of one run (lo_bits >= 7; 1 <= lo_bits always).
// (HALF: each word holds two 16-bit counters, see hist_add_page_chunk16: both go out in the same sweep)
// Returns the bytes of the vector this thread read + wrote.
// FLIGHT = read-modify-writes of 16 bytes a thread has in flight: four; eight where a word holds two counters (both halves go out in
// one sweep: k = 17 -2 %, k = 13 -2 % -- at k = 17 a workgroup spends three quarters of its time in this flush, 1 MiB of the vector
// per bucket against 0.27 MB of pages, one workgroup per CU; eight in flight for 15-bit bins made k = 15 4 % slower, same device).
template <bool HALF = false>
__device__ __forceinline__ uint32_t hist_flush_runs(const uint32_t *hist_words, unsigned long long *__restrict__ dst, int lo_bits, int hi_shift, bool only_writer,
                                                int tid, bool dst_is_zero, uint64_t half_stride = 0 /* HALF: where the bins of the high halves start, in counters */)
{
    const uint32_t lom = (1u << lo_bits) - 1u;
    constexpr int PAIRS = BUCKET_BINS / 2, NH = HALF ? 2 : 1, FLIGHT = HALF ? 8 : 4, U = FLIGHT / NH;
    // the counters of bins 2p, 2p + 1 (HALF: h = 0 the low halves of the two words, h = 1 the high halves: bins 2p + 32768, 2p + 32769)
    auto pair = [&](int p, int h) -> uint2 {
        uint2 w = reinterpret_cast<const uint2 *>(hist_words)[p];
        if (HALF) { w.x = (w.x >> (16 * h)) & 0xFFFFu; w.y = (w.y >> (16 * h)) & 0xFFFFu; }
        return w;
    };
    auto at = [&](int p, int h) -> ulonglong2 * {
        const uint32_t i = 2u * (uint32_t)p;
        return reinterpret_cast<ulonglong2 *>(dst + (HALF && h ? half_stride : 0ull) + (((uint64_t)(i >> lo_bits)) << hi_shift) + (i & lom));
    };
    uint32_t moved = 0;
    if (only_writer && dst_is_zero) {
        for (int base = 0; base < PAIRS; base += U * P2_THREADS) {
#pragma unroll
            for (int u = 0; u < U; u++) {
#pragma unroll
                for (int h = 0; h < NH; h++) {
                    const uint2 c = pair(base + u * P2_THREADS + tid, h);
                    *at(base + u * P2_THREADS + tid, h) = make_ulonglong2((unsigned long long)c.x, (unsigned long long)c.y);
                }
            }
        }
        moved = 16u * (uint32_t)(NH * PAIRS / P2_THREADS);
    } else if (only_writer) {
        for (int base = 0; base < PAIRS; base += U * P2_THREADS) {
            uint2 c[U][NH];
            ulonglong2 v[U][NH];
#pragma unroll
            for (int u = 0; u < U; u++)
#pragma unroll
                for (int h = 0; h < NH; h++) c[u][h] = pair(base + u * P2_THREADS + tid, h);
#pragma unroll
            for (int u = 0; u < U; u++)
#pragma unroll
                for (int h = 0; h < NH; h++) v[u][h] = (c[u][h].x | c[u][h].y) ? *at(base + u * P2_THREADS + tid, h) : make_ulonglong2(0ull, 0ull);
#pragma unroll
            for (int u = 0; u < U; u++)
#pragma unroll
                for (int h = 0; h < NH; h++)
                    if (c[u][h].x | c[u][h].y) {
                        v[u][h].x += c[u][h].x; v[u][h].y += c[u][h].y;
                        *at(base + u * P2_THREADS + tid, h) = v[u][h];
                        moved += 32u;
                    }
        }
    } else {
        for (int p = tid; p < PAIRS; p += P2_THREADS) {
#pragma unroll
            for (int h = 0; h < NH; h++) {
                const uint2 c = pair(p, h);
                unsigned long long *const a = reinterpret_cast<unsigned long long *>(at(p, h));
                if (c.x) { __hip_atomic_fetch_add(a, (unsigned long long)c.x, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); moved += 16u; }
                if (c.y) { __hip_atomic_fetch_add(a + 1, (unsigned long long)c.y, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); moved += 16u; }
            }
        }
    }
    return moved;
}

// BINS16 (k = 13, k = 17): the elements are 16-bit bins, two 16-bit counters per histogram word (hist_add_page_chunk16)
template <bool BINS16>
__global__ void __launch_bounds__(P2_THREADS)
page_hist_kernel(const uint8_t *__restrict__ pages, const PageEntry *__restrict__ list, const uint32_t *__restrict__ page_base,
                 const uint32_t *__restrict__ slice_base, uint32_t nbuckets, unsigned long long *__restrict__ table,
                 int lo_bits /* id bits below the bucket field: bucket b's first run starts at table + (b << lo_bits) */,
                 int hi_shift /* where the leading histogram bits sit in the id: lo_bits + all bucket bits */,
                 int table_is_zero /* host: the vector was cleared and no batch has been added to it since */, DevCounters *ctr)
{
    constexpr int CH = SC_PAGE_BYTES / 16;                // 16-byte chunks per page (64): one wave per page
    constexpr int PPS = P2_THREADS / CH;                  // pages per step (16)
    __shared__ uint32_t hist[BUCKET_BINS];
    __shared__ WrapListT<BINS16 ? WRAP_MAX : 1u> wl;
    __shared__ unsigned long long s_moved;
    const int tid = threadIdx.x;
    uint32_t b, s, nslices;
    if (!p2_locate(slice_base, nbuckets, blockIdx.x, &b, &s, &nslices)) return;
    const uint32_t P0 = page_base[b], n = page_base[b + 1] - P0;
    const uint32_t g0 = P0 + (uint32_t)((uint64_t)n * s / nslices), g1 = P0 + (uint32_t)((uint64_t)n * (s + 1) / nslices);
    if (g1 == g0) return;
    const uint32_t ch = (uint32_t)tid & (CH - 1), first = ch * 8u;
    for (int i = tid; i < BUCKET_BINS; i += P2_THREADS) hist[i] = 0;
    if (tid == 0) s_moved = 0;
    if (BINS16 && tid == 0) wl.n = 0;
    __syncthreads();
    uint32_t i = g0 + (uint32_t)tid / CH;
    for (; i + 3u * PPS < g1; i += 4u * PPS) {   // four pages in flight per wave
        PageEntry e[4];
        uint4 x[4];
#pragma unroll
        for (int u = 0; u < 4; u++) e[u] = list[i + u * PPS];
#pragma unroll
        for (int u = 0; u < 4; u++) x[u] = load_once16(reinterpret_cast<const uint4 *>(pages + (size_t)e[u].page * SC_PAGE_BYTES) + ch);
#pragma unroll
        for (int u = 0; u < 4; u++) {
            const uint32_t nv = e[u].nelems > first ? e[u].nelems - first : 0u;
            if constexpr (BINS16) hist_add_page_chunk16(hist, x[u], nv, wl); else hist_add_page_chunk(hist, x[u], nv);
        }
    }
    for (; i < g1; i += PPS) {
        const PageEntry e = list[i];
        const uint4 x = load_once16(reinterpret_cast<const uint4 *>(pages + (size_t)e.page * SC_PAGE_BYTES) + ch);
        const uint32_t nv = e.nelems > first ? e.nelems - first : 0u;
        if constexpr (BINS16) hist_add_page_chunk16(hist, x, nv, wl); else hist_add_page_chunk(hist, x, nv);
    }
    __syncthreads();
    unsigned long long *const dst = table + ((uint64_t)b << lo_bits);
    const bool zero = table_is_zero != 0 && ctr->table_dirty == 0;
    // bytes of the vector moved: one atomic per workgroup (through an LDS word that the histogram does not use)
    auto account = [&](uint32_t moved) {
        unsigned long long m = wave_sum((unsigned long long)moved);
        if ((tid & 63) == 0 && m) atomicAdd(&s_moved, m);
        __syncthreads();
        if (tid == 0 && s_moved) __hip_atomic_fetch_add(&ctr->table_bytes, s_moved, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    };
    if constexpr (!BINS16) { account(hist_flush_runs(hist, dst, lo_bits, hi_shift, nslices == 1, tid, zero)); return; }
    // the bin's leading bit sits above the 15 - lo_bits leading bits the histogram index holds: the high halves' bins start there
    account(hist_flush_runs<true>(hist, dst, lo_bits, hi_shift, nslices == 1, tid, zero, (uint64_t)(1u << (BIN_BITS - lo_bits)) << hi_shift));
    if (wl.n == 0) return;                                 // (block-uniform: written before the barrier above)
    // counters that wrapped: +- 65536 each, after this workgroup's own (possibly non-atomic) update of those bins has landed
    __threadfence();
    __syncthreads();
    const uint32_t nw = wl.n < WRAP_MAX ? wl.n : WRAP_MAX;
    if (tid == 0 && wl.n > WRAP_MAX) __hip_atomic_fetch_add(&ctr->internal_err, 1ull, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    const uint32_t lom = (1u << lo_bits) - 1u;
    for (uint32_t q = (uint32_t)tid; q < nw; q += P2_THREADS) {
        const uint32_t v = wl.e[q] & 0xFFFFu;
        const unsigned long long d = (wl.e[q] >> 16) ? 0ull - 65536ull : 65536ull;
        __hip_atomic_fetch_add(dst + ((uint64_t)(v >> lo_bits) << hi_shift) + (v & lom), d, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
}

// ---------------------------------------------------------------------------------
// host: 8 <= k <= 12
// ---------------------------------------------------------------------------------
struct ScatterState {
    uint8_t *d_pages = nullptr; size_t pages_cap = 0, page_bytes = 0;   // in pages of page_bytes
    uint32_t *d_tag = nullptr;                                  // [pages_cap]
    PageEntry *d_list = nullptr;                                // [pages_cap]
    uint32_t *d_bkt = nullptr;                                  // bkt_pages [NB] | bkt_elems [NB] | page_base [NB + 1] | slice_base [NB + 1]
    size_t bkt_cap = 0;                                         // NB the arrays were sized for
    int grid = 0;                                               // persistent workgroups (0 = SC_GRID)
    int lo_bits = 0;                                            // id bits below the bucket field; 0 = SC_LO_BITS_ONE_LEVEL / _TWO_LEVEL (SC_LO_BITS_MAX: buckets from the leading id bits, uneven in canonical mode)
    int contig_pages = 1;                                       // 1: workgroup w's pages are w * wg_pages + p (level 1 at k = 15: 2.26 -> 2.18 ms), 0: w + p * G
    int wide_lines = 1;                                         // 1: k <= 12 writes its pages in 128-byte pieces (engine option sc_wide_lines)
};

inline void scatter_free(ScatterState &st)
{
    if (st.d_pages) (void)hipFree(st.d_pages);
    if (st.d_tag) (void)hipFree(st.d_tag);
    if (st.d_list) (void)hipFree(st.d_list);
    if (st.d_bkt) (void)hipFree(st.d_bkt);
    st = ScatterState();
}

// tiles per launch: sub-batches of 2 Gi positions (page numbers stay well inside 32 bits), and fewer than 4096 tiles per
// workgroup (a thread's packed 16 + 16-bit statistics of <= 16 per tile cannot carry)
inline uint64_t scatter_max_tiles(uint32_t Gmax, uint32_t tile_pos = SC_TILE_POS)
{
    const uint64_t a = (1ull << 31) / tile_pos, b = 4095ull * Gmax;
    return a < b ? a : b;
}

// pages a workgroup can need: every element it can emit, one partial page per ring, one spare
inline uint32_t scatter_wg_pages(uint32_t tiles_per_wg, int rings, uint32_t page_elems, uint32_t tile_pos = SC_TILE_POS, uint32_t extra_elems = 0)
{
    return (uint32_t)(((uint64_t)tiles_per_wg * tile_pos + extra_elems + page_elems - 1) / page_elems) + (uint32_t)rings + 1u;
}

// EXPAND mode: fills of N-windows (4 or 16 per window) go through the rings like every other id; a workgroup's page sequence has
// room for half as many of them as it has window positions (0.5 % N at k = 12 makes a quarter), at least one full pass of
// 16 per thread; beyond that a workgroup adds its fills to the vector directly
inline uint32_t scatter_extra_elems(uint32_t tiles_per_wg, uint32_t tile_pos, int n_expand)
{
    if (!n_expand) return 0u;
    const uint64_t half = (uint64_t)tiles_per_wg * tile_pos / 2, floor_ = 16u * 1024u + 1024u * 64u + 2048u;      // (a round of 1024 threads, what the rings hold, slack)
    const uint64_t v = half > floor_ ? half : floor_;
    return (uint32_t)(v < 0x7FFFFFFFull ? v : 0x7FFFFFFFull);
}

inline int scatter_reserve(ScatterState &st, hipStream_t stream, size_t npages, size_t nb, size_t page_bytes = SC_PAGE_BYTES)
{
    if (st.pages_cap < npages || st.page_bytes != page_bytes) {
        if (st.d_pages) { if (hipStreamSynchronize(stream) != hipSuccess) return 1; (void)hipFree(st.d_pages); (void)hipFree(st.d_tag); (void)hipFree(st.d_list); st.d_pages = nullptr; st.d_tag = nullptr; st.d_list = nullptr; st.pages_cap = 0; }
        if (hipMalloc((void **)&st.d_pages, npages * page_bytes) != hipSuccess ||
            hipMalloc((void **)&st.d_tag, npages * sizeof(uint32_t)) != hipSuccess ||
            hipMalloc((void **)&st.d_list, npages * sizeof(PageEntry)) != hipSuccess) {
            (void)hipGetLastError();
            if (st.d_pages) (void)hipFree(st.d_pages);
            if (st.d_tag) (void)hipFree(st.d_tag);
            if (st.d_list) (void)hipFree(st.d_list);
            st.d_pages = nullptr; st.d_tag = nullptr; st.d_list = nullptr;
            return 2;
        }
        st.pages_cap = npages;
        st.page_bytes = page_bytes;
    }
    if (st.bkt_cap < nb) {
        if (st.d_bkt) { if (hipStreamSynchronize(stream) != hipSuccess) return 1; (void)hipFree(st.d_bkt); st.d_bkt = nullptr; st.bkt_cap = 0; }
        if (hipMalloc((void **)&st.d_bkt, (4 * nb + 2) * sizeof(uint32_t)) != hipSuccess) { (void)hipGetLastError(); return 2; }
        st.bkt_cap = nb;
    }
    return 0;
}

// k = 13 in ONE level: 26 id bits = 10 bucket bits + 16 bin bits.  1024 rings of 64 u16 elements (128 KiB of LDS) leave room for
// one workgroup per CU, so that workgroup has 1024 threads (16 waves per CU, as two workgroups of 512 have) and a tile of 1023
// chunks; a bucket's 65536 bins share the 32768 histogram words as two 16-bit counters each (page_hist_kernel<true>, as at
// k = 17).  Level 2 of the two-level path existed at k = 13 only to split one more bit pair: 4.48 -> see DESIGN.md section 5.
constexpr int SC1_THREADS = 1024, SC1_RINGS = 1024, SC1_GRID = 256, SC1_K = 13;
constexpr int SC1_TILE_POS = (SC1_THREADS - 1) * 16;

// The one-level path in two stages, so that they can run on different streams: stage 1 = the scatter kernel of a sub-batch into the
// pages of `st`, stage 2 = the page sort and the histogram pass over those pages.  ScGeom: what the host works out once per batch.
struct ScGeom {
    bool big; bool wide /* k <= 12 in 128-byte pieces: one workgroup of 1024 threads per CU, 512 rings of 128 elements (u16w) */;
    int binb, nb, rings, sub_log2, nb_bits, lo_bits, hi_shift;
    uint32_t tile_stride, tile_pos, Gmax;
    uint64_t ntiles_all, max_tiles;
};
struct ScLaunch { uint32_t nt, G, npages; };

inline ScGeom scatter_geometry(const ScatterState &st, size_t nbytes, int k, uint32_t grid_default = 0)
{
    ScGeom g;
    g.big = k == SC1_K;                                                  // 1024 threads, 1024 rings, 16-bit bins
    g.binb = g.big ? 16 : BIN_BITS;
    g.nb = 1 << (2 * k - g.binb);                                        // buckets: 2 (k = 8) .. 512 (k = 12), 1024 (k = 13)
    g.rings = g.big ? SC1_RINGS : 512;
    g.lo_bits = st.lo_bits ? st.lo_bits : SC_LO_BITS_ONE_LEVEL;          // (lo_bits = 15: bucket = leading id bits, for comparison)
    g.wide = !g.big && st.wide_lines != 0;
    g.tile_stride = (g.big || g.wide) ? (uint32_t)SC1_THREADS - 1u : (uint32_t)SC_TILE_STRIDE;
    g.tile_pos = g.tile_stride * 16u;
    g.sub_log2 = 0; g.nb_bits = 2 * k - g.binb;
    while ((g.nb << g.sub_log2) < g.rings) g.sub_log2++;                 // few buckets: each gets several rings (no same-address pile-up)
    g.hi_shift = g.lo_bits + g.nb_bits;
    g.ntiles_all = ((nbytes + 15) / 16 + g.tile_stride - 1) / g.tile_stride;
    g.Gmax = st.grid > 0 ? (uint32_t)st.grid : (grid_default ? grid_default : (uint32_t)((g.big || g.wide) ? SC1_GRID : SC_GRID));
    g.max_tiles = scatter_max_tiles(g.Gmax, g.tile_pos);
    return g;
}

// scratch for the largest sub-batch; 0 ok, 1 stream error, 2 no room
inline int scatter_reserve_for(ScatterState &st, hipStream_t stream, const ScGeom &g, int n_expand)
{
    const uint64_t nt = g.ntiles_all < g.max_tiles ? g.ntiles_all : g.max_tiles;
    const uint32_t G = (uint32_t)(nt < g.Gmax ? nt : g.Gmax);
    const uint32_t tpw = (uint32_t)((nt + G - 1) / G);
    const uint32_t wg_pages = scatter_wg_pages(tpw, g.rings, 512, g.tile_pos, scatter_extra_elems(tpw, g.tile_pos, n_expand));
    const int rc = scatter_reserve(st, stream, (size_t)G * wg_pages, (size_t)g.nb);
    if (rc == 2) { partition_error_ref() = "scratch allocation failed"; return 2; }
    if (rc) { partition_error_ref() = "stream error"; return 1; }
    return 0;
}

inline int scatter_stage1(ScatterState &st, hipStream_t stream, const ScGeom &g, uint64_t t0, const uint8_t *d_bases, size_t nbytes, const RecStarts &rs, int k,
                          int canonical, int n_expand, unsigned long long *d_table, DevCounters *d_ctr, ProfHook &prof, ScLaunch *L)
{
    constexpr int C = 64;
    const int nb = g.nb, lo_bits = g.lo_bits, nb_bits = g.nb_bits, sub_log2 = g.sub_log2;
    const uint32_t nt = (uint32_t)((g.ntiles_all - t0) < g.max_tiles ? (g.ntiles_all - t0) : g.max_tiles);
    const uint32_t G = nt < g.Gmax ? nt : g.Gmax;
    ScOut out;
    out.pages = st.d_pages; out.tag = st.d_tag;
    out.extra_elems = scatter_extra_elems((nt + G - 1) / G, g.tile_pos, n_expand);
    out.wg_pages = scatter_wg_pages((nt + G - 1) / G, g.rings, 512, g.tile_pos, out.extra_elems);
    out.wg_range = nullptr; out.contig = (uint32_t)st.contig_pages; out.wg_base = 0; out.grid = 0;
    const uint32_t npages = G * out.wg_pages;
    L->nt = nt; L->G = G; L->npages = npages;
    if (hipMemsetAsync(st.d_tag, 0xFF, (size_t)npages * sizeof(uint32_t), stream) != hipSuccess ||
        hipMemsetAsync(st.d_bkt, 0, 2 * (size_t)nb * sizeof(uint32_t), stream) != hipSuccess) { partition_error_ref() = "memset failed"; return 1; }
    prof.begin_on(KDB_KERNEL_SCATTER, stream);
#define KDB_LAUNCH_SC1(EL, CC, E, CN, KK, RG, TH, RAG)                                                                                     \
    hipLaunchKernelGGL((scatter_bases_kernel<uint32_t, EL, RG, CC, 16, E, CN, KK, TH, RAG>), dim3(G), dim3(TH), 0, stream, d_bases,       \
                       (uint64_t)nbytes, (uint32_t)t0, nt, k, lo_bits, nb_bits, sub_log2, out, d_table, d_ctr, rs)
#define KDB_LAUNCH_SC(EL, CC, E, CN, KK, RG, TH) do { KDB_LAUNCH_SC1(EL, CC, E, CN, KK, RG, TH, false); KDB_LAUNCH_SC1(EL, CC, E, CN, KK, RG, TH, true); } while (0)
#define KDB_LAUNCH_SC_MODES(EL, CC, KK, RG, TH)                                                                                            \
    do {                                                                                                                                   \
        if (n_expand) { if (canonical) KDB_LAUNCH_SC(EL, CC, true, true, KK, RG, TH); else KDB_LAUNCH_SC(EL, CC, true, false, KK, RG, TH); }   \
        else          { if (canonical) KDB_LAUNCH_SC(EL, CC, false, true, KK, RG, TH); else KDB_LAUNCH_SC(EL, CC, false, false, KK, RG, TH); } \
    } while (0)
    if (g.big) {
        if (lo_bits == SC_LO_BITS_ONE_LEVEL) KDB_LAUNCH_SC_MODES(uint16_t, C, SC1_K, SC1_RINGS, SC1_THREADS);      // shifts and masks compiled in
        else                                 KDB_LAUNCH_SC_MODES(uint16_t, C, 0, SC1_RINGS, SC1_THREADS);
    } else if (g.wide) {
        // 128-byte pieces: one workgroup of 1024 threads per CU, 512 rings of 128 elements
        // (shifts, masks and rings per bucket compiled in for the default bucket field: two dozen scalar registers stay free)
        const bool compiled = lo_bits == SC_LO_BITS_ONE_LEVEL && (nb << sub_log2) == 512;
        if (compiled && k == 12)      KDB_LAUNCH_SC_MODES(u16w, 128, 12, 512, SC1_THREADS);                                          // BASELINE's headline k
        else if (compiled && k == 11) KDB_LAUNCH_SC_MODES(u16w, 128, 11, 512, SC1_THREADS);
        else if (compiled && k == 10) KDB_LAUNCH_SC_MODES(u16w, 128, 10, 512, SC1_THREADS);
        else if (compiled && k == 9)  KDB_LAUNCH_SC_MODES(u16w, 128, 9, 512, SC1_THREADS);
        else                          KDB_LAUNCH_SC_MODES(u16w, 128, 0, 512, SC1_THREADS);
    } else if (k == 12 && lo_bits == SC_LO_BITS_ONE_LEVEL && sub_log2 == 0) {
        KDB_LAUNCH_SC_MODES(uint16_t, C, 12, 512, SC_THREADS);                                         // ... in 64-byte lines, two workgroups per CU
    } else {
        KDB_LAUNCH_SC_MODES(uint16_t, C, 0, 512, SC_THREADS);
    }
#undef KDB_LAUNCH_SC_MODES
#undef KDB_LAUNCH_SC
#undef KDB_LAUNCH_SC1
    prof.end();
    if (hipGetLastError() != hipSuccess) { partition_error_ref() = "paged scatter failed to launch"; return 1; }
    return 0;
}

inline int scatter_stage2(ScatterState &st, hipStream_t stream, const ScGeom &g, const ScLaunch &L, unsigned long long *d_table, DevCounters *d_ctr, ProfHook &prof)
{
    const int nb = g.nb;
    const uint32_t npages = L.npages;
    uint32_t *const bkt_pages = st.d_bkt, *const bkt_elems = st.d_bkt + nb, *const page_base = st.d_bkt + 2 * nb, *const slice_base = st.d_bkt + 3 * nb + 1;
    prof.begin_on(KDB_KERNEL_PAGE_SORT, stream);
    const uint32_t pgrid = (npages + 4095u) / 4096u < 256u ? (npages + 4095u) / 4096u : 256u;
    const uint32_t target = 512u;                                    // P2 workgroups in all (fewer, larger slices win: single-slice buckets flush without atomics)
    const uint32_t est_pages = (uint32_t)(((uint64_t)L.nt * g.tile_pos * 2) / SC_PAGE_BYTES) + 1u;
    uint32_t slice_pages = (est_pages + target - 1) / target;
    if (slice_pages < 128u) slice_pages = 128u;                      // >= 64 Ki elements per histogram
    hipLaunchKernelGGL(pages_count_kernel, dim3(pgrid), dim3(PAGES_THREADS), 0, stream, (const uint32_t *)st.d_tag, npages, (uint32_t)nb, bkt_pages, bkt_elems,
                       &d_ctr->pages_bases, 32u);
    hipLaunchKernelGGL(pages_scan_kernel, dim3(1), dim3(1024), 0, stream, (const uint32_t *)bkt_pages, (const uint32_t *)bkt_elems, (uint32_t)nb,
                       page_base, slice_base, slice_pages, d_ctr);
    hipLaunchKernelGGL(pages_place_kernel, dim3(pgrid), dim3(PAGES_THREADS), 0, stream, (const uint32_t *)st.d_tag, npages, (uint32_t)nb, bkt_pages,
                       (const uint32_t *)page_base, st.d_list);
    prof.end();
    prof.begin_on(KDB_KERNEL_PAGE_HIST, stream);
    const uint32_t p2_grid = npages / slice_pages + (uint32_t)nb + 1u;
    if (g.big)
        hipLaunchKernelGGL(page_hist_kernel<true>, dim3(p2_grid), dim3(P2_THREADS), 0, stream, (const uint8_t *)st.d_pages, (const PageEntry *)st.d_list,
                           (const uint32_t *)page_base, (const uint32_t *)slice_base, (uint32_t)nb, d_table, g.lo_bits, g.hi_shift, 0, d_ctr);
    else
        hipLaunchKernelGGL(page_hist_kernel<false>, dim3(p2_grid), dim3(P2_THREADS), 0, stream, (const uint8_t *)st.d_pages, (const PageEntry *)st.d_list,
                           (const uint32_t *)page_base, (const uint32_t *)slice_base, (uint32_t)nb, d_table, g.lo_bits, g.hi_shift, 0, d_ctr);
    prof.end();
    if (hipGetLastError() != hipSuccess) { partition_error_ref() = "paged scatter failed to launch"; return 1; }
    return 0;
}

// returns 0 ok, 1 error (partition_error()), 2 no room for the scratch (nothing was counted)
inline int scatter_count(ScatterState &st, hipStream_t stream, const uint8_t *d_bases, size_t nbytes, const RecStarts &rs, int k, int canonical, int n_expand,
                         unsigned long long *d_table, DevCounters *d_ctr, ProfHook &prof)
{
    const ScGeom g = scatter_geometry(st, nbytes, k);
    { const int rc = scatter_reserve_for(st, stream, g, n_expand); if (rc) return rc; }
    for (uint64_t t0 = 0; t0 < g.ntiles_all; t0 += g.max_tiles) {
        ScLaunch L;
        if (scatter_stage1(st, stream, g, t0, d_bases, nbytes, rs, k, canonical, n_expand, d_table, d_ctr, prof, &L)) return 1;
        if (scatter_stage2(st, stream, g, L, d_table, d_ctr, prof)) return 1;
    }
    return 0;
}

// ---------------------------------------------------------------------------------
// The one-level path with the two stages of CONSECUTIVE batches side by side: the scatter kernel is bound by
// VALU issue, the histogram pass by HBM and LDS atomics, and on one stream they take turns.  Two sets of pages; batch i's stage 2 runs
// on a second stream while batch i + 1's stage 1 runs on the first.  The histogram passes stay in order (one stream: their plain
// read-modify-writes of the vector never meet each other), and a scatter kernel that runs beside a pass leaves its degenerate ids in
// a side list (hot_add) that is added behind the pass of its own batch.  DROP mode and batches of one sub-batch only (the callers fall
// back to scatter_count otherwise).  CU masks on the two streams (hipExtStreamCreateWithCUMask) give each stage its own CUs: both stages
// want most of a CU's LDS, so without masks a pass only finds room where the scatter kernel's persistent workgroups have ended.
// ---------------------------------------------------------------------------------
struct OverlapState {
    ScatterState sc[2];
    unsigned long long *side[2] = {nullptr, nullptr};
    size_t side_cap = 0;                                                  // pairs
    hipEvent_t scattered[2] = {nullptr, nullptr}, hist_done[2] = {nullptr, nullptr};
    bool used[2] = {false, false};
    int next = 0, last = -1;                                              // set of the next batch; set whose pass was launched last (-1: none outstanding)
    uint32_t grid = 0;                                                    // persistent scatter workgroups (0: the usual number)
};

inline void overlap_free(OverlapState &ov)
{
    for (int j = 0; j < 2; j++) {
        scatter_free(ov.sc[j]);
        if (ov.side[j]) (void)hipFree(ov.side[j]);
        if (ov.scattered[j]) (void)hipEventDestroy(ov.scattered[j]);
        if (ov.hist_done[j]) (void)hipEventDestroy(ov.hist_done[j]);
    }
    ov = OverlapState();
}

// 0 ok, 1 error, 2 no room, 3 this batch does not fit the overlapped form (several sub-batches): count it with scatter_count
inline int scatter_count_overlapped(OverlapState &ov, hipStream_t s_scatter, hipStream_t s_hist, const uint8_t *d_bases, size_t nbytes, const RecStarts &rs, int k,
                                    int canonical, unsigned long long *d_table, DevCounters *d_ctr, ProfHook &prof)
{
    const int j = ov.next;
    ScatterState &st = ov.sc[j];
    st.grid = ov.sc[0].grid; st.lo_bits = ov.sc[0].lo_bits; st.contig_pages = ov.sc[0].contig_pages; st.wide_lines = ov.sc[0].wide_lines;
    const ScGeom g = scatter_geometry(st, nbytes, k, ov.grid);
    if (g.ntiles_all > g.max_tiles) return 3;
    for (int q = 0; q < 2; q++) {
        if (!ov.scattered[q] && hipEventCreateWithFlags(&ov.scattered[q], hipEventDisableTiming) != hipSuccess) { partition_error_ref() = "event creation failed"; return 1; }
        if (!ov.hist_done[q] && hipEventCreateWithFlags(&ov.hist_done[q], hipEventDisableTiming) != hipSuccess) { partition_error_ref() = "event creation failed"; return 1; }
    }
    // this set's pages are read by the pass of the batch before last: the scatter stream waits for it (and a reallocation drains it)
    if (ov.used[j] && hipStreamWaitEvent(s_scatter, ov.hist_done[j], 0) != hipSuccess) { partition_error_ref() = "stream wait failed"; return 1; }
    {
        const uint64_t nt = g.ntiles_all;
        const uint32_t G = (uint32_t)(nt < g.Gmax ? nt : g.Gmax);
        const uint32_t tpw = (uint32_t)((nt + G - 1) / G);
        const size_t need = (size_t)G * scatter_wg_pages(tpw, g.rings, 512, g.tile_pos, 0);
        if ((st.pages_cap < need || st.bkt_cap < (size_t)g.nb) && ov.used[j] && hipEventSynchronize(ov.hist_done[j]) != hipSuccess) { partition_error_ref() = "stream error"; return 1; }
        const int rc = scatter_reserve_for(st, s_scatter, g, 0);
        if (rc) return rc;
        const size_t side_need = (size_t)g.Gmax * SC_HOT + 65536;
        if (ov.side_cap < side_need) {
            for (int q = 0; q < 2; q++) {
                if (ov.side[q]) { if (ov.used[q] && hipEventSynchronize(ov.hist_done[q]) != hipSuccess) { partition_error_ref() = "stream error"; return 1; } (void)hipFree(ov.side[q]); ov.side[q] = nullptr; }
                if (hipMalloc((void **)&ov.side[q], (2 + 2 * side_need) * sizeof(unsigned long long)) != hipSuccess) { (void)hipGetLastError(); ov.side_cap = 0; partition_error_ref() = "scratch allocation failed"; return 2; }
            }
            ov.side_cap = side_need;
        }
    }
    hipLaunchKernelGGL(hot_side_kernel, dim3(1), dim3(1), 0, s_scatter, d_ctr, ov.side[j], (unsigned long long)ov.side_cap);
    ScLaunch L;
    if (scatter_stage1(st, s_scatter, g, 0, d_bases, nbytes, rs, k, canonical, 0, d_table, d_ctr, prof, &L)) return 1;
    hipLaunchKernelGGL(hot_side_kernel, dim3(1), dim3(1), 0, s_scatter, d_ctr, (unsigned long long *)nullptr, 0ull);      // (whatever runs next on this stream adds directly again)
    if (hipEventRecord(ov.scattered[j], s_scatter) != hipSuccess || hipStreamWaitEvent(s_hist, ov.scattered[j], 0) != hipSuccess) { partition_error_ref() = "stream error"; return 1; }
    if (scatter_stage2(st, s_hist, g, L, d_table, d_ctr, prof)) return 1;
    hipLaunchKernelGGL(apply_hot_kernel, dim3(8), dim3(256), 0, s_hist, (const unsigned long long *)ov.side[j], d_table);
    if (hipGetLastError() != hipSuccess || hipEventRecord(ov.hist_done[j], s_hist) != hipSuccess) { partition_error_ref() = "paged scatter failed to launch"; return 1; }
    ov.used[j] = true;
    ov.last = j;
    ov.next = j ^ 1;
    return 0;
}

// ---------------------------------------------------------------------------------
// host: 13 <= k <= 17, two levels.   id = [ hi : 9 (k = 17: 10) ][ d1 : 2k - 24 (k = 17: 9) ][ d2 : 9 ][ lo : 6 ]
//   level 1  scatter_bases_kernel   residues -> u32 remainders (id without d1), pages tagged d1
//   level 2  scatter_ids_kernel     those pages -> u16 bins (hi | lo), pages tagged d1 << 9 | d2, into an ARENA that
//                                   several batches share
//   flush    one counting sort of the arena's tags, one page_hist_kernel: the sweep over the 4^k vector (8 GiB at
//            k = 15, 128 GiB at k = 17: more than everything else in a batch) is paid once per <= PAGED_PENDING_MAX batches,
//            at kdb_sync / kdb_finish, or when the arena is full -- not once per batch
// ---------------------------------------------------------------------------------
constexpr int PAGED_PENDING_MAX = 64;
// a level-1 ring must take the arrivals of a round (8176 ids x ROUND / 16 / rings, spread evenly by the mid-bit digits) on
// top of an incomplete line:
constexpr int L1_RINGS = 256, L1_C = 64, L1_ROUND = 8;      // k <= 16 (<= 256 digits), u24 elements (64 KiB of LDS): 16 arrivals a round, two flush rounds per tile
// k = 17 (512 digits), u32 elements (64 KiB): 8 arrivals a round on top of < 16 left over, two rounds per tile.  (Four rounds of 4 never
// refuse an element but pay four barrier pairs: level 1 2.46-2.54 ms; two rounds repeat one for ~0.02 % of the rings: 2.09 ms.)
constexpr int L1W_RINGS = 512, L1W_C = 32, L1W_ROUND = 8;
// (128 rings x 128 elements with one round per tile: 2.30 ms as u32, 2.6-2.7 ms as u24 against 2.28 ms for 256 x 64 -- measured, k = 13..15)

struct TwoLevelPaged {
    ScatterState l1;                       // level-1 pages / tags / list (reused by every batch)
    uint8_t *d_pages2 = nullptr; size_t cap2 = 0;      // the arena, in pages
    uint32_t *d_tag2 = nullptr;
    PageEntry *d_list2 = nullptr;
    uint32_t *d_bkt2 = nullptr; size_t nb2_cap = 0;
    uint32_t *d_wg_range = nullptr;        // [SC_GRID + 1] of the batch being scattered
    size_t used2 = 0;                      // arena pages the pending batches can have taken at most (their worst cases added up)
    // What they really took is only known on the device: l2_plan_kernel hands every batch the pages behind the previous one's
    // last (d_cursor), not the host's worst case -- round 3 charged a fixed ~0.4 GiB per batch, four times the payload of a
    // 64 MiB host-fed chunk.  The host learns the cursor by asynchronous read-backs (a few probes in flight, polled before
    // every batch) and then only has to assume the worst for the batches behind the last probe that has landed.
    static constexpr int PROBES = 8;
    uint32_t *d_cursor = nullptr;
    uint32_t *h_probe = nullptr;           // pinned [PROBES]
    hipEvent_t ev_probe[PROBES] = {nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr};
    size_t probe_bound[PROBES] = {0};      // used2 (worst cases added up) when the probe was sent
    bool probe_live[PROBES] = {false};
    int probe_next = 0;
    size_t slack = 0;                      // used2 - slack = the host's present bound on the cursor
    size_t tags_dirty = 0;                 // arena tags [0, tags_dirty) may have been written since they were last cleared
    int pending = 0;                       // batches in the arena
    int k_pending = 0;
    int defer = 1;
    size_t budget_bytes = 0;               // arena size; 0 = decide at first use (85 % of the free memory, less reserve_bytes)
    int l1_one_round = 1;                  // 1: level 1 of k <= 15 (<= 64 leading digits) with 128 rings of 256 elements: one placement round per tile (engine option l1_one_round)
    int l1_wide = 1;                       // 1: level 1 writes its pages in 128-byte pieces, one workgroup of 1024 threads per CU (engine option l1_wide_lines)
    int l2_wide = 1;                       // 1: level 2 writes its pages in 128-byte pieces, one workgroup of 1024 threads per CU (engine option l2_wide_lines)
    int l1k = 1;                           // 1: k = 15 (canonical, DROP) runs level 1's kernel compiled for that k; 0: the generic one (comparison)
    size_t reserve_bytes = 0;              // device memory the arena must leave free whatever it grows to (RCCL's buffers and the reduce's scratch: kmerdb_amd/distributed.py)
    size_t free_at_sizing = 0;             // what hipMemGetInfo reported when the budget was decided
    bool table_is_zero = false;            // the engine cleared the vector and nothing has been added since
    bool filled_up = false;                // the last flush came because the arena was full
    bool grow_failed = false;              // a larger arena could not be allocated: no further attempts
    uint64_t reallocs = 0;                 // (re)allocations of the arena so far
    int grow = 1;                          // 0: the arena keeps its first size; 1: it doubles when that pays (below); 2: whenever it has filled up
    uint64_t full_flushes = 0;             // flushes forced by a full arena since it got its present size
    int first_batches = 8;                 // the arena's first size, in batches like the first one (engine option arena_batches)
    uint64_t flushes = 0, flushed_batches = 0;     // histogram passes over the arena so far, and the batches they added to the vector
};

inline void twolevel_paged_free(TwoLevelPaged &tp)
{
    scatter_free(tp.l1);
    if (tp.d_pages2) (void)hipFree(tp.d_pages2);
    if (tp.d_tag2) (void)hipFree(tp.d_tag2);
    if (tp.d_list2) (void)hipFree(tp.d_list2);
    if (tp.d_bkt2) (void)hipFree(tp.d_bkt2);
    if (tp.d_wg_range) (void)hipFree(tp.d_wg_range);
    if (tp.d_cursor) (void)hipFree(tp.d_cursor);
    if (tp.h_probe) (void)hipHostFree(tp.h_probe);
    for (int i = 0; i < TwoLevelPaged::PROBES; i++) if (tp.ev_probe[i]) (void)hipEventDestroy(tp.ev_probe[i]);
    const int defer = tp.defer, grow = tp.grow, first_batches = tp.first_batches, l1k = tp.l1k, l2_wide = tp.l2_wide, l1_wide = tp.l1_wide, l1_one_round = tp.l1_one_round;
    const size_t budget = tp.budget_bytes, reserve = tp.reserve_bytes;
    const ScatterState keep = tp.l1;
    tp = TwoLevelPaged();
    tp.defer = defer; tp.budget_bytes = budget; tp.reserve_bytes = reserve; tp.grow = grow; tp.first_batches = first_batches; tp.l1k = l1k; tp.l2_wide = l2_wide; tp.l1_wide = l1_wide; tp.l1_one_round = l1_one_round;
    tp.l1.grid = keep.grid; tp.l1.lo_bits = keep.lo_bits; tp.l1.contig_pages = keep.contig_pages;
}

// kdb_reset / after a flush: no batch is pending any more (the cursor and the tags are cleared when the next cycle begins)
inline void twolevel_paged_drop(TwoLevelPaged &tp)
{
    if (tp.used2 - tp.slack > tp.tags_dirty) tp.tags_dirty = tp.used2 - tp.slack;
    tp.used2 = 0; tp.slack = 0; tp.pending = 0;
    for (int i = 0; i < TwoLevelPaged::PROBES; i++) tp.probe_live[i] = false;
}

// what the device has told the host about the cursor so far: the newest probe that has landed decides
inline void twolevel_paged_poll(TwoLevelPaged &tp)
{
    for (int i = 0; i < TwoLevelPaged::PROBES; i++) {
        if (!tp.probe_live[i] || hipEventQuery(tp.ev_probe[i]) != hipSuccess) continue;
        tp.probe_live[i] = false;
        // the cursor stood at `actual` when the worst cases added up to probe_bound: every batch since adds at most its worst case to both
        const size_t actual = tp.h_probe[i], bound = tp.probe_bound[i];
        if (bound >= actual && bound - actual > tp.slack) tp.slack = bound - actual;
    }
    (void)hipGetLastError();               // (hipErrorNotReady of a probe still in flight is no error)
}

inline void paged_bits(int k, int *d1_bits, int *bin_bits)
{
    *bin_bits = k == 17 ? 16 : 15;
    *d1_bits = 2 * k - *bin_bits - 9;                                   // 2, 4, 6, 8 (k = 13..16), 9 (k = 17)
}

// the histogram pass over everything in the arena
inline int twolevel_paged_flush(TwoLevelPaged &tp, hipStream_t stream, unsigned long long *d_table, DevCounters *d_ctr, ProfHook &prof)
{
    if (tp.pending == 0) return 0;
    const int k = tp.k_pending;
    int d1, binb;
    paged_bits(k, &d1, &binb);
    const uint32_t nb2 = 1u << (d1 + 9);
    const int table_is_zero = tp.table_is_zero ? 1 : 0;
    tp.table_is_zero = false;
    uint32_t *const bkt_pages = tp.d_bkt2, *const bkt_elems = tp.d_bkt2 + nb2, *const page_base = tp.d_bkt2 + 2 * (size_t)nb2,
             *const slice_base = tp.d_bkt2 + 3 * (size_t)nb2 + 1;
    twolevel_paged_poll(tp);
    const uint32_t npages = (uint32_t)(tp.used2 - tp.slack);            // (an upper bound on the cursor: tags behind it say "no page")
    if (npages == 0) {
        // the device's cursor has told the host that the pending batches took no page at all (not one countable window among them: reads of
        // N's in drop mode): nothing to sort, nothing to add -- and a launch with an empty grid is an error (found by tests/fuzz_gpu.py, round 4)
        tp.flushes++; tp.flushed_batches += (uint64_t)tp.pending;
        twolevel_paged_drop(tp);
        return 0;
    }
    if (hipMemsetAsync(tp.d_bkt2, 0, 2 * (size_t)nb2 * sizeof(uint32_t), stream) != hipSuccess) { partition_error_ref() = "memset failed"; return 1; }
    prof.begin(KDB_KERNEL_PAGE_SORT);
    const uint32_t pgrid = (npages + 4095u) / 4096u < 2048u ? (npages + 4095u) / 4096u : 2048u;     // (small chunks: few leading digits per LDS window)
    uint32_t slice_pages = (npages + 2047u) / 2048u;
    if (slice_pages < 128u) slice_pages = 128u;
    hipLaunchKernelGGL(pages_count_kernel, dim3(pgrid), dim3(PAGES_THREADS), 0, stream, (const uint32_t *)tp.d_tag2, npages, nb2, bkt_pages, bkt_elems,
                       &d_ctr->pages_ids, 32u);
    hipLaunchKernelGGL(pages_scan_kernel, dim3(1), dim3(1024), 0, stream, (const uint32_t *)bkt_pages, (const uint32_t *)bkt_elems, nb2, page_base, slice_base,
                       slice_pages, (DevCounters *)nullptr);
    hipLaunchKernelGGL(pages_place_kernel, dim3(pgrid), dim3(PAGES_THREADS), 0, stream, (const uint32_t *)tp.d_tag2, npages, nb2, bkt_pages,
                       (const uint32_t *)page_base, tp.d_list2);
    prof.end();
    prof.begin(KDB_KERNEL_PAGE_HIST);
    const uint32_t p2_grid = npages / slice_pages + nb2 + 1u;
    const int lo_bits = tp.l1.lo_bits ? tp.l1.lo_bits : SC_LO_BITS_TWO_LEVEL, hi_shift = lo_bits + d1 + 9;
    if (binb == 16)
        hipLaunchKernelGGL(page_hist_kernel<true>, dim3(p2_grid), dim3(P2_THREADS), 0, stream, (const uint8_t *)tp.d_pages2, (const PageEntry *)tp.d_list2,
                           (const uint32_t *)page_base, (const uint32_t *)slice_base, nb2, d_table, lo_bits, hi_shift, table_is_zero, d_ctr);
    else
        hipLaunchKernelGGL(page_hist_kernel<false>, dim3(p2_grid), dim3(P2_THREADS), 0, stream, (const uint8_t *)tp.d_pages2, (const PageEntry *)tp.d_list2,
                           (const uint32_t *)page_base, (const uint32_t *)slice_base, nb2, d_table, lo_bits, hi_shift, table_is_zero, d_ctr);
    prof.end();
    tp.flushes++; tp.flushed_batches += (uint64_t)tp.pending;
    twolevel_paged_drop(tp);
    if (hipGetLastError() != hipSuccess) { partition_error_ref() = "histogram pass over the page arena failed to launch"; return 1; }
    return 0;
}

// returns 0 ok, 1 error (partition_error()), 2 no room for the scratch (nothing of the batch was counted)
inline int twolevel_paged_count(TwoLevelPaged &tp, hipStream_t stream, const uint8_t *d_bases, size_t nbytes, const RecStarts &rs,
                                size_t max_windows /* nbytes - records x (k - 1) */,
                                int k, int canonical, int n_expand, unsigned long long *d_table, DevCounters *d_ctr, ProfHook &prof)
{
    int d1, binb;
    paged_bits(k, &d1, &binb);
    const int nb1 = 1 << d1;
    const uint32_t nb2 = 1u << (d1 + 9);
    const bool wide = k == 17;
    const int lo_bits = tp.l1.lo_bits ? tp.l1.lo_bits : SC_LO_BITS_TWO_LEVEL;
    // k <= 15 (option l1_one_round): 128 rings of 256 three-byte elements instead of 256 of 128 -- a ring then takes a whole tile's arrivals (128 +- 34 on top of < 64
    // left over) and the tile is placed in ONE round: two barriers and one flush per tile instead of four and two; level 1 1.82 -> 1.67 ms at k = 15 (the 64 lanes of a
    // request meet in 128 rings more often than in 256: not the 30 % the instruction count promised).  k = 16 has 256 digits: 256 rings of 256 elements do not fit.
    const bool one_round1 = tp.l1_one_round != 0 && tp.l1_wide != 0 && !wide && (1 << d1) <= 128;
    const int rings1 = wide ? L1W_RINGS : (one_round1 ? 128 : L1_RINGS);
    // level-1 elements: 24-bit remainders in three bytes (k <= 16), 25-bit ones in four (k = 17)
    const uint32_t l1_page_elems = wide ? 256u : 512u;
    const size_t l1_page_bytes = wide ? (size_t)ElemFmt<uint32_t>::PAGE_BYTES : (size_t)ElemFmt<u24>::PAGE_BYTES;
    int sub_log2 = 0;
    while ((nb1 << sub_log2) < rings1) sub_log2++;
    // level 1 in 128-byte pieces (option l1_wide_lines): one workgroup of 1024 threads per CU, rings of twice the elements, tiles of 1023 chunks
    const bool wide1 = tp.l1_wide != 0;
    const uint32_t tile_stride1 = wide1 ? (uint32_t)SC1_THREADS - 1u : (uint32_t)SC_TILE_STRIDE, tile_pos1 = tile_stride1 * 16u;
    const uint64_t ntiles_all = ((nbytes + 15) / 16 + tile_stride1 - 1) / tile_stride1;
    const uint32_t Gmax = tp.l1.grid > 0 ? (uint32_t)tp.l1.grid : (uint32_t)(wide1 ? SC1_GRID : SC_GRID);
    const uint64_t max_tiles = scatter_max_tiles(Gmax, tile_pos1);
    if (tp.pending && tp.k_pending != k) { if (twolevel_paged_flush(tp, stream, d_table, d_ctr, prof)) return 1; }
    // level-1 scratch for the largest sub-batch; small arrays
    {
        const uint64_t nt = ntiles_all < max_tiles ? ntiles_all : max_tiles;
        const uint32_t G = (uint32_t)(nt < Gmax ? nt : Gmax);
        const uint32_t tpw = (uint32_t)((nt + G - 1) / G);
        const int rc = scatter_reserve(tp.l1, stream, (size_t)G * scatter_wg_pages(tpw, rings1, l1_page_elems, tile_pos1, scatter_extra_elems(tpw, tile_pos1, n_expand)),
                                       (size_t)nb1, l1_page_bytes);
        if (rc == 2) { partition_error_ref() = "scratch allocation failed"; return 2; }
        if (rc) { partition_error_ref() = "stream error"; return 1; }
    }
    if (!tp.d_wg_range && hipMalloc((void **)&tp.d_wg_range, (SC_GRID + 1) * sizeof(uint32_t)) != hipSuccess) { (void)hipGetLastError(); partition_error_ref() = "scratch allocation failed"; return 2; }
    if (!tp.d_cursor) {
        if (hipMalloc((void **)&tp.d_cursor, sizeof(uint32_t)) != hipSuccess ||
            hipHostMalloc((void **)&tp.h_probe, TwoLevelPaged::PROBES * sizeof(uint32_t), hipHostMallocDefault) != hipSuccess) { (void)hipGetLastError(); partition_error_ref() = "scratch allocation failed"; return 2; }
        for (int i = 0; i < TwoLevelPaged::PROBES; i++)
            if (hipEventCreateWithFlags(&tp.ev_probe[i], hipEventDisableTiming) != hipSuccess) { partition_error_ref() = "event creation failed"; return 1; }
    }
    if (tp.nb2_cap < nb2) {
        if (tp.d_bkt2) { if (hipStreamSynchronize(stream) != hipSuccess) return 1; (void)hipFree(tp.d_bkt2); tp.d_bkt2 = nullptr; tp.nb2_cap = 0; }
        if (hipMalloc((void **)&tp.d_bkt2, (4 * (size_t)nb2 + 2) * sizeof(uint32_t)) != hipSuccess) { (void)hipGetLastError(); partition_error_ref() = "scratch allocation failed"; return 2; }
        tp.nb2_cap = nb2;
    }
    uint32_t *const bkt_pages1 = tp.l1.d_bkt, *const bkt_elems1 = tp.l1.d_bkt + nb1, *const page_base1 = tp.l1.d_bkt + 2 * nb1, *const slice_base1 = tp.l1.d_bkt + 3 * nb1 + 1;
    for (uint64_t t0 = 0; t0 < ntiles_all; t0 += max_tiles) {
        const uint32_t nt = (uint32_t)((ntiles_all - t0) < max_tiles ? (ntiles_all - t0) : max_tiles);
        const uint32_t G = nt < Gmax ? nt : Gmax;
        ScOut out1;
        out1.pages = tp.l1.d_pages; out1.tag = tp.l1.d_tag; out1.wg_range = nullptr; out1.contig = (uint32_t)tp.l1.contig_pages; out1.wg_base = 0; out1.grid = 0;
        out1.extra_elems = scatter_extra_elems((nt + G - 1) / G, tile_pos1, n_expand);
        out1.wg_pages = scatter_wg_pages((nt + G - 1) / G, rings1, l1_page_elems, tile_pos1, out1.extra_elems);
        const uint32_t npages1 = G * out1.wg_pages;
        // what level 2 can need at most (l2_plan_kernel hands out exactly what it does need, within this)
        const bool wide2 = tp.l2_wide != 0;                               // level 2 in 128-byte pieces: one workgroup of 1024 threads per CU
        const uint32_t G2 = wide2 ? (uint32_t)SC1_GRID : (uint32_t)SC_GRID;
        // (a page per 512 of the elements level 1 can emit -- at most one per position, N expansions go straight to the vector --
        //  plus level 1's partial pages rounded up, plus a partial page per ring and digit span of every level-2 workgroup)
        //  A batch of records that are all at least k long has nbytes - records x (k - 1) windows; one that is not fails at the
        //  sync, and until then a scatter kernel that runs out of its page sequence stops writing (internal_err), never out of bounds.
        // (N-expansion mode: plus the fills of N-windows that level 1 may send through its rings)
        const size_t pos = (size_t)nt * tile_pos1, elems = (pos < max_windows ? pos : max_windows) + (size_t)G * out1.extra_elems;
        const size_t need2 = (elems + 511) / 512 + (size_t)G * (size_t)rings1 + 2 * (size_t)G2 + 512 * ((size_t)nb1 + G2) + 16;
        // room in the arena (acquired before any kernel of the sub-batch runs: "no room" must leave nothing counted)
        if (tp.budget_bytes == 0) {
            size_t free_b = 0, total_b = 0;
            (void)hipMemGetInfo(&free_b, &total_b);
            // 85 % of what is free now (the vector and level 1's scratch are allocated already): every batch more in the arena
            // makes the sweep of the 4^k vector cheaper per batch (k = 17: 61 ms per flush, 24 batches at 70 % -> 2.5 ms each)
            tp.free_at_sizing = free_b;
            tp.budget_bytes = free_b / 100 * 85;
            // (a job that ends in a reduce: RCCL allocates its channel and peer-to-peer buffers at the first collective of each kind, and the
            //  sharded reduce shapes want a chunk of scratch -- the arena, which sizes itself on what is free, must not have taken that room)
            if (tp.reserve_bytes && tp.budget_bytes + tp.reserve_bytes > free_b) tp.budget_bytes = free_b > tp.reserve_bytes ? free_b - tp.reserve_bytes : 0;
            if (tp.budget_bytes > (192ull << 30)) tp.budget_bytes = 192ull << 30;
            if (tp.budget_bytes < (1ull << 30)) tp.budget_bytes = 1ull << 30;
        }
        size_t budget_pages = tp.defer ? tp.budget_bytes / SC_PAGE_BYTES : 0;
        if (tp.grow_failed && budget_pages > tp.cap2) budget_pages = tp.cap2;       // (a larger arena could not be had: what there is, is the budget)
        if (budget_pages < need2) budget_pages = need2;
        // worth enlarging: the arena filled up, a quarter more (at least) is within the budget -- and the larger arena pays.
        // Fresh device memory costs ~46 ms per GiB on this runtime (hipMalloc of 128 GiB: 5.9 s, tools/malloc_time.py); a flush
        // forced by a full arena costs one sweep of the vector (16 B per counter at ~5.5 TB/s: 50 ms at k = 17, 3 ms at k = 15),
        // and twice the arena saves every second one.  The arena doubles once the sweeps it would have saved so far add up to
        // the price of the allocation (so a job never spends more than about twice what the best fixed size would have cost
        // it): k = 17 after ~90 forced flushes at 8 batches per flush, k <= 15 in effect never.
        bool may_grow = tp.filled_up && tp.cap2 + tp.cap2 / 4 <= budget_pages && tp.grow != 0;
        if (may_grow && tp.grow == 1) {
            const size_t next_cap = 2 * tp.cap2 < budget_pages ? 2 * tp.cap2 : budget_pages;
            const double alloc_ms = (double)next_cap * (double)SC_PAGE_BYTES / (double)(1ull << 30) * 46.0;
            const double sweep_ms = (double)(1ull << (2 * k)) * 16.0 / 5.5e9;
            may_grow = (double)tp.full_flushes * sweep_ms * 0.5 >= alloc_ms;
        }
        twolevel_paged_poll(tp);
        if (tp.used2 - tp.slack + need2 > tp.cap2 || tp.cap2 == 0 || may_grow) {
            if (tp.pending) { if (twolevel_paged_flush(tp, stream, d_table, d_ctr, prof)) return 1; }
            tp.filled_up = false;
            // The arena grows with the job: room for eight batches like this one at first, twice as much every time it has
            // filled up, until the budget is reached -- a small job (or several processes on one device) never holds tens
            // of GiB it does not use, a long one amortises the sweep of the vector over as many batches as fit.
            size_t want_cap = tp.cap2 == 0 ? (size_t)tp.first_batches * need2 : 2 * tp.cap2;
            if (want_cap > budget_pages) want_cap = budget_pages;
            if (want_cap < need2) want_cap = need2;
            if (tp.cap2 < need2 || (may_grow && tp.cap2 < want_cap) || tp.cap2 == 0) {
                if (hipStreamSynchronize(stream) != hipSuccess) return 1;
                const size_t old_cap = tp.cap2;
                if (tp.d_pages2) { (void)hipFree(tp.d_pages2); (void)hipFree(tp.d_tag2); (void)hipFree(tp.d_list2); tp.d_pages2 = nullptr; tp.d_tag2 = nullptr; tp.d_list2 = nullptr; tp.cap2 = 0; }
                // what is wanted; failing that what there was (and no further attempts to grow); failing that just this batch
                const size_t tries[3] = {want_cap, old_cap >= need2 ? old_cap : need2, need2};
                for (int attempt = 0; attempt < 3 && !tp.d_pages2; attempt++) {
                    const size_t cap = tries[attempt];
                    if (attempt && cap == tries[attempt - 1]) continue;
                    if (hipMalloc((void **)&tp.d_pages2, cap * (size_t)SC_PAGE_BYTES) == hipSuccess &&
                        hipMalloc((void **)&tp.d_tag2, cap * sizeof(uint32_t)) == hipSuccess &&
                        hipMalloc((void **)&tp.d_list2, cap * sizeof(PageEntry)) == hipSuccess) { tp.cap2 = cap; tp.tags_dirty = cap; if (attempt) tp.grow_failed = true; break; }
                    (void)hipGetLastError();
                    if (tp.d_pages2) (void)hipFree(tp.d_pages2);
                    if (tp.d_tag2) (void)hipFree(tp.d_tag2);
                    if (tp.d_list2) (void)hipFree(tp.d_list2);
                    tp.d_pages2 = nullptr; tp.d_tag2 = nullptr; tp.d_list2 = nullptr;
                }
                tp.reallocs++;
                tp.full_flushes = 0;
                if (!tp.d_pages2) { partition_error_ref() = "scratch allocation failed"; return t0 == 0 ? 2 : 1; }
            }
        }
        if (tp.pending == 0) {
            // a new cycle: the cursor goes back to the start of the arena, and the tags the last cycle (or a fresh allocation) left behind are cleared
            if (tp.tags_dirty > tp.cap2) tp.tags_dirty = tp.cap2;
            if (hipMemsetAsync(tp.d_cursor, 0, sizeof(uint32_t), stream) != hipSuccess ||
                (tp.tags_dirty && hipMemsetAsync(tp.d_tag2, 0xFF, tp.tags_dirty * sizeof(uint32_t), stream) != hipSuccess)) { partition_error_ref() = "memset failed"; return 1; }
            tp.tags_dirty = 0;
        }
        if (hipMemsetAsync(tp.l1.d_tag, 0xFF, (size_t)npages1 * sizeof(uint32_t), stream) != hipSuccess ||
            hipMemsetAsync(tp.l1.d_bkt, 0, 2 * (size_t)nb1 * sizeof(uint32_t), stream) != hipSuccess) { partition_error_ref() = "memset failed"; return 1; }
        // ---- level 1
        prof.begin(KDB_KERNEL_SCATTER);
#define KDB_LAUNCH_L1K(ID, EL, RG, CC, RD, TH, E, CN, RAG, KK)                                                                                 \
    hipLaunchKernelGGL((scatter_bases_kernel<ID, EL, RG, CC, RD, E, CN, KK, TH, RAG>), dim3(G), dim3(TH), 0, stream, d_bases,                 \
                       (uint64_t)nbytes, (uint32_t)t0, nt, k, lo_bits + 9, d1, sub_log2, out1, d_table, d_ctr, rs)
#define KDB_LAUNCH_L1R(ID, EL, RG, CC, RD, TH, E, CN, RAG) KDB_LAUNCH_L1K(ID, EL, RG, CC, RD, TH, E, CN, RAG, 0)
#define KDB_LAUNCH_L1(ID, EL, RG, CC, RD, TH, E, CN) do { KDB_LAUNCH_L1R(ID, EL, RG, CC, RD, TH, E, CN, false); KDB_LAUNCH_L1R(ID, EL, RG, CC, RD, TH, E, CN, true); } while (0)
#define KDB_LAUNCH_L1_MODES(ID, EL, RG, CC, RD, TH)                                                                                            \
    do {                                                                                                                                       \
        if (n_expand) { if (canonical) KDB_LAUNCH_L1(ID, EL, RG, CC, RD, TH, true, true); else KDB_LAUNCH_L1(ID, EL, RG, CC, RD, TH, true, false); }     \
        else          { if (canonical) KDB_LAUNCH_L1(ID, EL, RG, CC, RD, TH, false, true); else KDB_LAUNCH_L1(ID, EL, RG, CC, RD, TH, false, false); }   \
    } while (0)
        const bool compiled15 = !wide && k == 15 && !n_expand && canonical && lo_bits == SC_LO_BITS_TWO_LEVEL && tp.l1k;
        const bool compiled17 = wide && wide1 && !n_expand && canonical && lo_bits == SC_LO_BITS_TWO_LEVEL && tp.l1k;      // BASELINE config 4's level 1, likewise
        // (k = 14 and k = 16 likewise, canonical drop mode: the kernels of the default path)
        const bool compiled14 = !wide && k == 14 && !n_expand && canonical && lo_bits == SC_LO_BITS_TWO_LEVEL && tp.l1k;
        const bool compiled16 = !wide && k == 16 && !n_expand && canonical && lo_bits == SC_LO_BITS_TWO_LEVEL && tp.l1k;
        if (one_round1) {
            if (compiled15) {
                KDB_LAUNCH_L1K(uint32_t, u24w, 128, 256, 16, SC1_THREADS, false, true, false, 15);
                KDB_LAUNCH_L1K(uint32_t, u24w, 128, 256, 16, SC1_THREADS, false, true, true, 15);
            } else if (compiled14) {
                KDB_LAUNCH_L1K(uint32_t, u24w, 128, 256, 16, SC1_THREADS, false, true, false, 14);
                KDB_LAUNCH_L1K(uint32_t, u24w, 128, 256, 16, SC1_THREADS, false, true, true, 14);
            } else KDB_LAUNCH_L1_MODES(uint32_t, u24w, 128, 256, 16, SC1_THREADS);
        } else if (wide1 && compiled16) {
            KDB_LAUNCH_L1K(uint32_t, u24w, L1_RINGS, 2 * L1_C, L1_ROUND, SC1_THREADS, false, true, false, 16);
            KDB_LAUNCH_L1K(uint32_t, u24w, L1_RINGS, 2 * L1_C, L1_ROUND, SC1_THREADS, false, true, true, 16);
        } else if (wide1) {
            if (compiled15) {
                KDB_LAUNCH_L1K(uint32_t, u24w, L1_RINGS, 2 * L1_C, L1_ROUND, SC1_THREADS, false, true, false, 15);
                KDB_LAUNCH_L1K(uint32_t, u24w, L1_RINGS, 2 * L1_C, L1_ROUND, SC1_THREADS, false, true, true, 15);
            } else if (!wide) KDB_LAUNCH_L1_MODES(uint32_t, u24w, L1_RINGS, 2 * L1_C, L1_ROUND, SC1_THREADS);
            else if (compiled17) {
                KDB_LAUNCH_L1K(uint64_t, u32w, L1W_RINGS, 2 * L1W_C, L1W_ROUND, SC1_THREADS, false, true, false, 17);
                KDB_LAUNCH_L1K(uint64_t, u32w, L1W_RINGS, 2 * L1W_C, L1W_ROUND, SC1_THREADS, false, true, true, 17);
            }
            else KDB_LAUNCH_L1_MODES(uint64_t, u32w, L1W_RINGS, 2 * L1W_C, L1W_ROUND, SC1_THREADS);
        } else if (compiled15) {
            // BASELINE config 3's kernel with its shifts and masks compiled in (as the k = 12 headline's)
            KDB_LAUNCH_L1K(uint32_t, u24, L1_RINGS, L1_C, L1_ROUND, SC_THREADS, false, true, false, 15);
            KDB_LAUNCH_L1K(uint32_t, u24, L1_RINGS, L1_C, L1_ROUND, SC_THREADS, false, true, true, 15);
        } else if (!wide) KDB_LAUNCH_L1_MODES(uint32_t, u24, L1_RINGS, L1_C, L1_ROUND, SC_THREADS);
        else KDB_LAUNCH_L1_MODES(uint64_t, uint32_t, L1W_RINGS, L1W_C, L1W_ROUND, SC_THREADS);
#undef KDB_LAUNCH_L1_MODES
#undef KDB_LAUNCH_L1
#undef KDB_LAUNCH_L1R
#undef KDB_LAUNCH_L1K
        prof.end();
        prof.begin(KDB_KERNEL_PAGE_SORT);
        const uint32_t pgrid = (npages1 + 4095u) / 4096u < 256u ? (npages1 + 4095u) / 4096u : 256u;
        hipLaunchKernelGGL(pages_count_kernel, dim3(pgrid), dim3(PAGES_THREADS), 0, stream, (const uint32_t *)tp.l1.d_tag, npages1, (uint32_t)nb1, bkt_pages1, bkt_elems1,
                           &d_ctr->pages_bases, wide ? 16u : 32u);
        hipLaunchKernelGGL(pages_scan_kernel, dim3(1), dim3(1024), 0, stream, (const uint32_t *)bkt_pages1, (const uint32_t *)bkt_elems1, (uint32_t)nb1, page_base1,
                           slice_base1, 1u << 20, d_ctr);
        hipLaunchKernelGGL(pages_place_kernel, dim3(pgrid), dim3(PAGES_THREADS), 0, stream, (const uint32_t *)tp.l1.d_tag, npages1, (uint32_t)nb1, bkt_pages1,
                           (const uint32_t *)page_base1, tp.l1.d_list);
        hipLaunchKernelGGL(l2_plan_kernel, dim3(1), dim3(1024), 0, stream, (const uint32_t *)page_base1, (uint32_t)nb1, G2, 512u, l1_page_elems, tp.d_wg_range, tp.d_cursor,
                           (uint32_t)tp.cap2, d_ctr);
        prof.end();
        tp.used2 += need2;
        {
            // tell the host where the cursor stands now (a probe slot that is still in flight is left alone: the bound stays valid without it)
            const int pi = tp.probe_next;
            if (!tp.probe_live[pi] && hipMemcpyAsync(&tp.h_probe[pi], tp.d_cursor, sizeof(uint32_t), hipMemcpyDeviceToHost, stream) == hipSuccess &&
                hipEventRecord(tp.ev_probe[pi], stream) == hipSuccess) {
                tp.probe_live[pi] = true; tp.probe_bound[pi] = tp.used2;
                tp.probe_next = (pi + 1) % TwoLevelPaged::PROBES;
            }
        }
        // ---- level 2
        ScOut out2;
        out2.pages = tp.d_pages2; out2.tag = tp.d_tag2; out2.wg_pages = 0; out2.wg_range = tp.d_wg_range; out2.contig = 0; out2.wg_base = 0; out2.grid = 0; out2.extra_elems = 0;
        prof.begin(KDB_KERNEL_SCATTER_L2);
#define KDB_LAUNCH_L2(IN, EL, CC, FX, TH)                                                                                                       \
    hipLaunchKernelGGL((scatter_ids_kernel<IN, EL, 512, CC, FX, TH>), dim3(G2), dim3(TH), 0, stream, (const uint8_t *)tp.l1.d_pages,            \
                       (const PageEntry *)tp.l1.d_list, (const uint32_t *)page_base1, (uint32_t)nb1, lo_bits, 9, out2, d_ctr)
        const bool fixed2 = lo_bits == SC_LO_BITS_TWO_LEVEL && tp.l1k;   // (shifts and masks compiled in)
        if (wide2) {
            if (wide && fixed2) KDB_LAUNCH_L2(uint32_t, u16w, 128, true, SC1_THREADS);
            else if (wide) KDB_LAUNCH_L2(uint32_t, u16w, 128, false, SC1_THREADS);
            else if (fixed2) KDB_LAUNCH_L2(u24, u16w, 128, true, SC1_THREADS);
            else KDB_LAUNCH_L2(u24, u16w, 128, false, SC1_THREADS);
        } else {
            if (wide) KDB_LAUNCH_L2(uint32_t, uint16_t, 64, false, SC_THREADS);
            else if (fixed2) KDB_LAUNCH_L2(u24, uint16_t, 64, true, SC_THREADS);
            else KDB_LAUNCH_L2(u24, uint16_t, 64, false, SC_THREADS);
        }
#undef KDB_LAUNCH_L2
        prof.end();
        tp.pending++;
        tp.k_pending = k;
        if (hipGetLastError() != hipSuccess) { partition_error_ref() = "two-level paged scatter failed to launch"; return 1; }
        // flush now if told not to defer, after PAGED_PENDING_MAX batches, or when another batch like this one would not fit the arena
        if (tp.defer && tp.used2 - tp.slack + need2 > tp.cap2) { tp.filled_up = true; tp.full_flushes++; }      // (the next batch finds the arena empty and may enlarge it)
        if (!tp.defer || tp.pending >= PAGED_PENDING_MAX || tp.used2 - tp.slack + need2 > tp.cap2) { if (twolevel_paged_flush(tp, stream, d_table, d_ctr, prof)) return 1; }
    }
    return 0;
}

}  // namespace kdb
