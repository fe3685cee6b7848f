// kdb_kernels.hip.h -- gfx950 (MI355X, CDNA4) device code of the k-mer counting engine.
//
// What the reference does per window (kmerdb/kmer.py:234-317 kmer_to_id, called
// from the window loop kmer.py:526-565 of shred, accumulated by
// kmerdb/parse.py:133-136) is restated here position-parallel:
//
//   * a workgroup owns a 16 KiB tile of the residue buffer (+ one 16-byte halo
//     chunk); every lane loads 16 B per instruction, lanes consecutive
//     (coalesced 1 KiB per wave-instruction);
//   * each 16-byte chunk is turned, once, into a 32-bit big-endian 2-bit word
//     (forward strand), its 2-bit-group-reversed complement (reverse strand,
//     little-endian), and 16-bit masks (not-ACGT, record start, is-N) in LDS;
//   * a k-mer starting at base i of chunk c is then two 64-bit shifts:
//         fwd = (F >> (64-2k-2i)) & (4^k-1)      F = fwd[c]:fwd[c+1]
//         rc  = (R >> 2i)         & (4^k-1)      R = rc[c+1]:rc[c]
//     and is valid iff no not-ACGT bit in [i,i+k) and no record start in (i,i+k);
//   * id = min(fwd, rc) (canonical, kmer.py:314-315) or fwd (kmer.py:317).
//
// No MFMA anywhere: this is integer scan + scatter-increment, bound by HBM
// streaming and by atomic / LDS-atomic rate, not by a contraction.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace kdb {

constexpr int TPB = 256;                          // 4 waves of 64
constexpr int CHUNKS_PER_THREAD = 4;
constexpr int TILE_CHUNKS = TPB * CHUNKS_PER_THREAD;   // 1024 chunks of 16 bases
constexpr int TILE_BYTES = TILE_CHUNKS * 16;           // 16 KiB of residues per workgroup

struct DevCounters {
    unsigned long long total_kmers;     // parse.py:136
    unsigned long long n_short;         // records shorter than k (kmer.py:461-463)
    unsigned long long n_bad;           // residues outside ACGTN
    unsigned long long unique;          // count_nonzero(counts)   parse.py:141
    unsigned long long sum;             // Sum(counts)
    unsigned long long bad_layout;      // offsets do not tile [0, nbytes) exactly (sticky until kdb_reset, like n_short / n_bad)
    unsigned long long not_uniform;     // kdb_submit_device_const batches whose records do not all have one length (sticky)
    unsigned long long internal_err;    // a kernel refused to write out of bounds (a sizing bug: the engine reports KDB_ERR_STATE)
    unsigned long long table_dirty;     // something was added to the vector directly since kdb_reset (degenerate ids): a deferred histogram pass must add, not store
    // ragged batches carry their record starts as bit 7 of a record's first byte: marks placed by mark_reads_kernel vs marks
    // the counting kernels met.  They differ iff the buffer held bytes with bit 7 set that are not this batch's starts
    // (stale marks of an aborted job, or bytes that are no residues): an error at the sync, never a silently dropped window
    unsigned long long marks_set, marks_seen;
    // what the LDS-histogram paths moved through HBM by their own account (cumulative; bench.py takes differences): pages that
    // hold elements and the 64-byte lines written into them, per scatter kernel; bytes of the vector read + written by the
    // histogram pass.  Counted where the work is done (page sort, histogram flush), not estimated.
    unsigned long long pages_bases, lines_bases;     // written by scatter_bases_kernel (read by scatter_ids_kernel or page_hist_kernel)
    unsigned long long pages_ids, lines_ids;         // written by scatter_ids_kernel (read by page_hist_kernel)
    unsigned long long table_bytes;                  // page_hist_kernel: bytes of the count vector read + written
    // per-batch record geometry (PER_BATCH_WORDS words zeroed before every batch, filled by lens_kernel)
    unsigned long long neg_min_len;     // max over records of ~len  (== ~min len)
    unsigned long long max_len;         // max record length; ~0 if the batch must use start marks
    unsigned long long wl_count;        // N-windows queued for expand_worklist_kernel in this batch (may exceed wl_cap)
    unsigned long long sus_count;       // residues that are neither ACGT nor N met in this batch (DROP mode; may exceed sus_cap)
    // set once per engine (EXPAND mode): work list of windows with more than two N's
    unsigned long long *wl;
    unsigned long long wl_cap;
    // set once per engine: byte positions of this batch's residues that are neither ACGT nor N, judged by resolve_suspects_kernel
    unsigned long long *sus;
    unsigned long long sus_cap;
    // set per batch by the overlapped one-level path (null otherwise): where a scatter kernel leaves the (id, count) pairs of degenerate
    // ids instead of adding them to the vector itself -- the histogram pass of the batch BEFORE runs beside it and updates its bins
    // with plain read-modify-writes.  [0] = entries appended, [1] = capacity, then the pairs; apply_hot_kernel adds them, in hist order.
    unsigned long long *hot_side;
};
constexpr int PER_BATCH_WORDS = 4;      // neg_min_len, max_len, wl_count, sus_count

// all records of the batch have the same length L  ->  record starts are the multiples of L and no
// start marks are needed (the usual shape of Illumina FASTQ); 0 otherwise
__device__ __forceinline__ uint32_t batch_uniform_len(const DevCounters *c)
{
    const unsigned long long mx = c->max_len, mn = ~c->neg_min_len;
    return (mx == mn && mx != 0 && mx < (1ull << 30)) ? (uint32_t)mx : 0u;
}

// ---------------------------------------------------------------------------------
// record geometry of a batch: min / max length, short-read check, layout check
// ---------------------------------------------------------------------------------
// Record starts of a ragged batch come straight from the offsets (round 3 wrote them into the residues as bit-7 marks and took
// them off again: two kernels of scattered byte read-modify-writes, 0.78 ms of a 2.05 ms step over 10 M reads of 35..150 bp).
// first_rec[B] = the record that holds byte B << FIRST_REC_SHIFT of the batch: a tile of a counting kernel starts its walk
// through the offsets there.  Filled by lens_kernel from the offsets it reads anyway.
constexpr int FIRST_REC_SHIFT = 12;

__global__ void __launch_bounds__(256)
lens_kernel(const uint64_t *__restrict__ offs, uint64_t nreads, uint64_t nbytes, int k, int first_is_continuation, DevCounters *ctr,
            uint32_t *__restrict__ first_rec /* [(nbytes >> FIRST_REC_SHIFT) + 1], or null */)
{
    // grid-stride over the records; one set of global atomics per workgroup (same-address atomics serialise
    // at the memory side: one per wave would cost milliseconds on a 10 M-read batch)
    __shared__ unsigned long long s_len[4], s_neg[4], s_short[4], s_bad[4];
    unsigned long long len = 0, neg = 0, nshort = 0, nbadl = 0;
    const uint64_t stride = (uint64_t)gridDim.x * blockDim.x;
    for (uint64_t rb = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; rb < nreads; rb += 4 * stride) {
        uint64_t sv[4], ev[4];
#pragma unroll
        for (int u = 0; u < 4; u++) {                            // four records' offsets in flight per lane
            const uint64_t r = rb + (uint64_t)u * stride;
            sv[u] = r < nreads ? offs[r] : 0ull;
            ev[u] = r < nreads ? offs[r + 1] : 0ull;
        }
#pragma unroll
        for (int u = 0; u < 4; u++) {
            const uint64_t r = rb + (uint64_t)u * stride, s = sv[u], e = ev[u];
            if (r >= nreads) break;
            unsigned long long l = e - s;
            neg = neg > ~l ? neg : ~l;
            if (!(r == 0 && first_is_continuation)) nshort += (l < (uint64_t)k) ? 1 : 0;      // (a continuation piece is not a record)
            // the offsets must rise from 0 to nbytes (device-resident offsets are the caller's: nothing was checked on the host)
            nbadl += (r == 0 && s != 0) + (r == nreads - 1 && e != nbytes) + (e < s || e > nbytes);
            if (r == 0 && first_is_continuation) l = ~0ull;          // a tiled long record: this batch needs marks
            len = len > l ? len : l;
            if (first_rec && e > s) {                                // the blocks whose first byte this record holds (offsets that do not tile the buffer fail the job anyway)
                const uint64_t e2 = e < nbytes ? e : nbytes;
                for (uint64_t b = (s + ((1ull << FIRST_REC_SHIFT) - 1ull)) >> FIRST_REC_SHIFT; (b << FIRST_REC_SHIFT) < e2; b++) first_rec[b] = (uint32_t)r;
            }
        }
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
        const unsigned long long l2 = __shfl_down(len, o, 64), n2 = __shfl_down(neg, o, 64);
        len = len > l2 ? len : l2;
        neg = neg > n2 ? neg : n2;
        nshort += __shfl_down(nshort, o, 64);
        nbadl += __shfl_down(nbadl, o, 64);
    }
    const int wave = threadIdx.x >> 6;
    if ((threadIdx.x & 63) == 0) { s_len[wave] = len; s_neg[wave] = neg; s_short[wave] = nshort; s_bad[wave] = nbadl; }
    __syncthreads();
    if (threadIdx.x == 0) {
        for (int w = 1; w < 4; w++) {
            len = len > s_len[w] ? len : s_len[w];
            neg = neg > s_neg[w] ? neg : s_neg[w];
            nshort += s_short[w];
            nbadl += s_bad[w];
        }
        // (both words only grow: a workgroup whose value is already there has nothing to add -- in a batch of equal-length reads that is
        //  all but the first few of the 1024, and a thousand atomics on one address take ~11 ns each at the memory side, one after the other)
        if (len > __hip_atomic_load(&ctr->max_len, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT))
            __hip_atomic_fetch_max(&ctr->max_len, len, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if (neg > __hip_atomic_load(&ctr->neg_min_len, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT))
            __hip_atomic_fetch_max(&ctr->neg_min_len, neg, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if (nshort) __hip_atomic_fetch_add(&ctr->n_short, nshort, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if (nbadl) __hip_atomic_fetch_add(&ctr->bad_layout, nbadl, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
}

// ---------------------------------------------------------------------------------
// record-boundary marks: bit 7 of the first residue of every record (skipped when lengths are uniform)
// ---------------------------------------------------------------------------------
__global__ void __launch_bounds__(256)
mark_reads_kernel(uint8_t *__restrict__ bases, const uint64_t *__restrict__ offs, uint64_t nreads,
                  int first_is_continuation, DevCounters *ctr)
{
    if (batch_uniform_len(ctr) || ctr->bad_layout) return;     // (offsets that do not tile the buffer: nothing is written, the job fails at the sync)
    const uint64_t stride = (uint64_t)gridDim.x * blockDim.x;
    for (uint64_t r = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; r < nreads; r += stride) {
        const uint64_t s = offs[r], e = offs[r + 1];
        if (e > s && !(r == 0 && first_is_continuation)) {
            const uint8_t b = bases[s];
            if (b & 0x80u) atomicAdd(&ctr->n_bad, 1ull);         // (not a residue: every batch takes its own marks off again)
            bases[s] = b | 0x80u;
        }
    }
    // (an empty record gets no mark -- and is a short read: that error comes first)
    if (blockIdx.x == 0 && threadIdx.x == 0) ctr->marks_set += nreads - (first_is_continuation ? 1ull : 0ull);
}

// after the batch's kernels have read the residues: the marks come off again, so the caller's buffer (kdb_submit_device)
// is what it was and can be submitted again with other offsets
__global__ void __launch_bounds__(256)
unmark_reads_kernel(uint8_t *__restrict__ bases, const uint64_t *__restrict__ offs, uint64_t nreads,
                    int first_is_continuation, const DevCounters *ctr)
{
    if (batch_uniform_len(ctr) || ctr->bad_layout) return;
    const uint64_t stride = (uint64_t)gridDim.x * blockDim.x;
    for (uint64_t r = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; r < nreads; r += stride) {
        const uint64_t s = offs[r], e = offs[r + 1];
        if (e > s && !(r == 0 && first_is_continuation))
            bases[s] = bases[s] & 0x7Fu;
    }
}

// host-fed batches: a byte with bit 7 set is not a residue (the reference raises on it: kmer.py:170); bit 7 is the
// engine's own record-start mark, so such bytes are counted as bad BEFORE mark_reads_kernel runs
__global__ void __launch_bounds__(256)
hibit_check_kernel(const uint8_t *__restrict__ bases, uint64_t nbytes, DevCounters *ctr)
{
    const uint4 *v4 = reinterpret_cast<const uint4 *>(bases);
    const uint64_t n16 = nbytes / 16, stride = (uint64_t)gridDim.x * blockDim.x;
    uint32_t nbad = 0;
    for (uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n16; i += stride) {
        const uint4 x = v4[i];
        const uint32_t m = (x.x | x.y | x.z | x.w) & 0x80808080u;
        if (m) nbad += __builtin_popcount(x.x & 0x80808080u) + __builtin_popcount(x.y & 0x80808080u) +
                       __builtin_popcount(x.z & 0x80808080u) + __builtin_popcount(x.w & 0x80808080u);
    }
    if (blockIdx.x == 0 && threadIdx.x < (nbytes & 15u)) nbad += bases[n16 * 16 + threadIdx.x] >> 7;
    if (__ballot(nbad != 0) == 0) return;
    unsigned long long w = 0;
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) nbad += (uint32_t)__shfl_xor((int)nbad, o, 64);
    w = nbad;
    if ((threadIdx.x & 63) == 0 && w) __hip_atomic_fetch_add(&ctr->n_bad, w, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

// kdb_submit_device_const: the caller's buffer is never written, so the batch must not need start marks
__global__ void require_uniform_kernel(DevCounters *ctr)
{
    const unsigned long long mx = ctr->max_len, mn = ~ctr->neg_min_len;
    if (!(mx == mn && mx != 0 && mx < (1ull << 30))) ctr->not_uniform += 1;
}

// start mask of a 16-base chunk when every record has length L: bit b set iff (chunk_start + b) % L == 0
__device__ __forceinline__ uint32_t uniform_starts(uint32_t x /* chunk_start % L */, uint32_t L)
{
    uint32_t d = x ? L - x : 0u, m = 0;
    if (L >= 16u) {                        // (kernel-uniform) at most one record starts in a chunk: no loop
        d = d < 16u ? d : 16u;
        return (1u << d) & 0xFFFFu;
    }
    while (d < 16u) { m |= 1u << d; d += L; }
    return m;
}

// ---------------------------------------------------------------------------------
// 16 residues -> packed words
// ---------------------------------------------------------------------------------
struct Enc {
    uint32_t fwd;     // base b at bits 30-2b, code A0 C1 G2 T3 (kmer.py:44-49); garbage where inv
    uint32_t rc;      // base b at bits 2b, complemented code (3-code)
    uint32_t inv;     // low 16: base b is not ACGT (or past the end of the buffer)
    uint32_t st;      // low 16: base b carries the record-start mark
    uint32_t nn;      // low 16: base b is 'N'
    uint32_t bad;     // low 16: base b is neither ACGT nor N (reference raises); in a uniform batch also: bit 7 set
};

__device__ __forceinline__ uint32_t nonzero_bytes(uint32_t z /* every byte < 0x80 */)
{
    return (z + 0x7F7F7F7Fu) & 0x80808080u;              // 0x80 in each non-zero byte
}

__device__ __forceinline__ uint32_t rev2(uint32_t x)     // reverse the order of the sixteen 2-bit groups
{
    uint32_t y = __builtin_bitreverse32(x);
    return ((y >> 1) & 0x55555555u) | ((y & 0x55555555u) << 1);
}

// ---------------------------------------------------------------------------------
// IUPAC codes other than N.  kmer_to_id returns None for a window that holds an 'N' BEFORE it looks at any other letter
// (kmer.py:287-289), so with replace_with_none=True (DROP) a window that holds, say, an R and an N is dropped like any
// N-window (kmer.py:541-544), and only an R in a window WITHOUT N reaches letterToBinaryNA and raises (KeyError, kmer.py:309).
// A record is therefore refused iff one of its windows holds such a code and no N -- iff the N-free stretch around the
// code, inside its record, is at least k long (tests/golden/iupac_next_to_n.json: the reference's own answers).  With
// replace_with_none=False (EXPAND) the reference's substitution code raises for such windows as well (kmer.py:545-555, :612,
// :652 ff.); every such code is an error there, as are letters outside the IUPAC alphabet in either mode (kmer.py:170).
// Rare by construction: this code only runs for a residue that is neither ACGT nor N.
// ---------------------------------------------------------------------------------
__device__ __forceinline__ bool is_iupac10(uint32_t c /* a byte without its bit 7 */)
{
    return (c & 0xE0u) == 0x40u && ((0x02CC2914u >> (c & 31u)) & 1u) != 0u;       // B D H K M R S V W Y
}

// does a window of its record [s, e) hold the residue at byte p and no N?
__device__ __forceinline__ bool residue_has_n_free_window(const uint8_t *__restrict__ bases, uint64_t s, uint64_t e, uint64_t p, int k)
{
    int run = 1;                                       // N-free residues of the record around p, p included; k are enough
#pragma unroll 1
    for (uint64_t q = p; q > s && run < k;) {
        q--;
        if ((bases[q] & 0x7Fu) == 0x4Eu) break;
        run++;
    }
#pragma unroll 1
    for (uint64_t q = p + 1; q < e && run < k; q++) {
        if ((bases[q] & 0x7Fu) == 0x4Eu) break;
        run++;
    }
    return run >= k;
}

// DROP mode: the counting kernels do not judge such a residue where they meet it (their front ends stay as lean as they are):
// its byte position goes to a short list, and resolve_suspects_kernel -- one small launch per batch -- decides (the record a
// position belongs to: binary search in the offsets).  A list that overflows is not an error (a masked assembly holds more than
// 65 536 shielded codes per GiB; round 4 refused such a batch, the reference counts it, kmer.py:287-289): sus_count keeps counting
// beyond sus_cap, which tells resolve_suspects_kernel to ignore the list and judge every residue of the batch itself.
// -> 0 (nothing is an error at this point)
__device__ __forceinline__ uint32_t defer_suspects16(uint32_t errs, uint64_t pos0, DevCounters *ctr)
{
#pragma unroll 1
    for (uint32_t m = errs; m; m &= m - 1u) {
        const unsigned long long slot = __hip_atomic_fetch_add(&ctr->sus_count, 1ull, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if (slot < ctr->sus_cap) ctr->sus[slot] = pos0 + (uint64_t)__builtin_ctz(m);
    }
    return 0u;
}

// a letter outside the IUPAC alphabet (kmer.py:170), or one of its ten codes in a window that no N shields (kmer.py:309)
__device__ __forceinline__ bool suspect_is_bad(const uint8_t *__restrict__ bases, uint64_t nbytes, const uint64_t *__restrict__ offs, uint64_t nreads, uint64_t p, int k)
{
    if (!is_iupac10(bases[p] & 0x7Fu)) return true;
    uint64_t s = 0, e = nbytes;
    if (offs) {                                        // the record that holds p: the last one that starts at or before it
        uint64_t lo = 0, hi = nreads - 1;
        while (lo < hi) { const uint64_t mid = (lo + hi + 1) >> 1; if (offs[mid] <= p) lo = mid; else hi = mid - 1; }
        s = offs[lo]; e = offs[lo + 1] < nbytes ? offs[lo + 1] : nbytes;
    }
    return residue_has_n_free_window(bases, s, e, p, k);
}

__global__ void __launch_bounds__(256)
resolve_suspects_kernel(const uint8_t *__restrict__ bases, uint64_t nbytes, const uint64_t *__restrict__ offs /* null: one record */, uint64_t nreads,
                        int k, DevCounters *ctr)
{
    const unsigned long long listed = ctr->sus_count;
    const unsigned long long tid = (unsigned long long)blockIdx.x * blockDim.x + threadIdx.x, nthreads = (unsigned long long)gridDim.x * blockDim.x;
    unsigned long long bad = 0;
    if (listed <= ctr->sus_cap) {
        for (unsigned long long i = tid; i < listed; i += nthreads)
            if (suspect_is_bad(bases, nbytes, offs, nreads, ctr->sus[i], k)) bad++;
    } else {
        // the list overflowed: every residue of the batch that is neither ACGT nor N (by its low seven bits: bit 7 is a record-start
        // mark in front of the direct-atomics kernel, and an error of its own everywhere else) is judged here.  Rare and not fast:
        // a few thousand lanes walk the whole batch, four 16-byte chunks in flight each.
        const uint64_t nwhole = nbytes / 16;
        for (uint64_t g0 = tid * 4; g0 < nwhole; g0 += nthreads * 4) {
            uint4 v[4];
#pragma unroll
            for (int u = 0; u < 4; u++) v[u] = g0 + u < nwhole ? *reinterpret_cast<const uint4 *>(bases + (g0 + u) * 16ull) : make_uint4(0x41414141u, 0x41414141u, 0x41414141u, 0x41414141u);
#pragma unroll
            for (int u = 0; u < 4; u++) {
                const uint32_t w[4] = {v[u].x, v[u].y, v[u].z, v[u].w};
#pragma unroll
                for (int q = 0; q < 4; q++) {
                    const uint32_t x7 = w[q] & 0x7F7F7F7Fu;
                    const uint32_t t = ((x7 ^ (x7 >> 1)) >> 1) & 0x03030303u;
                    const uint32_t other = nonzero_bytes(x7 ^ __builtin_amdgcn_perm(0u, 0x54474341u /* "ACGT" */, t)) & nonzero_bytes(x7 ^ 0x4E4E4E4Eu);
                    for (uint32_t m = other; m; m &= m - 1u)
                        if (suspect_is_bad(bases, nbytes, offs, nreads, (g0 + u) * 16ull + 4u * (uint32_t)q + ((uint32_t)__builtin_ctz(m) >> 3), k)) bad++;
                }
            }
        }
        if (tid == 0)
            for (uint64_t p = nwhole * 16; p < nbytes; p++) {
                const uint32_t c = bases[p] & 0x7Fu;
                if (c != 0x41u && c != 0x43u && c != 0x47u && c != 0x54u && c != 0x4Eu && suspect_is_bad(bases, nbytes, offs, nreads, p, k)) bad++;
            }
    }
    if (bad) __hip_atomic_fetch_add(&ctr->n_bad, bad, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

// 4 words whose bytes are 0x00 / 0x80 -> 16-bit mask, bit (4q+b) = byte b of word q.
// v_dot4_u32_u8 sums byte*weight: weights 1,2,4,8 (word 0/2) and 16,32,64,128 (word 1/3); the 0x80 scale is shifted out.
__device__ __forceinline__ uint32_t gather16(uint32_t y0, uint32_t y1, uint32_t y2, uint32_t y3)
{
    uint32_t lo = __builtin_amdgcn_udot4(y1, 0x80402010u, __builtin_amdgcn_udot4(y0, 0x08040201u, 0u, false), false);
    uint32_t hi = __builtin_amdgcn_udot4(y3, 0x80402010u, __builtin_amdgcn_udot4(y2, 0x08040201u, 0u, false), false);
    return ((hi << 8) | lo) >> 7;
}

// EXPAND: also produce the is-N mask.  NEED_BAD: also find residues that are neither ACGT nor N (the kernels that
// report them; the scatter kernels run after a counting kernel has already done so and skip that work).
// FULL: all 16 bytes exist (every chunk but the last one or two of a buffer): no existence masking.
template <bool EXPAND, bool NEED_BAD, bool FULL>
__device__ __forceinline__ Enc encode16(const uint32_t w[4], int nvalid /* 0..16 bytes that exist */, bool uniform, uint32_t ustarts)
{
    uint32_t fwd = 0;
    uint32_t notacgt[4], notn[4], start[4];
#pragma unroll
    for (int q = 0; q < 4; q++) {
        const uint32_t x = w[q];
        const uint32_t x7 = x & 0x7F7F7F7Fu;
        const uint32_t t = ((x7 >> 1) ^ (x7 >> 2)) & 0x03030303u;          // A0 C1 G2 T3 for the four letters
        // big-endian pack of the four codes: byte0*64 + byte1*16 + byte2*4 + byte3
        fwd |= __builtin_amdgcn_udot4(t, 0x01041040u, 0u, false) << (24 - 8 * q);
        // the code is only meaningful if the byte IS that letter: look the letter up again and compare
        const uint32_t expect = __builtin_amdgcn_perm(0u, 0x54474341u /* "ACGT" */, t);
        notacgt[q] = nonzero_bytes(x7 ^ expect);
        if (EXPAND || NEED_BAD) notn[q] = nonzero_bytes(x7 ^ 0x4E4E4E4Eu);
        start[q] = x & 0x80808080u;
    }
    const uint32_t exist = (FULL || nvalid >= 16) ? 0xFFFFu : ((1u << nvalid) - 1u);
    Enc e;
    e.fwd = fwd;
    e.rc = ~rev2(fwd);
    const uint32_t inv = gather16(notacgt[0], notacgt[1], notacgt[2], notacgt[3]);
    e.inv = FULL ? inv : ((inv | ~exist) & 0xFFFFu);
    e.st = uniform ? ustarts : gather16(start[0], start[1], start[2], start[3]);
    if (!FULL) e.st &= exist;
    e.bad = 0; e.nn = 0;
    if (EXPAND || NEED_BAD) {
        e.bad = gather16(notacgt[0] & notn[0], notacgt[1] & notn[1], notacgt[2] & notn[2], notacgt[3] & notn[3]) & exist;
        if (EXPAND) e.nn = inv & ~e.bad & exist;          // not ACGT and not bad == N
        // a uniform batch has no marks: a byte with bit 7 set is no residue (0xC1 is not 'A'; kmer.py:170 raises)
        if (NEED_BAD && uniform && ((start[0] | start[1] | start[2] | start[3]) != 0u))
            e.bad |= gather16(start[0], start[1], start[2], start[3]) & exist;
    }
    return e;
}

// load chunk g (16 bytes at 16*g) of a buffer of nbytes; never reads past nbytes
__device__ __forceinline__ int load_chunk(const uint8_t *__restrict__ bases, uint64_t nbytes, uint64_t g, uint32_t w[4])
{
    uint64_t b0 = g * 16ull;
    if (b0 + 16ull <= nbytes) {
        const uint4 v = *reinterpret_cast<const uint4 *>(bases + b0);
        w[0] = v.x; w[1] = v.y; w[2] = v.z; w[3] = v.w;
        return 16;
    }
    w[0] = w[1] = w[2] = w[3] = 0;
    if (b0 >= nbytes) return 0;
    int n = (int)(nbytes - b0);
    for (int i = 0; i < n; i++) w[i >> 2] |= (uint32_t)bases[b0 + i] << (8 * (i & 3));
    return n;
}

// LDS image of one tile: TILE_CHUNKS + 1 (halo) chunks.  EXPAND adds the is-N mask.
template <bool EXPAND>
struct TileLds {
    uint32_t fwd[TILE_CHUNKS + 1];
    uint32_t rc[TILE_CHUNKS + 1];
    uint32_t msk[TILE_CHUNKS + 1];                      // inv | st << 16
    uint32_t nn[EXPAND ? TILE_CHUNKS + 1 : 1];
};

// record starts of a uniform-length batch without a division per chunk: positions advance by a fixed step between
// the chunks of a thread, so the residue class mod L is kept by add + conditional subtract
struct UniformStarts {
    uint32_t L, stepmod, jmod;      // L = 0: the batch uses start marks
    __device__ __forceinline__ UniformStarts(uint32_t ulen, int threads)
        : L(ulen), stepmod(ulen ? (uint32_t)(threads * 16) % ulen : 0u), jmod(ulen ? (16u * threadIdx.x) % ulen : 0u) {}
};

template <bool EXPAND, bool NEED_BAD>
__device__ __forceinline__ uint32_t stage_chunk(TileLds<EXPAND> &L, const uint8_t *__restrict__ bases, uint64_t nbytes, uint64_t g, int c,
                                                bool uniform, uint32_t ustarts, DevCounters *ctr = nullptr)
{
    uint32_t w[4];
    Enc e;
    if ((g + 1) * 16ull <= nbytes) {                 // the hot path: a whole 16-byte chunk
        const uint4 v = *reinterpret_cast<const uint4 *>(bases + g * 16ull);
        w[0] = v.x; w[1] = v.y; w[2] = v.z; w[3] = v.w;
        e = encode16<EXPAND, NEED_BAD, true>(w, 16, uniform, ustarts);
    } else {
        const int nv = load_chunk(bases, nbytes, g, w);
        e = encode16<EXPAND, NEED_BAD, false>(w, nv, uniform, ustarts);
    }
    L.fwd[c] = e.fwd; L.rc[c] = e.rc; L.msk[c] = e.inv | (e.st << 16);
    if (EXPAND) L.nn[c] = e.nn;
    // DROP mode: residues that are neither ACGT nor N are judged by resolve_suspects_kernel (defer_suspects16); without a list every one is an error
    if (NEED_BAD && !EXPAND && e.bad && ctr) {
        uint32_t hib = 0;
        if (uniform) hib = gather16(w[0] & 0x80808080u, w[1] & 0x80808080u, w[2] & 0x80808080u, w[3] & 0x80808080u);      // (no residues: errors whatever stands around them)
        const uint32_t over = defer_suspects16(e.bad & ~hib, g * 16ull, ctr);
        return ((uint32_t)__builtin_popcount(e.bad & hib) + over) | (uniform ? 0u : (uint32_t)__builtin_popcount(e.st) << 16);
    }
    // low half: bad residues; high half: record-start marks met (ragged batches)
    return NEED_BAD ? (uint32_t)__builtin_popcount(e.bad) | (uniform ? 0u : (uint32_t)__builtin_popcount(e.st) << 16) : 0u;
}

// stage tile `tile` into LDS; *bad_count = residues outside ACGTN seen by this thread | start marks met << 16 (NEED_BAD only:
// see count_bad_and_marks).
// us.L != 0: all records have length us.L, record starts are computed instead of read from bit 7.
template <bool EXPAND, int THREADS = TPB, bool NEED_BAD = true>
__device__ __forceinline__ void stage_tile(TileLds<EXPAND> &L, const uint8_t *__restrict__ bases, uint64_t nbytes,
                                           uint64_t tile, uint32_t *bad_count, const UniformStarts &us, DevCounters *ctr = nullptr)
{
    const int j = threadIdx.x;
    uint32_t nbad = 0;
    uint32_t x = 0, tmod = 0;
    const bool uniform = us.L != 0;
    if (uniform) {
        tmod = (uint32_t)((tile * (uint64_t)TILE_BYTES) % us.L);             // wave-uniform
        x = tmod + us.jmod;
        if (x >= us.L) x -= us.L;
    }
#pragma unroll
    for (int q = 0; q < TILE_CHUNKS / THREADS; q++) {
        const int c = j + q * THREADS;
        nbad += stage_chunk<EXPAND, NEED_BAD>(L, bases, nbytes, tile * TILE_CHUNKS + (uint64_t)c, c, uniform,
                                              uniform ? uniform_starts(x, us.L) : 0u, ctr);
        if (uniform) { x += us.stepmod; if (x >= us.L) x -= us.L; }
    }
    if (j == 0) {                       // halo chunk: windows of the last 16 positions reach into it
        uint32_t hx = 0;
        if (uniform) { hx = tmod + (uint32_t)TILE_BYTES % us.L; if (hx >= us.L) hx -= us.L; }
        // its bad residues are counted by the tile that owns it
        (void)stage_chunk<EXPAND, false>(L, bases, nbytes, (tile + 1) * TILE_CHUNKS, TILE_CHUNKS, uniform,
                                         uniform ? uniform_starts(hx, us.L) : 0u);
    }
    *bad_count = nbad;
}

// the 32-position neighbourhood a lane needs for the 16 windows starting in chunk c
struct Hood {
    uint32_t f0, f1;   // forward words of chunk c, c+1 (base b of the pair at bits 62-2b of f0:f1)
    uint32_t r0, r1;   // reverse-strand words (base b at bits 2b of r1:r0, complemented)
    uint32_t V;        // bit b: base b of [chunk c, chunk c+1] is not ACGT
    uint32_t S;        // bit b: base b carries a record start
    __device__ __forceinline__ uint64_t F() const { return ((uint64_t)f0 << 32) | f1; }
    __device__ __forceinline__ uint64_t R() const { return ((uint64_t)r1 << 32) | r0; }
};

template <bool EXPAND>
__device__ __forceinline__ Hood load_hood(const TileLds<EXPAND> &L, int c)
{
    Hood h;
    h.f0 = L.fwd[c]; h.f1 = L.fwd[c + 1]; h.r0 = L.rc[c]; h.r1 = L.rc[c + 1];
    const uint32_t m0 = L.msk[c], m1 = L.msk[c + 1];
    h.V = (m0 & 0xFFFFu) | (m1 << 16);
    h.S = (m0 >> 16) | (m1 & 0xFFFF0000u);
    return h;
}

// window [i, i+k) is counted iff it has no non-ACGT base and no record start strictly inside it
__device__ __forceinline__ bool window_crosses(const Hood &h, int i, uint32_t k1mask) { return (((h.S >> 1) >> i) & k1mask) != 0; }

// bit i (0..15) set iff the window starting at base i of the chunk is NOT counted: a non-ACGT base in [i, i+k) or a
// record start in (i, i+k).  One sliding-window OR of length w = k - 1 per chunk on the 32-bit masks instead of two
// bit-field extracts and a compare per window.  A window of length w is the union of two windows of length m (the
// largest power of two <= w) that lie w - m apart, and the length-m OR comes from doubling steps whose shift amounts are
// 2^b while 2^b < m and 0 after -- kernel-uniform scalars, so the 13 instructions hold no select and no branch.
struct WinOr {
    uint32_t s0, s1, s2, s3, d, w;
    __device__ __forceinline__ explicit WinOr(int k)                  // 2 <= k <= 17
    {
        w = (uint32_t)(k - 1);
        uint32_t m = 1;
        while (2u * m <= w) m *= 2u;
        s0 = 1u < m ? 1u : 0u; s1 = 2u < m ? 2u : 0u; s2 = 4u < m ? 4u : 0u; s3 = 8u < m ? 8u : 0u;
        d = w - m;
    }
};

__device__ __forceinline__ uint32_t windows_bad16(const Hood &h, const WinOr &o)
{
    const uint32_t X = h.V;
    uint32_t p = h.V | (h.S >> 1);                         // either defect, over the k-1 positions i .. i+k-2
    p |= p >> o.s0; p |= p >> o.s1; p |= p >> o.s2; p |= p >> o.s3;      // OR over [i, i+m)
    return (p | (p >> o.d) | (X >> o.w)) & 0xFFFFu;        // OR over [i, i+w), + a non-ACGT base at position i+k-1
}

// (m & a) | (~m & b) in one instruction
__device__ __forceinline__ uint32_t bfi(uint32_t m, uint32_t a, uint32_t b)
{
    uint32_t r;
    asm("v_bfi_b32 %0, %1, %2, %3" : "=v"(r) : "v"(m), "v"(a), "v"(b));
    return r;
}

// k <= 16: ids are 32-bit; v_alignbit_b32 pulls the 16 bases starting at base i out of the word pair
struct IdParams32 { uint32_t fshift /* 32-2k */, mask /* 4^k-1 */; int canonical; };

__device__ __forceinline__ uint32_t window_id32(const Hood &h, int i, const IdParams32 &p)
{
    const uint32_t wf = (i == 0) ? h.f0 : __builtin_amdgcn_alignbit(h.f0, h.f1, 32 - 2 * i);
    uint32_t f = wf >> p.fshift;                                   // kmer.py:307-309
    if (p.canonical) {
        const uint32_t wr = (i == 0) ? h.r0 : __builtin_amdgcn_alignbit(h.r1, h.r0, 2 * i);
        const uint32_t r = wr & p.mask;                            // kmer.py:310-312
        f = f < r ? f : r;                                         // kmer.py:314-315
    }
    return f;
}

__device__ __forceinline__ uint64_t window_id64(const Hood &h, int i, int k, int canonical, uint64_t idmask)
{
    uint64_t f = (h.F() >> (64 - 2 * k - 2 * i)) & idmask;
    if (canonical) {
        uint64_t r = (h.R() >> (2 * i)) & idmask;
        f = f < r ? f : r;
    }
    return f;
}

template <typename ID> struct IdParams;
template <> struct IdParams<uint32_t> {
    IdParams32 p;
    __device__ __forceinline__ IdParams(int k, int canonical) { p.fshift = 32 - 2 * k; p.mask = (k >= 16) ? 0xFFFFFFFFu : ((1u << (2 * k)) - 1u); p.canonical = canonical; }
    __device__ __forceinline__ uint32_t id(const Hood &h, int i) const { return window_id32(h, i, p); }
    // same id with a run-time window index (rolled loops)
    __device__ __forceinline__ uint32_t id_dyn(const Hood &h, int i) const
    {
        uint32_t f = (uint32_t)(h.F() >> (32 - 2 * i)) >> p.fshift;
        if (p.canonical) { const uint32_t r = (uint32_t)(h.R() >> (2 * i)) & p.mask; f = f < r ? f : r; }
        return f;
    }
    __device__ __forceinline__ uint32_t id_any(const Hood &h, int i) const { return id_dyn(h, i); }
};
template <> struct IdParams<uint64_t> {
    int k, canonical; uint64_t idmask;
    __device__ __forceinline__ IdParams(int k_, int canonical_) : k(k_), canonical(canonical_), idmask((1ull << (2 * k_)) - 1ull) {}
    __device__ __forceinline__ uint64_t id(const Hood &h, int i) const { return window_id64(h, i, k, canonical, idmask); }
    __device__ __forceinline__ uint64_t id_any(const Hood &h, int i) const { return id(h, i); }
};

// all 4^m fills of a window with m N's (kmer.py:559-565, 586-621): one increment each.
// `base` = the window's forward id with the N positions zeroed; shift[j] = bit position of the j-th N.
__device__ __forceinline__ uint64_t fill_id(uint64_t base, const int *shift, int m, uint64_t f, int k, int canonical, uint64_t idmask)
{
    uint64_t id = base;
    for (int j = 0; j < m; j++) id |= ((f >> (2 * j)) & 3ull) << shift[j];
    if (canonical) {
        // reverse complement of a k-mer id: reverse the k 2-bit groups, complement
        uint64_t y = __builtin_bitreverse64(id);
        y = ((y >> 1) & 0x5555555555555555ull) | ((y & 0x5555555555555555ull) << 1);
        const uint64_t r = (~y >> (64 - 2 * k)) & idmask;
        id = id < r ? id : r;
    }
    return id;
}

// A window with m <= 2 N's is expanded in place (<= 16 increments).  Windows with more N's (all-N reads are common
// in real FASTQ, and 4^12 fills from one lane would take seconds) are queued; expand_worklist_kernel spreads each
// of them over a whole workgroup after the batch.  If the queue is full the window is expanded in place after all.
__device__ __noinline__ void expand_n_window(unsigned long long *__restrict__ table, uint64_t F, int i, int k,
                                              int canonical, uint64_t idmask, uint32_t nwin /* k bits, bit j = base j of window is N */,
                                              unsigned long long *emitted, DevCounters *ctr = nullptr)
{
    uint64_t base = (F >> (64 - 2 * k - 2 * i)) & idmask;
    const int m = __builtin_popcount(nwin);
    int shift[17];
    int n = 0;
    for (int j = 0; j < k; j++)
        if ((nwin >> j) & 1u) { shift[n++] = 2 * (k - 1 - j); base &= ~(3ull << (2 * (k - 1 - j))); }
    if (m > 2 && ctr && ctr->wl) {
        const unsigned long long slot = __hip_atomic_fetch_add(&ctr->wl_count, 1ull, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if (slot < ctr->wl_cap) { ctr->wl[slot] = base | ((unsigned long long)nwin << 34); return; }
    }
    const uint64_t nfill = 1ull << (2 * m);
    for (uint64_t f = 0; f < nfill; f++)
        __hip_atomic_fetch_add(&table[fill_id(base, shift, m, f, k, canonical, idmask)], 1ull, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    *emitted += nfill;
}

// one workgroup per queued window, the fills spread over its threads
__global__ void __launch_bounds__(256)
expand_worklist_kernel(unsigned long long *__restrict__ table, DevCounters *ctr, int k, int canonical)
{
    const unsigned long long n = ctr->wl_count < ctr->wl_cap ? ctr->wl_count : ctr->wl_cap;
    const uint64_t idmask = (1ull << (2 * k)) - 1ull;
    unsigned long long mine = 0;
    for (unsigned long long e = blockIdx.x; e < n; e += gridDim.x) {
        const unsigned long long ent = ctr->wl[e];
        const uint64_t base = ent & ((1ull << 34) - 1ull);
        const uint32_t nwin = (uint32_t)(ent >> 34);
        int shift[17];
        int m = 0;
        for (int j = 0; j < k; j++) if ((nwin >> j) & 1u) shift[m++] = 2 * (k - 1 - j);
        const uint64_t nfill = 1ull << (2 * m);
        for (uint64_t f = threadIdx.x; f < nfill; f += blockDim.x)
            __hip_atomic_fetch_add(&table[fill_id(base, shift, m, f, k, canonical, idmask)], 1ull, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if (threadIdx.x == 0) mine += nfill;
    }
    if (threadIdx.x == 0 && mine) __hip_atomic_fetch_add(&ctr->total_kmers, mine, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

// Same-address atomics serialise (64-way in LDS, far worse at the memory side), and real reads do contain
// poly-A/poly-G reads and microsatellites.  If the key of the first active lane is shared by >= 16 active lanes,
// those lanes are served by ONE atomic of their total; everything else proceeds as usual.  Wave-uniform result.
__device__ __forceinline__ bool wave_dominant(uint32_t key, uint64_t *same, uint32_t *key0)
{
    *key0 = (uint32_t)__builtin_amdgcn_readfirstlane((int)key);
    *same = __ballot(key == *key0);
    return __popcll(*same) >= 16;
}

__device__ __forceinline__ uint32_t lane_rank_in(uint64_t mask)      // number of set bits below this lane
{
    return __builtin_amdgcn_mbcnt_hi((uint32_t)(mask >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)mask, 0u));
}

// LDS histogram increment with the dominant-key shortcut (out of line: only degenerate stretches come here, and
// the hot loops stay small)
__device__ __noinline__ void lds_hist_add(uint32_t *hist, uint32_t bin)
{
    uint64_t same; uint32_t b0;
    if (wave_dominant(bin, &same, &b0)) {
        if (bin == b0) { if (lane_rank_in(same) == 0) atomicAdd(&hist[b0], (uint32_t)__popcll(same)); }
        else atomicAdd(&hist[bin], 1u);
    } else {
        atomicAdd(&hist[bin], 1u);
    }
}

// global 64-bit add of `cnt` to table[id] with the dominant-key shortcut.  MUST be called by all 64 lanes of the
// wave (convergent code); lanes with cnt == 0 add nothing.  ids are up to 34 bits: both halves are compared.
__device__ __forceinline__ void global_count_add(unsigned long long *__restrict__ table, uint64_t id, uint32_t cnt)
{
    const uint64_t active = __ballot(cnt != 0);
    if (active == 0) return;
    const int first = __ffsll((unsigned long long)active) - 1;
    const uint32_t lo0 = (uint32_t)__builtin_amdgcn_readlane((int)(uint32_t)id, first);
    const uint32_t hi0 = (uint32_t)__builtin_amdgcn_readlane((int)(uint32_t)(id >> 32), first);
    const bool mine = cnt != 0 && ((uint32_t)id == lo0) && ((uint32_t)(id >> 32) == hi0);
    const uint64_t same = __ballot(mine);
    if (__popcll(same) >= 16) {
        uint32_t v = mine ? cnt : 0u;
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) v += (uint32_t)__shfl_xor((int)v, o, 64);
        if (mine) { if (lane_rank_in(same) == 0) __hip_atomic_fetch_add(&table[id], (unsigned long long)v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
        else if (cnt) __hip_atomic_fetch_add(&table[id], (unsigned long long)cnt, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    } else if (cnt) {
        __hip_atomic_fetch_add(&table[id], (unsigned long long)cnt, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
}

__device__ __forceinline__ unsigned long long wave_sum(unsigned long long v)
{
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_down(v, o, 64);
    return v;
}

// a thread's stage_tile result (bad residues | start marks << 16) -> the engine's counters, one atomic per wave and counter
__device__ __forceinline__ void count_bad_and_marks(uint32_t packed, DevCounters *ctr)
{
    const unsigned long long wb = wave_sum((unsigned long long)(packed & 0xFFFFu)), wm = wave_sum((unsigned long long)(packed >> 16));
    if ((threadIdx.x & 63) == 0) {
        if (wb) __hip_atomic_fetch_add(&ctr->n_bad, wb, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if (wm) __hip_atomic_fetch_add(&ctr->marks_seen, wm, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
}

// ---------------------------------------------------------------------------------
// algo 1: encode + direct global atomics (any k <= 17)
// ---------------------------------------------------------------------------------
template <typename ID, bool EXPAND>
__global__ void __launch_bounds__(TPB)
count_direct_kernel(const uint8_t *__restrict__ bases, uint64_t nbytes, int k, int canonical,
                    unsigned long long *__restrict__ table, DevCounters *ctr)
{
    __shared__ TileLds<EXPAND> L;
    __shared__ unsigned long long s_tot[1];
    const int j = threadIdx.x;
    if (j < 1) s_tot[j] = 0;
    uint32_t nbad;
    stage_tile(L, bases, nbytes, blockIdx.x, &nbad, UniformStarts(batch_uniform_len(ctr), TPB), ctr);
    __syncthreads();

    const uint64_t idmask = (k == 32) ? ~0ull : ((1ull << (2 * k)) - 1ull);
    const uint32_t kmask = (k >= 32) ? 0xFFFFFFFFu : ((1u << k) - 1u);
    const uint32_t k1mask = kmask >> 1;
    const IdParams<ID> idp(k, canonical);
    unsigned long long emitted = 0;
    ID cur_id = 0;
    uint32_t cur_cnt = 0;

#pragma unroll 1
    for (int q = 0; q < CHUNKS_PER_THREAD; q++) {
        const int c = j + q * TPB;
        const Hood h = load_hood(L, c);
        uint32_t N32 = 0;
        if (EXPAND) N32 = (L.nn[c] & 0xFFFFu) | (L.nn[c + 1] << 16);
#pragma unroll
        for (int i = 0; i < 16; i++) {
            const bool crosses = window_crosses(h, i, k1mask);
            const uint32_t vwin = (h.V >> i) & kmask;
            const bool valid = (vwin == 0 && !crosses);
            const ID id = valid ? idp.id(h, i) : (ID)0;
            const bool extend = valid && cur_cnt != 0 && id == cur_id;       // in-lane run merging (homopolymers)
            const uint32_t fcnt = (valid && !extend) ? cur_cnt : 0u;         // a new run starts: flush the previous one
            const ID fid = cur_id;
            if (valid) {
                if (extend) cur_cnt++; else { cur_id = id; cur_cnt = 1; }
                emitted++;
            }
            global_count_add(table, (uint64_t)fid, fcnt);                    // convergent: all lanes call it
            if (EXPAND && !valid && !crosses) {
                const uint32_t nwin = (N32 >> i) & kmask;
                if (nwin == vwin)       // every non-ACGT base of the window is an N, and all of it exists
                    expand_n_window(table, h.F(), i, k, canonical, idmask, nwin, &emitted, ctr);
            }
        }
    }
    global_count_add(table, (uint64_t)cur_id, cur_cnt);

    unsigned long long wt = wave_sum(emitted);
    if ((j & 63) == 0 && wt) atomicAdd(&s_tot[0], wt);
    count_bad_and_marks(nbad, ctr);
    __syncthreads();
    if (j == 0 && s_tot[0]) __hip_atomic_fetch_add(&ctr->total_kmers, s_tot[0], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

// ---------------------------------------------------------------------------------
// shred for one record: id + validity per window position (kmer.py:573-577)
// ---------------------------------------------------------------------------------
__global__ void __launch_bounds__(TPB)
shred_kernel(const uint8_t *__restrict__ bases, uint64_t nbytes, int k, int canonical,
             unsigned long long *__restrict__ ids /* nbytes entries; ~0 where no window */, DevCounters *ctr)
{
    __shared__ TileLds<false> L;
    const int j = threadIdx.x;
    uint32_t nbad;
    stage_tile(L, bases, nbytes, blockIdx.x, &nbad, UniformStarts(batch_uniform_len(ctr), TPB), ctr);
    __syncthreads();
    const uint64_t idmask = (1ull << (2 * k)) - 1ull;
    const uint32_t kmask = (1u << k) - 1u;
    const uint32_t k1mask = kmask >> 1;
    for (int q = 0; q < CHUNKS_PER_THREAD; q++) {
        const int c = j + q * TPB;
        const Hood h = load_hood(L, c);
        const uint64_t p0 = ((uint64_t)blockIdx.x * TILE_CHUNKS + (uint64_t)c) * 16ull;
#pragma unroll
        for (int i = 0; i < 16; i++) {
            if (p0 + i >= nbytes) break;
            const bool crosses = window_crosses(h, i, k1mask);
            const uint32_t vwin = (h.V >> i) & kmask;
            ids[p0 + i] = (vwin == 0 && !crosses) ? window_id64(h, i, k, canonical, idmask) : ~0ull;
        }
    }
    count_bad_and_marks(nbad, ctr);
}

// ---------------------------------------------------------------------------------
// count_nonzero + sum of the table (parse.py:141; kmerdb/__init__.py:1901-1902)
// ---------------------------------------------------------------------------------
__global__ void __launch_bounds__(256)
stats_kernel(const unsigned long long *__restrict__ table, uint64_t nbins, DevCounters *ctr)
{
    unsigned long long nz = 0, sum = 0;
    const uint64_t stride = (uint64_t)gridDim.x * blockDim.x;
    for (uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; i < nbins; i += stride) {
        unsigned long long v = table[i];
        nz += (v != 0);
        sum += v;
    }
    nz = wave_sum(nz);
    sum = wave_sum(sum);
    if ((threadIdx.x & 63) == 0) {
        if (nz) __hip_atomic_fetch_add(&ctr->unique, nz, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if (sum) __hip_atomic_fetch_add(&ctr->sum, sum, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
}

// ---------------------------------------------------------------------------------
// nullomer_array = the ids whose count is zero, ascending (kmerdb/parse.py:139-140: np.array(range(4^k))[counts == 0]) by
// stream compaction on the device -- the host's np.flatnonzero over the 2^30 bins of k = 15 took 1.5 s.
// Tiles of NULL_TILE bins (8 consecutive bins per lane); the vector is cut into ranges of NULL_RANGE_TILES tiles:
//   null_count_kernel   zeros per tile                                          (one sweep of the vector)
//   null_scan_kernel    one workgroup per range: exclusive scan of its tiles' counts, the range's total
//   null_write_kernel   a tile's zero ids, in order, behind its offset           (a second sweep, range by range)
// ---------------------------------------------------------------------------------
constexpr int NULL_TPB = 256, NULL_PER_LANE = 8, NULL_TILE = NULL_TPB * NULL_PER_LANE;       // 2048 bins = 16 KiB of the vector
constexpr uint32_t NULL_RANGE_TILES = 1u << 14;                                               // a range: 2^25 bins, at most 256 MiB of ids

__device__ __forceinline__ uint32_t null_mask8(const unsigned long long *__restrict__ table, uint64_t i0, uint64_t nbins)
{
    uint32_t m = 0;
    if (i0 + NULL_PER_LANE <= nbins) {
#pragma unroll
        for (int j = 0; j < NULL_PER_LANE; j += 2) {
            const ulonglong2 v = *reinterpret_cast<const ulonglong2 *>(table + i0 + j);
            m |= (uint32_t)(v.x == 0) << j | (uint32_t)(v.y == 0) << (j + 1);
        }
    } else {
        for (int j = 0; j < NULL_PER_LANE; j++) if (i0 + j < nbins && table[i0 + j] == 0) m |= 1u << j;
    }
    return m;
}

__global__ void __launch_bounds__(NULL_TPB)
null_count_kernel(const unsigned long long *__restrict__ table, uint64_t nbins, uint64_t ntiles, uint32_t *__restrict__ tile_counts)
{
    __shared__ uint32_t wsum[NULL_TPB / 64];
    for (uint64_t tile = blockIdx.x; tile < ntiles; tile += gridDim.x) {
        const uint32_t m = null_mask8(table, tile * NULL_TILE + (uint64_t)threadIdx.x * NULL_PER_LANE, nbins);
        uint32_t c = (uint32_t)__popc(m);
        for (int o = 32; o > 0; o >>= 1) c += (uint32_t)__shfl_xor((int)c, o, 64);
        if ((threadIdx.x & 63) == 0) wsum[threadIdx.x >> 6] = c;
        __syncthreads();
        if (threadIdx.x == 0) tile_counts[tile] = wsum[0] + wsum[1] + wsum[2] + wsum[3];
        __syncthreads();
    }
}

// tile_counts[t] -> offset of tile t inside its range; range_totals[r] = zeros of range r
__global__ void __launch_bounds__(1024)
null_scan_kernel(uint32_t *__restrict__ tile_counts, uint64_t ntiles, unsigned long long *__restrict__ range_totals)
{
    __shared__ uint32_t part[1024];
    const uint64_t t0 = (uint64_t)blockIdx.x * NULL_RANGE_TILES;
    const uint32_t n = (uint32_t)(ntiles - t0 < NULL_RANGE_TILES ? ntiles - t0 : NULL_RANGE_TILES);
    constexpr uint32_t PER = NULL_RANGE_TILES / 1024;                   // consecutive tiles per thread
    const uint32_t a = threadIdx.x * PER;
    uint32_t s = 0;
    for (uint32_t j = 0; j < PER; j++) if (a + j < n) s += tile_counts[t0 + a + j];
    part[threadIdx.x] = s;
    __syncthreads();
    for (uint32_t d = 1; d < 1024; d <<= 1) {                           // inclusive scan of the 1024 partial sums
        const uint32_t v = threadIdx.x >= d ? part[threadIdx.x - d] : 0u;
        __syncthreads();
        part[threadIdx.x] += v;
        __syncthreads();
    }
    uint32_t run = part[threadIdx.x] - s;
    for (uint32_t j = 0; j < PER; j++)
        if (a + j < n) { const uint32_t c = tile_counts[t0 + a + j]; tile_counts[t0 + a + j] = run; run += c; }
    if (threadIdx.x == 1023) range_totals[blockIdx.x] = part[1023];
}

// the zero ids of the tiles [tile0, tile0 + ntiles_here) of one range, written to out[tile offset ...] in ascending order
__global__ void __launch_bounds__(NULL_TPB)
null_write_kernel(const unsigned long long *__restrict__ table, uint64_t nbins, uint64_t tile0, uint32_t ntiles_here,
                  const uint32_t *__restrict__ tile_offs, unsigned long long *__restrict__ out)
{
    __shared__ uint32_t wsum[NULL_TPB / 64];
    for (uint32_t tt = blockIdx.x; tt < ntiles_here; tt += gridDim.x) {
        const uint64_t tile = tile0 + tt;
        const uint64_t i0 = tile * NULL_TILE + (uint64_t)threadIdx.x * NULL_PER_LANE;
        const uint32_t m = null_mask8(table, i0, nbins);
        const uint32_t c = (uint32_t)__popc(m);
        uint32_t incl = c;                                              // inclusive scan over the wave
        for (int o = 1; o < 64; o <<= 1) { const uint32_t v = (uint32_t)__shfl_up((int)incl, o, 64); if ((int)(threadIdx.x & 63) >= o) incl += v; }
        if ((threadIdx.x & 63) == 63) wsum[threadIdx.x >> 6] = incl;
        __syncthreads();
        uint32_t base = tile_offs[tile];
        for (uint32_t w = 0; w < (threadIdx.x >> 6); w++) base += wsum[w];
        uint32_t at = base + incl - c;
        for (uint32_t mm = m; mm; mm &= mm - 1) out[at++] = i0 + (uint64_t)(__ffs((int)mm) - 1);
        __syncthreads();
    }
}

// ---------------------------------------------------------------------------------
// `counts = counts + counts_` across the files of a samplesheet, on the device (kmerdb/__init__.py:1888-1891):
// acc += table; the file's own count_nonzero / Sum (its metadata, parse.py:141) are taken in the same sweep and
// the file vector is cleared for the next file.
// ---------------------------------------------------------------------------------
__global__ void __launch_bounds__(256)
fold_kernel(unsigned long long *__restrict__ table, unsigned long long *__restrict__ acc, uint64_t nbins, DevCounters *ctr)
{
    unsigned long long nz = 0, sum = 0;
    const uint64_t stride = (uint64_t)gridDim.x * blockDim.x * 2;
    for (uint64_t i = ((uint64_t)blockIdx.x * blockDim.x + threadIdx.x) * 2; i < nbins; i += stride) {
        if (i + 1 < nbins) {
            ulonglong2 v = *reinterpret_cast<const ulonglong2 *>(table + i);
            if (v.x | v.y) {
                ulonglong2 a = *reinterpret_cast<const ulonglong2 *>(acc + i);
                a.x += v.x; a.y += v.y;
                *reinterpret_cast<ulonglong2 *>(acc + i) = a;
                *reinterpret_cast<ulonglong2 *>(table + i) = make_ulonglong2(0ull, 0ull);
                nz += (v.x != 0) + (v.y != 0);
                sum += v.x + v.y;
            }
        } else {
            const unsigned long long v = table[i];
            if (v) { acc[i] += v; table[i] = 0; nz++; sum += v; }
        }
    }
    nz = wave_sum(nz);
    sum = wave_sum(sum);
    if ((threadIdx.x & 63) == 0) {
        if (nz) __hip_atomic_fetch_add(&ctr->unique, nz, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if (sum) __hip_atomic_fetch_add(&ctr->sum, sum, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
}

// ---------------------------------------------------------------------------------
// kdb_reduce: own[i] += Sum over the peers' vectors at i, for i in [lo, hi) (lo, hi even or hi == nbins).  The peers'
// vectors live on other devices (xGMI peer access) or on this one; every engine runs this on its own slice at once.
// ---------------------------------------------------------------------------------
struct ReducePeers { const unsigned long long *p[15]; int n; };

__global__ void __launch_bounds__(256)
reduce_slice_kernel(unsigned long long *__restrict__ own, ReducePeers peers, uint64_t lo, uint64_t hi)
{
    const uint64_t stride = (uint64_t)gridDim.x * blockDim.x * 2;
    for (uint64_t i = lo + ((uint64_t)blockIdx.x * blockDim.x + threadIdx.x) * 2; i < hi; i += stride) {
        if (i + 1 < hi) {
            ulonglong2 a = *reinterpret_cast<const ulonglong2 *>(own + i);
            for (int q = 0; q < peers.n; q++) {
                const ulonglong2 v = *reinterpret_cast<const ulonglong2 *>(peers.p[q] + i);
                a.x += v.x; a.y += v.y;
            }
            *reinterpret_cast<ulonglong2 *>(own + i) = a;
        } else {
            unsigned long long a = own[i];
            for (int q = 0; q < peers.n; q++) a += peers.p[q][i];
            own[i] = a;
        }
    }
}

}  // namespace kdb
