/*
 * oracle/kmer_oracle.c -- TEST INFRASTRUCTURE, NOT PRODUCT CODE.
 *
 * A plain-C, CPU-only restatement of the k-mer counting semantics of the
 * reference (MatthewRalston/kmerdb v0.9.6).  It exists so that the HIP
 * engine in kmerdb_amd/csrc/ can be checked bit-for-bit on a box where the
 * reference's Python cannot run.  Only tests/, __graft_entry__.smoke() and
 * bench.py's cpu_baseline leg may load this library; the product path
 * (kmerdb_amd/) never does.
 *
 * The code deliberately follows the reference's *per-window* formulation
 * (recompute every window from scratch, forward loop then a loop over the
 * reverse complement, then min) instead of the rolling / position-parallel
 * bit tricks the GPU kernels use, so that the two are independent
 * derivations of the same counts.
 *
 * Pinning: see oracle/README.md and tests/test_oracle_golden.py -- this
 * restatement reproduces (a) the reference's own fixture pair
 * test/data/Cacetobutylicum_ATCC824.fasta.gz -> test_Cac_ATCC824.8.kdb
 * (forward, k=8, all 65,536 bins) and (b) vectors generated in the build
 * container by the reference's kmer.py/parse.py (tests/golden/).
 *
 * Reference lines restated (paths relative to /root/reference/):
 *   kmerdb/kmer.py:44-49     letterToBinaryNA  (A=0,C=1,G=2,T=3, uppercase only)
 *   kmerdb/kmer.py:234-317   kmer_to_id        (fwd id, rc id, min if canonical, None on 'N')
 *   kmerdb/kmer.py:430-483   validate_seqRecord_and_detect_IUPAC (len >= k)
 *   kmerdb/kmer.py:489-577   shred             (window loop, N: drop or 4^m expansion)
 *   kmerdb/kmer.py:586-621   substitute_residue_with_chars (all fills of the N's)
 *   kmerdb/parse.py:117-137  parsefile         (counts[id] += 1, total_kmers += 1)
 *   kmerdb/graph.py:108-216  make_edges_from_kmerids (consecutive k-mer pairs, N-free reads)
 */
#include <stdint.h>
#include <stddef.h>
#include <stdlib.h>
#include <string.h>
#include <pthread.h>

#define KDBO_OK            0
#define KDBO_SHORT_READ    1   /* kmer.py:461-463: ValueError, sequence shorter than k */
#define KDBO_BAD_RESIDUE   2   /* kmer.py:309 KeyError / :170 NameError: anything not in "ACGTN" */
#define KDBO_BAD_ARG       3

#define KDBO_N_DROP   0        /* replace_with_none=True  (kmer.py:541-544) */
#define KDBO_N_EXPAND 1        /* replace_with_none=False (kmer.py:545-565) */

/* kmer.py:44-49 */
static inline int code_of(uint8_t c)
{
    switch (c) {
    case 65: return 0;  /* A */
    case 67: return 1;  /* C */
    case 71: return 2;  /* G */
    case 84: return 3;  /* T */
    default: return -1;
    }
}

/* Bio.Seq.reverse_complement restricted to ACGT: complement then reverse. */
static inline uint8_t complement_of(uint8_t c)
{
    switch (c) {
    case 'A': return 'T';
    case 'C': return 'G';
    case 'G': return 'C';
    case 'T': return 'A';
    default:  return c;
    }
}

/*
 * kmer.py:234-317.  s has exactly k bytes.
 * returns 0 and *id_out on success, 1 if the k-mer contains 'N' (reference
 * returns None, :287-289 -- BEFORE it looks at any other letter: a window
 * that holds an N and another IUPAC code is None too), 2 on a non-ACGT byte
 * in a window without N (reference raises KeyError at :309).
 */
int kdbo_kmer_to_id(const uint8_t *s, int k, int canonicalize, uint64_t *id_out)
{
    uint64_t idx1 = 0, idx2 = 0;
    if (k < 1 || k > 32) return KDBO_BAD_ARG;
    for (int j = 0; j < k; j++) if (s[j] == 'N') return 1;      /* :287-289 */
    for (int j = 0; j < k; j++) if (code_of(s[j]) < 0) return 2; /* :309 KeyError */
    for (int j = 0; j < k; j++) {                      /* :307-309 */
        idx1 = idx1 << 2;
        idx1 = idx1 | (uint64_t)code_of(s[j]);
    }
    for (int j = k - 1; j >= 0; j--) {                 /* :310-312, over the reverse complement */
        idx2 = idx2 << 2;
        idx2 = idx2 | (uint64_t)code_of(complement_of(s[j]));
    }
    *id_out = canonicalize ? (idx1 < idx2 ? idx1 : idx2) : idx1;   /* :314-317 */
    return 0;
}

/* kmer.py:320-363 (nucleic acid branch): id -> k-mer string, out has k bytes. */
void kdbo_id_to_kmer(uint64_t id, int k, uint8_t *out)
{
    static const uint8_t letters[4] = { 'A', 'C', 'G', 'T' };
    for (int i = 0; i < k; i++) {
        out[k - 1 - i] = letters[id & 3u];
        id >>= 2;
    }
}

/* the ten IUPAC nucleotide codes besides ACGT and N (kmer.py: IUPAC_NA_DOUBLETS / _TRIPLETS); uppercase only */
static inline int is_iupac10(uint8_t c)
{
    switch (c) {
    case 'R': case 'Y': case 'S': case 'W': case 'K': case 'M': case 'B': case 'D': case 'H': case 'V': return 1;
    default: return 0;
    }
}

typedef void (*emit_fn)(void *ctx, uint64_t id, uint64_t pos);

/*
 * kmer.py:489-577 for one record.  Calls emit(ctx, id, pos) once per id in
 * the order the reference appends them (window order; for an N window, one
 * id per fill, all with pos = window start).
 */
static int shred_record(const uint8_t *seq, uint64_t len, int k, int canonicalize,
                        int n_mode, emit_fn emit, void *ctx)
{
    uint8_t buf[32];
    if (len < (uint64_t)k) return KDBO_SHORT_READ;     /* :461-463 */
    /* :519-521 (is_sequence_na / validate_seqRecord_and_detect_IUPAC): the whole record is validated before any window is
     * emitted -- letters outside the IUPAC nucleotide alphabet raise here (NameError :170 / ValueError :473).  The ten IUPAC
     * codes besides N pass this check; what becomes of them is decided window by window below. */
    for (uint64_t i = 0; i < len; i++)
        if (seq[i] != 'N' && code_of(seq[i]) < 0 && !is_iupac10(seq[i])) return KDBO_BAD_RESIDUE;
    /* replace_with_none=False: a window with such a code is handed to _substitute_na_doublets / _triplets (:545-555, :630-851),
     * which leave a code that occurs once in the window in place (:612 replaces "N", not the code) -- kmer_to_id then raises
     * KeyError -- and raise NameError for two different codes in one window: every fixture of tests/golden/iupac_next_to_n.json
     * raises except a record whose every window holds one code at least twice ("ANRRNA", k = 4).  This restatement raises for
     * every record with such a code in this mode (DESIGN.md section 1 names the one shape it does not reproduce). */
    if (n_mode == KDBO_N_EXPAND)
        for (uint64_t i = 0; i < len; i++) if (is_iupac10(seq[i])) return KDBO_BAD_RESIDUE;
    for (uint64_t i = 0; i + (uint64_t)k <= len; i++) {                  /* :526 */
        const uint8_t *w = seq + i;
        uint64_t id;
        int rc = kdbo_kmer_to_id(w, k, canonicalize, &id);               /* :528 */
        if (rc == 0) { emit(ctx, id, i); continue; }                     /* :538-540 */
        if (rc != 1) return KDBO_BAD_RESIDUE;
        if (n_mode == KDBO_N_DROP) continue;                             /* :542-544 */
        /* :559-565 + :586-621: every fill of the m N's with A,C,G,T */
        int npos[32], m = 0;
        for (int j = 0; j < k; j++) if (w[j] == 'N') npos[m++] = j;
        memcpy(buf, w, (size_t)k);
        uint64_t nfill = 1ull << (2 * m);
        for (uint64_t f = 0; f < nfill; f++) {
            static const uint8_t letters[4] = { 'A', 'C', 'G', 'T' };
            for (int j = 0; j < m; j++) buf[npos[j]] = letters[(f >> (2 * j)) & 3u];
            kdbo_kmer_to_id(buf, k, canonicalize, &id);
            emit(ctx, id, i);
        }
    }
    return KDBO_OK;
}

struct count_ctx { uint64_t *counts; uint64_t total; int atomic; };

static void emit_count(void *p, uint64_t id, uint64_t pos)
{
    struct count_ctx *c = (struct count_ctx *)p;
    (void)pos;
    if (c->atomic) __atomic_fetch_add(&c->counts[id], 1ull, __ATOMIC_RELAXED);
    else c->counts[id] += 1;                            /* parse.py:135 */
    c->total += 1;                                      /* parse.py:136 */
}

/*
 * parse.py:117-137 over records given as a flat byte buffer plus nreads+1
 * offsets (record r = bases[offsets[r] .. offsets[r+1])).  counts has 4^k
 * entries and is ADDED to (callers zero it for parsefile semantics).
 * On error returns the status and the index of the offending record.
 */
int kdbo_count(const uint8_t *bases, const uint64_t *offsets, uint64_t nreads,
               int k, int canonicalize, int n_mode,
               uint64_t *counts, uint64_t *total_kmers, uint64_t *err_read)
{
    struct count_ctx c = { counts, 0, 0 };
    if (k < 1 || k > 31) return KDBO_BAD_ARG;
    for (uint64_t r = 0; r < nreads; r++) {
        int rc = shred_record(bases + offsets[r], offsets[r + 1] - offsets[r],
                              k, canonicalize, n_mode, emit_count, &c);
        if (rc != KDBO_OK) { if (err_read) *err_read = r; return rc; }
    }
    if (total_kmers) *total_kmers = c.total;
    return KDBO_OK;
}

/* ---- the same loop sharded over host threads (cpu_baseline "all cores") ---- */

struct mt_job {
    const uint8_t *bases; const uint64_t *offsets; uint64_t r0, r1;
    int k, canonicalize, n_mode; uint64_t *counts; uint64_t total; int status; uint64_t err_read;
};

static void *mt_worker(void *p)
{
    struct mt_job *j = (struct mt_job *)p;
    struct count_ctx c = { j->counts, 0, 1 };
    j->status = KDBO_OK;
    for (uint64_t r = j->r0; r < j->r1; r++) {
        int rc = shred_record(j->bases + j->offsets[r], j->offsets[r + 1] - j->offsets[r],
                              j->k, j->canonicalize, j->n_mode, emit_count, &c);
        if (rc != KDBO_OK) { j->status = rc; j->err_read = r; break; }
    }
    j->total = c.total;
    return NULL;
}

int kdbo_count_mt(const uint8_t *bases, const uint64_t *offsets, uint64_t nreads,
                  int k, int canonicalize, int n_mode, int nthreads,
                  uint64_t *counts, uint64_t *total_kmers, uint64_t *err_read)
{
    if (nthreads < 1) nthreads = 1;
    if (k < 1 || k > 31) return KDBO_BAD_ARG;
    struct mt_job *jobs = (struct mt_job *)calloc((size_t)nthreads, sizeof *jobs);
    pthread_t *tids = (pthread_t *)calloc((size_t)nthreads, sizeof *tids);
    if (!jobs || !tids) { free(jobs); free(tids); return KDBO_BAD_ARG; }
    for (int t = 0; t < nthreads; t++) {
        jobs[t] = (struct mt_job){ bases, offsets, nreads * (uint64_t)t / (uint64_t)nthreads,
                                   nreads * (uint64_t)(t + 1) / (uint64_t)nthreads,
                                   k, canonicalize, n_mode, counts, 0, 0, 0 };
        pthread_create(&tids[t], NULL, mt_worker, &jobs[t]);
    }
    int status = KDBO_OK; uint64_t total = 0;
    for (int t = 0; t < nthreads; t++) {
        pthread_join(tids[t], NULL);
        total += jobs[t].total;
        if (jobs[t].status != KDBO_OK && status == KDBO_OK) {
            status = jobs[t].status; if (err_read) *err_read = jobs[t].err_read;
        }
    }
    if (total_kmers) *total_kmers = total;
    free(jobs); free(tids);
    return status;
}

/* ---- shred: ids + positions of one record (kmer.py:573-577) ---- */

struct shred_ctx { uint64_t *ids; uint64_t *pos; uint64_t n, cap; };

static void emit_shred(void *p, uint64_t id, uint64_t pos)
{
    struct shred_ctx *c = (struct shred_ctx *)p;
    if (c->n < c->cap) { c->ids[c->n] = id; c->pos[c->n] = pos; }
    c->n += 1;
}

/* returns status; *n_out = number of ids the reference would return (may exceed cap) */
int kdbo_shred(const uint8_t *seq, uint64_t len, int k, int canonicalize, int n_mode,
               uint64_t *ids, uint64_t *pos, uint64_t cap, uint64_t *n_out)
{
    struct shred_ctx c = { ids, pos, 0, cap };
    if (k < 1 || k > 31) return KDBO_BAD_ARG;
    int rc = shred_record(seq, len, k, canonicalize, n_mode, emit_shred, &c);
    if (n_out) *n_out = c.n;
    return rc;
}

/*
 * graph.py:108-216 restricted to N-free records: for a
 * record with n = L-k+1 k-mers the reference emits the rows
 * (pos j-1, id[j-1], pos j, id[j]) for j = 1..n-1.  The aggregate of those
 * rows is a weighted edge list; this oracle accumulates the weight of the
 * edge into edge_counts[(id1 << 2k) | id2] is NOT dense-feasible, so it is
 * keyed the way the build keys it: by the forward (k+1)-mer id at j-1
 * (id1 and id2 are both functions of that (k+1)-mer and of `canonicalize`).
 * edge_counts has 4^(k+1) entries and is ADDED to; counts (4^k) likewise.
 */
int kdbo_count_edges(const uint8_t *bases, const uint64_t *offsets, uint64_t nreads,
                     int k, uint64_t *edge_counts, uint64_t *total_edges, uint64_t *err_read)
{
    uint64_t total = 0;
    if (k < 1 || k > 30) return KDBO_BAD_ARG;
    for (uint64_t r = 0; r < nreads; r++) {
        const uint8_t *seq = bases + offsets[r];
        uint64_t len = offsets[r + 1] - offsets[r];
        if (len < (uint64_t)k) { if (err_read) *err_read = r; return KDBO_SHORT_READ; }
        for (uint64_t i = 0; i < len; i++)
            if (code_of(seq[i]) < 0) { if (err_read) *err_read = r; return KDBO_BAD_RESIDUE; }
        for (uint64_t j = 1; j + (uint64_t)k <= len; j++) {
            uint64_t e;
            kdbo_kmer_to_id(seq + j - 1, k + 1, 0, &e);   /* forward (k+1)-mer = (kmer j-1, kmer j) */
            edge_counts[e] += 1;
            total += 1;
        }
    }
    if (total_edges) *total_edges = total;
    return KDBO_OK;
}
