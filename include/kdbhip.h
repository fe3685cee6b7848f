/*
 * kdbhip.h -- C ABI of libkdbhip.so, the MI355X (gfx950) k-mer counting engine
 * that replaces the inner loops of kmerdb's `profile` hot path.
 *
 * The reference (MatthewRalston/kmerdb v0.9.6) is pure Python and has no FFI:
 * its boundary for this path is the Python function
 *     kmerdb/parse.py:90   parsefile(filepath, k, replace_with_none, canonicalize)
 * whose body (parse.py:117-137) loops  kmer.shred -> kmer.kmer_to_id  per
 * window and does  counts[kmer_id] += 1.  Every entry point below cites the
 * reference lines whose work it takes over.  Plain pointers and sizes only;
 * no torch / numpy types.  The ctypes stub a kmerdb maintainer would add is
 * shown in INTEGRATION.md; kmerdb_amd/_abi.py is that stub.
 *
 * Threading: one producer thread per engine handle.  All functions return an
 * int status (KDB_OK == 0); kdb_last_error() gives a thread-local message.
 */
#ifndef KDBHIP_H
#define KDBHIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define KDB_ABI_VERSION 6

/* status codes */
#define KDB_OK               0
#define KDB_ERR_ARG          1   /* bad argument (reference: TypeError / ValueError at parse.py:109-116) */
#define KDB_ERR_HIP          2   /* a HIP runtime call failed; message has hipGetErrorString */
#define KDB_ERR_SHORT_READ   3   /* a record shorter than k   (reference raises: kmer.py:461-463) */
#define KDB_ERR_BAD_RESIDUE  4   /* a byte outside "ACGTN"    (reference raises: kmer.py:309 KeyError, :170 NameError) */
#define KDB_ERR_NOMEM        5   /* 4^k table or staging does not fit on the device */
#define KDB_ERR_STATE        6   /* call sequence error */

/* N handling == the reference's `replace_with_none` flag (kmer.py:541-565) */
#define KDB_N_DROP    0          /* replace_with_none=True : windows containing N emit nothing          */
#define KDB_N_EXPAND  1          /* replace_with_none=False: a window with m N's emits all 4^m fills    */

typedef struct kdb_engine kdb_engine;

int         kdb_abi_version(void);
const char *kdb_last_error(void);
int         kdb_device_count(int *n_out);

/*
 * Create an engine on `device_id` holding a dense 4^k uint64 count vector in
 * HBM, zeroed.  Replaces  counts = np.zeros(4**k, dtype="uint64")  parse.py:117-120.
 * `d_table` may be NULL (engine hipMallocs and owns the vector) or a device
 * pointer to 4^k uint64 owned by the caller (e.g. a torch tensor that will be
 * handed to RCCL afterwards); it is zeroed by kdb_create either way.
 * 1 <= k <= 17 (4^17 * 8 B = 128 GiB; larger tables do not fit 288 GB).
 */
int kdb_create(int k, int canonicalize, int n_mode, int device_id, void *d_table, kdb_engine **out);
int kdb_destroy(kdb_engine *e);

/* Zero the count vector and the running totals (a new parsefile call). */
int kdb_reset(kdb_engine *e);

/*
 * Count every k-mer of `nreads` records.  Replaces the loop parse.py:128-137
 * (shred kmer.py:489-577 + kmer_to_id kmer.py:234-317 + counts[id] += 1).
 *   bases        : raw ASCII residues of all records, concatenated (no separators)
 *   read_offsets : nreads+1 offsets; record r = bases[read_offsets[r] .. read_offsets[r+1])
 * Windows never span records.  Asynchronous: the call returns once the data is
 * staged into pinned memory; copies and kernels run on the engine's streams,
 * double-buffered.  The caller's buffers may be reused as soon as it returns.
 * Records longer than a staging buffer are tiled with a k-1 base overlap.
 */
int kdb_submit(kdb_engine *e, const uint8_t *bases, size_t nbytes,
               const uint64_t *read_offsets, size_t nreads);

/*
 * Same as kdb_submit for buffers in pinned host memory (kdb_host_alloc / hipHostMalloc): the DMA reads
 * `bases` directly, with no staging copy, so `bases` must stay valid and unmodified until kdb_sync /
 * kdb_finish returns -- or until two further kdb_submit_pinned calls have returned (call N first waits for the
 * copies of call N-2, so cycling through three buffers is safe).  `read_offsets` may be reused at once.
 */
int kdb_submit_pinned(kdb_engine *e, const uint8_t *bases, size_t nbytes,
                      const uint64_t *read_offsets, size_t nreads);
/*
 * kdb_submit / kdb_submit_pinned with flags.  KDB_SUBMIT_CONTINUES: record 0 of this call is the next piece of the
 * LAST record of the previous call (a FASTA record longer than the reader's block, kmerdb/parse.py:50-85 streams
 * whole genomes): the caller starts the piece with the last k-1 residues it submitted, so that every window is seen
 * once; the piece gets no record start and is not subject to the "shorter than k" check.
 */
#define KDB_SUBMIT_PINNED     1
#define KDB_SUBMIT_CONTINUES  2
int kdb_submit_ex(kdb_engine *e, const uint8_t *bases, size_t nbytes,
                  const uint64_t *read_offsets, size_t nreads, int flags);
int kdb_host_alloc(void **out, size_t nbytes);     /* pinned host memory for kdb_submit_pinned */
int kdb_host_free(void *p);

/*
 * Same, for inputs already resident in HBM on the engine's device (device
 * pointers; d_bases 16-byte aligned).  The LDS-histogram paths (the default) only
 * read both buffers: a ragged batch's record starts are taken from the offsets.
 * The direct-atomics kernel ("algo" 1, and the fallback when the scatter
 * scratch does not fit in HBM) sets bit 7 of the first byte of every record of
 * a ragged batch in d_bases (its in-HBM record-boundary mark; the low 7 bits
 * are untouched) while it reads the batch and clears it again afterwards, so the
 * buffer is unchanged once the stream has drained and may be submitted again
 * with other offsets.  The buffers must stay alive until kdb_sync / kdb_finish
 * returns.
 */
int kdb_submit_device(kdb_engine *e, void *d_bases, size_t nbytes,
                      const void *d_read_offsets, size_t nreads);

/*
 * kdb_submit_device for a buffer the engine must not write (shared read-only between engines or streams).  On the
 * LDS-histogram paths that is every batch.  Where the direct-atomics kernel has to run ("algo" 1, or no room for the
 * scatter scratch) a batch of ragged records makes kdb_sync / kdb_finish return KDB_ERR_ARG: that kernel marks record
 * starts in the buffer (records of one length need no marks: the usual FASTQ shape).
 */
int kdb_submit_device_const(kdb_engine *e, const void *d_bases, size_t nbytes,
                            const void *d_read_offsets, size_t nreads);

/* Wait for all submitted work; surfaces KDB_ERR_SHORT_READ / KDB_ERR_BAD_RESIDUE. */
int kdb_sync(kdb_engine *e);

/*
 * Sync, then copy the count vector to `counts_out` (4^k uint64, caller-owned,
 * may be NULL to skip the copy) and report the totals parsefile derives at
 * parse.py:139-147:  total_kmers (Sum counts), unique_kmers (count_nonzero).
 * Does not reset the engine: further submits keep accumulating (the
 * `counts = counts + counts_` of kmerdb/__init__.py:1890 stays on the device).
 */
int kdb_finish(kdb_engine *e, uint64_t *counts_out, uint64_t *total_kmers, uint64_t *unique_kmers);

/*
 * Like kdb_finish for a vector the engine did not fill alone -- after the RCCL reduce of SURVEY 8(e) rank 0's
 * vector holds every rank's counts, so Sum(counts) no longer equals what this engine emitted and kdb_finish's
 * internal consistency check does not apply.  Reports Sum(counts) and count_nonzero(counts) of the vector as it is.
 */
int kdb_table_stats(kdb_engine *e, uint64_t *counts_out, uint64_t *sum_out, uint64_t *unique_out);

/*
 * nullomer_array of parse.py:139-140 -- the ids whose count is zero, ascending -- by stream compaction on the device
 * (np.flatnonzero over the 2^30 bins of k = 15 costs the host 1.5 s; the device sweeps the vector twice at HBM speed and
 * sends back only the ids).  Syncs first.  `folded` = 0: the count vector, 1: the samplesheet accumulator (kdb_fold_file).
 * *n_out = number of such ids (4^k - count_nonzero); ids_out may be NULL to ask for the number alone, otherwise it holds
 * `cap` entries (KDB_ERR_ARG if fewer than *n_out, with *n_out set).
 */
int kdb_nullomers(kdb_engine *e, int folded, uint64_t *ids_out, uint64_t cap, uint64_t *n_out);

/*
 * The reduce of SURVEY 8(e) for ONE process that drives several GPUs (an engine per device, records dealt out
 * between them -- a record is counted independently of every other, parse.py:128-137): sum the count vectors of
 * engines[0..n) into engines[root]'s.  All engines are synced first (a KDB_ERR_SHORT_READ / KDB_ERR_BAD_RESIDUE
 * of any of them is returned and nothing is reduced).  The 4^k index range is cut into n slices: engine j sums
 * slice j of every vector over xGMI peer access (all links busy in both directions), then the root collects the
 * n - 1 finished slices -- 2 x (n-1)/n x 4^k x 8 bytes over the root's links instead of (n-1) x 4^k x 8.
 * Integer sums: the result is the single-GPU vector bit for bit.  Afterwards engines[root] also carries the k-mers
 * emitted by all of them, so kdb_finish(engines[root], ...) reports the job's totals; the other engines' vectors
 * are partly overwritten -- kdb_reset them before further use.  Same k for all, 2 <= n <= KDB_REDUCE_MAX, engines
 * distinct; engines on the same device are allowed (tests on a one-GPU box).  KDB_ERR_HIP if two of the devices
 * cannot reach each other's memory.  (One process per GPU -- torch.distributed / RCCL -- is the other form:
 * kmerdb_amd/distributed.py reduces the same vectors with dist.reduce in 1 GiB pieces, see INTEGRATION.md.)
 */
#define KDB_REDUCE_MAX 16
int kdb_reduce(kdb_engine *const *engines, int n, int root);

/*
 * `counts = counts + counts_` over the files of a samplesheet (kmerdb/__init__.py:1888-1891) without leaving HBM.
 * kdb_fold_file: sync; add the engine's vector (one file's counts) to a second, engine-owned 4^k accumulator;
 * report that file's total_kmers / unique_kmers (its per-file metadata, parse.py:141-147); clear the file vector
 * and its totals for the next file -- one sweep.  kdb_finish_folded: copy the accumulator to `counts_out` (may be
 * NULL) and report its Sum / count_nonzero (__init__.py:1901-1902).  kdb_reset clears the accumulator too.
 * KDB_ERR_NOMEM if a second vector does not fit (k = 17): the host layer then sums on the host as before.
 */
int kdb_fold_file(kdb_engine *e, uint64_t *total_kmers, uint64_t *unique_kmers);
/* the same with the accumulator of ANOTHER engine (same k, same device): several engines count files at the same time
 * and fold into one sum; the caller serialises the folds into one accumulator (one at a time). */
int kdb_fold_file_into(kdb_engine *e, kdb_engine *acc, uint64_t *total_kmers, uint64_t *unique_kmers);
int kdb_finish_folded(kdb_engine *e, uint64_t *counts_out, uint64_t *total_kmers, uint64_t *unique_kmers);

/*
 * Device pointer of the count vector and its length 4^k (for the RCCL reduce by the host layer).  The vector
 * holds everything submitted so far only after kdb_sync / kdb_finish: submits are asynchronous, and for k >= 13
 * the histogram pass over scattered batches is deferred until then (see "defer_flush").
 */
int kdb_table(kdb_engine *e, void **d_table_out, uint64_t *nbins_out);

/* Error detail after KDB_ERR_SHORT_READ / KDB_ERR_BAD_RESIDUE: how many offenders were seen. */
int kdb_error_counts(kdb_engine *e, uint64_t *n_short_reads, uint64_t *n_bad_residues);

/*
 * An engine without a count vector, for kdb_shred / kdb_window_ids only (kmer.shred on single records must not
 * allocate 4^k * 8 bytes: 8 GiB at k = 15).  Counting entry points return KDB_ERR_STATE on it.  1 <= k <= 17.
 */
int kdb_create_ids(int k, int canonicalize, int device_id, kdb_engine **out);

/*
 * kmer.shred for one record on the device (kmer.py:489-577), N-free windows
 * only (n_mode is ignored: windows containing N emit nothing, as with
 * replace_with_none=True).  Writes one id per emitted window, in window order,
 * and its position.  ids_out/pos_out hold `cap` entries; *n_out is the number
 * of windows emitted (<= nbytes-k+1).
 */
int kdb_shred(kdb_engine *e, const uint8_t *seq, size_t nbytes,
              uint64_t *ids_out, uint64_t *pos_out, size_t cap, size_t *n_out);

/*
 * The id of the window starting at every residue position of a batch of records (the per-record lists
 * kmer.shred returns, kmer.py:573-577, laid out by position): ids_out[p] for p in [0, nbytes) is the k-mer id
 * of the window starting at residue p, or ~0 where no counted window starts (too close to the record's end,
 * or the window contains N).  Used by the De Bruijn edge list (graph.py:219-332).  Synchronous; nbytes <= 2^30.
 */
int kdb_window_ids(kdb_engine *e, const uint8_t *bases, size_t nbytes,
                   const uint64_t *read_offsets, size_t nreads, uint64_t *ids_out);

/*
 * Host-side record splitter (no GPU work): what Bio.SeqIO.parse does for kmerdb/parse.py:50-85 on this path.
 * FASTQ: parses whole records of text[0, n) into bases_out (concatenated residues) and offsets_out
 * (nreads+1 entries); *consumed_out = bytes of text used (stops before a trailing partial record unless at_eof).
 * The usual four-line records take the fast path; a text that is not of that form goes through the general grammar of
 * Bio.SeqIO's FASTQ reader (sequence and quality wrapped over several lines, a '+' line that repeats the title, quality
 * lines that start with '@'), and, like that reader, a quality character outside 33..126, differing sequence and quality
 * lengths or captions are malformed input.  kdb_parse_fastq_mt: the same on `nthreads` threads (the text is cut at record
 * starts, every piece is counted, then split into its final place; same outputs, same errors).
 * FASTA: the whole text; sequence lines are concatenated, blanks / CR dropped, text before the first '>' ignored.
 * header_spans_out (optional, 2 per record) = [start, end) of each header line in `text` (for record ids).
 * Malformed input -> KDB_ERR_ARG (the host layer raises ValueError).
 */
int kdb_parse_fastq(const uint8_t *text, size_t n, int at_eof, uint8_t *bases_out, size_t bases_cap,
                    uint64_t *offsets_out, size_t cap_reads, uint64_t *header_spans_out,
                    size_t *nreads_out, size_t *nbases_out, size_t *consumed_out);
int kdb_parse_fastq_mt(const uint8_t *text, size_t n, int at_eof, uint8_t *bases_out, size_t bases_cap,
                       uint64_t *offsets_out, size_t cap_reads, uint64_t *header_spans_out,
                       size_t *nreads_out, size_t *nbases_out, size_t *consumed_out, int nthreads);
int kdb_parse_fasta(const uint8_t *text, size_t n, uint8_t *bases_out, size_t bases_cap,
                    uint64_t *offsets_out, size_t cap_reads, uint64_t *header_spans_out,
                    size_t *nreads_out, size_t *nbases_out);

/*
 * kdb_parse_fasta for a file read in chunks: in_record = the chunk starts inside a record (then record 0 of the output
 * is that record's next piece; 2 = in the middle of one of its lines); header lines are consumed whole, sequence text as far as it goes;
 * *in_record_out = the state to pass with the next chunk.
 */
int kdb_parse_fasta_chunk(const uint8_t *text, size_t n, int at_eof, int in_record, uint8_t *bases_out, size_t bases_cap,
                          uint64_t *offsets_out, size_t cap_reads, uint64_t *header_spans_out,
                          size_t *nreads_out, size_t *nbases_out, size_t *consumed_out, int *in_record_out);

/*
 * Block-parallel inflate of BGZF input (bgzip / Bio.bgzf files; what the reference reads through gzip.open,
 * kmerdb/parse.py:64-72): inflates the whole BGZF members of src[0, n) that fit `cap` with `nthreads` threads, checks
 * their CRC32.  *consumed_out = input bytes used (stops before an incomplete member), *produced_out = bytes written.
 * KDB_ERR_ARG if the input is not BGZF or is corrupt.
 */
int kdb_bgzf_inflate(const uint8_t *src, size_t n, uint8_t *dst, size_t cap, int nthreads,
                     size_t *consumed_out, size_t *produced_out);

/*
 * Index of a BGZF file from its members' headers and trailers (nothing is inflated): coff_out[i] / uoff_out[i] =
 * compressed / uncompressed offset of member i, entry n = the file's sizes; *n_out = n + 1 entries.  With it a rank of
 * a multi-GPU job inflates only the members that hold the blocks it owns (the reference reads the whole file through
 * gzip.open on its one process, kmerdb/parse.py:64-72).  KDB_ERR_ARG if the file is not BGZF; KDB_ERR_NOMEM if `cap`
 * entries do not hold the index (call again with more).
 */
int kdb_bgzf_scan(const char *path, uint64_t *coff_out, uint64_t *uoff_out, size_t cap, size_t *n_out);

/*
 * One gzip stream -- what the reference opens with gzip.open (kmerdb/parse.py:64-72) -- inflated by a thread of its
 * own into a ring of 4 MiB buffers, ahead of the reader: inflating overlaps record splitting, hashing and counting.
 * kdb_gz_read fills `dst` with up to `cap` bytes (fewer only at the end of the stream; 0 = end); concatenated members
 * are concatenated output.  KDB_ERR_ARG for a file that cannot be opened or a corrupt / truncated stream.
 */
typedef struct kdb_gz kdb_gz;
int kdb_gz_open(const char *path, kdb_gz **out);
int kdb_gz_read(kdb_gz *g, uint8_t *dst, size_t cap, size_t *n_out);
int kdb_gz_close(kdb_gz *g);

/*
 * Host-side .kdb row writer (no GPU work): the per-row loop kmerdb/__init__.py:1980-1990 plus
 * Bio.bgzf.BgzfWriter._write_block.  Appends to `path` (which already holds the YAML header member written by
 * the host layer) the rows "{i}\t{i}\t{count}\t{count/total}\n", i = 0..nbins-1, as BGZF members of exactly
 * 65536 uncompressed bytes (last one partial, no EOF marker -- like the reference).  The frequency is printed as
 * Python prints numpy.float64 (shortest round-trip repr).  kdb_format_frequency exposes that formatter.
 */
int kdb_write_kdb_rows(const char *path, const uint64_t *counts, uint64_t nbins, uint64_t total_kmers,
                       int compresslevel, int nthreads, uint64_t *nblocks_out);
/*
 * The same with the deflate encoder named.  KDB_ENCODER_ROWS (what KDB_ENCODER_DEFAULT means unless the environment says
 * KDB_KDB_ENCODER=zlib): the row-aware encoder -- the text's own structure says where its repeats are (an id's leading
 * digits in the row before, the second id column in the first, a row's "count \t frequency \n" string in the last row with
 * that count), so LZ77 needs no search; one dynamic-Huffman block per member; `compresslevel` is not used.  About ten times
 * zlib level 6 per thread at a slightly better ratio.  KDB_ENCODER_ZLIB: zlib at `compresslevel`, what Bio.bgzf does for the
 * reference.  Either way the decompressed stream and the member boundaries are the reference's; the compressed bytes are not
 * comparable between encoders (nor between zlib versions).
 */
#define KDB_ENCODER_DEFAULT (-1)
#define KDB_ENCODER_ROWS      0
#define KDB_ENCODER_ZLIB      1
int kdb_write_kdb_rows_ex(const char *path, const uint64_t *counts, uint64_t nbins, uint64_t total_kmers,
                          int compresslevel, int nthreads, int encoder, uint64_t *nblocks_out);
int kdb_format_frequency(uint64_t count, uint64_t total, char *buf, size_t cap);
/*
 * kdb_finish's copy-back and kdb_write_kdb_rows_ex in one: the count vector (`folded` = 1: the samplesheet accumulator) comes back into
 * `counts_out` (4^k uint64, caller-owned) in pieces while the writer's threads are already formatting and deflating the rows that have
 * arrived -- the writer needs a chunk's own counts only, never the whole vector first.  Syncs the engine; the caller has the totals
 * (kdb_finish / kdb_finish_folded with counts_out = NULL) and has written the header member(s) to `path` before.  At k = 15 the 0.2 s of
 * the 8 GiB copy-back disappear under the 1.7 s of the rows.
 */
int kdb_copy_back_and_write_kdb_rows(kdb_engine *e, int folded, uint64_t *counts_out, const char *path, uint64_t total_kmers,
                                     int compresslevel, int nthreads, int encoder, uint64_t *nblocks_out);
/*
 * The way back (no GPU work): KDBReader._slurp, kmerdb/fileutil.py:308-466, reads the 4^k rows one by one through Bio.bgzf.  Here the
 * file's BGZF members are inflated in groups on `nthreads` threads and parsed where they were inflated.  Row "x \t kmer_id \t count \t f":
 * kmer_ids_out[x] = kmer_id, counts_out[kmer_id] = count, frequencies_out[kmer_id] = f (the file's column); x must be the row's line number
 * and there must be exactly `nbins` rows of four columns behind the header's delimiter line, else KDB_ERR_ARG.  The arrays hold `nbins`
 * entries each (zero them first: the reference's arrays start as zeros).  KDB_ERR_STATE: the file is not a sequence of BGZF members (one
 * plain gzip stream, say): the host layer reads it through gzip instead.
 */
int kdb_read_kdb_rows(const char *path, uint64_t nbins, uint64_t *kmer_ids_out, uint64_t *counts_out, double *frequencies_out,
                      int nthreads, uint64_t *nrows_out);

/*
 * Per-kernel timing with HIP events on the engine's compute stream (the stream
 * the kernels are launched on).  Enable, run submits, sync, then read back the
 * accumulated device time and launch count of each kernel.
 * kernel ids: see KDB_KERNEL_* ; name via kdb_prof_kernel_name.
 */
#define KDB_KERNEL_MARK          0   /* lens_kernel + hibit_check_kernel + mark_reads_kernel: record geometry, checks, start marks */
#define KDB_KERNEL_COUNT         1   /* count_direct_kernel (global atomics), count_lds_kernel (k <= 7), expand_worklist_kernel */
#define KDB_KERNEL_SCATTER       2   /* scatter_bases_kernel: residues -> pages of bins (k <= 12) or of remainders (level 1, k >= 13) */
#define KDB_KERNEL_SCATTER_L2    3   /* scatter_ids_kernel: level-1 pages -> pages of bins (k >= 13) */
#define KDB_KERNEL_PAGE_SORT     4   /* pages_count / pages_scan / pages_place (+ l2_plan): page tags -> one page list per bucket */
#define KDB_KERNEL_PAGE_HIST     5   /* page_hist_kernel: one 32768-bin LDS histogram per bucket, added to the vector */
#define KDB_KERNEL_STATS         6   /* stats_kernel / fold_kernel: count_nonzero, Sum, samplesheet accumulation */
#define KDB_N_KERNELS            7
int         kdb_prof_enable(kdb_engine *e, int on);
int         kdb_prof_reset(kdb_engine *e);
int         kdb_prof_get(kdb_engine *e, int kernel_id, double *total_ms, uint64_t *launches);
const char *kdb_prof_kernel_name(int kernel_id);

/*
 * Tuning knobs (ints); unknown names return KDB_ERR_ARG.
 *   set: "algo" 0 auto / 1 direct global atomics / 2 LDS-histogram paths (k <= 7 whole vector in LDS, else paged scatter);
 *        "defer_flush" 1/0 (k >= 13: add the scattered batches to the vector together -- at kdb_sync, after 64 batches or
 *        when the page arena is full -- instead of after every batch);  "pending_budget" (bytes the page arena may grow
 *        to; 0 = decide at first use: 85 % of the free device memory, at most 192 GiB);  "reserve_bytes" (device memory the arena
 *        must leave free when it sizes itself -- for what is allocated later beside it: RCCL's buffers at the first collective of each
 *        kind and the scratch of the end-of-job reduce, kmerdb_amd.distributed.reduce_reserve_bytes);  "arena_grow" (the arena starts with
 *        room for eight batches; 0: it stays that size, 1 (default): it doubles once the vector sweeps a larger one would have
 *        saved outweigh the allocation -- fresh device memory costs ~46 ms per GiB --, 2: it doubles whenever it has filled
 *        up: long-lived engines, benchmarks of the steady state);  "arena_batches" (1..64: room for that many batches
 *        like the first one instead of eight -- one allocation instead of a doubling series);  "sc_grid" (persistent
 *        workgroups of the scatter kernels);  "sc_top_bits" 1/0 (k <= 12: buckets from the leading id bits; diagnostic);
 *        "min_len";  "copy_threads", "accum_bytes" (-1 auto: 1 GiB for k >= 13), "stage_bytes", "stage_reads" (host staging);
 *        "one_level_max_k" 13/12 (k = 13 in one scatter level with 1024 rings, or through the two-level path);  "smallk_old" 0/1
 *        (k <= 8 in one CU's LDS, or as before round 4: k <= 7 count_lds_kernel, k = 8 paged scatter);  "overlap" 0/1 (one-level path,
 *        DROP mode: the scatter kernel of batch i + 1 on the compute stream beside the histogram pass of batch i on a second stream;
 *        off by default -- measured slower), "overlap_hist_cus" (> 0: the two streams get disjoint CU masks, that
 *        many CUs for the pass; a diagnostic that needs KDB_ALLOW_CU_MASKS=1 in the environment: on ROCm 7.2 a process that has created a
 *        CU-masked stream crashes or hangs in the runtime when a later hipMalloc runs out of memory), "overlap_mask_mode" (which CUs: 0 the
 *        first, 1 every n-th, 2 the first of every 32);  "sc_wide_lines" / "l1_wide_lines" / "l2_wide_lines" 1/0 (default 1: the scatter kernels of
 *        k <= 12 / level 1 / level 2 write their pages in 128-byte pieces, one workgroup of 1024 threads per CU; 0: 64-byte lines, two
 *        workgroups of 512 -- the form of rounds 2-4, kept for comparison: the memory system takes random 64-byte writes at 3.4-4.6 TB/s
 *        and 128-byte ones at 5.3);  "l1_compiled_k" 1/0 (k = 15: level 1 / level 2 with their shifts compiled in);
 *        "l1_one_round" 1/0 (default 1: level 1 of k <= 15 with 128 rings of 256 elements -- one placement round per tile -- instead of 256 of 128).
 *        The environment variable KDB_ENGINE_OPTS="name=value,..." sets options for every engine a process creates (experiments, the test
 *        suite under an option); an unknown name fails kdb_create.
 *   get: "sc_wide_lines", "l1_wide_lines", "l2_wide_lines", "l1_one_round", "reserve_bytes", "arena_budget_bytes" (what the arena may grow to, once decided), "free_at_sizing" (free device memory when it
 *        was decided), "free_hbm" (free device memory now), "overlap", "overlap_hist_cus", "overlap_scatter_grid",
 *        "algo", "stage_bytes", "stage_reads", "defer_flush", "k", "oom_fallbacks" (batches counted by direct atomics
 *        because scratch did not fit), "pending_batches" (scattered batches not yet added to the vector), "d2h_bytes"
 *        (bytes of count vector copied to the host so far), "folded_files", "sc_lo_bits", "sc_contig_pages", "arena_grow",
 *        "arena_batches", "arena_pages" / "arena_reallocs" (size of the page arena in 1 KiB pages; times it was (re)allocated),
 *        "arena_cursor" / "arena_used_bound" / "arena_worst_case" (pages the pending batches hold: on the device, by the host's
 *        present bound, and by their worst cases added up), "one_level_max_k", "smallk_old",
 *        "hist_flushes" / "flushed_batches" / "full_flushes" (k >= 13: histogram passes over the arena, the batches they added
 *        to the vector, and how many of the passes a full arena forced), "bytes_in" and the
 *        device counters "pages_bases", "lines_bases", "pages_ids", "lines_ids", "table_bytes", "total_kmers" (what the
 *        kernels moved since kdb_reset, by their own count; reading one synchronises the compute stream).
 */
/*
 * Diagnostic (bench.py: `roofline.pattern_ceilings`): what the memory system of the device delivers for the access patterns of the engine's kernels
 * with NO compute -- streamed residues in, random 64-byte lines / 128-byte pieces out, whole pages in (csrc/kdb_probe.hip.h;
 * tools/ubench_hbm_pattern.hip prints the full table).  Runs every pattern (kdb_hbm_pattern_count of them, named by kdb_hbm_pattern_name) for a few
 * milliseconds on two scratch regions of 4 GiB and returns GB/s (read + written) per pattern.  Nothing of the reference corresponds to it.
 */
int         kdb_hbm_pattern_count(void);
const char *kdb_hbm_pattern_name(int i);
int         kdb_hbm_pattern_probe(int device_id, double *gb_per_s_out, int n_out);
int kdb_set_option(kdb_engine *e, const char *name, int64_t value);
int kdb_get_option(kdb_engine *e, const char *name, int64_t *value);

#ifdef __cplusplus
}
#endif
#endif /* KDBHIP_H */
