// kdb_engine.hip -- host side of libkdbhip.so: the C ABI of include/kdbhip.h, the
// engine object (HBM count vector, streams, pinned double-buffered staging) and
// the kernel launches.  gfx950 only; no CPU fallback: every entry point that
// needs the device fails with KDB_ERR_HIP if HIP cannot provide one.
#include <hip/hip_runtime.h>

#include <cstdarg>
#include <cstddef>
#include <cstdio>
#include <cstdlib>
#include <algorithm>
#include <atomic>
#include <cstring>
#include <string>
#include <thread>
#include <vector>

#include "../../include/kdbhip.h"
#include "kdb_kernels.hip.h"
#include "kdb_hist.hip.h"
#include "kdb_scatter.hip.h"
#include "kdb_smallk.hip.h"
#include "kdb_probe.hip.h"
#include "kdb_hostparse.cpp.h"
#include "kdb_kdbwriter.cpp.h"

namespace {

thread_local std::string g_err;

int fail(int code, const char *fmt, ...)
{
    char buf[1024];
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(buf, sizeof buf, fmt, ap);
    va_end(ap);
    g_err = buf;
    return code;
}

#define HIP_TRY(expr)                                                                         \
    do {                                                                                      \
        hipError_t _e = (expr);                                                               \
        if (_e != hipSuccess)                                                                 \
            return fail(KDB_ERR_HIP, "%s failed: %s (%s:%d)", #expr, hipGetErrorString(_e),   \
                        __FILE__, __LINE__);                                                  \
    } while (0)

constexpr int NBUF = 2;                                  // double-buffered staging
const char *const KERNEL_NAMES[KDB_N_KERNELS] = {
    "lens+mark_reads_kernel", "count_kernel", "scatter_bases_kernel", "scatter_ids_kernel", "pages_sort_kernels", "page_hist_kernel", "stats_kernel"};

struct ProfSpan { hipEvent_t a, b; int kernel; };

// pageable -> pinned staging copy, split over a few host threads (one memcpy stream is ~10 GB/s, PCIe Gen5 is ~5x that)
void parallel_copy(uint8_t *dst, const uint8_t *src, size_t n, int nthreads)
{
    const size_t min_piece = 4u << 20;
    int t = (int)std::min<size_t>((size_t)std::max(nthreads, 1), std::max<size_t>(n / min_piece, 1));
    if (t <= 1) { if (n) memcpy(dst, src, n); return; }
    std::vector<std::thread> th;
    th.reserve((size_t)t - 1);
    const size_t piece = (n / (size_t)t + 63) & ~(size_t)63;
    for (int i = 1; i < t; i++) {
        const size_t a = std::min(n, piece * (size_t)i), b = std::min(n, piece * (size_t)(i + 1));
        if (b > a) th.emplace_back([=] { memcpy(dst + a, src + a, b - a); });
    }
    memcpy(dst, src, std::min(n, piece));
    for (auto &x : th) x.join();
}

// First touch of a fresh host buffer that is about to be overwritten whole (a np.empty() the vector is copied back into): the page
// faults of one copying thread cost 0.35 s of the 0.5 s an 8 GiB copy-back took; spread over a few threads they take a tenth.
// Writes one zero per 4 KiB page: only for buffers whose every byte is written afterwards.
void touch_pages(void *buf, size_t n, int nthreads)
{
    if (n < (64u << 20)) return;
    uint8_t *p = (uint8_t *)buf;
    const int t = std::max(1, std::min(nthreads, 64));
    std::vector<std::thread> th;
    const size_t piece = ((n / (size_t)t) + 4095) & ~(size_t)4095;
    for (int i = 0; i < t; i++) {
        const size_t a = std::min(n, piece * (size_t)i), b = std::min(n, piece * (size_t)(i + 1));
        if (b > a) th.emplace_back([=] { for (size_t o = a; o < b; o += 4096) ((volatile uint8_t *)p)[o] = 0; });
    }
    for (auto &x : th) x.join();
}

}  // namespace

struct kdb_engine {
    int k = 0, canonical = 1, n_mode = KDB_N_DROP, device = 0;
    uint64_t nbins = 0;
    unsigned long long *d_table = nullptr;
    bool owns_table = false;
    bool table_escaped = false;      // kdb_table handed the vector's address out: the caller may write it at any time
    kdb::DevCounters *d_ctr = nullptr;
    unsigned long long *d_worklist = nullptr;        // EXPAND mode: windows with > 2 N's, expanded by a workgroup each
    size_t worklist_cap = 1u << 20;
    uint32_t *d_first_rec = nullptr; size_t first_rec_cap = 0;       // record that holds every 4096th byte of the batch in hand (lens_kernel)
    unsigned long long *d_suspects = nullptr;        // DROP mode: positions of a batch's residues that are neither ACGT nor N (resolve_suspects_kernel)
    unsigned long long *sh_suspects = nullptr;       // the same for kdb_shred / kdb_window_ids
    size_t suspects_cap = 1u << 16;
    hipStream_t s_compute = nullptr, s_copy = nullptr;

    // pinned staging (allocated on first kdb_submit)
    size_t stage_bytes = 64ull << 20, stage_reads = 2ull << 20;
    uint8_t *h_bases[NBUF] = {nullptr, nullptr};
    uint64_t *h_offs[NBUF] = {nullptr, nullptr};
    uint8_t *d_bases[NBUF] = {nullptr, nullptr};
    uint64_t *d_offs[NBUF] = {nullptr, nullptr};
    hipEvent_t ev_copied[NBUF] = {nullptr, nullptr}, ev_done[NBUF] = {nullptr, nullptr};
    bool inflight[NBUF] = {false, false};
    hipEvent_t busy[NBUF] = {nullptr, nullptr};     // what must complete before staging buffer b is reused
    int next_buf = 0;
    bool staging_ready = false;
    int copy_threads = 8;

    // device-side accumulation of staged chunks (large k: every batch pays one sweep of the 4^k vector in P2,
    // so 64 MiB batches would be dominated by it; chunks are appended here and counted as one batch)
    int64_t accum_bytes = -1;                        // -1 auto (1 GiB for k >= 13, off below), 0 off
    size_t acc_cap = 0, acc_reads_cap = 0;
    uint8_t *d_acc_bases[2] = {nullptr, nullptr};
    uint64_t *d_acc_offs[2] = {nullptr, nullptr};
    hipEvent_t ev_acc_done[2] = {nullptr, nullptr};
    hipEvent_t ev_acc_copied = nullptr;
    // kdb_submit_pinned: the DMA reads the caller's buffer; call N waits for the copies of call N-2, so a caller that
    // cycles through three buffers (kmerdb_amd.reader) can never overwrite one that is still being read
    hipEvent_t ev_pin[2] = {nullptr, nullptr};
    bool pin_used[2] = {false, false};
    int pin_idx = 0;
    bool acc_inflight[2] = {false, false};
    int acc_slot = 0;
    size_t acc_nb = 0, acc_nr = 0;
    bool acc_ready = false;

    // options
    int64_t algo = 0;                 // 0 auto, 1 direct atomics, 2 LDS-histogram paths
    int min_len = 0;                  // records shorter than this are an error (0 = k)
    int smallk_old = 0;               // 1: k <= 7 through count_lds_kernel and k = 8 through the paged scatter, as before round 4 (for comparison)
    int one_level_max_k = kdb::SC1_K; // largest k counted with one scatter level (13: the 1024-ring kernel; 12: k = 13 takes the two-level path, for comparison)
    kdb::ScatterState sc;             // scratch of the paged-scatter path (8 <= k <= 12)
    // the one-level path with the scatter kernel of batch i + 1 beside the histogram pass of batch i ("overlap" option; DROP mode)
    int overlap = 0;                  // 0 off, 1 on
    int overlap_hist_cus = 0;         // > 0: the pass's stream is masked to that many CUs and the compute stream to the others; 0: no masks
    int overlap_mask_mode = 0;        // which CUs the pass gets: 0 the first ones, 1 every (n / H)-th, 2 the first H / 8 of every 32
    bool compute_masked = false;
    hipStream_t s_hist = nullptr;
    kdb::OverlapState ov;
    kdb::TwoLevelPaged tp;            // its two-level form (k = 13..17): level-1 scratch and the arena of pending level-2 pages
    int64_t oom_fallbacks = 0;        // batches that fell back to direct atomics because scratch did not fit

    // ids-only engines (kdb_create_ids) have no count vector; scratch of kdb_shred / kdb_window_ids (grow-only)
    bool tableless = false;
    uint8_t *sh_seq = nullptr; size_t sh_seq_cap = 0;
    uint64_t *sh_offs = nullptr; size_t sh_offs_cap = 0;
    unsigned long long *sh_ids = nullptr; size_t sh_ids_cap = 0;
    kdb::DevCounters *sh_ctr = nullptr;

    // samplesheet accumulator (kdb_fold_file): counts = counts + counts_ stays on the device
    unsigned long long *d_acc_table = nullptr;
    uint64_t folded_files = 0, folded_total = 0;
    uint64_t d2h_bytes = 0;           // bytes of count vector copied to the host so far (tests assert "one copy at the end")
    uint64_t bytes_in = 0;            // residue bytes handed to the counting kernels so far

    // profiling
    bool prof = false;
    std::vector<ProfSpan> spans;
    std::vector<hipEvent_t> ev_pool;
    double prof_ms[KDB_N_KERNELS] = {0};
    uint64_t prof_n[KDB_N_KERNELS] = {0};
};

namespace {

struct DeviceGuard {
    int prev = -1;
    explicit DeviceGuard(int dev) { if (hipGetDevice(&prev) != hipSuccess) prev = -1; (void)hipSetDevice(dev); }
    ~DeviceGuard() { if (prev >= 0) (void)hipSetDevice(prev); }
};

hipEvent_t prof_event(kdb_engine *e)
{
    hipEvent_t ev = nullptr;
    if (!e->ev_pool.empty()) { ev = e->ev_pool.back(); e->ev_pool.pop_back(); return ev; }
    if (hipEventCreate(&ev) != hipSuccess) return nullptr;
    return ev;
}

struct ProfScope {
    kdb_engine *e; int kernel; hipStream_t stream; hipEvent_t a = nullptr, b = nullptr;
    ProfScope(kdb_engine *e_, int kernel_, hipStream_t stream_ = nullptr) : e(e_), kernel(kernel_), stream(stream_ ? stream_ : e_->s_compute)
    {
        if (!e->prof) return;
        a = prof_event(e); b = prof_event(e);
        if (a) (void)hipEventRecord(a, stream);
    }
    ~ProfScope()
    {
        if (!e->prof || !a || !b) return;
        (void)hipEventRecord(b, stream);
        e->spans.push_back({a, b, kernel});
    }
};

struct EngineProf : kdb::ProfHook {
    kdb_engine *e; ProfScope *cur = nullptr;
    explicit EngineProf(kdb_engine *e_) : e(e_) {}
    void begin(int kernel) override { cur = new ProfScope(e, kernel); }
    void begin_on(int kernel, hipStream_t st) override { cur = new ProfScope(e, kernel, st); }
    void end() override { delete cur; cur = nullptr; }
    ~EngineProf() override { delete cur; }
};

int prof_collect(kdb_engine *e)
{
    for (auto &s : e->spans) {
        HIP_TRY(hipEventSynchronize(s.b));
        float ms = 0.f;
        HIP_TRY(hipEventElapsedTime(&ms, s.a, s.b));
        e->prof_ms[s.kernel] += ms;
        e->prof_n[s.kernel] += 1;
        e->ev_pool.push_back(s.a);
        e->ev_pool.push_back(s.b);
    }
    e->spans.clear();
    return KDB_OK;
}

int ensure_staging(kdb_engine *e)
{
    if (e->staging_ready) return KDB_OK;
    for (int b = 0; b < NBUF; b++) {
        HIP_TRY(hipHostMalloc((void **)&e->h_bases[b], e->stage_bytes + 64, hipHostMallocDefault));
        HIP_TRY(hipHostMalloc((void **)&e->h_offs[b], (e->stage_reads + 1) * sizeof(uint64_t), hipHostMallocDefault));
        HIP_TRY(hipMalloc((void **)&e->d_bases[b], e->stage_bytes + 64));
        HIP_TRY(hipMalloc((void **)&e->d_offs[b], (e->stage_reads + 1) * sizeof(uint64_t)));
        HIP_TRY(hipEventCreateWithFlags(&e->ev_copied[b], hipEventDisableTiming));
        HIP_TRY(hipEventCreateWithFlags(&e->ev_done[b], hipEventDisableTiming));
    }
    e->staging_ready = true;
    // (from k = 13 on every batch pays a sweep of the 4^k vector -- 0.5 GiB at k = 13 -- or, on the two-level path, partial pages of
    //  512 rings x 512 workgroups: 64 MiB chunks are appended on the device and counted 1 GiB at a time)
    const int64_t want = e->accum_bytes >= 0 ? e->accum_bytes : (e->k >= 13 ? (int64_t)1 << 30 : 0);
    if (want > 0) {
        e->acc_cap = (size_t)want < e->stage_bytes ? e->stage_bytes : (size_t)want;
        e->acc_reads_cap = e->acc_cap / 64 + e->stage_reads;
        for (int s = 0; s < 2; s++) {
            HIP_TRY(hipMalloc((void **)&e->d_acc_bases[s], e->acc_cap + 64));
            HIP_TRY(hipMalloc((void **)&e->d_acc_offs[s], (e->acc_reads_cap + 1) * sizeof(uint64_t)));
            HIP_TRY(hipEventCreateWithFlags(&e->ev_acc_done[s], hipEventDisableTiming));
        }
        HIP_TRY(hipEventCreateWithFlags(&e->ev_acc_copied, hipEventDisableTiming));
        e->acc_ready = true;
    }
    return KDB_OK;
}

// a histogram pass of the overlapped one-level path may still be running on s_hist: whatever else is about to touch the vector on
// the compute stream waits for it
int overlap_join(kdb_engine *e)
{
    if (e->ov.last < 0) return KDB_OK;
    HIP_TRY(hipStreamWaitEvent(e->s_compute, e->ov.hist_done[e->ov.last], 0));
    e->ov.last = -1;
    return KDB_OK;
}

// (re)create the compute stream and the histogram stream for the "overlap" options; nothing may be in flight
int overlap_streams(kdb_engine *e)
{
    const bool want_hist = e->overlap != 0, want_mask = want_hist && e->overlap_hist_cus > 0;
    if (e->s_hist) { HIP_TRY(hipStreamSynchronize(e->s_hist)); HIP_TRY(hipStreamDestroy(e->s_hist)); e->s_hist = nullptr; }
    kdb::overlap_free(e->ov);
    if (e->compute_masked || want_mask) {
        HIP_TRY(hipStreamSynchronize(e->s_compute));
        HIP_TRY(hipStreamDestroy(e->s_compute));
        e->s_compute = nullptr;
        e->compute_masked = false;
    }
    if (!want_hist) {
        if (!e->s_compute) HIP_TRY(hipStreamCreateWithFlags(&e->s_compute, hipStreamNonBlocking));
        return KDB_OK;
    }
    if (!want_mask) {
        if (!e->s_compute) HIP_TRY(hipStreamCreateWithFlags(&e->s_compute, hipStreamNonBlocking));
        HIP_TRY(hipStreamCreateWithFlags(&e->s_hist, hipStreamNonBlocking));
        e->ov.grid = 0;
        return KDB_OK;
    }
    // Diagnostic only.  On ROCm 7.2 a process that has created a CU-masked stream crashes or hangs inside the runtime when a LATER hipMalloc
    // runs out of memory (tools/experiments/repro_r05_oom_after_cumask.py: torch filling the device after such an engine was closed) --
    // and the partition does not pay anyway.  Masks are therefore only made when the environment asks for them.
    if (!getenv("KDB_ALLOW_CU_MASKS")) {
        e->overlap_hist_cus = 0;
        (void)overlap_streams(e);
        return fail(KDB_ERR_ARG, "overlap_hist_cus: CU-masked streams are a diagnostic (set KDB_ALLOW_CU_MASKS=1; see include/kdbhip.h)");
    }
    hipDeviceProp_t prop;
    HIP_TRY(hipGetDeviceProperties(&prop, e->device));
    const int ncu = prop.multiProcessorCount, H = e->overlap_hist_cus;
    if (H >= ncu || ncu > 1024) return fail(KDB_ERR_ARG, "overlap_hist_cus=%d of %d CUs", H, ncu);
    std::vector<uint32_t> mh((size_t)(ncu + 31) / 32, 0u), ms((size_t)(ncu + 31) / 32, 0u);
    int given = 0;
    for (int i = 0; i < ncu; i++) {
        bool hist;
        if (e->overlap_mask_mode == 1) hist = given < H && (int)(((long long)i * H) / ncu) != (int)(((long long)(i + 1) * H) / ncu);
        else if (e->overlap_mask_mode == 2) hist = given < H && (i % 32) < (H * 32 + ncu - 1) / ncu;
        else hist = i < H;
        if (hist) { mh[(size_t)i / 32] |= 1u << (i % 32); given++; } else ms[(size_t)i / 32] |= 1u << (i % 32);
    }
    HIP_TRY(hipExtStreamCreateWithCUMask(&e->s_compute, (uint32_t)ms.size(), ms.data()));
    e->compute_masked = true;
    HIP_TRY(hipExtStreamCreateWithCUMask(&e->s_hist, (uint32_t)mh.size(), mh.data()));
    e->ov.grid = 2u * (uint32_t)(ncu - given);                 // two persistent scatter workgroups per CU of the compute stream
    return KDB_OK;
}

enum : int { BATCH_HOST_FED = 1, BATCH_CONST_INPUT = 2 };
int grow(void **p, size_t *cap, size_t need_bytes);
int launch_batch(kdb_engine *e, uint8_t *d_bases, size_t nbytes, const uint64_t *d_offs, size_t nreads, int first_is_continuation, int flags);

// count what has been accumulated so far as one batch
int flush_accumulated(kdb_engine *e)
{
    if (!e->acc_ready || e->acc_nr == 0) return KDB_OK;
    const int s = e->acc_slot;
    HIP_TRY(hipEventRecord(e->ev_acc_copied, e->s_copy));
    HIP_TRY(hipStreamWaitEvent(e->s_compute, e->ev_acc_copied, 0));
    int rc = launch_batch(e, e->d_acc_bases[s], e->acc_nb, e->d_acc_offs[s], e->acc_nr, 0, BATCH_HOST_FED);
    if (rc != KDB_OK) return rc;
    HIP_TRY(hipEventRecord(e->ev_acc_done[s], e->s_compute));
    e->acc_inflight[s] = true;
    e->acc_slot = s ^ 1;
    e->acc_nb = e->acc_nr = 0;
    return KDB_OK;
}

// launch the counting kernels over one device-resident batch, on s_compute
int launch_batch(kdb_engine *e, uint8_t *d_bases, size_t nbytes, const uint64_t *d_offs, size_t nreads, int first_is_continuation, int flags)
{
    if (nreads == 0) return KDB_OK;
    if (e->tableless) return fail(KDB_ERR_STATE, "this engine was created by kdb_create_ids: it has no count vector");
    const uint64_t ntiles = (nbytes + kdb::TILE_BYTES - 1) / kdb::TILE_BYTES;
    if (ntiles > 0x7FFFFFFFull) return fail(KDB_ERR_ARG, "batch too large: %zu bytes", nbytes);       // (before anything is written into the caller's buffer)
    // The LDS-histogram paths take a ragged batch's record starts from the offsets (lens_kernel fills first_rec; nothing is written
    // into the residues).  Only the direct-atomics kernel (algo 1, or the fallback when the scatter scratch does not fit) and the
    // old k <= 7 kernel still mark record starts as bit 7 of a record's first byte: the marks go on right before such a kernel and
    // come off again right after it -- also when this function gives up half way.
    bool marked = false;
    auto mark = [&]() {
        ProfScope ps(e, KDB_KERNEL_MARK);
        const unsigned mg = (unsigned)std::min<uint64_t>((nreads + 255) / 256, 4096);
        if (flags & BATCH_CONST_INPUT)          // the caller's buffer is never written: these kernels then need records of one length
            hipLaunchKernelGGL(kdb::require_uniform_kernel, dim3(1), dim3(1), 0, e->s_compute, e->d_ctr);
        else {
            hipLaunchKernelGGL(kdb::mark_reads_kernel, dim3(mg), dim3(256), 0, e->s_compute, d_bases, d_offs, (uint64_t)nreads,
                               first_is_continuation, e->d_ctr);          // (grid-stride: a uniform batch returns at once)
            marked = true;
        }
    };
    auto unmark = [&]() {
        if (!marked) return;
        ProfScope ps(e, KDB_KERNEL_MARK);
        const unsigned ug = (unsigned)std::min<uint64_t>((nreads + 255) / 256, 4096);
        hipLaunchKernelGGL(kdb::unmark_reads_kernel, dim3(ug), dim3(256), 0, e->s_compute, d_bases, d_offs, (uint64_t)nreads, first_is_continuation,
                           (const kdb::DevCounters *)e->d_ctr);
        marked = false;
    };
    bool no_first_rec = false;
    {
        // first_rec: one entry per 4 KiB of residues (grow-only scratch).  The batch before this one may still be reading the old array
        // (submits of device-resident input are asynchronous): drain the stream before it is freed.  No room for a larger one: the
        // direct-atomics kernel needs none, the batch is counted there (as when the scatter scratch does not fit).
        const size_t need = ((nbytes >> kdb::FIRST_REC_SHIFT) + 2) * sizeof(uint32_t);
        if (e->first_rec_cap < need && e->d_first_rec) HIP_TRY(hipStreamSynchronize(e->s_compute));
        int grc = grow((void **)&e->d_first_rec, &e->first_rec_cap, need);
        if (grc == KDB_ERR_NOMEM) no_first_rec = true;
        else if (grc != KDB_OK) return grc;
    }
    {
        ProfScope ps(e, KDB_KERNEL_MARK);
        const dim3 grid((unsigned)((nreads + 255) / 256)), block(256);
        // only the per-batch words: n_short, n_bad, bad_layout, not_uniform stay set until kdb_reset
        HIP_TRY(hipMemsetAsync(&e->d_ctr->neg_min_len, 0, kdb::PER_BATCH_WORDS * sizeof(unsigned long long), e->s_compute));
        const dim3 lgrid(grid.x < 1024u ? grid.x : 1024u);
        hipLaunchKernelGGL(kdb::lens_kernel, lgrid, block, 0, e->s_compute, d_offs, (uint64_t)nreads, (uint64_t)nbytes,
                           e->min_len > 0 ? e->min_len : e->k, first_is_continuation, e->d_ctr, e->d_first_rec);
        // bytes with bit 7 set are no residues (kmer.py:170 raises).  No pass of its own looks for them: the counting kernels'
        // front end counts them as bad; where the direct-atomics kernel runs over a ragged batch (its start marks are bit 7),
        // mark_reads_kernel reports a record start that carries the bit already, and any other such byte is a mark too many
        // (marks_seen != marks_set at the sync).  Host-fed and device-resident input alike.
    }
    if (nbytes == 0) return KDB_OK;          // only zero-length records: all short reads
    if (nreads > 0xFFFFFFF0ull) return fail(KDB_ERR_ARG, "batch of %zu records: at most 2^32 - 16 per submit", nreads);
    e->bytes_in += nbytes;
    const kdb::RecStarts rs{d_offs, e->d_first_rec, (uint32_t)nreads, first_is_continuation ? 1u : 0u};
    int algo = (int)e->algo;
    if (algo == 0 || algo == 3) algo = 2;                    // LDS-histogram paths unless told otherwise (3: the paged scatter's old number)
    if (no_first_rec && algo == 2) { algo = 1; e->oom_fallbacks++; e->tp.table_is_zero = false; }
    const bool paged2 = algo == 2 && e->k > e->one_level_max_k;
    // only the deferred two-level flush may treat the vector as still all zero; everything else adds to it right away
    if (!paged2 || e->n_mode == KDB_N_EXPAND || !e->tp.defer) e->tp.table_is_zero = false;
    if (algo == 2) {
        EngineProf hook(e);
        const bool ex = e->n_mode == KDB_N_EXPAND;
        int rc;
        if (e->k <= kdb::SMALLK_LDS_MAX_K && !e->smallk_old)
            rc = kdb::smallk_lds_count(e->s_compute, d_bases, nbytes, rs, e->k, e->canonical, ex, e->sc.grid, e->d_table, e->d_ctr, hook);
        else if (e->k <= kdb::SMALLK_MAX) {
            mark();
            rc = kdb::smallk_count(e->s_compute, d_bases, nbytes, e->k, e->canonical, ex, e->d_table, e->d_ctr, hook);
            unmark();
        }
        else if (e->k <= e->one_level_max_k) {
            rc = 3;
            if (e->overlap && e->s_hist && !ex) {
                e->ov.sc[0].grid = e->sc.grid; e->ov.sc[0].lo_bits = e->sc.lo_bits; e->ov.sc[0].contig_pages = e->sc.contig_pages; e->ov.sc[0].wide_lines = e->sc.wide_lines;
                rc = kdb::scatter_count_overlapped(e->ov, e->s_compute, e->s_hist, d_bases, nbytes, rs, e->k, e->canonical, e->d_table, e->d_ctr, hook);
            }
            if (rc == 3) {
                { const int jrc = overlap_join(e); if (jrc != KDB_OK) return jrc; }
                rc = kdb::scatter_count(e->sc, e->s_compute, d_bases, nbytes, rs, e->k, e->canonical, ex, e->d_table, e->d_ctr, hook);
            }
        }
        else {
            const size_t lost = nreads * (size_t)(e->k - 1);
            rc = kdb::twolevel_paged_count(e->tp, e->s_compute, d_bases, nbytes, rs, nbytes > lost ? nbytes - lost : 0, e->k, e->canonical, ex, e->d_table, e->d_ctr, hook);
        }
        if (rc == 2) { e->oom_fallbacks++; algo = 1; e->tp.table_is_zero = false; }   // no room for the scatter scratch: count this batch with direct atomics
        else if (rc != 0) return fail(KDB_ERR_HIP, "LDS-histogram path failed: %s", kdb::partition_error());
    }
    if (algo == 1) {
        { const int jrc = overlap_join(e); if (jrc != KDB_OK) return jrc; }
        mark();
        {
            ProfScope ps(e, KDB_KERNEL_COUNT);
            const bool ex = (e->n_mode == KDB_N_EXPAND);
            const dim3 grid((unsigned)ntiles), block(kdb::TPB);
#define KDB_LAUNCH_DIRECT(ID, EX)                                                                              \
    hipLaunchKernelGGL((kdb::count_direct_kernel<ID, EX>), grid, block, 0, e->s_compute, d_bases, (uint64_t)nbytes, \
                       e->k, e->canonical, e->d_table, e->d_ctr)
            if (e->k <= 16) { if (ex) KDB_LAUNCH_DIRECT(uint32_t, true); else KDB_LAUNCH_DIRECT(uint32_t, false); }
            else            { if (ex) KDB_LAUNCH_DIRECT(uint64_t, true); else KDB_LAUNCH_DIRECT(uint64_t, false); }
#undef KDB_LAUNCH_DIRECT
        }
        unmark();
    }
    if (e->n_mode == KDB_N_EXPAND && e->d_worklist) {
        ProfScope ps(e, KDB_KERNEL_COUNT);
        hipLaunchKernelGGL(kdb::expand_worklist_kernel, dim3(1024), dim3(256), 0, e->s_compute, e->d_table, e->d_ctr, e->k, e->canonical);
    }
    if (e->n_mode == KDB_N_DROP) {
        // residues that are neither ACGT nor N: an IUPAC code is only an error in a window that no N shields (kmer.py:287-289)
        ProfScope ps(e, KDB_KERNEL_MARK);
        hipLaunchKernelGGL(kdb::resolve_suspects_kernel, dim3(4), dim3(256), 0, e->s_compute, (const uint8_t *)d_bases, (uint64_t)nbytes, d_offs, (uint64_t)nreads,
                           e->k, e->d_ctr);
    }
    HIP_TRY(hipGetLastError());
    return KDB_OK;
}

// batches the two-level path has scattered into the page arena but not yet added to the vector
int flush_pending_paged(kdb_engine *e)
{
    if (e->tp.pending == 0) return KDB_OK;
    EngineProf hook(e);
    if (kdb::twolevel_paged_flush(e->tp, e->s_compute, e->d_table, e->d_ctr, hook)) return fail(KDB_ERR_HIP, "LDS-histogram path failed: %s", kdb::partition_error());
    return KDB_OK;
}

int check_errors(kdb_engine *e)
{
    kdb::DevCounters c;
    HIP_TRY(hipMemcpy(&c, e->d_ctr, sizeof c, hipMemcpyDeviceToHost));
    if (c.n_short)
        return fail(KDB_ERR_SHORT_READ, "%llu record(s) shorter than k=%d (reference: kmer.py:461-463 raises)",
                    c.n_short, e->k);
    if (c.bad_layout)
        return fail(KDB_ERR_ARG, "read_offsets must rise from 0 to nbytes (records tile the residue buffer exactly)");
    if (c.internal_err)
        return fail(KDB_ERR_STATE, "internal: a scatter kernel ran out of its page sequence (%llu times); counts are incomplete", c.internal_err);
    if (c.not_uniform)
        return fail(KDB_ERR_ARG, "kdb_submit_device_const needs records of one length (the buffer is never marked); use kdb_submit_device");
    if (c.n_bad)
        return fail(KDB_ERR_BAD_RESIDUE, "%llu residue(s) outside ACGTN%s (reference: kmer.py:309 / :170 raises)", c.n_bad,
                    e->n_mode == KDB_N_DROP ? " that no N in their window shields" : "");
    if (c.marks_seen != c.marks_set)
        return fail(KDB_ERR_BAD_RESIDUE, "the residue buffer holds %lld byte(s) with bit 7 set that are not record starts of their batch "
                                         "(not residues -- kmer.py:170 raises -- or marks left in a device buffer by a job that was aborted)",
                    (long long)(c.marks_seen - c.marks_set));
    return KDB_OK;
}

int grow(void **p, size_t *cap, size_t need_bytes)
{
    if (*cap >= need_bytes) return KDB_OK;
    if (*p) { (void)hipFree(*p); *p = nullptr; *cap = 0; }
    size_t want = need_bytes + need_bytes / 4 + 256;
    hipError_t err = hipMalloc(p, want);
    if (err != hipSuccess) { (void)hipGetLastError(); return fail(KDB_ERR_NOMEM, "hipMalloc(%zu) failed: %s", want, hipGetErrorString(err)); }
    *cap = want;
    return KDB_OK;
}

// scratch of kdb_shred / kdb_window_ids: kept across calls (three hipMalloc + hipFree per record otherwise)
int shred_scratch(kdb_engine *e, size_t nbytes, size_t nreads)
{
    int rc;
    if ((rc = grow((void **)&e->sh_seq, &e->sh_seq_cap, nbytes + 64)) != KDB_OK) return rc;
    if ((rc = grow((void **)&e->sh_ids, &e->sh_ids_cap, nbytes * 8ull)) != KDB_OK) return rc;
    if (nreads && (rc = grow((void **)&e->sh_offs, &e->sh_offs_cap, (nreads + 1) * sizeof(uint64_t))) != KDB_OK) return rc;
    if (!e->sh_ctr) HIP_TRY(hipMalloc((void **)&e->sh_ctr, sizeof(kdb::DevCounters)));
    if (!e->sh_suspects) HIP_TRY(hipMalloc((void **)&e->sh_suspects, e->suspects_cap * sizeof(unsigned long long)));
    return KDB_OK;
}

int create_common(kdb_engine *e, kdb_engine **out)
{
    hipError_t err;
    if ((err = hipStreamCreateWithFlags(&e->s_compute, hipStreamNonBlocking)) != hipSuccess ||
        (err = hipStreamCreateWithFlags(&e->s_copy, hipStreamNonBlocking)) != hipSuccess ||
        (err = hipMalloc((void **)&e->d_ctr, sizeof(kdb::DevCounters))) != hipSuccess) {
        kdb_destroy(e);
        return fail(KDB_ERR_HIP, "engine setup failed: %s", hipGetErrorString(err));
    }
    int rc = kdb_reset(e);
    if (rc != KDB_OK) { kdb_destroy(e); return rc; }
    // KDB_ENGINE_OPTS="name=value,name=value": tuning options for every engine of the process (experiments and the test suite under an
    // option: tools/experiments/); an unknown name or a refused value fails the creation -- loudly, like kdb_set_option itself
    if (const char *opts = getenv("KDB_ENGINE_OPTS")) {
        std::string all(opts);
        for (size_t at = 0; at < all.size();) {
            size_t end = all.find(',', at);
            if (end == std::string::npos) end = all.size();
            const std::string kv = all.substr(at, end - at);
            at = end + 1;
            if (kv.empty()) continue;
            const size_t eq = kv.find('=');
            if (eq == std::string::npos) { kdb_destroy(e); return fail(KDB_ERR_ARG, "KDB_ENGINE_OPTS: '%s' is not name=value", kv.c_str()); }
            rc = kdb_set_option(e, kv.substr(0, eq).c_str(), (int64_t)strtoll(kv.c_str() + eq + 1, nullptr, 0));
            if (rc != KDB_OK) { kdb_destroy(e); return rc; }
        }
    }
    *out = e;
    return KDB_OK;
}

}  // namespace

// ---- the memory system's ceilings for the kernels' access patterns (diagnostic; bench.py) ----
namespace {
struct ProbePattern { const char *name; void (*launch)(const uint8_t *, uint64_t, uint8_t *, uint64_t, uint32_t, uint32_t *, hipStream_t); uint32_t read_bytes, write_bytes; };
template <int RM, int WM, int NR, int NW>
void probe_launch(const uint8_t *src, uint64_t sb, uint8_t *dst, uint64_t db, uint32_t steps, uint32_t *sink, hipStream_t st)
{
    hipLaunchKernelGGL((kdbprobe::pattern<RM, WM, NR, NW>), dim3(512), dim3(512), 0, st, src, sb, dst, db, steps, sink);
}
using namespace kdbprobe;
#define KDB_PROBE(name, RM, WM, NR, NW) {name, probe_launch<RM, WM, NR, NW>, (uint32_t)(NR * (RM == R_PAGES15 ? 1536 : RM ? 1024 : 0)), (uint32_t)(NW * (WM ? 1024 : 0))}
const ProbePattern PROBES[] = {
    KDB_PROBE("stream_read", R_STREAM, W_NONE, 4, 0),
    KDB_PROBE("stream_write", R_NONE, W_STREAM, 0, 4),
    KDB_PROBE("pages_1k_read", R_PAGES, W_NONE, 4, 0),                         // the histogram pass: whole 1 KiB pages, four in flight per wave
    KDB_PROBE("lines_64_write", R_NONE, W_LINES_SC1, 0, 2),                     // what rounds 2-4 wrote
    KDB_PROBE("pieces_128_write", R_NONE, W_CHUNK128_SC1, 0, 2),                // what round 5 writes
    KDB_PROBE("scatter_64", R_STREAM, W_LINES_SC1, 1, 2),                       // one-level scatter: 1 KiB of residues in per 2 KiB of elements out
    KDB_PROBE("scatter_128", R_STREAM, W_CHUNK128_SC1, 1, 2),
    KDB_PROBE("level1_64", R_STREAM, W_LINES_SC1, 1, 3),                        // level 1: 3 bytes out per residue
    KDB_PROBE("level1_128", R_STREAM, W_CHUNK128_SC1, 1, 3),
    KDB_PROBE("level2_64", R_PAGES15, W_LINES_SC1, 2, 2),                       // level 2: two 1.5 KiB pages in per 2 KiB out
    KDB_PROBE("level2_128", R_PAGES15, W_CHUNK128_SC1, 2, 2),
};
#undef KDB_PROBE
constexpr int N_PROBES = (int)(sizeof(PROBES) / sizeof(PROBES[0]));
}  // namespace

extern "C" {

int kdb_abi_version(void) { return KDB_ABI_VERSION; }

const char *kdb_last_error(void) { return g_err.c_str(); }

int kdb_device_count(int *n_out)
{
    if (!n_out) return fail(KDB_ERR_ARG, "n_out is NULL");
    HIP_TRY(hipGetDeviceCount(n_out));
    return KDB_OK;
}

const char *kdb_prof_kernel_name(int kernel_id)
{
    if (kernel_id < 0 || kernel_id >= KDB_N_KERNELS) return "";
    return KERNEL_NAMES[kernel_id];
}

int kdb_create(int k, int canonicalize, int n_mode, int device_id, void *d_table, kdb_engine **out)
{
    if (!out) return fail(KDB_ERR_ARG, "out is NULL");
    *out = nullptr;
    if (k < 1 || k > 17) return fail(KDB_ERR_ARG, "k=%d outside 1..17 (4^k uint64 table must fit in HBM)", k);
    if (n_mode != KDB_N_DROP && n_mode != KDB_N_EXPAND) return fail(KDB_ERR_ARG, "n_mode=%d", n_mode);
    int ndev = 0;
    HIP_TRY(hipGetDeviceCount(&ndev));
    if (device_id < 0 || device_id >= ndev) return fail(KDB_ERR_ARG, "device_id=%d but %d device(s) visible", device_id, ndev);
    DeviceGuard g(device_id);
    kdb_engine *e = new kdb_engine();
    e->k = k; e->canonical = canonicalize ? 1 : 0; e->n_mode = n_mode; e->device = device_id;
    e->nbins = 1ull << (2 * k);
    if (d_table) {
        e->d_table = (unsigned long long *)d_table;
        e->owns_table = false;
    } else {
        size_t free_b = 0, total_b = 0;
        (void)hipMemGetInfo(&free_b, &total_b);
        const unsigned long long need = e->nbins * 8ull;
        if (free_b && need > free_b) {
            delete e;
            return fail(KDB_ERR_NOMEM, "4^%d uint64 table needs %llu bytes, device has %zu free", k, need, free_b);
        }
        hipError_t me = hipMalloc((void **)&e->d_table, need);
        if (me != hipSuccess) { delete e; return fail(KDB_ERR_NOMEM, "hipMalloc(%llu) failed: %s", need, hipGetErrorString(me)); }
        e->owns_table = true;
    }
    return create_common(e, out);
}

int kdb_create_ids(int k, int canonicalize, int device_id, kdb_engine **out)
{
    if (!out) return fail(KDB_ERR_ARG, "out is NULL");
    *out = nullptr;
    if (k < 1 || k > 17) return fail(KDB_ERR_ARG, "k=%d outside 1..17", k);
    int ndev = 0;
    HIP_TRY(hipGetDeviceCount(&ndev));
    if (device_id < 0 || device_id >= ndev) return fail(KDB_ERR_ARG, "device_id=%d but %d device(s) visible", device_id, ndev);
    DeviceGuard g(device_id);
    kdb_engine *e = new kdb_engine();
    e->k = k; e->canonical = canonicalize ? 1 : 0; e->n_mode = KDB_N_DROP; e->device = device_id;
    e->nbins = 0; e->tableless = true;
    return create_common(e, out);
}

int kdb_destroy(kdb_engine *e)
{
    if (!e) return KDB_OK;
    DeviceGuard g(e->device);
    if (e->s_compute) (void)hipStreamSynchronize(e->s_compute);
    if (e->s_copy) (void)hipStreamSynchronize(e->s_copy);
    if (e->s_hist) { (void)hipStreamSynchronize(e->s_hist); (void)hipStreamDestroy(e->s_hist); }
    kdb::overlap_free(e->ov);
    kdb::scatter_free(e->sc);
    kdb::twolevel_paged_free(e->tp);
    for (auto &s : e->spans) { (void)hipEventDestroy(s.a); (void)hipEventDestroy(s.b); }
    for (auto ev : e->ev_pool) (void)hipEventDestroy(ev);
    for (int b = 0; b < NBUF; b++) {
        if (e->h_bases[b]) (void)hipHostFree(e->h_bases[b]);
        if (e->h_offs[b]) (void)hipHostFree(e->h_offs[b]);
        if (e->d_bases[b]) (void)hipFree(e->d_bases[b]);
        if (e->d_offs[b]) (void)hipFree(e->d_offs[b]);
        if (e->ev_copied[b]) (void)hipEventDestroy(e->ev_copied[b]);
        if (e->ev_done[b]) (void)hipEventDestroy(e->ev_done[b]);
    }
    for (int s2 = 0; s2 < 2; s2++) {
        if (e->d_acc_bases[s2]) (void)hipFree(e->d_acc_bases[s2]);
        if (e->d_acc_offs[s2]) (void)hipFree(e->d_acc_offs[s2]);
        if (e->ev_acc_done[s2]) (void)hipEventDestroy(e->ev_acc_done[s2]);
    }
    if (e->ev_acc_copied) (void)hipEventDestroy(e->ev_acc_copied);
    for (int s2 = 0; s2 < 2; s2++) if (e->ev_pin[s2]) (void)hipEventDestroy(e->ev_pin[s2]);
    if (e->d_worklist) (void)hipFree(e->d_worklist);
    if (e->d_suspects) (void)hipFree(e->d_suspects);
    if (e->d_first_rec) (void)hipFree(e->d_first_rec);
    if (e->sh_suspects) (void)hipFree(e->sh_suspects);
    if (e->sh_seq) (void)hipFree(e->sh_seq);
    if (e->sh_offs) (void)hipFree(e->sh_offs);
    if (e->sh_ids) (void)hipFree(e->sh_ids);
    if (e->sh_ctr) (void)hipFree(e->sh_ctr);
    if (e->d_acc_table) (void)hipFree(e->d_acc_table);
    if (e->d_ctr) (void)hipFree(e->d_ctr);
    if (e->owns_table && e->d_table) (void)hipFree(e->d_table);
    if (e->s_compute) (void)hipStreamDestroy(e->s_compute);
    if (e->s_copy) (void)hipStreamDestroy(e->s_copy);
    delete e;
    return KDB_OK;
}

int kdb_reset(kdb_engine *e)
{
    if (!e) return fail(KDB_ERR_ARG, "engine is NULL");
    DeviceGuard g(e->device);
    e->acc_nb = e->acc_nr = 0;                       // anything not yet counted is dropped with the vector
    kdb::twolevel_paged_drop(e->tp);
    HIP_TRY(hipStreamSynchronize(e->s_copy));
    HIP_TRY(hipStreamSynchronize(e->s_compute));
    if (e->s_hist) { HIP_TRY(hipStreamSynchronize(e->s_hist)); e->ov.last = -1; }
    if (e->nbins) HIP_TRY(hipMemsetAsync(e->d_table, 0, e->nbins * 8ull, e->s_compute));
    if (e->d_acc_table) HIP_TRY(hipMemsetAsync(e->d_acc_table, 0, e->nbins * 8ull, e->s_compute));
    e->folded_files = e->folded_total = 0;
    e->tp.table_is_zero = e->owns_table && !e->table_escaped;             // (a caller-owned vector may be written by the caller at any time)
    HIP_TRY(hipMemsetAsync(e->d_ctr, 0, sizeof(kdb::DevCounters), e->s_compute));
    unsigned long long wl[2] = {0, 0}, sus[2] = {0, 0};
    if (e->n_mode == KDB_N_EXPAND) {
        if (!e->d_worklist) HIP_TRY(hipMalloc((void **)&e->d_worklist, e->worklist_cap * sizeof(unsigned long long)));
        wl[0] = (unsigned long long)(uintptr_t)e->d_worklist; wl[1] = (unsigned long long)e->worklist_cap;
        HIP_TRY(hipMemcpyAsync(&e->d_ctr->wl, wl, sizeof wl, hipMemcpyHostToDevice, e->s_compute));
    } else if (!e->tableless) {
        if (!e->d_suspects) HIP_TRY(hipMalloc((void **)&e->d_suspects, e->suspects_cap * sizeof(unsigned long long)));
        sus[0] = (unsigned long long)(uintptr_t)e->d_suspects; sus[1] = (unsigned long long)e->suspects_cap;
        HIP_TRY(hipMemcpyAsync(&e->d_ctr->sus, sus, sizeof sus, hipMemcpyHostToDevice, e->s_compute));
    }
    HIP_TRY(hipStreamSynchronize(e->s_compute));
    return KDB_OK;
}

int kdb_submit_device(kdb_engine *e, void *d_bases, size_t nbytes, const void *d_read_offsets, size_t nreads)
{
    if (!e) return fail(KDB_ERR_ARG, "engine is NULL");
    if (nreads && (!d_bases || !d_read_offsets)) return fail(KDB_ERR_ARG, "NULL device buffer");
    if (((uintptr_t)d_bases & 15u) != 0) return fail(KDB_ERR_ARG, "d_bases must be 16-byte aligned");
    DeviceGuard g(e->device);
    return launch_batch(e, (uint8_t *)d_bases, nbytes, (const uint64_t *)d_read_offsets, nreads, 0, 0);
}

int kdb_submit_device_const(kdb_engine *e, const void *d_bases, size_t nbytes, const void *d_read_offsets, size_t nreads)
{
    if (!e) return fail(KDB_ERR_ARG, "engine is NULL");
    if (nreads && (!d_bases || !d_read_offsets)) return fail(KDB_ERR_ARG, "NULL device buffer");
    if (((uintptr_t)d_bases & 15u) != 0) return fail(KDB_ERR_ARG, "d_bases must be 16-byte aligned");
    DeviceGuard g(e->device);
    return launch_batch(e, (uint8_t *)const_cast<void *>(d_bases), nbytes, (const uint64_t *)d_read_offsets, nreads, 0, BATCH_CONST_INPUT);
}

static int submit_impl(kdb_engine *e, const uint8_t *bases, size_t nbytes, const uint64_t *offs, size_t nreads, bool src_pinned, bool first_continues = false)
{
    if (!e) return fail(KDB_ERR_ARG, "engine is NULL");
    if (nreads == 0) return KDB_OK;
    if (!offs || (!bases && nbytes)) return fail(KDB_ERR_ARG, "NULL host buffer");
    if (offs[nreads] < offs[0] || offs[nreads] > nbytes) return fail(KDB_ERR_ARG, "read_offsets exceed nbytes");
    DeviceGuard g(e->device);
    int rc = ensure_staging(e);
    if (rc != KDB_OK) return rc;
    if (src_pinned) {
        const int pi = e->pin_idx;                 // holds the event of call N-2
        if (!e->ev_pin[pi]) HIP_TRY(hipEventCreateWithFlags(&e->ev_pin[pi], hipEventDisableTiming));
        if (e->pin_used[pi]) HIP_TRY(hipEventSynchronize(e->ev_pin[pi]));
    }
    const uint64_t cap = e->stage_bytes;
    const uint64_t overlap = (uint64_t)(e->k - 1);
    size_t r = 0;
    bool cont = first_continues;                 // the next piece continues a record split across buffers (or across calls)
    uint64_t carry = first_continues ? offs[0] : 0;     // where that piece starts
    while (r < nreads) {
        const int b = e->next_buf;
        if (e->inflight[b]) { HIP_TRY(hipEventSynchronize(e->busy[b])); e->inflight[b] = false; }
        uint8_t *hb = e->h_bases[b];
        uint64_t *ho = e->h_offs[b];
        const int first_is_cont = cont ? 1 : 0;
        const uint64_t start = cont ? carry : offs[r];
        // records are adjacent in `bases`: take the longest run of whole records that fits the buffer
        const size_t rmax = (nreads - r < e->stage_reads) ? nreads : r + e->stage_reads;
        const uint64_t *ub = std::upper_bound(offs + r + 1, offs + rmax + 1, start + cap);
        size_t r1 = (size_t)(ub - offs) - 1;            // offs[r1] <= start + cap
        uint64_t nb;
        size_t nr;
        ho[0] = 0;
        if (r1 <= r) {                                   // the record at r alone exceeds a buffer: tile it with k-1 overlap
            nb = cap; nr = 1; ho[1] = cap;
            carry = start + cap - overlap; cont = true;
        } else {
            nb = offs[r1] - start; nr = r1 - r;
            uint64_t prev = start;
            for (size_t i = 1; i <= nr; i++) {
                const uint64_t o = offs[r + i];
                if (o < prev) return fail(KDB_ERR_ARG, "read_offsets not monotone at %zu", r + i);
                prev = o;
                ho[i] = o - start;
            }
            r = r1; cont = false;
        }
        const bool accumulate = e->acc_ready && !first_is_cont && !cont && nb <= e->acc_cap && nr <= e->acc_reads_cap;
        if (accumulate) {
            if (e->acc_nb + nb > e->acc_cap || e->acc_nr + nr > e->acc_reads_cap) { rc = flush_accumulated(e); if (rc != KDB_OK) return rc; }
            const int s = e->acc_slot;
            if (e->acc_inflight[s]) { HIP_TRY(hipEventSynchronize(e->ev_acc_done[s])); e->acc_inflight[s] = false; }
            for (size_t i = 0; i <= nr; i++) ho[i] += e->acc_nb;                 // rebase onto the accumulated batch
            const uint8_t *src = src_pinned ? bases + start : hb;
            if (!src_pinned) parallel_copy(hb, bases + start, nb, e->copy_threads);
            HIP_TRY(hipMemcpyAsync(e->d_acc_bases[s] + e->acc_nb, src, nb, hipMemcpyHostToDevice, e->s_copy));
            // entries acc_nr .. acc_nr + nr of the offsets (entry acc_nr == acc_nb is rewritten with the same value)
            HIP_TRY(hipMemcpyAsync(e->d_acc_offs[s] + e->acc_nr, ho, (nr + 1) * sizeof(uint64_t), hipMemcpyHostToDevice, e->s_copy));
            HIP_TRY(hipEventRecord(e->ev_copied[b], e->s_copy));
            e->busy[b] = e->ev_copied[b];                                          // staging is free once the copies are done
            e->acc_nb += nb; e->acc_nr += nr;
        } else {
            rc = flush_accumulated(e);                                             // keep the two paths from interleaving on a buffer
            if (rc != KDB_OK) return rc;
            if (src_pinned) {
                HIP_TRY(hipMemcpyAsync(e->d_bases[b], bases + start, nb, hipMemcpyHostToDevice, e->s_copy));
            } else {
                parallel_copy(hb, bases + start, nb, e->copy_threads);
                HIP_TRY(hipMemcpyAsync(e->d_bases[b], hb, nb, hipMemcpyHostToDevice, e->s_copy));
            }
            HIP_TRY(hipMemcpyAsync(e->d_offs[b], ho, (nr + 1) * sizeof(uint64_t), hipMemcpyHostToDevice, e->s_copy));
            HIP_TRY(hipEventRecord(e->ev_copied[b], e->s_copy));
            HIP_TRY(hipStreamWaitEvent(e->s_compute, e->ev_copied[b], 0));
            rc = launch_batch(e, e->d_bases[b], nb, e->d_offs[b], nr, first_is_cont, BATCH_HOST_FED);
            if (rc != KDB_OK) return rc;
            HIP_TRY(hipEventRecord(e->ev_done[b], e->s_compute));
            e->busy[b] = e->ev_done[b];
        }
        e->inflight[b] = true;
        e->next_buf = (b + 1) % NBUF;
    }
    if (src_pinned) {
        HIP_TRY(hipEventRecord(e->ev_pin[e->pin_idx], e->s_copy));
        e->pin_used[e->pin_idx] = true;
        e->pin_idx ^= 1;
    }
    return KDB_OK;
}

int kdb_submit(kdb_engine *e, const uint8_t *bases, size_t nbytes, const uint64_t *offs, size_t nreads)
{
    return submit_impl(e, bases, nbytes, offs, nreads, false);
}

int kdb_submit_pinned(kdb_engine *e, const uint8_t *bases, size_t nbytes, const uint64_t *offs, size_t nreads)
{
    return submit_impl(e, bases, nbytes, offs, nreads, true);
}

int kdb_submit_ex(kdb_engine *e, const uint8_t *bases, size_t nbytes, const uint64_t *offs, size_t nreads, int flags)
{
    if (flags & ~(KDB_SUBMIT_PINNED | KDB_SUBMIT_CONTINUES)) return fail(KDB_ERR_ARG, "kdb_submit_ex: unknown flags 0x%x", flags);
    return submit_impl(e, bases, nbytes, offs, nreads, (flags & KDB_SUBMIT_PINNED) != 0, (flags & KDB_SUBMIT_CONTINUES) != 0);
}

int kdb_host_alloc(void **out, size_t nbytes)
{
    if (!out) return fail(KDB_ERR_ARG, "out is NULL");
    *out = nullptr;
    hipError_t err = hipHostMalloc(out, nbytes ? nbytes : 1, hipHostMallocDefault);
    if (err != hipSuccess) return fail(KDB_ERR_NOMEM, "hipHostMalloc(%zu) failed: %s", nbytes, hipGetErrorString(err));
    return KDB_OK;
}

int kdb_host_free(void *p)
{
    if (p) HIP_TRY(hipHostFree(p));
    return KDB_OK;
}

int kdb_sync(kdb_engine *e)
{
    if (!e) return fail(KDB_ERR_ARG, "engine is NULL");
    DeviceGuard g(e->device);
    { int rc = flush_accumulated(e); if (rc != KDB_OK) return rc; }
    { int rc = flush_pending_paged(e); if (rc != KDB_OK) return rc; }
    HIP_TRY(hipStreamSynchronize(e->s_copy));
    HIP_TRY(hipStreamSynchronize(e->s_compute));
    if (e->s_hist) { HIP_TRY(hipStreamSynchronize(e->s_hist)); e->ov.last = -1; }
    for (int b = 0; b < NBUF; b++) e->inflight[b] = false;
    e->acc_inflight[0] = e->acc_inflight[1] = false;
    if (e->prof) { int rc = prof_collect(e); if (rc != KDB_OK) return rc; }
#ifdef KDB_SC_PROF
    {
        unsigned long long h[32];
        if (hipMemcpyFromSymbol(h, HIP_SYMBOL(kdb::g_sc_prof), sizeof h) == hipSuccess && (h[8] || h[24])) {
            static const char *names[8] = {"ids", "requests+stage-next+writes", "barrier 1", "flush", "barrier 2", "drain", "flush: word, page turn, line list", "flush: list read, line read, store"};
            static const char *kern[2] = {"scatter_bases_kernel", "scatter_ids_kernel"};
            for (int s2 = 0; s2 < 2; s2++) {
                const unsigned long long *g = h + 16 * s2;
                if (!g[8]) continue;
                unsigned long long tot = 0;
                for (int q = 0; q < 8; q++) tot += g[q];
                fprintf(stderr, "[sc_prof] %s: %llu workgroups; wave-cycles by phase:", kern[s2], g[8]);
                for (int q = 0; q < 8; q++) fprintf(stderr, " %s %.1f%%", names[q], 100.0 * (double)g[q] / (double)tot);
                fprintf(stderr, " | total %.3g wave-cycles\n", (double)tot);
            }
            memset(h, 0, sizeof h);
            (void)hipMemcpyToSymbol(HIP_SYMBOL(kdb::g_sc_prof), h, sizeof h);
            // the last launch of each kernel: when its workgroups ended, relative to the launch's span (KDB_SC_WG_DUMP=file: every workgroup's)
            static unsigned long long wg[2][2][1024];
            if (hipMemcpyFromSymbol(wg, HIP_SYMBOL(kdb::g_sc_wg), sizeof wg) == hipSuccess) {
                for (int s2 = 0; s2 < 2; s2++) {
                    std::vector<unsigned long long> st, en;
                    for (int w = 0; w < 1024; w++) if (wg[s2][1][w]) { st.push_back(wg[s2][0][w]); en.push_back(wg[s2][1][w]); }
                    if (en.empty()) continue;
                    const unsigned long long t00 = *std::min_element(st.begin(), st.end()), t11 = *std::max_element(en.begin(), en.end());
                    if (const char *dump = getenv("KDB_SC_WG_DUMP")) {
                        if (FILE *f = fopen(dump, "a")) {
                            fprintf(f, "%s", kern[s2]);
                            for (size_t w = 0; w < en.size(); w++) fprintf(f, " %.4f", (double)(en[w] - t00) / (double)(t11 - t00));
                            fprintf(f, "\n");
                            fclose(f);
                        }
                    }
                    std::sort(en.begin(), en.end());
                    const double t0 = (double)t00, span = (double)t11 - t0;
                    fprintf(stderr, "[sc_prof] %s, last launch: %zu workgroups, %.0f us; ends at min %.1f%% p10 %.1f%% p50 %.1f%% p90 %.1f%% of it\n", kern[s2], en.size(), span / 100.0,
                            100.0 * ((double)en.front() - t0) / span, 100.0 * ((double)en[en.size() / 10] - t0) / span, 100.0 * ((double)en[en.size() / 2] - t0) / span,
                            100.0 * ((double)en[en.size() * 9 / 10] - t0) / span);
                }
                memset(wg, 0, sizeof wg);
                (void)hipMemcpyToSymbol(HIP_SYMBOL(kdb::g_sc_wg), wg, sizeof wg);
            }
        }
    }
#endif
    return check_errors(e);
}

// count_nonzero + Sum over `vec` (stats_kernel), optional copy to the host; leaves the results in *c
static int vector_stats(kdb_engine *e, const unsigned long long *vec, uint64_t *counts_out, kdb::DevCounters *c)
{
    HIP_TRY(hipMemsetAsync(&e->d_ctr->unique, 0, 2 * sizeof(unsigned long long), e->s_compute));
    {
        ProfScope ps(e, KDB_KERNEL_STATS);
        unsigned grid = (unsigned)((e->nbins + 255) / 256);
        if (grid > 256u * 16u) grid = 256u * 16u;
        hipLaunchKernelGGL(kdb::stats_kernel, dim3(grid), dim3(256), 0, e->s_compute, vec, e->nbins, e->d_ctr);
    }
    HIP_TRY(hipGetLastError());
    if (counts_out) {
        touch_pages(counts_out, e->nbins * 8ull, 2 * e->copy_threads);          // (while stats_kernel sweeps the vector)
        HIP_TRY(hipMemcpyAsync(counts_out, vec, e->nbins * 8ull, hipMemcpyDeviceToHost, e->s_compute));
        e->d2h_bytes += e->nbins * 8ull;
    }
    HIP_TRY(hipStreamSynchronize(e->s_compute));
    if (e->prof) { int rc = prof_collect(e); if (rc != KDB_OK) return rc; }
    HIP_TRY(hipMemcpy(c, e->d_ctr, sizeof *c, hipMemcpyDeviceToHost));
    return KDB_OK;
}

int kdb_finish(kdb_engine *e, uint64_t *counts_out, uint64_t *total_kmers, uint64_t *unique_kmers)
{
    if (!e) return fail(KDB_ERR_ARG, "engine is NULL");
    if (e->tableless) return fail(KDB_ERR_STATE, "this engine was created by kdb_create_ids: it has no count vector");
    DeviceGuard g(e->device);
    int rc = kdb_sync(e);
    if (rc != KDB_OK) return rc;
    kdb::DevCounters c;
    if ((rc = vector_stats(e, e->d_table, counts_out, &c)) != KDB_OK) return rc;
    if (c.sum != c.total_kmers)
        return fail(KDB_ERR_STATE, "internal: Sum(counts)=%llu but %llu k-mers were emitted", c.sum, c.total_kmers);
    if (total_kmers) *total_kmers = c.total_kmers;
    if (unique_kmers) *unique_kmers = c.unique;
    return KDB_OK;
}

int kdb_table_stats(kdb_engine *e, uint64_t *counts_out, uint64_t *sum_out, uint64_t *unique_out)
{
    if (!e) return fail(KDB_ERR_ARG, "engine is NULL");
    if (e->tableless) return fail(KDB_ERR_STATE, "this engine was created by kdb_create_ids: it has no count vector");
    DeviceGuard g(e->device);
    int rc = kdb_sync(e);
    if (rc != KDB_OK) return rc;
    kdb::DevCounters c;
    if ((rc = vector_stats(e, e->d_table, counts_out, &c)) != KDB_OK) return rc;
    if (sum_out) *sum_out = c.sum;
    if (unique_out) *unique_out = c.unique;
    return KDB_OK;
}

int kdb_nullomers(kdb_engine *e, int folded, uint64_t *ids_out, uint64_t cap, uint64_t *n_out)
{
    if (!e) return fail(KDB_ERR_ARG, "engine is NULL");
    if (e->tableless) return fail(KDB_ERR_STATE, "this engine was created by kdb_create_ids: it has no count vector");
    if (folded && !e->d_acc_table) return fail(KDB_ERR_STATE, "kdb_nullomers(folded) before any kdb_fold_file");
    DeviceGuard g(e->device);
    int rc = kdb_sync(e);
    if (rc != KDB_OK) return rc;
    const unsigned long long *vec = folded ? e->d_acc_table : e->d_table;
    if (((uintptr_t)vec & 15u) != 0) return fail(KDB_ERR_ARG, "the count vector must be 16-byte aligned for kdb_nullomers");
    const uint64_t ntiles = (e->nbins + kdb::NULL_TILE - 1) / kdb::NULL_TILE;
    const uint64_t nranges = (ntiles + kdb::NULL_RANGE_TILES - 1) / kdb::NULL_RANGE_TILES;
    struct Scratch {
        uint32_t *tile_counts = nullptr; unsigned long long *range_totals = nullptr, *out[2] = {nullptr, nullptr};
        hipEvent_t written[2] = {nullptr, nullptr}, copied[2] = {nullptr, nullptr};
        ~Scratch()
        {
            (void)hipFree(tile_counts); (void)hipFree(range_totals);
            for (int b = 0; b < 2; b++) { (void)hipFree(out[b]); if (written[b]) (void)hipEventDestroy(written[b]); if (copied[b]) (void)hipEventDestroy(copied[b]); }
        }
    } sc;
    if (hipMalloc((void **)&sc.tile_counts, ntiles * sizeof(uint32_t)) != hipSuccess || hipMalloc((void **)&sc.range_totals, nranges * sizeof(unsigned long long)) != hipSuccess) {
        (void)hipGetLastError();
        return fail(KDB_ERR_NOMEM, "kdb_nullomers: no room for %llu tile counts", (unsigned long long)ntiles);
    }
    hipLaunchKernelGGL(kdb::null_count_kernel, dim3((unsigned)std::min<uint64_t>(ntiles, 256u * 16u)), dim3(kdb::NULL_TPB), 0, e->s_compute, vec, e->nbins, ntiles, sc.tile_counts);
    hipLaunchKernelGGL(kdb::null_scan_kernel, dim3((unsigned)nranges), dim3(1024), 0, e->s_compute, sc.tile_counts, ntiles, sc.range_totals);
    HIP_TRY(hipGetLastError());
    std::vector<unsigned long long> totals(nranges);
    HIP_TRY(hipMemcpyAsync(totals.data(), sc.range_totals, nranges * sizeof(unsigned long long), hipMemcpyDeviceToHost, e->s_compute));
    HIP_TRY(hipStreamSynchronize(e->s_compute));
    uint64_t total = 0, largest = 0;
    for (uint64_t r = 0; r < nranges; r++) { total += totals[r]; largest = std::max<uint64_t>(largest, totals[r]); }
    if (n_out) *n_out = total;
    if (!ids_out || total == 0) return KDB_OK;
    if (total > cap) return fail(KDB_ERR_ARG, "kdb_nullomers: %llu ids do not fit the %llu entries given", (unsigned long long)total, (unsigned long long)cap);
    for (int b = 0; b < (nranges > 1 ? 2 : 1); b++) {
        if (hipMalloc((void **)&sc.out[b], largest * 8ull) != hipSuccess) { (void)hipGetLastError(); return fail(KDB_ERR_NOMEM, "kdb_nullomers: no room for %llu ids on the device", (unsigned long long)largest); }
        HIP_TRY(hipEventCreateWithFlags(&sc.written[b], hipEventDisableTiming));
        HIP_TRY(hipEventCreateWithFlags(&sc.copied[b], hipEventDisableTiming));
    }
    touch_pages(ids_out, total * 8ull, 2 * e->copy_threads);
    uint64_t at = 0;
    for (uint64_t r = 0; r < nranges; r++) {
        if (!totals[r]) continue;
        const int b = (int)(r & 1);
        const uint64_t tile0 = r * kdb::NULL_RANGE_TILES;
        const uint32_t here = (uint32_t)std::min<uint64_t>(kdb::NULL_RANGE_TILES, ntiles - tile0);
        HIP_TRY(hipStreamWaitEvent(e->s_compute, sc.copied[b], 0));                 // (the copy of range r - 2 has left this buffer; a no-op before the first record)
        hipLaunchKernelGGL(kdb::null_write_kernel, dim3(std::min<uint32_t>(here, 256u * 16u)), dim3(kdb::NULL_TPB), 0, e->s_compute, vec, e->nbins, tile0, here,
                           (const uint32_t *)sc.tile_counts, sc.out[b]);
        HIP_TRY(hipGetLastError());
        HIP_TRY(hipEventRecord(sc.written[b], e->s_compute));
        HIP_TRY(hipStreamWaitEvent(e->s_copy, sc.written[b], 0));
        HIP_TRY(hipMemcpyAsync(ids_out + at, sc.out[b], totals[r] * 8ull, hipMemcpyDeviceToHost, e->s_copy));
        HIP_TRY(hipEventRecord(sc.copied[b], e->s_copy));
        at += totals[r];
    }
    HIP_TRY(hipStreamSynchronize(e->s_copy));
    HIP_TRY(hipStreamSynchronize(e->s_compute));
    return KDB_OK;
}

int kdb_reduce(kdb_engine *const *engines, int n, int root)
{
    if (!engines || n < 2 || n > KDB_REDUCE_MAX) return fail(KDB_ERR_ARG, "kdb_reduce: n=%d engines (2..%d)", n, KDB_REDUCE_MAX);
    if (root < 0 || root >= n) return fail(KDB_ERR_ARG, "kdb_reduce: root=%d of %d", root, n);
    for (int j = 0; j < n; j++) {
        kdb_engine *e = engines[j];
        if (!e) return fail(KDB_ERR_ARG, "kdb_reduce: engine %d is NULL", j);
        if (e->tableless) return fail(KDB_ERR_STATE, "kdb_reduce: engine %d was created by kdb_create_ids: it has no count vector", j);
        if (e->k != engines[0]->k) return fail(KDB_ERR_ARG, "kdb_reduce: engine %d counts k=%d, engine 0 k=%d", j, e->k, engines[0]->k);
        if (((uintptr_t)e->d_table & 15u) != 0) return fail(KDB_ERR_ARG, "kdb_reduce: the count vector of engine %d is not 16-byte aligned", j);
        for (int i = 0; i < j; i++)
            if (engines[i] == e || engines[i]->d_table == e->d_table) return fail(KDB_ERR_ARG, "kdb_reduce: engines %d and %d are the same vector", i, j);
    }
    for (int j = 0; j < n; j++) { int rc = kdb_sync(engines[j]); if (rc != KDB_OK) return rc; }       // flushes what k >= 13 deferred
    for (int j = 0; j < n; j++)
        for (int i = 0; i < n; i++) {
            const int dj = engines[j]->device, di = engines[i]->device;
            if (dj == di) continue;
            DeviceGuard g(dj);
            int can = 0;
            HIP_TRY(hipDeviceCanAccessPeer(&can, dj, di));
            if (!can) return fail(KDB_ERR_HIP, "kdb_reduce: device %d cannot access the memory of device %d", dj, di);
            const hipError_t pe = hipDeviceEnablePeerAccess(di, 0);
            if (pe != hipSuccess && pe != hipErrorPeerAccessAlreadyEnabled)
                return fail(KDB_ERR_HIP, "hipDeviceEnablePeerAccess(%d) on device %d failed: %s", di, dj, hipGetErrorString(pe));
            (void)hipGetLastError();
        }
    const uint64_t nbins = engines[0]->nbins;
    uint64_t bound[KDB_REDUCE_MAX + 1];
    for (int j = 0; j < n; j++) bound[j] = (nbins * (uint64_t)j / (uint64_t)n) & ~1ull;
    bound[n] = nbins;
    struct Events {
        hipEvent_t ev[KDB_REDUCE_MAX] = {nullptr}; int dev[KDB_REDUCE_MAX] = {0};
        ~Events() { for (int j = 0; j < KDB_REDUCE_MAX; j++) if (ev[j]) { DeviceGuard g(dev[j]); (void)hipEventDestroy(ev[j]); } }
    } evs;
    // every engine sums its slice of all the vectors
    for (int j = 0; j < n; j++) {
        kdb_engine *e = engines[j];
        if (bound[j + 1] <= bound[j]) continue;
        DeviceGuard g(e->device);
        kdb::ReducePeers peers;
        peers.n = 0;
        for (int i = 0; i < n; i++) if (i != j) peers.p[peers.n++] = engines[i]->d_table;
        const uint64_t pairs = (bound[j + 1] - bound[j] + 1) / 2;
        unsigned grid = (unsigned)std::min<uint64_t>((pairs + 255) / 256, 256u * 16u);
        hipLaunchKernelGGL(kdb::reduce_slice_kernel, dim3(grid), dim3(256), 0, e->s_compute, e->d_table, peers, bound[j], bound[j + 1]);
        HIP_TRY(hipGetLastError());
        evs.dev[j] = e->device;
        HIP_TRY(hipEventCreateWithFlags(&evs.ev[j], hipEventDisableTiming));
        HIP_TRY(hipEventRecord(evs.ev[j], e->s_compute));
        e->tp.table_is_zero = false;
    }
    // the root collects the finished slices
    kdb_engine *r = engines[root];
    r->tp.table_is_zero = false;
    {
        DeviceGuard g(r->device);
        for (int j = 0; j < n; j++) {
            if (j == root || !evs.ev[j]) continue;
            kdb_engine *e = engines[j];
            const size_t bytes = (size_t)(bound[j + 1] - bound[j]) * 8u;
            HIP_TRY(hipStreamWaitEvent(r->s_compute, evs.ev[j], 0));
            if (e->device == r->device) HIP_TRY(hipMemcpyAsync(r->d_table + bound[j], e->d_table + bound[j], bytes, hipMemcpyDeviceToDevice, r->s_compute));
            else HIP_TRY(hipMemcpyPeerAsync(r->d_table + bound[j], r->device, e->d_table + bound[j], e->device, bytes, r->s_compute));
        }
        HIP_TRY(hipStreamSynchronize(r->s_compute));
    }
    // ... and now stands for all of them: Sum(counts) == k-mers emitted must hold for the root's kdb_finish
    unsigned long long total = 0;
    for (int j = 0; j < n; j++) {
        DeviceGuard g(engines[j]->device);
        unsigned long long t = 0;
        HIP_TRY(hipStreamSynchronize(engines[j]->s_compute));
        HIP_TRY(hipMemcpy(&t, &engines[j]->d_ctr->total_kmers, sizeof t, hipMemcpyDeviceToHost));
        total += t;
    }
    {
        DeviceGuard g(r->device);
        HIP_TRY(hipMemcpy(&r->d_ctr->total_kmers, &total, sizeof total, hipMemcpyHostToDevice));
    }
    return KDB_OK;
}

int kdb_fold_file(kdb_engine *e, uint64_t *total_kmers, uint64_t *unique_kmers) { return kdb_fold_file_into(e, e, total_kmers, unique_kmers); }

int kdb_fold_file_into(kdb_engine *e, kdb_engine *acc, uint64_t *total_kmers, uint64_t *unique_kmers)
{
    if (!e || !acc) return fail(KDB_ERR_ARG, "engine is NULL");
    if (e->tableless || acc->tableless) return fail(KDB_ERR_STATE, "this engine was created by kdb_create_ids: it has no count vector");
    if (e->k != acc->k || e->device != acc->device) return fail(KDB_ERR_ARG, "kdb_fold_file_into: the engines differ in k or device");
    DeviceGuard g(e->device);
    int rc = kdb_sync(e);
    if (rc != KDB_OK) return rc;
    if (((uintptr_t)e->d_table & 15u) != 0) return fail(KDB_ERR_ARG, "the count vector must be 16-byte aligned for kdb_fold_file");
    if (!acc->d_acc_table) {
        hipError_t me = hipMalloc((void **)&acc->d_acc_table, acc->nbins * 8ull);
        if (me != hipSuccess) { (void)hipGetLastError(); acc->d_acc_table = nullptr; return fail(KDB_ERR_NOMEM, "no room for a second 4^%d vector (accumulator): %s", e->k, hipGetErrorString(me)); }
        HIP_TRY(hipMemsetAsync(acc->d_acc_table, 0, acc->nbins * 8ull, e->s_compute));     // (ordered before this fold; later folds start after it has finished)
    }
    HIP_TRY(hipMemsetAsync(&e->d_ctr->unique, 0, 2 * sizeof(unsigned long long), e->s_compute));
    {
        ProfScope ps(e, KDB_KERNEL_STATS);
        unsigned grid = (unsigned)((e->nbins / 2 + 255) / 256);
        if (grid > 256u * 16u) grid = 256u * 16u;
        if (grid == 0) grid = 1;
        hipLaunchKernelGGL(kdb::fold_kernel, dim3(grid), dim3(256), 0, e->s_compute, e->d_table, acc->d_acc_table, e->nbins, e->d_ctr);
    }
    HIP_TRY(hipGetLastError());
    HIP_TRY(hipStreamSynchronize(e->s_compute));
    if (e->prof) { rc = prof_collect(e); if (rc != KDB_OK) return rc; }
    kdb::DevCounters c;
    HIP_TRY(hipMemcpy(&c, e->d_ctr, sizeof c, hipMemcpyDeviceToHost));
    if (c.sum != c.total_kmers)
        return fail(KDB_ERR_STATE, "internal: Sum(counts)=%llu but %llu k-mers were emitted", c.sum, c.total_kmers);
    if (total_kmers) *total_kmers = c.total_kmers;
    if (unique_kmers) *unique_kmers = c.unique;
    acc->folded_files++;
    acc->folded_total += c.total_kmers;
    // the file vector is all zero again: a new file starts (what kdb_reset does, without a second sweep of the vector)
    HIP_TRY(hipMemsetAsync(&e->d_ctr->total_kmers, 0, sizeof(unsigned long long), e->s_compute));
    HIP_TRY(hipStreamSynchronize(e->s_compute));
    e->tp.table_is_zero = e->owns_table && !e->table_escaped;
    return KDB_OK;
}

int kdb_finish_folded(kdb_engine *e, uint64_t *counts_out, uint64_t *total_kmers, uint64_t *unique_kmers)
{
    if (!e) return fail(KDB_ERR_ARG, "engine is NULL");
    if (!e->d_acc_table) return fail(KDB_ERR_STATE, "kdb_finish_folded before any kdb_fold_file");
    DeviceGuard g(e->device);
    int rc = kdb_sync(e);
    if (rc != KDB_OK) return rc;
    kdb::DevCounters c;
    if ((rc = vector_stats(e, e->d_acc_table, counts_out, &c)) != KDB_OK) return rc;
    if (c.sum != e->folded_total)
        return fail(KDB_ERR_STATE, "internal: Sum(accumulated counts)=%llu but the folded files held %llu k-mers", c.sum, (unsigned long long)e->folded_total);
    if (total_kmers) *total_kmers = c.sum;
    if (unique_kmers) *unique_kmers = c.unique;
    return KDB_OK;
}

int kdb_table(kdb_engine *e, void **d_table_out, uint64_t *nbins_out)
{
    if (!e) return fail(KDB_ERR_ARG, "engine is NULL");
    if (d_table_out) { *d_table_out = e->d_table; e->table_escaped = true; e->tp.table_is_zero = false; }      // (the caller may write it from now on: RCCL reduces into it)
    if (nbins_out) *nbins_out = e->nbins;
    return KDB_OK;
}

int kdb_error_counts(kdb_engine *e, uint64_t *n_short, uint64_t *n_bad)
{
    if (!e) return fail(KDB_ERR_ARG, "engine is NULL");
    DeviceGuard g(e->device);
    { int rc = flush_accumulated(e); if (rc != KDB_OK) return rc; }
    HIP_TRY(hipStreamSynchronize(e->s_compute));
    kdb::DevCounters c;
    HIP_TRY(hipMemcpy(&c, e->d_ctr, sizeof c, hipMemcpyDeviceToHost));
    if (n_short) *n_short = c.n_short;
    if (n_bad) *n_bad = c.n_bad;
    return KDB_OK;
}

int kdb_shred(kdb_engine *e, const uint8_t *seq, size_t nbytes, uint64_t *ids_out, uint64_t *pos_out, size_t cap,
              size_t *n_out)
{
    if (!e) return fail(KDB_ERR_ARG, "engine is NULL");
    if (n_out) *n_out = 0;
    if (!seq && nbytes) return fail(KDB_ERR_ARG, "seq is NULL");
    if (nbytes < (size_t)e->k)
        return fail(KDB_ERR_SHORT_READ, "record of %zu residues is shorter than k=%d (reference: kmer.py:461-463 raises)",
                    nbytes, e->k);
    if (nbytes > (1ull << 30)) return fail(KDB_ERR_ARG, "kdb_shred: at most 2^30 residues per call");
    DeviceGuard g(e->device);
    int rc = shred_scratch(e, nbytes, 0);
    if (rc != KDB_OK) return rc;
    std::vector<unsigned long long> ids(nbytes);
    kdb::DevCounters c;
    memset(&c, 0, sizeof c);
    const unsigned long long sus[2] = {(unsigned long long)(uintptr_t)e->sh_suspects, (unsigned long long)e->suspects_cap};
    HIP_TRY(hipMemcpyAsync(e->sh_seq, seq, nbytes, hipMemcpyHostToDevice, e->s_compute));
    HIP_TRY(hipMemsetAsync(e->sh_ctr, 0, sizeof c, e->s_compute));
    HIP_TRY(hipMemcpyAsync(&e->sh_ctr->sus, sus, sizeof sus, hipMemcpyHostToDevice, e->s_compute));
    hipLaunchKernelGGL(kdb::hibit_check_kernel, dim3(64), dim3(256), 0, e->s_compute, e->sh_seq, (uint64_t)nbytes, e->sh_ctr);
    const unsigned ntiles = (unsigned)((nbytes + kdb::TILE_BYTES - 1) / kdb::TILE_BYTES);
    hipLaunchKernelGGL(kdb::shred_kernel, dim3(ntiles), dim3(kdb::TPB), 0, e->s_compute, e->sh_seq, (uint64_t)nbytes, e->k,
                       e->canonical, e->sh_ids, e->sh_ctr);
    hipLaunchKernelGGL(kdb::resolve_suspects_kernel, dim3(1), dim3(256), 0, e->s_compute, (const uint8_t *)e->sh_seq, (uint64_t)nbytes, (const uint64_t *)nullptr,
                       (uint64_t)1, e->k, e->sh_ctr);
    HIP_TRY(hipGetLastError());
    HIP_TRY(hipMemcpyAsync(ids.data(), e->sh_ids, nbytes * 8ull, hipMemcpyDeviceToHost, e->s_compute));
    HIP_TRY(hipMemcpyAsync(&c, e->sh_ctr, sizeof c, hipMemcpyDeviceToHost, e->s_compute));
    HIP_TRY(hipStreamSynchronize(e->s_compute));
    if (c.n_bad) return fail(KDB_ERR_BAD_RESIDUE, "%llu residue(s) outside ACGTN (reference: kmer.py:309 / :170 raises)", c.n_bad);
    size_t n = 0;
    for (size_t p = 0; p + (size_t)e->k <= nbytes; p++) {
        if (ids[p] == ~0ull) continue;
        if (n < cap) { if (ids_out) ids_out[n] = ids[p]; if (pos_out) pos_out[n] = p; }
        n++;
    }
    if (n_out) *n_out = n;
    return KDB_OK;
}

int kdb_window_ids(kdb_engine *e, const uint8_t *bases, size_t nbytes, const uint64_t *offs, size_t nreads, uint64_t *ids_out)
{
    if (!e) return fail(KDB_ERR_ARG, "engine is NULL");
    if (nreads == 0 || nbytes == 0) return KDB_OK;
    if (!bases || !offs || !ids_out) return fail(KDB_ERR_ARG, "NULL buffer");
    if (offs[0] != 0 || offs[nreads] != nbytes) return fail(KDB_ERR_ARG, "read_offsets must start at 0 and end at nbytes");
    if (nbytes > (1ull << 30)) return fail(KDB_ERR_ARG, "kdb_window_ids: at most 2^30 residues per call");
    DeviceGuard g(e->device);
    int rc = shred_scratch(e, nbytes, nreads);
    if (rc != KDB_OK) return rc;
    kdb::DevCounters c;
    memset(&c, 0, sizeof c);
    HIP_TRY(hipMemcpyAsync(e->sh_seq, bases, nbytes, hipMemcpyHostToDevice, e->s_compute));
    HIP_TRY(hipMemcpyAsync(e->sh_offs, offs, (nreads + 1) * sizeof(uint64_t), hipMemcpyHostToDevice, e->s_compute));
    const unsigned long long sus[2] = {(unsigned long long)(uintptr_t)e->sh_suspects, (unsigned long long)e->suspects_cap};
    HIP_TRY(hipMemsetAsync(e->sh_ctr, 0, sizeof c, e->s_compute));
    HIP_TRY(hipMemcpyAsync(&e->sh_ctr->sus, sus, sizeof sus, hipMemcpyHostToDevice, e->s_compute));
    const dim3 grid((unsigned)((nreads + 255) / 256)), block(256), lgrid(grid.x < 1024u ? grid.x : 1024u);
    hipLaunchKernelGGL(kdb::lens_kernel, lgrid, block, 0, e->s_compute, e->sh_offs, (uint64_t)nreads, (uint64_t)nbytes,
                       e->min_len > 0 ? e->min_len : e->k, 0, e->sh_ctr, (uint32_t *)nullptr);
    hipLaunchKernelGGL(kdb::hibit_check_kernel, dim3(1024), dim3(256), 0, e->s_compute, e->sh_seq, (uint64_t)nbytes, e->sh_ctr);
    hipLaunchKernelGGL(kdb::mark_reads_kernel, dim3(grid.x < 4096u ? grid.x : 4096u), block, 0, e->s_compute, e->sh_seq, e->sh_offs, (uint64_t)nreads, 0,
                       e->sh_ctr);
    const unsigned ntiles = (unsigned)((nbytes + kdb::TILE_BYTES - 1) / kdb::TILE_BYTES);
    hipLaunchKernelGGL(kdb::shred_kernel, dim3(ntiles), dim3(kdb::TPB), 0, e->s_compute, e->sh_seq, (uint64_t)nbytes, e->k,
                       e->canonical, e->sh_ids, e->sh_ctr);
    hipLaunchKernelGGL(kdb::resolve_suspects_kernel, dim3(4), dim3(256), 0, e->s_compute, (const uint8_t *)e->sh_seq, (uint64_t)nbytes, (const uint64_t *)e->sh_offs,
                       (uint64_t)nreads, e->k, e->sh_ctr);
    HIP_TRY(hipGetLastError());
    HIP_TRY(hipMemcpyAsync(ids_out, e->sh_ids, nbytes * 8ull, hipMemcpyDeviceToHost, e->s_compute));
    HIP_TRY(hipMemcpyAsync(&c, e->sh_ctr, sizeof c, hipMemcpyDeviceToHost, e->s_compute));
    HIP_TRY(hipStreamSynchronize(e->s_compute));
    if (c.bad_layout) return fail(KDB_ERR_ARG, "read_offsets must rise from 0 to nbytes");
    if (c.n_short) return fail(KDB_ERR_SHORT_READ, "%llu record(s) shorter than k=%d (reference: kmer.py:461-463 raises)", c.n_short, e->k);
    if (c.n_bad) return fail(KDB_ERR_BAD_RESIDUE, "%llu residue(s) outside ACGTN (reference: kmer.py:309 / :170 raises)", c.n_bad);
    return KDB_OK;
}

int kdb_parse_fastq(const uint8_t *text, size_t n, int at_eof, uint8_t *bases_out, size_t bases_cap, uint64_t *offsets_out,
                    size_t cap_reads, uint64_t *header_spans_out, size_t *nreads_out, size_t *nbases_out, size_t *consumed_out)
{
    if ((!text && n) || !bases_out || !offsets_out || !nreads_out || !nbases_out || !consumed_out) return fail(KDB_ERR_ARG, "NULL argument");
    const char *why = "";
    int rc = kdbhost::parse_fastq(text, n, at_eof, bases_out, bases_cap, offsets_out, cap_reads, header_spans_out, nreads_out,
                                  nbases_out, consumed_out, &why);
    if (rc) return fail(KDB_ERR_ARG, "kdb_parse_fastq: %s", why);
    return KDB_OK;
}

int kdb_parse_fastq_mt(const uint8_t *text, size_t n, int at_eof, uint8_t *bases_out, size_t bases_cap, uint64_t *offsets_out,
                       size_t cap_reads, uint64_t *header_spans_out, size_t *nreads_out, size_t *nbases_out, size_t *consumed_out, int nthreads)
{
    if ((!text && n) || !bases_out || !offsets_out || !nreads_out || !nbases_out || !consumed_out) return fail(KDB_ERR_ARG, "NULL argument");
    const char *why = "";
    int rc = kdbhost::parse_fastq_mt(text, n, at_eof, bases_out, bases_cap, offsets_out, cap_reads, header_spans_out, nreads_out,
                                     nbases_out, consumed_out, &why, nthreads);
    if (rc) return fail(KDB_ERR_ARG, "kdb_parse_fastq: %s", why);
    return KDB_OK;
}

int kdb_parse_fasta(const uint8_t *text, size_t n, uint8_t *bases_out, size_t bases_cap, uint64_t *offsets_out, size_t cap_reads,
                    uint64_t *header_spans_out, size_t *nreads_out, size_t *nbases_out)
{
    if ((!text && n) || !bases_out || !offsets_out || !nreads_out || !nbases_out) return fail(KDB_ERR_ARG, "NULL argument");
    const char *why = "";
    int rc = kdbhost::parse_fasta(text, n, bases_out, bases_cap, offsets_out, cap_reads, header_spans_out, nreads_out, nbases_out, &why);
    if (rc) return fail(KDB_ERR_ARG, "kdb_parse_fasta: %s", why);
    return KDB_OK;
}

int kdb_parse_fasta_chunk(const uint8_t *text, size_t n, int at_eof, int in_record, uint8_t *bases_out, size_t bases_cap, uint64_t *offsets_out,
                          size_t cap_reads, uint64_t *header_spans_out, size_t *nreads_out, size_t *nbases_out, size_t *consumed_out, int *in_record_out)
{
    if ((!text && n) || !bases_out || !offsets_out || !nreads_out || !nbases_out || !consumed_out || !in_record_out) return fail(KDB_ERR_ARG, "NULL argument");
    const char *why = "";
    int rc = kdbhost::parse_fasta_chunk(text, n, at_eof, in_record, bases_out, bases_cap, offsets_out, cap_reads, header_spans_out, nreads_out, nbases_out,
                                        consumed_out, in_record_out, &why);
    if (rc) return fail(KDB_ERR_ARG, "kdb_parse_fasta_chunk: %s", why);
    return KDB_OK;
}

int kdb_bgzf_inflate(const uint8_t *src, size_t n, uint8_t *dst, size_t cap, int nthreads, size_t *consumed_out, size_t *produced_out)
{
    if ((!src && n) || !dst || !consumed_out || !produced_out) return fail(KDB_ERR_ARG, "NULL argument");
    const char *why = "";
    if (kdbhost::bgzf_inflate(src, n, dst, cap, nthreads, consumed_out, produced_out, &why)) return fail(KDB_ERR_ARG, "kdb_bgzf_inflate: %s", why);
    return KDB_OK;
}

int kdb_bgzf_scan(const char *path, uint64_t *coff_out, uint64_t *uoff_out, size_t cap, size_t *n_out)
{
    if (!path || !coff_out || !uoff_out || !n_out) return fail(KDB_ERR_ARG, "NULL argument");
    const char *why = "";
    const int rc = kdbhost::bgzf_scan(path, coff_out, uoff_out, cap, n_out, &why);
    if (rc == 2) return fail(KDB_ERR_NOMEM, "kdb_bgzf_scan('%s'): %s", path, why);
    if (rc) return fail(KDB_ERR_ARG, "kdb_bgzf_scan('%s'): %s", path, why);
    return KDB_OK;
}

struct kdb_gz { kdbhost::GzStream *s; };

int kdb_gz_open(const char *path, kdb_gz **out)
{
    if (!path || !out) return fail(KDB_ERR_ARG, "NULL argument");
    *out = nullptr;
    const char *why = "";
    kdbhost::GzStream *s = kdbhost::gz_open(path, &why);
    if (!s) return fail(KDB_ERR_ARG, "kdb_gz_open('%s'): %s", path, why);
    *out = new kdb_gz{s};
    return KDB_OK;
}

int kdb_gz_read(kdb_gz *g, uint8_t *dst, size_t cap, size_t *n_out)
{
    if (!g || (!dst && cap) || !n_out) return fail(KDB_ERR_ARG, "NULL argument");
    *n_out = 0;
    if (g->s->read(dst, cap, n_out)) return fail(KDB_ERR_ARG, "kdb_gz_read: %s", g->s->err.c_str());
    return KDB_OK;
}

int kdb_gz_close(kdb_gz *g)
{
    if (!g) return KDB_OK;
    delete g->s;
    delete g;
    return KDB_OK;
}

int kdb_write_kdb_rows_ex(const char *path, const uint64_t *counts, uint64_t nbins, uint64_t total_kmers, int compresslevel,
                          int nthreads, int encoder, uint64_t *nblocks_out)
{
    if (!path || (!counts && nbins)) return fail(KDB_ERR_ARG, "NULL argument");
    if (encoder != KDB_ENCODER_DEFAULT && encoder != KDB_ENCODER_ROWS && encoder != KDB_ENCODER_ZLIB)
        return fail(KDB_ERR_ARG, "kdb_write_kdb_rows: unknown encoder %d", encoder);
    if (compresslevel < 0 || compresslevel > 9) return fail(KDB_ERR_ARG, "kdb_write_kdb_rows: compresslevel %d (0..9)", compresslevel);
    const char *why = "";
    if (kdbhost::write_kdb_rows(path, counts, nbins, total_kmers, compresslevel, nthreads, nblocks_out, &why, encoder))
        return fail(KDB_ERR_ARG, "kdb_write_kdb_rows('%s'): %s", path, why);
    return KDB_OK;
}

int kdb_write_kdb_rows(const char *path, const uint64_t *counts, uint64_t nbins, uint64_t total_kmers, int compresslevel,
                       int nthreads, uint64_t *nblocks_out)
{
    return kdb_write_kdb_rows_ex(path, counts, nbins, total_kmers, compresslevel, nthreads, KDB_ENCODER_DEFAULT, nblocks_out);
}

int kdb_copy_back_and_write_kdb_rows(kdb_engine *e, int folded, uint64_t *counts_out, const char *path, uint64_t total_kmers, int compresslevel,
                                     int nthreads, int encoder, uint64_t *nblocks_out)
{
    if (!e || !counts_out || !path) return fail(KDB_ERR_ARG, "NULL argument");
    if (e->tableless) return fail(KDB_ERR_STATE, "this engine was created by kdb_create_ids: it has no count vector");
    if (folded && !e->d_acc_table) return fail(KDB_ERR_STATE, "kdb_copy_back_and_write_kdb_rows(folded) before any kdb_fold_file");
    if (encoder != KDB_ENCODER_DEFAULT && encoder != KDB_ENCODER_ROWS && encoder != KDB_ENCODER_ZLIB)
        return fail(KDB_ERR_ARG, "kdb_copy_back_and_write_kdb_rows: unknown encoder %d", encoder);
    DeviceGuard g(e->device);
    { int rc = kdb_sync(e); if (rc != KDB_OK) return rc; }
    const unsigned long long *vec = folded ? e->d_acc_table : e->d_table;
    const uint64_t nbins = e->nbins;
    touch_pages(counts_out, nbins * 8ull, 2 * e->copy_threads);
    // the vector comes back in pieces on a thread of its own; the writer's threads start on a chunk of rows as soon as it is there
    std::atomic<uint64_t> rows_ready{0};
    hipError_t copy_err = hipSuccess;
    const int device = e->device;
    std::thread copier([&] {
        if (hipSetDevice(device) != hipSuccess) { copy_err = hipErrorInvalidDevice; rows_ready.store(~0ull); return; }
        const uint64_t piece = (128ull << 20) / 8;
        for (uint64_t at = 0; at < nbins; at += piece) {
            const uint64_t n = std::min(piece, nbins - at);
            const hipError_t err = hipMemcpy(counts_out + at, vec + at, n * 8ull, hipMemcpyDeviceToHost);
            if (err != hipSuccess) { copy_err = err; rows_ready.store(~0ull); return; }
            rows_ready.store(at + n, std::memory_order_release);
        }
    });
    const char *why = "";
    const int wrc = kdbhost::write_kdb_rows(path, counts_out, nbins, total_kmers, compresslevel, nthreads, nblocks_out, &why, encoder, &rows_ready);
    copier.join();
    e->d2h_bytes += nbins * 8ull;
    if (copy_err != hipSuccess) return fail(KDB_ERR_HIP, "copying the count vector back failed: %s", hipGetErrorString(copy_err));
    if (wrc) return fail(KDB_ERR_ARG, "kdb_copy_back_and_write_kdb_rows('%s'): %s", path, why);
    return KDB_OK;
}

int kdb_read_kdb_rows(const char *path, uint64_t nbins, uint64_t *kmer_ids_out, uint64_t *counts_out, double *frequencies_out, int nthreads,
                      uint64_t *nrows_out)
{
    if (!path || !kmer_ids_out || !counts_out || !frequencies_out || !nrows_out) return fail(KDB_ERR_ARG, "NULL argument");
    const char *why = "";
    const int rc = kdbhost::read_kdb_rows(path, nbins, kmer_ids_out, counts_out, frequencies_out, nthreads, nrows_out, &why);
    if (rc == 3) return fail(KDB_ERR_STATE, "kdb_read_kdb_rows('%s'): not a file of BGZF members", path);
    if (rc) return fail(KDB_ERR_ARG, "kdb_read_kdb_rows('%s'): %s", path, why);
    return KDB_OK;
}

int kdb_format_frequency(uint64_t count, uint64_t total, char *buf, size_t cap)
{
    if (!buf || cap < 40) return fail(KDB_ERR_ARG, "buffer too small");
    int n = kdbhost::py_float_repr((double)count / (double)total, buf);
    buf[n] = 0;
    return KDB_OK;
}

int kdb_prof_enable(kdb_engine *e, int on)
{
    if (!e) return fail(KDB_ERR_ARG, "engine is NULL");
    e->prof = on != 0;
    return KDB_OK;
}

int kdb_prof_reset(kdb_engine *e)
{
    if (!e) return fail(KDB_ERR_ARG, "engine is NULL");
    DeviceGuard g(e->device);
    HIP_TRY(hipStreamSynchronize(e->s_compute));
    int rc = prof_collect(e);
    if (rc != KDB_OK) return rc;
    for (int i = 0; i < KDB_N_KERNELS; i++) { e->prof_ms[i] = 0; e->prof_n[i] = 0; }
    return KDB_OK;
}

int kdb_prof_get(kdb_engine *e, int kernel_id, double *total_ms, uint64_t *launches)
{
    if (!e) return fail(KDB_ERR_ARG, "engine is NULL");
    if (kernel_id < 0 || kernel_id >= KDB_N_KERNELS) return fail(KDB_ERR_ARG, "kernel_id=%d", kernel_id);
    DeviceGuard g(e->device);
    HIP_TRY(hipStreamSynchronize(e->s_compute));
    int rc = prof_collect(e);
    if (rc != KDB_OK) return rc;
    if (total_ms) *total_ms = e->prof_ms[kernel_id];
    if (launches) *launches = e->prof_n[kernel_id];
    return KDB_OK;
}

// ---- the memory system's ceilings for the kernels' access patterns (diagnostic; bench.py): the table of patterns is above, before the C interface ----
int kdb_hbm_pattern_count(void) { return N_PROBES; }

const char *kdb_hbm_pattern_name(int i) { return (i >= 0 && i < N_PROBES) ? PROBES[i].name : ""; }

int kdb_hbm_pattern_probe(int device_id, double *gb_per_s_out, int n_out)
{
    if (!gb_per_s_out || n_out < N_PROBES) return fail(KDB_ERR_ARG, "kdb_hbm_pattern_probe: room for %d results needed", N_PROBES);
    int ndev = 0;
    HIP_TRY(hipGetDeviceCount(&ndev));
    if (device_id < 0 || device_id >= ndev) return fail(KDB_ERR_ARG, "device_id=%d but %d device(s) visible", device_id, ndev);
    DeviceGuard g(device_id);
    const uint64_t region = 4ull << 30;                  // far beyond the 256 MB memory-side cache: every piece is written (read) once
    uint8_t *src = nullptr, *dst = nullptr; uint32_t *sink = nullptr;
    hipStream_t st = nullptr;
    hipEvent_t a = nullptr, b = nullptr;
    int rc = KDB_OK;
    if (hipMalloc((void **)&src, region) != hipSuccess || hipMalloc((void **)&dst, region) != hipSuccess || hipMalloc((void **)&sink, 64) != hipSuccess) {
        (void)hipGetLastError();
        rc = fail(KDB_ERR_NOMEM, "kdb_hbm_pattern_probe: no room for two regions of 4 GiB");
    } else if (hipStreamCreateWithFlags(&st, hipStreamNonBlocking) != hipSuccess || hipEventCreate(&a) != hipSuccess || hipEventCreate(&b) != hipSuccess ||
               hipMemsetAsync(src, 1, region, st) != hipSuccess || hipMemsetAsync(dst, 0, region, st) != hipSuccess) {
        rc = fail(KDB_ERR_HIP, "kdb_hbm_pattern_probe: setup failed");
    } else {
        for (int i = 0; i < N_PROBES && rc == KDB_OK; i++) {
            const uint64_t per_step = 512ull * 8ull * ((uint64_t)PROBES[i].read_bytes + PROBES[i].write_bytes);
            const uint32_t steps = (uint32_t)(6.0e9 / (double)per_step) + 1;
            float best = 1e30f;
            for (int rep = 0; rep < 3; rep++) {
                (void)hipEventRecord(a, st);
                PROBES[i].launch(src, region, dst, region, steps, sink, st);
                (void)hipEventRecord(b, st);
                if (hipEventSynchronize(b) != hipSuccess) { rc = fail(KDB_ERR_HIP, "kdb_hbm_pattern_probe: pattern '%s' failed", PROBES[i].name); break; }
                float ms = 0;
                (void)hipEventElapsedTime(&ms, a, b);
                if (rep && ms < best) best = ms;
            }
            gb_per_s_out[i] = (double)steps * (double)per_step / 1e6 / (double)best;
        }
    }
    if (a) (void)hipEventDestroy(a);
    if (b) (void)hipEventDestroy(b);
    if (st) (void)hipStreamDestroy(st);
    if (src) (void)hipFree(src);
    if (dst) (void)hipFree(dst);
    if (sink) (void)hipFree(sink);
    return rc;
}

int kdb_set_option(kdb_engine *e, const char *name, int64_t value)
{
    if (!e || !name) return fail(KDB_ERR_ARG, "NULL argument");
    if (!strcmp(name, "algo")) {
        if (value < 0 || value > 3) return fail(KDB_ERR_ARG, "algo=%lld (0 auto, 1 direct atomics, 2 LDS-histogram paths)", (long long)value);
        e->algo = value; return KDB_OK;
    }
    if (!strcmp(name, "defer_flush")) {
        DeviceGuard g(e->device);
        if (!value) { int rc = flush_pending_paged(e); if (rc != KDB_OK) return rc; }
        e->tp.defer = value ? 1 : 0; return KDB_OK;
    }
    if (!strcmp(name, "pending_budget")) {
        if (value < 0) return fail(KDB_ERR_ARG, "pending_budget=%lld", (long long)value);
        e->tp.budget_bytes = (size_t)value; return KDB_OK;
    }
    if (!strcmp(name, "l1_compiled_k")) { e->tp.l1k = value ? 1 : 0; return KDB_OK; }
    if (!strcmp(name, "l1_wide_lines")) { e->tp.l1_wide = value ? 1 : 0; return KDB_OK; }
    if (!strcmp(name, "l1_one_round")) { e->tp.l1_one_round = value ? 1 : 0; return KDB_OK; }
    if (!strcmp(name, "l2_wide_lines")) { e->tp.l2_wide = value ? 1 : 0; return KDB_OK; }
    if (!strcmp(name, "reserve_bytes")) {
        if (value < 0) return fail(KDB_ERR_ARG, "reserve_bytes=%lld", (long long)value);
        e->tp.reserve_bytes = (size_t)value; return KDB_OK;
    }
    if (!strcmp(name, "arena_grow")) {
        if (value < 0 || value > 2) return fail(KDB_ERR_ARG, "arena_grow=%lld (0 never, 1 when it pays, 2 whenever the arena has filled up)", (long long)value);
        e->tp.grow = (int)value; return KDB_OK;
    }
    if (!strcmp(name, "smallk_old")) { e->smallk_old = value ? 1 : 0; return KDB_OK; }
    if (!strcmp(name, "one_level_max_k")) {
        if (value != 12 && value != kdb::SC1_K) return fail(KDB_ERR_ARG, "one_level_max_k=%lld (12 or %d)", (long long)value, kdb::SC1_K);
        DeviceGuard g(e->device);
        { int rc = flush_pending_paged(e); if (rc != KDB_OK) return rc; }
        e->one_level_max_k = (int)value; return KDB_OK;
    }
    if (!strcmp(name, "arena_batches")) {
        if (value < 1 || value > kdb::PAGED_PENDING_MAX) return fail(KDB_ERR_ARG, "arena_batches=%lld (1..%d: the arena's first size, in batches like the first one)", (long long)value, kdb::PAGED_PENDING_MAX);
        e->tp.first_batches = (int)value; return KDB_OK;
    }
    if (!strcmp(name, "sc_grid")) {
        if (value < 0 || value > 1024) return fail(KDB_ERR_ARG, "sc_grid=%lld (0..1024)", (long long)value);
        e->sc.grid = (int)value; e->tp.l1.grid = (int)value; return KDB_OK;
    }
    if (!strcmp(name, "overlap") || !strcmp(name, "overlap_hist_cus") || !strcmp(name, "overlap_mask_mode")) {
        if (value < 0 || value > 1024) return fail(KDB_ERR_ARG, "%s=%lld", name, (long long)value);
        DeviceGuard g(e->device);
        { int rc = kdb_sync(e); if (rc != KDB_OK) return rc; }
        if (!strcmp(name, "overlap")) e->overlap = value ? 1 : 0;
        else if (!strcmp(name, "overlap_hist_cus")) e->overlap_hist_cus = (int)value;
        else e->overlap_mask_mode = (int)value;
        return overlap_streams(e);
    }
#ifdef KDB_SC_PROF
    if (!strcmp(name, "sc_ablate")) { int v = (int)value; (void)hipMemcpyToSymbol(HIP_SYMBOL(kdb::g_sc_ablate), &v, sizeof v); return KDB_OK; }
#endif
    if (!strcmp(name, "sc_lo_bits") || !strcmp(name, "sc_top_bits")) {
        // id = [ hi ][ bucket fields ][ lo ]: how many of a bucket's 15 (k = 17: 16) bin bits sit below the bucket field
        // (sc_top_bits=1 is the old name of sc_lo_bits=15: buckets from the leading id bits)
        int64_t lo = !strcmp(name, "sc_top_bits") ? (value ? kdb::SC_LO_BITS_MAX : 0) : value;
        if (lo < 0 || lo > kdb::SC_LO_BITS_MAX) return fail(KDB_ERR_ARG, "sc_lo_bits=%lld (0 = the default of the path, 1..%d)", (long long)lo, kdb::SC_LO_BITS_MAX);
        DeviceGuard g(e->device);
        { int rc = flush_pending_paged(e); if (rc != KDB_OK) return rc; }        // (pending batches were scattered with the old split)
        e->sc.lo_bits = (int)lo; e->tp.l1.lo_bits = (int)lo; return KDB_OK;
    }
    if (!strcmp(name, "sc_contig_pages")) { e->sc.contig_pages = value ? 1 : 0; e->tp.l1.contig_pages = value ? 1 : 0; return KDB_OK; }
    if (!strcmp(name, "sc_wide_lines")) { e->sc.wide_lines = value ? 1 : 0; return KDB_OK; }
    if (!strcmp(name, "accum_bytes")) {
        if (e->staging_ready) return fail(KDB_ERR_STATE, "staging already allocated");
        if (value < -1) return fail(KDB_ERR_ARG, "accum_bytes=%lld (-1 auto, 0 off, else bytes)", (long long)value);
        e->accum_bytes = value; return KDB_OK;
    }
    if (!strcmp(name, "stage_bytes")) {
        if (e->staging_ready) return fail(KDB_ERR_STATE, "staging already allocated");
        if (value < 4096 || (value & 15)) return fail(KDB_ERR_ARG, "stage_bytes=%lld (>=4096, multiple of 16)", (long long)value);
        e->stage_bytes = (size_t)value; return KDB_OK;
    }
    if (!strcmp(name, "min_len")) {
        if (value < 0 || value > 64) return fail(KDB_ERR_ARG, "min_len=%lld", (long long)value);
        e->min_len = (int)value; return KDB_OK;
    }
    if (!strcmp(name, "copy_threads")) {
        if (value < 1 || value > 64) return fail(KDB_ERR_ARG, "copy_threads=%lld", (long long)value);
        e->copy_threads = (int)value; return KDB_OK;
    }
    if (!strcmp(name, "stage_reads")) {
        if (e->staging_ready) return fail(KDB_ERR_STATE, "staging already allocated");
        if (value < 1) return fail(KDB_ERR_ARG, "stage_reads=%lld", (long long)value);
        e->stage_reads = (size_t)value; return KDB_OK;
    }
    return fail(KDB_ERR_ARG, "unknown option '%s'", name);
}

int kdb_get_option(kdb_engine *e, const char *name, int64_t *value)
{
    if (!e || !name || !value) return fail(KDB_ERR_ARG, "NULL argument");
    if (!strcmp(name, "algo")) { *value = e->algo; return KDB_OK; }
    if (!strcmp(name, "reserve_bytes")) { *value = (int64_t)e->tp.reserve_bytes; return KDB_OK; }
    if (!strcmp(name, "arena_budget_bytes")) { *value = (int64_t)e->tp.budget_bytes; return KDB_OK; }
    if (!strcmp(name, "free_at_sizing")) { *value = (int64_t)e->tp.free_at_sizing; return KDB_OK; }
    if (!strcmp(name, "free_hbm")) {
        DeviceGuard g(e->device);
        size_t free_b = 0, total_b = 0;
        HIP_TRY(hipMemGetInfo(&free_b, &total_b));
        *value = (int64_t)free_b; return KDB_OK;
    }
    if (!strcmp(name, "overlap")) { *value = e->overlap; return KDB_OK; }
    if (!strcmp(name, "overlap_hist_cus")) { *value = e->overlap_hist_cus; return KDB_OK; }
    if (!strcmp(name, "overlap_scatter_grid")) { *value = e->ov.grid; return KDB_OK; }
    if (!strcmp(name, "stage_bytes")) { *value = (int64_t)e->stage_bytes; return KDB_OK; }
    if (!strcmp(name, "stage_reads")) { *value = (int64_t)e->stage_reads; return KDB_OK; }
    if (!strcmp(name, "k")) { *value = e->k; return KDB_OK; }
    if (!strcmp(name, "defer_flush")) { *value = e->tp.defer; return KDB_OK; }
    if (!strcmp(name, "sc_lo_bits")) { *value = e->sc.lo_bits; return KDB_OK; }
    if (!strcmp(name, "sc_contig_pages")) { *value = e->sc.contig_pages; return KDB_OK; }
    if (!strcmp(name, "sc_wide_lines")) { *value = e->sc.wide_lines; return KDB_OK; }
    if (!strcmp(name, "l1_wide_lines")) { *value = e->tp.l1_wide; return KDB_OK; }
    if (!strcmp(name, "l2_wide_lines")) { *value = e->tp.l2_wide; return KDB_OK; }
    if (!strcmp(name, "l1_one_round")) { *value = e->tp.l1_one_round; return KDB_OK; }
    if (!strcmp(name, "oom_fallbacks")) { *value = e->oom_fallbacks; return KDB_OK; }
    if (!strcmp(name, "pending_batches")) { *value = (int64_t)e->tp.pending; return KDB_OK; }
    if (!strcmp(name, "d2h_bytes")) { *value = (int64_t)e->d2h_bytes; return KDB_OK; }
    if (!strcmp(name, "folded_files")) { *value = (int64_t)e->folded_files; return KDB_OK; }
    if (!strcmp(name, "bytes_in")) { *value = (int64_t)e->bytes_in; return KDB_OK; }
    if (!strcmp(name, "arena_pages")) { *value = (int64_t)e->tp.cap2; return KDB_OK; }
    if (!strcmp(name, "arena_used_bound") || !strcmp(name, "arena_worst_case") || !strcmp(name, "arena_cursor")) {
        // pages of the arena the pending batches hold: the host's present bound (worst cases minus what the read-backs of the device's
        // cursor have shown to be unused), the worst cases added up, and the device's cursor itself (synchronises the compute stream)
        DeviceGuard g(e->device);
        if (!strcmp(name, "arena_cursor")) {
            uint32_t c = 0;
            HIP_TRY(hipStreamSynchronize(e->s_compute));
            if (e->tp.d_cursor && e->tp.pending) HIP_TRY(hipMemcpy(&c, e->tp.d_cursor, sizeof c, hipMemcpyDeviceToHost));
            *value = (int64_t)c; return KDB_OK;
        }
        kdb::twolevel_paged_poll(e->tp);
        *value = (int64_t)(!strcmp(name, "arena_worst_case") ? e->tp.used2 : e->tp.used2 - e->tp.slack); return KDB_OK;
    }
    if (!strcmp(name, "arena_reallocs")) { *value = (int64_t)e->tp.reallocs; return KDB_OK; }
    if (!strcmp(name, "arena_grow")) { *value = e->tp.grow; return KDB_OK; }
    if (!strcmp(name, "arena_batches")) { *value = e->tp.first_batches; return KDB_OK; }
    if (!strcmp(name, "one_level_max_k")) { *value = e->one_level_max_k; return KDB_OK; }
    if (!strcmp(name, "smallk_old")) { *value = e->smallk_old; return KDB_OK; }
    if (!strcmp(name, "hist_flushes")) { *value = (int64_t)e->tp.flushes; return KDB_OK; }
    if (!strcmp(name, "flushed_batches")) { *value = (int64_t)e->tp.flushed_batches; return KDB_OK; }
    if (!strcmp(name, "full_flushes")) { *value = (int64_t)e->tp.full_flushes; return KDB_OK; }
    {
        // HBM traffic by the engine's own account (cumulative since kdb_reset; the device is synchronised to read them)
        static const struct { const char *name; size_t off; } dev[] = {
            {"pages_bases", offsetof(kdb::DevCounters, pages_bases)}, {"lines_bases", offsetof(kdb::DevCounters, lines_bases)},
            {"pages_ids", offsetof(kdb::DevCounters, pages_ids)},     {"lines_ids", offsetof(kdb::DevCounters, lines_ids)},
            {"table_bytes", offsetof(kdb::DevCounters, table_bytes)}, {"total_kmers", offsetof(kdb::DevCounters, total_kmers)}};
        for (const auto &d : dev)
            if (!strcmp(name, d.name)) {
                DeviceGuard g(e->device);
                HIP_TRY(hipStreamSynchronize(e->s_compute));
                unsigned long long v = 0;
                HIP_TRY(hipMemcpy(&v, (const char *)e->d_ctr + d.off, sizeof v, hipMemcpyDeviceToHost));
                *value = (int64_t)v;
                return KDB_OK;
            }
    }
    return fail(KDB_ERR_ARG, "unknown option '%s'", name);
}

}  // extern "C"
