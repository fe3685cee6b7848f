// kdb_hostparse.cpp.h -- host-side record splitter (no GPU): FASTQ / FASTA text -> flat residue buffer + offsets.
// Takes over what Bio.SeqIO.parse does for kmerdb/parse.py:50-85 on the path: record ids are the first
// whitespace token of the header, FASTA lines are concatenated (blanks / CR stripped), case is preserved.
// memchr-driven; one pass over the text.
#pragma once
#include <cstdint>
#include <cstring>
#include <atomic>
#include <condition_variable>
#include <cstdio>
#include <mutex>
#include <string>
#include <thread>
#include <vector>
#include <zlib.h>
#include <fcntl.h>
#include <sys/mman.h>
#include <sys/stat.h>
#include <unistd.h>

#include "kdb_inflate.cpp.h"

namespace kdbhost {

// returns 0 ok; 1 malformed; 2 output capacity exceeded
inline int parse_fastq(const uint8_t *text, size_t n, int at_eof, uint8_t *bases, size_t bases_cap, uint64_t *offs, size_t cap_reads,
                       uint64_t *hdr /* optional [2*cap_reads]: header line start/end */, size_t *nreads_out, size_t *nbases_out,
                       size_t *consumed_out, const char **why)
{
    size_t pos = 0, nreads = 0, nb = 0;
    offs[0] = 0;
    while (pos < n) {
        // skip blank lines between records
        if (text[pos] == '\n' || text[pos] == '\r') { pos++; continue; }
        const uint8_t *l0 = text + pos;
        const uint8_t *e0 = (const uint8_t *)memchr(l0, '\n', n - pos);
        if (!e0) break;
        const uint8_t *l1 = e0 + 1;
        const uint8_t *e1 = (const uint8_t *)memchr(l1, '\n', (size_t)(text + n - l1));
        if (!e1) break;
        const uint8_t *l2 = e1 + 1;
        const uint8_t *e2 = (const uint8_t *)memchr(l2, '\n', (size_t)(text + n - l2));
        if (!e2) break;
        const uint8_t *l3 = e2 + 1;
        const uint8_t *e3 = (const uint8_t *)memchr(l3, '\n', (size_t)(text + n - l3));
        if (!e3) { if (!at_eof) break; e3 = text + n; }          // last record of the file without trailing newline
        if (*l0 != '@') { *why = "FASTQ record does not start with '@'"; return 1; }
        if (l2 >= text + n || *l2 != '+') { *why = "FASTQ third line does not start with '+'"; return 1; }
        size_t slen = (size_t)(e1 - l1), qlen = (size_t)(e3 - l3);
        if (slen && l1[slen - 1] == '\r') slen--;
        if (qlen && l3[qlen - 1] == '\r') qlen--;
        if (slen != qlen) { *why = "FASTQ sequence and quality lengths differ"; return 1; }
        if (nreads >= cap_reads || nb + slen > bases_cap) { *why = "output capacity exceeded"; return 2; }
        memcpy(bases + nb, l1, slen);
        nb += slen;
        if (hdr) { size_t hl = (size_t)(e0 - l0); if (hl && l0[hl - 1] == '\r') hl--; hdr[2 * nreads] = (uint64_t)(l0 - text); hdr[2 * nreads + 1] = (uint64_t)(l0 - text) + hl; }
        offs[++nreads] = nb;
        pos = (size_t)(e3 - text) + (e3 < text + n ? 1 : 0);
    }
    if (at_eof) {
        // anything left must be blank
        for (size_t i = pos; i < n; i++)
            if (text[i] != '\n' && text[i] != '\r' && text[i] != ' ') { *why = "truncated FASTQ record at end of file"; return 1; }
        pos = n;
    }
    *nreads_out = nreads; *nbases_out = nb; *consumed_out = pos;
    return 0;
}

inline int parse_fasta(const uint8_t *text, size_t n, uint8_t *bases, size_t bases_cap, uint64_t *offs, size_t cap_reads,
                       uint64_t *hdr, size_t *nreads_out, size_t *nbases_out, const char **why)
{
    size_t pos = 0, nreads = 0, nb = 0;
    bool in_record = false;
    offs[0] = 0;
    while (pos < n) {
        const uint8_t *l = text + pos;
        const uint8_t *e = (const uint8_t *)memchr(l, '\n', n - pos);
        size_t len = e ? (size_t)(e - l) : n - pos;
        pos += len + (e ? 1 : 0);
        if (len && l[len - 1] == '\r') len--;
        if (len && l[0] == '>') {
            if (in_record) offs[nreads] = nb;
            if (nreads >= cap_reads) { *why = "output capacity exceeded"; return 2; }
            if (hdr) { hdr[2 * nreads] = (uint64_t)(l - text); hdr[2 * nreads + 1] = (uint64_t)(l - text) + len; }
            nreads++;
            in_record = true;
            continue;
        }
        if (!in_record || !len) continue;                      // text before the first header is ignored
        if (nb + len > bases_cap) { *why = "output capacity exceeded"; return 2; }
        // copy the line, dropping blanks inside it (Biopython strips spaces and CR from FASTA sequence lines)
        if (!memchr(l, ' ', len) && !memchr(l, '\t', len) && !memchr(l, '\r', len)) {
            memcpy(bases + nb, l, len);
            nb += len;
        } else {
            for (size_t i = 0; i < len; i++) { const uint8_t c = l[i]; if (c != ' ' && c != '\t' && c != '\r') bases[nb++] = c; }
        }
    }
    if (in_record) offs[nreads] = nb;
    *nreads_out = nreads; *nbases_out = nb;
    return 0;
}

// parse_fasta for one chunk of a file that is read in pieces: `in_record` says that the chunk starts inside a record
// (a record longer than a chunk).  Its sequence lines before the first header then form record 0 of the output, a
// CONTINUATION piece (no header span).  Only complete lines are consumed unless at_eof.  *in_record_out = the chunk
// ends inside a record (1; 2 = in the middle of one of its lines -- pass the value back as `in_record` of the next chunk).
inline int parse_fasta_chunk(const uint8_t *text, size_t n, int at_eof, int in_record_in, uint8_t *bases, size_t bases_cap, uint64_t *offs,
                             size_t cap_reads, uint64_t *hdr, size_t *nreads_out, size_t *nbases_out, size_t *consumed_out, int *in_record_out,
                             const char **why)
{
    size_t pos = 0, nreads = 0, nb = 0;
    bool in_record = in_record_in != 0;
    bool mid_line = in_record_in == 2;                       // the chunk starts in the middle of a sequence line: its first line is sequence whatever it starts with
    offs[0] = 0;
    if (in_record) {                                         // record 0 = the continuation piece (possibly empty)
        if (cap_reads < 1) { *why = "output capacity exceeded"; return 2; }
        if (hdr) { hdr[0] = 0; hdr[1] = 0; }
        nreads = 1;
    }
    while (pos < n) {
        const uint8_t *l = text + pos;
        const uint8_t *e = (const uint8_t *)memchr(l, '\n', n - pos);
        // incomplete last line: a header waits for its end (the caller brings it back with the next chunk); sequence
        // text is taken as far as it goes (unwrapped FASTA keeps a whole chromosome on one line)
        const bool seq_line = mid_line || (in_record && l[0] != '>');
        if (!e && !at_eof && !seq_line) break;
        size_t len = e ? (size_t)(e - l) : n - pos;
        pos += len + (e ? 1 : 0);
        const bool was_mid = mid_line;
        mid_line = !e && !at_eof;                            // consumed up to the end of the chunk without a newline
        if (len && l[len - 1] == '\r') len--;
        if (len && l[0] == '>' && !was_mid) {
            if (in_record) offs[nreads] = nb;
            if (nreads >= cap_reads) { *why = "output capacity exceeded"; return 2; }
            if (hdr) { hdr[2 * nreads] = (uint64_t)(l - text); hdr[2 * nreads + 1] = (uint64_t)(l - text) + len; }
            nreads++;
            in_record = true;
            continue;
        }
        if (!in_record || !len) continue;
        if (nb + len > bases_cap) { *why = "output capacity exceeded"; return 2; }
        if (!memchr(l, ' ', len) && !memchr(l, '\t', len) && !memchr(l, '\r', len)) {
            memcpy(bases + nb, l, len);
            nb += len;
        } else {
            for (size_t i = 0; i < len; i++) { const uint8_t c = l[i]; if (c != ' ' && c != '\t' && c != '\r') bases[nb++] = c; }
        }
    }
    if (in_record) offs[nreads] = nb;
    *nreads_out = nreads; *nbases_out = nb; *consumed_out = pos; *in_record_out = in_record ? (mid_line ? 2 : 1) : 0;
    return 0;
}

// ---------------------------------------------------------------------------------------------------------------
// BGZF (blocked gzip: what bgzip and Bio.bgzf write; every member carries its compressed size in a 'BC' extra
// subfield and its uncompressed size in ISIZE) inflated block-parallel.  Python's gzip module inflates one member
// after the other on one core; here the members of a chunk are found first and then inflated by `nthreads` threads,
// each into its final place.  Returns 0 ok, 1 malformed / not BGZF, 2 a block does not fit `cap`.
// ---------------------------------------------------------------------------------------------------------------
struct BgzfBlock { size_t src, csize, dst, isize, data_off; };

// 0: whole block at src[at..]; 1: not a BGZF member; 2: incomplete (need more input)
inline int bgzf_block_at(const uint8_t *src, size_t n, size_t at, BgzfBlock *b)
{
    if (n - at < 18) return 2;
    const uint8_t *p = src + at;
    if (p[0] != 0x1f || p[1] != 0x8b || p[2] != 8 || !(p[3] & 4)) return 1;
    const size_t xlen = (size_t)p[10] | ((size_t)p[11] << 8);
    if (n - at < 12 + xlen) return 2;
    size_t bsize = 0, x = 12;
    bool found = false;
    while (x + 4 <= 12 + xlen) {
        const size_t slen = (size_t)p[x + 2] | ((size_t)p[x + 3] << 8);
        if (p[x] == 'B' && p[x + 1] == 'C' && slen == 2 && x + 6 <= 12 + xlen) { bsize = ((size_t)p[x + 4] | ((size_t)p[x + 5] << 8)) + 1; found = true; }
        x += 4 + slen;
    }
    if (!found || (p[3] & ~4)) return 1;                     // (no FNAME / FCOMMENT / FHCRC in BGZF members)
    if (bsize < 12 + xlen + 8) return 1;
    if (n - at < bsize) return 2;
    b->src = at; b->csize = bsize; b->data_off = 12 + xlen;
    b->isize = (size_t)p[bsize - 4] | ((size_t)p[bsize - 3] << 8) | ((size_t)p[bsize - 2] << 16) | ((size_t)p[bsize - 1] << 24);
    return 0;
}

inline int bgzf_inflate(const uint8_t *src, size_t n, uint8_t *dst, size_t cap, int nthreads, size_t *consumed, size_t *produced, const char **why)
{
    std::vector<BgzfBlock> blocks;
    size_t at = 0, out = 0;
    *consumed = 0; *produced = 0;
    while (at < n) {
        BgzfBlock b;
        const int rc = bgzf_block_at(src, n, at, &b);
        if (rc == 1) { *why = "not a BGZF member"; return 1; }
        if (rc == 2) break;
        if (out + b.isize > cap) { if (blocks.empty()) { *why = "a BGZF block does not fit the output buffer"; return 2; } break; }
        b.dst = out;
        blocks.push_back(b);
        out += b.isize; at += b.csize;
    }
    if (blocks.empty()) return 0;
    std::atomic<size_t> next(0);
    std::atomic<int> bad(0);
    auto work = [&] {
        z_stream zs;
        for (;;) {
            const size_t i = next.fetch_add(1);
            if (i >= blocks.size() || bad.load()) return;
            const BgzfBlock &b = blocks[i];
            memset(&zs, 0, sizeof zs);
            if (inflateInit2(&zs, -15) != Z_OK) { bad = 1; return; }
            zs.next_in = const_cast<Bytef *>(src + b.src + b.data_off);
            zs.avail_in = (uInt)(b.csize - b.data_off - 8);
            zs.next_out = dst + b.dst;
            zs.avail_out = (uInt)b.isize;
            const int rc = b.isize ? inflate(&zs, Z_FINISH) : Z_STREAM_END;
            const bool ok = (rc == Z_STREAM_END || (b.isize == 0 && rc == Z_BUF_ERROR)) && zs.total_out == b.isize;
            inflateEnd(&zs);
            const uint8_t *t = src + b.src + b.csize - 8;
            const uint32_t want = (uint32_t)t[0] | ((uint32_t)t[1] << 8) | ((uint32_t)t[2] << 16) | ((uint32_t)t[3] << 24);
            if (!ok || (uint32_t)crc32(crc32(0L, Z_NULL, 0), dst + b.dst, (uInt)b.isize) != want) { bad = 1; return; }
        }
    };
    int t = nthreads < 1 ? 1 : nthreads;
    if ((size_t)t > blocks.size()) t = (int)blocks.size();
    std::vector<std::thread> th;
    for (int i = 1; i < t; i++) th.emplace_back(work);
    work();
    for (auto &x : th) x.join();
    if (bad.load()) { *why = "corrupt BGZF block (inflate or CRC32 failed)"; return 1; }
    *consumed = at; *produced = out;
    return 0;
}

// ---------------------------------------------------------------------------------------------------------------
// Index of a BGZF file: cumulative compressed and uncompressed offsets of its members (n + 1 entries each), from the
// members' own headers (BSIZE) and trailers (ISIZE): 36 bytes read per member, nothing inflated.  With it a rank of a
// multi-GPU job inflates only the members that hold ITS blocks (kmerdb_amd.reader.ShardedBlockReader).
// Returns 0 ok, 1 not BGZF / unreadable, 2 `cap` entries do not hold the index (*n_out = members found so far + 1).
// ---------------------------------------------------------------------------------------------------------------
inline int bgzf_scan(const char *path, uint64_t *coff, uint64_t *uoff, size_t cap, size_t *n_out, const char **why)
{
    *n_out = 0;
    FILE *f = fopen(path, "rb");
    if (!f) { *why = "cannot open the file"; return 1; }
    struct Closer { FILE *f; ~Closer() { fclose(f); } } closer{f};
    if (fseeko(f, 0, SEEK_END) != 0) { *why = "cannot seek"; return 1; }
    const uint64_t size = (uint64_t)ftello(f);
    uint64_t at = 0, u = 0;
    size_t n = 0;
    // one seek + read per member: a member's 4-byte ISIZE trailer and the next member's header lie side by side
    uint8_t buf[4 + 32];
    uint8_t *const h = buf + 4;
    size_t got = 0;
    if (size) {
        if (fseeko(f, 0, SEEK_SET) != 0) { *why = "cannot seek"; return 1; }
        got = fread(h, 1, 32, f);
    }
    while (at < size) {
        if (n + 1 >= cap) { *n_out = n + 1; *why = "index buffer too small"; return 2; }
        if (got < 18 || h[0] != 0x1f || h[1] != 0x8b || h[2] != 8 || h[3] != 4) { *why = "not a BGZF member"; return 1; }
        const size_t xlen = (size_t)h[10] | ((size_t)h[11] << 8);
        size_t bsize = 0, x = 12;
        while (x + 6 <= 12 + xlen && x + 6 <= got) {
            const size_t slen = (size_t)h[x + 2] | ((size_t)h[x + 3] << 8);
            if (h[x] == 'B' && h[x + 1] == 'C' && slen == 2) { bsize = ((size_t)h[x + 4] | ((size_t)h[x + 5] << 8)) + 1; break; }
            x += 4 + slen;
        }
        if (bsize < 12 + xlen + 8 || at + bsize > size) { *why = "truncated or malformed BGZF member"; return 1; }
        if (fseeko(f, (off_t)(at + bsize - 4), SEEK_SET) != 0) { *why = "cannot seek"; return 1; }
        const size_t r = fread(buf, 1, sizeof buf, f);
        if (r < 4) { *why = "cannot read a member trailer"; return 1; }
        got = r - 4;                                       // what follows the trailer is the next member's header (or the end of the file)
        coff[n] = at; uoff[n] = u;
        u += (uint64_t)buf[0] | ((uint64_t)buf[1] << 8) | ((uint64_t)buf[2] << 16) | ((uint64_t)buf[3] << 24);
        at += bsize;
        n++;
    }
    if (n + 1 > cap) { *n_out = n + 1; *why = "index buffer too small"; return 2; }
    coff[n] = at; uoff[n] = u;
    *n_out = n + 1;
    return 0;
}

// ---------------------------------------------------------------------------------------------------------------
// One gzip stream (what the reference opens with gzip.open, kmerdb/parse.py:64-72) inflated by a thread of its own, ahead
// of the reader, so that inflating overlaps record splitting, hashing, copying and counting without Python's GIL or its
// gzip module in the way.  A deflate stream cannot be entered in the middle: one decoding thread per file is all there is
// (BGZF files go through bgzf_inflate instead), so the decoder itself is the lever: kdb_inflate.cpp.h (1.3-1.5 x zlib's
// rate on FASTQ text).  The CRC-32 of every member is checked by a second thread that trails the decoder through the
// same buffers.  Concatenated members are concatenated output, like gzip.open.
//
// The file is mapped; the output ring has NBUF slots of [ 32 KiB history | BUF data | slack ]: the decoder writes straight
// into a slot (after copying the previous slot's last 32 KiB in front of it), a slot never holds bytes of two members.
// ---------------------------------------------------------------------------------------------------------------
struct GzStream {
    static constexpr size_t BUF = 4u << 20, NBUF = 16, SLACK = 512;
    struct Slot { std::vector<uint8_t> mem; size_t len = 0; bool member_end = false; uint32_t crc_want = 0, isize_want = 0; };
    int fd = -1;
    const uint8_t *map = nullptr;
    size_t map_len = 0;
    std::thread th_inflate, th_check;
    std::mutex mu;
    std::condition_variable cv;
    Slot slot[NBUF];
    size_t tail = 0, chead = 0, head = 0;   // slots [head, tail) hold unread data; [chead, tail) are not CRC-checked yet (indices grow)
    size_t rpos = 0;                        // read position inside slot `head`
    bool produced_all = false, checked_all = false, stop = false;
    std::string err;

    uint8_t *data(size_t i) { return slot[i % NBUF].mem.data() + Inflater::WINDOW; }

    void set_error(const char *why)
    {
        std::lock_guard<std::mutex> lk(mu);
        if (err.empty()) err = why;
        produced_all = true;
        cv.notify_all();
    }

    // gzip member header (RFC 1952) at p: returns the first byte of the deflate data, nullptr if malformed / truncated
    static const uint8_t *skip_header(const uint8_t *p, const uint8_t *end)
    {
        if (end - p < 10 || p[0] != 0x1f || p[1] != 0x8b || p[2] != 8 || (p[3] & 0xE0)) return nullptr;
        const uint8_t flg = p[3];
        p += 10;
        if (flg & 4) { if (end - p < 2) return nullptr; const size_t xlen = (size_t)p[0] | ((size_t)p[1] << 8); p += 2; if ((size_t)(end - p) < xlen) return nullptr; p += xlen; }
        for (int bit = 8; bit <= 16; bit <<= 1)                       // FNAME, FCOMMENT: zero-terminated
            if (flg & bit) { const uint8_t *z = (const uint8_t *)memchr(p, 0, (size_t)(end - p)); if (!z) return nullptr; p = z + 1; }
        if (flg & 2) { if (end - p < 2) return nullptr; p += 2; }
        return p;
    }

    void run_inflate()
    {
        Inflater *inf = new Inflater();
        struct Del { Inflater *p; ~Del() { delete p; } } del{inf};
        const uint8_t *p = map, *end = map + map_len;
        size_t hist = 0;                         // valid history bytes in front of the current slot's data (same member)
        uint32_t isize = 0;
        bool in_member = false;
        for (;;) {
            if (!in_member) {
                while (p < end && *p == 0) p++;                       // (zero padding after a member is tolerated, like gzip does)
                if (p == end) break;
                const uint8_t *d = skip_header(p, end);
                if (!d) { set_error(p == map ? "not a gzip file" : "trailing garbage after the gzip stream"); return; }
                inf->reset(d, end);
                in_member = true; hist = 0; isize = 0;
            }
            size_t i;
            {
                std::unique_lock<std::mutex> lk(mu);
                cv.wait(lk, [&] { return stop || tail - (head < chead ? head : chead) < NBUF; });
                if (stop) return;
                i = tail;
            }
            uint8_t *base = data(i), *out = base;
            if (hist && i > 0) memcpy(base - hist, data(i - 1) + slot[(i - 1) % NBUF].len - hist, hist);
            if (!inf->run(out, base + BUF, base - hist)) { set_error(inf->err); return; }
            Slot &s = slot[i % NBUF];
            s.len = (size_t)(out - base);
            isize += (uint32_t)s.len;
            s.member_end = inf->state == Inflater::DONE;
            if (s.member_end) {
                p = inf->input_position();
                if (end - p < 8) { set_error("truncated gzip stream"); return; }
                s.crc_want = (uint32_t)p[0] | ((uint32_t)p[1] << 8) | ((uint32_t)p[2] << 16) | ((uint32_t)p[3] << 24);
                s.isize_want = (uint32_t)p[4] | ((uint32_t)p[5] << 8) | ((uint32_t)p[6] << 16) | ((uint32_t)p[7] << 24);
                if (s.isize_want != isize) { set_error("incorrect length check"); return; }
                p += 8;
                in_member = false;
            } else {
                hist = Inflater::WINDOW;                              // (a slot that does not end its member is full: BUF >= the window)
            }
            {
                std::lock_guard<std::mutex> lk(mu);
                tail++;
            }
            cv.notify_all();
        }
        if (in_member) { set_error("truncated gzip stream"); return; }
        std::lock_guard<std::mutex> lk(mu);
        produced_all = true;
        cv.notify_all();
    }

    void run_check()
    {
        uLong crc = crc32(0L, Z_NULL, 0);
        for (;;) {
            size_t i;
            {
                std::unique_lock<std::mutex> lk(mu);
                cv.wait(lk, [&] { return stop || chead < tail || produced_all; });
                if (stop) return;
                if (chead == tail) { checked_all = true; cv.notify_all(); return; }      // (produced_all, or an error: nothing more comes)
                i = chead;
            }
            Slot &s = slot[i % NBUF];
            size_t off = 0;
            while (off < s.len) { const size_t n = s.len - off < (1u << 30) ? s.len - off : (1u << 30); crc = crc32(crc, data(i) + off, (uInt)n); off += n; }
            bool bad = false;
            if (s.member_end) { bad = (uint32_t)crc != s.crc_want; crc = crc32(0L, Z_NULL, 0); }
            {
                std::lock_guard<std::mutex> lk(mu);
                if (bad && err.empty()) { err = "incorrect data check"; produced_all = true; }
                chead++;
            }
            cv.notify_all();
        }
    }

    // up to `cap` bytes; fewer only at the end of the stream (0 = end, reported once every member's CRC has been checked).
    // Returns 0 ok, 1 error (err)
    int read(uint8_t *dst, size_t cap, size_t *n_out)
    {
        size_t n = 0;
        while (n < cap) {
            std::unique_lock<std::mutex> lk(mu);
            cv.wait(lk, [&] { return head < tail || (produced_all && checked_all); });
            if (!err.empty()) return 1;
            if (head == tail) break;                                  // everything produced, checked and read
            const size_t avail = slot[head % NBUF].len - rpos, take = avail < cap - n ? avail : cap - n;
            const uint8_t *src = data(head) + rpos;
            lk.unlock();
            memcpy(dst + n, src, take);
            n += take;
            lk.lock();
            rpos += take;
            if (rpos == slot[head % NBUF].len) { rpos = 0; head++; lk.unlock(); cv.notify_all(); }
        }
        *n_out = n;
        return 0;
    }

    ~GzStream()
    {
        { std::lock_guard<std::mutex> lk(mu); stop = true; }
        cv.notify_all();
        if (th_inflate.joinable()) th_inflate.join();
        if (th_check.joinable()) th_check.join();
        if (map && map_len) munmap(const_cast<uint8_t *>(map), map_len);
        if (fd >= 0) close(fd);
    }
};

inline GzStream *gz_open(const char *path, const char **why)
{
    const int fd = open(path, O_RDONLY);
    if (fd < 0) { *why = "cannot open the file"; return nullptr; }
    struct stat st;
    if (fstat(fd, &st) != 0 || st.st_size <= 0) { close(fd); *why = "cannot stat the file (or it is empty)"; return nullptr; }
    void *m = mmap(nullptr, (size_t)st.st_size, PROT_READ, MAP_PRIVATE, fd, 0);
    if (m == MAP_FAILED) { close(fd); *why = "cannot map the file"; return nullptr; }
    (void)madvise(m, (size_t)st.st_size, MADV_SEQUENTIAL);
    GzStream *g = new GzStream();
    g->fd = fd; g->map = (const uint8_t *)m; g->map_len = (size_t)st.st_size;
    for (auto &sl : g->slot) sl.mem.resize(Inflater::WINDOW + GzStream::BUF + GzStream::SLACK);
    g->th_inflate = std::thread([g] { g->run_inflate(); });
    g->th_check = std::thread([g] { g->run_check(); });
    return g;
}

}  // namespace kdbhost
