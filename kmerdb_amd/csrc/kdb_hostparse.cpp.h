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

#include <memory>

#include "kdb_crc32.cpp.h"
#include "kdb_inflate.cpp.h"

namespace kdbhost {

// every byte of q[0, n) a printable quality character, 33 <= c <= 126 (Bio.SeqIO's "fastq" iterator raises on anything else)
inline bool fastq_quality_ok(const uint8_t *q, size_t n)
{
    const uint64_t ones = ~0ull / 255u;
    size_t i = 0;
    for (; i + 8 <= n; i += 8) {
        uint64_t x;
        memcpy(&x, q + i, 8);
        const uint64_t hi = x & (ones * 0x80u);                              // > 127
        const uint64_t lt33 = (x - ones * 33u) & ~x & (ones * 0x80u);        // < 33 (exact where no byte has bit 7 set)
        const uint64_t y = x ^ (ones * 0x7Fu);
        const uint64_t is127 = (y - ones) & ~y & (ones * 0x80u);             // == 127
        if (hi | lt33 | is127) return false;
    }
    for (; i < n; i++) if (q[i] < 33 || q[i] > 126) return false;
    return true;
}

// The general FASTQ grammar of Bio.SeqIO.QualityIO.FastqGeneralIterator (what kmerdb/parse.py:70-72 reads with): a title line '@...',
// sequence lines up to a line that starts with '+' (which may repeat the title), then quality lines until they hold as many
// characters as the sequence -- a quality line may itself start with '@'.  Wrapped (multi-line) records are rare; parse_fastq
// tries the four-line form first and comes here when a file is not of that form.  Same outputs and return codes.
inline int parse_fastq_general(const uint8_t *text, size_t n, int at_eof, uint8_t *bases, size_t bases_cap, uint64_t *offs, size_t cap_reads,
                               uint64_t *hdr, size_t *nreads_out, size_t *nbases_out, size_t *consumed_out, const char **why)
{
    size_t pos = 0, nreads = 0, nb = 0;
    offs[0] = 0;
    auto line_at = [&](size_t at, size_t *len, size_t *next) -> int {      // 1: a complete line (or the last one at EOF); 0: none yet
        if (at >= n) return 0;
        const uint8_t *e = (const uint8_t *)memchr(text + at, '\n', n - at);
        if (!e) { if (!at_eof) return 0; *len = n - at; *next = n; }
        else { *len = (size_t)(e - (text + at)); *next = at + *len + 1; }
        while (*len && (text[at + *len - 1] == '\r' || text[at + *len - 1] == ' ' || text[at + *len - 1] == '\t')) (*len)--;     // rstrip()
        return 1;
    };
    while (pos < n) {
        if (text[pos] == '\n' || text[pos] == '\r') { pos++; continue; }
        size_t tlen, at;
        if (!line_at(pos, &tlen, &at)) break;
        if (tlen == 0) { pos = at; continue; }                               // a line of blanks
        if (text[pos] != '@') { *why = "FASTQ record does not start with '@'"; return 1; }
        const size_t title = pos, nb0 = nb;
        size_t slen = 0;
        bool plus = false, partial = false;
        for (;;) {                                                           // sequence lines
            size_t len, next;
            if (!line_at(at, &len, &next)) { partial = true; break; }
            if (text[at] == '+') {
                if (len > 1 && (len - 1 != tlen - 1 || memcmp(text + at + 1, text + title + 1, len - 1) != 0)) { *why = "FASTQ sequence and quality captions differ"; return 1; }
                at = next; plus = true; break;
            }
            if (memchr(text + at, ' ', len) || memchr(text + at, '\t', len)) { *why = "whitespace is not allowed in a FASTQ sequence"; return 1; }
            if (nb + len > bases_cap) { *why = "output capacity exceeded"; return 2; }
            memcpy(bases + nb, text + at, len);
            nb += len; slen += len;
            at = next;
        }
        size_t qlen = 0;
        bool first_q = true;
        while (plus && !partial) {                                           // quality lines
            size_t len, next;
            if (!line_at(at, &len, &next)) { if (first_q || qlen < slen) partial = true; break; }
            if (!first_q && text[at] == '@' && qlen >= slen) break;          // the next record's title
            if (!first_q && len == 0 && qlen >= slen) { at = next; continue; }      // blank lines behind a record
            if (!fastq_quality_ok(text + at, len)) { *why = "invalid character in a FASTQ quality string"; return 1; }
            qlen += len;
            first_q = false;
            at = next;
            if (qlen > slen) break;
        }
        if (partial) {
            nb = nb0;
            if (at_eof) { *why = plus ? "end of file inside a FASTQ quality string" : "end of file without FASTQ quality information"; return 1; }
            break;                                                           // the rest of this record comes with the next chunk
        }
        if (qlen != slen) { *why = "FASTQ sequence and quality lengths differ"; return 1; }
        if (nreads >= cap_reads) { *why = "output capacity exceeded"; return 2; }
        if (hdr) { hdr[2 * nreads] = (uint64_t)title; hdr[2 * nreads + 1] = (uint64_t)(title + tlen); }
        offs[++nreads] = nb;
        pos = at;
    }
    if (at_eof) pos = n;
    *nreads_out = nreads; *nbases_out = nb; *consumed_out = pos;
    return 0;
}

// The usual FASTQ: four lines per record.  returns 0 ok; 1 malformed; 2 output capacity exceeded; 3 (strict only) not of the
// four-line form -- the caller goes to parse_fastq_general.  COUNT_ONLY: nothing is written (pass 1 of parse_fastq_mt).
template <bool COUNT_ONLY>
inline int parse_fastq_strict(const uint8_t *text, size_t n, int at_eof, uint8_t *bases, size_t bases_cap, uint64_t *offs, size_t cap_reads,
                              uint64_t *hdr /* optional [2*cap_reads]: header line start/end */, uint64_t hdr_base, uint64_t offs_base,
                              size_t *nreads_out, size_t *nbases_out, size_t *consumed_out, const char **why)
{
    size_t pos = 0, nreads = 0, nb = 0;
    if (!COUNT_ONLY) offs[0] = offs_base;
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
        if (*l0 != '@') return 3;
        if (l2 >= text + n || *l2 != '+') return 3;
        size_t slen = (size_t)(e1 - l1), qlen = (size_t)(e3 - l3);
        if (slen && l1[slen - 1] == '\r') slen--;
        if (qlen && l3[qlen - 1] == '\r') qlen--;
        if (slen != qlen) return 3;
        // what the next record must look like for this one to be a whole four-line record: a quality line of a wrapped record can
        // have the length of its first sequence line -- the general grammar decides then
        if (e3 + 1 < text + n && e3[1] != '@' && e3[1] != '\n' && e3[1] != '\r') return 3;
        if (!COUNT_ONLY) {
            size_t p2 = (size_t)(e2 - l2);
            if (p2 && l2[p2 - 1] == '\r') p2--;
            if (p2 > 1) return 3;                                  // a '+' line that repeats the title: checked by the general grammar
            if (!fastq_quality_ok(l3, qlen)) { *why = "invalid character in a FASTQ quality string"; return 1; }
            if (nreads >= cap_reads || nb + slen > bases_cap) { *why = "output capacity exceeded"; return 2; }
            memcpy(bases + nb, l1, slen);
            if (hdr) { size_t hl = (size_t)(e0 - l0); if (hl && l0[hl - 1] == '\r') hl--; hdr[2 * nreads] = hdr_base + (uint64_t)(l0 - text); hdr[2 * nreads + 1] = hdr_base + (uint64_t)(l0 - text) + hl; }
            offs[nreads + 1] = offs_base + nb + slen;
        }
        nb += slen;
        nreads++;
        pos = (size_t)(e3 - text) + (e3 < text + n ? 1 : 0);
    }
    if (at_eof) {
        // anything left must be blank
        for (size_t i = pos; i < n; i++)
            if (text[i] != '\n' && text[i] != '\r' && text[i] != ' ') return 3;
        pos = n;
    }
    *nreads_out = nreads; *nbases_out = nb; *consumed_out = pos;
    return 0;
}

// returns 0 ok; 1 malformed; 2 output capacity exceeded
inline int parse_fastq(const uint8_t *text, size_t n, int at_eof, uint8_t *bases, size_t bases_cap, uint64_t *offs, size_t cap_reads,
                       uint64_t *hdr /* optional [2*cap_reads]: header line start/end */, size_t *nreads_out, size_t *nbases_out,
                       size_t *consumed_out, const char **why)
{
    const int rc = parse_fastq_strict<false>(text, n, at_eof, bases, bases_cap, offs, cap_reads, hdr, 0, 0, nreads_out, nbases_out, consumed_out, why);
    if (rc != 3) return rc;
    return parse_fastq_general(text, n, at_eof, bases, bases_cap, offs, cap_reads, hdr, nreads_out, nbases_out, consumed_out, why);
}

// the first four-line record that starts at or behind `from`: a line that starts with '@' whose next line but one starts with '+'
// (a quality line may start with '@' too, but the line two below it is then a sequence line); n if there is none
inline size_t fastq_resync(const uint8_t *text, size_t n, size_t from)
{
    size_t at = from;
    if (at > 0) {
        const uint8_t *e = (const uint8_t *)memchr(text + at - 1, '\n', n - (at - 1));
        if (!e) return n;
        at = (size_t)(e - text) + 1;
    }
    while (at < n) {
        const uint8_t *e0 = (const uint8_t *)memchr(text + at, '\n', n - at);
        if (!e0) return n;
        if (text[at] == '@') {
            const uint8_t *l1 = e0 + 1;
            const uint8_t *e1 = l1 < text + n ? (const uint8_t *)memchr(l1, '\n', (size_t)(text + n - l1)) : nullptr;
            if (!e1) return n;
            if (e1 + 1 < text + n && e1[1] == '+') return at;
        }
        at = (size_t)(e0 - text) + 1;
    }
    return n;
}

// parse_fastq on `nthreads` threads: the text is cut at record starts (fastq_resync), every piece is counted (pass 1), then split into
// its final place behind the pieces before it (pass 2) -- one thread split 3.7 GB/s of text, less than sixteen inflating threads deliver.
// Any piece that is not plain four-line FASTQ sends the whole text through parse_fastq (same result, same errors, one thread).
inline int parse_fastq_mt(const uint8_t *text, size_t n, int at_eof, uint8_t *bases, size_t bases_cap, uint64_t *offs, size_t cap_reads,
                          uint64_t *hdr, size_t *nreads_out, size_t *nbases_out, size_t *consumed_out, const char **why, int nthreads,
                          size_t MIN_PIECE = 2u << 20 /* (tests: smaller) */)
{
    int T = nthreads;
    if ((size_t)T > n / MIN_PIECE) T = (int)(n / MIN_PIECE);
    if (T <= 1) return parse_fastq(text, n, at_eof, bases, bases_cap, offs, cap_reads, hdr, nreads_out, nbases_out, consumed_out, why);
    std::vector<size_t> cut((size_t)T + 1), nr((size_t)T), nbv((size_t)T), used((size_t)T);
    std::vector<int> rcs((size_t)T, 0);
    std::vector<const char *> whys((size_t)T, "");
    cut[0] = 0; cut[(size_t)T] = n;
    auto piece = [&](int i, bool count_only, size_t r0, size_t b0) {
        const uint8_t *p = text + cut[(size_t)i];
        const size_t len = cut[(size_t)i + 1] - cut[(size_t)i];
        const int eof = (i == T - 1) ? at_eof : 0;
        size_t a = 0, b = 0, c = 0;
        int rc;
        if (count_only) rc = parse_fastq_strict<true>(p, len, eof, nullptr, 0, nullptr, 0, nullptr, 0, 0, &a, &b, &c, &whys[(size_t)i]);
        else rc = parse_fastq_strict<false>(p, len, eof, bases + b0, bases_cap - b0, offs + r0, cap_reads - r0, hdr ? hdr + 2 * r0 : nullptr, (uint64_t)cut[(size_t)i], (uint64_t)b0,
                                            &a, &b, &c, &whys[(size_t)i]);
        if (rc == 0 && i < T - 1 && c != len) rc = 3;              // an inner piece ends where the next record starts: all of it must go
        rcs[(size_t)i] = rc; nr[(size_t)i] = a; nbv[(size_t)i] = b; used[(size_t)i] = c;
    };
    {
        std::vector<std::thread> th;
        for (int i = 1; i < T; i++) th.emplace_back([&, i] { cut[(size_t)i] = fastq_resync(text, n, n / (size_t)T * (size_t)i); });
        for (auto &x : th) x.join();
        for (int i = 1; i < T; i++) if (cut[(size_t)i] < cut[(size_t)i - 1]) cut[(size_t)i] = cut[(size_t)i - 1];
    }
    auto run_all = [&](bool count_only, const std::vector<size_t> &r0, const std::vector<size_t> &b0) {
        std::vector<std::thread> th;
        for (int i = 1; i < T; i++) th.emplace_back([&, i] { piece(i, count_only, r0[(size_t)i], b0[(size_t)i]); });
        piece(0, count_only, r0[0], b0[0]);
        for (auto &x : th) x.join();
    };
    std::vector<size_t> r0((size_t)T + 1, 0), b0((size_t)T + 1, 0);
    run_all(true, r0, b0);
    bool plain = true;
    for (int i = 0; i < T; i++) if (rcs[(size_t)i] != 0) plain = false;
    if (plain) {
        for (int i = 0; i < T; i++) { r0[(size_t)i + 1] = r0[(size_t)i] + nr[(size_t)i]; b0[(size_t)i + 1] = b0[(size_t)i] + nbv[(size_t)i]; }
        if (r0[(size_t)T] > cap_reads || b0[(size_t)T] > bases_cap) { *why = "output capacity exceeded"; return 2; }
        run_all(false, r0, b0);
        for (int i = 0; i < T; i++) {
            if (rcs[(size_t)i] == 1 || rcs[(size_t)i] == 2) { *why = whys[(size_t)i]; return rcs[(size_t)i]; }
            if (rcs[(size_t)i] != 0) plain = false;
        }
    }
    if (!plain) return parse_fastq(text, n, at_eof, bases, bases_cap, offs, cap_reads, hdr, nreads_out, nbases_out, consumed_out, why);
    *nreads_out = r0[(size_t)T]; *nbases_out = b0[(size_t)T]; *consumed_out = cut[(size_t)T - 1] + used[(size_t)T - 1];
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
    // A member is decoded by the engine's own DEFLATE decoder (kdb_inflate.cpp.h, 1.3-1.5 x zlib's inflate on FASTQ text) into a buffer
    // of the thread's own -- the decoder may write a few hundred bytes past a member's end, and the neighbouring members are being
    // written by other threads -- and copied to its place once its CRC-32 (carry-less multiplication: kdb_crc32.cpp.h) has been checked.
    auto work = [&] {
        std::unique_ptr<Inflater> inf(new Inflater());
        std::vector<uint8_t> tmp(65536 + 1024);
        for (;;) {
            const size_t i = next.fetch_add(1);
            if (i >= blocks.size() || bad.load()) return;
            const BgzfBlock &b = blocks[i];
            if (b.isize > 65536) { bad = 1; return; }
            const uint8_t *d0 = src + b.src + b.data_off, *d1 = src + b.src + b.csize - 8;
            inf->reset(d0, d1);
            uint8_t *out = tmp.data();
            bool ok = inf->run(out, tmp.data() + b.isize, tmp.data());
            // (an empty member -- bgzip's end-of-file marker -- is one empty final block: run() returns before it reads the header)
            if (ok && inf->state != Inflater::DONE) { uint8_t *o2 = out; ok = inf->run(o2, o2 + 1, tmp.data()) && o2 == out && inf->state == Inflater::DONE; }
            ok = ok && (size_t)(out - tmp.data()) == b.isize;
            const uint8_t *t = src + b.src + b.csize - 8;
            const uint32_t want = (uint32_t)t[0] | ((uint32_t)t[1] << 8) | ((uint32_t)t[2] << 16) | ((uint32_t)t[3] << 24);
            if (!ok || crc32_bytes(tmp.data(), b.isize) != want) { bad = 1; return; }
            memcpy(dst + b.dst, tmp.data(), b.isize);
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
