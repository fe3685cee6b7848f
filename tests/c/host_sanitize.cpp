// Host-side code of libkdbhip (record splitter, .kdb row writer) under AddressSanitizer + UBSan (CPU build; the GPU
// pool has no sanitizer runs).  Random and malformed FASTQ/FASTA text with EXACTLY sized output buffers, so that any
// overrun is caught; the writer is run on random count vectors and its output re-read with zlib.
// Build + run: tests/test_host_sanitize.py
#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <random>
#include <string>
#include <vector>
#include <zlib.h>

#include "../../kmerdb_amd/csrc/kdb_hostparse.cpp.h"
#include "../../kmerdb_amd/csrc/kdb_kdbwriter.cpp.h"

static std::string random_text(std::mt19937_64 &g, bool fastq)
{
    static const char alpha[] = "ACGTNacgt\n\r @>+I!~\t";
    std::string s;
    const int nrec = (int)(g() % 20);
    for (int r = 0; r < nrec; r++) {
        const int mode = (int)(g() % 10);
        if (mode == 0) {                       // garbage
            const int n = (int)(g() % 60);
            for (int i = 0; i < n; i++) s.push_back(alpha[g() % (sizeof alpha - 1)]);
            continue;
        }
        const int L = (int)(g() % 80);
        std::string seq;
        for (int i = 0; i < L; i++) seq.push_back("ACGTN"[g() % 5]);
        if (fastq) {
            s += "@r" + std::to_string(r) + (g() % 2 ? " extra" : "") + (g() % 7 == 0 ? "\r\n" : "\n");
            s += seq + "\n+\n";
            std::string q((g() % 9 == 0) ? (size_t)(g() % 80) : (size_t)L, 'I');     // sometimes the wrong quality length
            s += q + (g() % 5 == 0 && r == nrec - 1 ? "" : "\n");
        } else {
            s += ">c" + std::to_string(r) + " desc\n";
            for (int i = 0; i < L; i += 17) s += seq.substr(i, 17) + (g() % 11 == 0 ? "\r\n" : "\n");
            if (g() % 6 == 0) s += "\n";
        }
    }
    if (g() % 4 == 0 && !s.empty()) s.resize(g() % s.size());                         // truncated file
    return s;
}

int main(int argc, char **argv)
{
    const int iters = argc > 1 ? atoi(argv[1]) : 20000;
    std::mt19937_64 g(12345);
    size_t ok = 0, bad = 0;
    for (int it = 0; it < iters; it++) {
        const bool fastq = g() % 2;
        const std::string t = random_text(g, fastq);
        const size_t n = t.size();
        // capacities as kmerdb_amd/reader.py computes them; allocated exactly so that ASan sees any overrun
        size_t cap_reads = fastq ? n / 6 + 2 : 1;
        if (!fastq) for (char c : t) cap_reads += (c == '>');
        std::vector<uint8_t> text(t.begin(), t.end());
        std::vector<uint8_t> bases(n ? n : 1);
        std::vector<uint64_t> offs(cap_reads + 1), hdr(2 * cap_reads);
        size_t nreads = 0, nbases = 0, consumed = 0;
        const char *why = "";
        int rc;
        if (fastq) rc = kdbhost::parse_fastq(text.data(), n, (int)(g() % 2), bases.data(), bases.size(), offs.data(), cap_reads, hdr.data(), &nreads, &nbases, &consumed, &why);
        else rc = kdbhost::parse_fasta(text.data(), n, bases.data(), bases.size(), offs.data(), cap_reads, hdr.data(), &nreads, &nbases, &why);
        if (rc == 0) {
            ok++;
            if (nreads > cap_reads || nbases > bases.size() || offs[nreads] != nbases) { fprintf(stderr, "inconsistent result\n"); return 1; }
            for (size_t r = 0; r < nreads; r++) if (offs[r] > offs[r + 1]) { fprintf(stderr, "offsets not monotone\n"); return 1; }
        } else bad++;
    }
    // FASTQ on several threads == on one: same records, same residues, same header spans, same bytes consumed -- or the same refusal.
    // Texts of a few hundred records cut into pieces of a few hundred bytes, with wrapped records, '@' quality lines, CR LF and garbage among them.
    for (int it = 0; it < iters / 8; it++) {
        std::string t;
        const int parts = 1 + (int)(g() % 6);
        for (int p = 0; p < parts; p++) {
            if (g() % 5 == 0) t += random_text(g, true);
            const int nrec = (int)(g() % 120);
            for (int r = 0; r < nrec; r++) {
                const int L = 1 + (int)(g() % 90);
                std::string seq, q;
                for (int i = 0; i < L; i++) { seq.push_back("ACGTN"[g() % 5]); q.push_back((char)(33 + g() % 94)); }
                if (g() % 3 == 0) q[0] = '@';
                const std::string nl = g() % 9 == 0 ? "\r\n" : "\n";
                t += "@read" + std::to_string(r) + nl;
                if (g() % 40 == 0) {                                     // a wrapped record
                    const int w = 1 + (int)(g() % 30);
                    for (int i = 0; i < L; i += w) t += seq.substr((size_t)i, (size_t)w) + nl;
                    t += (g() % 2 ? "+read" + std::to_string(r) : std::string("+")) + nl;
                    for (int i = 0; i < L; i += w) t += q.substr((size_t)i, (size_t)w) + nl;
                } else t += seq + nl + "+" + nl + q + nl;
            }
        }
        if (g() % 6 == 0 && !t.empty()) t.resize(g() % t.size());
        const size_t n = t.size(), cap_reads = n / 6 + 2;
        std::vector<uint8_t> text(t.begin(), t.end()), b1(n ? n : 1), b2(n ? n : 1);
        std::vector<uint64_t> o1(cap_reads + 1), o2(cap_reads + 1), h1(2 * cap_reads), h2(2 * cap_reads);
        size_t r1 = 0, r2 = 0, n1 = 0, n2 = 0, c1 = 0, c2 = 0;
        const char *w1 = "", *w2 = "";
        const int at_eof = (int)(g() % 2), threads = 2 + (int)(g() % 7);
        const int rc1 = kdbhost::parse_fastq(text.data(), n, at_eof, b1.data(), b1.size(), o1.data(), cap_reads, h1.data(), &r1, &n1, &c1, &w1);
        const int rc2 = kdbhost::parse_fastq_mt(text.data(), n, at_eof, b2.data(), b2.size(), o2.data(), cap_reads, h2.data(), &r2, &n2, &c2, &w2, threads, 64 + g() % 3000);
        if (rc1 != rc2) { fprintf(stderr, "fastq mt: rc %d vs %d (%s / %s)\n", rc1, rc2, w1, w2); return 1; }
        if (rc1 == 0 && (r1 != r2 || n1 != n2 || c1 != c2 || memcmp(b1.data(), b2.data(), n1) != 0 || memcmp(o1.data(), o2.data(), (r1 + 1) * 8) != 0 ||
                         memcmp(h1.data(), h2.data(), 2 * r1 * 8) != 0)) { fprintf(stderr, "fastq mt differs from the one-thread parse (%zu/%zu reads)\n", r1, r2); return 1; }
        if (rc1 == 0) ok++; else bad++;
    }
    // chunked FASTA parsing (streamed files): cutting the text anywhere and carrying the unconsumed rest must give the
    // residues of the whole-text parse, piece after piece
    for (int it = 0; it < iters / 4; it++) {
        const std::string t = random_text(g, false);
        const size_t n = t.size();
        size_t cap_reads = 2;
        for (char c : t) cap_reads += (c == '>');
        std::vector<uint8_t> text(t.begin(), t.end()), bases(n ? n : 1), bases2(n ? n : 1);
        std::vector<uint64_t> offs(cap_reads + 1), offs2(cap_reads + 1);
        size_t nreads = 0, nbases = 0;
        const char *why = "";
        if (kdbhost::parse_fasta(text.data(), n, bases.data(), bases.size(), offs.data(), cap_reads, nullptr, &nreads, &nbases, &why)) continue;
        std::string whole((const char *)bases.data(), nbases), pieces;
        size_t pos = 0;
        int in_record = 0;
        while (pos < n) {
            const size_t len = std::min<size_t>(n - pos, 1 + g() % 97);
            size_t nr = 0, nb = 0, consumed = 0;
            int in_out = 0;
            const int at_eof = pos + len == n;
            if (kdbhost::parse_fasta_chunk(text.data() + pos, len, at_eof, in_record, bases2.data(), bases2.size(), offs2.data(), cap_reads, nullptr, &nr, &nb,
                                           &consumed, &in_out, &why)) { fprintf(stderr, "chunk parse failed: %s\n", why); return 1; }
            if (nb > bases2.size() || nr > cap_reads || offs2[nr] != nb || consumed > len) { fprintf(stderr, "chunk parse inconsistent\n"); return 1; }
            pieces.append((const char *)bases2.data(), nb);
            in_record = in_out;
            if (consumed == 0 && !at_eof) {              // an incomplete header line longer than the piece: extend the piece
                size_t more = len;
                while (consumed == 0 && pos + more < n) {
                    more = std::min<size_t>(n - pos, more + 64);
                    if (kdbhost::parse_fasta_chunk(text.data() + pos, more, pos + more == n, in_record, bases2.data(), bases2.size(), offs2.data(), cap_reads,
                                                   nullptr, &nr, &nb, &consumed, &in_out, &why)) { fprintf(stderr, "chunk parse failed: %s\n", why); return 1; }
                }
                pieces.append((const char *)bases2.data(), nb);
                in_record = in_out;
                if (consumed == 0) break;
            }
            pos += consumed;
        }
        (void)0;
        if (pieces != whole) { fprintf(stderr, "chunked FASTA parse differs from the whole-text parse (%zu vs %zu residues)\n", pieces.size(), whole.size()); return 1; }
    }
    // BGZF inflate: members written here with zlib, whole and truncated input, 1..8 threads, exact output buffers; corrupt input must be refused
    for (int it = 0; it < 60; it++) {
        std::string plain;
        const size_t want = g() % 300000;
        while (plain.size() < want) plain += random_text(g, true);
        std::string comp;
        std::vector<size_t> ends;
        for (size_t at = 0; at <= plain.size(); at += 60000) {
            const size_t len = std::min<size_t>(60000, plain.size() - at);
            std::vector<uint8_t> buf(compressBound(len) + 64);
            z_stream zs; memset(&zs, 0, sizeof zs);
            deflateInit2(&zs, 6, Z_DEFLATED, -15, 8, Z_DEFAULT_STRATEGY);
            zs.next_in = (Bytef *)plain.data() + at; zs.avail_in = (uInt)len; zs.next_out = buf.data(); zs.avail_out = (uInt)buf.size();
            deflate(&zs, Z_FINISH);
            const size_t clen = zs.total_out;
            deflateEnd(&zs);
            const size_t bsize = clen + 25;
            const uint8_t hdr[18] = {0x1f, 0x8b, 8, 4, 0, 0, 0, 0, 0, 0xff, 6, 0, 'B', 'C', 2, 0, (uint8_t)(bsize & 0xff), (uint8_t)(bsize >> 8)};
            comp.append((const char *)hdr, 18);
            comp.append((const char *)buf.data(), clen);
            const uint32_t crc = (uint32_t)crc32(crc32(0L, Z_NULL, 0), (const Bytef *)plain.data() + at, (uInt)len), isz = (uint32_t)len;
            for (int b = 0; b < 4; b++) comp.push_back((char)(crc >> (8 * b)));
            for (int b = 0; b < 4; b++) comp.push_back((char)(isz >> (8 * b)));
            ends.push_back(comp.size());
            if (len < 60000) break;
        }
        const size_t cut = g() % 3 == 0 ? g() % (comp.size() + 1) : comp.size();        // truncated input: only whole members are used
        std::vector<uint8_t> src(comp.begin(), comp.begin() + cut), dst(plain.size() ? plain.size() : 1);
        size_t consumed = 0, produced = 0;
        const char *why = "";
        if (kdbhost::bgzf_inflate(src.data(), src.size(), dst.data(), plain.size(), 1 + (int)(g() % 8), &consumed, &produced, &why)) { fprintf(stderr, "bgzf: %s\n", why); return 1; }
        size_t whole_members = 0;
        for (size_t e : ends) if (e <= cut) whole_members = e;
        if (consumed != whole_members || produced > plain.size() || memcmp(dst.data(), plain.data(), produced) != 0) { fprintf(stderr, "bgzf inflate wrong\n"); return 1; }
        if (src.size() > 40) {
            src[20 + g() % (src.size() - 20)] ^= 0x55;
            (void)kdbhost::bgzf_inflate(src.data(), src.size(), dst.data(), plain.size(), 2, &consumed, &produced, &why);    // must not crash or overrun
        }
    }
    // writer: random vectors, 1..8 threads; the output must gunzip to the rows it was given
    for (int it = 0; it < 40; it++) {
        const uint64_t nb = 1ull << (2 * (1 + g() % 6));
        std::vector<uint64_t> counts(nb);
        uint64_t total = 0;
        for (auto &c : counts) { c = (g() % 3 == 0) ? 0 : g() % 100000; total += c; }
        if (total == 0) { counts[0] = 1; total = 1; }
        const char *path = "/tmp/kdb_sanitize_rows.bin";
        remove(path);
        uint64_t nblocks = 0;
        const char *why = "";
        if (kdbhost::write_kdb_rows(path, counts.data(), nb, total, 6, 1 + (int)(g() % 8), &nblocks, &why) != 0) { fprintf(stderr, "writer failed: %s\n", why); return 1; }
        gzFile f = gzopen(path, "rb");
        std::string all; char buf[65536]; int r;
        while ((r = gzread(f, buf, sizeof buf)) > 0) all.append(buf, (size_t)r);
        gzclose(f);
        size_t lines = 0; for (char c : all) lines += (c == '\n');
        if (lines != nb) { fprintf(stderr, "writer: %zu lines for %llu bins\n", lines, (unsigned long long)nb); return 1; }
    }
    printf("host sanitize ok: %zu parsed, %zu rejected, writer round trips ok\n", ok, bad);
    return 0;
}
