// Host-side code of libkdbhip (record splitter, .kdb row writer) under AddressSanitizer + UBSan (CPU build; the GPU
// pool has no sanitizer runs).  Random and malformed FASTQ/FASTA text with EXACTLY sized output buffers, so that any
// overrun is caught; the writer is run on random count vectors and its output re-read with zlib.
// Build + run: tests/test_host_sanitize.py
#include <cstdio>
#include <cstdlib>
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
