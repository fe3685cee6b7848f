// writer_check.cpp -- the .kdb row writer (kdb_kdbwriter.cpp.h) on its own: every file it writes must gunzip (zlib) to exactly
// the rows format_rows() states, in members of 65536 bytes, for both encoders, any thread count, any count distribution.
//   writer_check check [iters]            random vectors (k = 1..9), all checked byte for byte
//   writer_check bench K THREADS [enc]    time the k = K vector (Poisson-like counts) into /dev/shm; prints rows/s, MB/s, ratio
#include <zlib.h>

#include <chrono>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <random>
#include <atomic>
#include <string>
#include <thread>
#include <vector>

#include "../../kmerdb_amd/csrc/kdb_kdbwriter.cpp.h"

static bool members_ok(const char *path, uint64_t want_members, uint64_t text_len)
{
    FILE *f = fopen(path, "rb");
    if (!f) return false;
    std::vector<uint8_t> raw;
    uint8_t buf[1 << 16];
    size_t r;
    while ((r = fread(buf, 1, sizeof buf, f)) > 0) raw.insert(raw.end(), buf, buf + r);
    fclose(f);
    size_t pos = 0;
    uint64_t n = 0, total = 0;
    while (pos < raw.size()) {
        if (pos + 18 > raw.size() || raw[pos] != 0x1f || raw[pos + 1] != 0x8b || raw[pos + 12] != 'B' || raw[pos + 13] != 'C') return false;
        const size_t bsize = (size_t)(raw[pos + 16] | (raw[pos + 17] << 8)) + 1;
        if (pos + bsize > raw.size()) return false;
        const uint32_t isize = (uint32_t)raw[pos + bsize - 4] | ((uint32_t)raw[pos + bsize - 3] << 8) | ((uint32_t)raw[pos + bsize - 2] << 16) | ((uint32_t)raw[pos + bsize - 1] << 24);
        n++;
        total += isize;
        pos += bsize;
        if (pos < raw.size() && isize != 65536) return false;          // every member but the last is full
    }
    return n == want_members && total == text_len;
}

static std::string gunzip_all(const char *path)
{
    gzFile f = gzopen(path, "rb");
    std::string all;
    static char buf[1 << 20];
    int r;
    while ((r = gzread(f, buf, sizeof buf)) > 0) all.append(buf, (size_t)r);
    gzclose(f);
    return all;
}

int main(int argc, char **argv)
{
    const std::string mode = argc > 1 ? argv[1] : "check";
    if (mode == "bench") {
        const int k = argc > 2 ? atoi(argv[2]) : 12;
        const int threads = argc > 3 ? atoi(argv[3]) : 8;
        const int enc = argc > 4 ? atoi(argv[4]) : 0;
        const double mean = argc > 5 ? atof(argv[5]) : 83.0;
        const uint64_t nb = 1ull << (2 * k);
        std::vector<uint64_t> counts(nb);
        std::mt19937_64 g(7);
        std::poisson_distribution<int> pd(mean);
        uint64_t total = 0;
        for (uint64_t i = 0; i < nb; i++) { counts[i] = (i & 1) ? 0 : (uint64_t)pd(g); total += counts[i]; }      // (half the bins empty, like a canonical vector)
        const char *path = "/dev/shm/kdb_writer_bench.bin";
        for (int rep = 0; rep < 3; rep++) {
            remove(path);
            fclose(fopen(path, "wb"));
            uint64_t nblocks = 0;
            const char *why = "";
            const auto t0 = std::chrono::steady_clock::now();
            if (kdbhost::write_kdb_rows(path, counts.data(), nb, total, 6, threads, &nblocks, &why, enc)) { fprintf(stderr, "failed: %s\n", why); return 1; }
            const double dt = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
            FILE *f = fopen(path, "rb"); fseek(f, 0, SEEK_END); const long sz = ftell(f); fclose(f);
            printf("k=%d enc=%s threads=%d: %.3f s, %.1f M rows/s, text %.1f MB at %.1f MB/s, file %.1f MB (ratio %.2f)\n", k, enc ? "zlib" : "rows", threads, dt,
                   (double)nb / dt / 1e6, (double)nblocks * 65536 / 1e6, (double)nblocks * 65536 / 1e6 / dt, (double)sz / 1e6, (double)nblocks * 65536 / (double)sz);
        }
        if (k <= 12) {
            std::string want;
            kdbhost::format_rows(counts.data(), 0, nb, (double)total, want);
            if (gunzip_all(path) != want) { fprintf(stderr, "bench output differs from the rows\n"); return 1; }
            printf("verified\n");
        }
        remove(path);
        return 0;
    }
    const int iters = argc > 2 ? atoi(argv[2]) : 60;
    std::mt19937_64 g(20240612);
    size_t checked = 0;
    for (int it = 0; it < iters; it++) {
        const int k = 1 + (int)(g() % 9);
        const uint64_t nb = 1ull << (2 * k);
        std::vector<uint64_t> counts(nb);
        const int shape = (int)(g() % 6);
        uint64_t total = 0;
        for (uint64_t i = 0; i < nb; i++) {
            uint64_t c;
            switch (shape) {
            case 0: c = g() % 3 == 0 ? 0 : g() % 200; break;                        // small counts, many repeats
            case 1: c = g() % 100000; break;                                        // beyond the 65536-entry table
            case 2: c = (g() % 5 == 0) ? (g() >> (g() % 40)) : g() % 50; break;     // a few huge ones (20-digit counts)
            case 3: c = 0; break;                                                   // all nullomers
            case 4: c = 1000000 + g() % 1000; break;                                // every count above the table
            default: c = i % 7; break;
            }
            counts[i] = c;
            total += c;
        }
        if (total == 0) { counts[nb - 1] = 1; total = 1; }
        std::string want;
        kdbhost::format_rows(counts.data(), 0, nb, (double)total, want);
        for (int enc = 0; enc < 2; enc++) {
            const char *path = "/tmp/kdb_writer_check.bin";
            remove(path);
            fclose(fopen(path, "wb"));
            uint64_t nblocks = 0;
            const char *why = "";
            const int threads = 1 + (int)(g() % 9);
            if (kdbhost::write_kdb_rows(path, counts.data(), nb, total, 1 + (int)(g() % 9), threads, &nblocks, &why, enc)) { fprintf(stderr, "writer failed: %s\n", why); return 1; }
            if (!members_ok(path, (want.size() + 65535) / 65536, want.size()) || nblocks != (want.size() + 65535) / 65536) {
                fprintf(stderr, "it %d k %d shape %d enc %d threads %d: member structure wrong\n", it, k, shape, enc, threads);
                return 1;
            }
            if (gunzip_all(path) != want) { fprintf(stderr, "it %d k %d shape %d enc %d threads %d: text differs\n", it, k, shape, enc, threads); return 1; }
            {
                // ... and back: the native reader (members inflated and parsed in groups, on several threads) returns the vector.  A header
                // member with the delimiter line goes in front, as in a real file.
                const char *rp = "/tmp/kdb_writer_check_r.bin";
                FILE *hf = fopen(rp, "wb");
                std::vector<uint8_t> hm;
                const std::string header = std::string("k: ") + std::to_string(k) + "\n\n========================\n";
                kdbhost::bgzf_block((const uint8_t *)header.data(), header.size(), 6, hm);
                fwrite(hm.data(), 1, hm.size(), hf);
                FILE *rf = fopen(path, "rb");
                static char cb[1 << 16];
                size_t got;
                while ((got = fread(cb, 1, sizeof cb, rf)) > 0) fwrite(cb, 1, got, hf);
                fclose(rf); fclose(hf);
                std::vector<uint64_t> ids(nb, 0), back(nb, 0);
                std::vector<double> fr(nb, 0.0);
                uint64_t nrows = 0;
                const char *rwhy = "";
                const int rrc = kdbhost::read_kdb_rows(rp, nb, ids.data(), back.data(), fr.data(), 1 + (int)(g() % 6), &nrows, &rwhy);
                if (rrc != 0 || nrows != nb || back != counts) { fprintf(stderr, "it %d k %d shape %d enc %d: reader rc %d (%s), %llu rows\n", it, k, shape, enc, rrc, rwhy, (unsigned long long)nrows); return 1; }
                for (uint64_t i = 0; i < nb; i++)
                    if (ids[i] != i || fr[i] != (double)counts[i] / (double)total) { fprintf(stderr, "it %d: reader row %llu wrong\n", it, (unsigned long long)i); return 1; }
                remove(rp);
            }
            remove(path);
            checked++;
        }
    }
    // the counts arrive while the writer runs (a device-to-host copy in pieces): a producer thread fills the vector front to back and
    // moves the watermark; the file must be the one the finished vector gives.  And a producer that gives up (~0) ends the job with an error.
    for (int it = 0; it < 6; it++) {
        const int k = 7 + it % 3;
        const uint64_t nb = 1ull << (2 * k);
        std::vector<uint64_t> src(nb), dst(nb, 0xDEADBEEFDEADBEEFull);
        uint64_t total = 0;
        for (auto &c : src) { c = g() % 3 == 0 ? 0 : g() % 70000; total += c; }
        std::string want;
        kdbhost::format_rows(src.data(), 0, nb, (double)total, want);
        std::atomic<uint64_t> ready{0};
        const bool give_up = it == 5;
        const uint64_t piece = 1 + g() % 5000;
        std::thread producer([&] {
            for (uint64_t at = 0; at < nb; at += piece) {
                const uint64_t n = std::min(piece, nb - at);
                memcpy(dst.data() + at, src.data() + at, n * 8);
                ready.store(at + n, std::memory_order_release);
                if (give_up && at > nb / 2) { ready.store(~0ull); return; }
                if ((at / piece) % 7 == 0) std::this_thread::sleep_for(std::chrono::microseconds(200));
            }
        });
        const char *path = "/tmp/kdb_writer_check.bin";
        remove(path);
        fclose(fopen(path, "wb"));
        uint64_t nblocks = 0;
        const char *why = "";
        const int rc = kdbhost::write_kdb_rows(path, dst.data(), nb, total, 6, 1 + (int)(g() % 6), &nblocks, &why, 0, &ready);
        producer.join();
        if (give_up) { if (rc == 0) { fprintf(stderr, "a producer that gave up did not fail the job\n"); return 1; } continue; }
        if (rc != 0 || gunzip_all(path) != want) { fprintf(stderr, "streamed counts: rc %d (%s) or text differs\n", rc, why); return 1; }
        checked++;
    }
    // the CRC against zlib's, odd lengths and alignments
    for (int it = 0; it < 2000; it++) {
        const size_t n = g() % 5000, off = g() % 9;
        std::vector<uint8_t> v(n + off);
        for (auto &b : v) b = (uint8_t)g();
        if (kdbhost::crc32_bytes(v.data() + off, n) != (uint32_t)crc32(crc32(0L, Z_NULL, 0), v.data() + off, (uInt)n)) { fprintf(stderr, "crc differs\n"); return 1; }
    }
    printf("writer check ok: %zu files\n", checked);
    return 0;
}
