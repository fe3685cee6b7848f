// kdb_kdbwriter.cpp.h -- host-side .kdb row writer (no GPU): the loop kmerdb/__init__.py:1980-1990 +
// Bio.bgzf.BgzfWriter._write_block as native code.  Rows  "{i}\t{kmer_id}\t{count}\t{frequency}\n"  for
// i = kmer_id = 0..4^k-1, cut into exactly 65536-byte chunks, each written as one BGZF member (gzip header
// with the 'BC' extra field, raw deflate level 6, CRC32, ISIZE); the last chunk is partial; no EOF marker is
// written (the reference never calls close(): kmerdb/__init__.py:1995-1998).  frequency = count / total as
// float64, printed like Python's str(numpy.float64): shortest round-trip digits, fixed notation for
// 1e-4 <= |x| < 1e16, otherwise scientific with a two-digit exponent.
#pragma once
#include <zlib.h>

#include <charconv>
#include <cmath>
#include <cstdint>
#include <cstdio>
#include <cstring>
#include <string>
#include <thread>
#include <vector>

namespace kdbhost {

// Python repr(float) of x into buf; returns length
inline int py_float_repr(double x, char *buf)
{
    if (std::isnan(x)) { memcpy(buf, "nan", 3); return 3; }
    if (std::isinf(x)) { if (x < 0) { memcpy(buf, "-inf", 4); return 4; } memcpy(buf, "inf", 3); return 3; }
    char tmp[64];
    auto res = std::to_chars(tmp, tmp + sizeof tmp, x, std::chars_format::scientific);   // shortest round-trip digits
    *res.ptr = 0;
    // tmp = [-]d[.ddd]e[+-]XX
    char *p = tmp;
    int n = 0;
    if (*p == '-') { buf[n++] = '-'; p++; }
    char digits[32];
    int nd = 0;
    digits[nd++] = *p++;
    if (*p == '.') { p++; while (*p != 'e') digits[nd++] = *p++; }
    p++;                                   // 'e'
    const int exp10 = atoi(p);
    if (exp10 >= -4 && exp10 < 16) {       // fixed notation
        if (exp10 >= 0) {
            for (int i = 0; i <= exp10; i++) buf[n++] = i < nd ? digits[i] : '0';
            buf[n++] = '.';
            if (nd > exp10 + 1) for (int i = exp10 + 1; i < nd; i++) buf[n++] = digits[i];
            else buf[n++] = '0';
        } else {
            buf[n++] = '0'; buf[n++] = '.';
            for (int i = 0; i < -exp10 - 1; i++) buf[n++] = '0';
            for (int i = 0; i < nd; i++) buf[n++] = digits[i];
        }
    } else {                               // scientific: d[.ddd]e[+-]XX, at least two exponent digits
        buf[n++] = digits[0];
        if (nd > 1) { buf[n++] = '.'; for (int i = 1; i < nd; i++) buf[n++] = digits[i]; }
        buf[n++] = 'e';
        buf[n++] = exp10 < 0 ? '-' : '+';
        const int a = exp10 < 0 ? -exp10 : exp10;
        if (a >= 100) { buf[n++] = (char)('0' + a / 100); buf[n++] = (char)('0' + (a / 10) % 10); buf[n++] = (char)('0' + a % 10); }
        else { buf[n++] = (char)('0' + a / 10); buf[n++] = (char)('0' + a % 10); }
    }
    return n;
}

inline int fmt_u64(uint64_t v, char *buf)
{
    char t[24];
    int n = 0;
    do { t[n++] = (char)('0' + v % 10); v /= 10; } while (v);
    for (int i = 0; i < n; i++) buf[i] = t[n - 1 - i];
    return n;
}

// rows [r0, r1) appended to out
inline void format_rows(const uint64_t *counts, uint64_t r0, uint64_t r1, double total, std::string &out)
{
    char line[128];
    for (uint64_t i = r0; i < r1; i++) {
        int n = fmt_u64(i, line);
        line[n++] = '\t';
        n += fmt_u64(i, line + n);
        line[n++] = '\t';
        n += fmt_u64(counts[i], line + n);
        line[n++] = '\t';
        n += py_float_repr((double)counts[i] / total, line + n);
        line[n++] = '\n';
        out.append(line, (size_t)n);
    }
}

// one BGZF member for data[0, len), len <= 65536 (Bio.bgzf._write_block)
inline bool bgzf_block(const uint8_t *data, size_t len, int level, std::vector<uint8_t> &out)
{
    uint8_t comp[70000];
    size_t clen = 0;
    for (int attempt = 0; attempt < 2; attempt++) {
        z_stream zs;
        memset(&zs, 0, sizeof zs);
        if (deflateInit2(&zs, attempt == 0 ? level : 0, Z_DEFLATED, -15, 8, Z_DEFAULT_STRATEGY) != Z_OK) return false;
        zs.next_in = const_cast<Bytef *>(data); zs.avail_in = (uInt)len;
        zs.next_out = comp; zs.avail_out = sizeof comp;
        const int rc = deflate(&zs, Z_FINISH);
        clen = sizeof comp - zs.avail_out;
        deflateEnd(&zs);
        if (rc == Z_STREAM_END && clen < 65536 - 26) break;
        if (attempt == 1) return false;
    }
    const uint32_t crc = (uint32_t)crc32(crc32(0L, Z_NULL, 0), data, (uInt)len);
    const uint16_t bsize = (uint16_t)(clen + 25);
    static const uint8_t hdr[16] = {0x1f, 0x8b, 0x08, 0x04, 0, 0, 0, 0, 0, 0xff, 0x06, 0x00, 'B', 'C', 0x02, 0x00};
    out.insert(out.end(), hdr, hdr + 16);
    out.push_back((uint8_t)(bsize & 0xff)); out.push_back((uint8_t)(bsize >> 8));
    out.insert(out.end(), comp, comp + clen);
    for (int i = 0; i < 4; i++) out.push_back((uint8_t)(crc >> (8 * i)));
    const uint32_t isize = (uint32_t)len;
    for (int i = 0; i < 4; i++) out.push_back((uint8_t)(isize >> (8 * i)));
    return true;
}

// append the 4^k rows to `path` (which already holds the header member(s)); returns 0 ok
inline int write_kdb_rows(const char *path, const uint64_t *counts, uint64_t nbins, uint64_t total_kmers, int level, int nthreads,
                          uint64_t *nblocks_out, const char **why)
{
    FILE *f = fopen(path, "ab");
    if (!f) { *why = "cannot open output file for append"; return 1; }
    if (nthreads < 1) nthreads = 1;
    const double total = (double)total_kmers;
    const uint64_t WINDOW = 1ull << 22;               // rows formatted per round (bounded memory at large k)
    std::string carry;
    uint64_t nblocks = 0;
    bool ok = true;
    for (uint64_t w0 = 0; w0 < nbins && ok; w0 += WINDOW) {
        const uint64_t w1 = w0 + WINDOW < nbins ? w0 + WINDOW : nbins;
        std::vector<std::string> parts((size_t)nthreads);
        {
            std::vector<std::thread> th;
            for (int t = 0; t < nthreads; t++) {
                const uint64_t a = w0 + (w1 - w0) * (uint64_t)t / (uint64_t)nthreads, b = w0 + (w1 - w0) * (uint64_t)(t + 1) / (uint64_t)nthreads;
                th.emplace_back([&, t, a, b] { parts[(size_t)t].reserve((size_t)(b - a) * 40); format_rows(counts, a, b, total, parts[(size_t)t]); });
            }
            for (auto &x : th) x.join();
        }
        std::string text = std::move(carry);
        size_t tot = text.size();
        for (auto &p : parts) tot += p.size();
        text.reserve(tot);
        for (auto &p : parts) { text += p; std::string().swap(p); }
        const bool last = (w1 == nbins);
        const size_t nfull = text.size() / 65536;
        const size_t nblk = nfull + ((last && text.size() % 65536) ? 1 : 0);
        std::vector<std::vector<uint8_t>> outs(nblk);
        std::vector<char> good(nblk, 1);
        {
            std::vector<std::thread> th;
            for (int t = 0; t < nthreads; t++)
                th.emplace_back([&, t] {
                    for (size_t b = (size_t)t; b < nblk; b += (size_t)nthreads) {
                        const size_t off = b * 65536, len = (off + 65536 <= text.size()) ? 65536 : text.size() - off;
                        outs[b].reserve(len / 3 + 64);
                        if (!bgzf_block((const uint8_t *)text.data() + off, len, level, outs[b])) good[b] = 0;
                    }
                });
            for (auto &x : th) x.join();
        }
        for (size_t b = 0; b < nblk; b++) {
            if (!good[b]) { *why = "deflate failed"; ok = false; break; }
            if (fwrite(outs[b].data(), 1, outs[b].size(), f) != outs[b].size()) { *why = "short write"; ok = false; break; }
        }
        nblocks += nblk;
        if (!last) carry.assign(text, nfull * 65536, std::string::npos);
    }
    if (fclose(f) != 0 && ok) { *why = "close failed"; ok = false; }
    if (nblocks_out) *nblocks_out = nblocks;
    return ok ? 0 : 1;
}

}  // namespace kdbhost
