// kdb_kdbwriter.cpp.h -- host-side .kdb row writer (no GPU): the loop kmerdb/__init__.py:1980-1990 +
// Bio.bgzf.BgzfWriter._write_block as native code.  Rows  "{i}\t{kmer_id}\t{count}\t{frequency}\n"  for
// i = kmer_id = 0..4^k-1, cut into exactly 65536-byte chunks, each written as one BGZF member (gzip header
// with the 'BC' extra field, raw deflate, CRC32, ISIZE); the last chunk is partial; no EOF marker is
// written (the reference never calls close(): kmerdb/__init__.py:1995-1998).  frequency = count / total as
// float64, printed like Python's str(numpy.float64): shortest round-trip digits, fixed notation for
// 1e-4 <= |x| < 1e16, otherwise scientific with a two-digit exponent.
//
// Round 5: one pipeline, no window barriers, no concatenation, no pass over the whole vector first.
//   a thread takes the next chunk of rows and
//   (1)  adds up the chunk's text length (a row's length is 2 x digits(i) + 2 + the length of its "count \t frequency \n"
//        string, which depends on the count alone: a table for counts < 65536); the lengths are committed in chunk order ->
//        the text offset of the chunk's first row -> which 65536-byte members the chunk owns (those that START inside its text);
//   (2)  formats its rows (the id is a decimal string incremented in place, the count string a 32-byte copy out of the table) up to the
//        end of the last member it owns, deflating every member as it is formatted;
//   (3)  writes its own chunk (pwrite) once the file offset is known: the compressed sizes are committed in chunk order too.
//   The counts may still be arriving while this runs (rows_ready): a chunk waits for its own rows only.
// The deflate stream is made by a ROW-AWARE encoder (default; "zlib" = zlib at the level asked for, for comparison):
// LZ77 needs no search when the text's structure says where the repeats are -- the leading digits of an id repeat
// the row before, the second id column repeats the first, and a row's "count \t frequency \n" string repeats the
// last row with that count (a 65536-entry table of latest positions, every candidate checked byte for byte) --
// followed by one dynamic Huffman block per member -- a thread's first member with codes made from its own tokens (two passes),
// every later one in one pass with the codes of a member shortly before (StreamCoder).  The decompressed stream and the member boundaries are what the
// reference's file has; the compressed bytes differ from zlib's as they do between zlib versions.
#pragma once
#include <fcntl.h>
#include <unistd.h>
#include <zlib.h>

#include <algorithm>
#include <atomic>
#include <charconv>
#include <chrono>
#include <cmath>
#include <condition_variable>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <mutex>
#include <string>
#include <thread>
#include <vector>

#include <sys/mman.h>
#include <sys/stat.h>

#include <memory>

#include "kdb_crc32.cpp.h"
#include "kdb_hostparse.cpp.h"

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

inline int ndigits_u64(uint64_t v)
{
    int n = 1;
    while (v >= 10) { v /= 10; n++; }
    return n;
}

// "count \t frequency \n" into buf (at most 46 bytes); returns its length
inline int format_count_string(uint64_t count, double total, char *buf)
{
    int n = fmt_u64(count, buf);
    buf[n++] = '\t';
    n += py_float_repr((double)count / total, buf + n);
    buf[n++] = '\n';
    return n;
}

// rows [r0, r1) appended to out (the plain statement of the row format; the pipeline below writes the same bytes)
inline void format_rows(const uint64_t *counts, uint64_t r0, uint64_t r1, double total, std::string &out)
{
    char line[128];
    for (uint64_t i = r0; i < r1; i++) {
        int n = fmt_u64(i, line);
        line[n++] = '\t';
        n += fmt_u64(i, line + n);
        line[n++] = '\t';
        n += format_count_string(counts[i], total, line + n);
        out.append(line, (size_t)n);
    }
}

// ---- one BGZF member around a raw deflate payload ---------------------------------------------------------------
constexpr size_t BGZF_TEXT = 65536;            // uncompressed bytes per member (the reference's KDBWriter cuts there)
constexpr size_t BGZF_HEAD = 18, BGZF_TAIL = 8;
constexpr size_t BGZF_MAX_PAYLOAD = 65536 - 26;

inline void bgzf_frame(uint8_t *member, size_t clen, uint32_t crc, uint32_t isize)
{
    static const uint8_t hdr[16] = {0x1f, 0x8b, 0x08, 0x04, 0, 0, 0, 0, 0, 0xff, 0x06, 0x00, 'B', 'C', 0x02, 0x00};
    memcpy(member, hdr, 16);
    const uint16_t bsize = (uint16_t)(clen + 25);
    member[16] = (uint8_t)(bsize & 0xff); member[17] = (uint8_t)(bsize >> 8);
    uint8_t *t = member + BGZF_HEAD + clen;
    for (int i = 0; i < 4; i++) t[i] = (uint8_t)(crc >> (8 * i));
    for (int i = 0; i < 4; i++) t[4 + i] = (uint8_t)(isize >> (8 * i));
}

// zlib payload for data[0, len) into out (capacity 70000); 0 on failure (Bio.bgzf._write_block: level, then stored)
inline size_t zlib_payload(const uint8_t *data, size_t len, int level, uint8_t *out)
{
    for (int attempt = 0; attempt < 2; attempt++) {
        z_stream zs;
        memset(&zs, 0, sizeof zs);
        if (deflateInit2(&zs, attempt == 0 ? level : 0, Z_DEFLATED, -15, 8, Z_DEFAULT_STRATEGY) != Z_OK) return 0;
        zs.next_in = const_cast<Bytef *>(data); zs.avail_in = (uInt)len;
        zs.next_out = out; zs.avail_out = 70000;
        const int rc = deflate(&zs, Z_FINISH);
        const size_t clen = 70000 - zs.avail_out;
        deflateEnd(&zs);
        if (rc == Z_STREAM_END && clen < BGZF_MAX_PAYLOAD) return clen;
    }
    return 0;
}

// one BGZF member for data[0, len), len <= 65536, by zlib (kept for the .kdbg writer's callers and the tests)
inline bool bgzf_block(const uint8_t *data, size_t len, int level, std::vector<uint8_t> &out)
{
    uint8_t comp[70000 + 32];
    const size_t clen = zlib_payload(data, len, level, comp + BGZF_HEAD);
    if (!clen) return false;
    bgzf_frame(comp, clen, crc32_bytes(data, len), (uint32_t)len);
    out.insert(out.end(), comp, comp + BGZF_HEAD + clen + BGZF_TAIL);
    return true;
}

// ---- deflate: dynamic Huffman block over a token list -------------------------------------------------------------
struct DeflateTables {
    uint8_t len_sym[259];          // match length 3..258 -> litlen symbol - 257
    uint8_t len_xbits[29];
    uint16_t len_base[29];
    uint8_t dist_xbits[30];
    uint16_t dist_base[30];
    uint8_t dist_sym_lo[257];      // distance 1..256 -> symbol
    DeflateTables()
    {
        static const uint8_t lx[29] = {0, 0, 0, 0, 0, 0, 0, 0, 1, 1, 1, 1, 2, 2, 2, 2, 3, 3, 3, 3, 4, 4, 4, 4, 5, 5, 5, 5, 0};
        int b = 3;
        for (int s = 0; s < 29; s++) {
            len_xbits[s] = lx[s];
            len_base[s] = (uint16_t)(s == 28 ? 258 : b);
            if (s < 28) { for (int j = 0; j < (1 << lx[s]) && b + j <= 258; j++) len_sym[b + j] = (uint8_t)s; b += 1 << lx[s]; }
        }
        len_sym[258] = 28;
        int d = 1;
        for (int s = 0; s < 30; s++) {
            const int x = s < 2 ? 0 : (s >> 1) - 1;
            dist_xbits[s] = (uint8_t)x;
            dist_base[s] = (uint16_t)d;
            for (int j = 0; j < (1 << x); j++) if (d + j <= 256) dist_sym_lo[d + j] = (uint8_t)s;
            d += 1 << x;
        }
    }
    inline int dist_sym(uint32_t dist) const          // 1..32768
    {
        if (dist <= 256) return dist_sym_lo[dist];
        const uint32_t v = dist - 1;                  // >= 256: symbol = 2 * floor(log2 v) + the bit below the leading one
        const int hb = 31 - __builtin_clz(v);
        return 2 * hb + (int)((v >> (hb - 1)) & 1u);
    }
};
inline const DeflateTables &deflate_tables() { static const DeflateTables T; return T; }

// code lengths for freq[0, n), none longer than maxbits; at least two symbols get a code (inflaters want a complete or
// a two-leaf tree).  Two-queue Huffman over the symbols sorted by frequency; an over-long tree is flattened by moving
// codes between length classes until the Kraft sum is 1 again, and the sorted symbols are given the class sizes in order.
inline void huffman_lengths(const uint32_t *freq, int n, int maxbits, uint8_t *len)
{
    int order[288];
    int m = 0;
    for (int i = 0; i < n; i++) { len[i] = 0; if (freq[i]) order[m++] = i; }
    if (m == 0) { len[0] = 1; if (n > 1) len[1] = 1; return; }
    if (m == 1) { len[order[0]] = 1; len[order[0] == 0 ? 1 : 0] = 1; return; }
    std::sort(order, order + m, [&](int a, int b) { return freq[a] != freq[b] ? freq[a] < freq[b] : a < b; });
    uint64_t w[576];
    int parent[576];
    for (int i = 0; i < m; i++) w[i] = freq[order[i]];
    int leaf = 0, inner = m, next = m;                 // leaves [0, m) ascending, inner nodes [m, next) ascending
    while (next < 2 * m - 1) {
        int pick[2];
        for (int j = 0; j < 2; j++) {
            if (leaf < m && (inner >= next || w[leaf] <= w[inner])) pick[j] = leaf++;
            else pick[j] = inner++;
        }
        w[next] = w[pick[0]] + w[pick[1]];
        parent[pick[0]] = parent[pick[1]] = next;
        next++;
    }
    int depth[576];
    depth[2 * m - 2] = 0;
    for (int i = 2 * m - 3; i >= 0; i--) depth[i] = depth[parent[i]] + 1;
    int cnt[64] = {0};
    for (int i = 0; i < m; i++) cnt[depth[i] > 63 ? 63 : depth[i]]++;
    int over = 0;
    for (int b = maxbits + 1; b < 64; b++) { over += cnt[b]; cnt[b] = 0; }
    if (over) {
        cnt[maxbits] += over;
        uint64_t kraft = 0;                            // in units of 2^-maxbits
        for (int b = 1; b <= maxbits; b++) kraft += (uint64_t)cnt[b] << (maxbits - b);
        while (kraft > (1ull << maxbits)) {            // one code from the longest class pairs up with a shorter one moved down a level
            cnt[maxbits]--;
            for (int b = maxbits - 1; b >= 1; b--)
                if (cnt[b]) { cnt[b]--; cnt[b + 1] += 2; break; }
            kraft--;
        }
    }
    int at = 0;                                        // rarest symbols first: the longest codes
    for (int b = maxbits; b >= 1; b--)
        for (int j = 0; j < cnt[b]; j++) len[order[at++]] = (uint8_t)b;
}

// canonical codes of RFC 1951 3.2.2, bit-reversed (Huffman codes go into the stream starting with their first bit)
inline void huffman_codes(const uint8_t *len, int n, uint16_t *code)
{
    int cnt[16] = {0}, nextc[16];
    for (int i = 0; i < n; i++) cnt[len[i]]++;
    cnt[0] = 0;
    int c = 0;
    for (int b = 1; b < 16; b++) { c = (c + cnt[b - 1]) << 1; nextc[b] = c; }
    for (int i = 0; i < n; i++) {
        if (!len[i]) { code[i] = 0; continue; }
        unsigned v = (unsigned)nextc[len[i]]++, r = 0;
        for (int b = 0; b < len[i]; b++) { r = (r << 1) | (v & 1u); v >>= 1; }
        code[i] = (uint16_t)r;
    }
}

struct BitSink {
    uint8_t *p;
    uint64_t acc = 0;
    int n = 0;
    explicit BitSink(uint8_t *out) : p(out) {}
    inline void put(uint32_t bits, int len)            // len <= 32
    {
        acc |= (uint64_t)bits << n;
        n += len;
        if (n >= 32) { const uint32_t w = (uint32_t)acc; memcpy(p, &w, 4); p += 4; acc >>= 32; n -= 32; }
    }
    inline uint8_t *finish()
    {
        while (n > 0) { *p++ = (uint8_t)acc; acc >>= 8; n -= 8; }
        n = 0;
        return p;
    }
};

// tokens of one member: a literal byte, or 1 << 31 | distance symbol << 24 | length << 15 | distance - 1
struct MemberCoder {
    std::vector<uint32_t> toks;
    uint32_t ntok = 0;
    uint32_t lf[288], df[32];
    MemberCoder() : toks(BGZF_TEXT + 8) { begin(); }
    void begin() { ntok = 0; memset(lf, 0, sizeof lf); memset(df, 0, sizeof df); }
    inline void lit(uint8_t b) { toks[ntok++] = b; lf[b]++; }
    inline void lits(const char *s, int n) { for (int i = 0; i < n; i++) lit((uint8_t)s[i]); }
    inline void match(uint32_t len, uint32_t dist)
    {
        const DeflateTables &T = deflate_tables();
        const uint32_t ds = (uint32_t)T.dist_sym(dist);
        toks[ntok++] = 0x80000000u | (ds << 24) | (len << 15) | (dist - 1);
        lf[257 + T.len_sym[len]]++;
        df[ds]++;
    }

    // one final dynamic-Huffman block holding the tokens; `out` has room for 2 x 65536 bytes; returns the payload's length
    size_t finish(uint8_t *out)
    {
        const DeflateTables &T = deflate_tables();
        lf[256]++;                                     // end of block
        uint8_t ll[288], dl[32];
        uint16_t lc[288], dc[32];
        huffman_lengths(lf, 286, 15, ll);
        huffman_lengths(df, 30, 15, dl);
        huffman_codes(ll, 286, lc);
        huffman_codes(dl, 30, dc);
        int hlit = 286, hdist = 30;
        while (hlit > 257 && !ll[hlit - 1]) hlit--;
        while (hdist > 1 && !dl[hdist - 1]) hdist--;
        // the two length arrays as one run-length coded sequence of code-length symbols
        uint8_t seq[320];
        int ns = 0;
        for (int i = 0; i < hlit; i++) seq[ns++] = ll[i];
        for (int i = 0; i < hdist; i++) seq[ns++] = dl[i];
        uint8_t cls[320], clx[320];
        int ncl = 0;
        uint32_t cf[19] = {0};
        for (int i = 0; i < ns;) {
            int run = 1;
            while (i + run < ns && seq[i + run] == seq[i]) run++;
            const int v = seq[i];
            int left = run;
            if (v == 0) {
                while (left >= 11) { const int r = left > 138 ? 138 : left; cls[ncl] = 18; clx[ncl++] = (uint8_t)(r - 11); cf[18]++; left -= r; }
                if (left >= 3) { cls[ncl] = 17; clx[ncl++] = (uint8_t)(left - 3); cf[17]++; left = 0; }
                while (left-- > 0) { cls[ncl] = 0; clx[ncl++] = 0; cf[0]++; }
            } else {
                cls[ncl] = (uint8_t)v; clx[ncl++] = 0; cf[v]++; left--;
                while (left >= 3) { const int r = left > 6 ? 6 : left; cls[ncl] = 16; clx[ncl++] = (uint8_t)(r - 3); cf[16]++; left -= r; }
                while (left-- > 0) { cls[ncl] = (uint8_t)v; clx[ncl++] = 0; cf[v]++; }
            }
            i += run;
        }
        uint8_t cl[19];
        uint16_t cc[19];
        huffman_lengths(cf, 19, 7, cl);
        huffman_codes(cl, 19, cc);
        static const uint8_t clorder[19] = {16, 17, 18, 0, 8, 7, 9, 6, 10, 5, 11, 4, 12, 3, 13, 2, 14, 1, 15};
        int hclen = 19;
        while (hclen > 4 && !cl[clorder[hclen - 1]]) hclen--;
        BitSink bs(out);
        bs.put(1, 1);                                  // BFINAL
        bs.put(2, 2);                                  // BTYPE = dynamic Huffman
        bs.put((uint32_t)(hlit - 257), 5);
        bs.put((uint32_t)(hdist - 1), 5);
        bs.put((uint32_t)(hclen - 4), 4);
        for (int i = 0; i < hclen; i++) bs.put(cl[clorder[i]], 3);
        for (int i = 0; i < ncl; i++) {
            bs.put(cc[cls[i]], cl[cls[i]]);
            if (cls[i] == 16) bs.put(clx[i], 2);
            else if (cls[i] == 17) bs.put(clx[i], 3);
            else if (cls[i] == 18) bs.put(clx[i], 7);
        }
        uint32_t mbits[259];                           // a match length's code and extra bits, and how many bits that is
        uint8_t mlen[259];
        for (int len = 3; len <= 258; len++) {
            const int ls = T.len_sym[len];
            mbits[len] = (uint32_t)lc[257 + ls] | ((uint32_t)(len - T.len_base[ls]) << ll[257 + ls]);
            mlen[len] = (uint8_t)(ll[257 + ls] + T.len_xbits[ls]);
        }
        for (uint32_t i = 0; i < ntok; i++) {
            const uint32_t t = toks[i];
            if (!(t & 0x80000000u)) { bs.put(lc[t], ll[t]); continue; }
            const uint32_t len = (t >> 15) & 0x1FFu, dist = (t & 0x7FFFu) + 1, ds = (t >> 24) & 0x1Fu;
            bs.put(mbits[len], mlen[len]);
            bs.put((uint32_t)dc[ds] | ((dist - T.dist_base[ds]) << dl[ds]), dl[ds] + T.dist_xbits[ds]);
        }
        bs.put(lc[256], ll[256]);
        return (size_t)(bs.finish() - out);
    }
};

// The same block written in ONE pass: the Huffman codes are those of a member this thread wrote shortly before (the statistics of
// neighbouring members differ little), so a token goes into the bit stream where it is found -- no token list, no counting pass, no code
// construction per member (three sorts of 316 symbols were a tenth of a member's time).  Every symbol the rows' text can produce has a
// code (floors below); should one ever lack it, or the payload outgrow a BGZF member, `bad` tells the caller to deflate the member with
// zlib instead -- never a corrupt stream.  The statistics of the member being written are kept for the next rebuild.
struct StreamCoder {
    uint8_t ll[288], dl[32];
    uint16_t lc[288], dc[32];
    uint32_t mbits[259];
    uint8_t mlen[259];
    uint8_t hdr[512];                      // the block header (BFINAL .. the coded code lengths) as whole bytes + a tail of hdr_tail_bits bits
    size_t hdr_bytes = 0;
    uint32_t hdr_tail = 0;
    int hdr_tail_bits = 0;
    uint32_t lf[288], df[32];
    BitSink bs{nullptr};
    uint8_t *out0 = nullptr;
    bool bad = false, ready = false;

    // codes from the statistics (lf_in, df_in) of an earlier member
    void build_from(const uint32_t *lf_in, const uint32_t *df_in)
    {
        const DeflateTables &T = deflate_tables();
        uint32_t f[288], d[32];
        for (int i = 0; i < 288; i++) f[i] = i < 286 ? lf_in[i] : 0;
        for (int i = 0; i < 32; i++) d[i] = i < 30 ? df_in[i] : 0;
        // whatever the rows' text can hold gets a code: digits, the separators, the letters of "e-05", "inf" and "nan"; end of block;
        // match lengths up to 66 (an id, or a count string, is at most 46 bytes); every distance
        static const char plausible[] = "0123456789\t\n.e-+naif";
        for (const char *c = plausible; *c; c++) if (!f[(uint8_t)*c]) f[(uint8_t)*c] = 1;
        if (!f[256]) f[256] = 1;
        for (int sy = 257; sy <= 276; sy++) if (!f[sy]) f[sy] = 1;
        for (int sy = 0; sy < 30; sy++) if (!d[sy]) d[sy] = 1;
        huffman_lengths(f, 286, 15, ll);
        huffman_lengths(d, 30, 15, dl);
        huffman_codes(ll, 286, lc);
        huffman_codes(dl, 30, dc);
        for (int len = 3; len <= 258; len++) {
            const int ls = T.len_sym[len];
            mbits[len] = (uint32_t)lc[257 + ls] | ((uint32_t)(len - T.len_base[ls]) << ll[257 + ls]);
            mlen[len] = ll[257 + ls] ? (uint8_t)(ll[257 + ls] + T.len_xbits[ls]) : 0;
        }
        // the header, once
        uint8_t tmp[600];
        BitSink h(tmp);
        int hlit = 286, hdist = 30;
        while (hlit > 257 && !ll[hlit - 1]) hlit--;
        while (hdist > 1 && !dl[hdist - 1]) hdist--;
        uint8_t seq[320];
        int ns = 0;
        for (int i = 0; i < hlit; i++) seq[ns++] = ll[i];
        for (int i = 0; i < hdist; i++) seq[ns++] = dl[i];
        uint8_t cls[320], clx[320];
        int ncl = 0;
        uint32_t cf[19] = {0};
        for (int i = 0; i < ns;) {
            int run = 1;
            while (i + run < ns && seq[i + run] == seq[i]) run++;
            const int v = seq[i];
            int left = run;
            if (v == 0) {
                while (left >= 11) { const int r = left > 138 ? 138 : left; cls[ncl] = 18; clx[ncl++] = (uint8_t)(r - 11); cf[18]++; left -= r; }
                if (left >= 3) { cls[ncl] = 17; clx[ncl++] = (uint8_t)(left - 3); cf[17]++; left = 0; }
                while (left-- > 0) { cls[ncl] = 0; clx[ncl++] = 0; cf[0]++; }
            } else {
                cls[ncl] = (uint8_t)v; clx[ncl++] = 0; cf[v]++; left--;
                while (left >= 3) { const int r = left > 6 ? 6 : left; cls[ncl] = 16; clx[ncl++] = (uint8_t)(r - 3); cf[16]++; left -= r; }
                while (left-- > 0) { cls[ncl] = (uint8_t)v; clx[ncl++] = 0; cf[v]++; }
            }
            i += run;
        }
        uint8_t cl[19];
        uint16_t cc[19];
        huffman_lengths(cf, 19, 7, cl);
        huffman_codes(cl, 19, cc);
        static const uint8_t clorder[19] = {16, 17, 18, 0, 8, 7, 9, 6, 10, 5, 11, 4, 12, 3, 13, 2, 14, 1, 15};
        int hclen = 19;
        while (hclen > 4 && !cl[clorder[hclen - 1]]) hclen--;
        h.put(1, 1);
        h.put(2, 2);
        h.put((uint32_t)(hlit - 257), 5);
        h.put((uint32_t)(hdist - 1), 5);
        h.put((uint32_t)(hclen - 4), 4);
        for (int i = 0; i < hclen; i++) h.put(cl[clorder[i]], 3);
        for (int i = 0; i < ncl; i++) {
            h.put(cc[cls[i]], cl[cls[i]]);
            if (cls[i] == 16) h.put(clx[i], 2);
            else if (cls[i] == 17) h.put(clx[i], 3);
            else if (cls[i] == 18) h.put(clx[i], 7);
        }
        // whole bytes so far + the bits still in the accumulator
        hdr_bytes = (size_t)(h.p - tmp);
        size_t tail_bits = (size_t)h.n;
        while (tail_bits >= 8) { tmp[hdr_bytes++] = (uint8_t)h.acc; h.acc >>= 8; tail_bits -= 8; }
        memcpy(hdr, tmp, hdr_bytes);
        hdr_tail = (uint32_t)(h.acc & ((1u << tail_bits) - 1u));
        hdr_tail_bits = (int)tail_bits;
        ready = true;
    }

    void begin(uint8_t *out)
    {
        out0 = out;
        memcpy(out, hdr, hdr_bytes);
        bs = BitSink(out + hdr_bytes);
        bs.acc = hdr_tail;
        bs.n = hdr_tail_bits;
        memset(lf, 0, sizeof lf);
        memset(df, 0, sizeof df);
        bad = false;
    }
    inline void lit(uint8_t b) { lf[b]++; bad |= ll[b] == 0; bs.put(lc[b], ll[b]); }
    inline void lits(const char *t, int n) { for (int i = 0; i < n; i++) lit((uint8_t)t[i]); }
    inline void match(uint32_t len, uint32_t dist)
    {
        const DeflateTables &T = deflate_tables();
        const uint32_t ds = (uint32_t)T.dist_sym(dist);
        lf[257 + T.len_sym[len]]++;
        df[ds]++;
        bad |= (mlen[len] == 0) | (dl[ds] == 0);
        bs.put(mbits[len], mlen[len]);
        bs.put((uint32_t)dc[ds] | ((dist - T.dist_base[ds]) << dl[ds]), dl[ds] + T.dist_xbits[ds]);
    }
    size_t finish()
    {
        lf[256]++;
        bs.put(lc[256], ll[256]);
        return (size_t)(bs.finish() - out0);
    }
};

// ---- "count \t frequency \n" for every count below 65536 -------------------------------------------------------------
struct CountStrings {
    static constexpr uint32_t N = 65536, STRIDE = 32;      // 5 digits + tab + at most 23 characters of repr + newline = 30
    std::vector<char> text;
    std::vector<uint8_t> len;
    double total;
    explicit CountStrings(double total_) : text((size_t)N * STRIDE), len(N), total(total_) {}
    void build(uint32_t a, uint32_t b)
    {
        for (uint32_t c = a; c < b; c++) len[c] = (uint8_t)format_count_string(c, total, &text[(size_t)c * STRIDE]);
    }
};

// Sum of digits(i) over i in [r0, r1)
inline uint64_t digits_sum(uint64_t r0, uint64_t r1)
{
    uint64_t s = 0, lo = 0, hi = 10;
    for (int d = 1; d <= 20 && lo < r1; d++) {
        const uint64_t a = r0 > lo ? r0 : lo, b = (r1 < hi || d == 20) ? r1 : hi;
        if (b > a) s += (uint64_t)d * (b - a);
        lo = hi;
        hi = d >= 19 ? ~0ull : hi * 10;
    }
    return s;
}

struct GrowBuf {                                            // (std::vector would zero every member's worst-case room)
    uint8_t *p = nullptr;
    size_t n = 0, cap = 0;
    ~GrowBuf() { free(p); }
    bool room(size_t extra)
    {
        if (n + extra <= cap) return true;
        size_t c = cap ? cap : (1u << 20);
        while (c < n + extra) c *= 2;
        uint8_t *q = (uint8_t *)realloc(p, c);
        if (!q) return false;
        p = q; cap = c;
        return true;
    }
};

struct KdbRowsJob {
    const uint64_t *counts;
    uint64_t nbins;
    double total;
    int level;
    bool rows_encoder;
    uint64_t chunk_rows, nchunks;

    CountStrings cs;
    KdbRowsJob(const uint64_t *c, uint64_t n, uint64_t total_kmers, int level_, bool rows)
        : counts(c), nbins(n), total((double)total_kmers), level(level_), rows_encoder(rows), cs((double)total_kmers) {}

    uint64_t chunk_text_bytes(uint64_t c) const
    {
        const uint64_t r0 = c * chunk_rows, r1 = std::min(nbins, r0 + chunk_rows);
        uint64_t s = 2 * digits_sum(r0, r1) + 2 * (r1 - r0);
        char tmp[64];
        for (uint64_t r = r0; r < r1; r++) {
            const uint64_t v = counts[r];
            s += v < CountStrings::N ? cs.len[v] : (uint64_t)format_count_string(v, total, tmp);
        }
        return s;
    }
};

// The rows of one chunk and the members it owns.  One per worker thread; its buffers are reused from chunk to chunk.
struct ChunkWorker {
    const KdbRowsJob &J;
    std::vector<char> text;
    std::vector<uint32_t> lastpos;      // count & 0xFFFF -> 1 + offset in `text` of the latest row string with that count
    MemberCoder mc;                     // two passes: a member's tokens, then its own Huffman codes (this thread's first member)
    StreamCoder sc;                     // one pass with the codes of a member before (every later member)
    bool direct = false;                // how the open member is being coded
    int since_rebuild = 0;
    static constexpr int REBUILD_EVERY = 4;
    explicit ChunkWorker(const KdbRowsJob &j) : J(j), lastpos(65536, 0) {}

    inline void lit(uint8_t b) { if (direct) sc.lit(b); else mc.lit(b); }
    inline void lits(const char *t, int n) { if (direct) sc.lits(t, n); else mc.lits(t, n); }
    inline void match(uint32_t len, uint32_t dist) { if (direct) sc.match(len, dist); else mc.match(len, dist); }

    // a member opens: room for its worst case in `out` (the buffer may move now, not while the member is being written)
    bool open_member(GrowBuf &out)
    {
        if (!out.room(BGZF_HEAD + 2 * BGZF_TEXT + 4096 + BGZF_TAIL)) return false;
        if (!J.rows_encoder) return true;
        direct = sc.ready;
        if (direct) sc.begin(out.p + out.n + BGZF_HEAD); else mc.begin();
        return true;
    }

    // close the member text[m0, m1): payload, frame
    bool close_member(size_t m0, size_t m1, GrowBuf &out)
    {
        uint8_t *member = out.p + out.n;
        const uint8_t *data = (const uint8_t *)text.data() + m0;
        size_t clen;
        if (!J.rows_encoder) {
            clen = zlib_payload(data, m1 - m0, J.level, member + BGZF_HEAD);
        } else if (direct) {
            clen = sc.finish();
            // (cannot happen with the rows' text; if it ever does, zlib writes the member: the file stays right)
            if (sc.bad || clen >= BGZF_MAX_PAYLOAD) clen = zlib_payload(data, m1 - m0, 6, member + BGZF_HEAD);
            if (++since_rebuild >= REBUILD_EVERY) { sc.build_from(sc.lf, sc.df); since_rebuild = 0; }
        } else {
            clen = mc.finish(member + BGZF_HEAD);
            if (clen >= BGZF_MAX_PAYLOAD) clen = zlib_payload(data, m1 - m0, 6, member + BGZF_HEAD);
            sc.build_from(mc.lf, mc.df);
            since_rebuild = 0;
        }
        if (!clen) return false;
        bgzf_frame(member, clen, crc32_bytes(data, m1 - m0), (uint32_t)(m1 - m0));
        out.n += BGZF_HEAD + clen + BGZF_TAIL;
        return true;
    }

    // T0, T1: the text offsets of this chunk's first row and of the next chunk's (known once every chunk before has been measured)
    bool run(uint64_t c, uint64_t T0, uint64_t T1, GrowBuf &out, uint64_t *nmembers)
    {
        out.n = 0;
        *nmembers = 0;
        const uint64_t first = (T0 + BGZF_TEXT - 1) / BGZF_TEXT * BGZF_TEXT;      // the members that start in [T0, T1) are this chunk's
        if (first >= T1) return true;
        // up to the end of the last member it owns -- or to the last row, if the text ends before that (its length is not known yet
        // when a chunk in the middle of the vector is written: the chunks behind may still be on their way from the device)
        const uint64_t E = ((T1 - 1) / BGZF_TEXT + 1) * BGZF_TEXT;
        const size_t need = (size_t)(E - T0);
        if (text.size() < need + 256) text.resize(need + 256 + (need >> 3));
        char *buf = text.data();
        const bool ROWS = J.rows_encoder;
        const char *cstext = J.cs.text.data();
        const uint8_t *cslen = J.cs.len.data();

        uint64_t r = c * J.chunk_rows;
        char id[24];
        int L = fmt_u64(r, id);
        memset(id + L, 0, sizeof id - (size_t)L);
        int P = 0;                                  // leading digits this row's id shares with the row before
        size_t pos = 0, prev_s = 0;
        bool have_prev = false;
        bool in_member = false;
        size_t cut = (size_t)(first - T0);          // where the open member ends, or (before the first one) where it begins
        size_t mstart = 0;
        while (pos < need && r < J.nbins) {
            const size_t s = pos;
            char *p = buf + s;
            memcpy(p, id, 24); p += L; *p++ = '\t';
            memcpy(p, id, 24); p += L; *p++ = '\t';
            const uint64_t v = J.counts[r];
            int C;
            if (v < CountStrings::N) { memcpy(p, cstext + (size_t)v * CountStrings::STRIDE, CountStrings::STRIDE); C = cslen[v]; }
            else C = format_count_string(v, J.total, p);
            const size_t cs_at = s + 2 * (size_t)L + 2, rowend = cs_at + (size_t)C;
            if (in_member && rowend <= cut) {
                if (ROWS) {
                    if (P >= 3 && have_prev && prev_s >= mstart) { match((uint32_t)P, (uint32_t)(s - prev_s)); lits(id + P, L - P); }
                    else lits(id, L);
                    lit('\t');
                    if (L >= 2) match((uint32_t)L + 1, (uint32_t)L + 1);
                    else { lit((uint8_t)id[0]); lit('\t'); }
                    uint32_t &lp = lastpos[v & 0xFFFFu];
                    const size_t cand = (size_t)lp - 1;           // (lp == 0: none -> a huge value that fails the range test)
                    if (lp && cand >= mstart && cand < cs_at && cs_at - cand <= 32768 && memcmp(buf + cand, buf + cs_at, (size_t)C) == 0)
                        match((uint32_t)C, (uint32_t)(cs_at - cand));
                    else lits(buf + cs_at, C);
                    lp = (uint32_t)(cs_at + 1);
                }
            } else {
                // the row meets a member boundary (or lies before this chunk's first member): byte by byte
                for (size_t b = s; b < rowend; b++) {
                    if (b == cut) {
                        if (in_member) { if (!close_member(mstart, cut, out)) return false; (*nmembers)++; in_member = false; }
                        if (cut < need) { if (!open_member(out)) return false; in_member = true; mstart = cut; cut = std::min(need, cut + BGZF_TEXT); }
                        else break;
                    }
                    if (in_member && ROWS) lit((uint8_t)buf[b]);
                }
            }
            prev_s = s;
            have_prev = true;
            pos = rowend;
            if (in_member && pos == cut) {              // a row that ends exactly on the boundary
                if (!close_member(mstart, cut, out)) return false;
                (*nmembers)++;
                in_member = false;
                if (cut < need) { if (!open_member(out)) return false; in_member = true; mstart = cut; cut = std::min(need, cut + BGZF_TEXT); }
            }
            // next id: increment the decimal string in place
            r++;
            int j = L - 1;
            while (j >= 0 && id[j] == '9') id[j--] = '0';
            if (j < 0) { memmove(id + 1, id, (size_t)L); id[0] = '1'; L++; P = 0; }
            else { id[j]++; P = j; }
        }
        if (in_member) {                                // the rows ended inside the member: the file's last, partial one
            if (!close_member(mstart, pos, out)) return false;
            (*nmembers)++;
        }
        return true;
    }
};

// append the 4^k rows to `path` (which already holds the header member(s)); returns 0 ok.
// encoder: 0 = row-aware (default), 1 = zlib at `level`, -1 = KDB_KDB_ENCODER from the environment ("zlib" / "rows"), else row-aware
// rows_ready (optional): the counts are still being filled in, front to back (a device-to-host copy in pieces): *rows_ready = how many
// are there; ~0 = the producer failed
inline int write_kdb_rows(const char *path, const uint64_t *counts, uint64_t nbins, uint64_t total_kmers, int level, int nthreads,
                          uint64_t *nblocks_out, const char **why, int encoder = -1, const std::atomic<uint64_t> *rows_ready = nullptr)
{
    if (encoder < 0) {
        const char *env = getenv("KDB_KDB_ENCODER");
        encoder = (env && strcmp(env, "zlib") == 0) ? 1 : 0;
    }
    if (nblocks_out) *nblocks_out = 0;
    if (nbins == 0) return 0;
    const int fd = open(path, O_WRONLY | O_CREAT, 0644);
    if (fd < 0) { *why = "cannot open output file for append"; return 1; }
    const off_t base = lseek(fd, 0, SEEK_END);             // behind the header member(s)
    if (base < 0) { close(fd); *why = "cannot seek in the output file"; return 1; }
    if (nthreads < 1) nthreads = 1;
    KdbRowsJob J(counts, nbins, total_kmers, level, encoder == 0);
    // a chunk: rows whose text is a few MB (dozens of members: the rows a chunk formats beyond its own, to finish its last member, stay a few %)
    J.chunk_rows = std::min<uint64_t>(65536, std::max<uint64_t>(4096, nbins / (4 * (uint64_t)nthreads)));
    J.nchunks = (nbins + J.chunk_rows - 1) / J.chunk_rows;
    if ((uint64_t)nthreads > J.nchunks) nthreads = (int)J.nchunks;

    // Two quantities are committed in chunk order, each by whichever thread completes the prefix: the TEXT offset of a chunk (known once
    // every chunk before it has been measured -- a pass over its counts that takes a twentieth of writing it) and its FILE offset (known once
    // every chunk before it has been deflated).  Between the two a thread formats and deflates its chunk; behind the second it writes its own
    // chunk (pwrite): the writes go on side by side (one writer thread moved 2-3 GB/s into tmpfs -- less than sixteen encoders make).
    // No pass over the whole vector comes first, so the rows may still be arriving: `rows_ready` (optional) says how many counts are there.
    struct Ordered {                                          // prefix sums that become final in index order
        std::vector<uint64_t> off; std::vector<char> have; uint64_t committed = 0;
        explicit Ordered(uint64_t n) : off(n + 1, 0), have(n, 0) {}
        bool publish(uint64_t c, uint64_t len)                // (under the job's mutex) -> something became final
        {
            off[c + 1] = len; have[c] = 1;
            bool moved = false;
            while (committed < have.size() && have[committed]) { off[committed + 1] += off[committed]; committed++; moved = true; }
            return moved;
        }
    } text_off(J.nchunks), file_off(J.nchunks);
    std::mutex mu;
    std::condition_variable cv_commit, cv_phase;
    std::atomic<uint64_t> next_chunk{0}, nblocks{0};
    int phase_arrived = 0, phase = 0;                        // a reusable barrier of nthreads parties
    bool failed = false;
    const char *fail_why = "";

    auto barrier = [&] {
        std::unique_lock<std::mutex> lk(mu);
        const int my = phase;
        if (++phase_arrived == nthreads) { phase_arrived = 0; phase++; cv_phase.notify_all(); }
        else cv_phase.wait(lk, [&] { return phase != my; });
    };
    auto wait_rows = [&](uint64_t upto) -> bool {            // the counts of rows [0, upto) are in `counts`
        if (!rows_ready) return true;
        for (int spin = 0;; spin++) {
            const uint64_t have = rows_ready->load(std::memory_order_acquire);
            if (have == ~0ull) return false;                 // (the producer gave up)
            if (have >= upto) return true;
            if (spin < 64) std::this_thread::yield(); else std::this_thread::sleep_for(std::chrono::microseconds(50));
        }
    };

    auto worker = [&](int t) {
        J.cs.build((uint32_t)((uint64_t)CountStrings::N * (uint64_t)t / (uint64_t)nthreads), (uint32_t)((uint64_t)CountStrings::N * (uint64_t)(t + 1) / (uint64_t)nthreads));
        barrier();                                           // the table is complete
        ChunkWorker w(J);
        GrowBuf out;
        for (;;) {
            const uint64_t c = next_chunk.fetch_add(1);
            if (c >= J.nchunks) break;
            // a chunk's rows, and the few thousand behind them that finish its last member (a row is at least 10 bytes)
            bool ok = wait_rows(std::min(nbins, (c + 1) * J.chunk_rows + BGZF_TEXT / 8));
            const uint64_t len = ok ? J.chunk_text_bytes(c) : 0;
            uint64_t T0 = 0;
            {
                std::unique_lock<std::mutex> lk(mu);
                if (!ok) { failed = true; fail_why = "the rows stopped arriving"; }
                if (text_off.publish(c, len) || failed) cv_commit.notify_all();
                cv_commit.wait(lk, [&] { return failed || text_off.committed >= c; });
                if (failed) break;
                T0 = text_off.off[c];
            }
            uint64_t nm = 0;
            ok = w.run(c, T0, T0 + len, out, &nm);
            uint64_t at = 0;
            {
                std::unique_lock<std::mutex> lk(mu);
                if (!ok) { failed = true; fail_why = "deflate failed"; }
                if (file_off.publish(c, out.n) || failed) cv_commit.notify_all();
                cv_commit.wait(lk, [&] { return failed || file_off.committed >= c; });
                if (failed) break;
                at = file_off.off[c];
            }
            const uint8_t *p = out.p;
            size_t left = out.n;
            while (left) {
                const ssize_t wr = pwrite(fd, p, left, base + (off_t)at);
                if (wr < 0) { if (errno == EINTR) continue; break; }
                p += wr; left -= (size_t)wr; at += (uint64_t)wr;
            }
            if (left) {
                std::lock_guard<std::mutex> lk(mu);
                failed = true; fail_why = "short write";
                cv_commit.notify_all();
                break;
            }
            nblocks.fetch_add(nm);
        }
        if (failed) { std::lock_guard<std::mutex> lk(mu); cv_commit.notify_all(); }
    };

    std::vector<std::thread> th;
    for (int t = 1; t < nthreads; t++) th.emplace_back(worker, t);
    worker(0);
    for (auto &x : th) x.join();
    if (close(fd) != 0 && !failed) { failed = true; fail_why = "close failed"; }
    if (nblocks_out) *nblocks_out = nblocks.load();
    if (failed) { *why = fail_why; return 1; }
    return 0;
}


// ---------------------------------------------------------------------------------------------------------------
// The rows back into arrays: KDBReader._slurp (kmerdb/fileutil.py:308-466) reads 4^k text rows one by one through Bio.bgzf;
// here the file's BGZF members are inflated in groups on `nthreads` threads (the engine's own DEFLATE decoder, kdb_inflate.cpp.h) and
// every group's whole lines are parsed where they were inflated; a line that straddles two groups is put together afterwards.
// Row "x \t kmer_id \t count \t frequency": kmer_ids[x] = kmer_id, counts[kmer_id] = count, freqs[kmer_id] = the file's frequency column
// (fileutil.py:367-369); x must be the line's number (:361) and there must be exactly `nbins` rows of four columns (:354).
// Returns 0 ok, 1 malformed (why), 3 the file is not made of BGZF members alone (the caller reads it as one gzip stream instead).
// ---------------------------------------------------------------------------------------------------------------
struct KdbRowGroup {
    size_t m0 = 0, m1 = 0;                       // members [m0, m1)
    std::string head, tail;                      // the bytes before the first newline / behind the last one
    uint64_t first_x = 0, nlines = 0;            // whole lines parsed in place: x = first_x .. first_x + nlines - 1
    bool any_newline = false;
};

inline bool kdb_parse_row(const char *p, const char *e, uint64_t nbins, uint64_t *x_out, uint64_t *kmer_ids, uint64_t *counts, double *freqs, const char **why)
{
    uint64_t v[3];
    for (int c = 0; c < 3; c++) {
        if (p >= e || *p < '0' || *p > '9') { *why = "a .kdb row does not have four columns of numbers"; return false; }
        uint64_t a = 0;
        while (p < e && *p >= '0' && *p <= '9') { a = a * 10u + (uint64_t)(*p - '0'); p++; }
        if (p >= e || *p != '\t') { *why = "a .kdb row does not have four tab-separated columns"; return false; }
        p++;
        v[c] = a;
    }
    const char *fe = e;
    while (fe > p && (fe[-1] == '\r' || fe[-1] == ' ')) fe--;
    if (memchr(p, '\t', (size_t)(fe - p))) { *why = "a .kdb row has more than four columns"; return false; }
    double f = 0.0;
    const auto res = std::from_chars(p, fe, f);
    if (res.ec != std::errc() || res.ptr != fe) { *why = "the frequency column of a .kdb row is not a number"; return false; }
    if (v[0] >= nbins || v[1] >= nbins) { *why = "a .kdb row's index or k-mer id is not below 4^k"; return false; }
    *x_out = v[0];
    kmer_ids[v[0]] = v[1];
    counts[v[1]] = v[2];
    freqs[v[1]] = f;
    return true;
}

inline int read_kdb_rows(const char *path, uint64_t nbins, uint64_t *kmer_ids, uint64_t *counts, double *freqs, int nthreads, uint64_t *nrows_out,
                         const char **why)
{
    *nrows_out = 0;
    const int fd = open(path, O_RDONLY);
    if (fd < 0) { *why = "cannot open the file"; return 1; }
    struct stat st;
    if (fstat(fd, &st) != 0 || st.st_size <= 0) { close(fd); *why = "cannot read the file"; return 1; }
    const size_t n = (size_t)st.st_size;
    const uint8_t *src = (const uint8_t *)mmap(nullptr, n, PROT_READ, MAP_PRIVATE, fd, 0);
    close(fd);
    if (src == MAP_FAILED) { *why = "cannot map the file"; return 1; }
    struct Unmap { const uint8_t *p; size_t n; ~Unmap() { munmap((void *)p, n); } } unmap{src, n};
    std::vector<BgzfBlock> blocks;
    for (size_t at = 0; at < n;) {
        BgzfBlock b;
        const int rc = bgzf_block_at(src, n, at, &b);
        if (rc != 0 || b.isize > 65536) return 3;
        blocks.push_back(b);
        at += b.csize;
    }
    if (blocks.empty()) return 3;
    auto inflate_members = [&](size_t m0, size_t m1, std::string &text, Inflater &inf) -> bool {
        size_t total = 0;
        for (size_t m = m0; m < m1; m++) total += blocks[m].isize;
        text.resize(total + 1024);
        uint8_t *base = (uint8_t *)&text[0];
        size_t at = 0;
        for (size_t m = m0; m < m1; m++) {
            const BgzfBlock &b = blocks[m];
            inf.reset(src + b.src + b.data_off, src + b.src + b.csize - 8);
            uint8_t *out = base + at;
            bool ok = inf.run(out, base + at + b.isize, base + at);
            if (ok && inf.state != Inflater::DONE) { uint8_t *o2 = out; ok = inf.run(o2, o2 + 1, base + at) && o2 == out && inf.state == Inflater::DONE; }
            const uint8_t *t = src + b.src + b.csize - 8;
            const uint32_t want = (uint32_t)t[0] | ((uint32_t)t[1] << 8) | ((uint32_t)t[2] << 16) | ((uint32_t)t[3] << 24);
            if (!ok || (size_t)(out - (base + at)) != b.isize || crc32_bytes(base + at, b.isize) != want) return false;
            at += b.isize;
        }
        text.resize(total);
        return true;
    };
    // the header: members from the start until the delimiter line has been seen
    static const char delim[] = "\n========================\n";
    const size_t dl = sizeof delim - 1;
    size_t body_member = 0, body_skip = 0;          // the rows start `body_skip` bytes into member `body_member`
    {
        Inflater inf;
        std::string text, one;
        bool found = false;
        for (size_t m = 0; m < blocks.size() && !found; m++) {
            const size_t before = text.size();
            if (!inflate_members(m, m + 1, one, inf)) { *why = "corrupt BGZF member (inflate or CRC-32 failed)"; return 1; }
            text += one;
            const size_t from = before > dl ? before - dl : 0;
            const size_t at = text.find(delim, from, dl);
            if (at != std::string::npos) {
                const size_t body = at + dl;          // uncompressed offset of the first row
                size_t off = 0;
                for (size_t q = 0; q <= m; q++) { if (body < off + blocks[q].isize || q == m) { body_member = q; body_skip = body - off; break; } off += blocks[q].isize; }
                if (body_skip == blocks[body_member].isize) { body_member++; body_skip = 0; }
                found = true;
            }
            if (text.size() > (64u << 20)) break;
        }
        if (!found) { *why = "no .kdb header delimiter"; return 1; }
    }
    const size_t GROUP = 64;
    std::vector<KdbRowGroup> groups;
    for (size_t m = body_member; m < blocks.size(); m += GROUP) { KdbRowGroup g; g.m0 = m; g.m1 = std::min(blocks.size(), m + GROUP); groups.push_back(g); }
    if (groups.empty()) { if (nbins) { *why = "the file holds no rows"; return 1; } return 0; }
    std::atomic<size_t> next(0);
    std::atomic<int> bad(0);
    std::mutex mu;
    const char *first_why = "";
    auto fail_with = [&](const char *w) { std::lock_guard<std::mutex> lk(mu); if (!bad.load()) { first_why = w; bad = 1; } };
    auto work = [&] {
        std::unique_ptr<Inflater> inf(new Inflater());
        std::string text;
        for (;;) {
            const size_t gi = next.fetch_add(1);
            if (gi >= groups.size() || bad.load()) return;
            KdbRowGroup &g = groups[gi];
            if (!inflate_members(g.m0, g.m1, text, *inf)) { fail_with("corrupt BGZF member (inflate or CRC-32 failed)"); return; }
            const char *p = text.data() + (gi == 0 ? body_skip : 0), *e = text.data() + text.size();
            const char *nl = (const char *)memchr(p, '\n', (size_t)(e - p));
            if (!nl) { g.head.assign(p, (size_t)(e - p)); continue; }          // (no line ends in this group: all of it is one fragment)
            g.any_newline = true;
            if (gi == 0) nl = p - 1;                                            // the first row starts right here: nothing belongs to a group before
            else g.head.assign(p, (size_t)(nl - p));
            const char *ls = nl + 1;
            uint64_t expect = 0;
            bool have = false;
            while (ls < e) {
                const char *le = (const char *)memchr(ls, '\n', (size_t)(e - ls));
                if (!le) break;
                uint64_t x = 0;
                const char *w = "";
                if (!kdb_parse_row(ls, le, nbins, &x, kmer_ids, counts, freqs, &w)) { fail_with(w); return; }
                if (!have) { g.first_x = x; expect = x; have = true; }
                if (x != expect) { fail_with("a .kdb row's index does not match its line number"); return; }
                expect++;
                g.nlines++;
                ls = le + 1;
            }
            g.tail.assign(ls, (size_t)(e - ls));
        }
    };
    int t = nthreads < 1 ? 1 : nthreads;
    if ((size_t)t > groups.size()) t = (int)groups.size();
    {
        std::vector<std::thread> th;
        for (int i = 1; i < t; i++) th.emplace_back(work);
        work();
        for (auto &x : th) x.join();
    }
    if (bad.load()) { *why = first_why; return 1; }
    // the lines that straddle groups, and the line numbers across all of them
    uint64_t line = 0;
    std::string carry;
    auto stitched = [&](const std::string &l) -> bool {
        uint64_t x = 0;
        const char *w = "";
        if (!kdb_parse_row(l.data(), l.data() + l.size(), nbins, &x, kmer_ids, counts, freqs, &w)) { *why = w; return false; }
        if (x != line) { *why = "a .kdb row's index does not match its line number"; return false; }
        line++;
        return true;
    };
    for (size_t gi = 0; gi < groups.size(); gi++) {
        KdbRowGroup &g = groups[gi];
        carry += g.head;
        if (!g.any_newline) continue;
        if (!carry.empty() && !stitched(carry)) return 1;          // (the end of the group before + the start of this one: one whole line)
        carry.clear();
        if (g.nlines) { if (g.first_x != line) { *why = "a .kdb row's index does not match its line number"; return 1; } line += g.nlines; }
        carry = g.tail;
    }
    if (!carry.empty()) {                            // a last line without its newline
        bool blank = true;
        for (char c : carry) if (c != ' ' && c != '\r' && c != '\n') blank = false;
        if (!blank && !stitched(carry)) return 1;
    }
    *nrows_out = line;
    if (line != nbins) { *why = "the file does not hold 4^k rows"; return 1; }
    return 0;
}

}  // namespace kdbhost
