// kdb_inflate.cpp.h -- a DEFLATE (RFC 1951) decoder for the host feed (no GPU): one gzip stream of FASTQ text is a serial
// dependency chain, so the only way to read `.fastq.gz` -- what the reference opens with gzip.open, kmerdb/parse.py:64-72 --
// faster is a faster decoder.  zlib 1.2.11's inflate runs at ~250-350 MB/s of output on this class of host; this one keeps a
// 64-bit bit buffer that is refilled with one unaligned 8-byte load, decodes literals / lengths through an 11-bit primary table
// (+ subtables for longer codes) and distances through an 8-bit one, and copies matches eight bytes at a time.
//
// Streaming: the compressed input is one contiguous byte range (the caller maps the file); output goes to a caller-owned arena
//     [ 32 KiB history | data ... | slack ]
// that the caller drains and re-bases (last 32 KiB moved to the front) whenever `run` returns because the data part is full.
// The decoder can be resumed at any symbol boundary.  Errors: every malformed input that zlib rejects is rejected here too
// (bad block type, bad stored length, over-subscribed or incomplete code sets, distance too far back, invalid symbol, input
// ends inside the stream); tests/test_reader_cpu.py compares it with zlib on every block type and on corrupted streams.
#pragma once
#include <cstdint>
#include <cstring>

namespace kdbhost {

struct Inflater {
    static constexpr uint32_t WINDOW = 32768;
    static constexpr int LIT_BITS = 11, DIST_BITS = 8;
    static constexpr uint32_t F_LIT = 1u << 12, F_EOB = 1u << 13, F_SUB = 1u << 14, F_BAD = 1u << 15;

    // input
    const uint8_t *in = nullptr, *in_end = nullptr;
    uint64_t bitbuf = 0;
    uint32_t bitcnt = 0;
    // block state
    enum { HEADER, STORED, CODED, DONE } state = HEADER;
    bool last_block = false;
    uint32_t stored_left = 0;
    const char *err = nullptr;
    // entry: bits 0..4 code length, bits 8..11 extra bits (or, for F_SUB, the subtable's index bits), flags, bits 16..31 value
    uint32_t lit[(1u << LIT_BITS) + 1024];
    uint32_t dist[(1u << DIST_BITS) + 512];

    void reset(const uint8_t *p, const uint8_t *e) { in = p; in_end = e; bitbuf = 0; bitcnt = 0; state = HEADER; last_block = false; stored_left = 0; err = nullptr; }

    // bytes of input not yet consumed (whole bytes still in the bit buffer count as unconsumed)
    const uint8_t *input_position() const { return in - (bitcnt >> 3); }

    inline void refill()
    {
        if (in_end - in >= 8) {
            uint64_t w;
            memcpy(&w, in, 8);
            bitbuf |= w << bitcnt;
            in += (63 - bitcnt) >> 3;
            bitcnt |= 56;
        } else {
            while (bitcnt <= 56 && in < in_end) { bitbuf |= (uint64_t)*in++ << bitcnt; bitcnt += 8; }
        }
    }
    inline uint32_t peek(uint32_t n) const { return (uint32_t)(bitbuf & ((1ull << n) - 1ull)); }
    inline void drop(uint32_t n) { bitbuf >>= n; bitcnt -= n; }

    static uint32_t bitrev(uint32_t c, int n)
    {
        uint32_t r = 0;
        for (int i = 0; i < n; i++) { r = (r << 1) | (c & 1u); c >>= 1; }
        return r;
    }

    // canonical Huffman code -> lookup table.  lens[0..n): code length per symbol (0 = unused); entry_of(sym) gives the entry without
    // its length field.  Returns false for an over-subscribed code, and for an incomplete one unless it is a single 1-bit code of
    // a literal/length or distance set (`allow_incomplete`; what zlib's inflate_table accepts).  Unused table slots hold F_BAD.
    template <typename F>
    static bool build(uint32_t *table, int primary_bits, size_t table_cap, const uint8_t *lens, int n, bool allow_incomplete, F entry_of)
    {
        int count[16] = {0};
        for (int i = 0; i < n; i++) count[lens[i]]++;
        if (count[0] == n) { if (!allow_incomplete) return false; for (size_t i = 0; i < (1u << primary_bits); i++) table[i] = F_BAD | 1u; return true; }
        int left = 1, max_len = 0;
        for (int l = 1; l <= 15; l++) { left = (left << 1) - count[l]; if (left < 0) return false; if (count[l]) max_len = l; }
        if (left > 0 && !(allow_incomplete && max_len == 1)) return false;
        uint32_t next[16];
        uint32_t code = 0;
        for (int l = 1; l <= 15; l++) { code = (code + (uint32_t)count[l - 1] * (l > 1 ? 1u : 0u)) << 1; next[l] = code; }
        // (count[0] must not enter the code of length 1: handled by the factor above)
        const uint32_t psize = 1u << primary_bits;
        for (uint32_t i = 0; i < psize; i++) table[i] = F_BAD | 1u;
        // longest code under every primary prefix that needs a subtable
        uint8_t sub_len[1u << LIT_BITS];
        memset(sub_len, 0, psize);
        uint32_t codes[320];
        for (int s = 0; s < n; s++) {
            const int l = lens[s];
            if (!l) continue;
            const uint32_t r = bitrev(next[l]++, l);
            codes[s] = r;
            if (l > primary_bits) { uint8_t &m = sub_len[r & (psize - 1)]; if (l > m) m = (uint8_t)l; }
        }
        size_t used = psize;
        for (uint32_t p = 0; p < psize; p++) {
            if (!sub_len[p]) continue;
            const uint32_t sb = (uint32_t)sub_len[p] - (uint32_t)primary_bits;
            if (used + (1u << sb) > table_cap) return false;
            table[p] = ((uint32_t)used << 16) | F_SUB | (sb << 8) | (uint32_t)primary_bits;
            for (uint32_t i = 0; i < (1u << sb); i++) table[used + i] = F_BAD | (uint32_t)sub_len[p];
            used += 1u << sb;
        }
        for (int s = 0; s < n; s++) {
            const int l = lens[s];
            if (!l) continue;
            const uint32_t r = codes[s], e = entry_of(s) | (uint32_t)l;
            if (l <= primary_bits) {
                for (uint32_t i = r; i < psize; i += 1u << l) table[i] = e;
            } else {
                const uint32_t pe = table[r & (psize - 1)], sb = (pe >> 8) & 0xFu, base = pe >> 16;
                for (uint32_t i = r >> primary_bits; i < (1u << sb); i += 1u << (l - primary_bits)) table[base + i] = e;
            }
        }
        return true;
    }

    static uint32_t lit_entry(int s)
    {
        static const uint16_t lbase[29] = {3, 4, 5, 6, 7, 8, 9, 10, 11, 13, 15, 17, 19, 23, 27, 31, 35, 43, 51, 59, 67, 83, 99, 115, 131, 163, 195, 227, 258};
        static const uint8_t lext[29] = {0, 0, 0, 0, 0, 0, 0, 0, 1, 1, 1, 1, 2, 2, 2, 2, 3, 3, 3, 3, 4, 4, 4, 4, 5, 5, 5, 5, 0};
        if (s < 256) return ((uint32_t)s << 16) | F_LIT;
        if (s == 256) return F_EOB;
        if (s > 285) return F_BAD;
        return ((uint32_t)lbase[s - 257] << 16) | ((uint32_t)lext[s - 257] << 8);
    }
    static uint32_t dist_entry(int s)
    {
        static const uint16_t dbase[30] = {1, 2, 3, 4, 5, 7, 9, 13, 17, 25, 33, 49, 65, 97, 129, 193, 257, 385, 513, 769, 1025, 1537, 2049, 3073, 4097, 6145, 8193, 12289, 16385, 24577};
        static const uint8_t dext[30] = {0, 0, 0, 0, 1, 1, 2, 2, 3, 3, 4, 4, 5, 5, 6, 6, 7, 7, 8, 8, 9, 9, 10, 10, 11, 11, 12, 12, 13, 13};
        if (s > 29) return F_BAD;
        return ((uint32_t)dbase[s] << 16) | ((uint32_t)dext[s] << 8);
    }

    bool fail(const char *why) { err = why; return false; }

    // Sequence lines are runs of literals with 2-3-bit codes: `multi[i]` holds the literals (up to four) whose codes fit the
    // LIT_BITS-bit index i together -- bytes 0..3 the literals, byte 4 how many (0: look the index up in `lit`), byte 5 the bits
    // they take -- so one lookup emits up to four bytes.
    uint64_t multi[1u << LIT_BITS];
    void build_multi()
    {
        for (uint32_t i = 0; i < (1u << LIT_BITS); i++) {
            uint64_t lits = 0;
            uint32_t n = 0, used = 0;
            while (n < 4) {
                const uint32_t e = lit[i >> used];
                if ((e & (F_LIT | F_SUB)) != F_LIT) break;
                const uint32_t l = e & 0x1Fu;
                if (used + l > (uint32_t)LIT_BITS) break;
                lits |= (uint64_t)((e >> 16) & 0xFFu) << (8 * n);
                n++; used += l;
            }
            multi[i] = lits | ((uint64_t)n << 32) | ((uint64_t)used << 40);
        }
    }

    bool read_header()
    {
        refill();
        if (bitcnt < 3) return fail("deflate stream ends inside a block header");
        last_block = peek(1) != 0;
        const uint32_t type = (peek(3) >> 1);
        drop(3);
        if (type == 0) {
            drop(bitcnt & 7u);                                   // to the byte boundary; whole bytes in the buffer go back to the input
            in -= bitcnt >> 3; bitbuf = 0; bitcnt = 0;
            if (in_end - in < 4) return fail("deflate stream ends inside a stored block header");
            const uint32_t len = (uint32_t)in[0] | ((uint32_t)in[1] << 8), nlen = (uint32_t)in[2] | ((uint32_t)in[3] << 8);
            if ((len ^ 0xFFFFu) != nlen) return fail("invalid stored block lengths");
            in += 4;
            stored_left = len;
            state = STORED;
            return true;
        }
        uint8_t lens[320];
        if (type == 1) {
            for (int i = 0; i < 144; i++) lens[i] = 8;
            for (int i = 144; i < 256; i++) lens[i] = 9;
            for (int i = 256; i < 280; i++) lens[i] = 7;
            for (int i = 280; i < 288; i++) lens[i] = 8;
            if (!build(lit, LIT_BITS, sizeof lit / 4, lens, 288, false, lit_entry)) return fail("internal: fixed code");
            for (int i = 0; i < 30; i++) lens[i] = 5;
            lens[30] = lens[31] = 5;                             // (two codes that never occur in valid data; they complete the code)
            if (!build(dist, DIST_BITS, sizeof dist / 4, lens, 32, false, dist_entry)) return fail("internal: fixed distance code");
            build_multi();
            state = CODED;
            return true;
        }
        if (type != 2) return fail("invalid block type");
        refill();
        if (bitcnt < 14) return fail("deflate stream ends inside a block header");
        const int hlit = (int)peek(5) + 257; drop(5);
        const int hdist = (int)peek(5) + 1; drop(5);
        const int hclen = (int)peek(4) + 4; drop(4);
        if (hlit > 286 || hdist > 30) return fail("too many length or distance symbols");
        static const uint8_t order[19] = {16, 17, 18, 0, 8, 7, 9, 6, 10, 5, 11, 4, 12, 3, 13, 2, 14, 1, 15};
        uint8_t cl[19] = {0};
        for (int i = 0; i < hclen; i++) {
            refill();
            if (bitcnt < 3) return fail("deflate stream ends inside a block header");
            cl[order[i]] = (uint8_t)peek(3);
            drop(3);
        }
        uint32_t cltab[128 + 8];
        if (!build(cltab, 7, sizeof cltab / 4, cl, 19, false, [](int s) { return (uint32_t)s << 16; })) return fail("invalid code lengths set");
        int i = 0;
        while (i < hlit + hdist) {
            refill();
            const uint32_t e = cltab[peek(7)];
            if (e & F_BAD) return fail("invalid code lengths set");
            if ((e & 0x1Fu) > bitcnt) return fail("deflate stream ends inside a block header");
            drop(e & 0x1Fu);
            const uint32_t sym = e >> 16;
            if (sym < 16) { lens[i++] = (uint8_t)sym; continue; }
            uint32_t rep, val = 0;
            if (sym == 16) {
                if (i == 0) return fail("invalid bit length repeat");
                val = lens[i - 1];
                if (bitcnt < 2) return fail("deflate stream ends inside a block header");
                rep = 3 + peek(2); drop(2);
            } else if (sym == 17) {
                if (bitcnt < 3) return fail("deflate stream ends inside a block header");
                rep = 3 + peek(3); drop(3);
            } else {
                if (bitcnt < 7) return fail("deflate stream ends inside a block header");
                rep = 11 + peek(7); drop(7);
            }
            if (i + (int)rep > hlit + hdist) return fail("invalid bit length repeat");
            while (rep--) lens[i++] = (uint8_t)val;
        }
        if (lens[256] == 0) return fail("invalid code -- missing end-of-block");
        if (!build(lit, LIT_BITS, sizeof lit / 4, lens, hlit, true, lit_entry)) return fail("invalid literal/lengths set");
        if (!build(dist, DIST_BITS, sizeof dist / 4, lens + hlit, hdist, true, dist_entry)) return fail("invalid distances set");
        build_multi();
        state = CODED;
        return true;
    }

    // Decode until `out` reaches `out_limit` (the caller keeps >= 320 bytes of slack behind it), the stream ends (state == DONE) or an
    // error (returns false, err set).  `lowest`: the first byte of valid history (a distance may not reach before it).
    bool run(uint8_t *&out, uint8_t *out_limit, const uint8_t *lowest)
    {
        while (out < out_limit) {
            if (state == DONE) return true;
            if (state == HEADER) { if (!read_header()) return false; continue; }
            if (state == STORED) {
                size_t n = stored_left;
                if ((size_t)(out_limit - out) < n) n = (size_t)(out_limit - out);
                if ((size_t)(in_end - in) < n) return fail("deflate stream ends inside a stored block");
                memcpy(out, in, n);
                out += n; in += n; stored_left -= (uint32_t)n;
                if (stored_left == 0) state = last_block ? DONE : HEADER;
                continue;
            }
            // Huffman-coded block
            for (;;) {
                if (out >= out_limit) return true;
                refill();
                {
                    // up to three lookups of up to four literals each on one refill (3 x 11 bits)
                    uint64_t m = multi[peek(LIT_BITS)];
                    if ((uint32_t)(m >> 32) & 0xFFu) {
                        if ((uint32_t)(m >> 40) > bitcnt) return fail("deflate stream ends inside a block");
                        uint32_t w = (uint32_t)m;
                        memcpy(out, &w, 4);
                        out += (uint32_t)(m >> 32) & 0xFFu;
                        drop((uint32_t)(m >> 40));
                        m = multi[peek(LIT_BITS)];
                        if (((uint32_t)(m >> 32) & 0xFFu) && bitcnt >= 32) {
                            w = (uint32_t)m;
                            memcpy(out, &w, 4);
                            out += (uint32_t)(m >> 32) & 0xFFu;
                            drop((uint32_t)(m >> 40));
                            m = multi[peek(LIT_BITS)];
                            if (((uint32_t)(m >> 32) & 0xFFu) && bitcnt >= 16) {
                                w = (uint32_t)m;
                                memcpy(out, &w, 4);
                                out += (uint32_t)(m >> 32) & 0xFFu;
                                drop((uint32_t)(m >> 40));
                            }
                        }
                        continue;
                    }
                }
                uint32_t e = lit[peek(LIT_BITS)];
                if (e & F_SUB) e = lit[(e >> 16) + (((uint32_t)(bitbuf >> LIT_BITS)) & ((1u << ((e >> 8) & 0xFu)) - 1u))];
                if ((e & 0x1Fu) > bitcnt) return fail("deflate stream ends inside a block");
                if (e & F_LIT) {                                 // (a literal whose code is longer than the index)
                    drop(e & 0x1Fu);
                    *out++ = (uint8_t)(e >> 16);
                    continue;
                }
                if (e & F_BAD) return fail("invalid literal/length code");
                drop(e & 0x1Fu);
                if (e & F_EOB) { state = last_block ? DONE : HEADER; break; }
                const uint32_t lx = (e >> 8) & 0xFu;
                if (lx > bitcnt) return fail("deflate stream ends inside a block");
                uint32_t len = (e >> 16) + peek(lx);
                drop(lx);
                uint32_t d = dist[peek(DIST_BITS)];
                if (d & F_SUB) d = dist[(d >> 16) + (((uint32_t)(bitbuf >> DIST_BITS)) & ((1u << ((d >> 8) & 0xFu)) - 1u))];
                if (d & F_BAD) return fail("invalid distance code");
                if ((d & 0x1Fu) > bitcnt) return fail("deflate stream ends inside a block");
                drop(d & 0x1Fu);
                const uint32_t dx = (d >> 8) & 0xFu;
                if (dx > bitcnt) return fail("deflate stream ends inside a block");
                const uint32_t distance = (d >> 16) + peek(dx);
                drop(dx);
                if ((size_t)(out - lowest) < distance) return fail("invalid distance too far back");
                const uint8_t *src = out - distance;
                uint8_t *dst = out;
                out += len;
                if (distance >= 8) {
                    // eight bytes at a time; may write up to 7 bytes past the match (slack), never reads ahead of what is written
                    do { uint64_t w; memcpy(&w, src, 8); memcpy(dst, &w, 8); src += 8; dst += 8; } while (dst < out);
                } else if (distance == 1) {
                    memset(dst, *src, len);
                } else {
                    do { *dst++ = *src++; } while (dst < out);
                }
            }
        }
        return true;
    }
};

}  // namespace kdbhost
