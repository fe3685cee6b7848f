// kdb_crc32.cpp.h -- CRC-32 of gzip / BGZF members (host side, no GPU): the .kdb writer frames 16 000 members per GiB of text and the
// BGZF reader checks as many; zlib 1.2.11's crc32 (about 1 GB/s) was a third of a writer thread's time.  Carry-less multiplication
// where the CPU has it (x86-64 PCLMULQDQ: ~10 x), tables otherwise; both checked against zlib's in tests/c/writer_check.cpp.
#pragma once
#include <cstddef>
#include <cstdint>
#include <cstring>
#if defined(__x86_64__)
#include <immintrin.h>
#endif

namespace kdbhost {

// ---- CRC-32 (the gzip polynomial, reflected), eight bytes per step ---------------------------------------------
struct Crc32Tables {
    uint32_t t[8][256];
    Crc32Tables()
    {
        for (uint32_t i = 0; i < 256; i++) {
            uint32_t c = i;
            for (int b = 0; b < 8; b++) c = (c & 1) ? 0xEDB88320u ^ (c >> 1) : c >> 1;
            t[0][i] = c;
        }
        for (uint32_t i = 0; i < 256; i++)
            for (int s = 1; s < 8; s++) t[s][i] = (t[s - 1][i] >> 8) ^ t[0][t[s - 1][i] & 0xFF];
    }
};
inline const Crc32Tables &crc32_tables() { static const Crc32Tables T; return T; }

// c: the running register (the CRC's complement), advanced over p[0, n) with the tables
inline uint32_t crc32_update_tables(uint32_t c, const uint8_t *p, size_t n)
{
    const Crc32Tables &T = crc32_tables();
    while (n && ((uintptr_t)p & 7u)) { c = (c >> 8) ^ T.t[0][(c ^ *p++) & 0xFF]; n--; }
    while (n >= 8) {
        uint64_t w;
        memcpy(&w, p, 8);
        w ^= c;                                                       // (little-endian host)
        c = T.t[7][w & 0xFF] ^ T.t[6][(w >> 8) & 0xFF] ^ T.t[5][(w >> 16) & 0xFF] ^ T.t[4][(w >> 24) & 0xFF] ^
            T.t[3][(w >> 32) & 0xFF] ^ T.t[2][(w >> 40) & 0xFF] ^ T.t[1][(w >> 48) & 0xFF] ^ T.t[0][w >> 56];
        p += 8; n -= 8;
    }
    while (n--) c = (c >> 8) ^ T.t[0][(c ^ *p++) & 0xFF];
    return c;
}

#if defined(__x86_64__)
// The same register over n >= 64 bytes, n a multiple of 16, by carry-less multiplication (Gopal et al., "Fast CRC computation
// for generic polynomials using PCLMULQDQ", Intel 2009): four 128-bit lanes folded 64 bytes at a time with x^(512+-32) mod P,
// then into one lane, down to 64 bits, and a Barrett reduction -- the constants are the paper's for the reflected 0xEDB88320.
__attribute__((target("pclmul,sse4.1"))) inline uint32_t crc32_update_clmul(uint32_t c, const uint8_t *p, size_t n)
{
    const __m128i k1k2 = _mm_set_epi64x(0x01c6e41596ll, 0x0154442bd4ll);
    const __m128i k3k4 = _mm_set_epi64x(0x00ccaa009ell, 0x01751997d0ll);
    const __m128i k5 = _mm_set_epi64x(0, 0x0163cd6124ll);
    const __m128i poly = _mm_set_epi64x(0x01f7011641ll, 0x01db710641ll);
    __m128i x1 = _mm_loadu_si128((const __m128i *)(p + 0)), x2 = _mm_loadu_si128((const __m128i *)(p + 16));
    __m128i x3 = _mm_loadu_si128((const __m128i *)(p + 32)), x4 = _mm_loadu_si128((const __m128i *)(p + 48));
    x1 = _mm_xor_si128(x1, _mm_cvtsi32_si128((int)c));
    p += 64; n -= 64;
    while (n >= 64) {
        const __m128i a1 = _mm_clmulepi64_si128(x1, k1k2, 0x00), a2 = _mm_clmulepi64_si128(x2, k1k2, 0x00);
        const __m128i a3 = _mm_clmulepi64_si128(x3, k1k2, 0x00), a4 = _mm_clmulepi64_si128(x4, k1k2, 0x00);
        x1 = _mm_clmulepi64_si128(x1, k1k2, 0x11); x2 = _mm_clmulepi64_si128(x2, k1k2, 0x11);
        x3 = _mm_clmulepi64_si128(x3, k1k2, 0x11); x4 = _mm_clmulepi64_si128(x4, k1k2, 0x11);
        x1 = _mm_xor_si128(_mm_xor_si128(x1, a1), _mm_loadu_si128((const __m128i *)(p + 0)));
        x2 = _mm_xor_si128(_mm_xor_si128(x2, a2), _mm_loadu_si128((const __m128i *)(p + 16)));
        x3 = _mm_xor_si128(_mm_xor_si128(x3, a3), _mm_loadu_si128((const __m128i *)(p + 32)));
        x4 = _mm_xor_si128(_mm_xor_si128(x4, a4), _mm_loadu_si128((const __m128i *)(p + 48)));
        p += 64; n -= 64;
    }
    __m128i a = _mm_clmulepi64_si128(x1, k3k4, 0x00);
    x1 = _mm_xor_si128(_mm_xor_si128(_mm_clmulepi64_si128(x1, k3k4, 0x11), x2), a);
    a = _mm_clmulepi64_si128(x1, k3k4, 0x00);
    x1 = _mm_xor_si128(_mm_xor_si128(_mm_clmulepi64_si128(x1, k3k4, 0x11), x3), a);
    a = _mm_clmulepi64_si128(x1, k3k4, 0x00);
    x1 = _mm_xor_si128(_mm_xor_si128(_mm_clmulepi64_si128(x1, k3k4, 0x11), x4), a);
    while (n >= 16) {
        a = _mm_clmulepi64_si128(x1, k3k4, 0x00);
        x1 = _mm_xor_si128(_mm_xor_si128(_mm_clmulepi64_si128(x1, k3k4, 0x11), _mm_loadu_si128((const __m128i *)p)), a);
        p += 16; n -= 16;
    }
    const __m128i mask32 = _mm_setr_epi32(~0, 0, ~0, 0);
    x2 = _mm_clmulepi64_si128(x1, k3k4, 0x10);
    x1 = _mm_xor_si128(_mm_srli_si128(x1, 8), x2);
    x2 = _mm_srli_si128(x1, 4);
    x1 = _mm_xor_si128(_mm_clmulepi64_si128(_mm_and_si128(x1, mask32), k5, 0x00), x2);
    x2 = _mm_clmulepi64_si128(_mm_and_si128(x1, mask32), poly, 0x10);
    x2 = _mm_clmulepi64_si128(_mm_and_si128(x2, mask32), poly, 0x00);
    x1 = _mm_xor_si128(x1, x2);
    return (uint32_t)_mm_extract_epi32(x1, 1);
}
inline bool crc32_have_clmul() { static const bool have = __builtin_cpu_supports("pclmul") && __builtin_cpu_supports("sse4.1"); return have; }
#endif

inline uint32_t crc32_bytes(const uint8_t *p, size_t n)
{
    uint32_t c = 0xFFFFFFFFu;
#if defined(__x86_64__)
    if (n >= 64 && crc32_have_clmul()) {
        const size_t m = n & ~(size_t)15;
        c = crc32_update_clmul(c, p, m);
        p += m; n -= m;
    }
#endif
    return ~crc32_update_tables(c, p, n);
}

}  // namespace kdbhost
