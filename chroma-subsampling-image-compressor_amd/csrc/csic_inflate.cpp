// csic_inflate.cpp -- zlib-stream decoder (RFC 1950 / 1951), CRC-32 and Adler-32 for the PNG reader (host only).
//
// The reader of csic_png.cpp stands in for scrimage's `ImmutableImage.loader().fromFile` (ImageProcessorModel.scala:14-16).
// From files, the whole path is bound by PNG decoding (profiles/r03_host_io.json), and of a 4K frame's decode time two
// thirds were zlib's inflate, a tenth each its crc32 and adler32.  This file replaces the three:
//   * inflate: 64-bit bit buffer refilled with one unaligned 8-byte load, an 11-bit literal/length table and an 8-bit
//     distance table with second-level tables behind them (one 32-bit entry decodes codeword + base + extra-bit count);
//     where two literal codewords fit the table index, ONE entry decodes both (a literal costs a dependent table load,
//     about 7 cycles: filtered image data is mostly literals of 3-6 bits, so pairs nearly halve that chain); up to
//     three lookups per refill; matches copied 8 bytes at a time (short distances widened to a multiple >= 8
//     first: distance 3 and 4 are the common ones behind PNG's Sub/Avg/Paeth residuals).  A bounds-checked loop takes
//     over near either end of the buffers, so no byte outside [in, in+n) or [out, out+n) is ever touched.
//   * accepts and rejects what zlib's inflate does: over-subscribed or incomplete code sets (the single 1-bit code
//     excepted), a missing end-of-block code, repeat without a previous length, too many symbols, distances beyond the
//     start of the output, stored-block length mismatch, header / dictionary / check-value errors; the output must be
//     EXACTLY the size the caller derived from IHDR (as the reader demanded of uncompress()).
//   * crc32: carry-less multiplication where the CPU has PCLMULQDQ, else slicing by 8;  adler32: SSSE3, 16 bytes per
//     step, else scalar (both checked at run time).
// Checked against zlib itself (uncompress / crc32 / adler32) on random streams of every block type and level, and on
// mutated streams, under ASan / UBSan in tests/cpp/host_sanitize.cpp.
#include <cstdint>
#include <cstdlib>
#include <cstring>

#if defined(__x86_64__)
#include <immintrin.h>
#endif

#include "csic_internal.h"

namespace {

constexpr int LL_BITS = 11, D_BITS = 8, CL_BITS = 7, MAX_CODE_LEN = 15;
constexpr int LL_CAP = (1 << LL_BITS) + 288 * (1 << (MAX_CODE_LEN - LL_BITS));
constexpr int D_CAP = (1 << D_BITS) + 32 * (1 << (MAX_CODE_LEN - D_BITS));
constexpr int CL_CAP = 1 << CL_BITS;

// entry: [31:16] literal (a pair: second literal in [31:24]) | length base | distance base | start of a second-level table
//        [15:12] kind   [9] literal pair
//        [8:5] bits of the codeword (a pair: of its first; a pointer to a second-level table: that table's index width)
//        [4:0] bits to drop: codeword + extra bits (both codewords of a pair; a pointer: the first-level index width)
constexpr uint32_t K_LIT = 1u << 12, K_EOB = 1u << 13, K_SUB = 1u << 14, K_BAD = 1u << 15, K_LIT2 = 1u << 9;

struct Tables {
    uint32_t ll[LL_CAP];
    uint32_t d[D_CAP];
};

uint32_t g_ll_sym[288], g_d_sym[32], g_cl_sym[19];      // per symbol: [31:16] value, [15:12] kind, [3:0] extra bits
Tables g_fixed;

inline unsigned bit_reverse(unsigned code, int len)
{
    unsigned r = 0;
    for (int i = 0; i < len; ++i) { r = (r << 1) | (code & 1); code >>= 1; }
    return r;
}

// Literal pairs: where the index bits behind a literal's codeword hold a second complete literal codeword, the entry
// takes both.  Descending, so that table[i >> l1] (always below i) is still the single-symbol entry when it is read.
void add_literal_pairs(uint32_t *table, int table_bits)
{
    for (int i = (1 << table_bits) - 1; i >= 0; --i) {
        const uint32_t e1 = table[i];
        if (!(e1 & K_LIT)) continue;
        const int l1 = (int)(e1 & 31);
        const uint32_t e2 = table[i >> l1];
        const int l2 = (int)(e2 & 31);
        if ((e2 & K_LIT) && l1 + l2 <= table_bits)
            table[i] = (e1 & 0x00FF0000u) | ((e2 & 0x00FF0000u) << 8) | K_LIT | K_LIT2 | ((uint32_t)l1 << 5) | (uint32_t)(l1 + l2);
    }
}

enum Strict { ANY_INCOMPLETE_IS_BAD, SINGLE_CODE_MAY_BE_INCOMPLETE };

// Canonical Huffman code -> lookup tables (RFC 1951 3.2.2).  false: over-subscribed or (see Strict) incomplete.
bool build_table(const uint8_t *lens, int n, int table_bits, const uint32_t *sym_entry, uint32_t *table, Strict strict)
{
    int count[MAX_CODE_LEN + 1] = {0};
    for (int s = 0; s < n; ++s) ++count[lens[s]];
    count[0] = 0;
    int max_len = 0;
    for (int l = 1; l <= MAX_CODE_LEN; ++l) if (count[l]) max_len = l;
    long left = 1;
    for (int l = 1; l <= MAX_CODE_LEN; ++l) {
        left = (left << 1) - count[l];
        if (left < 0) return false;
    }
    if (left > 0 && max_len > 0 && (strict == ANY_INCOMPLETE_IS_BAD || max_len != 1)) return false;
    unsigned next[MAX_CODE_LEN + 2], code = 0;
    for (int l = 1; l <= MAX_CODE_LEN; ++l) { code = (code + count[l - 1]) << 1; next[l] = code; }
    const int primary = 1 << table_bits;
    for (int i = 0; i < primary; ++i) table[i] = K_BAD;
    const int sub_bits = max_len > table_bits ? max_len - table_bits : 0;
    int sub_next = primary;
    for (int s = 0; s < n; ++s) {
        const int l = lens[s];
        if (!l) continue;
        const unsigned rev = bit_reverse(next[l]++, l);
        if (l <= table_bits) {
            const uint32_t e = (sym_entry[s] & 0xFFFFF000u) | ((uint32_t)l << 5) | ((uint32_t)l + (sym_entry[s] & 15));
            for (int i = (int)rev; i < primary; i += 1 << l) table[i] = e;
        } else {
            const unsigned prefix = rev & (unsigned)(primary - 1);
            if (!(table[prefix] & K_SUB)) {
                for (int i = 0; i < (1 << sub_bits); ++i) table[sub_next + i] = K_BAD;
                table[prefix] = ((uint32_t)sub_next << 16) | K_SUB | ((uint32_t)sub_bits << 5) | (uint32_t)table_bits;
                sub_next += 1 << sub_bits;
            }
            const int start = (int)(table[prefix] >> 16), rest = l - table_bits;
            const uint32_t e = (sym_entry[s] & 0xFFFFF000u) | ((uint32_t)rest << 5) | ((uint32_t)rest + (sym_entry[s] & 15));
            for (int i = (int)(rev >> table_bits); i < (1 << sub_bits); i += 1 << rest) table[start + i] = e;
        }
    }
    return true;
}

struct StaticInit {
    StaticInit()
    {
        static const uint16_t lbase[29] = {3, 4, 5, 6, 7, 8, 9, 10, 11, 13, 15, 17, 19, 23, 27, 31, 35, 43, 51, 59, 67, 83, 99, 115, 131, 163, 195, 227, 258};
        static const uint8_t lext[29] = {0, 0, 0, 0, 0, 0, 0, 0, 1, 1, 1, 1, 2, 2, 2, 2, 3, 3, 3, 3, 4, 4, 4, 4, 5, 5, 5, 5, 0};
        static const uint16_t dbase[30] = {1, 2, 3, 4, 5, 7, 9, 13, 17, 25, 33, 49, 65, 97, 129, 193, 257, 385, 513, 769, 1025, 1537, 2049, 3073, 4097, 6145, 8193, 12289, 16385, 24577};
        static const uint8_t dext[30] = {0, 0, 0, 0, 1, 1, 2, 2, 3, 3, 4, 4, 5, 5, 6, 6, 7, 7, 8, 8, 9, 9, 10, 10, 11, 11, 12, 12, 13, 13};
        for (int s = 0; s < 256; ++s) g_ll_sym[s] = ((uint32_t)s << 16) | K_LIT;
        g_ll_sym[256] = K_EOB;
        for (int s = 257; s < 286; ++s) g_ll_sym[s] = ((uint32_t)lbase[s - 257] << 16) | lext[s - 257];
        g_ll_sym[286] = g_ll_sym[287] = K_BAD;                  // in the fixed code, never valid in data
        for (int s = 0; s < 30; ++s) g_d_sym[s] = ((uint32_t)dbase[s] << 16) | dext[s];
        g_d_sym[30] = g_d_sym[31] = K_BAD;
        for (int s = 0; s < 19; ++s) g_cl_sym[s] = (uint32_t)s << 16;
        uint8_t lens[288 + 32];
        for (int s = 0; s < 288; ++s) lens[s] = s < 144 ? 8 : s < 256 ? 9 : s < 280 ? 7 : 8;
        for (int s = 0; s < 32; ++s) lens[288 + s] = 5;
        build_table(lens, 288, LL_BITS, g_ll_sym, g_fixed.ll, SINGLE_CODE_MAY_BE_INCOMPLETE);
        add_literal_pairs(g_fixed.ll, LL_BITS);
        build_table(lens + 288, 32, D_BITS, g_d_sym, g_fixed.d, SINGLE_CODE_MAY_BE_INCOMPLETE);
    }
};
const StaticInit g_static_init;

#if !defined(__BYTE_ORDER__) || __BYTE_ORDER__ != __ORDER_LITTLE_ENDIAN__
#error "the bit buffer of this decoder is refilled with little-endian 8-byte loads"
#endif
inline uint64_t load64(const unsigned char *p) { uint64_t v; std::memcpy(&v, p, 8); return v; }
inline void store64(unsigned char *p, uint64_t v) { std::memcpy(p, &v, 8); }
inline void store16(unsigned char *p, uint32_t v) { const uint16_t h = (uint16_t)v; std::memcpy(p, &h, 2); }

// The precise bit reader used for headers and near the ends of the buffers: `buf` holds exactly `bits` valid bits,
// `in` is the next unread byte.
struct Bits {
    const unsigned char *in, *end;
    uint64_t buf = 0;
    unsigned bits = 0;
    void fill() { while (bits < 56 && in < end) { buf |= (uint64_t)*in++ << bits; bits += 8; } }
    bool need(unsigned n) { if (bits < n) fill(); return bits >= n; }
    unsigned take(unsigned n) { const unsigned v = (unsigned)(buf & ((1ull << n) - 1)); buf >>= n; bits -= n; return v; }
    void unread_whole_bytes() { in -= bits >> 3; bits &= 7; buf &= (1ull << bits) - 1; }
};

enum { INF_OK = 0, INF_CORRUPT = 1, INF_TRUNCATED = 2, INF_TOO_LONG = 3 };

// One Huffman-coded block.  Fast loop while both buffers keep their margins, then the checked loop to the end-of-block code.
// An entry's low bits say how far to shift for codeword AND extra bits at once, so the extra bits are read from a copy of the
// bit buffer, off the chain  table load -> shift -> next table load  that bounds a match-heavy stream.
#define CSIC_EXTRA(saved, e) ((unsigned)(((saved) & ((1ull << ((e) & 31)) - 1)) >> (((e) >> 5) & 15)))
int inflate_block(Bits &br, const Tables &t, unsigned char *const out0, unsigned char *&outp, unsigned char *const out_end)
{
    unsigned char *out = outp;
    {
        const unsigned char *in = br.in;
        uint64_t bitbuf = br.buf, saved;
        unsigned bitsleft = br.bits;                                            // <= 63
#define CSIC_REFILL() do { bitbuf |= load64(in) << bitsleft; in += (63 - bitsleft) >> 3; bitsleft |= 56; } while (0)
#define CSIC_LL(e) do { e = t.ll[bitbuf & ((1u << LL_BITS) - 1)];                                                   \
                        if (e & K_SUB) { bitbuf >>= LL_BITS; bitsleft -= LL_BITS;                                    \
                                         e = t.ll[(e >> 16) + (bitbuf & ((1u << ((e >> 5) & 15)) - 1))]; }          \
                        saved = bitbuf; bitbuf >>= (e & 31); bitsleft -= (e & 31); } while (0)
#define CSIC_EMIT(e) do { store16(out, e >> 16); out += 1 + ((e >> 9) & 1); } while (0)   /* the pair's second byte, or a byte of slack */
        bool eob = false;
        while (br.end - in >= 16 && out_end - out >= 288) {
            uint32_t e;
            CSIC_REFILL();                                                      // >= 56 bits: three literal lookups (<= 15 each, the
            CSIC_LL(e);                                                         // last may be a length: <= 20), or length + distance (<= 48)
            if (e & K_LIT) {
                CSIC_EMIT(e);
                CSIC_LL(e);
                if (e & K_LIT) {
                    CSIC_EMIT(e);
                    CSIC_LL(e);
                    if (e & K_LIT) { CSIC_EMIT(e); continue; }
                }
                CSIC_REFILL();                                                  // this length's extra bits are in `saved`
            }
            if (e & (K_EOB | K_BAD)) {
                if (e & K_BAD) return INF_CORRUPT;
                eob = true;
                break;
            }
            const unsigned len = (e >> 16) + CSIC_EXTRA(saved, e);
            uint32_t d = t.d[bitbuf & ((1u << D_BITS) - 1)];
            if (d & K_SUB) {
                bitbuf >>= D_BITS; bitsleft -= D_BITS;
                d = t.d[(d >> 16) + (bitbuf & ((1u << ((d >> 5) & 15)) - 1))];
            }
            saved = bitbuf; bitbuf >>= (d & 31); bitsleft -= (d & 31);
            if (d & K_BAD) return INF_CORRUPT;
            const size_t dist = (d >> 16) + CSIC_EXTRA(saved, d);
            if (dist > (size_t)(out - out0)) return INF_CORRUPT;
            unsigned char *dst = out, *const stop = out + len;
            const unsigned char *src = out - dist;
            out = stop;
            if (dist >= 8) {
                store64(dst, load64(src));                                      // most matches are shorter than 8: no loop to mispredict
                if (len > 8) {
                    store64(dst + 8, load64(src + 8));
                    if (len > 16) { dst += 16; src += 16; do { store64(dst, load64(src)); dst += 8; src += 8; } while (dst < stop); }
                }
            } else if (dist == 1) {
                const uint64_t v = 0x0101010101010101ull * src[0];
                do { store64(dst, v); dst += 8; } while (dst < stop);
            } else {
                // Widen the distance byte by byte to its first multiple that is also a multiple of 8, then copy words at that
                // distance: each load then reads exactly what ONE earlier store wrote (a load that straddles two recent stores
                // cannot be forwarded and stalls: 1 GB/s instead of 8 on the long distance-3 runs of flat image areas).
                const size_t wide = dist * (8 >> (dist % 2 ? 0 : dist % 4 ? 1 : 2));
                size_t lead = wide - dist;
                while (lead-- && dst < stop) { *dst = *(dst - dist); ++dst; }
                src = dst - wide;
                while (dst < stop) { store64(dst, load64(src)); dst += 8; src += 8; }
            }
        }
#undef CSIC_EMIT
#undef CSIC_LL
#undef CSIC_REFILL
        br.in = in; br.buf = bitbuf; br.bits = bitsleft;
        br.unread_whole_bytes();
        if (eob) { outp = out; return INF_OK; }
    }
    for (;;) {
        br.fill();
        uint64_t saved = br.buf;
        uint32_t e = t.ll[saved & ((1u << LL_BITS) - 1)];
        unsigned used = 0;
        if (e & K_SUB) { used = LL_BITS; saved >>= LL_BITS; e = t.ll[(e >> 16) + (saved & ((1u << ((e >> 5) & 15)) - 1))]; }
        if (e & K_BAD) return br.bits < MAX_CODE_LEN && br.in == br.end ? INF_TRUNCATED : INF_CORRUPT;
        if ((e & K_LIT2) && ((e & 31) > br.bits || out_end - out < 2)) e = (e & 0x00FF0000u) | K_LIT | ((e >> 5) & 15);   // the first of the pair alone
        used += e & 31;
        if (used > br.bits) return INF_TRUNCATED;
        br.take(used);
        if (e & K_LIT) {
            if (out == out_end) return INF_TOO_LONG;
            *out++ = (unsigned char)(e >> 16);
            if (e & K_LIT2) *out++ = (unsigned char)(e >> 24);
            continue;
        }
        if (e & K_EOB) { outp = out; return INF_OK; }
        const unsigned len = (e >> 16) + CSIC_EXTRA(saved, e);
        br.fill();
        saved = br.buf;
        uint32_t d = t.d[saved & ((1u << D_BITS) - 1)];
        used = 0;
        if (d & K_SUB) { used = D_BITS; saved >>= D_BITS; d = t.d[(d >> 16) + (saved & ((1u << ((d >> 5) & 15)) - 1))]; }
        if (d & K_BAD) return br.bits < MAX_CODE_LEN && br.in == br.end ? INF_TRUNCATED : INF_CORRUPT;
        used += d & 31;
        if (used > br.bits) return INF_TRUNCATED;
        br.take(used);
        const size_t dist = (d >> 16) + CSIC_EXTRA(saved, d);
        if (dist > (size_t)(out - out0)) return INF_CORRUPT;
        if (len > (size_t)(out_end - out)) return INF_TOO_LONG;
        for (unsigned i = 0; i < len; ++i) { *out = *(out - dist); ++out; }
    }
}
#undef CSIC_EXTRA

int read_dynamic_tables(Bits &br, Tables &t)
{
    if (!br.need(14)) return INF_TRUNCATED;
    const unsigned hlit = br.take(5) + 257, hdist = br.take(5) + 1, hclen = br.take(4) + 4;
    if (hlit > 286 || hdist > 30) return INF_CORRUPT;
    static const uint8_t order[19] = {16, 17, 18, 0, 8, 7, 9, 6, 10, 5, 11, 4, 12, 3, 13, 2, 14, 1, 15};
    uint8_t cl_lens[19] = {0};
    for (unsigned i = 0; i < hclen; ++i) {
        if (!br.need(3)) return INF_TRUNCATED;
        cl_lens[order[i]] = (uint8_t)br.take(3);
    }
    uint32_t cl[CL_CAP];
    if (!build_table(cl_lens, 19, CL_BITS, g_cl_sym, cl, ANY_INCOMPLETE_IS_BAD)) return INF_CORRUPT;
    uint8_t lens[286 + 30];
    const unsigned total = hlit + hdist;
    for (unsigned i = 0; i < total;) {
        br.fill();
        const uint32_t e = cl[br.buf & ((1u << CL_BITS) - 1)];
        if (e & K_BAD) return br.bits < CL_BITS && br.in == br.end ? INF_TRUNCATED : INF_CORRUPT;
        if ((e & 31) > br.bits) return INF_TRUNCATED;
        br.take(e & 31);
        const unsigned sym = e >> 16;
        if (sym < 16) { lens[i++] = (uint8_t)sym; continue; }
        unsigned rep, val = 0;
        if (sym == 16) {
            if (i == 0) return INF_CORRUPT;
            if (!br.need(2)) return INF_TRUNCATED;
            val = lens[i - 1]; rep = 3 + br.take(2);
        } else if (sym == 17) {
            if (!br.need(3)) return INF_TRUNCATED;
            rep = 3 + br.take(3);
        } else {
            if (!br.need(7)) return INF_TRUNCATED;
            rep = 11 + br.take(7);
        }
        if (i + rep > total) return INF_CORRUPT;
        while (rep--) lens[i++] = (uint8_t)val;
    }
    if (lens[256] == 0) return INF_CORRUPT;                                     // no end-of-block code
    if (!build_table(lens, (int)hlit, LL_BITS, g_ll_sym, t.ll, SINGLE_CODE_MAY_BE_INCOMPLETE)) return INF_CORRUPT;
    add_literal_pairs(t.ll, LL_BITS);
    if (!build_table(lens + hlit, (int)hdist, D_BITS, g_d_sym, t.d, SINGLE_CODE_MAY_BE_INCOMPLETE)) return INF_CORRUPT;
    return INF_OK;
}

uint32_t g_crc_table[8][256];
struct CrcInit {
    CrcInit()
    {
        for (uint32_t i = 0; i < 256; ++i) {
            uint32_t c = i;
            for (int k = 0; k < 8; ++k) c = (c & 1) ? 0xEDB88320u ^ (c >> 1) : c >> 1;
            g_crc_table[0][i] = c;
        }
        for (uint32_t i = 0; i < 256; ++i)
            for (int k = 1; k < 8; ++k) g_crc_table[k][i] = g_crc_table[0][g_crc_table[k - 1][i] & 0xFF] ^ (g_crc_table[k - 1][i] >> 8);
    }
};
const CrcInit g_crc_init;

constexpr uint32_t ADLER_MOD = 65521;

uint32_t adler32_scalar(uint32_t adler, const unsigned char *p, size_t n)
{
    uint32_t s1 = adler & 0xFFFF, s2 = adler >> 16;
    while (n) {
        size_t k = n < 5552 ? n : 5552;                                         // largest run that cannot overflow 32 bits
        n -= k;
        while (k--) { s1 += *p++; s2 += s1; }
        s1 %= ADLER_MOD; s2 %= ADLER_MOD;
    }
    return (s2 << 16) | s1;
}

#if defined(__x86_64__)
__attribute__((target("ssse3"))) uint32_t adler32_ssse3(uint32_t adler, const unsigned char *p, size_t n)
{
    uint32_t s1 = adler & 0xFFFF, s2 = adler >> 16;
    const __m128i tap = _mm_setr_epi8(16, 15, 14, 13, 12, 11, 10, 9, 8, 7, 6, 5, 4, 3, 2, 1);
    const __m128i zero = _mm_setzero_si128(), ones = _mm_set1_epi16(1);
    while (n >= 16) {
        size_t k = n < 4096 ? (n & ~(size_t)15) : 4096;                         // 256 steps: every 32-bit lane stays below 2^31
        n -= k;
        __m128i v_s1 = zero, v_s2 = zero, v_ps = zero;
        const uint32_t s1_before = s1;
        for (size_t i = 0; i < k; i += 16) {
            const __m128i v = _mm_loadu_si128((const __m128i *)(p + i));
            v_ps = _mm_add_epi32(v_ps, v_s1);
            v_s1 = _mm_add_epi32(v_s1, _mm_sad_epu8(v, zero));
            v_s2 = _mm_add_epi32(v_s2, _mm_madd_epi16(_mm_maddubs_epi16(v, tap), ones));
        }
        v_s2 = _mm_add_epi32(v_s2, _mm_slli_epi32(v_ps, 4));
        v_s1 = _mm_add_epi32(v_s1, _mm_shuffle_epi32(v_s1, _MM_SHUFFLE(1, 0, 3, 2)));
        v_s2 = _mm_add_epi32(v_s2, _mm_shuffle_epi32(v_s2, _MM_SHUFFLE(1, 0, 3, 2)));
        v_s2 = _mm_add_epi32(v_s2, _mm_shuffle_epi32(v_s2, _MM_SHUFFLE(2, 3, 0, 1)));
        s2 = (uint32_t)(((uint64_t)s2 + (uint64_t)k * s1_before + (uint32_t)_mm_cvtsi128_si32(v_s2)) % ADLER_MOD);
        s1 = (s1 + (uint32_t)_mm_cvtsi128_si32(v_s1)) % ADLER_MOD;
        p += k;
    }
    return adler32_scalar((s2 << 16) | s1, p, n);
}
#endif

#if defined(__x86_64__)
// CRC-32 by carry-less multiplication (Gopal et al., "Fast CRC Computation for Generic Polynomials Using PCLMULQDQ
// Instruction", Intel 2009): four 128-bit lanes folded over 64 bytes per step with x^(512+-32) mod P, then folded into
// one, reduced to 64 bits and to 32 by Barrett reduction.  The constants are those of the reflected polynomial
// 0xEDB88320; n must be a multiple of 16 and at least 64; `c` and the result are the raw (inverted) register.
__attribute__((target("pclmul,sse4.1"))) uint32_t crc32_clmul(uint32_t c, const unsigned char *p, size_t n)
{
    const __m128i k1k2 = _mm_set_epi64x(0x01c6e41596ll, 0x0154442bd4ll), k3k4 = _mm_set_epi64x(0x00ccaa009ell, 0x01751997d0ll);
    const __m128i k5 = _mm_set_epi64x(0, 0x0163cd6124ll), poly = _mm_set_epi64x(0x01f7011641ll, 0x01db710641ll);
    const __m128i *v = (const __m128i *)p;
    __m128i x1 = _mm_xor_si128(_mm_loadu_si128(v), _mm_cvtsi32_si128((int)c)), x2 = _mm_loadu_si128(v + 1), x3 = _mm_loadu_si128(v + 2),
            x4 = _mm_loadu_si128(v + 3);
    v += 4; n -= 64;
    for (; n >= 64; v += 4, n -= 64) {
        const __m128i l1 = _mm_clmulepi64_si128(x1, k1k2, 0x00), l2 = _mm_clmulepi64_si128(x2, k1k2, 0x00),
                      l3 = _mm_clmulepi64_si128(x3, k1k2, 0x00), l4 = _mm_clmulepi64_si128(x4, k1k2, 0x00);
        x1 = _mm_xor_si128(_mm_xor_si128(_mm_clmulepi64_si128(x1, k1k2, 0x11), l1), _mm_loadu_si128(v));
        x2 = _mm_xor_si128(_mm_xor_si128(_mm_clmulepi64_si128(x2, k1k2, 0x11), l2), _mm_loadu_si128(v + 1));
        x3 = _mm_xor_si128(_mm_xor_si128(_mm_clmulepi64_si128(x3, k1k2, 0x11), l3), _mm_loadu_si128(v + 2));
        x4 = _mm_xor_si128(_mm_xor_si128(_mm_clmulepi64_si128(x4, k1k2, 0x11), l4), _mm_loadu_si128(v + 3));
    }
#define CSIC_FOLD(acc, next) _mm_xor_si128(_mm_xor_si128(_mm_clmulepi64_si128(acc, k3k4, 0x11), _mm_clmulepi64_si128(acc, k3k4, 0x00)), next)
    x1 = CSIC_FOLD(x1, x2); x1 = CSIC_FOLD(x1, x3); x1 = CSIC_FOLD(x1, x4);
    for (; n >= 16; ++v, n -= 16) x1 = CSIC_FOLD(x1, _mm_loadu_si128(v));
#undef CSIC_FOLD
    const __m128i mask32 = _mm_setr_epi32(-1, 0, -1, 0);
    x2 = _mm_clmulepi64_si128(x1, k3k4, 0x10);
    x1 = _mm_xor_si128(_mm_srli_si128(x1, 8), x2);                               // 128 -> 96 bits
    x2 = _mm_srli_si128(x1, 4);
    x1 = _mm_xor_si128(_mm_clmulepi64_si128(_mm_and_si128(x1, mask32), k5, 0x00), x2);   // -> 64 bits
    x2 = _mm_and_si128(_mm_clmulepi64_si128(_mm_and_si128(x1, mask32), poly, 0x10), mask32);
    x1 = _mm_xor_si128(x1, _mm_clmulepi64_si128(x2, poly, 0x00));                // Barrett
    return (uint32_t)_mm_extract_epi32(x1, 1);
}
#endif

} // namespace

namespace csic {

uint32_t crc32_update(uint32_t crc, const unsigned char *p, size_t n)
{
    uint32_t c = ~crc;
#if defined(__x86_64__)
    static const bool have_clmul = __builtin_cpu_supports("pclmul") && __builtin_cpu_supports("sse4.1") && !std::getenv("CSIC_NO_SIMD");   // (the variable: tests of the portable paths)
    if (have_clmul && n >= 64) {
        const size_t body = n & ~(size_t)15;
        c = crc32_clmul(c, p, body);
        p += body; n -= body;
    }
#endif
    while (n && ((uintptr_t)p & 7)) { c = g_crc_table[0][(c ^ *p++) & 0xFF] ^ (c >> 8); --n; }
    while (n >= 8) {
        const uint64_t v = load64(p) ^ c;
        c = g_crc_table[7][v & 0xFF] ^ g_crc_table[6][(v >> 8) & 0xFF] ^ g_crc_table[5][(v >> 16) & 0xFF] ^ g_crc_table[4][(v >> 24) & 0xFF] ^
            g_crc_table[3][(v >> 32) & 0xFF] ^ g_crc_table[2][(v >> 40) & 0xFF] ^ g_crc_table[1][(v >> 48) & 0xFF] ^ g_crc_table[0][v >> 56];
        p += 8; n -= 8;
    }
    while (n--) c = g_crc_table[0][(c ^ *p++) & 0xFF] ^ (c >> 8);
    return ~c;
}

uint32_t adler32_update(uint32_t adler, const unsigned char *p, size_t n)
{
#if defined(__x86_64__)
    static const bool have_ssse3 = __builtin_cpu_supports("ssse3") && !std::getenv("CSIC_NO_SIMD");
    if (have_ssse3) return adler32_ssse3(adler, p, n);
#endif
    return adler32_scalar(adler, p, n);
}

// Decodes the zlib stream in[0, in_len) into out[0, out_len); succeeds (0) only if the stream is well formed, ends with
// the right Adler-32 and holds exactly out_len bytes.  Non-zero: 1 corrupt, 2 truncated, 3 more data than out_len, 4 less.
int zlib_decode_exact(const unsigned char *in, size_t in_len, unsigned char *out, size_t out_len)
{
    if (in_len < 2) return INF_TRUNCATED;
    if ((in[0] & 0x0F) != 8 || (in[0] >> 4) > 7 || ((in[0] << 8) | in[1]) % 31 != 0 || (in[1] & 0x20)) return INF_CORRUPT;
    Bits br;
    br.in = in + 2; br.end = in + in_len;
    unsigned char *o = out, *const out_end = out + out_len;
    Tables dyn;
    for (;;) {
        if (!br.need(3)) return INF_TRUNCATED;
        const unsigned last = br.take(1), type = br.take(2);
        if (type == 0) {
            br.take(br.bits & 7);
            if (!br.need(32)) return INF_TRUNCATED;
            const unsigned len = br.take(16), nlen = br.take(16);
            if ((len ^ nlen) != 0xFFFF) return INF_CORRUPT;
            br.unread_whole_bytes();                                            // bits is a multiple of 8 here: nothing stays behind
            if ((size_t)(br.end - br.in) < len) return INF_TRUNCATED;
            if ((size_t)(out_end - o) < len) return INF_TOO_LONG;
            if (len) std::memcpy(o, br.in, len);
            o += len; br.in += len;
        } else if (type == 1) {
            const int st = inflate_block(br, g_fixed, out, o, out_end);
            if (st != INF_OK) return st;
        } else if (type == 2) {
            int st = read_dynamic_tables(br, dyn);
            if (st != INF_OK) return st;
            st = inflate_block(br, dyn, out, o, out_end);
            if (st != INF_OK) return st;
        } else {
            return INF_CORRUPT;
        }
        if (last) break;
    }
    br.take(br.bits & 7);
    br.unread_whole_bytes();
    if (br.end - br.in < 4) return INF_TRUNCATED;
    if (o != out_end) return 4;
    const uint32_t want = ((uint32_t)br.in[0] << 24) | ((uint32_t)br.in[1] << 16) | ((uint32_t)br.in[2] << 8) | br.in[3];
    return adler32_update(1, out, out_len) == want ? INF_OK : INF_CORRUPT;
}

} // namespace csic
