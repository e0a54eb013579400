// csic_png.cpp -- minimal PNG codec for the host I/O either side of the hot path (host only; reading: own inflate /
// CRC / Adler, csic_inflate.cpp; writing: zlib's deflate).
//
// Stands in for the scrimage calls of the reference's helper object:
//   ImmutableImage.loader().fromFile(file)              ImageProcessorModel.scala:14-16
//   image.output(new PngWriter(), outputFile)           ImageProcessorModel.scala:18-22
// and for `pixel.red()/green()/blue()` (:47-48): decoding yields straight 8-bit samples packed as ARGB
// ints; input alpha is dropped and ancillary chunks (gAMA, cHRM, ...) are NOT applied -- the reference's
// golden images pin exactly that behaviour.  The decoder writes directly into a caller
// buffer, which may be a pinned staging buffer from csic_pipeline_acquire_input.
//
// Supported: non-interlaced PNG, colour types 0/2/3/4/6, bit depths 1/2/4/8/16 (16-bit samples keep
// their high byte).  The encoder writes 8-bit RGB (what the reference's spec dumps are:
// BufferedImage.TYPE_INT_RGB, ChromaSubsamplerImageSpec.scala:88) with per-row adaptive filtering.
//
// The reader is what bounds the path from files (profiles/r03_host_io.json), so 8-bit RGB / RGBA -- what the reference's
// inputs and every encoder's default are -- have their own row loops: the Sub / Average filters carry the previous
// pixel in registers instead of re-loading the bytes just stored, Paeth works on a whole pixel in one SSE2 register (runs of
// Average / Paeth rows: two or four rows at a time, skewed by a pixel, so that several dependency chains are in flight),
// and the conversion to ARGB words is one byte shuffle per four pixels where the CPU has SSSE3.  Everything else
// (grey, palette, 16-bit, sub-byte depths) takes the plain byte loops.
#include <zlib.h>

#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <atomic>
#include <memory>
#include <new>
#include <thread>
#include <utility>
#include <vector>

#if defined(__x86_64__)
#include <immintrin.h>
#endif

#include "csic_internal.h"
#include "csic_trace.h"

using namespace csic;

namespace {

struct Header {
    uint32_t w = 0, h = 0;
    int depth = 0, ctype = 0, interlace = 0;
};

uint32_t be32(const unsigned char *p) { return ((uint32_t)p[0] << 24) | ((uint32_t)p[1] << 16) | ((uint32_t)p[2] << 8) | p[3]; }

const unsigned char kSig[8] = {0x89, 'P', 'N', 'G', 0x0D, 0x0A, 0x1A, 0x0A};

struct FileBytes {                                               // a whole file, not zero-filled first
    std::unique_ptr<unsigned char[]> mem;
    size_t len = 0;
    const unsigned char *data() const { return mem.get(); }
    size_t size() const { return len; }
    const unsigned char &operator[](size_t i) const { return mem[i]; }
};

int read_file(const char *path, FileBytes &buf)
{
    FILE *fp = std::fopen(path, "rb");
    if (!fp) return set_error(CSIC_EIO, "cannot open %s", path);
    std::fseek(fp, 0, SEEK_END);
    long n = std::ftell(fp);
    std::fseek(fp, 0, SEEK_SET);
    if (n < 0) { std::fclose(fp); return set_error(CSIC_EIO, "cannot size %s", path); }
    buf.mem.reset(new unsigned char[(size_t)n + 1]);
    buf.len = (size_t)n;
    size_t got = n ? std::fread(buf.mem.get(), 1, (size_t)n, fp) : 0;
    std::fclose(fp);
    if (got != (size_t)n) return set_error(CSIC_EIO, "short read on %s", path);
    return CSIC_OK;
}

typedef std::vector<std::pair<const unsigned char *, size_t>> Spans;

// Walks the chunk list: fills the header, the palette and the list of IDAT payloads (pointers into `f`).
int parse(const char *path, const FileBytes &f, Header &hd, std::vector<unsigned char> *plte, Spans *idat)
{
    if (f.size() < 8 + 25 || std::memcmp(f.data(), kSig, 8) != 0) return set_error(CSIC_EFORMAT, "%s is not a PNG file", path);
    size_t pos = 8;
    bool have_ihdr = false, have_iend = false;
    while (pos + 12 <= f.size()) {
        const uint32_t len = be32(&f[pos]);
        const unsigned char *type = &f[pos + 4];
        if (pos + 12 + (size_t)len > f.size()) return set_error(CSIC_EFORMAT, "%s: truncated chunk", path);
        const unsigned char *data = &f[pos + 8];
        const uint32_t crc = be32(&f[pos + 8 + len]);
        if (crc32_update(0, type, 4 + (size_t)len) != crc) return set_error(CSIC_EFORMAT, "%s: chunk CRC mismatch", path);
        if (!std::memcmp(type, "IHDR", 4)) {
            if (len != 13) return set_error(CSIC_EFORMAT, "%s: bad IHDR", path);
            hd.w = be32(data); hd.h = be32(data + 4);
            hd.depth = data[8]; hd.ctype = data[9]; hd.interlace = data[12];
            if (data[10] != 0 || data[11] != 0) return set_error(CSIC_EFORMAT, "%s: unknown compression/filter method", path);
            have_ihdr = true;
        } else if (!std::memcmp(type, "PLTE", 4)) {
            if (plte) plte->assign(data, data + len);
        } else if (!std::memcmp(type, "IDAT", 4)) {
            if (idat && len) idat->emplace_back(data, (size_t)len);
        } else if (!std::memcmp(type, "IEND", 4)) {
            have_iend = true;
            break;
        }
        pos += 12 + (size_t)len;
    }
    if (!have_ihdr || !have_iend) return set_error(CSIC_EFORMAT, "%s: missing IHDR or IEND", path);
    if (hd.w == 0 || hd.h == 0 || hd.w > 0x7FFFFFFFu || hd.h > 0x7FFFFFFFu) return set_error(CSIC_EFORMAT, "%s: bad dimensions", path);
    static const int ok_depth[7][6] = {{1, 2, 4, 8, 16, 0}, {0}, {8, 16, 0}, {1, 2, 4, 8, 0}, {8, 16, 0}, {0}, {8, 16, 0}};
    bool ok = false;
    if (hd.ctype >= 0 && hd.ctype <= 6)
        for (int k = 0; k < 6 && ok_depth[hd.ctype][k]; ++k) ok |= ok_depth[hd.ctype][k] == hd.depth;
    if (!ok) return set_error(CSIC_EFORMAT, "%s: unsupported colour type %d / bit depth %d", path, hd.ctype, hd.depth);
    if (hd.interlace != 0) return set_error(CSIC_EFORMAT, "%s: interlaced PNG is not supported", path);
    return CSIC_OK;
}

int channels(int ctype) { return ctype == 0 ? 1 : ctype == 2 ? 3 : ctype == 3 ? 1 : ctype == 4 ? 2 : 4; }

inline int paeth(int a, int b, int c)
{
    const int p = a + b - c, pa = std::abs(p - a), pb = std::abs(p - b), pc = std::abs(p - c);
    return (pa <= pb && pa <= pc) ? a : (pb <= pc ? b : c);
}

// ---- 8-bit RGB / RGBA rows: BPP = 3 / 4 bytes per pixel -------------------------------------------------------------
template <int BPP> void unfilter_sub(unsigned char *cur, size_t npx)
{
    unsigned a[BPP] = {0};
    for (size_t x = 0; x < npx; ++x, cur += BPP)
        for (int c = 0; c < BPP; ++c) { a[c] = (a[c] + cur[c]) & 255u; cur[c] = (unsigned char)a[c]; }
}

template <int BPP> void unfilter_avg(unsigned char *cur, const unsigned char *prev, size_t npx)
{
    unsigned a[BPP] = {0};
    for (size_t x = 0; x < npx; ++x, cur += BPP, prev += BPP)
        for (int c = 0; c < BPP; ++c) { a[c] = (cur[c] + ((a[c] + prev[c]) >> 1)) & 255u; cur[c] = (unsigned char)a[c]; }
}

// One pixel per step, its channels in the 16-bit lanes of one register.  Reads 4 bytes per pixel: with BPP = 3 the byte
// behind a row is the next row's filter byte or the buffer's padding (never stored to; its lane is not used).
template <int BPP> void unfilter_paeth(unsigned char *cur, const unsigned char *prev, size_t npx)
{
#if defined(__x86_64__)
    const __m128i zero = _mm_setzero_si128(), low = _mm_set1_epi16(0xFF);
    __m128i a = zero, c = zero;
    for (size_t x = 0; x < npx; ++x, cur += BPP, prev += BPP) {
        uint32_t wb, wr;
        std::memcpy(&wb, prev, 4); std::memcpy(&wr, cur, 4);
        const __m128i b = _mm_unpacklo_epi8(_mm_cvtsi32_si128((int)wb), zero), r = _mm_unpacklo_epi8(_mm_cvtsi32_si128((int)wr), zero);
        __m128i pa = _mm_sub_epi16(b, c), pb = _mm_sub_epi16(a, c);                       // p - a, p - b with p = a + b - c
        __m128i pc = _mm_add_epi16(pa, pb);
        pa = _mm_max_epi16(pa, _mm_sub_epi16(zero, pa));
        pb = _mm_max_epi16(pb, _mm_sub_epi16(zero, pb));
        pc = _mm_max_epi16(pc, _mm_sub_epi16(zero, pc));
        const __m128i least = _mm_min_epi16(pc, _mm_min_epi16(pa, pb));
        const __m128i is_a = _mm_cmpeq_epi16(least, pa), is_b = _mm_cmpeq_epi16(least, pb);   // ties: a before b before c
        const __m128i b_or_c = _mm_or_si128(_mm_and_si128(is_b, b), _mm_andnot_si128(is_b, c));
        const __m128i pred = _mm_or_si128(_mm_and_si128(is_a, a), _mm_andnot_si128(is_a, b_or_c));
        a = _mm_and_si128(_mm_add_epi16(r, pred), low);
        c = b;
        const uint32_t o = (uint32_t)_mm_cvtsi128_si32(_mm_packus_epi16(a, a));
        std::memcpy(cur, &o, BPP);
    }
#else
    int a[BPP] = {0}, c[BPP] = {0};
    for (size_t x = 0; x < npx; ++x, cur += BPP, prev += BPP)
        for (int k = 0; k < BPP; ++k) { a[k] = (cur[k] + paeth(a[k], prev[k], c[k])) & 255; c[k] = prev[k]; cur[k] = (unsigned char)a[k]; }
#endif
}

template <int BPP> void to_argb_scalar(const unsigned char *px, uint32_t *out, size_t npx)
{
    for (size_t x = 0; x < npx; ++x, px += BPP) out[x] = 0xFF000000u | ((uint32_t)px[0] << 16) | ((uint32_t)px[1] << 8) | px[2];
}

#if defined(__x86_64__)
// 4 pixels per step from one 16-byte load (BPP = 3: 12 of them used, so the steps stop 4 bytes short of the row's end).
template <int BPP> __attribute__((target("ssse3"))) void to_argb_ssse3(const unsigned char *px, uint32_t *out, size_t npx)
{
    const __m128i pick = BPP == 3 ? _mm_setr_epi8(2, 1, 0, -1, 5, 4, 3, -1, 8, 7, 6, -1, 11, 10, 9, -1)
                                  : _mm_setr_epi8(2, 1, 0, -1, 6, 5, 4, -1, 10, 9, 8, -1, 14, 13, 12, -1);
    const __m128i alpha = _mm_set1_epi32((int)0xFF000000u);
    size_t x = 0;
    for (; (x + 4) * BPP + (16 - 4 * BPP) <= npx * BPP; x += 4)
        _mm_storeu_si128((__m128i *)(out + x), _mm_or_si128(_mm_shuffle_epi8(_mm_loadu_si128((const __m128i *)(px + x * BPP)), pick), alpha));
    to_argb_scalar<BPP>(px + x * BPP, out + x, npx - x);
}
#endif

#if defined(__x86_64__)
// Average and Paeth are serial in x (about 4 and 11 cycles of dependent arithmetic per pixel), but row y + 1 at pixel x - 1
// needs of row y only pixels x - 1 and x - 2: two consecutive rows of these filters run as two independent chains in one
// loop, one pixel apart, and row y + 1 takes its "above" and "above left" from the registers row y has just produced.
template <int FT> inline __m128i filter_step(__m128i r, __m128i a, __m128i b, __m128i c)   // 16-bit lanes; FT: 3 Average, 4 Paeth
{
    const __m128i zero = _mm_setzero_si128(), low = _mm_set1_epi16(0xFF);
    if (FT == 3) return _mm_and_si128(_mm_add_epi16(r, _mm_srli_epi16(_mm_add_epi16(a, b), 1)), low);
    __m128i pa = _mm_sub_epi16(b, c), pb = _mm_sub_epi16(a, c);
    __m128i pc = _mm_add_epi16(pa, pb);
    pa = _mm_max_epi16(pa, _mm_sub_epi16(zero, pa));
    pb = _mm_max_epi16(pb, _mm_sub_epi16(zero, pb));
    pc = _mm_max_epi16(pc, _mm_sub_epi16(zero, pc));
    const __m128i least = _mm_min_epi16(pc, _mm_min_epi16(pa, pb));
    const __m128i is_a = _mm_cmpeq_epi16(least, pa), is_b = _mm_cmpeq_epi16(least, pb);
    const __m128i b_or_c = _mm_or_si128(_mm_and_si128(is_b, b), _mm_andnot_si128(is_b, c));
    return _mm_and_si128(_mm_add_epi16(r, _mm_or_si128(_mm_and_si128(is_a, a), _mm_andnot_si128(is_a, b_or_c))), low);
}

inline __m128i load_px(const unsigned char *p) { uint32_t w; std::memcpy(&w, p, 4); return _mm_unpacklo_epi8(_mm_cvtsi32_si128((int)w), _mm_setzero_si128()); }
template <int BPP> inline void store_px(unsigned char *p, __m128i v) { const uint32_t o = (uint32_t)_mm_cvtsi128_si32(_mm_packus_epi16(v, v)); std::memcpy(p, &o, BPP); }

template <int BPP, int FT0, int FT1> void unfilter_two_rows(unsigned char *cur0, unsigned char *cur1, const unsigned char *prev, size_t npx)
{
    const __m128i zero = _mm_setzero_si128();
    __m128i c0 = load_px(prev), a0 = filter_step<FT0>(load_px(cur0), zero, c0, zero);    // row 0, pixel 0
    store_px<BPP>(cur0, a0);
    __m128i a1 = zero, c1 = zero;
    for (size_t x = 1; x < npx; ++x) {
        const __m128i b0 = load_px(prev + x * BPP);
        const __m128i n0 = filter_step<FT0>(load_px(cur0 + x * BPP), a0, b0, c0);       // row 0, pixel x
        a1 = filter_step<FT1>(load_px(cur1 + (x - 1) * BPP), a1, a0, c1);               // row 1, pixel x - 1: above = a0, above left = c1
        store_px<BPP>(cur0 + x * BPP, n0);
        store_px<BPP>(cur1 + (x - 1) * BPP, a1);
        c1 = a0; c0 = b0; a0 = n0;
    }
    a1 = filter_step<FT1>(load_px(cur1 + (npx - 1) * BPP), a1, a0, c1);
    store_px<BPP>(cur1 + (npx - 1) * BPP, a1);
}

// Four rows of ONE filter type, two to a register: rows 0 | 1 in the halves of A, rows 2 | 3 in B, row r at pixel x - r.  One
// filter_step then serves two rows, and the two registers are two chains in flight: the arithmetic of four pixels in the time the
// chain of one takes.  A row's "above" is the row before it one step earlier -- the other half of the register, or the other
// register -- and its "above left" is the "above" of the step before.  Rows that have not started yet are kept at zero (what
// lies left of the image); rows that have ended feed only rows that have ended too.
template <int BPP, int FT> void unfilter_four_rows(unsigned char *const cur[4], const unsigned char *prev, size_t npx)
{
    const __m128i zero = _mm_setzero_si128();
    const __m128i lo_half = _mm_set_epi64x(0, -1);
    auto load2 = [&](const unsigned char *p, const unsigned char *q) {
        uint32_t a, b;
        std::memcpy(&a, p, 4); std::memcpy(&b, q, 4);
        return _mm_unpacklo_epi8(_mm_unpacklo_epi32(_mm_cvtsi32_si128((int)a), _mm_cvtsi32_si128((int)b)), zero);
    };
    auto store = [&](int r, ptrdiff_t x, uint32_t v) { if (x >= 0 && x < (ptrdiff_t)npx) std::memcpy(cur[r] + x * BPP, &v, BPP); };
    __m128i A = zero, B = zero, cA = zero, cB = zero;
    const ptrdiff_t n = (ptrdiff_t)npx;
    for (ptrdiff_t i = 0; i < n + 3; ++i) {
        const __m128i rA = load2(cur[0] + i * BPP, cur[1] + (i - 1) * BPP), rB = load2(cur[2] + (i - 2) * BPP, cur[3] + (i - 3) * BPP);
        const __m128i bA = _mm_unpacklo_epi64(load_px(prev + i * BPP), A);                                   // [ row above | row 0 one step ago ]
        const __m128i bB = _mm_castpd_si128(_mm_shuffle_pd(_mm_castsi128_pd(A), _mm_castsi128_pd(B), 1));   // [ row 1 | row 2 ], one step ago
        __m128i nA = filter_step<FT>(rA, A, bA, cA), nB = filter_step<FT>(rB, B, bB, cB);
        if (i < 3) {                                                                                          // rows 1..3 start one step apart
            if (i == 0) { nA = _mm_and_si128(nA, lo_half); nB = zero; }
            else if (i == 1) nB = zero;
            else nB = _mm_and_si128(nB, lo_half);
        }
        cA = bA; cB = bB; A = nA; B = nB;
        const __m128i pa = _mm_packus_epi16(nA, nA), pb = _mm_packus_epi16(nB, nB);
        if (i >= 3 && i < n) {                                                                                // all four rows inside the image
            const uint32_t v0 = (uint32_t)_mm_cvtsi128_si32(pa), v1 = (uint32_t)_mm_cvtsi128_si32(_mm_srli_si128(pa, 4));
            const uint32_t v2 = (uint32_t)_mm_cvtsi128_si32(pb), v3 = (uint32_t)_mm_cvtsi128_si32(_mm_srli_si128(pb, 4));
            std::memcpy(cur[0] + i * BPP, &v0, BPP); std::memcpy(cur[1] + (i - 1) * BPP, &v1, BPP);
            std::memcpy(cur[2] + (i - 2) * BPP, &v2, BPP); std::memcpy(cur[3] + (i - 3) * BPP, &v3, BPP);
        } else {
            store(0, i, (uint32_t)_mm_cvtsi128_si32(pa)); store(1, i - 1, (uint32_t)_mm_cvtsi128_si32(_mm_srli_si128(pa, 4)));
            store(2, i - 2, (uint32_t)_mm_cvtsi128_si32(pb)); store(3, i - 3, (uint32_t)_mm_cvtsi128_si32(_mm_srli_si128(pb, 4)));
        }
    }
}

template <int BPP> void unfilter_two_rows(int ft0, int ft1, unsigned char *cur0, unsigned char *cur1, const unsigned char *prev, size_t npx)
{
    if (ft0 == 3) { if (ft1 == 3) unfilter_two_rows<BPP, 3, 3>(cur0, cur1, prev, npx); else unfilter_two_rows<BPP, 3, 4>(cur0, cur1, prev, npx); }
    else          { if (ft1 == 3) unfilter_two_rows<BPP, 4, 3>(cur0, cur1, prev, npx); else unfilter_two_rows<BPP, 4, 4>(cur0, cur1, prev, npx); }
}
#endif

template <int BPP> int decode_rows_8bit(const char *path, unsigned char *raw, size_t W, size_t H, uint32_t *dst, const unsigned char *zero_row)
{
    const size_t stride = W * BPP;
#if defined(__x86_64__)
    static const bool have_ssse3 = __builtin_cpu_supports("ssse3") && !std::getenv("CSIC_NO_SIMD");   // (the variable: tests of the portable loop)
#endif
    auto emit = [&](const unsigned char *px, size_t y) {
#if defined(__x86_64__)
        if (have_ssse3) { to_argb_ssse3<BPP>(px, dst + y * W, W); return; }
#endif
        to_argb_scalar<BPP>(px, dst + y * W, W);
    };
    for (size_t y = 0; y < H; ++y) {
        unsigned char *cur = raw + y * (stride + 1) + 1;
        const unsigned char *prev = y ? cur - (stride + 1) : zero_row;
        const int ft = cur[-1];
#if defined(__x86_64__)
        if ((ft == 3 || ft == 4) && y + 3 < H && W >= 4 && cur[stride] == ft && cur[2 * stride + 1] == ft && cur[3 * stride + 2] == ft) {
            unsigned char *const rows[4] = {cur, cur + (stride + 1), cur + 2 * (stride + 1), cur + 3 * (stride + 1)};   // (filter bytes of the next three rows)
            if (ft == 3) unfilter_four_rows<BPP, 3>(rows, prev, W); else unfilter_four_rows<BPP, 4>(rows, prev, W);
            for (int r = 0; r < 4; ++r) emit(rows[r], y + (size_t)r);
            y += 3;
            continue;
        }
        if ((ft == 3 || ft == 4) && y + 1 < H && (cur[stride] == 3 || cur[stride] == 4)) {       // cur[stride]: the next row's filter byte
            unfilter_two_rows<BPP>(ft, cur[stride], cur, cur + stride + 1, prev, W);
            emit(cur, y);
            emit(cur + stride + 1, y + 1);
            ++y;
            continue;
        }
#endif
        switch (ft) {
        case 0: break;
        case 1: unfilter_sub<BPP>(cur, W); break;
        case 2: for (size_t i = 0; i < stride; ++i) cur[i] = (unsigned char)(cur[i] + prev[i]); break;
        case 3: unfilter_avg<BPP>(cur, prev, W); break;
        case 4: unfilter_paeth<BPP>(cur, prev, W); break;
        default: return set_error(CSIC_EFORMAT, "%s: bad filter type %d", path, ft);
        }
        emit(cur, y);
    }
    return CSIC_OK;
}

} // namespace

static int png_info_impl(const char *path, int32_t *width, int32_t *height)
{
    if (!path || !width || !height) return set_error(CSIC_EINVAL_NULL, "argument is NULL");
    FileBytes f;
    int st = read_file(path, f);
    if (st != CSIC_OK) return st;
    Header hd;
    st = parse(path, f, hd, nullptr, nullptr);
    if (st != CSIC_OK) return st;
    *width = (int32_t)hd.w; *height = (int32_t)hd.h;
    clear_error();
    return CSIC_OK;
}

static int png_read_impl(const char *path, uint32_t *dst, size_t dst_px)
{
    if (!path || !dst) return set_error(CSIC_EINVAL_NULL, "argument is NULL");
    FileBytes f;
    std::vector<unsigned char> plte, joined;
    Spans idat;
    int st = read_file(path, f);
    if (st != CSIC_OK) return st;
    Header hd;
    st = parse(path, f, hd, &plte, &idat);
    if (st != CSIC_OK) return st;
    const size_t W = hd.w, H = hd.h;
    if (dst_px != W * H) return set_error(CSIC_EINVAL_SIZE, "%s is %zux%zu = %zu pixels, destination holds %zu", path, W, H, W * H, dst_px);
    const int nch = channels(hd.ctype);
    const size_t bits_pp = (size_t)nch * hd.depth;
    const size_t stride = (W * bits_pp + 7) / 8;
    const size_t bpp = bits_pp >= 8 ? bits_pp / 8 : 1;           // filter distance in bytes
    const unsigned char *z = nullptr;
    size_t zlen = 0;
    if (idat.size() == 1) { z = idat[0].first; zlen = idat[0].second; }   // one IDAT chunk: decode it where it lies
    else {
        for (const auto &sp : idat) zlen += sp.second;
        joined.reserve(zlen);
        for (const auto &sp : idat) joined.insert(joined.end(), sp.first, sp.first + sp.second);
        z = joined.data();
    }
    const size_t rawlen = (stride + 1) * H, pad = 16;                     // the padding: see unfilter_paeth
    std::unique_ptr<unsigned char[]> rawbuf(new unsigned char[rawlen + pad]);
    unsigned char *raw = rawbuf.get();
    std::memset(raw + rawlen, 0, pad);
    if (zlib_decode_exact(z, zlen, raw, rawlen) != 0) return set_error(CSIC_EFORMAT, "%s: corrupt image data", path);
    if (hd.ctype == 3 && plte.size() < 3) return set_error(CSIC_EFORMAT, "%s: palette image without PLTE", path);
    const std::vector<unsigned char> zero_row(stride + pad, 0);
    if (hd.depth == 8 && (hd.ctype == 2 || hd.ctype == 6)) {
        st = hd.ctype == 2 ? decode_rows_8bit<3>(path, raw, W, H, dst, zero_row.data()) : decode_rows_8bit<4>(path, raw, W, H, dst, zero_row.data());
        if (st != CSIC_OK) return st;
        clear_error();
        return CSIC_OK;
    }
    for (size_t y = 0; y < H; ++y) {
        unsigned char *row = &raw[y * (stride + 1)];
        const int ft = row[0];
        unsigned char *cur = row + 1;
        const unsigned char *prev = y ? cur - (stride + 1) : zero_row.data();
        switch (ft) {
        case 0: break;
        case 1: for (size_t i = bpp; i < stride; ++i) cur[i] = (unsigned char)(cur[i] + cur[i - bpp]); break;
        case 2: for (size_t i = 0; i < stride; ++i) cur[i] = (unsigned char)(cur[i] + prev[i]); break;
        case 3:
            for (size_t i = 0; i < stride; ++i) cur[i] = (unsigned char)(cur[i] + (((i >= bpp ? cur[i - bpp] : 0) + prev[i]) >> 1));
            break;
        case 4:
            for (size_t i = 0; i < stride; ++i)
                cur[i] = (unsigned char)(cur[i] + paeth(i >= bpp ? cur[i - bpp] : 0, prev[i], i >= bpp ? prev[i - bpp] : 0));
            break;
        default: return set_error(CSIC_EFORMAT, "%s: bad filter type %d", path, ft);
        }
        uint32_t *out = dst + y * W;
        const size_t sb = hd.depth == 16 ? 2 : 1;                // bytes per sample (16-bit: keep the high byte)
        for (size_t x = 0; x < W; ++x) {
            unsigned r, g, b;
            if (hd.depth >= 8) {
                const unsigned char *px = cur + x * nch * sb;
                if (hd.ctype == 2 || hd.ctype == 6) { r = px[0]; g = px[sb]; b = px[2 * sb]; }
                else if (hd.ctype == 3) {
                    const size_t idx = px[0];
                    if (3 * idx + 2 >= plte.size()) return set_error(CSIC_EFORMAT, "%s: palette index out of range", path);
                    r = plte[3 * idx]; g = plte[3 * idx + 1]; b = plte[3 * idx + 2];
                } else { r = g = b = px[0]; }
            } else {                                             // 1/2/4-bit grey or palette
                const unsigned v = (cur[(x * hd.depth) >> 3] >> (8 - hd.depth - ((x * hd.depth) & 7))) & ((1u << hd.depth) - 1);
                if (hd.ctype == 3) {
                    if (3 * (size_t)v + 2 >= plte.size()) return set_error(CSIC_EFORMAT, "%s: palette index out of range", path);
                    r = plte[3 * v]; g = plte[3 * v + 1]; b = plte[3 * v + 2];
                } else { r = g = b = v * 255u / ((1u << hd.depth) - 1); }
            }
            out[x] = 0xFF000000u | (r << 16) | (g << 8) | b;
        }
    }
    clear_error();
    return CSIC_OK;
}

// ---- writer ---------------------------------------------------------------------------------------------------------
// One image is filtered and deflated on several threads: rows are filtered in blocks (an encoder's filters work on ORIGINAL
// pixels: nothing is serial), and the filtered stream is deflated in pieces of kDeflatePiece bytes, each an independent raw
// deflate run primed with the 32 KiB in front of it (deflateSetDictionary) and closed with a sync flush, the last one with
// Z_FINISH -- concatenated they are one valid deflate stream (what pigz does).  The pieces depend on the data alone, never on
// the thread count, so the file is the same bytes on 1 thread and on 16; an image of up to kDeflatePiece filtered bytes is one
// piece = what compress2() produces.  zlib's deflate is 90 % of an encode: a 4096x4096 result (50 MB filtered) takes seconds
// on one thread.
namespace {

constexpr size_t kDeflatePiece = (size_t)1 << 20;
constexpr size_t kFilterRows = 32;

// fn(task) for task in [0, ntasks) on up to `threads` threads (the caller's included); false if a task ran out of memory
template <class F> bool parallel_for(size_t ntasks, int threads, F fn)
{
    std::atomic<size_t> next{0};
    std::atomic<bool> ok{true};
    auto work = [&] {
        try { for (size_t t; (t = next.fetch_add(1)) < ntasks;) fn(t); }
        catch (...) { ok = false; next = ntasks; }
    };
    std::vector<std::thread> pool;
    const size_t extra = threads > 1 ? (size_t)threads - 1 : 0;
    try { for (size_t k = 0; k < extra && k + 1 < ntasks; ++k) pool.emplace_back(work); } catch (...) { /* fewer threads, same result */ }
    work();
    for (auto &t : pool) t.join();
    return ok.load();
}

// Rows [y0, y1) of `src` as filtered PNG rows (filter byte + 3 bytes per pixel) at raw + y * (stride + 1).
void filter_rows(const uint32_t *src, size_t W, size_t y0, size_t y1, unsigned char *raw)
{
    const size_t stride = W * 3, lead = 16;
    // Rows carry `lead` zero bytes in front (the pixels "left of the image"), so every candidate is one loop without edge
    // cases that the compiler vectorises.
    std::vector<unsigned char> row_a(lead + stride, 0), row_b(lead + stride, 0), cand(5 * stride);
    unsigned char *cur = row_a.data() + lead, *prev = row_b.data() + lead;
    auto unpack = [&](size_t y, unsigned char *dst) {
        for (size_t x = 0; x < W; ++x) {
            const uint32_t v = src[y * W + x];
            dst[3 * x] = (unsigned char)(v >> 16); dst[3 * x + 1] = (unsigned char)(v >> 8); dst[3 * x + 2] = (unsigned char)v;
        }
    };
    if (y0 > 0) unpack(y0 - 1, prev);
    for (size_t y = y0; y < y1; ++y) {
        unpack(y, cur);
        unsigned char *c0 = cand.data(), *c1 = c0 + stride, *c2 = c1 + stride, *c3 = c2 + stride, *c4 = c3 + stride;
        const unsigned char *left = cur - 3, *up_left = prev - 3;
        for (size_t i = 0; i < stride; ++i) c0[i] = cur[i];
        for (size_t i = 0; i < stride; ++i) c1[i] = (unsigned char)(cur[i] - left[i]);
        for (size_t i = 0; i < stride; ++i) c2[i] = (unsigned char)(cur[i] - prev[i]);
        for (size_t i = 0; i < stride; ++i) c3[i] = (unsigned char)(cur[i] - ((left[i] + prev[i]) >> 1));
        for (size_t i = 0; i < stride; ++i) {
            const int a = left[i], b = prev[i], c = up_left[i];
            const int pa = b > c ? b - c : c - b, pb = a > c ? a - c : c - a, pc = a + b > 2 * c ? a + b - 2 * c : 2 * c - a - b;
            const int pred = (pa <= pb && pa <= pc) ? a : (pb <= pc ? b : c);
            c4[i] = (unsigned char)(cur[i] - pred);
        }
        // adaptive filtering: minimum sum of absolute (signed) residuals, the first of equal sums
        int best_ft = 0; unsigned long best = ~0ul;
        for (int ft = 0; ft < 5; ++ft) {
            const unsigned char *c = cand.data() + (size_t)ft * stride;
            unsigned long sum = 0;
            for (size_t i = 0; i < stride; ++i) sum += c[i] < 128 ? c[i] : 256 - c[i];
            if (sum < best) { best = sum; best_ft = ft; }
        }
        unsigned char *row = raw + y * (stride + 1);
        row[0] = (unsigned char)best_ft;
        std::memcpy(row + 1, cand.data() + (size_t)best_ft * stride, stride);
        std::swap(cur, prev);
    }
}

// Piece k of the filtered stream as raw deflate data; every piece but the last ends on a byte boundary (sync flush).
bool deflate_piece(const unsigned char *raw, size_t rawlen, size_t k, int level, std::vector<unsigned char> &out)
{
    const size_t off = k * kDeflatePiece, len = rawlen - off < kDeflatePiece ? rawlen - off : kDeflatePiece;
    const bool last = off + len == rawlen;
    z_stream zs;
    std::memset(&zs, 0, sizeof zs);
    if (deflateInit2(&zs, level, Z_DEFLATED, -15, 8, Z_DEFAULT_STRATEGY) != Z_OK) return false;
    bool ok = true;
    if (off > 0) {
        const size_t dlen = off < 32768 ? off : 32768;
        ok = deflateSetDictionary(&zs, raw + off - dlen, (uInt)dlen) == Z_OK;
    }
    if (ok) {
        out.resize((size_t)deflateBound(&zs, (uLong)len) + 64);
        zs.next_in = const_cast<unsigned char *>(raw + off); zs.avail_in = (uInt)len;
        zs.next_out = out.data(); zs.avail_out = (uInt)out.size();
        const int rc = deflate(&zs, last ? Z_FINISH : Z_SYNC_FLUSH);
        ok = last ? rc == Z_STREAM_END : (rc == Z_OK && zs.avail_in == 0 && zs.avail_out > 0);
        if (ok) out.resize((size_t)zs.total_out);
    }
    deflateEnd(&zs);
    return ok;
}

int writer_threads(int asked)
{
    if (asked > 0) return asked > 64 ? 64 : asked;
    if (const char *e = std::getenv("CSIC_PNG_THREADS")) { const int v = std::atoi(e); if (v > 0) return v > 64 ? 64 : v; }
    const int budget = host_cpu_budget();                  // affinity mask and cgroup quota, not the CPUs the machine shows
    return budget > 16 ? 16 : budget;
}

} // namespace

static int png_write_impl(const char *path, const uint32_t *src, int32_t width, int32_t height, int32_t level, int threads)
{
    if (!path || !src) return set_error(CSIC_EINVAL_NULL, "argument is NULL");
    if (width <= 0 || height <= 0) return set_error(CSIC_EINVAL_DIMS, "width and height must be positive. Got %dx%d", width, height);
    if (level < 0 || level > 9) level = 6;
    const size_t W = (size_t)width, H = (size_t)height, stride = W * 3, rawlen = (stride + 1) * H;
    const size_t npieces = (rawlen + kDeflatePiece - 1) / kDeflatePiece, nblocks = (H + kFilterRows - 1) / kFilterRows;
    threads = npieces > 1 ? writer_threads(threads) : 1;
    std::unique_ptr<unsigned char[]> raw(new unsigned char[rawlen]);
    if (!parallel_for(nblocks, threads, [&](size_t b) {
            const size_t y0 = b * kFilterRows;
            filter_rows(src, W, y0, y0 + kFilterRows < H ? y0 + kFilterRows : H, raw.get());
        }))
        return set_error(CSIC_ENOMEM, "out of host memory");
    std::vector<std::vector<unsigned char>> pieces(npieces);
    std::atomic<bool> deflated{true};
    if (!parallel_for(npieces, threads, [&](size_t k) { if (!deflate_piece(raw.get(), rawlen, k, level, pieces[k])) deflated = false; }))
        return set_error(CSIC_ENOMEM, "out of host memory");
    if (!deflated.load()) return set_error(CSIC_EIO, "deflate failed");
    // the zlib wrapper around the pieces: header as deflateInit(level) writes it, Adler-32 of the filtered stream
    const unsigned flevel = level < 2 ? 0 : level < 6 ? 1 : level == 6 ? 2 : 3;
    unsigned header = (0x78u << 8) | (flevel << 6);
    header += 31 - header % 31;
    const uint32_t adler = adler32_update(1, raw.get(), rawlen);
    pieces.front().insert(pieces.front().begin(), {(unsigned char)(header >> 8), (unsigned char)header});
    for (int sh = 24; sh >= 0; sh -= 8) pieces.back().push_back((unsigned char)(adler >> sh));
    FILE *fp = std::fopen(path, "wb");
    if (!fp) return set_error(CSIC_EIO, "cannot create %s", path);
    auto chunk = [&](const char *type, const unsigned char *data, size_t len) {
        unsigned char hdr[8] = {(unsigned char)(len >> 24), (unsigned char)(len >> 16), (unsigned char)(len >> 8), (unsigned char)len,
                                (unsigned char)type[0], (unsigned char)type[1], (unsigned char)type[2], (unsigned char)type[3]};
        uint32_t crc = crc32_update(0, hdr + 4, 4);
        if (len) crc = crc32_update(crc, data, len);
        unsigned char tail[4] = {(unsigned char)(crc >> 24), (unsigned char)(crc >> 16), (unsigned char)(crc >> 8), (unsigned char)crc};
        bool ok = std::fwrite(hdr, 1, 8, fp) == 8;
        if (len) ok = ok && std::fwrite(data, 1, len, fp) == len;
        return ok && std::fwrite(tail, 1, 4, fp) == 4;
    };
    unsigned char ihdr[13] = {(unsigned char)(width >> 24), (unsigned char)(width >> 16), (unsigned char)(width >> 8), (unsigned char)width,
                              (unsigned char)(height >> 24), (unsigned char)(height >> 16), (unsigned char)(height >> 8), (unsigned char)height,
                              8, 2, 0, 0, 0};
    bool ok = std::fwrite(kSig, 1, 8, fp) == 8 && chunk("IHDR", ihdr, 13);
    for (const auto &piece : pieces) ok = ok && chunk("IDAT", piece.data(), piece.size());     // one IDAT per piece (each far below 2^31 bytes)
    ok = ok && chunk("IEND", nullptr, 0);
    ok = (std::fclose(fp) == 0) && ok;
    if (!ok) return set_error(CSIC_EIO, "write to %s failed", path);
    clear_error();
    return CSIC_OK;
}

// No C++ exception may cross the C ABI: allocation failures become CSIC_ENOMEM.
#define CSIC_NOEXCEPT_CALL(expr)                                                        \
    try { return (expr); }                                                              \
    catch (const std::bad_alloc &) { return set_error(CSIC_ENOMEM, "out of host memory"); } \
    catch (...) { return set_error(CSIC_EIO, "unexpected failure in the PNG codec"); }

extern "C" {

int csic_png_info(const char *path, int32_t *width, int32_t *height) { CSIC_NOEXCEPT_CALL(png_info_impl(path, width, height)) }

int csic_png_read_argb(const char *path, uint32_t *dst, size_t dst_px) { CSIC_NOEXCEPT_CALL(png_read_impl(path, dst, dst_px)) }

int csic_png_write_argb(const char *path, const uint32_t *src, int32_t width, int32_t height, int32_t level)
{
    CSIC_NOEXCEPT_CALL(png_write_impl(path, src, width, height, level, 0))
}

} // extern "C"

namespace csic {
int png_write_argb_threads(const char *path, const uint32_t *src, int32_t width, int32_t height, int32_t level, int threads)
{
    CSIC_NOEXCEPT_CALL(png_write_impl(path, src, width, height, level, threads))
}
} // namespace csic
