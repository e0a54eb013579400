// csic_png.cpp -- minimal PNG codec for the host I/O either side of the hot path (host only, zlib).
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
#include <zlib.h>

#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <new>
#include <vector>

#include "csic_internal.h"

using namespace csic;

namespace {

struct Header {
    uint32_t w = 0, h = 0;
    int depth = 0, ctype = 0, interlace = 0;
};

uint32_t be32(const unsigned char *p) { return ((uint32_t)p[0] << 24) | ((uint32_t)p[1] << 16) | ((uint32_t)p[2] << 8) | p[3]; }

const unsigned char kSig[8] = {0x89, 'P', 'N', 'G', 0x0D, 0x0A, 0x1A, 0x0A};

int read_file(const char *path, std::vector<unsigned char> &buf)
{
    FILE *fp = std::fopen(path, "rb");
    if (!fp) return set_error(CSIC_EIO, "cannot open %s", path);
    std::fseek(fp, 0, SEEK_END);
    long n = std::ftell(fp);
    std::fseek(fp, 0, SEEK_SET);
    if (n < 0) { std::fclose(fp); return set_error(CSIC_EIO, "cannot size %s", path); }
    buf.resize((size_t)n);
    size_t got = n ? std::fread(buf.data(), 1, (size_t)n, fp) : 0;
    std::fclose(fp);
    if (got != (size_t)n) return set_error(CSIC_EIO, "short read on %s", path);
    return CSIC_OK;
}

// Walks the chunk list: fills the header, the palette and the concatenated IDAT stream.
int parse(const char *path, const std::vector<unsigned char> &f, Header &hd, std::vector<unsigned char> *plte,
          std::vector<unsigned char> *idat)
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
        if ((uint32_t)crc32(crc32(0L, Z_NULL, 0), type, 4 + len) != crc) return set_error(CSIC_EFORMAT, "%s: chunk CRC mismatch", path);
        if (!std::memcmp(type, "IHDR", 4)) {
            if (len != 13) return set_error(CSIC_EFORMAT, "%s: bad IHDR", path);
            hd.w = be32(data); hd.h = be32(data + 4);
            hd.depth = data[8]; hd.ctype = data[9]; hd.interlace = data[12];
            if (data[10] != 0 || data[11] != 0) return set_error(CSIC_EFORMAT, "%s: unknown compression/filter method", path);
            have_ihdr = true;
        } else if (!std::memcmp(type, "PLTE", 4)) {
            if (plte) plte->assign(data, data + len);
        } else if (!std::memcmp(type, "IDAT", 4)) {
            if (idat) idat->insert(idat->end(), data, data + len);
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

} // namespace

static int png_info_impl(const char *path, int32_t *width, int32_t *height)
{
    if (!path || !width || !height) return set_error(CSIC_EINVAL_NULL, "argument is NULL");
    std::vector<unsigned char> f;
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
    std::vector<unsigned char> f, plte, idat;
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
    std::vector<unsigned char> raw((stride + 1) * H);
    uLongf rawlen = (uLongf)raw.size();
    if (uncompress(raw.data(), &rawlen, idat.data(), (uLong)idat.size()) != Z_OK || rawlen != raw.size())
        return set_error(CSIC_EFORMAT, "%s: corrupt image data", path);
    if (hd.ctype == 3 && plte.size() < 3) return set_error(CSIC_EFORMAT, "%s: palette image without PLTE", path);
    std::vector<unsigned char> prev(stride, 0);
    for (size_t y = 0; y < H; ++y) {
        unsigned char *row = &raw[y * (stride + 1)];
        const int ft = row[0];
        unsigned char *cur = row + 1;
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
        std::memcpy(prev.data(), cur, stride);
    }
    clear_error();
    return CSIC_OK;
}

static int png_write_impl(const char *path, const uint32_t *src, int32_t width, int32_t height, int32_t level)
{
    if (!path || !src) return set_error(CSIC_EINVAL_NULL, "argument is NULL");
    if (width <= 0 || height <= 0) return set_error(CSIC_EINVAL_DIMS, "width and height must be positive. Got %dx%d", width, height);
    if (level < 0 || level > 9) level = 6;
    const size_t W = (size_t)width, H = (size_t)height, stride = W * 3;
    std::vector<unsigned char> raw((stride + 1) * H), cur(stride), prev(stride, 0), cand(stride);
    for (size_t y = 0; y < H; ++y) {
        for (size_t x = 0; x < W; ++x) {
            const uint32_t v = src[y * W + x];
            cur[3 * x] = (unsigned char)(v >> 16); cur[3 * x + 1] = (unsigned char)(v >> 8); cur[3 * x + 2] = (unsigned char)v;
        }
        // adaptive filtering: minimum sum of absolute (signed) residuals
        int best_ft = 0; unsigned long best = ~0ul;
        unsigned char *row = &raw[y * (stride + 1)];
        for (int ft = 0; ft < 5; ++ft) {
            unsigned long sum = 0;
            for (size_t i = 0; i < stride; ++i) {
                const int a = i >= 3 ? cur[i - 3] : 0, b = prev[i], c = i >= 3 ? prev[i - 3] : 0;
                const int pred = ft == 0 ? 0 : ft == 1 ? a : ft == 2 ? b : ft == 3 ? ((a + b) >> 1) : paeth(a, b, c);
                const unsigned char d = (unsigned char)(cur[i] - pred);
                cand[i] = d;
                sum += d < 128 ? d : 256 - d;
            }
            if (sum < best) { best = sum; best_ft = ft; row[0] = (unsigned char)ft; std::memcpy(row + 1, cand.data(), stride); }
        }
        (void)best_ft;
        prev = cur;
    }
    uLongf clen = compressBound((uLong)raw.size());
    std::vector<unsigned char> comp(clen);
    if (compress2(comp.data(), &clen, raw.data(), (uLong)raw.size(), level) != Z_OK) return set_error(CSIC_EIO, "deflate failed");
    FILE *fp = std::fopen(path, "wb");
    if (!fp) return set_error(CSIC_EIO, "cannot create %s", path);
    auto chunk = [&](const char *type, const unsigned char *data, uint32_t len) {
        unsigned char hdr[8] = {(unsigned char)(len >> 24), (unsigned char)(len >> 16), (unsigned char)(len >> 8), (unsigned char)len,
                                (unsigned char)type[0], (unsigned char)type[1], (unsigned char)type[2], (unsigned char)type[3]};
        uLong crc = crc32(crc32(0L, Z_NULL, 0), hdr + 4, 4);
        if (len) crc = crc32(crc, data, len);
        unsigned char tail[4] = {(unsigned char)(crc >> 24), (unsigned char)(crc >> 16), (unsigned char)(crc >> 8), (unsigned char)crc};
        bool ok = std::fwrite(hdr, 1, 8, fp) == 8;
        if (len) ok = ok && std::fwrite(data, 1, len, fp) == len;
        return ok && std::fwrite(tail, 1, 4, fp) == 4;
    };
    unsigned char ihdr[13] = {(unsigned char)(width >> 24), (unsigned char)(width >> 16), (unsigned char)(width >> 8), (unsigned char)width,
                              (unsigned char)(height >> 24), (unsigned char)(height >> 16), (unsigned char)(height >> 8), (unsigned char)height,
                              8, 2, 0, 0, 0};
    bool ok = std::fwrite(kSig, 1, 8, fp) == 8 && chunk("IHDR", ihdr, 13) && chunk("IDAT", comp.data(), (uint32_t)clen) && chunk("IEND", nullptr, 0);
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
    CSIC_NOEXCEPT_CALL(png_write_impl(path, src, width, height, level))
}

} // extern "C"
