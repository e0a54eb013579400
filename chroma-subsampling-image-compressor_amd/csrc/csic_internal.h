// csic_internal.h -- shared between the host-only translation unit and the HIP one.
#pragma once
#include "csic.h"

#include <cstdarg>
#include <cstddef>
#include <cstdint>

namespace csic {

// Derived, validated geometry of one parameter set.
struct Geometry {
    int32_t W, H;        // input
    int32_t Wo, Ho;      // output, ceil(W/f) x ceil(H/f)
    int32_t f;           // spatial factor
    int32_t h, v;        // chroma horizontal / vertical hold factors: h = 4/a, v = (b == 0) ? 2 : 1
    int32_t s_first;     // 1 = spatial stage sits before the chroma stage (order class S-before-C)
    int32_t last_sample_col; // ((W-1)/h)*h : column of the last chroma sample of a (chroma) row
    uint32_t mask_y, mask_cb, mask_cr; // quantiser AND masks, 0xFF << (8 - bits)
};

int  set_error(int status, const char *fmt, ...) __attribute__((format(printf, 2, 3)));
void clear_error();
int  derive_geometry(const csic_params *p, Geometry *g);   // validates first
void planar_layout(const Geometry &g, const csic_params *p, csic_planar_layout *layout);   // csic.h: CSIC_FMT_PLANAR

// Exact unsigned division by a run-time constant without a divide (k_generic's stream-index arithmetic): for
// 1 <= d < 2^31 and every n < 2^31,  n / d == (uint64(n) * m) >> k  with  k = 31 + ceil(log2 d),  m = ceil(2^k / d) < 2^32.
// (Error term e = m*d - 2^k < d <= 2^ceil(log2 d), and n * e < 2^31 * 2^ceil(log2 d) = 2^k.)  Host side, csic_host.cpp;
// checked against the hardware divide over edge cases and random pairs in tests/cpp/host_sanitize.cpp.
void magic_div(uint32_t d, uint32_t *m, uint32_t *k);

// csic_inflate.cpp: what the PNG reader needs of zlib, faster.  zlib_decode_exact: 0 if in[0, in_len) is a well-formed zlib
// stream of exactly out_len bytes with the right check value (1 corrupt, 2 truncated, 3 longer, 4 shorter than out_len).
int      zlib_decode_exact(const unsigned char *in, size_t in_len, unsigned char *out, size_t out_len);
uint32_t crc32_update(uint32_t crc, const unsigned char *p, size_t n);      // crc32_update(0, ..) starts a CRC, as zlib's crc32
uint32_t adler32_update(uint32_t adler, const unsigned char *p, size_t n);  // adler32_update(1, ..) starts a sum, as zlib's adler32

// csic_png.cpp: csic_png_write_argb on a given number of threads (<= 0: CSIC_PNG_THREADS or up to 16); the bytes written do
// not depend on it.  The file pools of csic_files.hip, parallel over files already, ask for 1.
int png_write_argb_threads(const char *path, const uint32_t *src, int32_t width, int32_t height, int32_t level, int threads);

} // namespace csic
