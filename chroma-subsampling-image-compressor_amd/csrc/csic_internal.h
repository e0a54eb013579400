// csic_internal.h -- shared between the host-only translation unit and the HIP one.
#pragma once
#include "csic.h"

#include <cstdarg>
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

// Exact unsigned division by a run-time constant without a divide (k_generic's stream-index arithmetic): for
// 1 <= d < 2^31 and every n < 2^31,  n / d == (uint64(n) * m) >> k  with  k = 31 + ceil(log2 d),  m = ceil(2^k / d) < 2^32.
// (Error term e = m*d - 2^k < d <= 2^ceil(log2 d), and n * e < 2^31 * 2^ceil(log2 d) = 2^k.)  Host side, csic_host.cpp;
// checked against the hardware divide over edge cases and random pairs in tests/cpp/host_sanitize.cpp.
void magic_div(uint32_t d, uint32_t *m, uint32_t *k);

} // namespace csic
