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

} // namespace csic
