// csic_kernel_ops.h -- device-side building blocks shared by the kernel translation units (csic_kernels.hip, csic_planar.hip):
// the wave prologue, address-space-qualified pixel pointers, streaming (non-temporal) accesses, the per-pixel arithmetic of
// the forward / inverse transforms and the quantiser, and -- in the CSIC_DEBUG build -- the range checks of every global access.
// Citations are relative to /root/reference/.
#pragma once
#include "csic_hip_common.h"

namespace csic {

// Wave prologue.  hipcc sinks every kernel-argument s_load to its first use, and blockDim/gridDim come
// from the hidden-argument area, so a kernel with early exits pays 3-4 DEPENDENT scalar-load round
// trips (~0.4 us) before its first global load -- 5 % of the f=1 kernel, whose waves hold a single
// 16-byte load in flight.  Pinning the arguments in SGPRs at entry turns that into one batch and one
// s_waitcnt; the launch geometry travels in KArgs for the same reason.
__device__ __forceinline__ void pin_args(const KArgs &a)
{
    asm volatile("" ::"s"(a.in), "s"(a.out), "s"(a.W), "s"(a.H), "s"(a.Wo), "s"(a.Ho), "s"(a.last_sample_col));
    asm volatile("" ::"s"(a.my), "s"(a.mcb), "s"(a.mcr), "s"(a.in_frame_px), "s"(a.out_frame_px), "s"(a.bdx),
                 "s"(a.bdy), "s"(a.row_step), "s"(a.ip), "s"(a.op), "s"(a.in_tab), "s"(a.out_tab));
}

// Pixel pointers carry their address space.  A pointer that was itself loaded from memory (frame-table mode) is
// a generic pointer to the compiler, and every access through it becomes a flat_load / flat_store with a 64-bit
// per-lane address and an lgkmcnt dependency; the frames are always global memory, so say so.
#define CSIC_GLOBAL __attribute__((address_space(1)))
#define CSIC_CONSTANT __attribute__((address_space(4)))
typedef const uint32_t CSIC_GLOBAL *gin_t;
typedef uint32_t CSIC_GLOBAL *gout_t;

// Base of the frame this block works on (grid z = frame): consecutive frames behind a.in / a.out, or -- frame-table mode --
// whatever the device-resident tables name.  The tables are read through the constant address space (they are never
// written while a kernel runs): one wave-uniform s_load, the frame base stays in SGPRs.
__device__ __forceinline__ gin_t frame_in(const KArgs &a)
{
    if (a.in_tab) return (gin_t)((const uint64_t CSIC_CONSTANT *)(uintptr_t)a.in_tab)[blockIdx.z];
    return (gin_t)(uintptr_t)a.in + (int64_t)blockIdx.z * a.in_frame_px;
}
__device__ __forceinline__ gout_t frame_out(const KArgs &a)
{
    if (a.out_tab) return (gout_t)((const uint64_t CSIC_CONSTANT *)(uintptr_t)a.out_tab)[blockIdx.z];
    return (gout_t)(uintptr_t)a.out + (int64_t)blockIdx.z * a.out_frame_px;
}

enum { R_FLOOR = CSIC_ROUND_FLOOR_HW, R_TRUNC = CSIC_ROUND_TRUNC_SW };
enum { F_ARGB = CSIC_FMT_ARGB8888, F_YCC = CSIC_FMT_YCBCR888X };

// ------------------------------------------------------------------------------------------------
// streaming memory access
// ------------------------------------------------------------------------------------------------
// Frames are read once and written once and are far larger than L2 (4 MiB/XCD): non-temporal
// ("nt") loads and stores keep the stream from displacing itself in the cache hierarchy.  Measured on
// MI355X at 8192x8192, f=2 (tools/ubench.hip): 35.9 -> 33.5 us per frame, the same gain a plain
// 16 B/lane copy kernel sees (6.03 -> 6.41 TB/s).
typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
typedef uint32_t u32x2 __attribute__((ext_vector_type(2)));

template <bool NT> __device__ __forceinline__ uint32_t ld1(gin_t p)
{ return NT ? __builtin_nontemporal_load(p) : *p; }
template <bool NT> __device__ __forceinline__ u32x4 ld4(gin_t p)
{ typedef const u32x4 CSIC_GLOBAL *vp; return NT ? __builtin_nontemporal_load((vp)p) : *(vp)p; }
template <bool NT> __device__ __forceinline__ void st1(gout_t p, uint32_t v)
{ if (NT) __builtin_nontemporal_store(v, p); else *p = v; }
template <bool NT> __device__ __forceinline__ void st2(gout_t p, u32x2 v)
{ typedef u32x2 CSIC_GLOBAL *vp; if (NT) __builtin_nontemporal_store(v, (vp)p); else *(vp)p = v; }
template <bool NT> __device__ __forceinline__ void st4(gout_t p, u32x4 v)
{ typedef u32x4 CSIC_GLOBAL *vp; if (NT) __builtin_nontemporal_store(v, (vp)p); else *(vp)p = v; }
// the same through plain pointers (kernel arguments: the compiler infers global itself; tools/ubench*.hip)
template <bool NT> __device__ __forceinline__ uint32_t ld1(const uint32_t *p) { return ld1<NT>((gin_t)(uintptr_t)p); }
template <bool NT> __device__ __forceinline__ u32x4 ld4(const uint32_t *p) { return ld4<NT>((gin_t)(uintptr_t)p); }
template <bool NT> __device__ __forceinline__ void st1(uint32_t *p, uint32_t v) { st1<NT>((gout_t)(uintptr_t)p, v); }
template <bool NT> __device__ __forceinline__ void st2(uint32_t *p, u32x2 v) { st2<NT>((gout_t)(uintptr_t)p, v); }
template <bool NT> __device__ __forceinline__ void st4(uint32_t *p, u32x4 v) { st4<NT>((gout_t)(uintptr_t)p, v); }

// ------------------------------------------------------------------------------------------------
// frame accessors -- and, in the CSIC_DEBUG build (`make debug` -> libcsic_hip_debug.so), the bounds of every access
// ------------------------------------------------------------------------------------------------
// Every pixel kernel addresses its frame as  base + offset-in-pixels.  The accessors below are what the kernels call; in
// the product build they are the bare non-temporal access.  With -DCSIC_DEBUG each one first checks that
// [off, off + n) lies inside the frame's extent -- (H - 1) * pitch + W pixels of input, (Ho - 1) * pitch + Wo of output,
// the last row being only as long as the image -- and executes s_trap otherwise: the queue reports a hardware exception and
// the launch fails instead of reading or writing a neighbour's memory silently.  This is the GPU-side stand-in for a
// sanitizer (GPU AddressSanitizer / xnack+ are not available on the pool); tests: `CSIC_LIB=.../libcsic_hip_debug.so pytest -m gpu`.
#if defined(CSIC_DEBUG) && CSIC_DEBUG
#define CSIC_CHECK(cond) do { if (!(cond)) __builtin_trap(); } while (0)
#else
#define CSIC_CHECK(cond) do { } while (0)
#endif
__device__ __forceinline__ int64_t in_extent(const KArgs &a) { return (int64_t)(a.H - 1) * a.ip + a.W; }
__device__ __forceinline__ int64_t out_extent(const KArgs &a) { return (int64_t)(a.Ho - 1) * a.op + a.Wo; }

template <bool NT> __device__ __forceinline__ uint32_t in1(const KArgs &a, gin_t in, int64_t off)
{ CSIC_CHECK(off >= 0 && off + 1 <= in_extent(a)); (void)a; return ld1<NT>(in + off); }
template <bool NT> __device__ __forceinline__ u32x4 in4(const KArgs &a, gin_t in, int64_t off)
{ CSIC_CHECK(off >= 0 && off + 4 <= in_extent(a)); (void)a; return ld4<NT>(in + off); }
template <bool NT> __device__ __forceinline__ void out1(const KArgs &a, gout_t out, int64_t off, uint32_t v)
{ CSIC_CHECK(off >= 0 && off + 1 <= out_extent(a)); (void)a; st1<NT>(out + off, v); }
template <bool NT> __device__ __forceinline__ void out2(const KArgs &a, gout_t out, int64_t off, u32x2 v)
{ CSIC_CHECK(off >= 0 && off + 2 <= out_extent(a)); (void)a; st2<NT>(out + off, v); }
template <bool NT> __device__ __forceinline__ void out4(const KArgs &a, gout_t out, int64_t off, u32x4 v)
{ CSIC_CHECK(off >= 0 && off + 4 <= out_extent(a)); (void)a; st4<NT>(out + off, v); }

// The same with a 32-bit pixel offset, for kernels that are only launched on frames whose extents fit 2^30 pixels (the flat
// kernels): the byte offset is a uint32, and "uniform base + zero-extended 32-bit lane offset" is the saddr form of
// global_load / global_store -- the frame base stays in SGPRs and no lane computes a 64-bit address.
template <bool NT> __device__ __forceinline__ uint32_t in1n(const KArgs &a, gin_t in, uint32_t off)
{
    CSIC_CHECK((int64_t)off + 1 <= in_extent(a)); (void)a;
    return ld1<NT>((gin_t)((const char CSIC_GLOBAL *)in + (uint64_t)(off << 2)));
}
template <bool NT> __device__ __forceinline__ u32x4 in4n(const KArgs &a, gin_t in, uint32_t off)
{
    CSIC_CHECK((int64_t)off + 4 <= in_extent(a)); (void)a;
    return ld4<NT>((gin_t)((const char CSIC_GLOBAL *)in + (uint64_t)(off << 2)));
}
template <bool NT> __device__ __forceinline__ void out4n(const KArgs &a, gout_t out, uint32_t off, u32x4 v)
{
    CSIC_CHECK((int64_t)off + 4 <= out_extent(a)); (void)a;
    st4<NT>((gout_t)((char CSIC_GLOBAL *)out + (uint64_t)(off << 2)), v);
}
template <bool NT> __device__ __forceinline__ void out1n(const KArgs &a, gout_t out, uint32_t off, uint32_t v)
{
    CSIC_CHECK((int64_t)off + 1 <= out_extent(a)); (void)a;
    st1<NT>((gout_t)((char CSIC_GLOBAL *)out + (uint64_t)(off << 2)), v);
}

// A kernel that picks, block-uniformly, between a straight-line body and a bounds-checked copy of it ends both with the same
// store; LLVM's SimplifyCFG then sinks that store into a common tail -- and the hardware waits there with s_waitcnt vmcnt(0) for
// every earlier load AND store of the wave before it may issue the last one.  On the headline kernel that was 1 us per launch
// (8192x8192 f = 2: 31.8 -> 32.8 us, 79.0 -> 76.8 % of the roofline; found in round 4 by diffing the ISA against round 3's after
// a refactoring that did not touch the kernel's arithmetic -- the number of such sinks in the object went from 76 to 248).  An
// empty volatile asm as the LAST statement of the straight-line body makes the two tails differ: nothing is emitted, nothing
// sinks.  (tools: `make -C csrc asm`, then grep sink.split in build/asm/*.s)
#ifdef CSIC_NO_TAIL_BARRIER      // (A/B builds only: tools/, never shipped)
__device__ __forceinline__ void keep_tail_apart() {}
#else
__device__ __forceinline__ void keep_tail_apart() { asm volatile("; end of the straight-line body" ::: ); }
#endif

// ------------------------------------------------------------------------------------------------
// per-pixel arithmetic
// ------------------------------------------------------------------------------------------------
// Pixel bytes (little endian uint32): b0 = B, b1 = G, b2 = R, b3 = A.

// Y = (77R + 150G + 29B + 128) >> 8.  The sum is non-negative, so floor == trunc, and
// 77 + 150 + 29 = 256 bounds it by 255: no clamp can fire.  One v_dot4_u32_u8 + one shift.
__device__ __forceinline__ uint32_t fwd_y(uint32_t px)
{
    return __builtin_amdgcn_udot4(px, 0x004D961Du /* A:0 R:77 G:150 B:29 */, 128u, false) >> 8;
}

// Cb/Cr with the UNSIGNED dot product.  The negatively weighted bytes are complemented first
// (-43 R = 43 (255 - R) - 43*255), which turns every coefficient into a u8 and moves the sign into a constant:
//   cbI = -43R - 85G + 128B = udot4(px ^ 0x00FFFF00, {R:43, G:85, B:128}) - 32640
//   crI = 128R - 107G - 21B = udot4(px ^ 0x0000FFFF, {R:128, G:107, B:21}) - 32640
// FLOOR: ((cbI + 128) >> 8) + 128 == (cbI + 32896) >> 8 == (udot + 256) >> 8, with the +256 riding in the
// dot's accumulator: xor, v_dot4_u32_u8, shift, min -- 4 VALU ops per channel (the signed v_dot4c_i32_i8
// form needs an accumulator-init move and a subtract on top).  The only clamp that can fire is 256 -> 255
// (SURVEY.md App. A.1).  TRUNC (Scala's '/' rounds toward zero): negative numerators cbI + 128 < 0, i.e.
// udot < 32512, round up instead: (udot + 511) >> 8.
template <int ROUND>
__device__ __forceinline__ void fwd_c(uint32_t px, uint32_t &cb, uint32_t &cr)
{
    const uint32_t ub = __builtin_amdgcn_udot4(px ^ 0x00FFFF00u, 0x002B5580u /* A:0 R:43  G:85  B:128 */, 256u, false);
    const uint32_t ur = __builtin_amdgcn_udot4(px ^ 0x0000FFFFu, 0x00806B15u /* A:0 R:128 G:107 B:21  */, 256u, false);
    if (ROUND == R_FLOOR) {
        cb = min(ub >> 8, 255u);
        cr = min(ur >> 8, 255u);
    } else {
        // ub, ur carry the +256 already: the "negative numerator" test is udot + 256 < 32768
        cb = min((ub + (ub < 32768u ? 255u : 0u)) >> 8, 255u);
        cr = min((ur + (ur < 32768u ? 255u : 0u)) >> 8, 255u);
    }
}

// (Y, Cb, Cr) of an input pixel: the forward transform, or plain unpacking for a YCbCr input stream
template <int ROUND, int INFMT>
__device__ __forceinline__ uint32_t in_y(uint32_t px) { return INFMT == F_YCC ? (px & 0xFFu) : fwd_y(px); }
template <int ROUND, int INFMT>
__device__ __forceinline__ void in_c(uint32_t px, uint32_t &cb, uint32_t &cr)
{
    if (INFMT == F_YCC) { cb = (px >> 8) & 0xFFu; cr = (px >> 16) & 0xFFu; }
    else fwd_c<ROUND>(px, cb, cr);
}

// Chroma-dependent part of the inverse transform, shared by all pixels that hold the same chroma.
//   R = clamp((298Y + 409(Cr-128) + 128) >> 8)                 = clamp((298Y + KR) >> 8)
//   G = clamp((298Y - 100(Cb-128) - 208(Cr-128) + 128) >> 8)   = clamp((298Y + KG) >> 8)
//   B = clamp((298Y + 516(Cb-128) + 128) >> 8)                 = clamp((298Y + KB) >> 8)
struct ChromaTerm {
    int kr, kg, kb;      // ARGB output
    uint32_t ycc_hi;     // YCC output: Cb << 8 | Cr << 16
};

// from already quantised Cb, Cr
template <int FMT>
__device__ __forceinline__ ChromaTerm chroma_term_q(uint32_t cb, uint32_t cr)
{
    ChromaTerm t;
    if (FMT == F_ARGB) {
        t.kr = __mul24((int)cr, 409) - 52224;
        t.kb = __mul24((int)cb, 516) - 65920;
        t.kg = 39552 - __mul24((int)cb, 100) - __mul24((int)cr, 208);
        t.ycc_hi = 0;
    } else {
        t.kr = t.kg = t.kb = 0;
        t.ycc_hi = (cb << 8) | (cr << 16);
    }
    return t;
}

template <int ROUND, int FMT>
__device__ __forceinline__ ChromaTerm chroma_term(uint32_t cpx, uint32_t mcb, uint32_t mcr)
{
    uint32_t cb, cr;
    fwd_c<ROUND>(cpx, cb, cr);
    return chroma_term_q<FMT>(cb & mcb, cr & mcr);      // quantiser, ColorQuantizer.scala:43-44
}

__device__ __forceinline__ int clamp_u8(int v) { return min(max(v, 0), 255); }   // -> v_med3_i32

// from an already quantised Y
template <int FMT>
__device__ __forceinline__ uint32_t finish_y(uint32_t y, const ChromaTerm &t)
{
    if (FMT == F_ARGB) {
        // clamp(x >> 8, 0, 255) is byte 1 of clamp(x, 0, 65535) -- x < 0 gives 0, x > 65535 gives 0xFFFF -- so the three shifts
        // and the shift-or packing become one v_med3_i32 per channel and two v_perm_b32 for the pixel (selector 0x0c = 0x00,
        // 0x0d = 0xFF): 8 VALU per pixel instead of 13 (round 4; SQ_INSTS_VALU per wave of the headline kernel 208 -> 192, and 154 with the
        // flat kernels' 32-bit addressing: profiles/r04_counters.json).  Checked on
        // all 2^24 (Y, Cb, Cr) by tests/test_gpu_parity.py::test_exhaustive_cube.
        const int yy = __mul24((int)y, 298);
        const uint32_t r = (uint32_t)min(max(yy + t.kr, 0), 65535);
        const uint32_t g = (uint32_t)min(max(yy + t.kg, 0), 65535);
        const uint32_t b = (uint32_t)min(max(yy + t.kb, 0), 65535);
        const uint32_t gb = __builtin_amdgcn_perm(g, b, 0x0c0c0501u);           // { b.byte1, g.byte1, 0, 0 }
        return __builtin_amdgcn_perm(r, gb, 0x0d050100u);                       // { b, g, r.byte1, 0xFF }
    } else {
        return y | t.ycc_hi;
    }
}

template <int FMT>
__device__ __forceinline__ uint32_t finish(uint32_t ypx, uint32_t my, const ChromaTerm &t)
{
    return finish_y<FMT>(fwd_y(ypx) & my, t);           // quantiser, ColorQuantizer.scala:42
}

// ------------------------------------------------------------------------------------------------
// AVG sampling extension: pieces shared by k_avg (csic_kernels.hip) and the planar kernels (csic_planar.hip)
// ------------------------------------------------------------------------------------------------
typedef unsigned short u16x2 __attribute__((ext_vector_type(2)));

// Cb | Cr << 16 of one pixel, both clamped: the two 9-bit quotients leave their dot products through ONE v_perm_b32 and are
// clamped by ONE v_pk_min_u16 (6 VALU ops for the pair instead of 8); all the sums below then run two channels to a register.
template <int ROUND>
__device__ __forceinline__ u16x2 fwd_c_pk(uint32_t px)
{
    uint32_t ub = __builtin_amdgcn_udot4(px ^ 0x00FFFF00u, 0x002B5580u /* A:0 R:43  G:85  B:128 */, 256u, false);
    uint32_t ur = __builtin_amdgcn_udot4(px ^ 0x0000FFFFu, 0x00806B15u /* A:0 R:128 G:107 B:21  */, 256u, false);
    if (ROUND == R_TRUNC) {                              // see fwd_c
        ub += (ub < 32768u ? 255u : 0u);
        ur += (ur < 32768u ? 255u : 0u);
    }
    const uint32_t pk = __builtin_amdgcn_perm(ur, ub, 0x06050201u);      // { ub[23:8], ur[23:8] }
    const u16x2 lim = {255, 255};
    return __builtin_elementwise_min(__builtin_bit_cast(u16x2, pk), lim);
}

// One output pixel by the definition (orc_process_avg verbatim): clamped coordinates everywhere.
template <int ROUND, int FMT, int INFMT>
__device__ __forceinline__ uint32_t avg_pixel_generic(const KArgs &a, gin_t in, int ro, int co)
{
    const int h = a.hmask + 1, v = a.vmask + 1, f = a.f;
    const int nlog = (h == 4 ? 2 : h == 2 ? 1 : 0) + (v == 2 ? 1 : 0);
    const int flog2 = 2 * a.sc_shift;
    uint32_t sy = 0, sb = 0, sr = 0;
    for (int i = 0; i < f; ++i) {
        for (int j = 0; j < f; ++j) {
            const int r = min(ro * f + i, a.H - 1), c = min(co * f + j, a.W - 1);
            sy += in_y<ROUND, INFMT>(in1<false>(a, in, (int64_t)r * a.ip + c));
            const int r0 = r & ~a.vmask, c0 = c & ~a.hmask;
            uint32_t ab = 0, ar = 0;
            for (int ii = 0; ii < v; ++ii) {
                for (int jj = 0; jj < h; ++jj) {
                    const int rr = min(r0 + ii, a.H - 1), cc = min(c0 + jj, a.W - 1);
                    uint32_t cb, cr;
                    in_c<ROUND, INFMT>(in1<false>(a, in, (int64_t)rr * a.ip + cc), cb, cr);
                    ab += cb; ar += cr;
                }
            }
            sb += (ab + ((h * v) >> 1)) >> nlog;
            sr += (ar + ((h * v) >> 1)) >> nlog;
        }
    }
    sy = ((sy + ((f * f) >> 1)) >> flog2) & a.my;
    sb = ((sb + ((f * f) >> 1)) >> flog2) & a.mcb;
    sr = ((sr + ((f * f) >> 1)) >> flog2) & a.mcr;
    return finish_y<FMT>(sy, chroma_term_q<FMT>(sb, sr));
}

} // namespace csic
