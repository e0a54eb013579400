// csic_planar.hip -- out_format = CSIC_FMT_PLANAR: the same pixel pipeline with a genuinely subsampled result (one Y byte
// per output pixel, one Cb and one Cr byte per chroma SAMPLE POINT), and csic_reconstruct_device, its inverse.  The layout
// and the reconstruct rule are defined in include/csic.h (csic_planar_layout); SURVEY.md 8 f3, VERDICT r03 item 4.
//
// Why it is reference-derivable although the reference never builds the format (ChromaSubsampler.scala:57-65 re-emits the held
// chroma with every pixel; README.md:35-46 only describes the subsampled wire format): the packed YCbCr stream the reference
// does produce is, by App. A.3, a pure function of Y at every position and Cb / Cr at the sample points -- the planar frame
// stores exactly those, and reconstruct replays the hold (the in-row hold and the 4:x:0 "odd rows replay the last sample of
// the row above" quirk), so  reconstruct(planar(x)) == packed(x)  bit for bit (tests/test_gpu_planar.py, against the oracle).
//
// Kernels (wave64; HBM-bound byte work, no LDS, no MFMA):
//   k_planar_flat<ROUND, MODE, NT>  HOLD_DECIMATE.  Lanes over groups of 4 consecutive positions of the OUTPUT stream, K = 4
//                      groups per lane spaced by the block size: 4 Y bytes leave as one 4-byte store.
//                        MODE 2  factor 1, W % 4 == 0 (the stream IS the image): one 16-byte load per group; Cb / Cr of a group
//                                leave as one 4-, 2- or 1-byte store (h = 1, 2, 4); rows without sample points (4:x:0 odd rows)
//                                compute Y only.  One-wave blocks: a wave's four loads cover 4 KiB of consecutive pixels.
//                        MODE 0  anything whose groups may straddle chroma rows (module_width % 4 != 0: spatial before chroma
//                                with ragged widths): 4-byte loads, per-position chroma byte stores.
//                        MODE 1  (CSIC_TUNE_VARIANT 10 only) the first form of k_planar_strided's job, kept for A/B.
//   k_planar_strided<ROUND, NT>  HOLD_DECIMATE with a factor >= 2 (module_width % 4 == 0): k_decflat's mapping -- 4 positions per
//                      lane spaced by the block size, loads at stride f (rows r % f != 0 are never read) -- and a 4 x 4 byte
//                      transpose inside each quad of lanes (DPP + v_perm), after which a lane owns 4 consecutive positions and
//                      stores Y, Cb, Cr as dwords.
//   k_planar_avg_f1<ROUND, HE, VE, NT>  AVG extension, factor 1, whole 4 x VE tiles: a lane owns a 4-pixel x VE-row tile
//                      (VE 16-byte loads), so every h x v chroma block lies in its registers: true 4:2:2 / 4:2:0 / 4:1:1
//                      box-filtered chroma -- what the north star's prose describes -- at 1.5-3 bytes per pixel out.
//   k_planar_avg_gen<ROUND>  AVG, anything else: one output position per lane by the definition (avg_pixel_generic).
//   k_recon<FMT, FAST, NT>  planar -> packed: K = 4 groups of 4 positions per lane; per group one 4-byte Y load and the aligned
//                      dword that holds the group's chroma samples (branch-free: shifted into place), one 16-byte store;
//                      FAST needs module_width % 4 == 0, otherwise position by position.
// Measured on 8192x8192 (profiles/r04_bench_all_configs.jsonl, r04_planar_bt.log): 4:2:0 at factor 1 -- 5.5 algorithmic bytes per pixel
// instead of 8 -- k_planar_flat 77 % of the 8 TB/s roofline (59.8 us per frame against 87.3 us for the packed k_f1x4),
// k_planar_avg_f1 77 %; factor 2 (3 bytes per output pixel) k_planar_strided 72-74 % (the 4-consecutive form: 45 %);
// k_recon 67-72 %.
// Algorithmic bytes: input as for the packed path (4 * W * ceil(H / f), AVG: 4 * W * H) + csic_planar_layout.payload_bytes;
// reconstruct: payload_bytes + 4 * n.
#include <cstdio>
#include <cstring>

#include "csic_kernel_ops.h"
#include "csic_avg_tile.h"

namespace csic {

typedef uint8_t CSIC_GLOBAL *gbyte_t;
typedef const uint8_t CSIC_GLOBAL *gcbyte_t;
typedef unsigned short CSIC_GLOBAL *gshort_t;
typedef const unsigned short CSIC_GLOBAL *gcshort_t;

#if defined(CSIC_DEBUG) && CSIC_DEBUG
#define CSIC_PCHECK(e, off, nbytes) CSIC_CHECK((off) >= 0 && (int64_t)(off) + (nbytes) <= (e).frame_bytes)
#else
#define CSIC_PCHECK(e, off, nbytes) do { } while (0)
#endif

template <bool NT> __device__ __forceinline__ void pst1(const PExtra &e, gbyte_t base, int64_t off, uint32_t v)
{
    CSIC_PCHECK(e, off, 1); (void)e;
    if (NT) __builtin_nontemporal_store((uint8_t)v, base + off); else base[off] = (uint8_t)v;
}
template <bool NT> __device__ __forceinline__ void pst2(const PExtra &e, gbyte_t base, int64_t off, uint32_t v)
{
    CSIC_PCHECK(e, off, 2); (void)e;
    if (NT) __builtin_nontemporal_store((unsigned short)v, (gshort_t)(base + off)); else *(gshort_t)(base + off) = (unsigned short)v;
}
template <bool NT> __device__ __forceinline__ void pst4(const PExtra &e, gbyte_t base, int64_t off, uint32_t v)
{
    CSIC_PCHECK(e, off, 4); (void)e;
    if (NT) __builtin_nontemporal_store(v, (gout_t)(base + off)); else *(gout_t)(base + off) = v;
}
__device__ __forceinline__ uint32_t pld1(const PExtra &e, gcbyte_t base, int64_t off) { CSIC_PCHECK(e, off, 1); (void)e; return base[off]; }
__device__ __forceinline__ uint32_t pld2(const PExtra &e, gcbyte_t base, int64_t off) { CSIC_PCHECK(e, off, 2); (void)e; return *(gcshort_t)(base + off); }
template <bool NT> __device__ __forceinline__ uint32_t pld4(const PExtra &e, gcbyte_t base, int64_t off)
{
    CSIC_PCHECK(e, off, 4); (void)e;
    return NT ? __builtin_nontemporal_load((gin_t)(base + off)) : *(gin_t)(base + off);
}

// base of the planar frame this block works on (grid z = frame): consecutive buffers, or -- frame-table mode -- whatever the
// device-resident table names (read through the constant address space: one wave-uniform s_load, as frame_in / frame_out)
__device__ __forceinline__ gbyte_t planar_frame(const PExtra &e)
{
    if (e.planar_tab) return (gbyte_t)((const uint64_t CSIC_CONSTANT *)(uintptr_t)e.planar_tab)[blockIdx.z];
    return (gbyte_t)(uintptr_t)e.planar + (int64_t)blockIdx.z * e.frame_bytes;
}

// ------------------------------------------------------------------------------------------------
// forward, HOLD_DECIMATE
// ------------------------------------------------------------------------------------------------
// input offset of output stream position j: decimated (ro, co) = (j / Wo, j % Wo) -> image (ro * f, co * f)
__device__ __forceinline__ int64_t stream_in_off(const KArgs &a, uint32_t j)
{
    const uint32_t ro = (uint32_t)(((uint64_t)j * a.mWo) >> a.kWo);
    const uint32_t co = j - ro * (uint32_t)a.Wo;
    return (int64_t)(ro * (uint32_t)a.f) * a.ip + co * (uint32_t)a.f;
}

// one position on its own (the stream's ragged tail, and every position of MODE 0)
template <int ROUND, bool NT>
__device__ __forceinline__ void planar_one(const KArgs &a, const PExtra &e, gin_t in, gbyte_t fb, uint32_t j, uint32_t px, bool store_y)
{
    if (store_y) pst1<NT>(e, fb, (int64_t)j, fwd_y(px) & a.my);
    const uint32_t r = (uint32_t)(((uint64_t)j * e.mWm) >> e.kWm), c = j - r * (uint32_t)e.Wm;
    if ((c & ((1u << e.lhe) - 1u)) == 0 && (r & ((1u << e.lve) - 1u)) == 0) {          // a sample point emits its OWN chroma
        uint32_t cb, cr;
        fwd_c<ROUND>(px, cb, cr);
        const int64_t k = (int64_t)(r >> e.lve) * e.Wc + (c >> e.lhe);
        pst1<NT>(e, fb, e.cb_off + k, cb & a.mcb);
        pst1<NT>(e, fb, e.cr_off + k, cr & a.mcr);
    }
    (void)in;
}

template <int ROUND, int MODE, int K, bool NT, bool CHECK>
__device__ __forceinline__ void planar_flat_body(const KArgs &a, const PExtra &e, gin_t in, gbyte_t fb, uint32_t g0, uint32_t T, uint32_t ngroups)
{
    const uint32_t n = (uint32_t)e.n;
    uint32_t px[K][4];
    bool live[K], full[K];
#pragma unroll
    for (int k = 0; k < K; ++k) {
        const uint32_t g = g0 + (uint32_t)k * T;
        live[k] = !CHECK || g < ngroups;
        const uint32_t gc = CHECK ? min(g, ngroups - 1) : g;                 // clamp instead of branching: the loads still issue back to back
        const uint32_t j0 = 4u * gc;
        full[k] = !CHECK || j0 + 3u < n;
        if (MODE == 2) {
            // factor 1, W % 4 == 0: the group is 4 consecutive pixels of one row (the stream IS the image)
            const u32x4 v = in4<NT>(a, in, stream_in_off(a, j0));
            px[k][0] = v.x; px[k][1] = v.y; px[k][2] = v.z; px[k][3] = v.w;
        } else {
#pragma unroll
            for (int q = 0; q < 4; ++q) px[k][q] = in1<NT>(a, in, stream_in_off(a, min(j0 + (uint32_t)q, n - 1u)));
        }
    }
#pragma unroll
    for (int k = 0; k < K; ++k) {
        if (!live[k]) continue;
        const uint32_t j0 = 4u * (g0 + (uint32_t)k * T);
        if (CHECK && !full[k]) {                                             // the stream's ragged tail: position by position
            for (uint32_t q = 0; q < 4u && j0 + q < n; ++q) planar_one<ROUND, NT>(a, e, in, fb, j0 + q, px[k][q], true);
            continue;
        }
        const uint32_t y4 = (fwd_y(px[k][0]) & a.my) | ((fwd_y(px[k][1]) & a.my) << 8) | ((fwd_y(px[k][2]) & a.my) << 16) |
                            ((fwd_y(px[k][3]) & a.my) << 24);
        pst4<NT>(e, fb, (int64_t)j0, y4);
        if (MODE == 0) {
#pragma unroll
            for (int q = 0; q < 4; ++q) planar_one<ROUND, NT>(a, e, in, fb, j0 + (uint32_t)q, px[k][q], false);
        } else {
            // module_width % 4 == 0: the group sits in ONE chroma row at a column that is a multiple of 4
            const uint32_t r = (uint32_t)(((uint64_t)j0 * e.mWm) >> e.kWm), c0 = j0 - r * (uint32_t)e.Wm;
            if ((r & ((1u << e.lve) - 1u)) != 0) continue;                   // no sample point in this row (wave-uniform almost everywhere)
            const int64_t k0 = (int64_t)(r >> e.lve) * e.Wc + (c0 >> e.lhe);
            uint32_t cb[4], cr[4];
            if (e.lhe == 0) {
#pragma unroll
                for (int q = 0; q < 4; ++q) { fwd_c<ROUND>(px[k][q], cb[q], cr[q]); cb[q] &= a.mcb; cr[q] &= a.mcr; }
                pst4<NT>(e, fb, e.cb_off + k0, cb[0] | (cb[1] << 8) | (cb[2] << 16) | (cb[3] << 24));
                pst4<NT>(e, fb, e.cr_off + k0, cr[0] | (cr[1] << 8) | (cr[2] << 16) | (cr[3] << 24));
            } else if (e.lhe == 1) {
                fwd_c<ROUND>(px[k][0], cb[0], cr[0]);
                fwd_c<ROUND>(px[k][2], cb[2], cr[2]);
                pst2<NT>(e, fb, e.cb_off + k0, (cb[0] & a.mcb) | ((cb[2] & a.mcb) << 8));
                pst2<NT>(e, fb, e.cr_off + k0, (cr[0] & a.mcr) | ((cr[2] & a.mcr) << 8));
            } else {
                fwd_c<ROUND>(px[k][0], cb[0], cr[0]);
                pst1<NT>(e, fb, e.cb_off + k0, cb[0] & a.mcb);
                pst1<NT>(e, fb, e.cr_off + k0, cr[0] & a.mcr);
            }
        }
    }
    if (!CHECK) keep_tail_apart();
}

constexpr int PLANAR_K = 4;

template <int ROUND, int MODE, bool NT>
__global__ void __launch_bounds__(256) k_planar_flat(KArgs a, PExtra e)
{
    pin_args(a);
    const uint32_t T = (uint32_t)e.T;
    const uint32_t ngroups = (uint32_t)((e.n + 3) >> 2);
    const uint32_t b0 = blockIdx.x * (T * PLANAR_K);
    const gin_t in = frame_in(a);
    const gbyte_t fb = planar_frame(e);
    // whole block inside the stream AND no ragged tail in it: the straight-line path
    if (b0 + T * PLANAR_K <= ngroups && (uint64_t)4 * (b0 + T * PLANAR_K) <= (uint64_t)e.n)
        planar_flat_body<ROUND, MODE, PLANAR_K, NT, false>(a, e, in, fb, b0 + threadIdx.x, T, ngroups);
    else
        planar_flat_body<ROUND, MODE, PLANAR_K, NT, true>(a, e, in, fb, b0 + threadIdx.x, T, ngroups);
}

// ------------------------------------------------------------------------------------------------
// k_planar_strided: factor f >= 2 (or any plan with module_width % 4 == 0 that is not the factor-1 image): k_decflat's mapping
// -- lane = 4 positions of the output stream spaced by the block size, so that every load instruction of a wave reads
// consecutive OUTPUT positions (input stride f * 4 bytes across lanes, the best the decimator allows) -- and then a 4 x 4
// transpose inside every quad of lanes, so that each lane ends up with 4 CONSECUTIVE positions and can store its Y (and Cb, Cr)
// bytes as one dword.  Loading 4 consecutive positions per lane instead (the first version of MODE 1) spread every load
// instruction over 4 x as many sectors at f = 2: 45 % of the roofline on 8192x8192 against 79 % for the packed k_decflat.
// The transpose is 4 DPP quad broadcasts + 2 v_perm + 1 v_lshl_or per plane: lane q of a quad collects byte q of its four lanes.
// ------------------------------------------------------------------------------------------------
__device__ __forceinline__ uint32_t quad_transpose_bytes(uint32_t w, uint32_t sel_lo, uint32_t sel_hi)
{
    const uint32_t a0 = (uint32_t)__builtin_amdgcn_update_dpp((int)w, (int)w, 0x00 /* quad_perm:[0,0,0,0] */, 0xF, 0xF, false);
    const uint32_t a1 = (uint32_t)__builtin_amdgcn_update_dpp((int)w, (int)w, 0x55 /* quad_perm:[1,1,1,1] */, 0xF, 0xF, false);
    const uint32_t a2 = (uint32_t)__builtin_amdgcn_update_dpp((int)w, (int)w, 0xAA /* quad_perm:[2,2,2,2] */, 0xF, 0xF, false);
    const uint32_t a3 = (uint32_t)__builtin_amdgcn_update_dpp((int)w, (int)w, 0xFF /* quad_perm:[3,3,3,3] */, 0xF, 0xF, false);
    // v_perm_b32(s0, s1, sel): selector bytes 0..3 pick from s1, 4..7 from s0
    const uint32_t lo = __builtin_amdgcn_perm(a1, a0, sel_lo);          // { a0.byte[q], a1.byte[q], 0, 0 }
    const uint32_t hi = __builtin_amdgcn_perm(a3, a2, sel_hi);          // { 0, 0, a2.byte[q], a3.byte[q] }
    return lo | hi;
}

template <int ROUND, bool NT, bool CHECK>
__device__ __forceinline__ void planar_strided_body(const KArgs &a, const PExtra &e, gin_t in, gbyte_t fb, uint32_t b0, uint32_t T)
{
    const uint32_t n = (uint32_t)e.n;
    const uint32_t tid = threadIdx.x, q = tid & 3u;
    uint32_t px[4];
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        const uint32_t j = b0 + (uint32_t)k * T + tid;
        px[k] = in1<NT>(a, in, stream_in_off(a, CHECK ? min(j, n - 1u) : j));
    }
    uint32_t wy = 0, wb = 0, wr = 0;                                       // this lane's 4 positions, byte k = position b0 + k * T + tid
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        uint32_t cb, cr;
        fwd_c<ROUND>(px[k], cb, cr);
        wy |= (fwd_y(px[k]) & a.my) << (8 * k);
        wb |= (cb & a.mcb) << (8 * k);
        wr |= (cr & a.mcr) << (8 * k);
    }
    // 0x0c = constant zero byte in v_perm_b32's selector
    const uint32_t sel_lo = 0x0c0c0000u | q | ((4u + q) << 8), sel_hi = 0x00000c0cu | (q << 16) | ((4u + q) << 24);
    const uint32_t y4 = quad_transpose_bytes(wy, sel_lo, sel_hi);
    const uint32_t cb4 = quad_transpose_bytes(wb, sel_lo, sel_hi);
    const uint32_t cr4 = quad_transpose_bytes(wr, sel_lo, sel_hi);
    // this lane now holds the group of 4 consecutive positions j0 .. j0 + 3 that its quad loaded in round k = q
    const uint32_t j0 = b0 + q * T + (tid & ~3u);
    if (CHECK && j0 >= n) return;
    if (CHECK && j0 + 3u >= n) {                                            // the stream's ragged tail: byte by byte
        for (uint32_t i = 0; j0 + i < n; ++i) {
            const uint32_t j = j0 + i;
            pst1<NT>(e, fb, (int64_t)j, (y4 >> (8 * i)) & 0xFFu);
            const uint32_t r = (uint32_t)(((uint64_t)j * e.mWm) >> e.kWm), c = j - r * (uint32_t)e.Wm;
            if ((c & ((1u << e.lhe) - 1u)) == 0 && (r & ((1u << e.lve) - 1u)) == 0) {
                const int64_t kk = (int64_t)(r >> e.lve) * e.Wc + (c >> e.lhe);
                pst1<NT>(e, fb, e.cb_off + kk, (cb4 >> (8 * i)) & 0xFFu);
                pst1<NT>(e, fb, e.cr_off + kk, (cr4 >> (8 * i)) & 0xFFu);
            }
        }
        return;
    }
    pst4<NT>(e, fb, (int64_t)j0, y4);
    // module_width % 4 == 0: the group sits in ONE chroma row at a column that is a multiple of 4
    const uint32_t r = (uint32_t)(((uint64_t)j0 * e.mWm) >> e.kWm), c0 = j0 - r * (uint32_t)e.Wm;
    if ((r & ((1u << e.lve) - 1u)) != 0) return;                            // a row without sample points
    const int64_t k0 = (int64_t)(r >> e.lve) * e.Wc + (c0 >> e.lhe);
    if (e.lhe == 0) {
        pst4<NT>(e, fb, e.cb_off + k0, cb4);
        pst4<NT>(e, fb, e.cr_off + k0, cr4);
    } else if (e.lhe == 1) {                                                // samples at positions 0 and 2 of the group
        pst2<NT>(e, fb, e.cb_off + k0, (cb4 & 0xFFu) | ((cb4 >> 8) & 0xFF00u));
        pst2<NT>(e, fb, e.cr_off + k0, (cr4 & 0xFFu) | ((cr4 >> 8) & 0xFF00u));
    } else {
        pst1<NT>(e, fb, e.cb_off + k0, cb4 & 0xFFu);
        pst1<NT>(e, fb, e.cr_off + k0, cr4 & 0xFFu);
    }
    if (!CHECK) keep_tail_apart();
}

template <int ROUND, bool NT>
__global__ void __launch_bounds__(256) k_planar_strided(KArgs a, PExtra e)
{
    pin_args(a);
    const uint32_t T = (uint32_t)e.T;                   // a multiple of 4: quads of lanes are quads of positions
    const uint32_t b0 = blockIdx.x * (T * 4u);
    const gin_t in = frame_in(a);
    const gbyte_t fb = planar_frame(e);
    // (every lane takes part in the quad exchange: no early exit before it)
    if ((uint64_t)b0 + (uint64_t)T * 4u <= (uint64_t)e.n) planar_strided_body<ROUND, NT, false>(a, e, in, fb, b0, T);
    else                                                   planar_strided_body<ROUND, NT, true>(a, e, in, fb, b0, T);
}

// ------------------------------------------------------------------------------------------------
// forward, AVG extension
// ------------------------------------------------------------------------------------------------
// factor 1, W % 4 == 0, H % VE == 0: a lane owns a 4-pixel x VE-row tile; K tiles per lane spaced by the block width
template <int ROUND, int HE, int VE, bool NT>
__global__ void __launch_bounds__(256) k_planar_avg_f1(KArgs a, PExtra e)
{
    constexpr int K = 2;
    constexpr int NLOG = (HE == 4 ? 2 : HE == 2 ? 1 : 0) + (VE == 2 ? 1 : 0);
    pin_args(a);
    const int W4 = a.W >> 2;
    const int x0 = blockIdx.x * (a.bdx * K) + threadIdx.x;
    if (x0 >= W4) return;
    const gin_t in = frame_in(a);
    const gbyte_t fb = planar_frame(e);
    const int ntr = a.H / VE;
    const u16x2 qmask = {(unsigned short)a.mcb, (unsigned short)a.mcr};
    for (int tr = blockIdx.y * a.bdy + threadIdx.y; tr < ntr; tr += a.row_step) {
        u32x4 p[K][VE];
#pragma unroll
        for (int t = 0; t < K; ++t) {
            const int x4 = min(x0 + t * a.bdx, W4 - 1);
#pragma unroll
            for (int i = 0; i < VE; ++i) p[t][i] = in4<NT>(a, in, (int64_t)(tr * VE + i) * a.ip + 4 * x4);
        }
#pragma unroll
        for (int t = 0; t < K; ++t) {
            const int x4 = x0 + t * a.bdx;
            if (x4 >= W4) continue;
            u16x2 C[VE][4];
#pragma unroll
            for (int i = 0; i < VE; ++i) {
                const uint32_t px[4] = {p[t][i].x, p[t][i].y, p[t][i].z, p[t][i].w};
                uint32_t y4 = 0;
#pragma unroll
                for (int j = 0; j < 4; ++j) { y4 |= (fwd_y(px[j]) & a.my) << (8 * j); C[i][j] = fwd_c_pk<ROUND>(px[j]); }
                pst4<NT>(e, fb, (int64_t)(tr * VE + i) * a.W + 4 * x4, y4);
            }
            // HE x VE box averages: 4 / HE samples per tile
            uint32_t cb = 0, cr = 0;
            const u16x2 half = {(HE * VE) >> 1, (HE * VE) >> 1};
#pragma unroll
            for (int bj = 0; bj < 4; bj += HE) {
                u16x2 s = {0, 0};
#pragma unroll
                for (int i = 0; i < VE; ++i)
#pragma unroll
                    for (int j = 0; j < HE; ++j) s += C[i][bj + j];
                s = ((s + half) >> (unsigned short)NLOG) & qmask;
                cb |= (uint32_t)s.x << (8 * (bj / HE));
                cr |= (uint32_t)s.y << (8 * (bj / HE));
            }
            const int64_t k0 = (int64_t)tr * e.Wc + x4 * (4 / HE);
            if (HE == 1) { pst4<NT>(e, fb, e.cb_off + k0, cb); pst4<NT>(e, fb, e.cr_off + k0, cr); }
            else if (HE == 2) { pst2<NT>(e, fb, e.cb_off + k0, cb); pst2<NT>(e, fb, e.cr_off + k0, cr); }
            else { pst1<NT>(e, fb, e.cb_off + k0, cb); pst1<NT>(e, fb, e.cr_off + k0, cr); }
        }
    }
}

// any factor, any shape with a whole tile: k_avg's body (csic_avg_tile.h: 4 x max(f, v) tiles in registers, edge blocks for the
// cut tiles) computing packed YCbCr -- under AVG every pixel carries its chroma block's average -- with a sink that scatters the
// bytes into the planes instead of packing pixels: the Y byte of every position, Cb / Cr at the sample points, which the AVG
// layout puts at every (max(1, h / f))-th column of every (max(1, v / f))-th row of the output (csic.h).  A tile yields 4 / f
// consecutive positions of max(f, v) / f rows (one position per lane pair at f = 8).  Stores are as wide as the run is long where
// the plane's rows keep them aligned (wave-uniform), single bytes otherwise.
template <int LHE, int LVE>
struct PlanarSink {
    const PExtra &e;
    gbyte_t fb;
    int Wo;
    // n bytes (n = 1, 2, 4; v holds them little-endian) at byte offset off of the planar frame
    template <int NB> __device__ __forceinline__ void run(int64_t off, uint32_t v, bool aligned) const
    {
        if (NB == 1) pst1<true>(e, fb, off, v & 0xFFu);
        else if (aligned) { if (NB == 2) pst2<true>(e, fb, off, v); else pst4<true>(e, fb, off, v); }
        else {
#pragma unroll
            for (int i = 0; i < NB; ++i) pst1<true>(e, fb, off + i, (v >> (8 * i)) & 0xFFu);
        }
    }
    template <int N> __device__ __forceinline__ void put(int row, int col, const uint32_t (&o)[N]) const
    {
        static_assert(N == 1 || N == 2 || N == 4, "a tile yields 1, 2 or 4 positions per row");
        constexpr int NS = (N >> LHE) > 0 ? (N >> LHE) : 1;             // chroma samples among the N positions (N < hold_h: one or none)
        uint32_t y = 0, cb = 0, cr = 0;
#pragma unroll
        for (int i = 0; i < N; ++i) y |= (o[i] & 0xFFu) << (8 * i);
#pragma unroll
        for (int i = 0; i < NS; ++i) {
            cb |= ((o[i << LHE] >> 8) & 0xFFu) << (8 * i);
            cr |= ((o[i << LHE] >> 16) & 0xFFu) << (8 * i);
        }
        // col is a multiple of N: a run is aligned iff its plane's rows are a multiple of its length apart
        run<N>((int64_t)row * Wo + col, y, (Wo & (N - 1)) == 0);
        if ((row & ((1 << LVE) - 1)) == 0 && (N >= (1 << LHE) || (col & ((1 << LHE) - 1)) == 0)) {
            const int64_t k = (int64_t)(row >> LVE) * e.Wc + (col >> LHE);
            const bool al = (e.Wc & (NS - 1)) == 0;
            run<NS>(e.cb_off + k, cb, al);
            run<NS>(e.cr_off + k, cr, al);
        }
    }
    __device__ __forceinline__ void put_edge(int row, int col, uint32_t v) const
    {
        pst1<false>(e, fb, (int64_t)row * Wo + col, v & 0xFFu);
        if ((col & ((1 << LHE) - 1)) == 0 && (row & ((1 << LVE) - 1)) == 0) {
            const int64_t k = (int64_t)(row >> LVE) * e.Wc + (col >> LHE);
            pst1<false>(e, fb, e.cb_off + k, (v >> 8) & 0xFFu);
            pst1<false>(e, fb, e.cr_off + k, (v >> 16) & 0xFFu);
        }
    }
};

template <int ROUND, int F, int HH, int VV>
__global__ void __launch_bounds__(256) k_planar_avg_tile(KArgs a, PExtra e)
{
    constexpr int HOLD_H = HH > F ? HH / F : 1, HOLD_V = VV > F ? VV / F : 1;        // csic_planar_layout under AVG
    constexpr int LHE = HOLD_H == 4 ? 2 : HOLD_H == 2 ? 1 : 0, LVE = HOLD_V == 2 ? 1 : 0;
    pin_args(a);
    const PlanarSink<LHE, LVE> sink{e, planar_frame(e), a.Wo};
    avg_kernel_body<ROUND, F_YCC, F, HH, VV, true, (F <= 2 ? 2 : 1)>(a, frame_in(a), sink);
}

// any factor, any shape: one output position per lane by the definition
template <int ROUND>
__global__ void __launch_bounds__(256) k_planar_avg_gen(KArgs a, PExtra e)
{
    pin_args(a);
    const int co = blockIdx.x * a.bdx + threadIdx.x;
    if (co >= a.Wo) return;
    const gin_t in = frame_in(a);
    const gbyte_t fb = planar_frame(e);
    for (int ro = blockIdx.y * a.bdy + threadIdx.y; ro < a.Ho; ro += a.row_step) {
        const uint32_t ycc = avg_pixel_generic<ROUND, F_YCC, F_ARGB>(a, in, ro, co);    // Y | Cb << 8 | Cr << 16, quantised
        pst1<false>(e, fb, (int64_t)ro * a.Wo + co, ycc & 0xFFu);
        if ((co & ((1 << e.lhe) - 1)) == 0 && (ro & ((1 << e.lve) - 1)) == 0) {           // module_width == Wo under AVG
            const int64_t k = (int64_t)(ro >> e.lve) * e.Wc + (co >> e.lhe);
            pst1<false>(e, fb, e.cb_off + k, (ycc >> 8) & 0xFFu);
            pst1<false>(e, fb, e.cr_off + k, (ycc >> 16) & 0xFFu);
        }
    }
}

// ------------------------------------------------------------------------------------------------
// reconstruct: planar -> packed
// ------------------------------------------------------------------------------------------------
// chroma sample index of stream position (r, c) -- csic.h, csic_planar_layout
__device__ __forceinline__ int64_t recon_index(const PExtra &e, uint32_t r, uint32_t c)
{
    if ((r & ((1u << e.lve) - 1u)) == 0 || !e.replay_last) return (int64_t)(r >> e.lve) * e.Wc + (c >> e.lhe);
    return (int64_t)((r - 1u) >> e.lve) * e.Wc + (e.Wc - 1);          // ChromaSubsampler.scala:52-65: the last sample of the row above
}

// one group of (up to) 4 positions, position by position: the stream's ragged tail, and groups that may straddle chroma rows
template <int FMT, bool NT>
__device__ __forceinline__ void recon_slow(const PExtra &e, gcbyte_t fb, gout_t out, uint32_t j0, uint32_t n)
{
    uint32_t o[4];
    const uint32_t cnt = min(4u, n - j0);
    for (uint32_t q = 0; q < cnt; ++q) {
        const uint32_t j = j0 + q;
        const uint32_t r = (uint32_t)(((uint64_t)j * e.mWm) >> e.kWm), c = j - r * (uint32_t)e.Wm;
        const int64_t k = recon_index(e, r, c);
        o[q] = finish_y<FMT>(pld1(e, fb, (int64_t)j), chroma_term_q<FMT>(pld1(e, fb, e.cb_off + k), pld1(e, fb, e.cr_off + k)));
    }
    if (cnt == 4u) { const u32x4 ov = {o[0], o[1], o[2], o[3]}; st4<NT>(out + j0, ov); }
    else for (uint32_t q = 0; q < cnt; ++q) out[j0 + q] = o[q];
}

#ifndef CSIC_RECON_K
#define CSIC_RECON_K 4
#endif
constexpr int RECON_K = CSIC_RECON_K;

// K groups per lane spaced by the block size: all their loads (one Y dword and the group's chroma samples each) are in flight
// before the first inverse transform starts.  The first version took one group per lane -- 8 bytes in flight per lane -- and
// ran at 58 % of the roofline on a kernel that is three quarters stores.
template <int FMT, bool FAST, bool NT, bool CHECK>
__device__ __forceinline__ void recon_body(const PExtra &e, gcbyte_t fb, gout_t out, uint32_t g0, uint32_t T, uint32_t ngroups)
{
    const uint32_t n = (uint32_t)e.n;
    uint32_t y4[RECON_K], cbw[RECON_K], crw[RECON_K], sh[RECON_K];
    bool live[RECON_K], fast[RECON_K];
#pragma unroll
    for (int k = 0; k < RECON_K; ++k) {
        const uint32_t g = g0 + (uint32_t)k * T;
        live[k] = !CHECK || g < ngroups;
        const uint32_t j0 = 4u * (CHECK ? min(g, ngroups - 1u) : g);
        fast[k] = FAST && (!CHECK || j0 + 3u < n);
        y4[k] = cbw[k] = crw[k] = sh[k] = 0;
        if (fast[k]) {
            // module_width % 4 == 0: the group lies in one chroma row at a column that is a multiple of 4, so its samples are
            // 4, 2 or 1 consecutive bytes (h = 1, 2, 4) -- or ONE byte on a row that replays the last sample of the row above.
            // Branch-free: the aligned dword that holds them is loaded and shifted (planes are padded to 256 bytes), position q
            // then takes byte q >> sh.  With a branch per case the K groups' loads did not stay in flight together.
            y4[k] = pld4<NT>(e, fb, (int64_t)j0);
            const uint32_t r = (uint32_t)(((uint64_t)j0 * e.mWm) >> e.kWm), c0 = j0 - r * (uint32_t)e.Wm;
            const bool replay = (r & ((1u << e.lve) - 1u)) != 0 && e.replay_last;
            const int64_t k0 = replay ? (int64_t)((r - 1u) >> e.lve) * e.Wc + (e.Wc - 1) : (int64_t)(r >> e.lve) * e.Wc + (c0 >> e.lhe);
            sh[k] = replay ? 2u : (uint32_t)e.lhe;
            const uint32_t bit = 8u * (uint32_t)(k0 & 3);
            cbw[k] = pld4<false>(e, fb, (e.cb_off + k0) & ~(int64_t)3) >> bit;
            crw[k] = pld4<false>(e, fb, (e.cr_off + k0) & ~(int64_t)3) >> bit;
        }
    }
#pragma unroll
    for (int k = 0; k < RECON_K; ++k) {
        if (!live[k]) continue;
        const uint32_t j0 = 4u * (g0 + (uint32_t)k * T);
        if (!fast[k]) { recon_slow<FMT, NT>(e, fb, out, j0, n); continue; }
        // the chroma half of the inverse transform once per DISTINCT sample of the group (4, 2 or 1), not once per pixel: the kernel
        // is as much VALU as memory (27 VALU per pixel on 5.5 bytes with a term per pixel; 69 % of the roofline)
        uint32_t o[4];
        const uint32_t y0 = y4[k] & 0xFFu, y1 = (y4[k] >> 8) & 0xFFu, y2 = (y4[k] >> 16) & 0xFFu, y3 = y4[k] >> 24;
        if (sh[k] == 2u) {
            const ChromaTerm t = chroma_term_q<FMT>(cbw[k] & 0xFFu, crw[k] & 0xFFu);
            o[0] = finish_y<FMT>(y0, t); o[1] = finish_y<FMT>(y1, t); o[2] = finish_y<FMT>(y2, t); o[3] = finish_y<FMT>(y3, t);
        } else if (sh[k] == 1u) {
            const ChromaTerm t0 = chroma_term_q<FMT>(cbw[k] & 0xFFu, crw[k] & 0xFFu);
            const ChromaTerm t1 = chroma_term_q<FMT>((cbw[k] >> 8) & 0xFFu, (crw[k] >> 8) & 0xFFu);
            o[0] = finish_y<FMT>(y0, t0); o[1] = finish_y<FMT>(y1, t0); o[2] = finish_y<FMT>(y2, t1); o[3] = finish_y<FMT>(y3, t1);
        } else {
            o[0] = finish_y<FMT>(y0, chroma_term_q<FMT>(cbw[k] & 0xFFu, crw[k] & 0xFFu));
            o[1] = finish_y<FMT>(y1, chroma_term_q<FMT>((cbw[k] >> 8) & 0xFFu, (crw[k] >> 8) & 0xFFu));
            o[2] = finish_y<FMT>(y2, chroma_term_q<FMT>((cbw[k] >> 16) & 0xFFu, (crw[k] >> 16) & 0xFFu));
            o[3] = finish_y<FMT>(y3, chroma_term_q<FMT>(cbw[k] >> 24, crw[k] >> 24));
        }
        const u32x4 ov = {o[0], o[1], o[2], o[3]};
        st4<NT>(out + j0, ov);
    }
    if (!CHECK) keep_tail_apart();
}

template <int FMT, bool FAST, bool NT>
__global__ void __launch_bounds__(256) k_recon(KArgs a, PExtra e)
{
    pin_args(a);
    (void)a;
    const uint32_t T = (uint32_t)e.T;
    const uint32_t ngroups = (uint32_t)((e.n + 3) >> 2);
    const uint32_t b0 = blockIdx.x * (T * RECON_K);
    const gcbyte_t fb = (gcbyte_t)planar_frame(e);
    const gout_t out = (gout_t)(uintptr_t)e.packed + (int64_t)blockIdx.z * e.n;
    if (b0 + T * RECON_K <= ngroups && (uint64_t)4 * (b0 + T * RECON_K) <= (uint64_t)e.n)
        recon_body<FMT, FAST, NT, false>(e, fb, out, b0 + threadIdx.x, T, ngroups);
    else
        recon_body<FMT, FAST, NT, true>(e, fb, out, b0 + threadIdx.x, T, ngroups);
}

// ------------------------------------------------------------------------------------------------
// host side
// ------------------------------------------------------------------------------------------------
using PlanarFn = void (*)(KArgs, PExtra);

static int ilog2(int x) { int l = 0; while ((1 << l) < x) ++l; return l; }

static void fill_extra(const csic_planar_layout &L, PExtra *e)
{
    std::memset(e, 0, sizeof *e);
    e->cb_off = L.cb_offset; e->cr_off = L.cr_offset; e->frame_bytes = L.frame_bytes;
    e->n = (int64_t)L.y_width * L.y_height;
    e->Wm = L.module_width; e->Wc = L.chroma_width; e->lhe = ilog2(L.hold_h); e->lve = ilog2(L.hold_v); e->replay_last = L.replay_last;
    magic_div((uint32_t)L.module_width, &e->mWm, &e->kWm);
}

// which forward kernel a plan takes: 0 = flat MODE 0, 1 = strided, 2 = flat MODE 2, 3 = avg_f1, 4 = avg_gen, 5 = flat MODE 1, 6 = avg_tile
static int forward_kind(const csic_plan *pl, const csic_planar_layout &L)
{
    const csic_params &p = plan_params(pl);
    const Geometry &g = plan_geometry(pl);
    const bool general = plan_variant(pl) == 9;                 // CSIC_TUNE_VARIANT 9: the general kernels (A/B, tests)
    if (p.sampling == CSIC_SAMPLING_AVG) {
        const bool f1_whole = g.f == 1 && g.W % 4 == 0 && g.H % g.v == 0;
        // factor 1 on frames of whole tiles: the dedicated kernel (Y bytes straight from the loaded pixels, no per-pixel write-back of
        // the block averages) is 3 points ahead of k_avg's body there -- 8192x8192 4:2:0: 77.5 against 74.1 % of the roofline,
        // profiles/r04_planar_avg_tile_ab.log; CSIC_TUNE_VARIANT 12 takes the tile body all the same (A/B, tests)
        if (!general && f1_whole && (plan_variant(pl) != 12 || !plan_nontemporal(pl))) return 3;
        if (!general && plan_nontemporal(pl)) {                  // k_avg's body wherever k_avg itself would run
            LaunchDesc d;
            bool tile = false;
            if (planar_avg_geometry(pl, 1, &d, &tile) == CSIC_OK && tile) return 6;
        }
        return (!general && f1_whole) ? 3 : 4;
    }
    if (general || L.module_width % 4 != 0) return 0;
    if (g.f == 1 && g.W % 4 == 0) return 2;
    return plan_variant(pl) == 10 ? 5 : 1;
}

template <int ROUND, bool NT>
static PlanarFn pick_forward(int kind, int he, int ve)
{
    switch (kind) {
    case 0: return k_planar_flat<ROUND, 0, NT>;
    case 1: return k_planar_strided<ROUND, NT>;
    case 5: return k_planar_flat<ROUND, 1, NT>;           // the first form of kind 1 (4 consecutive positions per lane): A/B, CSIC_TUNE_VARIANT 10
    case 2: return k_planar_flat<ROUND, 2, NT>;
    case 3:
        if (ve == 1) return he == 1 ? k_planar_avg_f1<ROUND, 1, 1, NT> : he == 2 ? k_planar_avg_f1<ROUND, 2, 1, NT> : k_planar_avg_f1<ROUND, 4, 1, NT>;
        return he == 1 ? k_planar_avg_f1<ROUND, 1, 2, NT> : he == 2 ? k_planar_avg_f1<ROUND, 2, 2, NT> : k_planar_avg_f1<ROUND, 4, 2, NT>;
    default: return k_planar_avg_gen<ROUND>;
    }
}

template <int ROUND, int F>
static PlanarFn pick_avg_tile_f(int h, int v)
{
    if (v == 1) return h == 1 ? k_planar_avg_tile<ROUND, F, 1, 1> : h == 2 ? k_planar_avg_tile<ROUND, F, 2, 1> : k_planar_avg_tile<ROUND, F, 4, 1>;
    return h == 1 ? k_planar_avg_tile<ROUND, F, 1, 2> : h == 2 ? k_planar_avg_tile<ROUND, F, 2, 2> : k_planar_avg_tile<ROUND, F, 4, 2>;
}
template <int ROUND>
static PlanarFn pick_avg_tile(int f, int h, int v)
{
    return f == 1 ? pick_avg_tile_f<ROUND, 1>(h, v) : f == 2 ? pick_avg_tile_f<ROUND, 2>(h, v)
         : f == 4 ? pick_avg_tile_f<ROUND, 4>(h, v) : pick_avg_tile_f<ROUND, 8>(h, v);
}

void planar_kernel_name(const csic_plan *pl, char *buf, size_t len)
{
    csic_planar_layout L;
    planar_layout(plan_geometry(pl), &plan_params(pl), &L);
    const char *rn = plan_params(pl).rounding == CSIC_ROUND_FLOOR_HW ? "floor" : "trunc";
    const char *nt = plan_nontemporal(pl) ? "nt" : "cached";
    switch (forward_kind(pl, L)) {
    case 0: snprintf(buf, len, "k_planar_flat<%s,general,h%d,v%d,%s>", rn, L.hold_h, L.hold_v, nt); break;
    case 1: snprintf(buf, len, "k_planar_strided<%s,f%d,h%d,v%d,%s>", rn, plan_geometry(pl).f, L.hold_h, L.hold_v, nt); break;
    case 5: snprintf(buf, len, "k_planar_flat<%s,f%d,h%d,v%d,%s>", rn, plan_geometry(pl).f, L.hold_h, L.hold_v, nt); break;
    case 2: snprintf(buf, len, "k_planar_flat<%s,f1x4,h%d,v%d,%s>", rn, L.hold_h, L.hold_v, nt); break;
    case 3: snprintf(buf, len, "k_planar_avg_f1<%s,h%d,v%d,%s>", rn, L.hold_h, L.hold_v, nt); break;
    case 6: snprintf(buf, len, "k_planar_avg_tile<%s,f%d,h%d,v%d,nt>", rn, plan_geometry(pl).f, plan_geometry(pl).h, plan_geometry(pl).v); break;
    default: snprintf(buf, len, "k_planar_avg_gen<%s,h%d,v%d>", rn, L.hold_h, L.hold_v); break;
    }
}

static int launch_pk(PlanarFn fn, dim3 grid, dim3 block, KArgs a, PExtra e, hipStream_t stream)
{
    void *params[2] = {&a, &e};
    HIP_TRY(hipLaunchKernel(reinterpret_cast<const void *>(fn), grid, block, params, 0, stream));
    return CSIC_OK;
}

// kernel, grid, block and both argument blocks of one forward launch over nz <= 65535 frames; the frame pointers stay unset
static int planar_resolve(const csic_plan *pl, int nz, PlanarLaunchDesc *d)
{
    const csic_params &p = plan_params(pl);
    const Geometry &g = plan_geometry(pl);
    csic_planar_layout L;
    planar_layout(g, &p, &L);
    const int kind = forward_kind(pl, L);
    const bool nt = plan_nontemporal(pl);
    const bool floor_r = p.rounding == CSIC_ROUND_FLOOR_HW;
    PlanarFn fn = floor_r ? (nt ? pick_forward<R_FLOOR, true>(kind, L.hold_h, L.hold_v) : pick_forward<R_FLOOR, false>(kind, L.hold_h, L.hold_v))
                          : (nt ? pick_forward<R_TRUNC, true>(kind, L.hold_h, L.hold_v) : pick_forward<R_TRUNC, false>(kind, L.hold_h, L.hold_v));
    if (kind == 6) fn = floor_r ? pick_avg_tile<R_FLOOR>(g.f, g.h, g.v) : pick_avg_tile<R_TRUNC>(g.f, g.h, g.v);
    KArgs &a = d->args;
    PExtra &e = d->extra;
    fill_base_args(g, g.W, g.Wo, &a);
    fill_extra(L, &e);
    dim3 grid, block;
    if (kind == 6) {
        // k_avg's own geometry (block shape, edge blocks, XCD rotation): prepare_common through the plan's packed twin
        LaunchDesc ld;
        bool tile = false;
        const int st = planar_avg_geometry(pl, nz, &ld, &tile);
        if (st != CSIC_OK) return st;
        if (!tile) return set_error(CSIC_EHIP, "internal: the planar AVG tile kernel was selected for a plan k_avg does not take");
        a = ld.args;
        grid = ld.grid; block = ld.block;
        e.T = 256;
    } else if (kind == 1) {
        // k_planar_strided: 4 positions per lane, T * 4 positions per block
        const int bt = plan_block_threads(pl);
        const int T = (bt == 64 || bt == 128 || bt == 256) ? bt : 256;
        e.T = T;
        block = dim3((unsigned)T, 1, 1);
        grid = dim3((unsigned)((e.n + (int64_t)T * 4 - 1) / ((int64_t)T * 4)), 1, (unsigned)nz);
        a.bdx = T; a.bdy = 1; a.row_step = 1;
    } else if (kind <= 2 || kind == 5) {
        const int64_t ngroups = (e.n + 3) / 4;
        const int bt = plan_block_threads(pl);
        // one-wave blocks for the 16-byte-load kernel: a wave's four loads then cover 4 KiB of consecutive pixels
        // (8192x8192 4:2:0: 71.7 % of the roofline with 256-thread blocks, 77.6 % with 64; profiles/r04_planar_bt.log)
        const int T = (bt == 64 || bt == 128 || bt == 256) ? bt : (kind == 2 ? 64 : 256);
        e.T = T;
        block = dim3((unsigned)T, 1, 1);
        grid = dim3((unsigned)((ngroups + (int64_t)T * PLANAR_K - 1) / ((int64_t)T * PLANAR_K)), 1, (unsigned)nz);
        a.bdx = T; a.bdy = 1; a.row_step = 1;
    } else {
        // row-tiled kernels: lanes along x (tiles of 4 pixels for avg_f1 with 2 tiles per lane, output pixels for avg_gen)
        const int lanes_x = kind == 3 ? (g.W / 4 + 1) / 2 : g.Wo;
        const int rows = kind == 3 ? g.H / g.v : g.Ho;
        int bx = 1;
        while (bx < lanes_x && bx < 256) bx <<= 1;
        const int by = 256 / bx;
        unsigned gy = (unsigned)((rows + by - 1) / by);
        if (gy > 65535u) gy = 65535u;
        block = dim3(bx, by, 1);
        grid = dim3((unsigned)((lanes_x + bx - 1) / bx), gy, (unsigned)nz);
        a.bdx = bx; a.bdy = by; a.row_step = (int32_t)gy * by;
        e.T = 256;
    }
    d->fn = fn; d->grid = grid; d->block = block;
    return CSIC_OK;
}

int planar_enqueue(const PlanarLaunchDesc &d, hipStream_t stream) { return launch_pk(d.fn, d.grid, d.block, d.args, d.extra, stream); }

int planar_forward(const csic_plan *pl, const void *d_in, void *d_planar, int nframes, hipStream_t stream)
{
    if (!d_in || !d_planar) return set_error(CSIC_EINVAL_NULL, "device buffer is NULL");
    if (((uintptr_t)d_planar & 255u) || ((uintptr_t)d_in & 3u))
        return set_error(CSIC_EINVAL_SIZE, "a planar frame buffer must be 256-byte aligned (and the input 4-byte aligned)");
    for (int f0 = 0; f0 < nframes; f0 += 65535) {            // grid z limit
        const int nz = nframes - f0 < 65535 ? nframes - f0 : 65535;
        PlanarLaunchDesc d;
        int st = planar_resolve(pl, nz, &d);
        if (st != CSIC_OK) return st;
        d.args.in = static_cast<const uint32_t *>(d_in) + (int64_t)f0 * d.args.in_frame_px;
        d.extra.planar = static_cast<uint8_t *>(d_planar) + (int64_t)f0 * d.extra.frame_bytes;
        st = planar_enqueue(d, stream);
        if (st != CSIC_OK) return st;
    }
    return CSIC_OK;
}

int planar_prepare_table(const csic_plan *pl, const void *const *d_in_tab, void *const *d_planar_tab, uintptr_t align_bits, int nframes,
                         PlanarLaunchDesc *d)
{
    if (!pl) return set_error(CSIC_EINVAL_NULL, "plan is NULL");
    if (!d_in_tab || !d_planar_tab) return set_error(CSIC_EINVAL_NULL, "frame table is NULL");
    if (nframes <= 0 || nframes > 65535) return set_error(CSIC_EINVAL_SIZE, "nframes per launch must be in 1..65535. Got %d", nframes);
    if (align_bits & 3u) return set_error(CSIC_EINVAL_SIZE, "frame buffers must be 4-byte aligned");
    const int st = planar_resolve(pl, nframes, d);
    if (st != CSIC_OK) return st;
    d->args.in_tab = reinterpret_cast<const uint32_t *const *>(d_in_tab);
    d->extra.planar_tab = reinterpret_cast<const uint64_t *>(d_planar_tab);
    return CSIC_OK;
}

} // namespace csic

using namespace csic;

extern "C" int csic_reconstruct_device(csic_plan *plan, const void *d_planar, void *d_out, int32_t nframes, int32_t out_format,
                                       void *hip_stream)
{
    if (!plan) return set_error(CSIC_EINVAL_NULL, "plan is NULL");
    if (!d_planar || !d_out) return set_error(CSIC_EINVAL_NULL, "device buffer is NULL");
    if (nframes <= 0) return set_error(CSIC_EINVAL_SIZE, "nframes must be positive. Got %d", nframes);
    if (out_format != CSIC_FMT_ARGB8888 && out_format != CSIC_FMT_YCBCR888X)
        return set_error(CSIC_EINVAL_FORMAT, "csic_reconstruct_device writes ARGB8888(0) or YCBCR888X(1). Got %d", out_format);
    if (((uintptr_t)d_planar & 255u) || ((uintptr_t)d_out & 15u))
        return set_error(CSIC_EINVAL_SIZE, "a planar frame buffer must be 256-byte aligned and the packed output 16-byte aligned");
    const csic_params &p = plan_params(plan);
    const Geometry &g = plan_geometry(plan);
    csic_planar_layout L;
    planar_layout(g, &p, &L);
    CSIC_DEVICE_SCOPE(plan_device(plan));
    const bool fast = L.module_width % 4 == 0 && plan_variant(plan) != 9;
    const bool nt = plan_nontemporal(plan);
    PlanarFn fn;
    if (out_format == CSIC_FMT_ARGB8888)
        fn = fast ? (nt ? k_recon<F_ARGB, true, true> : k_recon<F_ARGB, true, false>) : (nt ? k_recon<F_ARGB, false, true> : k_recon<F_ARGB, false, false>);
    else
        fn = fast ? (nt ? k_recon<F_YCC, true, true> : k_recon<F_YCC, true, false>) : (nt ? k_recon<F_YCC, false, true> : k_recon<F_YCC, false, false>);
    hipStream_t stream = static_cast<hipStream_t>(hip_stream);
    for (int f0 = 0; f0 < nframes; f0 += 65535) {
        const int nz = nframes - f0 < 65535 ? nframes - f0 : 65535;
        KArgs a;
        fill_base_args(g, g.W, g.Wo, &a);
        PExtra e;
        fill_extra(L, &e);
        e.planar = const_cast<uint8_t *>(static_cast<const uint8_t *>(d_planar)) + (int64_t)f0 * L.frame_bytes;
        e.packed = static_cast<uint32_t *>(d_out) + (int64_t)f0 * e.n;
        const int bt = plan_block_threads(plan);
        const int T = (bt == 64 || bt == 128 || bt == 256) ? bt : 64;
        e.T = T;
        const int64_t ngroups = (e.n + 3) / 4, per_block = (int64_t)T * RECON_K;
        const int st = launch_pk(fn, dim3((unsigned)((ngroups + per_block - 1) / per_block), 1, (unsigned)nz), dim3((unsigned)T, 1, 1), a, e, stream);
        if (st != CSIC_OK) return st;
    }
    clear_error();
    return CSIC_OK;
}
