// csic_kernels.hip -- the fused pixel pipeline for gfx950 (MI355X) and the device half of the C ABI.
//
// One launch turns packed ARGB input pixels into packed reconstructed ARGB (or YCbCr) output pixels:
//   forward RGB->YCbCr  (RGB2YCbCr.scala:33-65 floor form / :95-121 trunc form)
//   chroma sample-and-hold (ChromaSubsampler.scala:47-65), closed form of SURVEY.md App. A.3
//   top-left decimation  (SpatialDownsampler.scala:33-55)
//   LSB-truncating quantiser (ColorQuantizer.scala:29-31,42-44)
//   inverse YCbCr->RGB   (YCbCr2RGB.scala:17-26; applied per output pixel by the reference's
//                         harness, ImageCompressorTopApp.scala:118)
// Every output pixel is a pure function of at most two input pixels (the one that supplies Y and the
// one whose chroma is held), so the stream is embarrassingly parallel; the kernels are HBM-bound
// integer/byte work -- no MFMA, no LDS staging (there is no reuse to stage).
//
// Kernel families (all written for wave64; 256-thread blocks laid out bx * by, bx chosen so that rows tile
// exactly whenever possible):
//   k_dec    : the workhorse.  One lane = K (= 4) output pixels spaced by the block width, so every load
//              and store instruction is lane-contiguous in the OUTPUT (4-byte nt loads with stride f*4
//              bytes across lanes, dense 4-byte nt stores); only rows r % f == 0 are touched.  The in-row
//              chroma hold is a DPP quad_perm on the loaded pixel, the 4:x:0 / spatial-before-chroma
//              "replay one pixel for the whole row" cases a per-row broadcast.  Serves f = 2/4/8 in both
//              order classes (spatial-before-chroma when f | W and h | Wo) and f = 1 for 4:x:0, any width
//              and pointers that are only 4-byte aligned.
//   k_f1flat : factor 1, width % 4 == 0 (round 4).  Lanes over groups of 4 consecutive pixels of the flat frame, 4 groups per lane
//              spaced by the one-wave block: four 16-byte nt loads in flight per lane, a wave covers 4 KiB of consecutive pixels;
//              held chroma is reused inside the group (h in {2,4} divides 4), 4:x:0 odd rows fetch one row-uniform pixel.
//   k_f1x4   : its predecessor (one group per lane, 2-D blocks over rows); CSIC_TUNE_VARIANT 11, A/B only.
//   k_dec2v  : factor 2 variants with 16-byte loads (tuning knob, not the default).
//   k_generic: one lane = one output pixel, run-time parameters, SURVEY.md App. A.3/A.4 verbatim: the
//              remaining spatial-before-chroma shapes (chroma counters run on the decimated stream
//              modulo the FULL width, ImageCompressorTop.scala:52-58), frames narrower than a quad, and a
//              YCbCr input stream (single-stage driving, the reference's spec style).
//   k_avg / k_avg_generic : the AVG sampling EXTENSION (box-filter chroma + average pooling) -- not
//              reference semantics, never the default; see the section comment further down.
//
// Citations are relative to /root/reference/.
#include <cstdio>
#include <cstring>
#include <new>
#include <tuple>
#include <utility>

#include "csic_kernel_ops.h"
#include "csic_avg_tile.h"

namespace csic {

// ------------------------------------------------------------------------------------------------
// k_f1x4: factor 1, W % 4 == 0, 4 pixels per lane
// ------------------------------------------------------------------------------------------------
template <int ROUND, int FMT, int HH, int VV, bool NT>
__global__ void __launch_bounds__(256) k_f1x4(KArgs a)
{
    pin_args(a);
    const int W4 = a.W >> 2;
    const int x4 = blockIdx.x * a.bdx + threadIdx.x;
    if (x4 >= W4) return;
    const gin_t in = frame_in(a);
    const gout_t out = frame_out(a);
    const int row_step = a.row_step;
    for (int row = blockIdx.y * a.bdy + threadIdx.y; row < a.H; row += row_step) {
        const int64_t base = (int64_t)row * a.ip + 4 * x4;
        const int64_t obase = (int64_t)row * a.op + 4 * x4;
        // 4:x:0 odd row: no pixel is a sample point, the whole row replays the chroma latched at the last
        // sample of the previous row (ChromaSubsampler.scala:52-65).  Its load is
        // issued ahead of the 16-byte stream load so that the two latencies overlap.
        const bool odd = (VV == 2) && (row & 1);
        uint32_t cpx = 0;
        if (odd) cpx = in1<false>(a, in, (int64_t)(row - 1) * a.ip + a.last_sample_col);
        const u32x4 p = in4<NT>(a, in, base);
        const uint32_t px[4] = {p.x, p.y, p.z, p.w};
        uint32_t o[4];
        if (odd) {
            const ChromaTerm t = chroma_term<ROUND, FMT>(cpx, a.mcb, a.mcr);
#pragma unroll
            for (int i = 0; i < 4; ++i) o[i] = finish<FMT>(px[i], a.my, t);
        } else {
#pragma unroll
            for (int g = 0; g < 4; g += HH) {
                const ChromaTerm t = chroma_term<ROUND, FMT>(px[g], a.mcb, a.mcr);
#pragma unroll
                for (int i = g; i < g + HH; ++i) o[i] = finish<FMT>(px[i], a.my, t);
            }
        }
        const u32x4 ov = {o[0], o[1], o[2], o[3]};
        out4<NT>(a, out, obase, ov);
    }
}

// ------------------------------------------------------------------------------------------------
// k_f1flat: factor 1, W % 4 == 0 -- k_f1x4's arithmetic on k_planar_flat's mapping (round 4): lanes over groups of 4 consecutive
// pixels of the flat frame, K = 4 groups per lane spaced by the one-wave block, so a wave's four 16-byte loads (and stores)
// cover 4 KiB of consecutive pixels and four loads per lane are in flight.  Group -> (row, column) by an exact multiply-shift.
// ------------------------------------------------------------------------------------------------
template <int ROUND, int FMT, int HH, int VV, bool NT, bool CHECK>
__device__ __forceinline__ void f1flat_body(const KArgs &a, gin_t in, gout_t out, uint32_t g0, uint32_t T, uint32_t ngroups)
{
    constexpr int K = 4;
    u32x4 p[K];
    uint32_t cpx[K], row[K], off[K], oo[K];
    bool odd[K];
    // 32-bit offsets and stepping instead of K divisions, as in decflat_body: T groups = 4 T pixels = qT rows + rT columns
    const uint32_t W = (uint32_t)a.W, ip = (uint32_t)a.ip, op = (uint32_t)a.op;
    const uint32_t qT = (uint32_t)(((uint64_t)(4u * T) * a.mW) >> a.kW), rT = 4u * T - qT * W;
    const uint32_t in_n = qT * ip + rT, in_w = in_n + ip - W, out_n = qT * op + rT, out_w = out_n + op - W;
    uint32_t r = 0, col = 0, io = 0, ooff = 0;
#pragma unroll
    for (int k = 0; k < K; ++k) {
        if (CHECK || k == 0) {
            const uint32_t g = CHECK ? min(g0 + (uint32_t)k * T, ngroups - 1u) : g0;
            const uint32_t j0 = 4u * g;
            r = (uint32_t)(((uint64_t)j0 * a.mW) >> a.kW);
            col = j0 - __umul24(r, W);
            io = __umul24(r, ip) + col;                                          // rows and pitches fit 24 bits (prepare_common)
            ooff = __umul24(r, op) + col;
        } else {
            const bool wrap = col + rT >= W;
            col = wrap ? col + rT - W : col + rT;
            r += qT + (wrap ? 1u : 0u);
            io += wrap ? in_w : in_n;
            ooff += wrap ? out_w : out_n;
        }
        row[k] = r;
        off[k] = io;
        oo[k] = ooff;
        odd[k] = (VV == 2) && (r & 1u);
    }
    // 4:x:0 odd rows replay the last sample of the row above: one row-uniform pixel.  Issued AHEAD of the 16-byte stream loads (as
    // in k_f1x4), only by waves that sit on such a row, and once per row a lane visits -- its K groups are T * 4 pixels apart,
    // usually inside one row.
#pragma unroll
    for (int k = 0; k < K; ++k) {
        cpx[k] = 0;
        if (VV == 2) {
            const bool fresh = (k == 0) || row[k] != row[k > 0 ? k - 1 : 0];
            if (__builtin_amdgcn_ballot_w64(odd[k] && fresh) != 0)
                cpx[k] = in1n<false>(a, in, odd[k] ? __umul24(row[k] - 1u, ip) + (uint32_t)a.last_sample_col : off[k]);
            if (k > 0 && !fresh) cpx[k] = cpx[k - 1];
        }
    }
#pragma unroll
    for (int k = 0; k < K; ++k) p[k] = in4n<NT>(a, in, off[k]);
#pragma unroll
    for (int k = 0; k < K; ++k) {
        if (CHECK && g0 + (uint32_t)k * T >= ngroups) continue;
        const uint32_t px[4] = {p[k].x, p[k].y, p[k].z, p[k].w};
        uint32_t o[4];
        if (odd[k]) {
            const ChromaTerm t = chroma_term<ROUND, FMT>(cpx[k], a.mcb, a.mcr);
#pragma unroll
            for (int i = 0; i < 4; ++i) o[i] = finish<FMT>(px[i], a.my, t);
        } else {
#pragma unroll
            for (int g = 0; g < 4; g += HH) {
                const ChromaTerm t = chroma_term<ROUND, FMT>(px[g], a.mcb, a.mcr);
#pragma unroll
                for (int i = g; i < g + HH; ++i) o[i] = finish<FMT>(px[i], a.my, t);
            }
        }
        const u32x4 ov = {o[0], o[1], o[2], o[3]};
        out4n<NT>(a, out, oo[k], ov);
    }
    if (!CHECK) keep_tail_apart();
}

template <int ROUND, int FMT, int HH, int VV, bool NT>
__global__ void __launch_bounds__(256) k_f1flat(KArgs a)
{
    pin_args(a);
    const uint32_t T = (uint32_t)a.bdx;
    const uint32_t ngroups = ((uint32_t)a.W >> 2) * (uint32_t)a.H;
    const uint32_t b0 = blockIdx.x * (T * 4u);
    const gin_t in = frame_in(a);
    const gout_t out = frame_out(a);
    if (b0 + T * 4u <= ngroups) f1flat_body<ROUND, FMT, HH, VV, NT, false>(a, in, out, b0 + threadIdx.x, T, ngroups);
    else                        f1flat_body<ROUND, FMT, HH, VV, NT, true>(a, in, out, b0 + threadIdx.x, T, ngroups);
}

// ------------------------------------------------------------------------------------------------
// k_dec: factor F in {2,4,8}, both order classes.  K output pixels per lane.
//
// Chroma source of output pixel (ro, co), in DECIMATED coordinates:
//   chroma before spatial: the hold runs on image columns c = co*F: source column c - c % h.  With
//     h <= F that is the pixel itself (the chroma stage is unobservable); the one
//     exception, 4:1:1 with F = 2, is decimated column co & ~1.  Rows ro*F are always sample rows.
//   spatial before chroma (SROWS): the chroma counters run on the decimated stream modulo the FULL
//     width W (ImageCompressorTop.scala:52-58).  When F | W and h | Wo, chroma row r = ro / F; on
//     r % v == 0 the source is decimated column co & ~(h-1) of the same row; on odd r (4:x:0) every
//     pixel of the F decimated rows replays ONE pixel: the last sample of chroma row r-1.
// In both classes the in-row hold is a lane shuffle inside an aligned quad (HOLD in {1,2,4} lanes):
// one v_mov_b32 DPP quad_perm on the loaded pixel instead of a second gather.
// ------------------------------------------------------------------------------------------------
template <int HOLD>
__device__ __forceinline__ uint32_t hold_in_quad(uint32_t v)
{
    if (HOLD == 2) return (uint32_t)__builtin_amdgcn_update_dpp((int)v, (int)v, 0xA0 /* quad_perm:[0,0,2,2] */, 0xF, 0xF, false);
    if (HOLD == 4) return (uint32_t)__builtin_amdgcn_update_dpp((int)v, (int)v, 0x00 /* quad_perm:[0,0,0,0] */, 0xF, 0xF, false);
    return v;
}

// One row-chunk of k_dec.  CHECK = false is the block-uniform fast path (the whole chunk is inside the
// row): K independent loads are issued back to back, then K stores, with no per-lane branches -- with
// per-lane bounds checks hipcc wraps every access in its own exec-mask region and puts
// s_waitcnt vmcnt(0) in front of every store, which serialises the stores (38.0 vs 32.7 us per frame).
// BCAST: the whole row replays one chroma pixel (`bpx`).
template <int ROUND, int FMT, int F, int HOLD, bool BCAST, int K, bool NT, bool CHECK>
__device__ __forceinline__ void dec_chunk(const KArgs &a, gin_t in, int64_t rowoff, gout_t out, int64_t orowoff, int co0, int bx,
                                          uint32_t bpx)
{
    uint32_t px[K];
#pragma unroll
    for (int k = 0; k < K; ++k) {
        // partial chunk: clamp the column instead of branching, so the K loads still issue back to back
        // (out-of-row lanes re-read the last pixel of the row and simply do not store)
        const int co = CHECK ? min(co0 + k * bx, a.Wo - 1) : co0 + k * bx;
        px[k] = in1<NT>(a, in, rowoff + co * F);
    }
    if (BCAST) {
        const ChromaTerm t = chroma_term<ROUND, FMT>(bpx, a.mcb, a.mcr);
#pragma unroll
        for (int k = 0; k < K; ++k) {
            const int co = co0 + k * bx;
            if (!CHECK || co < a.Wo) out1<NT>(a, out, orowoff + co, finish<FMT>(px[k], a.my, t));
        }
    } else {
        uint32_t cpx[K];
#pragma unroll
        for (int k = 0; k < K; ++k) cpx[k] = hold_in_quad<HOLD>(px[k]);   // all lanes; a source lane <= its reader
#pragma unroll
        for (int k = 0; k < K; ++k) {
            const int co = co0 + k * bx;
            if (!CHECK || co < a.Wo) {
                const ChromaTerm t = chroma_term<ROUND, FMT>(cpx[k], a.mcb, a.mcr);
                out1<NT>(a, out, orowoff + co, finish<FMT>(px[k], a.my, t));
            }
        }
    }
}

template <int ROUND, int FMT, int F, int HOLD, bool SROWS, int K, bool NT, bool CHECK>
__device__ __forceinline__ void dec_rows(const KArgs &a, gin_t in, gout_t out, int co0, int bx,
                                         int ro0, int row_step)
{
    for (int ro = ro0; ro < a.Ho; ro += row_step) {
        const int64_t rowoff = (int64_t)(ro * F) * a.ip;
        const int64_t orowoff = (int64_t)ro * a.op;
        if (SROWS) {
            const int r = ro >> a.sc_shift;                               // chroma row = ro / F
            if (r & a.vmask) {                                            // odd chroma row of 4:x:0
                const int srow = ((r - 1) << a.sc_shift) + a.bc_row_off; // decimated row of the held sample
                const uint32_t bpx = in1<false>(a, in, (int64_t)(srow * F) * a.ip + a.bc_col_in);
                dec_chunk<ROUND, FMT, F, HOLD, true, K, NT, CHECK>(a, in, rowoff, out, orowoff, co0, bx, bpx);
                keep_tail_apart();                                       // (or its last store is merged with the other chunk's)
                continue;
            }
        }
        dec_chunk<ROUND, FMT, F, HOLD, false, K, NT, CHECK>(a, in, rowoff, out, orowoff, co0, bx, 0u);
    }
}

template <int ROUND, int FMT, int F, int HOLD, bool SROWS, int K, bool NT>
__global__ void __launch_bounds__(256) k_dec(KArgs a)
{
    pin_args(a);
    const int bx = a.bdx;
    const int cbase = blockIdx.x * (bx * K);
    const int co0 = cbase + threadIdx.x;
    const gin_t in = frame_in(a);
    const gout_t out = frame_out(a);
    const int row_step = a.row_step;
    const int ro0 = blockIdx.y * a.bdy + threadIdx.y;
    // two separate row loops (the condition is uniform over the block) so that the fast path keeps
    // its own straight-line load/store sequence
    if (cbase + bx * K <= a.Wo) dec_rows<ROUND, FMT, F, HOLD, SROWS, K, NT, false>(a, in, out, co0, bx, ro0, row_step);
    else                        dec_rows<ROUND, FMT, F, HOLD, SROWS, K, NT, true>(a, in, out, co0, bx, ro0, row_step);
}

// ------------------------------------------------------------------------------------------------
// k_decflat: factor F in {2,4,8}, both order classes wherever k_dec's fast path applies (and HOLD | Wo) -- first written for
// chroma before spatial with h <= F (the chroma stage is unobservable, no cross-lane hold) on rows that
// k_dec cannot cut into whole blocks -- Wo % K != 0 (1000-wide frames at f = 4 / 8: Wo = 250 / 125) or a lane count with no
// usable divisor (Wo = 1028: 257 lanes) --, for rows of a few partly filled waves (125, 90, 44, 160 lanes) and, since it is
// level or slightly ahead there too, for rows of whole waves: everything but k_dec's one-wave-block shapes (dec_prefers_flat).  There k_dec puts its last chunk -- for narrow rows EVERY chunk -- on the
// bounds-checked path, whose exec-mask regions and s_waitcnt vmcnt(0) before every store serialise the wave (1000x1000
// f = 8: 64.8 % of the HBM roofline against 80.5 % for 1024x1024, with HBM traffic only 1.05x the algorithmic bytes:
// profiles/r03_pmc_summary / pmc_traffic.json sq1000_csq, sq1024_csq).  Here the lanes cover the flat DECIMATED stream
// instead: output index i -> (ro, co) = (i / Wo, i % Wo) by an exact multiply-shift (magic_div), K indices per lane spaced
// by the block size, so every block but the frame's last runs the straight-line path and a wave stores 256 contiguous
// bytes per instruction whatever the row width (row ends no longer split stores into partial lines: WRITE_SIZE was 1.29x
// the output bytes for k_dec on these rows).
// ------------------------------------------------------------------------------------------------
// HOLD / SROWS as in k_dec.  The in-row hold stays a DPP inside an aligned quad of FLAT indices: a block starts at a multiple of
// 4, and with HOLD | Wo a hold group never straddles two rows.  SROWS (spatial before chroma, 4:x:0): on an odd chroma row --
// (ro / F) odd -- every pixel replays the last sample of the chroma row above; lanes of one wave may sit in different rows
// here, so the held pixel is a per-lane load whose address is SELECTED (own pixel on even rows: a cache hit), not branched on.
// Addressing (round 4): the flat kernels are launched only for frames whose extents fit 2^30 pixels and whose rows and pitches fit
// 24 bits (prepare_common sends anything larger to the row kernels), so a pixel's BYTE offset inside its frame is a uint32 and
// every access takes the "scalar base + 32-bit lane offset" form of global_load / global_store; row * pitch is one full-rate
// v_mad_u32_u24.  Only the first of a lane's K indices is divided (the exact multiply-shift); the others follow by stepping:
// T = qT * Wo + rT, so (ro, co) += (qT, rT) with one conditional wrap, and the two offsets advance by one of two wave-uniform
// strides each.  The straight-line path of the headline kernel: 192 -> 154 VALU instructions per lane (4 output pixels), 64-bit
// multiply-adds 20 -> 1, quarter-rate v_mul_lo_u32 4 -> 0; k_f1flat 625 -> 601 (16 pixels), k_flatgen 341 -> 299.  Timing is
// unchanged -- these kernels wait for HBM -- so this buys headroom, not speed (profiles/r04_narrow_offsets_ab.log).  Two things the
// A/B runs showed on the way: (1) written with plain 32-bit `ro * op + co`, hipcc formed the low half of a v_mad_u64_u32 whose
// don't-care high addend it parked in the register the lane's FIRST load writes -- a false dependency, an s_waitcnt on that load in
// front of the fourth load, 79.4 -> 77.1 % on the headline; the 24-bit multiplies leave nothing undefined to park.  (2) Issuing
// the K loads back to back after ALL the offset arithmetic (a sched_barrier) instead of interleaved with it costs the headline
// half a point to two points, whichever part of the arithmetic is moved behind the loads.
template <int ROUND, int FMT, int F, int HOLD, bool SROWS, int K, bool NT, bool CHECK>
__device__ __forceinline__ void decflat_body(const KArgs &a, gin_t in, gout_t out, uint32_t i0, uint32_t T, uint32_t n)
{
    uint32_t px[K], hp[K], oo[K], ho[K];
    bool odd[K];
    const uint32_t Wo = (uint32_t)a.Wo, ip = (uint32_t)a.ip, op = (uint32_t)a.op, rstep = (uint32_t)F * ip;
    // wave-uniform: how far one step of T indices moves (row, column) and the two offsets, without and with a wrap
    const uint32_t qT = (uint32_t)(((uint64_t)T * a.mWo) >> a.kWo), rT = T - qT * Wo;
    const uint32_t in_n = qT * rstep + rT * F, in_w = in_n + rstep - Wo * F;
    const uint32_t out_n = qT * op + rT, out_w = out_n + op - Wo;
    uint32_t ro = 0, co = 0, yoff = 0, ooff = 0;
#pragma unroll
    for (int k = 0; k < K; ++k) {
        if (CHECK || k == 0) {
            // the frame's last block: clamp instead of branching so that the K loads still issue back to back
            const uint32_t i = CHECK ? min(i0 + (uint32_t)k * T, n - 1) : i0;
            ro = (uint32_t)(((uint64_t)i * a.mWo) >> a.kWo);                     // i / Wo, exact for i < 2^31
            co = i - __umul24(ro, Wo);
            yoff = (__umul24(ro, ip) + co) * F;                                  // rows and pitches fit 24 bits (prepare_common):
            ooff = __umul24(ro, op) + co;                                        // one full-rate v_mad_u32_u24 each
        } else {
            const bool wrap = co + rT >= Wo;
            co = wrap ? co + rT - Wo : co + rT;
            ro += qT + (wrap ? 1u : 0u);
            yoff += wrap ? in_w : in_n;
            ooff += wrap ? out_w : out_n;
        }
        px[k] = in1n<NT>(a, in, yoff);
        oo[k] = ooff;
        if (SROWS) {
            const int r = (int)(ro >> a.sc_shift);                                // chroma row = ro / F
            odd[k] = (r & a.vmask) != 0;
            const int srow = ((r - 1) << a.sc_shift) + a.bc_row_off;             // decimated row of the held sample
            ho[k] = odd[k] ? __umul24((uint32_t)srow, rstep) + (uint32_t)a.bc_col_in : yoff;
        }
    }
    if (SROWS) {
        // the held pixels, only for waves that have a lane on an odd chroma row (a wave-uniform branch: chroma rows are F
        // decimated rows tall, so most waves are all-even -- no second load at all -- or all-odd -- one address for the wave)
#pragma unroll
        for (int k = 0; k < K; ++k) {
            hp[k] = 0;
            if (__builtin_amdgcn_ballot_w64(odd[k]) != 0) hp[k] = in1n<false>(a, in, ho[k]);
        }
    }
    uint32_t cpx[K];
#pragma unroll
    for (int k = 0; k < K; ++k) cpx[k] = hold_in_quad<HOLD>(px[k]);              // all lanes; a source lane <= its reader
#pragma unroll
    for (int k = 0; k < K; ++k) {
        if (!CHECK || i0 + (uint32_t)k * T < n) {
            const uint32_t c = (SROWS && odd[k]) ? hp[k] : cpx[k];
            const ChromaTerm t = chroma_term<ROUND, FMT>(c, a.mcb, a.mcr);
            out1n<NT>(a, out, oo[k], finish<FMT>(px[k], a.my, t));
        }
    }
    if (!CHECK) keep_tail_apart();
}

template <int ROUND, int FMT, int F, int HOLD, bool SROWS, int K, bool NT>
__global__ void __launch_bounds__(256) k_decflat(KArgs a)
{
    pin_args(a);
    const uint32_t T = (uint32_t)a.bdx;
    const uint32_t n = (uint32_t)a.Wo * (uint32_t)a.Ho;
    const uint32_t b0 = blockIdx.x * (T * K);
    const gin_t in = frame_in(a);
    const gout_t out = frame_out(a);
    if (b0 + T * K <= n) decflat_body<ROUND, FMT, F, HOLD, SROWS, K, NT, false>(a, in, out, b0 + threadIdx.x, T, n);
    else                 decflat_body<ROUND, FMT, F, HOLD, SROWS, K, NT, true>(a, in, out, b0 + threadIdx.x, T, n);
}

// ------------------------------------------------------------------------------------------------
// k_dec2v: factor 2, chroma before spatial, h <= 2, W % 8 == 0 -- 16-byte loads.
//   VAR 1: one lane = 4 input pixels (one dwordx4 load)  -> 2 output pixels (one dwordx2 store),
//          loads and stores both dense across lanes.
//   VAR 2: one lane = 8 input pixels (two dwordx4 loads of its own 32 contiguous bytes)
//          -> 4 output pixels (one dwordx4 store); load instructions are 50 % dense, stores dense.
// ------------------------------------------------------------------------------------------------
template <int ROUND, int FMT, int VAR, bool NT>
__global__ void __launch_bounds__(256) k_dec2v(KArgs a)
{
    constexpr int OPL = (VAR == 1) ? 2 : 4;             // output pixels per lane
    pin_args(a);
    const int nx = a.Wo / OPL;                          // lanes per row (W % 8 == 0 -> exact)
    const int x = blockIdx.x * a.bdx + threadIdx.x;
    if (x >= nx) return;
    const gin_t in = frame_in(a);
    const gout_t out = frame_out(a);
    const int row_step = a.row_step;
    for (int ro = blockIdx.y * a.bdy + threadIdx.y; ro < a.Ho; ro += row_step) {
        const int64_t ioff = (int64_t)(ro * 2) * a.ip + (int64_t)x * (OPL * 2);
        const int64_t ooff = (int64_t)ro * a.op + (int64_t)x * OPL;
        if (VAR == 1) {
            const u32x4 p = in4<NT>(a, in, ioff);
            const ChromaTerm t0 = chroma_term<ROUND, FMT>(p.x, a.mcb, a.mcr);
            const ChromaTerm t1 = chroma_term<ROUND, FMT>(p.z, a.mcb, a.mcr);
            const u32x2 ov = {finish<FMT>(p.x, a.my, t0), finish<FMT>(p.z, a.my, t1)};
            out2<NT>(a, out, ooff, ov);
        } else {
            const u32x4 p = in4<NT>(a, in, ioff);
            const u32x4 q = in4<NT>(a, in, ioff + 4);
            const ChromaTerm t0 = chroma_term<ROUND, FMT>(p.x, a.mcb, a.mcr);
            const ChromaTerm t1 = chroma_term<ROUND, FMT>(p.z, a.mcb, a.mcr);
            const ChromaTerm t2 = chroma_term<ROUND, FMT>(q.x, a.mcb, a.mcr);
            const ChromaTerm t3 = chroma_term<ROUND, FMT>(q.z, a.mcb, a.mcr);
            const u32x4 ov = {finish<FMT>(p.x, a.my, t0), finish<FMT>(p.z, a.my, t1),
                              finish<FMT>(q.x, a.my, t2), finish<FMT>(q.z, a.my, t3)};
            out4<NT>(a, out, ooff, ov);
        }
    }
}

// ------------------------------------------------------------------------------------------------
// AVG sampling extension (CSIC_SAMPLING_AVG) -- NOT reference semantics: csic_avg_tile.h holds the tile kernel's body (shared
// with the planar output format); here it stores packed pixels.
// ------------------------------------------------------------------------------------------------
template <bool NT>
struct PackedSink {
    const KArgs &a;
    gout_t out;
    template <int N> __device__ __forceinline__ void put(int row, int col, const uint32_t (&o)[N]) const
    {
        const int64_t oo = (int64_t)row * a.op + col;
        if (N == 4) { const u32x4 ov = {o[0], o[1 % N], o[2 % N], o[3 % N]}; out4<NT>(a, out, oo, ov); }
        else if (N == 2) { const u32x2 ov = {o[0], o[1 % N]}; out2<NT>(a, out, oo, ov); }
        else out1<NT>(a, out, oo, o[0]);
    }
    __device__ __forceinline__ void put_edge(int row, int col, uint32_t v) const { out1<false>(a, out, (int64_t)row * a.op + col, v); }
};

template <int ROUND, int FMT, int F, int HH, int VV, bool NT, int TILES = (F <= 2 ? 2 : 1)>
__global__ void __launch_bounds__(256) k_avg(KArgs a)
{
    pin_args(a);
    const PackedSink<NT> sink{a, frame_out(a)};
    avg_kernel_body<ROUND, FMT, F, HH, VV, NT, TILES>(a, frame_in(a), sink);
}

// Any shape and a YCbCr input stream: one output pixel per lane.
template <int ROUND, int FMT, int INFMT>
__global__ void __launch_bounds__(256) k_avg_generic(KArgs a)
{
    pin_args(a);
    const int co = blockIdx.x * a.bdx + threadIdx.x;
    if (co >= a.Wo) return;
    const gin_t in = frame_in(a);
    const gout_t out = frame_out(a);
    for (int ro = blockIdx.y * a.bdy + threadIdx.y; ro < a.Ho; ro += a.row_step)
        out1<false>(a, out, (int64_t)ro * a.op + co, avg_pixel_generic<ROUND, FMT, INFMT>(a, in, ro, co));
}

// ------------------------------------------------------------------------------------------------
// k_generic: any parameters, one output pixel per lane (SURVEY.md App. A.3 / A.4 verbatim)
// ------------------------------------------------------------------------------------------------
template <int ROUND, int FMT, int INFMT>
__global__ void __launch_bounds__(256) k_generic(KArgs a)
{
    pin_args(a);
    const int co = blockIdx.x * a.bdx + threadIdx.x;
    if (co >= a.Wo) return;
    const gin_t in = frame_in(a);
    const gout_t out = frame_out(a);
    const int row_step = a.row_step;
    for (int ro = blockIdx.y * a.bdy + threadIdx.y; ro < a.Ho; ro += row_step) {
        const int64_t y_idx = (int64_t)(ro * a.f) * a.ip + co * a.f;
        int64_t c_idx;
        if (!a.s_first) {
            const int r = ro * a.f, c = co * a.f;       // chroma counters == image coordinates
            c_idx = ((r & a.vmask) == 0) ? (int64_t)r * a.ip + (c & ~a.hmask)
                                         : (int64_t)(r - 1) * a.ip + a.last_sample_col;
        } else {
            // chroma sits behind the decimator but was built with the full width
            // (ImageCompressorTop.scala:52-58): its column counter wraps every W decimated pixels.
            // The two divisions by run-time constants are exact multiply-shifts (magic_div, host side).
            const int j = ro * a.Wo + co;               // < 2^31 (validated)
            const int r = (int)(((uint64_t)(uint32_t)j * a.mW) >> a.kW), c = j - r * a.W;
            const int src = ((r & a.vmask) == 0) ? (j - (c & a.hmask)) : ((r - 1) * a.W + a.last_sample_col);
            const int sro = (int)(((uint64_t)(uint32_t)src * a.mWo) >> a.kWo), sco = src - sro * a.Wo;
            c_idx = (int64_t)(sro * a.f) * a.ip + sco * a.f;       // (counters above use the semantic W, addresses the pitch)
        }
        uint32_t cb, cr;
        in_c<ROUND, INFMT>(in1<false>(a, in, c_idx), cb, cr);
        const uint32_t y = in_y<ROUND, INFMT>(in1<false>(a, in, y_idx)) & a.my;
        out1<false>(a, out, (int64_t)ro * a.op + co, finish_y<FMT>(y, chroma_term_q<FMT>(cb & a.mcb, cr & a.mcr)));
    }
}

// ------------------------------------------------------------------------------------------------
// k_flatgen: the plans k_decflat's fast conditions exclude -- spatial before chroma where f does not divide W or h does not
// divide Wo (the chroma counters run over the decimated stream modulo the FULL width, ImageCompressorTop.scala:52-58, so a hold
// group starts at an arbitrary lane and may straddle two decimated rows) -- on the same mapping: lanes over the flat decimated
// stream, K indices per lane.  k_generic fetches the chroma source of every pixel with a second gather (58-68 % of the HBM
// roofline on 1000x1000); here the source of output j is output j - d's OWN pixel, d = (j mod W) mod h < 4, which the lane d
// places to the left has just loaded: one ds_bpermute instead of a load.  Only the first d lanes of a wave (their source sits in
// the previous wave) and the odd chroma rows of 4:x:0 (every pixel replays the last sample of the chroma row above) load a
// second pixel, through a wave-uniform branch that most waves skip.  Run-time parameters like k_generic: one kernel per
// (rounding, format).
// ------------------------------------------------------------------------------------------------
template <int ROUND, int FMT, int K, bool NT, bool CHECK>
__device__ __forceinline__ void flatgen_body(const KArgs &a, gin_t in, gout_t out, uint32_t i0, uint32_t T, uint32_t n)
{
    const int lane = (int)(threadIdx.x & 63u);
    uint32_t px[K], hp[K], yo[K], oo[K], ho[K];
    int dd[K];
    bool need[K];
    // 32-bit offsets; (ro, co) of the lane's 2nd..Kth index by stepping, as in decflat_body
    const uint32_t Wo = (uint32_t)a.Wo, f = (uint32_t)a.f, rstep = f * (uint32_t)a.ip;
    const uint32_t qT = (uint32_t)(((uint64_t)T * a.mWo) >> a.kWo), rT = T - qT * Wo;
    uint32_t ro = 0, co = 0;
#pragma unroll
    for (int k = 0; k < K; ++k) {
        const uint32_t j = CHECK ? min(i0 + (uint32_t)k * T, n - 1) : i0 + (uint32_t)k * T;
        if (CHECK || k == 0) {
            ro = (uint32_t)(((uint64_t)j * a.mWo) >> a.kWo);                      // j / Wo, exact for j < 2^31
            co = j - __umul24(ro, Wo);
        } else {
            const bool wrap = co + rT >= Wo;
            co = wrap ? co + rT - Wo : co + rT;
            ro += qT + (wrap ? 1u : 0u);
        }
        yo[k] = __umul24(ro, rstep) + co * f;                                    // rows, pitches and f * pitch fit 24 bits (prepare_common)
        px[k] = in1n<NT>(a, in, yo[k]);
        oo[k] = __umul24(ro, (uint32_t)a.op) + co;
        int r, d;
        if (a.s_first) {                                                          // counters over the decimated stream, width W
            r = (int)(((uint64_t)j * a.mW) >> a.kW);
            d = ((int)j - r * a.W) & a.hmask;
        } else {                                                                  // counters == image coordinates (rows ro * f are sample rows)
            r = 0;
            d = (int)((co * f) & (uint32_t)a.hmask) >> a.sc_shift;
        }
        const bool odd = (r & a.vmask) != 0;
        const int src = odd ? (r - 1) * a.W + a.last_sample_col : (int)j - d;    // flat index whose OWN pixel is the chroma source
        const uint32_t sro = (uint32_t)(((uint64_t)(uint32_t)src * a.mWo) >> a.kWo), sco = (uint32_t)src - sro * Wo;
        ho[k] = __umul24(sro, rstep) + sco * f;
        dd[k] = odd ? 0 : d;
        need[k] = odd || d > lane;                                                // not in a lane of this wave
    }
#pragma unroll
    for (int k = 0; k < K; ++k) {
        hp[k] = 0;
        if (__builtin_amdgcn_ballot_w64(need[k]) != 0) hp[k] = in1n<false>(a, in, need[k] ? ho[k] : yo[k]);
    }
#pragma unroll
    for (int k = 0; k < K; ++k) {
        const uint32_t nb = (uint32_t)__shfl((int)px[k], lane - dd[k], 64);       // all lanes take part
        if (!CHECK || i0 + (uint32_t)k * T < n) {
            const ChromaTerm t = chroma_term<ROUND, FMT>(need[k] ? hp[k] : nb, a.mcb, a.mcr);
            out1n<NT>(a, out, oo[k], finish<FMT>(px[k], a.my, t));
        }
    }
    if (!CHECK) keep_tail_apart();
}

template <int ROUND, int FMT, int K, bool NT>
__global__ void __launch_bounds__(256) k_flatgen(KArgs a)
{
    pin_args(a);
    const uint32_t T = (uint32_t)a.bdx;
    const uint32_t n = (uint32_t)a.Wo * (uint32_t)a.Ho;
    const uint32_t b0 = blockIdx.x * (T * K);
    const gin_t in = frame_in(a);
    const gout_t out = frame_out(a);
    if (b0 + T * K <= n) flatgen_body<ROUND, FMT, K, NT, false>(a, in, out, b0 + threadIdx.x, T, n);
    else                 flatgen_body<ROUND, FMT, K, NT, true>(a, in, out, b0 + threadIdx.x, T, n);
}

// ------------------------------------------------------------------------------------------------
// utilities: synthetic frames and checksum
// ------------------------------------------------------------------------------------------------
__host__ __device__ __forceinline__ uint32_t fmix32(uint32_t h)
{
    h ^= h >> 16; h *= 0x85ebca6bu; h ^= h >> 13; h *= 0xc2b2ae35u; h ^= h >> 16;
    return h;
}

__global__ void __launch_bounds__(256) k_synth(uint32_t *dst, int64_t npix, int64_t first_index, uint32_t salt)
{
    const int64_t stride = (int64_t)gridDim.x * blockDim.x;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < npix; i += stride)
        dst[i] = 0xFF000000u | (fmix32((uint32_t)(first_index + i) + salt) & 0x00FFFFFFu);
}

// plain 16 B/lane non-temporal copy: the streaming ceiling bench.py quotes next to the roofline
__global__ void __launch_bounds__(256) k_copy(const uint32_t *src, uint32_t *dst, int64_t n4)
{
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i < n4) st4<true>(dst + 4 * i, ld4<true>(src + 4 * i));
}

__global__ void __launch_bounds__(256) k_checksum(const uint32_t *src, int64_t npix, unsigned long long *sum)
{
    const int64_t stride = (int64_t)gridDim.x * blockDim.x;
    unsigned long long acc = 0;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < npix; i += stride)
        acc += fmix32(src[i] + 0x9E3779B9u * (uint32_t)i);
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) acc += __shfl_down(acc, off, 64);
    if ((threadIdx.x & 63) == 0) atomicAdd(sum, acc);
}

// one checked read of the frame described by `a` (csic_debug_probe_device: is the range check of this build live?)
__global__ void __launch_bounds__(64) k_debug_probe(KArgs a, int64_t off, uint32_t *sink)
{
    if (threadIdx.x == 0 && blockIdx.x == 0) *sink = in1<false>(a, (gin_t)(uintptr_t)a.in, off);
}

// ------------------------------------------------------------------------------------------------
// plan
// ------------------------------------------------------------------------------------------------
enum Family { FAM_F1X4, FAM_DEC, FAM_DEC2V1, FAM_DEC2V2, FAM_GENERIC, FAM_AVG, FAM_AVG_GENERIC, FAM_DECFLAT, FAM_F1FLAT };

} // namespace csic

struct csic_plan {
    csic_params p;
    csic::Geometry g;
    int device;
    int variant;
    int force_generic;
    int no_vec;          // 1 = no 16-byte vector kernels (set per launch for pointers that are only 4-byte aligned)
    int no_flat;         // 1 = no flat kernels (set per launch for frames whose extents pass 2^30 pixels: they address with 32-bit byte offsets)
    int dec_hold;        // k_dec: lane-hold distance of the selected kernel (1, 2 or 4)
    int no_nt;           // 1 = plain (cached) loads/stores instead of non-temporal ones
    int block_threads;   // 0 = default (256); 64 / 128 = smaller blocks (CSIC_TUNE_BLOCK_THREADS)
    // selection (recomputed by select())
    csic::Family fam;
    csic::KernelFn fn;
    int units_per_row;   // lanes needed along x
    int k_per_lane;      // x units consumed per lane (k_dec only)
    char name[96];
    // host path staging
    void *d_in, *d_out;
    unsigned long long *d_sum;
};

namespace csic {

template <int ROUND, int FMT, bool NT>
static KernelFn pick_f1x4(int h, int v)
{
    if (v == 1) {
        if (h == 1) return k_f1x4<ROUND, FMT, 1, 1, NT>;
        if (h == 2) return k_f1x4<ROUND, FMT, 2, 1, NT>;
        return k_f1x4<ROUND, FMT, 4, 1, NT>;
    }
    if (h == 1) return k_f1x4<ROUND, FMT, 1, 2, NT>;
    if (h == 2) return k_f1x4<ROUND, FMT, 2, 2, NT>;
    return k_f1x4<ROUND, FMT, 4, 2, NT>;
}

template <int ROUND, int FMT, bool NT>
static KernelFn pick_f1flat(int h, int v)
{
    if (v == 1) {
        if (h == 1) return k_f1flat<ROUND, FMT, 1, 1, NT>;
        if (h == 2) return k_f1flat<ROUND, FMT, 2, 1, NT>;
        return k_f1flat<ROUND, FMT, 4, 1, NT>;
    }
    if (h == 1) return k_f1flat<ROUND, FMT, 1, 2, NT>;
    if (h == 2) return k_f1flat<ROUND, FMT, 2, 2, NT>;
    return k_f1flat<ROUND, FMT, 4, 2, NT>;
}

constexpr int DEC_K = 4;

template <int ROUND, int FMT, int F, bool NT>
static KernelFn pick_dec_f(int hold, bool srows)
{
    if (srows) {
        if (hold == 1) return k_dec<ROUND, FMT, F, 1, true, DEC_K, NT>;
        if (hold == 2) return k_dec<ROUND, FMT, F, 2, true, DEC_K, NT>;
        return k_dec<ROUND, FMT, F, 4, true, DEC_K, NT>;
    }
    if (hold == 1) return k_dec<ROUND, FMT, F, 1, false, DEC_K, NT>;
    if (hold == 2) return k_dec<ROUND, FMT, F, 2, false, DEC_K, NT>;
    return k_dec<ROUND, FMT, F, 4, false, DEC_K, NT>;
}

template <int ROUND, int FMT, bool NT>
static KernelFn pick_dec(int f, int hold, bool srows)
{
    if (f == 1) return pick_dec_f<ROUND, FMT, 1, NT>(hold, srows);
    if (f == 2) return pick_dec_f<ROUND, FMT, 2, NT>(hold, srows);
    if (f == 4) return pick_dec_f<ROUND, FMT, 4, NT>(hold, srows);
    return pick_dec_f<ROUND, FMT, 8, NT>(hold, srows);
}

template <int ROUND, int FMT, int F, bool NT>
static KernelFn pick_decflat_f(int hold, bool srows)
{
    if (srows) {
        if (hold == 1) return k_decflat<ROUND, FMT, F, 1, true, DEC_K, NT>;
        if (hold == 2) return k_decflat<ROUND, FMT, F, 2, true, DEC_K, NT>;
        return k_decflat<ROUND, FMT, F, 4, true, DEC_K, NT>;
    }
    if (hold == 1) return k_decflat<ROUND, FMT, F, 1, false, DEC_K, NT>;
    if (hold == 2) return k_decflat<ROUND, FMT, F, 2, false, DEC_K, NT>;
    return k_decflat<ROUND, FMT, F, 4, false, DEC_K, NT>;
}

template <int ROUND, int FMT, bool NT>
static KernelFn pick_decflat(int f, int hold, bool srows)
{
    if (f == 2) return pick_decflat_f<ROUND, FMT, 2, NT>(hold, srows);
    if (f == 4) return pick_decflat_f<ROUND, FMT, 4, NT>(hold, srows);
    return pick_decflat_f<ROUND, FMT, 8, NT>(hold, srows);
}

// Block width k_dec would take for `lanes_x` lanes per row in blocks of `tpb` threads (prepare_common): a width that divides
// the row exactly when there is one between tpb / 2 and tpb lanes, else the power of two that leaves a partial chunk.
static int dec_block_x(int lanes_x, int tpb, int hold)
{
    int bx = 1;
    while (bx < lanes_x) bx <<= 1;
    if (bx > tpb) bx = tpb;
    if (lanes_x <= tpb) {
        if (lanes_x % hold == 0) bx = lanes_x;
    } else {
        for (int m = (lanes_x + tpb - 1) / tpb; m <= lanes_x / (tpb / 2); ++m)
            if (lanes_x % m == 0 && (lanes_x / m) % hold == 0) { bx = lanes_x / m; break; }
    }
    return bx;
}

// One-wave blocks for narrow rows (a row needs at most two waves and tiles into them): see prepare_common.
static bool dec_one_wave_blocks(int lanes_x, int f, int hold)
{
    if (lanes_x < 16 || lanes_x > 128) return false;
    bool tiles = (f >= 4 && lanes_x < 64) || (lanes_x & (lanes_x - 1)) == 0;       // 16, 32, 64, 128; any < 64 for f >= 4
    for (int w = 64; !tiles && w >= 48; --w) tiles = lanes_x % w == 0 && w % hold == 0;
    return tiles && (lanes_x >= 64 || lanes_x % hold == 0);
}

// Should a chroma-before-spatial, hold-free plan cover the flat decimated stream (k_decflat) instead of rows (k_dec)?
// Measured over 22 shapes (tools/probe_flat.py, profiles/r03_probe_flat.log; batched launches, k_dec | flat):
//  * rows k_dec cannot cut into whole blocks -- not a whole number of K-pixel lanes, or lanes without a usable divisor: every
//    block of k_dec runs its bounds-checked path (1000x1000 f = 4 / 8: 66 / 63 | 73 / 71 %; 1366x768 f = 2: 68 | 79 %);
//  * rows of a few partly filled waves that straddle two rows at odd offsets (1000x1000 f = 2: 70 | 78 %, 720x480 f = 2:
//    70 | 78 %, 352x288 f = 2: 70 | 81 %, 1280x720 f = 2: 78.5 | 81.5 %);
//  * rows of whole waves: level or slightly ahead (8192x8192 f = 2 / 4 / 8: 79.8 | 82.6, 76.3 | 76.9, 74.3 | 77.0 %; 3840x2160
//    f = 4: 75.3 | 76.7 %; 1080p / 4K f = 2: 79.6 | 80.4 %), and on the headline -- ONE 8192x8192 frame per launch -- four
//    interleaved repeats give 32.09 | 31.85 us = 78.4 | 79.0 % (profiles/r03_headline_flat_ab.jsonl);
//  * the one exception: shapes that take k_dec's one-wave blocks (512x512 f = 2: 77 | 65-73 %; 1024x1024 f = 8: 79 | 70-78 %;
//    640x480 f = 4 and 1920x1080 f = 4 level) stay with k_dec;
//  * the same picture with a lane hold and with spatial before chroma (profiles/r03_probe_flat_{csq411,scq444,scq422,scq420}.log;
//    4:2:0 spatial before chroma: 1000x1000 f = 2: 60 | 78 %, 352x288 f = 2: 61 | 80 %, 8192x8192 f = 2 / 4 / 8: 79.4 | 82.6,
//    77.0 | 77.8, 75.2 | 77.5 % -- once the held-pixel load of the odd chroma rows is skipped wave-uniformly; loaded
//    unconditionally it cost 8192x8192 f = 8 twenty points).
static bool dec_prefers_flat(const Geometry &g, int hold)
{
    if (g.Wo % DEC_K != 0) return true;
    return !dec_one_wave_blocks(g.Wo / DEC_K, g.f, hold);
}

template <int ROUND, int FMT, int F, bool NT>
static KernelFn pick_avg_f(int h, int v)
{
    if (v == 1) {
        if (h == 1) return k_avg<ROUND, FMT, F, 1, 1, NT>;
        if (h == 2) return k_avg<ROUND, FMT, F, 2, 1, NT>;
        return k_avg<ROUND, FMT, F, 4, 1, NT>;
    }
    if (h == 1) return k_avg<ROUND, FMT, F, 1, 2, NT>;
    if (h == 2) return k_avg<ROUND, FMT, F, 2, 2, NT>;
    return k_avg<ROUND, FMT, F, 4, 2, NT>;
}

template <int ROUND, int FMT, bool NT>
static KernelFn pick_avg(int f, int h, int v)
{
    if (f == 1) return pick_avg_f<ROUND, FMT, 1, NT>(h, v);
    if (f == 2) return pick_avg_f<ROUND, FMT, 2, NT>(h, v);
    if (f == 4) return pick_avg_f<ROUND, FMT, 4, NT>(h, v);
    return pick_avg_f<ROUND, FMT, 8, NT>(h, v);
}

// Can the k_dec family handle this geometry?  Chroma before spatial: always, except that a hold across
// lanes (4:1:1 with f = 2) needs whole quads in a row.  Spatial before chroma: only when chroma rows
// coincide with groups of decimated rows (f | W) and the in-row hold is lane-aligned (h | Wo).
static bool dec_fast_ok(const Geometry &g)
{
    const int hold = (g.s_first || g.f == 1) ? g.h : (g.h > g.f ? g.h / g.f : 1);
    const int lanes_x = (g.Wo + DEC_K - 1) / DEC_K;
    if (hold > 1 && lanes_x < 3) return false;            // block width < 4 lanes: quads would span rows
    if (!g.s_first) return true;
    return (g.W % g.f == 0) && (g.Wo % g.h == 0);
}

template <int ROUND, int FMT>
static void select_rf(csic_plan *pl)
{
    const Geometry &g = pl->g;
    const char *rn = ROUND == R_FLOOR ? "floor" : "trunc";
    const char *fn = FMT == F_ARGB ? "argb" : "ycc";
    const bool nt = !pl->no_nt;
    const char *ntn = nt ? "nt" : "cached";
    // A YCbCr input stream (single-stage driving, the reference's spec style) is a test-oriented path:
    // it is served by the run-time-parameter kernels only.
    const bool ycc_in = pl->p.in_format == CSIC_FMT_YCBCR888X;
    if (pl->p.sampling == CSIC_SAMPLING_AVG) {
        const int th = g.f > g.v ? g.f : g.v;
        const int tw = g.f == 8 ? 8 : 4;
        // any shape with at least one whole tile (pair of tiles at f = 8) and tile row: the frame's cut tiles take the definition's
        // clamped form inside k_avg; variant 8 keeps the rule of rounds 1-3 (whole tiles only, everything else generic) for A/B
        const bool whole = g.W % tw == 0 && g.H % th == 0;
        if (!pl->force_generic && !ycc_in && !pl->no_vec && g.W >= tw && g.H >= th && (whole || pl->variant != 8)) {
            pl->fam = FAM_AVG;
            pl->fn = nt ? pick_avg<ROUND, FMT, true>(g.f, g.h, g.v) : pick_avg<ROUND, FMT, false>(g.f, g.h, g.v);
            pl->units_per_row = (g.W + 3) / 4;
            pl->k_per_lane = (g.f <= 2) ? 2 : 1;          // TILES of k_avg
            snprintf(pl->name, sizeof pl->name, "k_avg<%s,%s,f%d,h%d,v%d,%s>", rn, fn, g.f, g.h, g.v, ntn);
        } else {
            pl->fam = FAM_AVG_GENERIC;
            pl->fn = ycc_in ? (KernelFn)k_avg_generic<ROUND, FMT, F_YCC> : (KernelFn)k_avg_generic<ROUND, FMT, F_ARGB>;
            pl->units_per_row = g.Wo;
            pl->k_per_lane = 1;
            snprintf(pl->name, sizeof pl->name, "k_avg_generic<%s,%s%s>", rn, fn, ycc_in ? ",ycc-in" : "");
        }
        return;
    }
    // f = 1: the 16-byte kernel wins whenever it applies (8192^2: 4:4:4 84.0 vs 88.0 us, 4:2:0 84.5 vs 85.9 us for
    // the 4-byte k_dec<f1>, which serves the other widths / alignments; variant 4 forces k_dec<f1> for A/B).
    const bool f1x4_ok = !pl->force_generic && !ycc_in && !pl->no_vec && g.f == 1 && g.W % 4 == 0;
    if (f1x4_ok && pl->variant != 11 && pl->variant != 4 && !pl->no_flat) {
        // the flat mapping (round 4): ahead of k_f1x4 at every chroma mode and on 10 of 12 frame sizes -- 8192x8192 4:2:0 77.0 ->
        // 79.3 %, 4:2:2 77.5 -> 80.3 %, 4:4:4 78.8 -> 79.8 %, 4:1:0 76.5 -> 79.5 %, 4096x4096 76.5 -> 79.1 %, 1000x1000 76.6 -> 78.9 %;
        // level (-0.5) on 3840x2160 and 1920x1080 (profiles/r04_f1flat_ab.log).  CSIC_TUNE_VARIANT 11 keeps k_f1x4 for A/B.
        pl->fam = FAM_F1FLAT;
        pl->fn = nt ? pick_f1flat<ROUND, FMT, true>(g.h, g.v) : pick_f1flat<ROUND, FMT, false>(g.h, g.v);
        pl->units_per_row = g.W / 4;
        pl->k_per_lane = 1;
        snprintf(pl->name, sizeof pl->name, "k_f1flat<%s,%s,h%d,v%d,%s>", rn, fn, g.h, g.v, ntn);
    } else if (f1x4_ok && (pl->variant != 4 || !dec_fast_ok(g))) {
        pl->fam = FAM_F1X4;
        pl->fn = nt ? pick_f1x4<ROUND, FMT, true>(g.h, g.v) : pick_f1x4<ROUND, FMT, false>(g.h, g.v);
        pl->units_per_row = g.W / 4;
        pl->k_per_lane = 1;
        snprintf(pl->name, sizeof pl->name, "k_f1x4<%s,%s,h%d,v%d,%s>", rn, fn, g.h, g.v, ntn);
    } else if (!pl->force_generic && !ycc_in && dec_fast_ok(g)) {
        // in-row chroma hold distance in decimated lanes; srows = chroma rows follow the decimated stream
        // (with f = 1 the decimated stream IS the image and both order classes coincide: any width, any
        // 4-byte-aligned pointer, 4-byte accesses)
        const bool srows = g.s_first != 0 || g.f == 1;
        const int hold = srows ? g.h : (g.h > g.f ? g.h / g.f : 1);
        if (g.f == 2 && hold == 1 && !srows && !pl->no_vec && g.W % 8 == 0 && (pl->variant == 1 || pl->variant == 2)) {
            pl->fam = pl->variant == 1 ? FAM_DEC2V1 : FAM_DEC2V2;
            if (pl->variant == 1) pl->fn = nt ? (KernelFn)k_dec2v<ROUND, FMT, 1, true> : (KernelFn)k_dec2v<ROUND, FMT, 1, false>;
            else                  pl->fn = nt ? (KernelFn)k_dec2v<ROUND, FMT, 2, true> : (KernelFn)k_dec2v<ROUND, FMT, 2, false>;
            pl->units_per_row = g.Wo / (pl->variant == 1 ? 2 : 4);
            pl->k_per_lane = 1;
            snprintf(pl->name, sizeof pl->name, "k_dec2v<%s,%s,var%d,%s>", rn, fn, pl->variant, ntn);
        } else if (g.f >= 2 && g.Wo % hold == 0 && pl->variant != 5 && !pl->no_flat && (pl->variant == 6 || dec_prefers_flat(g, hold))) {
            // lanes over the flat decimated stream (variant 5 keeps k_dec, variant 6 takes k_decflat wherever it applies: A/B);
            // a hold group must not straddle two rows: hold | Wo (spatial before chroma has that from dec_fast_ok)
            const bool sr = srows && g.v == 2;
            pl->fam = FAM_DECFLAT;
            pl->dec_hold = hold;
            pl->fn = nt ? pick_decflat<ROUND, FMT, true>(g.f, hold, sr) : pick_decflat<ROUND, FMT, false>(g.f, hold, sr);
            pl->units_per_row = g.Wo;
            pl->k_per_lane = DEC_K;
            if (hold == 1 && !srows)
                snprintf(pl->name, sizeof pl->name, "k_decflat<%s,%s,f%d,K%d,%s>", rn, fn, g.f, DEC_K, ntn);
            else
                snprintf(pl->name, sizeof pl->name, "k_decflat<%s,%s,f%d,hold%d,%s,K%d,%s>", rn, fn, g.f, hold,
                         srows ? (g.v == 2 ? "s>c,v2" : "s>c") : "c>s", DEC_K, ntn);
        } else {
            pl->fam = FAM_DEC;
            pl->dec_hold = hold;
            pl->fn = nt ? pick_dec<ROUND, FMT, true>(g.f, hold, srows && g.v == 2)
                        : pick_dec<ROUND, FMT, false>(g.f, hold, srows && g.v == 2);
            pl->units_per_row = g.Wo;
            pl->k_per_lane = DEC_K;
            snprintf(pl->name, sizeof pl->name, "k_dec<%s,%s,f%d,hold%d,%s,K%d,%s>", rn, fn, g.f, hold,
                     g.f == 1 ? (g.v == 2 ? "v2" : "v1") : srows ? (g.v == 2 ? "s>c,v2" : "s>c") : "c>s", DEC_K, ntn);
        }
    } else if (!pl->force_generic && !ycc_in && g.f >= 2 && pl->variant != 7 && !pl->no_flat) {
        // what k_dec / k_decflat cannot take (spatial before chroma with f not dividing W or h not dividing Wo; tiny frames with a
        // hold): the general flat kernel; variant 7 keeps the one-pixel-per-lane k_generic for A/B
        pl->fam = FAM_DECFLAT;
        pl->dec_hold = 1;
        pl->fn = nt ? (KernelFn)k_flatgen<ROUND, FMT, DEC_K, true> : (KernelFn)k_flatgen<ROUND, FMT, DEC_K, false>;
        pl->units_per_row = g.Wo;
        pl->k_per_lane = DEC_K;
        snprintf(pl->name, sizeof pl->name, "k_flatgen<%s,%s,K%d,%s>", rn, fn, DEC_K, ntn);
    } else {
        pl->fam = FAM_GENERIC;
        pl->fn = ycc_in ? (KernelFn)k_generic<ROUND, FMT, F_YCC> : (KernelFn)k_generic<ROUND, FMT, F_ARGB>;
        pl->units_per_row = g.Wo;
        pl->k_per_lane = 1;
        snprintf(pl->name, sizeof pl->name, "k_generic<%s,%s%s>", rn, fn, ycc_in ? ",ycc-in" : "");
    }
}

static void select(csic_plan *pl)
{
    if (pl->p.out_format == CSIC_FMT_PLANAR) {           // csic_planar.hip picks and names its kernels
        pl->fam = FAM_GENERIC; pl->fn = nullptr; pl->units_per_row = pl->g.Wo; pl->k_per_lane = 1;
        planar_kernel_name(pl, pl->name, sizeof pl->name);
        return;
    }
    const int r = pl->p.rounding, f = pl->p.out_format;
    if (r == R_FLOOR && f == F_ARGB) select_rf<R_FLOOR, F_ARGB>(pl);
    else if (r == R_FLOOR) select_rf<R_FLOOR, F_YCC>(pl);
    else if (f == F_ARGB) select_rf<R_TRUNC, F_ARGB>(pl);
    else select_rf<R_TRUNC, F_YCC>(pl);
}

static int pow2_ceil(int x) { int p = 1; while (p < x) p <<= 1; return p; }

// The geometry half of the kernel arguments (what does not depend on the kernel family or the launch shape).
void fill_base_args(const Geometry &g, int32_t ip, int32_t op, KArgs *pa)
{
    KArgs &a = *pa;
    std::memset(&a, 0, sizeof a);
    a.W = g.W; a.H = g.H; a.Wo = g.Wo; a.Ho = g.Ho;
    a.last_sample_col = g.last_sample_col;
    a.my = g.mask_y; a.mcb = g.mask_cb; a.mcr = g.mask_cr;
    a.f = g.f; a.hmask = g.h - 1; a.vmask = g.v - 1; a.s_first = g.s_first;
    a.ip = ip; a.op = op;
    a.in_frame_px = (int64_t)ip * g.H;
    a.out_frame_px = (int64_t)op * g.Ho;
    a.sc_shift = (g.f == 8) ? 3 : (g.f == 4) ? 2 : (g.f == 2) ? 1 : 0;
    a.bc_row_off = g.last_sample_col / g.Wo;             // only meaningful (and only used) when f | W
    a.bc_col_in = (g.last_sample_col % g.Wo) * g.f;
    magic_div((uint32_t)g.W, &a.mW, &a.kW);
    magic_div((uint32_t)g.Wo, &a.mWo, &a.kWo);
}

// Resolves kernel, grid and arguments for `nframes` frames (<= 65535, the grid z limit) whose base pointers OR to
// `align_bits`; the callers below fill in where the frames are.
static int prepare_common(const csic_plan *pl, uintptr_t align_bits, int nframes, int32_t in_pitch, int32_t out_pitch, LaunchDesc *d)
{
    if (nframes <= 0 || nframes > 65535)
        return set_error(CSIC_EINVAL_SIZE, "nframes per launch must be in 1..65535. Got %d", nframes);
    if (pl->p.out_format == CSIC_FMT_PLANAR)
        return set_error(CSIC_EINVAL_FORMAT, "planar plans go through csic_process_device / csic_process_batch_device / csic_process_host, "
                                              "csic_pipeline_* and fused frame graphs only (no row pitches, per-frame-launch graphs, file pools or csic_multi)");
    const Geometry &g = pl->g;

    Family fam = pl->fam;
    KernelFn fn = pl->fn;
    int units = pl->units_per_row, kpl = pl->k_per_lane, dec_hold = pl->dec_hold;
    // The vector kernels need 16-byte aligned frame bases; otherwise take the 4-byte-access kernels.
    const int32_t ip = in_pitch > 0 ? in_pitch : g.W, op = out_pitch > 0 ? out_pitch : g.Wo;
    if (ip < g.W || op < g.Wo)
        return set_error(CSIC_EINVAL_SIZE, "row pitch (%d, %d px) smaller than the frame width (%d, %d px)", ip, op, g.W, g.Wo);
    // (k_avg takes any 4-byte alignment: gfx950 executes its 16-byte accesses at any dword address, tools/ubench_unaligned.hip;
    // the others keep the rule because their 4-byte fallbacks are as fast as a misaligned vector access would be)
    const bool vec = (fam == FAM_F1X4 || fam == FAM_DEC2V1 || fam == FAM_DEC2V2 || fam == FAM_F1FLAT);
    // The flat kernels address a pixel by a 32-bit BYTE offset from its frame's base (decflat_body): frames whose extents -- pitch
    // included -- pass 2^30 pixels (4 GiB) take the row kernels, which keep 64-bit offsets.
    // (and whose rows and pitch -- times the factor -- fit 24 bits, for the full-rate 24-bit multiplies of the row offsets)
    const int64_t flat_limit = 1ll << 30;
    const bool too_wide = (fam == FAM_DECFLAT || fam == FAM_F1FLAT) &&
                          ((int64_t)(g.H - 1) * ip + g.W > flat_limit || (int64_t)(g.Ho - 1) * op + g.Wo > flat_limit ||
                           g.H >= (1 << 24) || (int64_t)ip * g.f >= (1 << 24) || op >= (1 << 24));
    const bool misaligned = vec && ((align_bits & 15u) || ((ip | op) & 3));
    if (too_wide || misaligned) {
        csic_plan tmp = *pl;
        if (too_wide) tmp.no_flat = 1;
        if (misaligned) tmp.no_vec = 1;
        select(&tmp);
        fam = tmp.fam; fn = tmp.fn; units = tmp.units_per_row; kpl = tmp.k_per_lane; dec_hold = tmp.dec_hold;
    }

    KArgs &a = d->args;
    fill_base_args(g, ip, op, &a);

    // Threads per block.  256 by default; k_dec takes two-wave blocks (128 threads) for a single frame of >= 64 MB whose rows
    // tile into full waves at that width: measured on one-frame-per-launch streams (profiles/r02_probe_block_shapes.log)
    // 8192x8192 f=2 32.48 -> 31.89 us, 16384x4096 32.56 -> 31.98, 8192x4096 17.44 -> 17.27, 6144x6144 19.38 -> 19.18,
    // 8192x8192 f=4 15.52 -> 15.37; no gain below ~64 MB (8192x2048: 9.88 / 9.88), none for batched launches, and a loss
    // where 128 lanes do not divide the row into full waves (7680x4320: 17.29 -> 17.93).  CSIC_TUNE_BLOCK_THREADS overrides.
    const int avg_th = g.f > g.v ? g.f : g.v;
    const int rows = (fam == FAM_F1X4) ? g.H : (fam == FAM_AVG) ? (g.H + avg_th - 1) / avg_th : g.Ho;
    const int lanes_x = (units + kpl - 1) / kpl;
    int tpb = 256;
    const bool forced = pl->block_threads == 64 || pl->block_threads == 128 || pl->block_threads == 256;
    const int hold = (fam == FAM_DEC && dec_hold > 0) ? dec_hold : 1;
    if (forced) tpb = pl->block_threads;
    else if (fam == FAM_DEC && nframes == 1 && units % kpl == 0 && lanes_x % 128 == 0 &&
             4ll * ((int64_t)g.W * g.Ho + (int64_t)g.Wo * g.Ho) >= (64ll << 20))
        tpb = 128;
    else if (fam == FAM_DEC && units % kpl == 0 && lanes_x >= 16 && lanes_x <= 128) {
        // Narrow rows (a row needs at most two waves): one-wave blocks, when the row tiles into them, beat blocks that stack
        // several rows -- batched launches, profiles/r02_probe_block_batched.log: 512x512 f=2 69.1 -> 73.5 %, f=8 69.3 -> 73.2,
        // 1024x1024 f=8 66.8 -> 78.3, 1920x1080 f=4 74.1 -> 75.6, f=8 71.0 -> 73.5; rows that do not tile (1000x1000 f=2: 125
        // lanes) lose (70.8 -> 63.4) and keep the default, as do rows of fewer than 16 lanes (128x128 f=4/8: -1 %).
        // With f >= 4, rows of fewer than 64 lanes fit one wave whatever their width (640x480 f=4, 40 lanes: 74.1 -> 77.7 %;
        // 352x288 f=4 s>c: 62.0 -> 72.4 %); at f = 2 that loses (352x288, 44 lanes: 70.5 -> 59.3 %) and only powers of two qualify.
        if (dec_one_wave_blocks(lanes_x, g.f, hold)) tpb = 64;
    }
    int bx = pow2_ceil(lanes_x);
    if (bx > tpb) bx = tpb;
    if (bx < 1) bx = 1;
    if (fam == FAM_DEC && units % kpl == 0) {
        // Rows that do not tile into power-of-two chunks (1920/3840-wide video: Wo = 960, 1920, ...) would put
        // their last chunk on the bounds-checked path.  A block width that divides the row exactly keeps
        // every block on the straight-line path (4K f=2: 70 % -> 80 % of HBM peak).  The width only has to
        // be a multiple of the lane-hold distance so that a DPP hold group never straddles two rows.
        bx = dec_block_x(lanes_x, tpb, hold);
    }
    if (fam == FAM_F1FLAT) {
        const int T = forced ? tpb : 64;
        const int64_t ngroups = (int64_t)(g.W / 4) * g.H, per_block = (int64_t)T * 4;
        d->block = dim3((unsigned)T, 1, 1);
        a.bdx = T; a.bdy = 1; a.row_step = 1;
        d->grid = dim3((unsigned)((ngroups + per_block - 1) / per_block), 1, (unsigned)nframes);
        d->fn = fn;
        return CSIC_OK;
    }
    if (fam == FAM_DECFLAT) {
        // lanes over the flat decimated stream: blocks of whole waves, K indices per lane spaced by the block size
        // Two-wave blocks at f = 2 (1000x1000 77.8 -> 78.6 %, 1366x768 77.1 -> 79.0, 352x288 80.5 -> 81.8, 8192x8192 80.0 -> 82.6)
        // and for long rows at f = 4 / 8 (3840x2160 f = 4: 74.7 -> 76.7 %, 8192x8192 f = 8: 75.9 -> 77.0); four-wave blocks for
        // short rows at f = 4 / 8 (1000x1000 f = 4: 72.3 % against 69.4 / 68.4 % with 128 / 64 threads; 1920x1080 f = 8: 75.7
        // against 75.2 / 71.1).                                                      profiles/r03_probe_flat.log
        const int T = forced ? tpb : ((g.f == 2 || g.Wo >= 512) ? 128 : 256);
        const int64_t per_block = (int64_t)T * kpl, n = (int64_t)g.Wo * g.Ho;
        d->block = dim3((unsigned)T, 1, 1);
        a.bdx = T; a.bdy = 1; a.row_step = 1;
        d->grid = dim3((unsigned)((n + per_block - 1) / per_block), 1, (unsigned)nframes);
        d->fn = fn;
        return CSIC_OK;
    }
    if (fam == FAM_F1X4 && !forced) {
        // One 16-byte load and store per lane: this kernel lives on the wave launch rate, so waves that exit at once (the idle
        // part of a block's last chunk) or run partly filled cost in proportion.  1280-wide rows are 320 lanes: [256][64 + 192
        // idle] runs at 61 %, 5 x 64 lanes (four rows to a block) at 78 % (profiles/r02_probe_block_batched_video.log).
        if (lanes_x % 64 == 0) {
            for (int w : {256, 192, 128, 64}) if (lanes_x % w == 0) { bx = w; break; }
        } else if (lanes_x <= 256) {
            bx = lanes_x;                                      // one partly filled wave per row instead of idle ones
        }
    }
    if (fam == FAM_AVG && !forced && lanes_x % 64 == 0 && lanes_x > 256 && lanes_x <= 512 && lanes_x % 256 != 0) {
        // the same for k_avg on rows of at most two blocks: 1280-wide f = 4 / 8 (320 lanes) 64 / 61 % -> 80 % with blocks of whole
        // waves that tile the row (profiles/r02_probe_block_avg.log)
        for (int w : {192, 128, 64}) if (lanes_x % w == 0) { bx = w; break; }
    } else if (fam == FAM_AVG && !forced && lanes_x > tpb && lanes_x % tpb != 0 && lanes_x <= 8 * tpb) {
        // rows of a few blocks that do not tile (1368-wide f = 4: 342 lanes = [256][86 + 170 idle]): equal blocks instead of a
        // nearly empty last one
        // (a multiple of 4 lanes: at f = 8 the two tiles of an output are neighbouring lanes of one quad -- the DPP swap -- so a
        // block must not start on an odd tile; tools/fuzz_gpu.py found 1032x8 with 129-lane blocks)
        const int m = (lanes_x + tpb - 1) / tpb;
        bx = ((lanes_x + m - 1) / m + 3) & ~3;
        if (bx > tpb) bx = tpb;
    }
    const int by = tpb / bx > 0 ? tpb / bx : 1;
    d->block = dim3(bx, by, 1);
    unsigned gx = (unsigned)((lanes_x + bx - 1) / bx);
    unsigned gy_edge = 0;
    if (fam == FAM_AVG) {
        // k_avg's edge blocks: one lane per output pixel that no whole tile produces (see k_avg), in rows of blocks below the grid
        const int W4f = g.W / 4, ntrf = g.H / avg_th;
        const int Cw = g.f == 8 ? W4f / 2 : W4f * (4 / g.f), Rw = g.f == 8 ? ntrf : ntrf * (avg_th / g.f);
        const int64_t nedge = (int64_t)(g.Wo - Cw) * g.Ho + (int64_t)(g.Ho - Rw) * Cw;
        const int64_t nblocks = (nedge + (int64_t)bx * by - 1) / ((int64_t)bx * by);
        gy_edge = (unsigned)((nblocks + gx - 1) / gx);
    }
    unsigned gy = (unsigned)((rows + by - 1) / by);
    if (gy > 65535u - gy_edge - 1u) gy = 65535u - gy_edge - 1u;  // kernels stride over rows
    if (gy_edge > 0 && nframes > 1 && (gx * (gy + gy_edge)) % 8u == 0) {
        // XCD-aware: workgroups go to the 8 XCDs round-robin in dispatch order, so with a multiple of 8 blocks per frame the
        // few (slower, latency-bound) edge blocks of EVERY frame of a batch land on the same XCDs.  1922x1082 at f = 2 -- 541 + 3
        // block rows -- ran at 66 % where 1922x1080 and 1922x1084 ran at 74 % (profiles/r04_avg_1922_sweep.log).  One more
        // (empty) block row per frame rotates them.
        gy_edge += 1;
    }
    a.bdx = bx; a.bdy = by; a.row_step = (int32_t)gy * by;
    a.edge_y0 = (fam == FAM_AVG) ? (int32_t)gy : 0x7FFFFFFF;
    d->grid = dim3(gx, gy + gy_edge, (unsigned)nframes);
    d->fn = fn;
    return CSIC_OK;
}

// The launch geometry k_avg would take for a planar AVG plan -- its packed twin's grid, block and KArgs (edge blocks included) --
// for the planar tile kernel, which runs k_avg's body (csic_avg_tile.h).  *tile = false: the tile kernel does not apply (no whole
// tile in the frame, CSIC_TUNE_FORCE_GENERIC / NO_VECTOR, variant 8 on a ragged shape).
int planar_avg_geometry(const csic_plan *pl, int nframes, LaunchDesc *d, bool *tile)
{
    csic_plan tmp = *pl;
    tmp.p.out_format = CSIC_FMT_YCBCR888X;
    select(&tmp);
    *tile = tmp.fam == FAM_AVG;
    if (!*tile) return CSIC_OK;
    return prepare_common(&tmp, 0, nframes, 0, 0, d);
}

int prepare_launch(const csic_plan *pl, const void *d_in, void *d_out, int nframes, int32_t in_pitch, int32_t out_pitch,
                   LaunchDesc *d)
{
    if (!pl) return set_error(CSIC_EINVAL_NULL, "plan is NULL");
    if (!d_in || !d_out) return set_error(CSIC_EINVAL_NULL, "device buffer is NULL");
    const int st = prepare_common(pl, (uintptr_t)d_in | (uintptr_t)d_out, nframes, in_pitch, out_pitch, d);
    if (st != CSIC_OK) return st;
    d->args.in = static_cast<const uint32_t *>(d_in);
    d->args.out = static_cast<uint32_t *>(d_out);
    return CSIC_OK;
}

int prepare_launch_table(const csic_plan *pl, const void *const *d_in_tab, void *const *d_out_tab, uintptr_t align_bits, int nframes,
                         LaunchDesc *d)
{
    if (!pl) return set_error(CSIC_EINVAL_NULL, "plan is NULL");
    if (!d_in_tab || !d_out_tab) return set_error(CSIC_EINVAL_NULL, "frame table is NULL");
    const int st = prepare_common(pl, align_bits, nframes, 0, 0, d);
    if (st != CSIC_OK) return st;
    d->args.in_tab = reinterpret_cast<const uint32_t *const *>(d_in_tab);
    d->args.out_tab = reinterpret_cast<uint32_t *const *>(d_out_tab);
    return CSIC_OK;
}

template <class T, size_t... I>
static void fill_ptrs(void **p, T &t, std::index_sequence<I...>) { ((p[I] = &std::get<I>(t)), ...); }

template <class... A, class... B>
static hipError_t launch_k(void (*k)(A...), dim3 grid, dim3 block, hipStream_t stream, B... args)
{
    std::tuple<A...> t(static_cast<A>(args)...);
    void *p[sizeof...(A)];
    fill_ptrs(p, t, std::index_sequence_for<A...>{});
    return hipLaunchKernel(reinterpret_cast<const void *>(k), grid, block, p, 0, stream);
}

// hipLaunchKernel reports this launch's own error: a pending error of the host's earlier, unrelated
// runtime calls is neither consumed nor mistaken for ours.
int enqueue(const LaunchDesc &d, hipStream_t stream)
{
    KArgs a = d.args;
    void *params[1] = {&a};
    HIP_TRY(hipLaunchKernel(reinterpret_cast<const void *>(d.fn), d.grid, d.block, params, 0, stream));
    return CSIC_OK;
}

static int launch(csic_plan *pl, const void *d_in, void *d_out, int nframes, hipStream_t stream,
                  int32_t in_pitch = 0, int32_t out_pitch = 0)
{
    if (!pl) return set_error(CSIC_EINVAL_NULL, "plan is NULL");
    if (nframes <= 0) return set_error(CSIC_EINVAL_SIZE, "nframes must be positive. Got %d", nframes);
    CSIC_DEVICE_SCOPE(pl->device);
    if (pl->p.out_format == CSIC_FMT_PLANAR) {
        if (in_pitch > 0 || out_pitch > 0)
            return set_error(CSIC_EINVAL_FORMAT, "planar output takes packed input rows and its own plane layout: no row pitches");
        const int st = planar_forward(pl, d_in, d_out, nframes, stream);
        if (st == CSIC_OK) clear_error();
        return st;
    }
    for (int f0 = 0; f0 < nframes; f0 += 65535) {     // grid z limit
        const int nz = (nframes - f0 < 65535) ? nframes - f0 : 65535;
        LaunchDesc d;
        int st = prepare_launch(pl, d_in, d_out, nz, in_pitch, out_pitch, &d);
        if (st != CSIC_OK) return st;
        d.args.in += (int64_t)f0 * d.args.in_frame_px;
        d.args.out += (int64_t)f0 * d.args.out_frame_px;
        st = enqueue(d, stream);
        if (st != CSIC_OK) return st;
    }
    clear_error();
    return CSIC_OK;
}

// entry points for csic_pipeline.hip
int launch_on_stream(csic_plan *pl, const void *d_in, void *d_out, int nframes, hipStream_t stream)
{
    return launch(pl, d_in, d_out, nframes, stream);
}
int plan_device(const csic_plan *pl) { return pl->device; }
const csic_params &plan_params(const csic_plan *pl) { return pl->p; }
const Geometry &plan_geometry(const csic_plan *pl) { return pl->g; }
int plan_variant(const csic_plan *pl) { return pl->variant; }
bool plan_nontemporal(const csic_plan *pl) { return !pl->no_nt; }
int plan_block_threads(const csic_plan *pl) { return pl->block_threads; }
int64_t plan_algorithmic_bytes(const csic_plan *pl)
{
    int64_t b = 0;
    return csic_algorithmic_bytes(&pl->p, &b) == CSIC_OK ? b : 0;
}
void plan_sizes(const csic_plan *pl, size_t *in_px, size_t *out_px)
{
    *in_px = (size_t)pl->g.W * pl->g.H;
    *out_px = (size_t)pl->g.Wo * pl->g.Ho;
    if (pl->p.out_format == CSIC_FMT_PLANAR) {           // what csic_process_host moves: the planar frame buffer, in 4-byte words
        csic_planar_layout L;
        planar_layout(pl->g, &pl->p, &L);
        *out_px = (size_t)(L.frame_bytes / 4);
    }
}
int32_t plan_width(const csic_plan *pl) { return pl->g.W; }
void plan_out_dims(const csic_plan *pl, int32_t *wo, int32_t *ho)
{
    *wo = pl->g.Wo;
    *ho = pl->g.Ho;
}

} // namespace csic

using namespace csic;

extern "C" {

int csic_device_count(void)
{
    int n = 0;
    hipError_t e = hipGetDeviceCount(&n);
    if (e == hipErrorNoDevice || (e == hipSuccess && n == 0))
        return set_error(CSIC_ENODEVICE, "no HIP device visible (libcsic_hip has no CPU fallback)");
    if (e != hipSuccess) return set_error(CSIC_EHIP, "hipGetDeviceCount failed: %s", hipGetErrorString(e));
    clear_error();
    return n;
}

int csic_plan_create(const csic_params *p, int device, csic_plan **out)
{
    if (!out) return set_error(CSIC_EINVAL_NULL, "out is NULL");
    *out = nullptr;
    Geometry g;
    int st = derive_geometry(p, &g);
    if (st != CSIC_OK) return st;
    const int n = csic_device_count();
    if (n < 0) return n;
    if (device < 0 || device >= n)
        return set_error(CSIC_ENODEVICE, "device %d out of range (have %d; there is no CPU backend)", device, n);
    csic_plan *pl = new (std::nothrow) csic_plan();
    if (!pl) return set_error(CSIC_ENOMEM, "out of host memory");
    std::memset(pl, 0, sizeof *pl);
    pl->p = *p; pl->g = g; pl->device = device;
    select(pl);
    *out = pl;
    clear_error();
    return CSIC_OK;
}

int csic_plan_destroy(csic_plan *plan)
{
    if (!plan) return CSIC_OK;
    if (plan->d_in || plan->d_out || plan->d_sum) {
        DeviceGuard guard(plan->device);
        if (guard.status() == 0) {
            if (plan->d_in) (void)hipFree(plan->d_in);
            if (plan->d_out) (void)hipFree(plan->d_out);
            if (plan->d_sum) (void)hipFree(plan->d_sum);
        }
    }
    delete plan;
    return CSIC_OK;
}

const char *csic_plan_kernel_name(const csic_plan *plan) { return plan ? plan->name : ""; }

int csic_plan_tune(csic_plan *plan, int32_t knob, int32_t value)
{
    if (!plan) return set_error(CSIC_EINVAL_NULL, "plan is NULL");
    if (knob == CSIC_TUNE_VARIANT) plan->variant = value;
    else if (knob == CSIC_TUNE_FORCE_GENERIC) plan->force_generic = value ? 1 : 0;
    else if (knob == CSIC_TUNE_NONTEMPORAL) plan->no_nt = value ? 0 : 1;
    else if (knob == CSIC_TUNE_NO_VECTOR) plan->no_vec = value ? 1 : 0;
    else if (knob == CSIC_TUNE_BLOCK_THREADS) {
        if (value != 0 && value != 64 && value != 128 && value != 256)
            return set_error(CSIC_EINVAL_SIZE, "block threads must be 0 (default), 64, 128 or 256. Got %d", value);
        plan->block_threads = value;
    }
    else return set_error(CSIC_EINVAL_SIZE, "unknown tuning knob %d", knob);
    select(plan);
    clear_error();
    return CSIC_OK;
}

int csic_process_device(csic_plan *plan, const void *d_in, void *d_out, void *hip_stream)
{
    return launch(plan, d_in, d_out, 1, static_cast<hipStream_t>(hip_stream));
}

int csic_process_batch_device(csic_plan *plan, const void *d_in, void *d_out, int32_t nframes, void *hip_stream)
{
    return launch(plan, d_in, d_out, nframes, static_cast<hipStream_t>(hip_stream));
}

int csic_process_pitched_device(csic_plan *plan, const void *d_in, int32_t in_pitch_px, void *d_out, int32_t out_pitch_px,
                                int32_t nframes, void *hip_stream)
{
    if (in_pitch_px <= 0 || out_pitch_px <= 0) return set_error(CSIC_EINVAL_SIZE, "row pitches must be positive");
    return launch(plan, d_in, d_out, nframes, static_cast<hipStream_t>(hip_stream), in_pitch_px, out_pitch_px);
}

int csic_debug_build(void)
{
#if defined(CSIC_DEBUG) && CSIC_DEBUG
    return 1;
#else
    return 0;
#endif
}

int csic_debug_probe_device(const void *d_frame, int32_t width, int32_t height, int64_t offset_px, void *d_sink, void *hip_stream)
{
    if (!d_frame || !d_sink) return set_error(CSIC_EINVAL_NULL, "argument is NULL");
    if (width <= 0 || height <= 0) return set_error(CSIC_EINVAL_DIMS, "width and height must be positive");
    Geometry g{};
    g.W = width; g.H = height; g.Wo = width; g.Ho = height; g.f = 1; g.h = 1; g.v = 1;
    KArgs a;
    fill_base_args(g, width, width, &a);
    a.in = static_cast<const uint32_t *>(d_frame);
    HIP_TRY(launch_k(k_debug_probe, dim3(1), dim3(64), static_cast<hipStream_t>(hip_stream), a, offset_px, static_cast<uint32_t *>(d_sink)));
    clear_error();
    return CSIC_OK;
}

int csic_plan_preferred_pitch(const csic_plan *plan, int32_t *in_pitch_px, int32_t *out_pitch_px)
{
    if (!plan || !in_pitch_px || !out_pitch_px) return set_error(CSIC_EINVAL_NULL, "argument is NULL");
    const Geometry &g = plan->g;
    // Measured on the round-4 kernels over 2048- to 16384-pixel rows x factors 1 / 2 / 4 / 8 x 13 (input pad, output pad) pairs
    // (tools/probe_pitch2.py, profiles/r04_probe_pitch.jsonl): packed rows are as fast as any padded layout, at every factor.
    //  * factor 2 / 4 / 8 (k_decflat / k_dec): 8192 f = 2: 82.9 % packed, 82.0-82.8 % padded; f = 8: 79.5 % packed, 74.5-77.7 %
    //    padded -- the +2-4 points round 2 measured for k_dec at 8192 went away with k_decflat's flat mapping;
    //  * factor 1: k_f1x4 (rounds 1-3) did gain 2-5 points from 1 KiB of padding on both sides (8192: 76.2 -> 80.3 %), and the
    //    first round-4 rule said so; k_f1flat reaches that rate on PACKED rows (79.7 / 79.1 %) and loses two points on padded
    //    ones (77.6 %: rows 3-4 of profiles/r04_bench_all_configs.jsonl as first measured with it; r04_probe_pitch_f1flat.jsonl).
    //  * pads of 16-64 pixels lose up to 10 points everywhere.
    // What looked like a property of the DRAMs was a property of two kernels' block-to-address mappings.  The answer is therefore
    // "packed" for every plan; the entry point stays so that a caller need not know that, and so that a future kernel can change it.
    *in_pitch_px = g.W;
    *out_pitch_px = g.Wo;
    clear_error();
    return CSIC_OK;
}

int csic_process_host(csic_plan *plan, const uint32_t *in, size_t in_px, uint32_t *out, size_t out_px)
{
    if (!plan) return set_error(CSIC_EINVAL_NULL, "plan is NULL");
    if (!in || !out) return set_error(CSIC_EINVAL_NULL, "host buffer is NULL");
    size_t need_in, need_out;
    plan_sizes(plan, &need_in, &need_out);
    if (in_px != need_in || out_px != need_out)
        return set_error(CSIC_EINVAL_SIZE, "expected %zu input and %zu output pixels, got %zu and %zu",
                         need_in, need_out, in_px, out_px);
    CSIC_DEVICE_SCOPE(plan->device);
    int st;
    if (!plan->d_in) HIP_TRY(hipMalloc(&plan->d_in, need_in * 4));
    if (!plan->d_out) HIP_TRY(hipMalloc(&plan->d_out, need_out * 4));
    HIP_TRY(hipMemcpyAsync(plan->d_in, in, need_in * 4, hipMemcpyHostToDevice, nullptr));
    st = launch(plan, plan->d_in, plan->d_out, 1, nullptr);
    if (st != CSIC_OK) return st;
    HIP_TRY(hipMemcpyAsync(out, plan->d_out, need_out * 4, hipMemcpyDeviceToHost, nullptr));
    HIP_TRY(hipStreamSynchronize(nullptr));
    clear_error();
    return CSIC_OK;
}

int csic_synth_frame_device(void *d_dst, int64_t npix, int64_t first_index, uint32_t seed, void *hip_stream)
{
    if (!d_dst) return set_error(CSIC_EINVAL_NULL, "d_dst is NULL");
    if (npix < 0) return set_error(CSIC_EINVAL_SIZE, "npix must be >= 0");
    if (npix == 0) return CSIC_OK;
    int64_t blocks = (npix + 255) / 256;
    if (blocks > 8192) blocks = 8192;
    HIP_TRY(launch_k(k_synth, dim3((unsigned)blocks), dim3(256), static_cast<hipStream_t>(hip_stream),
                     static_cast<uint32_t *>(d_dst), npix, first_index, seed * 0x9E3779B9u));
    clear_error();
    return CSIC_OK;
}

int csic_copy_device(void *d_dst, const void *d_src, int64_t npix, void *hip_stream)
{
    if (!d_dst || !d_src) return set_error(CSIC_EINVAL_NULL, "argument is NULL");
    if (npix < 0 || (npix & 3) || ((((uintptr_t)d_dst) | ((uintptr_t)d_src)) & 15u))
        return set_error(CSIC_EINVAL_SIZE, "csic_copy_device needs npix %% 4 == 0 and 16-byte aligned pointers");
    if (npix == 0) return CSIC_OK;
    const int64_t n4 = npix / 4, blocks = (n4 + 255) / 256;
    if (blocks > 0x7FFFFFFF) return set_error(CSIC_EINVAL_SIZE, "copy too large for one launch");
    HIP_TRY(launch_k(k_copy, dim3((unsigned)blocks), dim3(256), static_cast<hipStream_t>(hip_stream),
                     static_cast<const uint32_t *>(d_src), static_cast<uint32_t *>(d_dst), n4));
    clear_error();
    return CSIC_OK;
}

int csic_checksum_device(const void *d_src, int64_t npix, uint64_t *sum, void *hip_stream)
{
    if (!d_src || !sum) return set_error(CSIC_EINVAL_NULL, "argument is NULL");
    hipStream_t s = static_cast<hipStream_t>(hip_stream);
    unsigned long long *d_sum = nullptr;
    HIP_TRY(hipMalloc(reinterpret_cast<void **>(&d_sum), sizeof(unsigned long long)));
    hipError_t e = hipMemsetAsync(d_sum, 0, sizeof(unsigned long long), s);
    if (e == hipSuccess && npix > 0) {
        int64_t blocks = (npix + 255) / 256;
        if (blocks > 4096) blocks = 4096;
        e = launch_k(k_checksum, dim3((unsigned)blocks), dim3(256), s, static_cast<const uint32_t *>(d_src), npix, d_sum);
    }
    unsigned long long h = 0;
    if (e == hipSuccess) e = hipMemcpyAsync(&h, d_sum, sizeof h, hipMemcpyDeviceToHost, s);
    if (e == hipSuccess) e = hipStreamSynchronize(s);
    (void)hipFree(d_sum);
    if (e != hipSuccess) return set_error(CSIC_EHIP, "checksum failed: %s", hipGetErrorString(e));
    *sum = h;
    clear_error();
    return CSIC_OK;
}

} // extern "C"
