// csic_avg_tile.h -- the AVG extension's tile kernel body, shared by k_avg (packed output, csic_kernels.hip) and
// k_planar_avg_tile (planar output, csic_planar.hip).  Device code only; included after csic_kernel_ops.h.
#pragma once
#include "csic_kernel_ops.h"

namespace csic {

// ------------------------------------------------------------------------------------------------
// AVG sampling extension (CSIC_SAMPLING_AVG) -- NOT reference semantics.
// Box-filter chroma (h x v blocks) followed by f x f average pooling, order chroma -> spatial -> quant,
// integer rounding (sum + n/2) >> log2 n per stage, 8-bit values between stages, edge coordinates
// clamped.  This is what the reference's README and the north star describe; the reference code itself
// does sample-and-hold + decimation (k_dec / k_f1x4 above).  Normative statement: oracle/csic_oracle.c
// orc_process_avg.  All input rows are live here: algorithmic bytes = 4*W*H + 4*Wo*Ho.
//
// k_avg: one lane owns a 4-pixel-wide, TH-row tile (TH = max(v, f)) = TH 16-byte loads, so every chroma block and every
// pooling block with f <= 4 lies inside one lane's registers -- no LDS line buffer and no cross-lane traffic is needed.
// f = 8 spans two lanes: each sums its 4 x 8 half and the halves meet through one DPP quad_perm swap.
//
// Any frame shape (round 4; rounds 1-3 needed W % 4 == 0, H % TH == 0, Wo % 4 == 0 and 16-byte aligned rows, and everything
// else -- 1366x768, 1001x1001, but also 1368x768 at f = 4, whose OUTPUT rows are 342 pixels -- fell to the one-pixel-per-lane
// kernel at a tenth of the speed).  Tiles that lie inside the frame take the register path; the tiles of the last column /
// last tile row that the frame cuts take orc_process_avg's clamped definition verbatim, output pixel by output pixel
// (avg_edge_output) -- one lane per row and one row of waves per frame.  Clamped loads alone would NOT reproduce the
// definition: a pooling window that hangs over the edge by a whole chroma block would average a block made of the edge pixel
// only, where the definition re-reads the clamped pixel's own (partly real) block.  Rows and output rows need no alignment:
// gfx950 executes 16-byte global accesses at any 4-byte address (tools/ubench_unaligned.hip: 6.16-6.35 TB/s against 6.49
// aligned, no mismatches).
// ------------------------------------------------------------------------------------------------
// arithmetic + stores of one already loaded 4 x TH tile that lies inside the frame
template <int ROUND, int FMT, int F, int HH, int VV, int TH, class SINK>
__device__ __forceinline__ void avg_tile(const KArgs &a, const u32x4 (&p)[TH], const SINK &sink, int tr, int x4)
{
    constexpr int NLOG = (HH == 4 ? 2 : HH == 2 ? 1 : 0) + (VV == 2 ? 1 : 0);
    constexpr int FLOG2 = (F == 8 ? 6 : F == 4 ? 4 : F == 2 ? 2 : 0);
    const u16x2 qmask = {(unsigned short)a.mcb, (unsigned short)a.mcr};
    uint32_t Y[TH][4];
    u16x2 C[TH][4];
#pragma unroll
    for (int i = 0; i < TH; ++i) {
        const uint32_t px[4] = {p[i].x, p[i].y, p[i].z, p[i].w};
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            Y[i][j] = fwd_y(px[j]);
            C[i][j] = fwd_c_pk<ROUND>(px[j]);
        }
    }
    // chroma stage: HH x VV block averages, written back to every pixel of the block (sums <= 8 * 255: 16 bits hold them)
    if (HH * VV > 1) {
        const u16x2 half = {(HH * VV) >> 1, (HH * VV) >> 1};
#pragma unroll
        for (int bi = 0; bi < TH; bi += VV) {
#pragma unroll
            for (int bj = 0; bj < 4; bj += HH) {
                u16x2 s = {0, 0};
#pragma unroll
                for (int i = 0; i < VV; ++i)
#pragma unroll
                    for (int j = 0; j < HH; ++j) s += C[bi + i][bj + j];
                s = (s + half) >> (unsigned short)NLOG;
#pragma unroll
                for (int i = 0; i < VV; ++i)
#pragma unroll
                    for (int j = 0; j < HH; ++j) C[bi + i][bj + j] = s;
            }
        }
    }
    if (F <= 4) {
        constexpr int FF = (F <= 4) ? F : 4;        // (keeps the F = 8 instantiation well-formed)
        constexpr int NOX = 4 / FF, NOY = TH / FF;  // output pixels per tile
        const u16x2 half = {(FF * FF) >> 1, (FF * FF) >> 1};
#pragma unroll
        for (int oi = 0; oi < NOY; ++oi) {
            uint32_t o[NOX];
#pragma unroll
            for (int oj = 0; oj < NOX; ++oj) {
                uint32_t sy = 0;
                u16x2 sc = {0, 0};                   // <= 16 * 255
#pragma unroll
                for (int i = 0; i < FF; ++i)
#pragma unroll
                    for (int j = 0; j < FF; ++j) { sy += Y[oi * FF + i][oj * FF + j]; sc += C[oi * FF + i][oj * FF + j]; }
                sy = ((sy + ((FF * FF) >> 1)) >> FLOG2) & a.my;
                sc = ((sc + half) >> (unsigned short)FLOG2) & qmask;
                o[oj] = finish_y<FMT>(sy, chroma_term_q<FMT>(sc.x, sc.y));
            }
            sink.template put<NOX>(tr * NOY + oi, x4 * NOX, o);
        }
    } else {
        // F = 8: this lane's 4 x 8 half, then the neighbour's through a quad_perm [1,0,3,2] swap (sums <= 64 * 255 = 16 320)
        uint32_t sy = 0;
        u16x2 sc = {0, 0};
#pragma unroll
        for (int i = 0; i < TH; ++i)
#pragma unroll
            for (int j = 0; j < 4; ++j) { sy += Y[i][j]; sc += C[i][j]; }
        sy += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)sy, 0xB1 /* quad_perm:[1,0,3,2] */, 0xF, 0xF, false);
        const uint32_t scp = __builtin_bit_cast(uint32_t, sc);
        sc += __builtin_bit_cast(u16x2, (uint32_t)__builtin_amdgcn_update_dpp(0, (int)scp, 0xB1, 0xF, 0xF, false));
        if ((x4 & 1) == 0) {
            const u16x2 half = {32, 32};
            sy = ((sy + 32) >> 6) & a.my;
            sc = ((sc + half) >> (unsigned short)6) & qmask;
            const uint32_t o[1] = {finish_y<FMT>(sy, chroma_term_q<FMT>(sc.x, sc.y))};
            sink.template put<1>(tr, x4 >> 1, o);
        }
    }
}

// One output pixel whose pooling window the frame cuts, in registers and bit for bit the clamped definition.
// The window touches whole chroma blocks only (its origin is a multiple of F, blocks are powers of two): the REGION
// RW x RH = max(F, h) x max(F, v) pixels, aligned to itself, holds every block any pixel of the window belongs to.  All of it is
// loaded with clamped coordinates -- RW * RH independent loads in flight, where the run-time loops of avg_pixel_generic wait for
// each pixel in turn (1001x1001 f = 8: a lane spent 320 dependent loads on its one edge pixel while 63 waited; 21.8 % of the
// roofline against 82.8 % for 1000x1000).  Then, as the definition has it (orc_process_avg):
//   * a block whose origin is inside the frame averages its clamped pixels -- exactly what the clamped loads delivered;
//   * a pixel beyond the frame IS the clamped pixel: it takes that pixel's Y (the load gave it) and that pixel's BLOCK average,
//     not the average of its own, virtual block -- "hold the last real column, then the last real block row" over the region, a
//     chain of selects with compile-time indices (no dynamic register indexing).
template <int ROUND, int FMT, int F, int HH, int VV>
__device__ __forceinline__ uint32_t avg_edge_output(const KArgs &a, gin_t in, int ro, int co)
{
    constexpr int RW = F > HH ? F : HH, RH = F > VV ? F : VV;
    constexpr int NLOG = (HH == 4 ? 2 : HH == 2 ? 1 : 0) + (VV == 2 ? 1 : 0);
    constexpr int FLOG2 = (F == 8 ? 6 : F == 4 ? 4 : F == 2 ? 2 : 0);
    const int r0w = ro * F, c0w = co * F;
    const int rr0 = r0w & ~(RH - 1), rc0 = c0w & ~(RW - 1);
    const int imax = a.H - 1 - rr0, jmax = a.W - 1 - rc0;             // last real row / column in region coordinates (>= 0)
    const int wi0 = r0w - rr0, wj0 = c0w - rc0;                       // the window inside the region: all of it unless a chroma block
                                                                      // is larger than the window (h = 4 with F <= 2, v = 2 with F = 1)
    const u16x2 bhalf = {(HH * VV) >> 1, (HH * VV) >> 1}, zero = {0, 0};
    uint32_t sy = 0;
    u16x2 sc = {0, 0};
    u16x2 prev[RW];                                                   // the chroma of the block row above (the last real one, in the end)
#pragma unroll
    for (int j = 0; j < RW; ++j) prev[j] = zero;
    // one block row (VV pixel rows) at a time -- a rolled loop for the large regions: the whole kernel's occupancy is set by its
    // largest live range, and a 64-pixel region held in registers at once would cost the tile path two thirds of its waves
    constexpr int UNROLL = RW * RH <= 16 ? RH : 1;
#pragma unroll UNROLL
    for (int bi = 0; bi < RH; bi += VV) {
        uint32_t Y[VV][RW];
        u16x2 C[VV][RW];
#pragma unroll
        for (int i = 0; i < VV; ++i)
#pragma unroll
            for (int j = 0; j < RW; ++j) {
                const uint32_t px = in1<false>(a, in, (int64_t)min(rr0 + bi + i, a.H - 1) * a.ip + min(rc0 + j, a.W - 1));
                Y[i][j] = fwd_y(px);
                C[i][j] = fwd_c_pk<ROUND>(px);
            }
        // the block averages of this block row, one value per pixel column
        u16x2 R[RW];
#pragma unroll
        for (int bj = 0; bj < RW; bj += HH) {
            u16x2 s = {0, 0};
#pragma unroll
            for (int i = 0; i < VV; ++i)
#pragma unroll
                for (int j = 0; j < HH; ++j) s += C[i][bj + j];
            if (HH * VV > 1) s = (s + bhalf) >> (unsigned short)NLOG;
#pragma unroll
            for (int j = 0; j < HH; ++j) R[bj + j] = s;
        }
        // hold the last real column, then the last real (block) row
#pragma unroll
        for (int j = 1; j < RW; ++j) R[j] = (j > jmax) ? R[j - 1] : R[j];
#pragma unroll
        for (int j = 0; j < RW; ++j) { R[j] = (bi > imax) ? prev[j] : R[j]; prev[j] = R[j]; }
#pragma unroll
        for (int i = 0; i < VV; ++i)
#pragma unroll
            for (int j = 0; j < RW; ++j) {
                const bool inwin = (RH == F || (unsigned)(bi + i - wi0) < (unsigned)F) && (RW == F || (unsigned)(j - wj0) < (unsigned)F);
                sy += inwin ? Y[i][j] : 0u;
                sc += inwin ? R[j] : zero;
            }
    }
    const u16x2 half = {(F * F) >> 1, (F * F) >> 1};
    const u16x2 qmask = {(unsigned short)a.mcb, (unsigned short)a.mcr};
    sy = ((sy + ((F * F) >> 1)) >> FLOG2) & a.my;
    sc = ((sc + half) >> (unsigned short)FLOG2) & qmask;
    return finish_y<FMT>(sy, chroma_term_q<FMT>(sc.x, sc.y));
}

// TILES column groups per lane, spaced by the block width: all TILES * TH loads are issued before the first
// tile's arithmetic starts, so one tile's ~170 VALU ops overlap the other tiles' memory latency.
// Needs W >= 4 (8 at F = 8) and H >= TH: at least one whole tile for the clamped loads to fall back on (select_rf).
//
// Tiles the frame cuts produce nothing here.  The output pixels they would have produced -- the columns co >= Cw and the rows
// ro >= Rw that whole tiles do not reach -- belong to the EDGE BLOCKS appended to the grid (block rows blockIdx.y >= a.edge_y0;
// the host sizes them, prepare_common: the grid is a few blocks wide and hundreds tall, so rows of blocks waste none):
// one output pixel per lane through avg_edge_output, so that a wave of edge work is 64 lanes of edge work.  Two other placements were measured and dropped (profiles/r04_avg_edge_ab.log): evaluating a cut tile's outputs in
// the lane that owns the tile makes one lane of a wave do two to three tiles' work while 63 wait (1004x1000 f = 8: 73 % against
// 82 % for 1000x1000; 1020x1020: 64 %), and merely having that branch inside the tile loop cost the f = 2 tile path five
// points on frames that have no cut tile at all (8192x8192: 77.2 -> 71.9 %; the loop's load / arithmetic overlap did not
// survive the extra control flow).
// The kernel body takes its stores through a SINK (round 4): k_avg (csic_kernels.hip) packs pixels into the output frame, the
// planar tile kernel (csic_planar.hip) runs the same body with FMT = F_YCC and scatters Y / Cb / Cr bytes into the three planes.
//   sink.put<N>(row, col, o)   N consecutive output pixels of one row from a whole tile (N = 4 / F, or 1 at F = 8)
//   sink.put_edge(row, col, v) one output pixel of a cut tile
template <int ROUND, int FMT, int F, int HH, int VV, bool NT, int TILES, class SINK>
__device__ __forceinline__ void avg_kernel_body(const KArgs &a, gin_t in, const SINK &sink)
{
    constexpr int TH = (F > VV) ? F : VV;               // tile rows per lane
    const int W4 = (a.W + 3) >> 2, W4f = a.W >> 2;      // tiles per tile row (the last one possibly cut), whole tiles
    const int ntr = (a.H + TH - 1) / TH, ntrf = a.H / TH;
    if ((int)blockIdx.y >= a.edge_y0) {
        // edge blocks (block-uniform): lanes over the output pixels no whole tile produces -- the right-hand columns first
        // (all rows), then the bottom rows (the columns left of them)
        constexpr int NOXY = F <= 4 ? 4 / (F <= 4 ? F : 4) : 1;
        const int Cw = F == 8 ? (W4f >> 1) : W4f * NOXY;               // output columns / rows whole tiles produce
        const int Rw = F == 8 ? ntrf : ntrf * (TH / (F <= 4 ? F : 8));
        const int ecols = a.Wo - Cw, erows = a.Ho - Rw;
        const int nright = ecols * a.Ho, nbottom = erows * Cw;
        const int e = (((int)blockIdx.y - a.edge_y0) * (int)gridDim.x + (int)blockIdx.x) * (a.bdx * a.bdy) + (int)(threadIdx.y * a.bdx + threadIdx.x);
        int ro, co;
        if (e < nright) { ro = e / ecols; co = Cw + e - ro * ecols; }
        else if (e - nright < nbottom) { const int e2 = e - nright; const int q = e2 / Cw; ro = Rw + q; co = e2 - q * Cw; }
        else return;
        sink.put_edge(ro, co, avg_edge_output<ROUND, FMT, F, HH, VV>(a, in, ro, co));
        return;
    }
    const int x0 = blockIdx.x * (a.bdx * TILES) + threadIdx.x;
    if (x0 >= W4) return;
    for (int tr = blockIdx.y * a.bdy + threadIdx.y; tr < ntr; tr += a.row_step) {
        u32x4 p[TILES][TH];
        const int trc = min(tr, ntrf - 1);                     // a cut tile loads a whole one (unused) instead of branching
#pragma unroll
        for (int t = 0; t < TILES; ++t) {
            const int x4 = min(x0 + t * a.bdx, W4f - 1);       // clamp: out-of-row and cut tiles re-read the last whole one
#pragma unroll
            for (int i = 0; i < TH; ++i) p[t][i] = in4<NT>(a, in, (int64_t)(trc * TH + i) * a.ip + 4 * x4);
        }
#pragma unroll
        for (int t = 0; t < TILES; ++t) {
            const int x4 = x0 + t * a.bdx;
            // inside the frame?  F = 8: the PAIR of tiles that makes one output (the predicate is the same in both lanes)
            const bool whole = tr < ntrf && (F == 8 ? (x4 | 1) < W4f : x4 < W4f);
            if (whole) avg_tile<ROUND, FMT, F, HH, VV, TH>(a, p[t], sink, tr, x4);
        }
    }
}

} // namespace csic
