/*
 * csic_oracle.h -- CPU restatement of the reference's pixel-stream arithmetic.
 *
 * TEST INFRASTRUCTURE ONLY.  Nothing under oracle/ is part of the product path:
 * only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may load
 * this library, and there only as the checker / the reported CPU baseline.
 *
 * Parity status: PINNED.  The restatement reproduces all 29 golden PNGs the
 * reference commits (tests/golden/manifest.json) and the known-answer vectors
 * of its specs (tests/test_oracle_kat.py).  The single unpinned corner is the
 * spatial-before-chroma order class (no golden exists; follows
 * ImageCompressorTop.scala:44,52-58 + ChromaSubsampler.scala:37-38 by reading).
 *
 * The reference itself (Scala 2.13 + Chisel 3.6.1 on the JVM) cannot be built
 * in this image (no JDK/sbt/coursier, no network), so there is no oracle/_ref.
 *
 * All citations are relative to /root/reference/.
 */
#ifndef CSIC_ORACLE_H
#define CSIC_ORACLE_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* rounding of the forward transform */
#define ORC_ROUND_FLOOR_HW 0 /* RGB2YCbCr.scala:50-65, ReferenceModel.scala:15-17 */
#define ORC_ROUND_TRUNC_SW 1 /* RGB2YCbCr.scala:111-118 (Scala '/' truncates)     */

/* ProcessingStep ordinals, ImageCompressorTop.scala:7-9 */
#define ORC_OP_NOOP    0
#define ORC_OP_SPATIAL 1
#define ORC_OP_QUANT   2
#define ORC_OP_CHROMA  3

/* output pixel formats (uint32 per pixel, little endian) */
#define ORC_FMT_ARGB 0 /* byte0=B byte1=G byte2=R byte3=255  (Java int ARGB)  */
#define ORC_FMT_YCC  1 /* byte0=Y byte1=Cb byte2=Cr byte3=0  (what io.out carries) */

typedef struct orc_params {
    int32_t width, height;          /* input image                               */
    int32_t chroma_a, chroma_b;     /* J:a:b, ChromaSubsampler.scala:10-11       */
    int32_t y_bits, cb_bits, cr_bits; /* ColorQuantizer.scala:7-9                */
    int32_t factor;                 /* SpatialDownsampler.scala:6                */
    int32_t op[3];                  /* permutation of {1,2,3}                    */
    int32_t rounding;               /* ORC_ROUND_*                               */
    int32_t out_format;             /* ORC_FMT_*                                 */
    int32_t in_format;              /* ORC_FMT_ARGB, or ORC_FMT_YCC: the input already is the
                                       Y|Cb<<8|Cr<<16 stream a single stage is driven with in the
                                       reference's specs (forward transform skipped)            */
} orc_params;

/* ---- per-pixel functions -------------------------------------------------- */
void orc_rgb2ycbcr(int r, int g, int b, int rounding, int *y, int *cb, int *cr);
void orc_ycbcr2rgb(int y, int cb, int cr, int *r, int *g, int *b);
void orc_quantize(int y, int cb, int cr, int yb, int cbb, int crb,
                  int *yq, int *cbq, int *crq);

/* ---- whole-frame pipelines ------------------------------------------------ */
/* 0 on success, negative on invalid parameters (same rules as the reference's
 * require()s, see csic_oracle.c:orc_validate). */
int  orc_validate(const orc_params *p);
void orc_out_dims(const orc_params *p, int32_t *wo, int32_t *ho);

/* Streaming restatement: one pixel at a time through explicit stage state
 * machines wired in op[] order, exactly like ImageCompressorTop.scala:80-114.
 * `in` holds width*height ARGB pixels (alpha ignored); `out` receives
 * ceil(W/f)*ceil(H/f) pixels.  Returns the number of pixels emitted (<0 = error). */
long orc_process_stream(const orc_params *p, const uint32_t *in, uint32_t *out);

/* Closed-form restatement: every output pixel is
 * a gather of <= 2 input pixels.  Must agree with orc_process_stream. */
long orc_process_closed(const orc_params *p, const uint32_t *in, uint32_t *out);

/* Row-range variant of the closed form, for the row-stripe tests:
 * computes only output rows [ro0, ro1) into out + (ro0 * wo). */
long orc_process_closed_rows(const orc_params *p, const uint32_t *in, uint32_t *out,
                             int32_t ro0, int32_t ro1);

/* AVG sampling extension: box-filter chroma + average pooling, order chroma->spatial->quant only.
 * Build-defined semantics with NO counterpart in the reference ("parity unpinned"); see csic_oracle.c. */
long orc_process_avg(const orc_params *p, const uint32_t *in, uint32_t *out);

/* Closed form, output rows split over `nthreads` POSIX threads (CPU baseline on all host cores). */
long orc_process_closed_mt(const orc_params *p, const uint32_t *in, uint32_t *out, int nthreads);

/* ---- planar, subsampled form of the output stream (csic.h CSIC_FMT_PLANAR) -- */
/* Y at every output position, Cb / Cr at the positions where the chroma stage -- as the OUTPUT sees its
 * counters -- is at a sample point; see csic_oracle.c for the derivation and for what pins it. */
typedef struct orc_planar_layout {
    int32_t y_width, y_height, chroma_width, chroma_height;
    int32_t module_width, hold_h, hold_v, replay_last;
    int64_t chroma_samples;
} orc_planar_layout;
int  orc_planar_layout_of(const orc_params *p, int avg, orc_planar_layout *L);
long orc_planar_from_stream(const orc_planar_layout *L, const uint32_t *ycc, uint8_t *y, uint8_t *cb, uint8_t *cr);
long orc_planar_reconstruct(const orc_planar_layout *L, const uint8_t *y, const uint8_t *cb, const uint8_t *cr,
                            int fmt, uint32_t *out);

/* ---- per-stage helpers on YCbCr streams (used by the KAT tests) ----------- */
/* Chroma sample-and-hold on a stream of n (Y,Cb,Cr) triples, module width/height
 * W,H: ChromaSubsampler.scala:37-65 == ChromaSubsamplerImageSpec.scala:45-78. */
void orc_chroma_stream(const uint8_t *ycc_in, uint8_t *ycc_out, long n,
                       int W, int H, int a, int b);
/* Decimation of a W x H stream: SpatialDownsampler.scala:17-55.  Writes the
 * surviving stream indices into idx_out (capacity ceil*ceil); returns count. */
long orc_spatial_indices(int W, int H, int f, int64_t *idx_out);

/* ---- synthetic frames -------------------------------------- */
/* pixel i of frame k = 0xFF000000 | (fmix32(i + k*W*H + seed*0x9E3779B9) & 0xFFFFFF) */
void orc_synth_frame(uint32_t *dst, int64_t npix, int64_t first_index, uint32_t seed);

#ifdef __cplusplus
}
#endif
#endif
