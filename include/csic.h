/*
 * csic.h -- C ABI of libcsic_hip.so: the MI355X (gfx950) implementation of the reference's
 * pixel-stream hot path
 *
 *     RGB2YCbCr -> {ChromaSubsampler, SpatialDownsampler, ColorQuantizer in any order} -> YCbCr2RGB
 *
 * The reference (Scala/Chisel, /root/reference) has no FFI: its "interface" for this path is the
 * generator surface -- constructor parameter lists and require()s.  Each entry point below names the
 * reference interface it stands in for (paths relative to /root/reference/).  INTEGRATION.md shows the
 * JNI / Panama binding a maintainer of the reference would add on the Scala side.
 *
 * Conventions: plain C, no C++ types, no exceptions across the boundary.  Every function returning
 * `int` returns CSIC_OK (0) or a negative csic_status; a human-readable message for the calling
 * thread's last failure is available from csic_last_error().  The caller owns every pixel buffer; the
 * library owns only the opaque plan.  A plan is not thread-safe; distinct plans are independent.
 * There is NO CPU fallback: compute entry points fail with CSIC_ENODEVICE when no HIP device exists.
 *
 * Pixel layout (one uint32 per pixel, little endian):
 *   CSIC_FMT_ARGB8888 : byte0=B byte1=G byte2=R byte3=A  == Java `int` ARGB, what scrimage /
 *                       BufferedImage hand out (ImageProcessorModel.scala:47-48).  Input alpha is
 *                       ignored; output alpha is 255 (ImageCompressorTopApp.scala:139).
 *   CSIC_FMT_YCBCR888X: byte0=Y byte1=Cb byte2=Cr byte3=0 -- the PixelYCbCrBundle that
 *                       ImageCompressorTop.io.out carries (ImageCompressorTop.scala:35,
 *                       PixelBundle.scala:11-15), i.e. the pipeline WITHOUT the host-side inverse.  As
 *                       in_format it feeds a YCbCr stream straight into op1 (forward transform skipped):
 *                       how the reference's specs drive ONE stage (ChromaSubsamplerImageSpec.scala:150-170,
 *                       ColorQuantizerSpec.scala:72-100, SpatialDownsamplerSpec.scala:20-45).
 * Frames are row-major, tightly packed (row pitch = width * 4 bytes).
 */
#ifndef CSIC_H
#define CSIC_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define CSIC_ABI_VERSION 1

/* ---- status codes ------------------------------------------------------------------------------
 * Every CSIC_EINVAL_* corresponds to a require() that throws IllegalArgumentException when the
 * reference's generator is constructed; a host binding maps them to IllegalArgumentException. */
typedef enum csic_status {
    CSIC_OK                     =   0,
    CSIC_EINVAL_NULL            =  -1, /* null pointer argument                                       */
    CSIC_EINVAL_DIMS            =  -2, /* ImageProcessor.scala:22-23, ChromaSubsampler.scala:13-14     */
    CSIC_EINVAL_FACTOR          =  -3, /* ImageProcessor.scala:24, SpatialDownsampler.scala:8          */
    CSIC_EINVAL_CHROMA_A        =  -4, /* ImageProcessor.scala:27, ChromaSubsampler.scala:17           */
    CSIC_EINVAL_CHROMA_B        =  -5, /* ImageProcessor.scala:28, ChromaSubsampler.scala:18           */
    CSIC_EINVAL_BITS            =  -6, /* ColorQuantizer.scala:12-15                                   */
    CSIC_EINVAL_OP_PERMUTATION  =  -7, /* ImageCompressorTop.scala:27-31                               */
    CSIC_EINVAL_ROUNDING        =  -8,
    CSIC_EINVAL_FORMAT          =  -9,
    CSIC_EINVAL_NOT_DIVISIBLE   = -10, /* ImageProcessor.scala:25 (only when strict_divisible != 0)    */
    CSIC_EINVAL_SAMPLING        = -11,
    CSIC_EINVAL_STRIPE          = -12, /* row-stripe request that cannot be made independent           */
    CSIC_EINVAL_SIZE            = -13, /* buffer size does not match the plan                          */
    CSIC_ENODEVICE              = -20, /* no HIP device / bad device ordinal                           */
    CSIC_EHIP                   = -21, /* a HIP runtime call failed (message has the HIP error string) */
    CSIC_ENOMEM                 = -22,
    CSIC_ECAPTURE               = -23, /* the HIP stream is capturing and this operation cannot be captured */
    CSIC_EIO                    = -30, /* file cannot be opened / read / written                       */
    CSIC_EFORMAT                = -31  /* not a PNG, corrupt, or an unsupported PNG feature            */
} csic_status;

/* ---- enumerations -------------------------------------------------------------------------------*/
/* ProcessingStep ordinals, ImageCompressorTop.scala:7-9 */
#define CSIC_OP_NOOP      0
#define CSIC_OP_SPATIAL   1   /* ProcessingStep.SpatialSampling   */
#define CSIC_OP_QUANT     2   /* ProcessingStep.ColorQuantization */
#define CSIC_OP_CHROMA    3   /* ProcessingStep.ChromaSubsampling */

/* Forward-transform rounding: both exist in the reference and both are pinned by
 * committed golden images. */
#define CSIC_ROUND_FLOOR_HW 0 /* RTL + ReferenceModel: (x+128) >> 8     RGB2YCbCr.scala:50-65       */
#define CSIC_ROUND_TRUNC_SW 1 /* YCbCrUtils.rgbToYCbCr: (x+128) / 256   RGB2YCbCr.scala:111-118     */

/* Sampling semantics.
 * HOLD_DECIMATE (default) is the reference's: chroma sample-and-hold (ChromaSubsampler.scala:47-65) and
 *   top-left decimation (SpatialDownsampler.scala:33-55).  Everything "bit-exact" refers to this mode.
 * AVG is an EXTENSION with no counterpart in the reference code (its README and the project brief describe
 *   it): box-filter chroma over h x v blocks, then f x f average pooling, each (sum + n/2) >> log2 n on 8-bit
 *   values, edge coordinates clamped; defined for op = {CHROMA, SPATIAL, QUANT} only.  Its normative
 *   statement is oracle/csic_oracle.c:orc_process_avg; it has no reference parity by construction. */
#define CSIC_SAMPLING_HOLD_DECIMATE 0
#define CSIC_SAMPLING_AVG           1

#define CSIC_FMT_ARGB8888  0
#define CSIC_FMT_YCBCR888X 1
#define CSIC_FMT_PLANAR    2   /* out_format only: Y plane + Cb / Cr planes at the chroma sample points (csic_planar_layout) */

/* ---- parameters ---------------------------------------------------------------------------------
 * Field-for-field the constructor list of
 *   class ImageCompressorTop(width, height, chroma_param_a_config, chroma_param_b_config,
 *                            yTargetQuantBitsConfig, cbTargetQuantBitsConfig, crTargetQuantBitsConfig,
 *                            downFactorConfig, op1Type, op2Type, op3Type)   ImageCompressorTop.scala:11-25
 * (ImageProcessorParams, ImageProcessor.scala:15-21, is the subset width/height/factor/a/b with
 * bits = 8,8,8, op = {CHROMA, SPATIAL, QUANT} and strict_divisible = 1.) */
typedef struct csic_params {
    int32_t width, height;
    int32_t chroma_a, chroma_b;          /* J:a:b with J = 4: a in {4,2,1}, b in {a,0}                */
    int32_t y_bits, cb_bits, cr_bits;    /* 1..8 significant bits kept per channel                    */
    int32_t factor;                      /* spatial decimation factor 1,2,4,8                         */
    int32_t op[3];                       /* permutation of CSIC_OP_{SPATIAL,QUANT,CHROMA}             */
    int32_t rounding;                    /* CSIC_ROUND_*                                              */
    int32_t sampling;                    /* CSIC_SAMPLING_HOLD_DECIMATE (reference) or CSIC_SAMPLING_AVG  */
    int32_t in_format, out_format;       /* CSIC_FMT_* (CSIC_FMT_PLANAR: out_format only)             */
    int32_t strict_divisible;            /* 1 = enforce ImageProcessorParams' divisibility require()  */
} csic_params;

typedef struct csic_plan csic_plan;      /* opaque */

/* ---- host-only logic (usable without a GPU) -----------------------------------------------------*/
int  csic_abi_version(void);

/* Fills *p with 4:4:4, 8/8/8 bits, factor 1, order chroma->spatial->quant (the north-star order),
 * FLOOR_HW rounding, ARGB in/out, non-strict.  Mirrors ImageProcessorModel.getImageParams
 * (ImageProcessorModel.scala:33-41), which defaults chroma to 4:4:4. */
int  csic_params_default(csic_params *p, int32_t width, int32_t height);

/* All the reference's require()s in one call (see csic_status).  Replaces the construction-time checks
 * of ImageProcessorParams / ChromaSubsampler / SpatialDownsampler / ColorQuantizer / ImageCompressorTop. */
int  csic_validate(const csic_params *p);

/* Output frame size: ceil(W/f) x ceil(H/f) -- what SpatialDownsampler emits
 * (SpatialDownsampler.scala:33-55; 5x3,f=2 -> 3x2, SpatialDownsamplerSpec.scala:120-122). */
int  csic_out_dims(const csic_params *p, int32_t *out_width, int32_t *out_height);

/* Algorithmic HBM bytes of one frame, the roofline numerator of SURVEY.md 8(d):
 *   A = 4*W*ceil(H/f) + 4*ceil(W/f)*ceil(H/f)
 * (every input row that holds a surviving pixel, plus the output; rows r % f != 0 are dead).
 * With CSIC_SAMPLING_AVG every row is live: A = 4*W*H + 4*ceil(W/f)*ceil(H/f). */
int  csic_algorithmic_bytes(const csic_params *p, int64_t *bytes);

/* Row-stripe partition for `nranks` devices.  Stripe boundaries are aligned to
 * L = lcm(v, f) input rows (chroma before spatial) or v*f*f rows (spatial before chroma), which makes
 * every stripe an independent image of height *nrows: no halo, no collective.  Fails with
 * CSIC_EINVAL_STRIPE when the order class/shape cannot be split independently (spatial-before-chroma
 * with width % factor != 0).  Empty stripes (nrows == 0) are possible when nranks exceeds the number
 * of aligned row blocks. */
int  csic_stripe_rows(const csic_params *p, int32_t nranks, int32_t rank,
                      int32_t *row0, int32_t *nrows, int32_t *out_row0, int32_t *out_nrows);

/* Halo plan for stripes that are ALREADY partitioned at arbitrary rows (row_splits[0..nranks], with
 * row_splits[0] = 0 and row_splits[nranks] = height): rank r holds input rows [row_splits[r], row_splits[r+1]).
 * Every rank processes whole aligned L-row blocks [*proc_row0, *proc_row0 + *proc_nrows): it RECEIVES the
 * *halo_above = row_splits[r] - *proc_row0 rows in front of its stripe from rank r-1 and SENDS its trailing
 * *tail_below rows (those past its last aligned boundary) to rank r+1 -- one neighbour exchange of fewer than
 * L rows (the north star's "single RCCL halo exchange"; L as in csic_stripe_rows).  The processed range is an
 * independent image; its output rows are [*out_row0, *out_row0 + *out_nrows).  Fails with
 * CSIC_EINVAL_STRIPE if a stripe is shorter than the halo its successor needs. */
int  csic_stripe_halo(const csic_params *p, int32_t nranks, int32_t rank, const int32_t *row_splits,
                      int32_t *proc_row0, int32_t *proc_nrows, int32_t *halo_above, int32_t *tail_below,
                      int32_t *out_row0, int32_t *out_nrows);

/* ---- planar, genuinely subsampled output (out_format = CSIC_FMT_PLANAR) ------------------------------------------
 * Every packed output of the path is 4 bytes per pixel whatever the chroma mode: ChromaSubsampler re-emits the held Cb / Cr
 * with every pixel (ChromaSubsampler.scala:57-65), so nothing the reference produces is smaller than its input -- its README
 * describes the subsampled wire format (README.md:35-46), its code never builds it.  This format
 * stores each value the stream really carries ONCE: a Y plane with one byte per output pixel and two chroma planes with one
 * byte per chroma SAMPLE POINT.  4:2:0 at factor 1: 1.5 bytes per pixel instead of 4.
 *
 * Definition, on the output stream o[j], j = 0 .. n-1 (n = out_width * out_height, row-major), with the chroma stage's
 * counters as the output sees them -- c = j mod module_width, r = j div module_width:
 *     position j is a sample point  <=>  c % hold_h == 0  and  r % hold_v == 0
 *     sample index  k(j) = (r / hold_v) * chroma_width + c / hold_h,     chroma_width = ceil(module_width / hold_h)
 *     Y[j] = o[j].Y for every j;     Cb[k(j)] = o[j].Cb,  Cr[k(j)] = o[j].Cr  for every sample point j
 *   (module_width, hold_h, hold_v) follow from where the chroma stage sits:
 *     chroma before spatial, factor 1 : (W,  h, v)               -- the image's own 4:a:b grid
 *     chroma before spatial, factor f : (Wo, max(1, h / f), 1)   -- decimation keeps every f-th column of the held stream
 *     spatial before chroma           : (W,  h, v) over the DECIMATED stream: the chroma counters wrap at the FULL width
 *                                       (ImageCompressorTop.scala:52-58), so one chroma row spans f decimated rows and the
 *                                       last one may be partial (chroma_samples < chroma_width * chroma_height)
 *     CSIC_SAMPLING_AVG               : (Wo, max(1, h / f), max(1, v / f)), replay_last = 0
 * Values are what the packed YCbCr output holds at those positions (after the quantiser, in either rounding).
 *
 * csic_reconstruct_device is the inverse: planar -> packed ARGB (through YCbCrUtils.ycbcr2rgb, as the reference's harness
 * does per pixel, ImageCompressorTopApp.scala:118) or packed YCbCr.  Position j takes Y[j] and the chroma sample
 *     k(j)                                                      on rows r % hold_v == 0 (c / hold_h floors),
 *     ((r - 1) / hold_v) * chroma_width + chroma_width - 1      on the other rows when replay_last = 1: the reference's
 *                                                               4:x:0 hold replays the LAST sample of the row above for a
 *                                                               whole row (ChromaSubsampler.scala:52-65),
 *     (r / hold_v) * chroma_width + c / hold_h                  on the other rows when replay_last = 0 (AVG: plain box).
 * For every parameter set  reconstruct(planar(x)) == the packed output of the same parameters, bit for bit: under
 * HOLD_DECIMATE that pins the planar path to the reference exactly as far as the packed path is pinned; under AVG to
 * oracle/csic_oracle.c:orc_process_avg -- no reference parity by construction.
 *
 * Buffer: ONE allocation per frame, planes at 256-byte aligned offsets; frame k of a batch at k * frame_bytes.  Planes are
 * tightly packed (no row padding).  Bytes of a chroma plane beyond chroma_samples are never written. */
typedef struct csic_planar_layout {
    int32_t y_width, y_height;            /* = csic_out_dims                                                        */
    int32_t chroma_width, chroma_height;  /* samples per chroma row, chroma rows                                    */
    int32_t module_width;                 /* row length of the chroma counters over the output stream               */
    int32_t hold_h, hold_v;               /* one sample per hold_h positions, on every hold_v-th chroma row          */
    int32_t replay_last;                  /* 1: rows without samples replay the last sample of the row above (HOLD)  */
    int64_t chroma_samples;               /* samples per chroma plane that carry data                               */
    int64_t y_offset, cb_offset, cr_offset; /* byte offsets of the planes inside a frame's buffer                   */
    int64_t frame_bytes;                  /* bytes per frame buffer (multiple of 256)                               */
    int64_t payload_bytes;                /* y_width * y_height + 2 * chroma_samples: what the format really stores */
} csic_planar_layout;
/* usable without a GPU; p->out_format need not be CSIC_FMT_PLANAR (the layout of "these parameters, planar") */
int  csic_planar_layout_of(const csic_params *p, csic_planar_layout *layout);

const char *csic_strerror(int status);
const char *csic_last_error(void);       /* thread-local; "" when the last call succeeded */

/* ---- device path --------------------------------------------------------------------------------*/
int  csic_device_count(void);            /* >= 0, or CSIC_ENODEVICE / CSIC_EHIP */

/* Validates, selects and specialises the fused kernel for `p` on HIP device `device`.  Stands in for
 * `new ImageCompressorTop(...)` (ImageCompressorTopApp.scala:53-66) / `new ImageProcessor(params)`
 * (SpatialDownsamplerSpec.scala:180): parameters are generate-time constants there and plan-time
 * constants here.  There is no CPU backend: device must be >= 0. */
int  csic_plan_create(const csic_params *p, int device, csic_plan **out);
int  csic_plan_destroy(csic_plan *plan);

/* Name of the kernel variant the plan dispatches to (static string owned by the plan), for logs,
 * tests and matching rocprofv3 kernel names. */
const char *csic_plan_kernel_name(const csic_plan *plan);

/* Tuning knobs for A/B measurements; a knob the selected kernel does not have is ignored.
 *   CSIC_TUNE_VARIANT : kernel-family specific variant index (0 = default; 1, 2 = the 16-byte f = 2 kernels, 4 = k_dec for
 *                       f = 1, 5 = k_dec instead of k_decflat on rows that do not tile into whole blocks / waves, 6 = k_decflat wherever it applies, 7 = the one-pixel-per-lane k_generic instead of k_flatgen,
 *                       8 = AVG: the tile kernel only for frames of whole tiles, as in rounds 1-3, 9 = planar: the general kernels instead of the fast paths,
 *                       10 = planar, factor >= 2: 4 consecutive positions per lane (k_planar_flat) instead of the transposing k_planar_strided,
 *                       11 = factor 1: k_f1x4 (rounds 1-3's kernel) instead of k_f1flat,
 *                       12 = planar AVG at factor 1 on frames of whole tiles: k_avg's body with the planar sink instead of k_planar_avg_f1)
 *   CSIC_TUNE_FORCE_GENERIC : 1 = always use the one-thread-per-pixel generic kernel
 *   CSIC_TUNE_NONTEMPORAL   : 1 (default) = non-temporal loads/stores for the frame stream, 0 = cached
 *   CSIC_TUNE_NO_VECTOR     : 1 = never use the 16-byte-per-lane kernels
 *   CSIC_TUNE_BLOCK_THREADS : threads per block, 64 / 128 / 256 (0 = the library's choice) */
#define CSIC_TUNE_VARIANT        1
#define CSIC_TUNE_FORCE_GENERIC  2
#define CSIC_TUNE_NONTEMPORAL    3
#define CSIC_TUNE_NO_VECTOR      4   /* 1 = 4-byte accesses only (what unaligned pointers get automatically) */
#define CSIC_TUNE_BLOCK_THREADS  5   /* 0 (default) = the library's choice (256; 128 for a large single frame); 64 / 128 / 256 = forced */
int  csic_plan_tune(csic_plan *plan, int32_t knob, int32_t value);

/* One frame, device-resident: d_in holds width*height input pixels, d_out receives
 * out_width*out_height pixels.  Asynchronous on `hip_stream` (a hipStream_t, NULL = default stream);
 * no allocation, no synchronisation, hipGraph-capturable.  Replaces the per-pixel poke/peek loops
 * around the simulated RTL plus the host inverse transform
 * (ImageCompressorTopApp.scala:76-124, :118). */
int  csic_process_device(csic_plan *plan, const void *d_in, void *d_out, void *hip_stream);

/* `nframes` frames laid out back to back (frame k at d_in + k*W*H*4, d_out + k*Wo*Ho*4) in ONE launch
 * (frame index on the grid's z axis); same stream semantics as csic_process_device. */
int  csic_process_batch_device(csic_plan *plan, const void *d_in, void *d_out, int32_t nframes,
                               void *hip_stream);

/* The same for frames whose rows are NOT tightly packed: row r of frame k starts at
 * d_in + (k * height + r) * in_pitch_px pixels (resp. out_height / out_pitch_px for the output), with
 * in_pitch_px >= width and out_pitch_px >= out_width.  Padded decoder surfaces and regions of interest inside a
 * larger frame go through without a repacking copy (the reference streams pixels and has no pitch notion: the
 * semantic width -- chroma counters, ImageCompressorTop.scala:52-58 -- stays `width`; only addressing changes).
 * The 16-byte kernels need pitches that are multiples of 4 pixels, otherwise the 4-byte kernels are used. */
int  csic_process_pitched_device(csic_plan *plan, const void *d_in, int32_t in_pitch_px, void *d_out,
                                 int32_t out_pitch_px, int32_t nframes, void *hip_stream);

/* With out_format = CSIC_FMT_PLANAR, d_out of csic_process_device / csic_process_batch_device is a planar frame buffer of
 * csic_planar_layout.frame_bytes bytes per frame (256-byte aligned); csic_process_host and csic_pipeline_* hand the same
 * buffer to the host, and a frame graph of a planar plan (CSIC_FRAME_GRAPH_AUTO / _FUSED: one launch over the frames'
 * buffers) takes d_out[k] = frame k's planar buffer.  Row pitches, the per-frame-launch graph backends (_HIP, _DIRECT), the
 * file pools and csic_multi_* take packed formats only (CSIC_EINVAL_FORMAT otherwise).
 *
 * csic_reconstruct_device: `nframes` planar frames of `plan`'s parameters (the plan may have any out_format: only its
 * parameters matter) -> packed pixels, out_format = CSIC_FMT_ARGB8888 or CSIC_FMT_YCBCR888X, out_width * out_height per
 * frame back to back; asynchronous on `hip_stream`, no allocation, capturable. */
int  csic_reconstruct_device(csic_plan *plan, const void *d_planar, void *d_out, int32_t nframes, int32_t out_format,
                             void *hip_stream);

/* Row pitches, in pixels, at which frames of this plan stream fastest when the CALLER lays them out
 * (csic_process_pitched_device).  Measured on 2048- to 16384-pixel rows x every factor x 13 (input pad, output pad) pairs
 * (tools/probe_pitch2.py, profiles/r04_probe_pitch.jsonl, r04_probe_pitch_f1flat.jsonl): on the round-4 kernels packed rows are
 * as fast as any padded layout -- the gains rounds 2 and 3 saw from 1 KiB of padding (8192-pixel rows: +2-4 points at factor
 * 2 / 8 with k_dec, +2-5 points at factor 1 with k_f1x4) were properties of those kernels' block-to-address mappings and went
 * away with the flat mappings of k_decflat and k_f1flat; pads of 16-64 pixels lose up to 10 points.  So this returns the
 * widths themselves today.  It exists so that a caller that owns its surfaces asks instead of guessing. */
int  csic_plan_preferred_pitch(const csic_plan *plan, int32_t *in_pitch_px, int32_t *out_pitch_px);

/* ---- the range-checking build (diagnostics) ----------------------------------------------------------------------
 * `make -C chroma-subsampling-image-compressor_amd/csrc debug` builds libcsic_hip_debug.so from the same sources with
 * -DCSIC_DEBUG: every global access of the pixel kernels is checked against the frame's extent -- (H - 1) * pitch + W
 * pixels of input, (Ho - 1) * pitch + Wo of output, the planar frame's frame_bytes -- and a violation executes s_trap: the
 * queue reports a hardware exception and the process aborts instead of reading or writing a neighbour's memory silently.
 * It is the GPU-side stand-in for a sanitizer (GPU AddressSanitizer is not available on the pool); load it with
 * CSIC_LIB=<path> (Python host) or link it instead of libcsic_hip.so.  csic_debug_build() tells which build is loaded;
 * csic_debug_probe_device performs ONE checked read at pixel offset `offset_px` of a width x height frame (into *d_sink):
 * in the debug build an offset outside the frame traps, in the product build it is the caller's out-of-bounds read --
 * tests use it (in a child process) to show that the checks are live. */
int  csic_debug_build(void);
int  csic_debug_probe_device(const void *d_frame, int32_t width, int32_t height, int64_t offset_px, void *d_sink, void *hip_stream);

/* Convenience synchronous host path: H2D + kernel + D2H through plan-owned staging buffers.
 * in_px must equal width*height and out_px out_width*out_height (planar: frame_bytes / 4). */
int  csic_process_host(csic_plan *plan, const uint32_t *in, size_t in_px, uint32_t *out, size_t out_px);

/* Synthetic frame generator of SURVEY.md 8(d), on the device:
 *   dst[i] = 0xFF000000 | (fmix32((uint32)(first_index + i) + seed * 0x9E3779B9) & 0xFFFFFF)
 * Used by bench.py and the full-size property tests so that frames never cross PCIe. */
int  csic_synth_frame_device(void *d_dst, int64_t npix, int64_t first_index, uint32_t seed,
                             void *hip_stream);

/* Plain device-to-device copy of npix pixels with 16-byte-per-lane non-temporal accesses: the measured
 * streaming ceiling that bench.py reports next to the spec-peak roofline.  npix % 4 == 0,
 * 16-byte aligned pointers; asynchronous on `hip_stream`. */
int  csic_copy_device(void *d_dst, const void *d_src, int64_t npix, void *hip_stream);

/* 64-bit order-sensitive checksum of npix pixels in device memory (sum of fmix32-mixed
 * (pixel, index) pairs), for the full-size parity properties; synchronous. */
int  csic_checksum_device(const void *d_src, int64_t npix, uint64_t *sum, void *hip_stream);

/* ---- pre-recorded per-frame launches (BASELINE.json configs[4], "hipGraph-captured per-frame launch") ------
 * A stream of frames that live in SEPARATE device buffers (a decoder's surface pool) cannot use the one
 * contiguous batched launch of csic_process_batch_device, and a frame of 10-25 MB is only 1.3-3 us of HBM time
 * behind the ~1.7 us dependent-kernel boundary that every launch on a HIP stream -- eager or replayed from a
 * hipGraph chain -- pays.  A frame graph records one launch per frame once (frame k: d_in[k] -> d_out[k]; same
 * preconditions as csic_process_device) and replays them so that independent frames overlap.  The reference
 * processes images strictly one after the other (ImageCompressorTopApp.scala:23-145, a fresh DUT per image);
 * frames are independent, so no ordering between them has to be kept.  Three backends:
 *
 *   CSIC_FRAME_GRAPH_HIP     `branches` hipGraph chains (kernel nodes identical to the eager launch), chain 0
 *                            replayed on the caller's stream, the others on internal streams forked from and
 *                            joined to it by events: csic_frame_graph_launch is asynchronous and fully ordered
 *                            with `hip_stream`.  branches = 1 is the strictly serial graph that capturing a loop
 *                            of csic_process_device calls gives.  (ONE hipGraph with parallel branches is not
 *                            used: ROCm 7.2 replays such a graph node by node from the host, slower than a chain.)
 *   CSIC_FRAME_GRAPH_DIRECT  the same launches as pre-built AQL kernel-dispatch packets on `branches` (= queues,
 *                            default up to 3, max 8) user-mode HSA queues owned by the library, frame k on queue
 *                            k % queues, WITHOUT the barrier bit HIP sets on every packet: consecutive frames
 *                            overlap like the workgroups of one big launch (cfg 5 frame: 3.65 us in a hipGraph
 *                            chain, 1.76 us here, 1.75 us in one batched launch).  These queues are NOT HIP
 *                            streams: csic_frame_graph_submit starts the work immediately -- the caller makes
 *                            the inputs ready first (e.g. by synchronising the producing stream) -- and returns
 *                            a ticket; csic_frame_graph_wait blocks the host until that submission (and all
 *                            earlier ones of this graph) has finished, outputs visible to host and device.
 *                            csic_frame_graph_launch(graph, hip_stream) orders the same work with a HIP stream on
 *                            the device, asynchronously: the graph's signals are HIP "signal memory"
 *                            (hipExtMallocWithFlags(.., hipMallocSignalMemory) -- the value word of an HSA signal,
 *                            usable as hsa_signal_t in AQL barrier packets and readable / writable by kernels).
 *                            Every queue starts with a one-wave gate kernel that spins on the gate word; ONE one-wave
 *                            hand-off kernel on `hip_stream` opens the gate when the stream reaches it and then spins
 *                            until every queue's closing packet has zeroed its done word; work enqueued on
 *                            `hip_stream` afterwards sees the outputs.  (A dependency resolved by the command
 *                            processor polling a signal -- AQL barrier packets, hipStreamWaitValue64 -- costs about
 *                            10 us per hop on this part; a polling wave reacts within a microsecond.  Both spins are
 *                            bounded: after CSIC_DIRECT_TIMEOUT_MS (environment, default 30000) the kernel flags the
 *                            graph and returns, and the next launch / wait of that graph reports CSIC_EHIP.
 *                            CSIC_DIRECT_HANDOFF=cp in the environment keeps the command-processor form -- gate
 *                            barrier packets opened by hipStreamWriteValue64, hipStreamWaitValue64 on the closing
 *                            signal -- for comparison.)
 *                            csic_frame_graph_stream_ordered() tells whether the runtime offered this (else launch
 *                            degrades to hipStreamSynchronize + submit + wait).  A gate blocks its queues until the
 *                            stream reaches it, so do not make that stream's earlier work depend on a LATER direct
 *                            submission.  Up to 16 submissions/launches may be outstanding per graph (a 17th waits
 *                            for the oldest); host-ordered submissions are not ordered among themselves.
 *                            Queue count: the device runs 4 queues at once.  Host-ordered submissions are fastest
 *                            on 4 (cfg 5 frame 1.71 us, 3 queues 1.78 us).  A stream-ordered launch also keeps the
 *                            launch stream's own queue busy (its hand-off kernel runs as long as the frames do), and with
 *                            4 + 1 active queues the hardware scheduler time-slices them (64 frames: 250 us instead of
 *                            126 us) -- so csic_frame_graph_launch NEVER uses more than 3 queues: a graph created with
 *                            more keeps a second dealing of its frames over 3 queues for launches and uses all of them
 *                            for csic_frame_graph_submit.  A stream-ordered launch costs about 8 us of hand-off on top
 *                            of the host-ordered time (17-40 us in the command-processor form): record several frames
 *                            per graph.
 *                            Limits a caller must know:
 *                            - NOT capturable.  csic_frame_graph_launch on a capturing stream returns CSIC_ECAPTURE (the
 *                              packets would go out at capture time and a replay would find nothing behind its hand-off).
 *                            - The gate waves are dispatched when the host calls launch, not when the stream gets there:
 *                              stream work in front of a launch that runs LONGER than CSIC_DIRECT_TIMEOUT_MS makes the
 *                              gate give up -- the frames then run on inputs that may not be ready, the graph's error
 *                              word is set, and the next launch / wait / destroy of that graph reports CSIC_EHIP.  Raise
 *                              the timeout for such hosts; it bounds how long a wave may spin, nothing else.
 *                            - Waiting for ring space is bounded too (CSIC_DIRECT_SUBMIT_TIMEOUT_MS, default twice the
 *                              device bound + 5 s).  Room is checked on every queue BEFORE anything is written, so a
 *                              submission that cannot start fails cleanly (nothing queued, the engine stays usable).
 *                              Only a graph larger than the rings (> 4096 packets per queue, host-ordered) can fail between
 *                              two chunks; that marks the device's engine failed: every later DIRECT call on the device
 *                              returns CSIC_EHIP with the first failure's message, and what the stuck packets refer to
 *                              is leaked instead of freed.
 *
 *   CSIC_FRAME_GRAPH_FUSED   not per-frame launches at all: ONE kernel launch covers every frame (frame index on the
 *                            grid's z axis, the frame bases read from a device-resident pointer table the graph owns), so
 *                            frames in separate buffers run exactly like csic_process_batch_device's contiguous batch.
 *                            csic_frame_graph_launch is an ordinary asynchronous launch on `hip_stream`: fully ordered,
 *                            hipGraph-capturable, no internal streams or queues (`branches` is ignored).
 *
 * A graph, like a plan, is not thread-safe (one thread at a time per graph, including its ticket bookkeeping; different
 * graphs may be used from different threads, the library serialises their access to its queues).
 * csic_frame_graph_destroy waits for the graph's DIRECT submissions and for the HIP backend's internal streams; work that
 * csic_frame_graph_launch put on the CALLER's stream (chain 0 of the HIP backend, the fused launch) is the caller's to
 * synchronise before destroying the graph, as with any resource a stream still uses.
 * branches <= 0 selects the backend's default for the frame size (more overlap for smaller frames; measured table
 * in profiles/r02_small_launch.md).  The pointer arrays are read at creation only; the buffers they
 * name must stay valid for as long as the graph is launched.  csic_frame_graph_create == _create_ex with
 * CSIC_FRAME_GRAPH_AUTO, which today always resolves to FUSED: the plain entry point gives the fastest stream-ordered
 * path (cfg 5: 76 % of the HBM roofline against 42 % for hipGraph chains); per-frame launches are there for callers who
 * name them. */
#define CSIC_FRAME_GRAPH_HIP    0
#define CSIC_FRAME_GRAPH_DIRECT 1
#define CSIC_FRAME_GRAPH_FUSED  2
#define CSIC_FRAME_GRAPH_AUTO   3   /* the library's choice: FUSED (all frames of a graph share one plan, so it always applies) */
#define CSIC_FRAME_GRAPH_DEFAULT_BRANCHES 4   /* HIP backend, small frames (2 chains when a frame is >= 5 us of HBM time)          */
#define CSIC_FRAME_GRAPH_DEFAULT_QUEUES   3   /* DIRECT backend, small frames (2 queues from 2.5 us, 1 queue from 10 us per frame); see below */
typedef struct csic_frame_graph csic_frame_graph;
int  csic_frame_graph_create(csic_plan *plan, const void *const *d_in, void *const *d_out, int32_t nframes,
                             int32_t branches, csic_frame_graph **out);
int  csic_frame_graph_create_ex(csic_plan *plan, const void *const *d_in, void *const *d_out, int32_t nframes,
                                int32_t branches, int32_t backend, csic_frame_graph **out);
int  csic_frame_graph_launch(csic_frame_graph *graph, void *hip_stream);
int  csic_frame_graph_submit(csic_frame_graph *graph, int64_t *ticket);            /* DIRECT only */
int  csic_frame_graph_wait(csic_frame_graph *graph, int64_t ticket);               /* DIRECT only; ticket < 0 = all */
int  csic_frame_graph_count(const csic_frame_graph *graph, int32_t *nframes, int32_t *branches);
int  csic_frame_graph_backend(const csic_frame_graph *graph);                      /* the RESOLVED backend (never AUTO), or < 0 */
int  csic_frame_graph_launch_branches(const csic_frame_graph *graph);              /* chains / queues csic_frame_graph_launch uses: DIRECT min(branches, 3), FUSED 1 */
int  csic_frame_graph_stream_ordered(const csic_frame_graph *graph);               /* 1: launch() is asynchronous and ordered with its stream */
int  csic_frame_graph_destroy(csic_frame_graph *graph);

/* ---- PNG files (host only; reading: the library's own inflate, writing: zlib) ------------------------
 * The codec either side of the path: stands in for scrimage's loader / PngWriter behind
 * ImageProcessorModel.readImage / writeImage (ImageProcessorModel.scala:14-22).  Decoding yields straight
 * 8-bit samples as ARGB ints with alpha = 255 (input alpha dropped, gAMA/cHRM not applied -- the behaviour
 * the reference's golden images pin) and writes directly into `dst`, which may be a pinned
 * buffer from csic_pipeline_acquire_input.  Non-interlaced PNGs of every colour type / bit depth are read;
 * 8-bit RGB is written (`level` = zlib level 0..9; at level 1 a frame encodes about four times faster than at 6),
 * images of more than 1 MiB of filtered data on up to 16 threads (CSIC_PNG_THREADS=n sets the number; the bytes
 * written do not depend on it: the deflate stream is cut into pieces by the data alone).
 * The reader accepts and rejects exactly what zlib's inflate does (csrc/csic_inflate.cpp); environment variable
 * CSIC_NO_SIMD=1 keeps it off the PCLMULQDQ / SSSE3 paths it otherwise takes where the CPU has them. */
int  csic_png_info(const char *path, int32_t *width, int32_t *height);
int  csic_png_read_argb(const char *path, uint32_t *dst, size_t dst_px);
int  csic_png_write_argb(const char *path, const uint32_t *src, int32_t width, int32_t height, int32_t level);

/* ---- host-frame pipeline (the step either side of the hot path) -----------------------------------
 * Replaces the reference's per-image  readImage -> per-pixel poke ... peek -> writeImage  flow
 * (ImageProcessorModel.scala:14-52, ImageCompressorTopApp.scala:39-41,76-144) for streams of frames that
 * live in host memory: `depth` slots, each with pinned host input/output staging, device buffers and its
 * own HIP stream, so that frame k+1's H2D copy, frame k's kernel and frame k-1's D2H copy overlap.
 *
 *   csic_pipeline_acquire_input : pointer to the next slot's pinned input buffer (width*height pixels);
 *                                 decode/write the frame straight into it.  Fails with CSIC_EINVAL_SIZE
 *                                 when every slot still holds an uncollected frame.
 *   csic_pipeline_submit        : enqueue H2D + kernel + D2H for the acquired buffer (asynchronous).
 *   csic_pipeline_collect       : wait for the OLDEST submitted frame; *host_out points at its pinned
 *                                 output (out_width*out_height pixels; for a CSIC_FMT_PLANAR plan the planar
 *                                 frame buffer of csic_planar_layout_of, frame_bytes long -- 1.5 bytes per
 *                                 pixel instead of 4 on the way back for 4:2:0), valid until that slot is
 *                                 submitted again.  Frames complete in submission order; *ticket counts from 0.
 * The plan must outlive the pipeline.  Not thread-safe (one producer/consumer thread). */
typedef struct csic_pipeline csic_pipeline;
int  csic_pipeline_create(csic_plan *plan, int32_t depth, csic_pipeline **out);
int  csic_pipeline_destroy(csic_pipeline *pipeline);
int  csic_pipeline_acquire_input(csic_pipeline *pipeline, uint32_t **host_in);
int  csic_pipeline_submit(csic_pipeline *pipeline, int64_t *ticket);
int  csic_pipeline_collect(csic_pipeline *pipeline, const uint32_t **host_out, int64_t *ticket);
int  csic_pipeline_pending(const csic_pipeline *pipeline);
/* CSIC_PIPELINE_ZERO_COPY (default): the kernel reads the pinned host input and writes the pinned host
 *     output directly over PCIe: dead input rows (r % factor != 0) never cross the bus and both PCIe
 *     directions are busy inside one launch (8192x8192 sf=2: 2.4 ms/frame vs 5.9 ms staged).
 * CSIC_PIPELINE_STAGED: hipMemcpyAsync H2D -> kernel -> hipMemcpyAsync D2H through device buffers; keeps
 *     the CUs free while the copy engines move the frame. */
#define CSIC_PIPELINE_STAGED    0
#define CSIC_PIPELINE_ZERO_COPY 1
int  csic_pipeline_set_mode(csic_pipeline *pipeline, int32_t mode);

/* ---- PNG files in, PNG files out: the pipeline with the codec on worker threads ---------------------------------
 * The batch counterpart of the reference's per-image  readImage -> DUT -> writeImage  flow (ImageProcessorModel.scala:14-52,
 * ImageCompressorTopApp.scala:39-41,133-144) for `nfiles` PNGs that are all frames of `plan`: `decode_threads` workers
 * decode straight into pinned frame slots and launch the fused kernel on the slot's stream (zero-copy over PCIe, as
 * CSIC_PIPELINE_ZERO_COPY), `encode_threads` workers wait for a slot and write out_paths[i] (8-bit RGB, zlib `png_level`).
 * <= 0 threads = the library's choice (up to 32 decoders / 16 encoders, never more than files or host cores); the slots
 * between the two pools are bounded (workers + 2, at most 8 GiB of pinned memory).  final_width / final_height > 0 select
 * what the reference's collector keeps when the dimensions do not divide by the factor -- the first final_width *
 * final_height pixels of the output stream, final_width per row, missing pixels magenta (ImageCompressorTopApp.scala:44-45,
 * 133-142); 0 = the plan's output size.  Every output file is byte for byte what csic_png_read_argb -> csic_pipeline_* ->
 * csic_png_write_argb on one thread writes for the same input.  The parent directories of out_paths must exist.
 * Synchronous; the first failure stops the batch and is returned (message: "file <i>: ...").  *stats may be NULL. */
typedef struct csic_files_stats {
    int64_t frames;                   /* files written                                                                   */
    double  wall_s;                   /* first worker started -> last worker finished                                     */
    double  decode_s, encode_s;       /* summed over the workers of each pool: seconds inside the PNG decoder / encoder   */
    double  gpu_wait_s, slot_wait_s;  /* encoders waiting for the GPU; decoders waiting for a free slot                    */
    int32_t decode_threads, encode_threads, slots, max_in_flight;   /* slots = pinned frame slots actually allocated */
    int64_t in_pixels, out_pixels;
} csic_files_stats;
int  csic_process_png_files(csic_plan *plan, const char *const *in_paths, const char *const *out_paths, int32_t nfiles,
                            int32_t decode_threads, int32_t encode_threads, int32_t png_level,
                            int32_t final_width, int32_t final_height, csic_files_stats *stats);

/* ---- cycle-level model of the Decoupled pixel stream (host only) ---------------------------
 * What the reference's users simulate is not a function from frames to frames but hardware: modules that exchange one
 * pixel per ready/valid handshake (ImageCompressorTop.scala:33-38), and its tests check the handshake as well as the
 * pixels (back-pressure: SpatialDownsamplerSpec.scala:48-58; the cycle budget of the app's collector:
 * ImageCompressorTopApp.scala:110).  A csic_stream is that interface, clock edge by clock edge, for
 *   CSIC_STREAM_TOP        class ImageCompressorTop: RGB2YCbCr -> Queue(1) -> op1 -> Queue(1) -> op2 -> Queue(1) -> op3
 *                          (ImageCompressorTop.scala:63-65,80-114; the queues are chisel3.util.Queue, pipe = flow = false)
 *   CSIC_STREAM_PROCESSOR  class ImageProcessor: RGB2YCbCr -> ChromaSubsampler -> SpatialDownsampler, no queues
 *                          (ImageProcessor.scala:42-62)
 *   CSIC_STREAM_RGB2YCBCR / _CHROMA / _SPATIAL / _QUANT   one module alone, as the reference's specs drive it
 * built from the same csic_params (and rejected by the same require()s; FLOOR_HW and HOLD_DECIMATE only: the RTL has no
 * other rounding or sampling).  in_bits is a PixelBundle (CSIC_FMT_ARGB8888) where the chain starts with RGB2YCbCr, else
 * a PixelYCbCrBundle (CSIC_FMT_YCBCR888X); out_bits is the PixelYCbCrBundle on io.out, or -- params.out_format ==
 * CSIC_FMT_ARGB8888 -- that pixel put through YCbCrUtils.ycbcr2rgb as the harness does (ImageCompressorTopApp.scala:118).
 * sof / eol are accepted and, as in the RTL (SpatialDownsampler.scala:11-12), read by no logic.
 *
 * This is a simulator of interface timing (a few Mpixel/s on one host core), NOT a compute path: no csic_process_* /
 * csic_plan_* / csic_frame_graph_* call ever reaches it, and they still fail with CSIC_ENODEVICE without a GPU.  RTL /
 * FIRRTL emission would need Chisel and is out of scope.
 *
 *   csic_stream_eval : the combinational outputs (in_ready, out_valid, out_bits) for the present state and inputs -- a
 *                      chiseltest peek(); the state does not change.
 *   csic_stream_step : the same outputs (may be NULL), then one rising clock edge -- poke(...); clock.step().
 *   csic_stream_run  : the reference harness's two loops in one call (ImageCompressorTopApp.scala:76-124): the driver keeps
 *                      a pixel on the wires until the edge at which in_ready is high, the collector takes a pixel at every
 *                      edge where out_valid && out_ready and stops after max_out pixels or max_cycles cycles (< 0: until the
 *                      input is used up and the pipeline has drained).  in_valid_pattern / out_ready_pattern (cyclic over
 *                      the cycle number, NULL = always 1) add producer gaps and back-pressure.  *n_out pixels were
 *                      collected in *cycles cycles; the stream's state carries over to the next call. */
#define CSIC_STREAM_TOP        0
#define CSIC_STREAM_PROCESSOR  1
#define CSIC_STREAM_RGB2YCBCR  2
#define CSIC_STREAM_CHROMA     3
#define CSIC_STREAM_SPATIAL    4
#define CSIC_STREAM_QUANT      5
typedef struct csic_stream csic_stream;
typedef struct csic_stream_in  { int32_t in_valid; uint32_t in_bits; int32_t out_ready; int32_t sof, eol; } csic_stream_in;
typedef struct csic_stream_out { int32_t in_ready; int32_t out_valid; uint32_t out_bits; } csic_stream_out;
int     csic_stream_create(const csic_params *p, int32_t kind, csic_stream **out);
int     csic_stream_destroy(csic_stream *stream);
int     csic_stream_reset(csic_stream *stream);                       /* every register back to its RegInit value, cycle count 0 */
int     csic_stream_eval(const csic_stream *stream, const csic_stream_in *in, csic_stream_out *out);
int     csic_stream_step(csic_stream *stream, const csic_stream_in *in, csic_stream_out *out);
int64_t csic_stream_cycles(const csic_stream *stream);                /* clock edges since creation / reset */
int     csic_stream_depth(const csic_stream *stream);                 /* modules in the chain (queues included): 7 for TOP */
int     csic_stream_run(csic_stream *stream, const uint32_t *in, size_t n_in, uint32_t *out, size_t max_out, int64_t max_cycles,
                        const uint8_t *in_valid_pattern, size_t in_pattern_len,
                        const uint8_t *out_ready_pattern, size_t out_pattern_len, size_t *n_out, int64_t *cycles);

/* ---- several devices from one process --------------------------------------------------------------
 * The same aligned row-stripe partition as csic_stripe_rows, for hosts that own all GPUs in ONE process
 * (a JVM, a C++ service); bench.py and the Python driver use one process per GPU instead.  Stripe i runs on
 * devices[i] (a device may be listed more than once) on a stream of its own; stripes are independent images,
 * so there is no halo and no peer traffic.
 *   csic_multi_process_device : d_in[i] / d_out[i] are device pointers ON devices[i] holding stripe i's input
 *                               rows / receiving its output rows (see csic_multi_stripe); asynchronous --
 *                               finish with csic_multi_synchronize.
 *   csic_multi_process_host   : whole frame in host memory: scatter, process, gather; synchronous. */
typedef struct csic_multi csic_multi;
int  csic_multi_create(const csic_params *p, const int32_t *devices, int32_t ndev, csic_multi **out);
int  csic_multi_destroy(csic_multi *multi);
int  csic_multi_count(const csic_multi *multi);
int  csic_multi_stripe(const csic_multi *multi, int32_t idx, int32_t *device, int32_t *row0, int32_t *nrows,
                       int32_t *out_row0, int32_t *out_nrows);
int  csic_multi_process_device(csic_multi *multi, const void *const *d_in, void *const *d_out);
int  csic_multi_synchronize(csic_multi *multi);
int  csic_multi_process_host(csic_multi *multi, const uint32_t *in, size_t in_px, uint32_t *out, size_t out_px);

#ifdef __cplusplus
}
#endif
#endif /* CSIC_H */
