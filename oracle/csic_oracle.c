/*
 * csic_oracle.c -- CPU restatement of the reference pixel pipeline (see csic_oracle.h).
 * TEST INFRASTRUCTURE ONLY; never linked into, loaded by, or called from the product path.
 *
 * Two independent forms of the same semantics live here on purpose:
 *   orc_process_stream  -- explicit per-stage state machines (counters, held chroma),
 *                          wired in op[] order like the RTL top level;
 *   orc_process_closed  -- the per-output-pixel gather of SURVEY.md Appendix A.3/A.4.
 * tests/test_oracle_forms.py requires them to agree on random shapes and all six orders.
 *
 * Citations are relative to /root/reference/.
 */
#include "csic_oracle.h"

#include <pthread.h>
#include <string.h>

/* ------------------------------------------------------------------------- */
/* per-pixel arithmetic                                                       */
/* ------------------------------------------------------------------------- */

static inline int clamp255(int v) { return v < 0 ? 0 : (v > 255 ? 255 : v); }

/* floor(x / 256) without relying on the implementation-defined >> of negatives */
static inline int floor_div256(int x) { return x >= 0 ? (x >> 8) : -((-x + 255) >> 8); }

/* Scala Int '/' == C '/' : truncation toward zero */
static inline int trunc_div256(int x) { return x / 256; }

/* RGB2YCbCr.scala:33-35 (products), :55-65 (bias, floor shift, +128 offset), :37-47 (clamp)
 * ReferenceModel.scala:10-17 (same, '>> 8');  RGB2YCbCr.scala:100-118 (same, '/ 256'). */
void orc_rgb2ycbcr(int r, int g, int b, int rounding, int *y, int *cb, int *cr)
{
    int yi  =  77 * r + 150 * g +  29 * b;
    int cbi = -43 * r -  85 * g + 128 * b;
    int cri = 128 * r - 107 * g -  21 * b;
    if (rounding == ORC_ROUND_TRUNC_SW) {
        *y  = clamp255(trunc_div256(yi  + 128));
        *cb = clamp255(trunc_div256(cbi + 128) + 128);
        *cr = clamp255(trunc_div256(cri + 128) + 128);
    } else {
        *y  = clamp255(floor_div256(yi  + 128));
        *cb = clamp255(floor_div256(cbi + 128) + 128);
        *cr = clamp255(floor_div256(cri + 128) + 128);
    }
}

/* YCbCr2RGB.scala:17-26 == RGB2YCbCr.scala:123-132.  c = y (NOT y-16): the pair is lossy. */
void orc_ycbcr2rgb(int y, int cb, int cr, int *r, int *g, int *b)
{
    int c = y, d = cb - 128, e = cr - 128;
    *r = clamp255(floor_div256(298 * c + 409 * e + 128));
    *g = clamp255(floor_div256(298 * c - 100 * d - 208 * e + 128));
    *b = clamp255(floor_div256(298 * c + 516 * d + 128));
}

/* ColorQuantizer.scala:29-31,42-44 == ColorQuantizerSpec.scala:29-34 */
void orc_quantize(int y, int cb, int cr, int yb, int cbb, int crb, int *yq, int *cbq, int *crq)
{
    int sy = 8 - yb, scb = 8 - cbb, scr = 8 - crb;
    *yq  = (y  >> sy)  << sy;
    *cbq = (cb >> scb) << scb;
    *crq = (cr >> scr) << scr;
}

/* input pixel -> (Y, Cb, Cr): the forward transform, or plain unpacking when the caller feeds
 * a YCbCr stream (how ChromaSubsamplerImageSpec / ColorQuantizerSpec / SpatialDownsamplerSpec
 * drive one stage at a time) */
static inline void decode_px(const orc_params *p, uint32_t px, int *y, int *cb, int *cr)
{
    if (p->in_format == ORC_FMT_YCC) { *y = px & 0xFF; *cb = (px >> 8) & 0xFF; *cr = (px >> 16) & 0xFF; }
    else orc_rgb2ycbcr((px >> 16) & 0xFF, (px >> 8) & 0xFF, px & 0xFF, p->rounding, y, cb, cr);
}

/* ------------------------------------------------------------------------- */
/* validation                                                                  */
/* ------------------------------------------------------------------------- */

/* ImageProcessor.scala:22-28, ChromaSubsampler.scala:13-18, SpatialDownsampler.scala:7-8,
 * ColorQuantizer.scala:12-15, ImageCompressorTop.scala:27-31.  (The divisibility rule of
 * ImageProcessor.scala:25 is a property of ImageProcessorParams only and is checked by the
 * host mirror, not here.) */
int orc_validate(const orc_params *p)
{
    if (!p) return -1;
    if (p->width <= 0 || p->height <= 0) return -2;
    if (!(p->factor == 1 || p->factor == 2 || p->factor == 4 || p->factor == 8)) return -3;
    if (!(p->chroma_a == 4 || p->chroma_a == 2 || p->chroma_a == 1)) return -4;
    if (!(p->chroma_b == p->chroma_a || p->chroma_b == 0)) return -5;
    if (p->y_bits < 1 || p->y_bits > 8 || p->cb_bits < 1 || p->cb_bits > 8 ||
        p->cr_bits < 1 || p->cr_bits > 8) return -6;
    int seen[4] = {0, 0, 0, 0};
    for (int k = 0; k < 3; ++k) {
        if (p->op[k] < 1 || p->op[k] > 3) return -7;
        if (seen[p->op[k]]++) return -7;
    }
    if (p->rounding != ORC_ROUND_FLOOR_HW && p->rounding != ORC_ROUND_TRUNC_SW) return -8;
    if (p->out_format != ORC_FMT_ARGB && p->out_format != ORC_FMT_YCC) return -9;
    if (p->in_format != ORC_FMT_ARGB && p->in_format != ORC_FMT_YCC) return -9;
    return 0;
}

/* SpatialDownsampler emits one pixel per (row % f == 0, col % f == 0):
 * ceil(W/f) x ceil(H/f); KAT 5x3,f=2 -> 6 pixels (SpatialDownsamplerSpec.scala:120-122). */
void orc_out_dims(const orc_params *p, int32_t *wo, int32_t *ho)
{
    *wo = (p->width  + p->factor - 1) / p->factor;
    *ho = (p->height + p->factor - 1) / p->factor;
}

/* ------------------------------------------------------------------------- */
/* streaming form: explicit stage state machines                              */
/* ------------------------------------------------------------------------- */

typedef struct { int y, cb, cr; } ycc_t;

typedef struct {             /* ChromaSubsampler.scala:26-27,34-38 */
    int W, H, h, v;
    int pixel_counter, line_counter;
    int last_cb, last_cr;    /* RegInit(0.U) */
} chroma_state;

typedef struct {             /* SpatialDownsampler.scala:17-18 */
    int W, H, f;
    int col, row;
} spatial_state;

static void chroma_init(chroma_state *s, int W, int H, int a, int b)
{
    s->W = W; s->H = H;
    s->h = 4 / a;                          /* ChromaSubsampler.scala:26 */
    s->v = (b == 0 && a != 0) ? 2 : 1;     /* ChromaSubsampler.scala:27 */
    s->pixel_counter = 0; s->line_counter = 0;
    s->last_cb = 0; s->last_cr = 0;
}

/* one fire of ChromaSubsampler.scala:47-65 (+ Counter wrap :37-38) */
static ycc_t chroma_step(chroma_state *s, ycc_t in)
{
    ycc_t out;
    out.y = in.y;                                            /* :48 */
    int sample_h = (s->pixel_counter % s->h) == 0;          /* :52 */
    int sample_v = (s->line_counter  % s->v) == 0;          /* :53 */
    if (sample_h && sample_v) {                              /* :57-61 */
        out.cb = in.cb; out.cr = in.cr;
        s->last_cb = in.cb; s->last_cr = in.cr;
    } else {                                                 /* :62-65 */
        out.cb = s->last_cb; out.cr = s->last_cr;
    }
    if (++s->pixel_counter == s->W) {                        /* Counter(fire, imageWidth) */
        s->pixel_counter = 0;
        if (++s->line_counter == s->H) s->line_counter = 0;  /* Counter(fire && wrap, imageHeight) */
    }
    return out;
}

static void spatial_init(spatial_state *s, int W, int H, int f)
{
    s->W = W; s->H = H; s->f = f; s->col = 0; s->row = 0;
}

/* one fire of SpatialDownsampler.scala:20-55; returns 1 if the pixel is passed on */
static int spatial_step(spatial_state *s)
{
    int m = s->f - 1;
    int do_sample = ((s->col & m) == 0) && ((s->row & m) == 0);  /* :33-45 */
    if (s->col == s->W - 1) {                                     /* :21-31 */
        s->col = 0;
        s->row = (s->row == s->H - 1) ? 0 : s->row + 1;
    } else {
        s->col += 1;
    }
    return do_sample;
}

static inline uint32_t pack_out(const orc_params *p, ycc_t v)
{
    if (p->out_format == ORC_FMT_YCC)
        return (uint32_t)v.y | ((uint32_t)v.cb << 8) | ((uint32_t)v.cr << 16);
    int r, g, b;
    orc_ycbcr2rgb(v.y, v.cb, v.cr, &r, &g, &b);   /* ImageCompressorTopApp.scala:118 */
    return 0xFF000000u | ((uint32_t)r << 16) | ((uint32_t)g << 8) | (uint32_t)b; /* :139 */
}

long orc_process_stream(const orc_params *p, const uint32_t *in, uint32_t *out)
{
    if (orc_validate(p) != 0) return -1;
    const int W = p->width, H = p->height;
    chroma_state  cs; chroma_init(&cs, W, H, p->chroma_a, p->chroma_b);   /* full W,H: ImageCompressorTop.scala:52-58 */
    spatial_state ss; spatial_init(&ss, W, H, p->factor);                 /* full W,H: ImageCompressorTop.scala:44 */
    long n_out = 0;
    const long n_in = (long)W * H;
    for (long i = 0; i < n_in; ++i) {
        ycc_t v;                                                           /* alpha ignored: ImageProcessorModel.scala:48 */
        decode_px(p, in[i], &v.y, &v.cb, &v.cr);                           /* toYC, ImageCompressorTop.scala:80-81 */
        int alive = 1;
        for (int k = 0; k < 3 && alive; ++k) {                             /* op1 -> op2 -> op3, :83-114 */
            switch (p->op[k]) {
            case ORC_OP_SPATIAL: alive = spatial_step(&ss); break;
            case ORC_OP_QUANT:
                orc_quantize(v.y, v.cb, v.cr, p->y_bits, p->cb_bits, p->cr_bits, &v.y, &v.cb, &v.cr);
                break;
            case ORC_OP_CHROMA:  v = chroma_step(&cs, v); break;
            default: return -1;
            }
        }
        if (alive) out[n_out++] = pack_out(p, v);
    }
    return n_out;
}

/* ------------------------------------------------------------------------- */
/* closed form                                                                 */
/* ------------------------------------------------------------------------- */

static int spatial_before_chroma(const orc_params *p)
{
    int is = -1, ic = -1;
    for (int k = 0; k < 3; ++k) {
        if (p->op[k] == ORC_OP_SPATIAL) is = k;
        if (p->op[k] == ORC_OP_CHROMA)  ic = k;
    }
    return is < ic;
}

long orc_process_closed_rows(const orc_params *p, const uint32_t *in, uint32_t *out,
                             int32_t ro0, int32_t ro1)
{
    if (orc_validate(p) != 0) return -1;
    const long W = p->width;
    const int f = p->factor;
    const int h = 4 / p->chroma_a;
    const int v = (p->chroma_b == 0) ? 2 : 1;
    int32_t wo, ho; orc_out_dims(p, &wo, &ho);
    if (ro0 < 0 || ro1 > ho || ro0 > ro1) return -1;
    const int s_first = spatial_before_chroma(p);
    long n = 0;
    for (long ro = ro0; ro < ro1; ++ro) {
        for (long co = 0; co < wo; ++co) {
            long y_idx = (ro * f) * W + co * f;       /* pixel that supplies Y */
            long c_idx;                               /* pixel that supplies Cb,Cr */
            if (!s_first) {
                /* chroma runs on the full raster: Appendix A.3 with Wm = W on image coordinates */
                long r = ro * f, c = co * f;
                if (r % v == 0) c_idx = r * W + (c - c % h);
                else            c_idx = (r - 1) * W + ((W - 1) / h) * h;
            } else {
                /* chroma runs on the decimated stream but still wraps its column counter at the
                 * full width W (ImageCompressorTop.scala:52-58): Appendix A.4 */
                long j = ro * wo + co;
                long c = j % W, r = j / W;
                long src = (r % v == 0) ? (j - c % h) : ((r - 1) * W + ((W - 1) / h) * h);
                long sro = src / wo, sco = src % wo;
                c_idx = (sro * f) * W + sco * f;
            }
            uint32_t py = in[y_idx], pc = in[c_idx];
            ycc_t a, c2;
            decode_px(p, py, &a.y, &a.cb, &a.cr);
            decode_px(p, pc, &c2.y, &c2.cb, &c2.cr);
            a.cb = c2.cb; a.cr = c2.cr;
            orc_quantize(a.y, a.cb, a.cr, p->y_bits, p->cb_bits, p->cr_bits, &a.y, &a.cb, &a.cr);
            out[ro * wo + co] = pack_out(p, a);
            ++n;
        }
    }
    return n;
}

long orc_process_closed(const orc_params *p, const uint32_t *in, uint32_t *out)
{
    if (orc_validate(p) != 0) return -1;
    int32_t wo, ho; orc_out_dims(p, &wo, &ho);
    return orc_process_closed_rows(p, in, out, 0, ho);
}

/* ------------------------------------------------------------------------- */
/* AVG sampling extension (NOT reference semantics)                            */
/* ------------------------------------------------------------------------- */
/* The reference's chroma stage is sample-and-hold and its spatial stage is decimation
 * (ChromaSubsampler.scala:47-65, SpatialDownsampler.scala:33-55); its README and the project's
 * north star describe box-filter averaging instead.  This is the build-defined integer
 * specification of that variant -- there is nothing in the reference to be bit-exact against
 * ("parity unpinned"); this function is the normative statement the HIP kernels are tested on.
 *   order is fixed: forward -> chroma average -> spatial average -> quantise -> (inverse)
 *   chroma: block (r - r % v, c - c % h) of h x v pixels, coordinates clamped to the image,
 *           Cb' = (sum + n/2) >> log2(n), n = h*v; same for Cr; Y passes through
 *   spatial: output (ro, co) = per channel (sum over the f x f block at (ro*f, co*f), coordinates
 *           clamped, of the chroma-stage output + f*f/2) >> (2 log2 f)
 * Stages exchange 8-bit values, like the streaming pipeline would. */
static int ilog2(int x) { int l = 0; while ((1 << l) < x) ++l; return l; }

long orc_process_avg(const orc_params *p, const uint32_t *in, uint32_t *out)
{
    if (orc_validate(p) != 0) return -1;
    if (!(p->op[0] == ORC_OP_CHROMA && p->op[1] == ORC_OP_SPATIAL && p->op[2] == ORC_OP_QUANT)) return -1;
    const long W = p->width, H = p->height;
    const int f = p->factor, h = 4 / p->chroma_a, v = (p->chroma_b == 0) ? 2 : 1;
    const int nlog = ilog2(h * v), flog2 = 2 * ilog2(f);
    int32_t wo, ho; orc_out_dims(p, &wo, &ho);
    long n = 0;
    for (long ro = 0; ro < ho; ++ro) {
        for (long co = 0; co < wo; ++co) {
            int sy = 0, scb = 0, scr = 0;
            for (int i = 0; i < f; ++i) {
                for (int j = 0; j < f; ++j) {
                    long r = ro * f + i, c = co * f + j;
                    if (r > H - 1) r = H - 1;
                    if (c > W - 1) c = W - 1;
                    /* chroma-stage output at (r, c) */
                    int y, cb, cr;
                    decode_px(p, in[r * W + c], &y, &cb, &cr);
                    long r0 = r - r % v, c0 = c - c % h;
                    int acb = 0, acr = 0;
                    for (int ii = 0; ii < v; ++ii) {
                        for (int jj = 0; jj < h; ++jj) {
                            long rr = r0 + ii, cc = c0 + jj;
                            if (rr > H - 1) rr = H - 1;
                            if (cc > W - 1) cc = W - 1;
                            int y2, cb2, cr2;
                            decode_px(p, in[rr * W + cc], &y2, &cb2, &cr2);
                            acb += cb2; acr += cr2;
                        }
                    }
                    sy += y;
                    scb += (acb + ((h * v) >> 1)) >> nlog;
                    scr += (acr + ((h * v) >> 1)) >> nlog;
                }
            }
            ycc_t o;
            o.y  = (sy  + ((f * f) >> 1)) >> flog2;
            o.cb = (scb + ((f * f) >> 1)) >> flog2;
            o.cr = (scr + ((f * f) >> 1)) >> flog2;
            orc_quantize(o.y, o.cb, o.cr, p->y_bits, p->cb_bits, p->cr_bits, &o.y, &o.cb, &o.cr);
            out[ro * wo + co] = pack_out(p, o);
            ++n;
        }
    }
    return n;
}

/* Row-parallel closed form on `nthreads` POSIX threads (output rows are independent): the
 * "all host cores" CPU baseline of BASELINE.md section 2.  Same results as orc_process_closed. */
typedef struct { const orc_params *p; const uint32_t *in; uint32_t *out; int32_t ro0, ro1; long n; } mt_job;

static void *mt_worker(void *arg)
{
    mt_job *j = (mt_job *)arg;
    j->n = orc_process_closed_rows(j->p, j->in, j->out, j->ro0, j->ro1);
    return NULL;
}

long orc_process_closed_mt(const orc_params *p, const uint32_t *in, uint32_t *out, int nthreads)
{
    if (orc_validate(p) != 0 || nthreads < 1) return -1;
    if (nthreads > 1024) nthreads = 1024;
    int32_t wo, ho; orc_out_dims(p, &wo, &ho);
    pthread_t tid[1024];
    mt_job job[1024];
    int started = 0;
    for (int t = 0; t < nthreads; ++t) {
        job[t].p = p; job[t].in = in; job[t].out = out; job[t].n = 0;
        job[t].ro0 = (int32_t)((long)ho * t / nthreads);
        job[t].ro1 = (int32_t)((long)ho * (t + 1) / nthreads);
        if (pthread_create(&tid[t], NULL, mt_worker, &job[t]) != 0) break;
        ++started;
    }
    long n = 0;
    for (int t = 0; t < started; ++t) { pthread_join(tid[t], NULL); n += job[t].n; }
    for (int t = started; t < nthreads; ++t) { mt_worker(&job[t]); n += job[t].n; }   /* thread creation failed: run inline */
    return n;
}

/* ------------------------------------------------------------------------- */
/* planar, subsampled form of the output stream (include/csic.h, CSIC_FMT_PLANAR) */
/* ------------------------------------------------------------------------- */
/* The reference never builds this format (ChromaSubsampler.scala:57-65 re-emits the held chroma with every
 * pixel; README.md:35-46 only describes it).  What pins it to the reference all the same: the planes hold
 * nothing but values of the reference's OWN output stream -- Y at every position, Cb / Cr at the positions
 * where the chroma stage, as seen from the output, is at a sample point -- and orc_planar_reconstruct below is
 * ChromaSubsampler's latch (ChromaSubsampler.scala:29-65) replayed over them, so
 *     orc_planar_reconstruct(orc_planar_from_stream(s)) == s
 * for every stream s that orc_process_stream emits (tests/test_oracle_planar.py checks exactly that, for all
 * six orders).  Which counters the output sees:
 *   chroma before spatial, f = 1 : the image's own: row length W, holds h x v
 *   chroma before spatial, f > 1 : the decimator keeps image columns co * f of rows ro * f: every kept row is a
 *                                  sample row (f is even, v <= 2) and kept column co * f is a sample point iff
 *                                  (co * f) % h == 0, i.e. co % max(1, h / f) == 0: row length Wo, holds max(1, h/f) x 1
 *   spatial before chroma        : the stage runs on the decimated stream with its counters wrapping at the
 *                                  FULL width (ImageCompressorTop.scala:52-58): row length W, holds h x v
 *   AVG extension                : chroma blocks h x v pooled f x f: constant over max(1,h/f) x max(1,v/f) output pixels */
int orc_planar_layout_of(const orc_params *p, int avg, orc_planar_layout *L)
{
    if (orc_validate(p) != 0) return -1;
    int32_t wo, ho; orc_out_dims(p, &wo, &ho);
    const int h = 4 / p->chroma_a, v = (p->chroma_b == 0) ? 2 : 1, f = p->factor;
    L->y_width = wo; L->y_height = ho;
    if (avg) {
        L->module_width = wo; L->hold_h = h > f ? h / f : 1; L->hold_v = v > f ? v / f : 1; L->replay_last = 0;
    } else if (f == 1 || spatial_before_chroma(p)) {
        L->module_width = p->width; L->hold_h = h; L->hold_v = v; L->replay_last = 1;
    } else {
        L->module_width = wo; L->hold_h = h > f ? h / f : 1; L->hold_v = 1; L->replay_last = 1;
    }
    const int64_t n = (int64_t)wo * ho;
    const int64_t rows = (n + L->module_width - 1) / L->module_width;
    L->chroma_width = (L->module_width + L->hold_h - 1) / L->hold_h;
    L->chroma_height = (int32_t)((rows + L->hold_v - 1) / L->hold_v);
    int64_t cnt = 0;                       /* counted, not computed: one per sample point of the stream */
    for (int64_t r = 0; r < rows; r += L->hold_v) {
        const int64_t len = (r == rows - 1) ? n - r * L->module_width : L->module_width;
        cnt += (len + L->hold_h - 1) / L->hold_h;
    }
    L->chroma_samples = cnt;
    return 0;
}

/* ycc: the n = y_width * y_height packed output pixels (Y | Cb << 8 | Cr << 16) in stream order.  Samples are
 * appended in the order the stream meets them.  Returns the number of chroma samples written. */
long orc_planar_from_stream(const orc_planar_layout *L, const uint32_t *ycc, uint8_t *y, uint8_t *cb, uint8_t *cr)
{
    const int64_t n = (int64_t)L->y_width * L->y_height;
    int64_t c = 0, r = 0, k = 0;          /* the chroma stage's column / row counters over the output stream */
    for (int64_t j = 0; j < n; ++j) {
        y[j] = (uint8_t)(ycc[j] & 0xFF);
        if (c % L->hold_h == 0 && r % L->hold_v == 0) {
            cb[k] = (uint8_t)((ycc[j] >> 8) & 0xFF);
            cr[k] = (uint8_t)((ycc[j] >> 16) & 0xFF);
            ++k;
        }
        if (++c == L->module_width) { c = 0; ++r; }
    }
    return (long)k;
}

/* planes -> packed stream.  replay_last: ChromaSubsampler's latch -- a sample point loads the next sample, every
 * other position re-emits the latched one (so a row without sample points replays the last sample of the row
 * above, ChromaSubsampler.scala:52-65).  Otherwise (AVG): each sample covers its hold_h x hold_v box. */
long orc_planar_reconstruct(const orc_planar_layout *L, const uint8_t *y, const uint8_t *cb, const uint8_t *cr, int fmt, uint32_t *out)
{
    const int64_t n = (int64_t)L->y_width * L->y_height;
    int64_t c = 0, r = 0, k = 0;
    int last_cb = 0, last_cr = 0;         /* RegInit(0.U), ChromaSubsampler.scala:34-35 */
    for (int64_t j = 0; j < n; ++j) {
        int vcb, vcr;
        if (L->replay_last) {
            if (c % L->hold_h == 0 && r % L->hold_v == 0) { last_cb = cb[k]; last_cr = cr[k]; ++k; }
            vcb = last_cb; vcr = last_cr;
        } else {
            const int64_t kk = (r / L->hold_v) * L->chroma_width + c / L->hold_h;
            vcb = cb[kk]; vcr = cr[kk];
        }
        if (fmt == ORC_FMT_YCC) {
            out[j] = (uint32_t)y[j] | ((uint32_t)vcb << 8) | ((uint32_t)vcr << 16);
        } else {
            int rr, gg, bb;
            orc_ycbcr2rgb(y[j], vcb, vcr, &rr, &gg, &bb);
            out[j] = 0xFF000000u | ((uint32_t)rr << 16) | ((uint32_t)gg << 8) | (uint32_t)bb;
        }
        if (++c == L->module_width) { c = 0; ++r; }
    }
    return (long)n;
}

/* ------------------------------------------------------------------------- */
/* per-stage helpers for the KAT tests                                         */
/* ------------------------------------------------------------------------- */

void orc_chroma_stream(const uint8_t *ycc_in, uint8_t *ycc_out, long n, int W, int H, int a, int b)
{
    chroma_state cs; chroma_init(&cs, W, H, a, b);
    for (long i = 0; i < n; ++i) {
        ycc_t v = { ycc_in[3 * i], ycc_in[3 * i + 1], ycc_in[3 * i + 2] };
        v = chroma_step(&cs, v);
        ycc_out[3 * i] = (uint8_t)v.y; ycc_out[3 * i + 1] = (uint8_t)v.cb; ycc_out[3 * i + 2] = (uint8_t)v.cr;
    }
}

long orc_spatial_indices(int W, int H, int f, int64_t *idx_out)
{
    spatial_state ss; spatial_init(&ss, W, H, f);
    long n = 0;
    for (long i = 0; i < (long)W * H; ++i)
        if (spatial_step(&ss)) idx_out[n++] = i;
    return n;
}

/* ------------------------------------------------------------------------- */
/* synthetic frames                                                            */
/* ------------------------------------------------------------------------- */

static inline uint32_t fmix32(uint32_t h)
{
    h ^= h >> 16; h *= 0x85ebca6bu; h ^= h >> 13; h *= 0xc2b2ae35u; h ^= h >> 16;
    return h;
}

void orc_synth_frame(uint32_t *dst, int64_t npix, int64_t first_index, uint32_t seed)
{
    const uint32_t salt = seed * 0x9E3779B9u;
    for (int64_t i = 0; i < npix; ++i)
        dst[i] = 0xFF000000u | (fmix32((uint32_t)(first_index + i) + salt) & 0x00FFFFFFu);
}
