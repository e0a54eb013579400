// csic_stream.cpp -- cycle-level model of the reference's Decoupled pixel stream.  Host only.
//
// The reference is an RTL generator: what its users build and simulate is a chain of modules that talk over
// ready/valid ("Decoupled") interfaces, one pixel per handshake, and its tests check that handshake as well as the
// pixels (SpatialDownsamplerSpec.scala:48-58, the cycle budget of ImageCompressorTopApp.scala:110).  The GPU path
// answers "which pixels come out"; this file answers "in which cycle, and what do ready/valid do meanwhile" -- it is
// a SIMULATOR of the generated hardware's interface behaviour, one clock edge per csic_stream_step, a few Mpixel/s on a
// host core.  It is NOT a compute path of the library: nothing in csic_process_* / csic_plan_* / csic_frame_graph_*
// reaches this file (there is no CPU fallback for them), and the pixel arithmetic here exists only because the bits
// travel with the handshake.  RTL / FIRRTL emission needs Chisel and stays out of scope.
//
// Modules (citations relative to /root/reference/src/main/scala/jpeg/):
//   register slice   RGB2YCbCr.scala:67-91, ChromaSubsampler.scala:29-68, ColorQuantizer.scala:22-54:
//                    one output register + valid bit; in.ready = !valid || out.ready; on in.fire the register loads
//                    f(in.bits) and valid := 1, else on out.fire valid := 0.
//   SpatialDownsampler.scala:17-55: combinational pass/drop.  out.valid = in.valid && doSample, out.bits = in.bits,
//                    in.ready = doSample ? out.ready : true; col/row counters advance on in.fire.  sof / eol are inputs
//                    that no logic reads (:11-12) -- accepted and ignored here too.
//   Queue(gen, 1)    chisel3.util.Queue with pipe = false, flow = false (ImageCompressorTop.scala:63-65): one entry,
//                    enq.ready = !full, deq.valid = full; it can never enqueue and dequeue in the same cycle, so each
//                    queue passes at most one pixel every two cycles.
//   ImageCompressorTop.scala:80-114: in -> RGB2YCbCr -> Queue -> op1 -> Queue -> op2 -> Queue -> op3 -> out, with chroma and
//                    spatial built for the FULL width/height whatever their position (:44, :52-58).
//   ImageProcessor.scala:42-62: in -> RGB2YCbCr -> ChromaSubsampler -> SpatialDownsampler -> out, no queues.
#include "csic_internal.h"

#include <cstring>
#include <new>
#include <vector>

namespace {

using csic::Geometry;

enum NodeKind { N_RGB, N_CHROMA, N_QUANT, N_SPATIAL, N_QUEUE };

struct Node {
    NodeKind kind;
    bool valid = false;          // register slices: valid_reg; queue: maybe_full
    uint32_t bits = 0;           // the output register / the queue's one entry (Y | Cb << 8 | Cr << 16)
    // ChromaSubsampler.scala:34-38
    uint8_t last_cb = 0, last_cr = 0;
    int32_t px = 0, line = 0;
    // SpatialDownsampler.scala:17-18
    int32_t col = 0, row = 0;
};

inline uint32_t clamp255(int32_t v) { return v < 0 ? 0u : (v > 255 ? 255u : (uint32_t)v); }

// RGB2YCbCr.scala:33-35,55-65 (arithmetic shift = floor), clamps :37-47
inline uint32_t rgb2ycbcr_floor(uint32_t argb)
{
    const int32_t r = (argb >> 16) & 255, g = (argb >> 8) & 255, b = argb & 255;
    const int32_t y = (77 * r + 150 * g + 29 * b + 128) >> 8;
    const int32_t cb = ((-43 * r - 85 * g + 128 * b + 128) >> 8) + 128;
    const int32_t cr = ((128 * r - 107 * g - 21 * b + 128) >> 8) + 128;
    return clamp255(y) | (clamp255(cb) << 8) | (clamp255(cr) << 16);
}

// YCbCr2RGB.scala:17-26 -- what the harness applies to every pixel it collects (ImageCompressorTopApp.scala:118)
inline uint32_t ycbcr2rgb(uint32_t ycc)
{
    const int32_t c = ycc & 255, d = (int32_t)((ycc >> 8) & 255) - 128, e = (int32_t)((ycc >> 16) & 255) - 128;
    const uint32_t r = clamp255((298 * c + 409 * e + 128) >> 8);
    const uint32_t g = clamp255((298 * c - 100 * d - 208 * e + 128) >> 8);
    const uint32_t b = clamp255((298 * c + 516 * d + 128) >> 8);
    return 0xFF000000u | (r << 16) | (g << 8) | b;
}

} // namespace

struct csic_stream {
    csic_params p;
    Geometry g;
    int32_t kind;
    std::vector<Node> nodes;
    int64_t cycles = 0;
    bool in_is_rgb = true;       // interface 0 carries a PixelBundle (ARGB) -- else a PixelYCbCrBundle
    bool inverse_out = false;    // out_bits = ycbcr2rgb(io.out) (params.out_format == ARGB): the harness-side inverse
};

namespace {

constexpr int MAX_IF = 9;        // interfaces of the longest chain (7 nodes)

struct Wires {
    bool v[MAX_IF], r[MAX_IF];
    uint32_t b[MAX_IF];
};

inline bool do_sample(const csic_stream *s, const Node &n)
{
    const int32_t m = s->g.f - 1;                                  // SpatialDownsampler.scala:33-45: low bits of both counters zero
    return ((n.col & m) == 0) && ((n.row & m) == 0);
}

// the combinational network for the present state and inputs (no clock edge)
void eval(const csic_stream *s, const csic_stream_in *in, Wires &w)
{
    const int n = (int)s->nodes.size();
    w.v[0] = in->in_valid != 0;
    w.b[0] = in->in_bits;
    for (int k = 0; k < n; ++k) {
        const Node &nd = s->nodes[k];
        if (nd.kind == N_SPATIAL) {
            w.v[k + 1] = w.v[k] && do_sample(s, nd);               // :47
            w.b[k + 1] = w.b[k];                                   // :55
        } else {
            w.v[k + 1] = nd.valid;
            w.b[k + 1] = nd.bits;
        }
    }
    w.r[n] = in->out_ready != 0;
    for (int k = n - 1; k >= 0; --k) {
        const Node &nd = s->nodes[k];
        switch (nd.kind) {
        case N_SPATIAL: w.r[k] = do_sample(s, nd) ? w.r[k + 1] : true; break;   // :49-53
        case N_QUEUE:   w.r[k] = !nd.valid; break;                              // enq.ready = !full
        default:        w.r[k] = !nd.valid || w.r[k + 1]; break;                // RGB2YCbCr.scala:73 etc.
        }
    }
}

void clock_edge(csic_stream *s, const Wires &w)
{
    const int n = (int)s->nodes.size();
    for (int k = 0; k < n; ++k) {
        Node &nd = s->nodes[k];
        const bool fire_in = w.v[k] && w.r[k], fire_out = w.v[k + 1] && w.r[k + 1];
        switch (nd.kind) {
        case N_RGB:
            if (fire_in) { nd.bits = rgb2ycbcr_floor(w.b[k]); nd.valid = true; }      // RGB2YCbCr.scala:78-82
            else if (fire_out) nd.valid = false;                                      // :83-84
            break;
        case N_QUANT:
            if (fire_in) {                                                            // ColorQuantizer.scala:35-46
                nd.bits = (w.b[k] & 0xFF & s->g.mask_y) | (w.b[k] & ((uint32_t)s->g.mask_cb << 8)) | (w.b[k] & ((uint32_t)s->g.mask_cr << 16));
                nd.valid = true;
            } else if (fire_out) nd.valid = false;
            break;
        case N_CHROMA:
            if (fire_in) {                                                            // ChromaSubsampler.scala:47-65
                const uint32_t y = w.b[k] & 255, cb = (w.b[k] >> 8) & 255, cr = (w.b[k] >> 16) & 255;
                const bool sample = (nd.px % s->g.h == 0) && (nd.line % s->g.v == 0); // :52-55, counters BEFORE this fire
                if (sample) { nd.last_cb = (uint8_t)cb; nd.last_cr = (uint8_t)cr; }
                nd.bits = y | ((uint32_t)nd.last_cb << 8) | ((uint32_t)nd.last_cr << 16);
                nd.valid = true;
                if (++nd.px == s->g.W) {                                              // Counter(fire, imageWidth) :37
                    nd.px = 0;
                    if (++nd.line == s->g.H) nd.line = 0;                             // Counter(fire && wrap, imageHeight) :38
                }
            } else if (fire_out) nd.valid = false;
            break;
        case N_SPATIAL:
            if (fire_in) {                                                            // SpatialDownsampler.scala:20-31
                if (nd.col == s->g.W - 1) {
                    nd.col = 0;
                    nd.row = (nd.row == s->g.H - 1) ? 0 : nd.row + 1;
                } else nd.col += 1;
            }
            break;
        case N_QUEUE:
            if (fire_in) nd.bits = w.b[k];                                            // ram(0) := enq.bits
            if (fire_in != fire_out) nd.valid = fire_in;                              // maybe_full := do_enq
            break;
        }
    }
    s->cycles += 1;
}

inline void outputs(const csic_stream *s, const Wires &w, csic_stream_out *out)
{
    const int n = (int)s->nodes.size();
    out->in_ready = w.r[0] ? 1 : 0;
    out->out_valid = w.v[n] ? 1 : 0;
    out->out_bits = s->inverse_out ? ycbcr2rgb(w.b[n]) : (w.b[n] & 0x00FFFFFFu);
}

} // namespace

using namespace csic;

extern "C" {

int csic_stream_create(const csic_params *p, int32_t kind, csic_stream **out)
{
    if (!out) return set_error(CSIC_EINVAL_NULL, "out is NULL");
    *out = nullptr;
    Geometry g;
    int st = derive_geometry(p, &g);                      // every require() of the generators
    if (st != CSIC_OK) return st;
    if (kind < CSIC_STREAM_TOP || kind > CSIC_STREAM_QUANT) return set_error(CSIC_EINVAL_SIZE, "unknown stream-model kind %d", kind);
    if (p->sampling != CSIC_SAMPLING_HOLD_DECIMATE)
        return set_error(CSIC_EINVAL_SAMPLING, "the cycle-level model describes the reference's RTL: HOLD_DECIMATE only");
    const bool has_rgb = kind == CSIC_STREAM_TOP || kind == CSIC_STREAM_PROCESSOR || kind == CSIC_STREAM_RGB2YCBCR;
    if (has_rgb && p->rounding != CSIC_ROUND_FLOOR_HW)
        return set_error(CSIC_EINVAL_ROUNDING, "the RTL's RGB2YCbCr rounds with an arithmetic shift (FLOOR_HW); TRUNC_SW exists only in "
                                               "the spec-local software models");
    csic_stream *s = new (std::nothrow) csic_stream();
    if (!s) return set_error(CSIC_ENOMEM, "out of host memory");
    s->p = *p;
    s->g = g;
    s->kind = kind;
    s->in_is_rgb = has_rgb;
    s->inverse_out = p->out_format == CSIC_FMT_ARGB8888;
    auto add = [&](NodeKind k) { Node n; n.kind = k; s->nodes.push_back(n); };
    auto op_node = [&](int32_t op) { return op == CSIC_OP_SPATIAL ? N_SPATIAL : op == CSIC_OP_QUANT ? N_QUANT : N_CHROMA; };
    try {
        switch (kind) {
        case CSIC_STREAM_TOP:                             // ImageCompressorTop.scala:80-114
            add(N_RGB);
            for (int k = 0; k < 3; ++k) { add(N_QUEUE); add(op_node(p->op[k])); }
            break;
        case CSIC_STREAM_PROCESSOR:                       // ImageProcessor.scala:42-62
            add(N_RGB); add(N_CHROMA); add(N_SPATIAL);
            break;
        case CSIC_STREAM_RGB2YCBCR: add(N_RGB); break;
        case CSIC_STREAM_CHROMA:    add(N_CHROMA); break;
        case CSIC_STREAM_SPATIAL:   add(N_SPATIAL); break;
        default:                    add(N_QUANT); break;
        }
    } catch (const std::bad_alloc &) { delete s; return set_error(CSIC_ENOMEM, "out of host memory"); }
    *out = s;
    clear_error();
    return CSIC_OK;
}

int csic_stream_destroy(csic_stream *s)
{
    delete s;
    return CSIC_OK;
}

int csic_stream_reset(csic_stream *s)
{
    if (!s) return set_error(CSIC_EINVAL_NULL, "stream is NULL");
    for (Node &n : s->nodes) { const NodeKind k = n.kind; n = Node(); n.kind = k; }
    s->cycles = 0;
    clear_error();
    return CSIC_OK;
}

int csic_stream_eval(const csic_stream *s, const csic_stream_in *in, csic_stream_out *out)
{
    if (!s || !in || !out) return set_error(CSIC_EINVAL_NULL, "argument is NULL");
    Wires w;
    eval(s, in, w);
    outputs(s, w, out);
    return CSIC_OK;
}

int csic_stream_step(csic_stream *s, const csic_stream_in *in, csic_stream_out *out)
{
    if (!s || !in) return set_error(CSIC_EINVAL_NULL, "argument is NULL");
    Wires w;
    eval(s, in, w);
    if (out) outputs(s, w, out);
    clock_edge(s, w);
    return CSIC_OK;
}

int64_t csic_stream_cycles(const csic_stream *s) { return s ? s->cycles : (int64_t)set_error(CSIC_EINVAL_NULL, "stream is NULL"); }

int csic_stream_depth(const csic_stream *s) { return s ? (int)s->nodes.size() : set_error(CSIC_EINVAL_NULL, "stream is NULL"); }

int csic_stream_run(csic_stream *s, const uint32_t *in, size_t n_in, uint32_t *out, size_t max_out, int64_t max_cycles,
                    const uint8_t *in_valid_pattern, size_t in_pattern_len, const uint8_t *out_ready_pattern, size_t out_pattern_len,
                    size_t *n_out, int64_t *cycles)
{
    if (!s || (!in && n_in) || (!out && max_out)) return set_error(CSIC_EINVAL_NULL, "argument is NULL");
    size_t fed = 0, got = 0;
    int64_t used = 0;
    if (max_cycles < 0) {
        // run to completion: bounded all the same (a pattern without a single 1 would never drain)
        auto ones = [](const uint8_t *pat, size_t len) { if (!pat || !len) return true; for (size_t i = 0; i < len; ++i) if (pat[i]) return true; return false; };
        if (!ones(in_valid_pattern, in_pattern_len) || !ones(out_ready_pattern, out_pattern_len))
            return set_error(CSIC_EINVAL_SIZE, "a valid / ready pattern without a single 1 never finishes; give max_cycles");
    }
    const int64_t hard_cap = max_cycles >= 0 ? max_cycles : (int64_t)(n_in + 16) * 4 * (int64_t)(in_pattern_len + out_pattern_len + 1);
    Wires w;
    csic_stream_in pin;
    std::memset(&pin, 0, sizeof pin);
    // ImageCompressorTopApp.scala:76-124: the driver holds valid while pixels remain (a pixel stays on the wires until the edge
    // at which in.ready is seen), the collector samples out.valid every cycle with out.ready high and stops at max_out pixels
    // or when its cycle budget is spent.  The patterns (cyclic, may be NULL) add producer gaps and back-pressure.
    while (got < max_out && used < hard_cap) {
        if (fed >= n_in && max_cycles < 0) {
            // nothing left to feed: stop once the pipeline has drained and can never produce again
            bool any = false;
            for (const Node &n : s->nodes) any = any || n.valid;
            if (!any) break;
        }
        const bool offer = fed < n_in && (!in_valid_pattern || in_pattern_len == 0 || in_valid_pattern[(size_t)used % in_pattern_len] != 0);
        pin.in_valid = offer ? 1 : 0;
        pin.in_bits = offer ? in[fed] : 0;
        pin.out_ready = (!out_ready_pattern || out_pattern_len == 0 || out_ready_pattern[(size_t)used % out_pattern_len] != 0) ? 1 : 0;
        pin.sof = (offer && fed == 0) ? 1 : 0;                                        // :79-84 (no logic reads them)
        pin.eol = (offer && (int32_t)(fed % (size_t)s->g.W) == s->g.W - 1) ? 1 : 0;
        eval(s, &pin, w);
        const int n = (int)s->nodes.size();
        if (w.v[n] && w.r[n]) {
            out[got++] = s->inverse_out ? ycbcr2rgb(w.b[n]) : (w.b[n] & 0x00FFFFFFu);
        }
        if (w.v[0] && w.r[0]) fed += 1;
        clock_edge(s, w);
        used += 1;
    }
    if (n_out) *n_out = got;
    if (cycles) *cycles = used;
    clear_error();
    return CSIC_OK;
}

} // extern "C"
