// csic.hpp -- header-only C++17 host layer over the C ABI of csic.h, keeping the reference's generator
// names and argument lists (Scala originals under /root/reference/src/main/scala/jpeg/):
//   csic::ImageProcessorParams  <- case class ImageProcessorParams           ImageProcessor.scala:15-29
//   csic::ProcessingStep        <- object ProcessingStep extends ChiselEnum   ImageCompressorTop.scala:7-9
//   csic::ImageCompressorTop    <- class ImageCompressorTop(11 parameters)    ImageCompressorTop.scala:11-25
//   csic::ImageProcessor        <- class ImageProcessor(p)                    ImageProcessor.scala:31-63
// Every require() of the reference surfaces as csic::IllegalArgumentException thrown from the
// constructor, device failures as csic::RuntimeError.  No compute happens on the host.
#pragma once

#include <cstddef>
#include <cstdint>
#include <cstring>
#include <stdexcept>
#include <string>
#include <utility>
#include <vector>

#include "csic.h"

namespace csic {

struct IllegalArgumentException : std::invalid_argument {
    int status;
    IllegalArgumentException(int st, const std::string &m) : std::invalid_argument("requirement failed: " + m), status(st) {}
};

struct RuntimeError : std::runtime_error {
    int status;
    RuntimeError(int st, const std::string &m) : std::runtime_error(m), status(st) {}
};

inline int check(int status)
{
    if (status >= 0) return status;
    std::string msg = csic_last_error();
    if (msg.empty()) msg = csic_strerror(status);
    if (status <= CSIC_EINVAL_NULL && status >= CSIC_EINVAL_SIZE) throw IllegalArgumentException(status, msg);
    throw RuntimeError(status, msg);   // device, memory and file errors
}

enum class ProcessingStep : int32_t { NoOp = 0, SpatialSampling = 1, ColorQuantization = 2, ChromaSubsampling = 3 };
enum class Rounding : int32_t { FLOOR_HW = CSIC_ROUND_FLOOR_HW, TRUNC_SW = CSIC_ROUND_TRUNC_SW };
enum class PixelFormat : int32_t { ARGB8888 = CSIC_FMT_ARGB8888, YCBCR888X = CSIC_FMT_YCBCR888X, PLANAR = CSIC_FMT_PLANAR };

struct ImageProcessorParams {
    int width, height, factor, chromaParamA, chromaParamB;
    ImageProcessorParams(int width_, int height_, int factor_, int chromaParamA_, int chromaParamB_)
        : width(width_), height(height_), factor(factor_), chromaParamA(chromaParamA_), chromaParamB(chromaParamB_)
    {
        csic_params p = c_params();
        check(csic_validate(&p));      // the five require()s of ImageProcessor.scala:22-28
    }
    csic_params c_params(Rounding r = Rounding::FLOOR_HW, PixelFormat f = PixelFormat::ARGB8888) const
    {
        csic_params p;
        csic_params_default(&p, width, height);
        p.chroma_a = chromaParamA; p.chroma_b = chromaParamB; p.factor = factor;
        p.rounding = (int32_t)r; p.out_format = (int32_t)f; p.strict_divisible = 1;
        return p;
    }
};

// Stand-in for scrimage's image objects: width, height, packed ARGB ints (0xAARRGGBB).
struct Image {
    int width = 0, height = 0;
    std::vector<uint32_t> argb;
};

// object ImageProcessorModel, src/test/scala/jpeg/ImageProcessorModel.scala:9-53 -- I/O helpers, no arithmetic.
struct ImageProcessorModel {
    static Image readImage(const std::string &file)                                  // :14-16
    {
        Image im;
        int32_t w = 0, h = 0;
        check(csic_png_info(file.c_str(), &w, &h));
        im.width = w; im.height = h;
        im.argb.resize((size_t)w * h);
        check(csic_png_read_argb(file.c_str(), im.argb.data(), im.argb.size()));
        return im;
    }
    static void writeImage(const Image &im, const std::string &file, int compression = 6)   // :18-22
    {
        check(csic_png_write_argb(file.c_str(), im.argb.data(), im.width, im.height, compression));
    }
    static ImageProcessorParams getImageParams(const Image &im, int numPixelsPerCycle)      // :33-41
    {
        return ImageProcessorParams(im.width, im.height, numPixelsPerCycle, 4, 4);
    }
};

// One frame in the subsampled planar format (CSIC_FMT_PLANAR, csic.h: csic_planar_layout): the frame buffer as the device wrote
// it, and views of its three planes -- Y: y_width x y_height bytes, Cb / Cr: chroma_samples bytes each in sample order.
struct PlanarFrame {
    csic_planar_layout layout{};
    std::vector<uint8_t> bytes;                                   // layout.frame_bytes long
    const uint8_t *y() const { return bytes.data() + layout.y_offset; }
    const uint8_t *cb() const { return bytes.data() + layout.cb_offset; }
    const uint8_t *cr() const { return bytes.data() + layout.cr_offset; }
};

class ImageCompressorTop {
public:
    ImageCompressorTop(int width, int height, int chroma_param_a_config, int chroma_param_b_config,
                       int yTargetQuantBitsConfig, int cbTargetQuantBitsConfig, int crTargetQuantBitsConfig,
                       int downFactorConfig, ProcessingStep op1Type, ProcessingStep op2Type, ProcessingStep op3Type,
                       Rounding rounding = Rounding::FLOOR_HW, int device = 0)
        : device_(device)
    {
        csic_params_default(&params_, width, height);
        params_.chroma_a = chroma_param_a_config; params_.chroma_b = chroma_param_b_config;
        params_.y_bits = yTargetQuantBitsConfig; params_.cb_bits = cbTargetQuantBitsConfig; params_.cr_bits = crTargetQuantBitsConfig;
        params_.factor = downFactorConfig;
        params_.op[0] = (int32_t)op1Type; params_.op[1] = (int32_t)op2Type; params_.op[2] = (int32_t)op3Type;
        params_.rounding = (int32_t)rounding;
        check(csic_validate(&params_));                       // construction-time require()s
        check(csic_out_dims(&params_, &out_w_, &out_h_));
    }
    ImageCompressorTop(const ImageCompressorTop &) = delete;
    ImageCompressorTop &operator=(const ImageCompressorTop &) = delete;
    virtual ~ImageCompressorTop()
    {
        for (csic_plan *pl : plan_) csic_plan_destroy(pl);
    }

    int outWidth() const { return out_w_; }
    int outHeight() const { return out_h_; }

    // ARGB frame in -> reconstructed ARGB frame out (the DUT output put through YCbCrUtils.ycbcr2rgb,
    // ImageCompressorTopApp.scala:118)
    std::vector<uint32_t> process(const std::vector<uint32_t> &argb) { return run(PixelFormat::ARGB8888, argb); }
    // ARGB frame in -> io.out's PixelYCbCrBundle stream, packed Y | Cb << 8 | Cr << 16
    std::vector<uint32_t> processYCbCr(const std::vector<uint32_t> &argb) { return run(PixelFormat::YCBCR888X, argb); }
    // device-resident, asynchronous on `hip_stream`
    void processDevice(const void *d_in, void *d_out, void *hip_stream, PixelFormat f = PixelFormat::ARGB8888)
    {
        check(csic_process_device(plan(f), d_in, d_out, hip_stream));
    }
    // The subsampled wire format the reference's README describes and its code never builds (README.md:35-46,
    // ChromaSubsampler.scala:57-65): planarLayout() needs no GPU; processPlanar() moves one host frame; on the device,
    // processDevice(..., PixelFormat::PLANAR) writes layout.frame_bytes bytes per frame (256-byte aligned) and
    // reconstructDevice() turns planar frames back into the packed stream -- reconstruct(planar(x)) == process(x).
    csic_planar_layout planarLayout() const
    {
        csic_planar_layout lay;
        check(csic_planar_layout_of(&params_, &lay));
        return lay;
    }
    PlanarFrame processPlanar(const std::vector<uint32_t> &argb)
    {
        PlanarFrame fr;
        fr.layout = planarLayout();
        std::vector<uint32_t> words((size_t)(fr.layout.frame_bytes / 4));
        check(csic_process_host(plan(PixelFormat::PLANAR), argb.data(), argb.size(), words.data(), words.size()));
        fr.bytes.resize((size_t)fr.layout.frame_bytes);
        std::memcpy(fr.bytes.data(), words.data(), fr.bytes.size());
        return fr;
    }
    void reconstructDevice(const void *d_planar, void *d_out, int nframes, void *hip_stream, PixelFormat f = PixelFormat::ARGB8888)
    {
        check(csic_reconstruct_device(plan(PixelFormat::PLANAR), d_planar, d_out, nframes, (int32_t)f, hip_stream));
    }
    // what row pitch (pixels, input / output) a caller that owns its surfaces should allocate for csic_process_pitched_device
    std::pair<int, int> preferredPitch(PixelFormat f = PixelFormat::ARGB8888)
    {
        int32_t ip = 0, op = 0;
        check(csic_plan_preferred_pitch(plan(f), &ip, &op));
        return {ip, op};
    }
    const char *kernelName(PixelFormat f = PixelFormat::ARGB8888) { return csic_plan_kernel_name(plan(f)); }
    // the plan behind process(): what FrameGraph records launches of (owned by this object)
    csic_plan *nativePlan(PixelFormat f = PixelFormat::ARGB8888) { return plan(f); }

private:
    csic_plan *plan(PixelFormat f)
    {
        csic_plan *&pl = plan_[(int)f];
        if (!pl) {
            csic_params p = params_;
            p.out_format = (int32_t)f;
            check(csic_plan_create(&p, device_, &pl));
        }
        return pl;
    }
    std::vector<uint32_t> run(PixelFormat f, const std::vector<uint32_t> &argb)
    {
        std::vector<uint32_t> out((size_t)out_w_ * out_h_);
        check(csic_process_host(plan(f), argb.data(), argb.size(), out.data(), out.size()));
        return out;
    }
    csic_params params_{};
    csic_plan *plan_[3] = {nullptr, nullptr, nullptr};     // one per PixelFormat
    int32_t out_w_ = 0, out_h_ = 0;
    int device_;
};

// Pre-recorded per-frame launches (csic_frame_graph_*): frames in separate device buffers, recorded once and replayed
// so that small launches overlap.  HIP = hipGraph chains ordered with the caller's stream; DIRECT = AQL packets without
// barrier bits on the library's own queues -- launch(stream) orders them with a HIP stream on the device, submit()/wait()
// by the host.  The reference processes one image at a time (ImageCompressorTopApp.scala:53-68).
enum class FrameGraphBackend : int32_t { HIP = CSIC_FRAME_GRAPH_HIP, DIRECT = CSIC_FRAME_GRAPH_DIRECT, FUSED = CSIC_FRAME_GRAPH_FUSED,
                                         AUTO = CSIC_FRAME_GRAPH_AUTO };

class FrameGraph {
public:
    FrameGraph(csic_plan *plan, const std::vector<const void *> &d_in, const std::vector<void *> &d_out,
               FrameGraphBackend backend = FrameGraphBackend::AUTO, int branches = 0)
    {
        if (d_in.size() != d_out.size() || d_in.empty())
            throw IllegalArgumentException(CSIC_EINVAL_SIZE, "need as many output as input frames (> 0)");
        check(csic_frame_graph_create_ex(plan, d_in.data(), d_out.data(), (int32_t)d_in.size(), branches, (int32_t)backend, &g_));
    }
    FrameGraph(const FrameGraph &) = delete;
    FrameGraph &operator=(const FrameGraph &) = delete;
    ~FrameGraph() { csic_frame_graph_destroy(g_); }
    void launch(void *hip_stream) { check(csic_frame_graph_launch(g_, hip_stream)); }
    int64_t submit() { int64_t t = 0; check(csic_frame_graph_submit(g_, &t)); return t; }
    void wait(int64_t ticket = -1) { check(csic_frame_graph_wait(g_, ticket)); }
    bool streamOrdered() const { return csic_frame_graph_stream_ordered(g_) == 1; }
    int branches() const { int32_t n = 0, b = 0; csic_frame_graph_count(g_, &n, &b); return b; }
    int launchBranches() const { return csic_frame_graph_launch_branches(g_); }
    FrameGraphBackend backend() const { return (FrameGraphBackend)csic_frame_graph_backend(g_); }   // the resolved one, never AUTO

private:
    csic_frame_graph *g_ = nullptr;
};

// Cycle-level model of the Decoupled pixel stream (csic_stream_*): the generated hardware's ready/valid interface, one clock
// edge per step() -- chiseltest's poke / peek / step on the reference's modules (SpatialDownsamplerSpec.scala:48-58).  Host only;
// a simulator of interface timing, never a compute path.
class StreamModel {
public:
    StreamModel(const csic_params &p, int32_t kind) { check(csic_stream_create(&p, kind, &s_)); }
    StreamModel(const StreamModel &) = delete;
    StreamModel &operator=(const StreamModel &) = delete;
    ~StreamModel() { csic_stream_destroy(s_); }
    csic_stream_in in{0, 0, 0, 0, 0};                                       // poked inputs hold their value (un-poked: 0)
    csic_stream_out peek() const { csic_stream_out o; check(csic_stream_eval(s_, &in, &o)); return o; }
    csic_stream_out step() { csic_stream_out o; check(csic_stream_step(s_, &in, &o)); return o; }
    void reset() { check(csic_stream_reset(s_)); in = csic_stream_in{0, 0, 0, 0, 0}; }
    int64_t cycles() const { return csic_stream_cycles(s_); }
    // the app's driver + collector loops (ImageCompressorTopApp.scala:76-124); max_cycles < 0 = until drained
    std::vector<uint32_t> run(const std::vector<uint32_t> &pixels, size_t max_out, int64_t max_cycles = -1, int64_t *cycles_used = nullptr)
    {
        std::vector<uint32_t> out(max_out ? max_out : 1);
        size_t n = 0;
        check(csic_stream_run(s_, pixels.data(), pixels.size(), out.data(), max_out, max_cycles, nullptr, 0, nullptr, 0, &n, cycles_used));
        out.resize(n);
        return out;
    }

private:
    csic_stream *s_ = nullptr;
};

class ImageProcessor : public ImageCompressorTop {
public:
    explicit ImageProcessor(const ImageProcessorParams &p, Rounding rounding = Rounding::FLOOR_HW, int device = 0)
        : ImageCompressorTop(p.width, p.height, p.chromaParamA, p.chromaParamB, 8, 8, 8, p.factor,
                             ProcessingStep::ChromaSubsampling, ProcessingStep::SpatialSampling,
                             ProcessingStep::ColorQuantization, rounding, device) {}
};

} // namespace csic
