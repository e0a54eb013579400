// host_test.cpp -- exercises include/csic.hpp.  `host_test cpu` checks the construction-time
// require()s (no GPU needed); `host_test gpu` also pushes the reference's known-answer vectors through
// the HIP path (RGB2YCbCrTester.scala:12-18, ColorQuantizerSpec.scala:43-61, SpatialDownsamplerSpec.scala:26).
#include <cstdio>
#include <cstring>

#include "csic.hpp"
#include "../../chroma-subsampling-image-compressor_amd/csrc/csic_device_guard.h"

using namespace csic;
using PS = ProcessingStep;

static int fails = 0;
#define EXPECT(cond) do { if (!(cond)) { std::printf("FAIL %s:%d %s\n", __FILE__, __LINE__, #cond); ++fails; } } while (0)

template <class F> static int iae_status(F f)
{
    try { f(); } catch (const IllegalArgumentException &e) { return e.status; } catch (...) { return 1; }
    return 0;
}

static void cpu_checks()
{
    EXPECT(iae_status([] { ImageProcessorParams(16, 16, 2, 2, 0); }) == 0);
    EXPECT(iae_status([] { ImageProcessorParams(0, 16, 1, 4, 4); }) == CSIC_EINVAL_DIMS);
    EXPECT(iae_status([] { ImageProcessorParams(16, 16, 3, 4, 4); }) == CSIC_EINVAL_FACTOR);       // SpatialDownsamplerSpec.scala:147-151
    EXPECT(iae_status([] { ImageProcessorParams(15, 16, 2, 4, 4); }) == CSIC_EINVAL_NOT_DIVISIBLE); // ImageProcessor.scala:25
    EXPECT(iae_status([] { ImageProcessorParams(16, 16, 1, 3, 3); }) == CSIC_EINVAL_CHROMA_A);
    EXPECT(iae_status([] { ImageProcessorParams(16, 16, 1, 2, 1); }) == CSIC_EINVAL_CHROMA_B);
    EXPECT(iae_status([] { ImageCompressorTop(4, 4, 4, 4, 8, 8, 8, 2, PS::SpatialSampling, PS::SpatialSampling, PS::ChromaSubsampling); }) == CSIC_EINVAL_OP_PERMUTATION);
    EXPECT(iae_status([] { ImageCompressorTop(4, 4, 4, 4, 9, 8, 8, 2, PS::SpatialSampling, PS::ColorQuantization, PS::ChromaSubsampling); }) == CSIC_EINVAL_BITS);
    ImageCompressorTop t(5, 3, 4, 4, 8, 8, 8, 2, PS::SpatialSampling, PS::ColorQuantization, PS::ChromaSubsampling);
    EXPECT(t.outWidth() == 3 && t.outHeight() == 2);                                              // SpatialDownsamplerSpec.scala:120-122
    try { ImageProcessorParams(16, 16, 3, 4, 4); } catch (const IllegalArgumentException &e) {
        EXPECT(std::strstr(e.what(), "requirement failed: factor must be 1, 2, 4, or 8") != nullptr);
    }
    // the planar layout is host arithmetic: 64x16 4:2:0 at factor 1 stores 1024 Y bytes and 2 x 256 chroma samples
    {
        ImageCompressorTop p(64, 16, 2, 0, 8, 8, 8, 1, PS::ChromaSubsampling, PS::SpatialSampling, PS::ColorQuantization);
        const csic_planar_layout lay = p.planarLayout();
        EXPECT(lay.y_width == 64 && lay.y_height == 16 && lay.hold_h == 2 && lay.hold_v == 2 && lay.replay_last == 1);
        EXPECT(lay.chroma_width == 32 && lay.chroma_height == 8 && lay.chroma_samples == 256 && lay.payload_bytes == 1536);
        EXPECT(lay.cb_offset % 256 == 0 && lay.cr_offset % 256 == 0 && lay.frame_bytes % 256 == 0 && lay.frame_bytes >= lay.payload_bytes);
    }
}

// The device guard every csic_* entry point opens (csrc/csic_device_guard.h), driven by a fake runtime: it must
// switch only when needed, always put the caller's device back, and surface get/set failures without switching.
struct FakeRt {
    static int cur, sets, fail_get, fail_set_on;
    static int get(int *d) { if (fail_get) return 101; *d = cur; return 0; }
    static int set(int d) { if (d == fail_set_on) return 100; cur = d; ++sets; return 0; }
};
int FakeRt::cur = 0, FakeRt::sets = 0, FakeRt::fail_get = 0, FakeRt::fail_set_on = -1;

static void device_guard_checks()
{
    using Guard = BasicDeviceGuard<FakeRt>;
    FakeRt::cur = 3; FakeRt::sets = 0;
    { Guard g(3); EXPECT(g.status() == 0 && !g.switched() && FakeRt::cur == 3); }
    EXPECT(FakeRt::cur == 3 && FakeRt::sets == 0);                 // same device: no runtime call at all
    { Guard g(5); EXPECT(g.status() == 0 && g.switched() && g.previous() == 3 && FakeRt::cur == 5); }
    EXPECT(FakeRt::cur == 3 && FakeRt::sets == 2);                 // switched and restored
    { Guard outer(1); { Guard inner(2); EXPECT(FakeRt::cur == 2); } EXPECT(FakeRt::cur == 1); }
    EXPECT(FakeRt::cur == 3);                                      // nesting unwinds in order
    FakeRt::fail_set_on = 7;
    { Guard g(7); EXPECT(g.status() == 100 && !g.switched() && FakeRt::cur == 3); }
    EXPECT(FakeRt::cur == 3);                                      // failed switch: nothing to restore
    FakeRt::fail_set_on = -1; FakeRt::fail_get = 1;
    { Guard g(4); EXPECT(g.status() == 101 && !g.switched()); }
    FakeRt::fail_get = 0;
    EXPECT(FakeRt::cur == 3);
}

static uint32_t argb(int r, int g, int b) { return 0xFF000000u | (r << 16) | (g << 8) | b; }
static uint32_t ycc(int y, int cb, int cr) { return (uint32_t)y | (cb << 8) | (cr << 16); }

static void gpu_checks()
{
    // 5 primaries, FLOOR_HW (what the RTL is checked against) and TRUNC_SW
    const std::vector<uint32_t> prim = {argb(0, 0, 0), argb(255, 255, 255), argb(255, 0, 0), argb(0, 255, 0), argb(0, 0, 255)};
    {
        ImageCompressorTop t(5, 1, 4, 4, 8, 8, 8, 1, PS::ChromaSubsampling, PS::SpatialSampling, PS::ColorQuantization);
        const auto o = t.processYCbCr(prim);
        const uint32_t want[5] = {ycc(0, 128, 128), ycc(255, 128, 128), ycc(77, 85, 255), ycc(149, 43, 21), ycc(29, 255, 107)};
        for (int i = 0; i < 5; ++i) EXPECT(o[i] == want[i]);
        const auto rgb = t.process(prim);                          // inverse of the FLOOR results
        const uint32_t wrgb[5] = {argb(0, 0, 0), argb(255, 255, 255), argb(255, 3, 3), argb(2, 255, 2), argb(0, 1, 255)};
        for (int i = 0; i < 5; ++i) EXPECT(rgb[i] == wrgb[i]);
    }
    {
        ImageCompressorTop t(5, 1, 4, 4, 8, 8, 8, 1, PS::ChromaSubsampling, PS::SpatialSampling, PS::ColorQuantization, Rounding::TRUNC_SW);
        const auto o = t.processYCbCr(prim);
        const uint32_t want[5] = {ycc(0, 128, 128), ycc(255, 128, 128), ycc(77, 86, 255), ycc(149, 44, 22), ycc(29, 255, 108)};
        for (int i = 0; i < 5; ++i) EXPECT(o[i] == want[i]);
    }
    // decimation KAT: 4x4 grey ramp, f = 2 -> stream indices {0, 2, 8, 10}
    {
        std::vector<uint32_t> ramp(16);
        for (int i = 0; i < 16; ++i) ramp[i] = argb(i, i, i);
        ImageProcessor ip(ImageProcessorParams(4, 4, 2, 4, 4));
        const auto o = ip.processYCbCr(ramp);
        const int want[4] = {0, 2, 8, 10};
        EXPECT(o.size() == 4);
        for (int i = 0; i < 4; ++i) EXPECT((o[i] & 0xFF) == (uint32_t)want[i] && ((o[i] >> 8) & 0xFF) == 128);
    }
    // quantiser KAT Y3Cb3Cr2 on a grey pixel: (235,235,235) -> Y 235 & 0xE0 = 224, Cb 128 & 0xE0 = 128, Cr 128 & 0xC0 = 128
    {
        ImageCompressorTop t(1, 1, 4, 4, 3, 3, 2, 1, PS::ColorQuantization, PS::SpatialSampling, PS::ChromaSubsampling);
        const auto o = t.processYCbCr({argb(235, 235, 235)});
        EXPECT(o[0] == ycc(224, 128, 128));
    }
    // planar output against the packed YCbCr stream of the same object: the Y plane is the stream's Y bytes, the chroma planes
    // are its Cb / Cr at the sample points (4:2:0, factor 1: even rows, even columns); the preferred pitch is the packed one
    {
        const int W = 64, H = 16;
        std::vector<uint32_t> img((size_t)W * H);
        uint32_t x = 12345u;
        for (auto &px : img) { x = x * 1664525u + 1013904223u; px = 0xFF000000u | (x >> 8); }
        ImageCompressorTop t(W, H, 2, 0, 7, 6, 5, 1, PS::ChromaSubsampling, PS::SpatialSampling, PS::ColorQuantization);
        const auto packed = t.processYCbCr(img);
        const PlanarFrame fr = t.processPlanar(img);
        EXPECT((int64_t)fr.bytes.size() == fr.layout.frame_bytes && fr.layout.chroma_samples == (W / 2) * (H / 2));
        bool ok = true;
        for (int r = 0; r < H; ++r)
            for (int c = 0; c < W; ++c) {
                ok = ok && fr.y()[r * W + c] == (packed[r * W + c] & 0xFF);
                if (r % 2 == 0 && c % 2 == 0) {
                    const int k = (r / 2) * fr.layout.chroma_width + c / 2;
                    ok = ok && fr.cb()[k] == ((packed[r * W + c] >> 8) & 0xFF) && fr.cr()[k] == ((packed[r * W + c] >> 16) & 0xFF);
                }
            }
        EXPECT(ok);
        EXPECT(std::strncmp(t.kernelName(PixelFormat::PLANAR), "k_planar_flat", 13) == 0);
        const auto pitch = t.preferredPitch();
        EXPECT(pitch.first == W && pitch.second == W);
    }
    // wrong buffer size is an IllegalArgumentException, bad device a RuntimeError
    {
        ImageCompressorTop t(4, 4, 4, 4, 8, 8, 8, 1, PS::ChromaSubsampling, PS::SpatialSampling, PS::ColorQuantization);
        EXPECT(iae_status([&] { t.process(std::vector<uint32_t>(15)); }) == CSIC_EINVAL_SIZE);
        bool rt = false;
        try { ImageCompressorTop bad(4, 4, 4, 4, 8, 8, 8, 1, PS::ChromaSubsampling, PS::SpatialSampling, PS::ColorQuantization, Rounding::FLOOR_HW, 99); bad.process(std::vector<uint32_t>(16)); }
        catch (const RuntimeError &e) { rt = e.status == CSIC_ENODEVICE; }
        EXPECT(rt);
    }
}

// SpatialDownsamplerSpec.scala:155-230 in C++: in16x16.png -> ImageProcessorParams(w, h, 2, 2, 0) ->
// ImageProcessor -> 8x8, pixels pinned by APP_OUTPUT/spatial_downsampler_integration_420_sf2.png
static void integration_flow(const char *in_png, const char *golden_png, const char *out_png)
{
    const Image in = ImageProcessorModel::readImage(in_png);
    EXPECT(in.width == 16 && in.height == 16);
    ImageProcessor dut(ImageProcessorParams(in.width, in.height, 2, 2, 0));
    Image out;
    out.width = dut.outWidth(); out.height = dut.outHeight();
    out.argb = dut.process(in.argb);
    ImageProcessorModel::writeImage(out, out_png);
    const Image back = ImageProcessorModel::readImage(out_png), want = ImageProcessorModel::readImage(golden_png);
    EXPECT(back.width == 8 && back.height == 8 && back.argb == want.argb);
}

static void png_checks(const char *in_png)
{
    const Image im = ImageProcessorModel::readImage(in_png);
    EXPECT(im.width == 16 && im.height == 16 && im.argb.size() == 256);
    EXPECT((im.argb[0] >> 24) == 0xFF);
    const ImageProcessorParams p = ImageProcessorModel::getImageParams(im, 1);
    EXPECT(p.factor == 1 && p.chromaParamA == 4 && p.chromaParamB == 4);
    bool io = false;
    try { ImageProcessorModel::readImage("/nonexistent/x.png"); } catch (const RuntimeError &e) { io = e.status == CSIC_EIO; }
    EXPECT(io);
}

// the back-pressure KAT of SpatialDownsamplerSpec.scala:48-58 and a 4x4 frame through ImageCompressorTop, from C++
static void stream_model_checks()
{
    csic_params p;
    csic_params_default(&p, 4, 4);
    p.factor = 2;
    p.in_format = p.out_format = CSIC_FMT_YCBCR888X;
    StreamModel dut(p, CSIC_STREAM_SPATIAL);
    dut.in.out_ready = 0; dut.step();
    EXPECT(dut.peek().in_ready == 0);
    dut.in.out_ready = 1; dut.step();
    EXPECT(dut.peek().in_ready == 1 && dut.cycles() == 2);
    csic_params t;
    csic_params_default(&t, 4, 4);
    t.factor = 2; t.out_format = CSIC_FMT_YCBCR888X;
    StreamModel top(t, CSIC_STREAM_TOP);
    std::vector<uint32_t> px(16, 0xFFFFFFFFu);                        // white: (255, 128, 128) -- RGB2YCbCrTester.scala:13
    int64_t cyc = 0;
    const std::vector<uint32_t> out = top.run(px, 16, -1, &cyc);
    EXPECT(out.size() == 4 && out[0] == (255u | (128u << 8) | (128u << 16)) && cyc >= 2 * 16 - 2);
    bool iae = false;
    t.factor = 3;
    try { StreamModel bad(t, CSIC_STREAM_TOP); } catch (const IllegalArgumentException &e) { iae = e.status == CSIC_EINVAL_FACTOR; }
    EXPECT(iae);
}

int main(int argc, char **argv)
{
    cpu_checks();
    stream_model_checks();
    device_guard_checks();
    if (argc > 2) png_checks(argv[2]);
    if (argc > 1 && std::strcmp(argv[1], "gpu") == 0) {
        gpu_checks();
        if (argc > 4) integration_flow(argv[2], argv[3], argv[4]);
    }
    std::printf(fails ? "%d check(s) failed\n" : "all checks passed\n", fails);
    return fails ? 1 : 0;
}
