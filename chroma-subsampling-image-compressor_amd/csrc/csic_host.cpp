// csic_host.cpp -- host-only half of the C ABI: parameter validation (the reference's require()s),
// geometry, the algorithmic-byte model and the row-stripe partition.  No HIP calls in this file, so
// every function here works on a machine without a GPU.
//
// Citations are relative to /root/reference/.
#include "csic_internal.h"
#include "csic_trace.h"

#include <atomic>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <mutex>

#include <dlfcn.h>
#include <sched.h>

namespace csic {

// ---- optional roctx markers (csic_trace.h) ------------------------------------------------------------------------------
namespace trace {
namespace {
using push_fn = int (*)(const char *);
using pop_fn = int (*)();
std::atomic<int> g_state{0};                  // 0 = not looked yet, 1 = on, 2 = off
push_fn g_push = nullptr;
pop_fn g_pop = nullptr;
std::once_flag g_once;

void look()
{
    const char *env = std::getenv("CSIC_ROCTX");
    if (env && *env && std::strcmp(env, "0") != 0) {
        for (const char *name : {"librocprofiler-sdk-roctx.so.1", "librocprofiler-sdk-roctx.so", "libroctx64.so.4", "libroctx64.so"}) {
            void *h = dlopen(name, RTLD_NOW | RTLD_GLOBAL);
            if (!h) continue;
            g_push = reinterpret_cast<push_fn>(dlsym(h, "roctxRangePushA"));
            g_pop = reinterpret_cast<pop_fn>(dlsym(h, "roctxRangePop"));
            if (g_push && g_pop) break;
            g_push = nullptr; g_pop = nullptr;
        }
    }
    g_state.store(g_push && g_pop ? 1 : 2, std::memory_order_release);
}
} // namespace

bool enabled()
{
    int s = g_state.load(std::memory_order_acquire);
    if (s == 0) { std::call_once(g_once, look); s = g_state.load(std::memory_order_acquire); }
    return s == 1;
}
void push(const char *name) { if (g_push) (void)g_push(name); }
void pop() { if (g_pop) (void)g_pop(); }
} // namespace trace

int host_cpu_budget()
{
    int n = 0;
    cpu_set_t set;
    CPU_ZERO(&set);
    if (sched_getaffinity(0, sizeof set, &set) == 0) n = CPU_COUNT(&set);
    if (n <= 0) n = 4;
    // cgroup v2 cpu.max ("<quota> <period>" or "max ..."), then v1 cfs quota
    long long q = -1, per = -1;
    if (FILE *fp = std::fopen("/sys/fs/cgroup/cpu.max", "r")) {
        char a[32] = "", b[32] = "";
        if (std::fscanf(fp, "%31s %31s", a, b) == 2 && std::strcmp(a, "max") != 0) { q = std::atoll(a); per = std::atoll(b); }
        std::fclose(fp);
    } else {
        if (FILE *fq = std::fopen("/sys/fs/cgroup/cpu/cpu.cfs_quota_us", "r")) { if (std::fscanf(fq, "%lld", &q) != 1) q = -1; std::fclose(fq); }
        if (FILE *fr = std::fopen("/sys/fs/cgroup/cpu/cpu.cfs_period_us", "r")) { if (std::fscanf(fr, "%lld", &per) != 1) per = -1; std::fclose(fr); }
    }
    if (q > 0 && per > 0) {
        const int quota = (int)((q + per - 1) / per);
        if (quota >= 1 && quota < n) n = quota;
    }
    return n < 1 ? 1 : n;
}

static thread_local char g_last_error[512] = "";

int set_error(int status, const char *fmt, ...)
{
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_last_error, sizeof g_last_error, fmt, ap);
    va_end(ap);
    return status;
}

void clear_error() { g_last_error[0] = '\0'; }

static int validate_impl(const csic_params *p)
{
    if (!p) return set_error(CSIC_EINVAL_NULL, "params is NULL");
    // ImageProcessor.scala:22-23 "width/height must be positive"
    if (p->width <= 0 || p->height <= 0)
        return set_error(CSIC_EINVAL_DIMS, "width and height must be positive. Got %dx%d", p->width, p->height);
    // pixel indices are 32-bit in the kernels
    if ((int64_t)p->width * (int64_t)p->height >= (int64_t)1 << 31)
        return set_error(CSIC_EINVAL_DIMS, "width*height must be < 2^31. Got %dx%d", p->width, p->height);
    // ImageProcessor.scala:24, SpatialDownsampler.scala:8 "Factor must be 1, 2, 4, or 8"
    if (!(p->factor == 1 || p->factor == 2 || p->factor == 4 || p->factor == 8))
        return set_error(CSIC_EINVAL_FACTOR, "factor must be 1, 2, 4, or 8. Got %d", p->factor);
    // ImageProcessor.scala:27, ChromaSubsampler.scala:17
    if (!(p->chroma_a == 4 || p->chroma_a == 2 || p->chroma_a == 1))
        return set_error(CSIC_EINVAL_CHROMA_A, "chroma_a must be 4, 2, or 1. Got %d", p->chroma_a);
    // ImageProcessor.scala:28, ChromaSubsampler.scala:18
    if (!(p->chroma_b == p->chroma_a || p->chroma_b == 0))
        return set_error(CSIC_EINVAL_CHROMA_B, "chroma_b must be equal to chroma_a (%d) or 0. Got %d",
                         p->chroma_a, p->chroma_b);
    // ColorQuantizer.scala:12-15
    const int32_t bits[3] = {p->y_bits, p->cb_bits, p->cr_bits};
    static const char *const nm[3] = {"Y", "Cb", "Cr"};
    for (int k = 0; k < 3; ++k)
        if (bits[k] < 1 || bits[k] > 8)
            return set_error(CSIC_EINVAL_BITS, "%s target bits must be between 1 and 8. Got %d", nm[k], bits[k]);
    // ImageCompressorTop.scala:27-31
    int seen[4] = {0, 0, 0, 0};
    for (int k = 0; k < 3; ++k) {
        if (p->op[k] < CSIC_OP_SPATIAL || p->op[k] > CSIC_OP_CHROMA)
            return set_error(CSIC_EINVAL_OP_PERMUTATION, "op%d must be a valid reorderable operation. Got %d",
                             k + 1, p->op[k]);
        if (seen[p->op[k]]++)
            return set_error(CSIC_EINVAL_OP_PERMUTATION,
                             "op1, op2, and op3 must be distinct and form a permutation. Got %d,%d,%d",
                             p->op[0], p->op[1], p->op[2]);
    }
    if (p->rounding != CSIC_ROUND_FLOOR_HW && p->rounding != CSIC_ROUND_TRUNC_SW)
        return set_error(CSIC_EINVAL_ROUNDING, "rounding must be FLOOR_HW(0) or TRUNC_SW(1). Got %d", p->rounding);
    if (p->sampling != CSIC_SAMPLING_HOLD_DECIMATE && p->sampling != CSIC_SAMPLING_AVG)
        return set_error(CSIC_EINVAL_SAMPLING, "sampling must be HOLD_DECIMATE(0) or AVG(1). Got %d", p->sampling);
    if (p->sampling == CSIC_SAMPLING_AVG &&
        !(p->op[0] == CSIC_OP_CHROMA && p->op[1] == CSIC_OP_SPATIAL && p->op[2] == CSIC_OP_QUANT))
        return set_error(CSIC_EINVAL_SAMPLING, "the AVG extension is defined for the order chroma -> spatial -> quant only");
    if (p->in_format != CSIC_FMT_ARGB8888 && p->in_format != CSIC_FMT_YCBCR888X)
        return set_error(CSIC_EINVAL_FORMAT, "in_format must be ARGB8888(0) or YCBCR888X(1). Got %d", p->in_format);
    if (p->out_format != CSIC_FMT_ARGB8888 && p->out_format != CSIC_FMT_YCBCR888X && p->out_format != CSIC_FMT_PLANAR)
        return set_error(CSIC_EINVAL_FORMAT, "out_format must be ARGB8888(0), YCBCR888X(1) or PLANAR(2). Got %d", p->out_format);
    // ImageProcessor.scala:25 -- a rule of ImageProcessorParams only; the raw RTL accepts any size
    if (p->strict_divisible && (p->width % p->factor != 0 || p->height % p->factor != 0))
        return set_error(CSIC_EINVAL_NOT_DIVISIBLE,
                         "Image dimensions must be divisible by spatial downsampling factor. Got %dx%d, factor %d",
                         p->width, p->height, p->factor);
    return CSIC_OK;
}

int derive_geometry(const csic_params *p, Geometry *g)
{
    int st = validate_impl(p);
    if (st != CSIC_OK) return st;
    g->W = p->width; g->H = p->height; g->f = p->factor;
    g->Wo = (p->width + p->factor - 1) / p->factor;
    g->Ho = (p->height + p->factor - 1) / p->factor;
    g->h = 4 / p->chroma_a;                       // ChromaSubsampler.scala:26
    g->v = (p->chroma_b == 0) ? 2 : 1;            // ChromaSubsampler.scala:27
    int is = 0, ic = 0;
    for (int k = 0; k < 3; ++k) {
        if (p->op[k] == CSIC_OP_SPATIAL) is = k;
        if (p->op[k] == CSIC_OP_CHROMA) ic = k;
    }
    // With f == 1 the decimator is the identity and the two order classes coincide.
    g->s_first = (is < ic && p->factor > 1) ? 1 : 0;
    g->last_sample_col = ((p->width - 1) / g->h) * g->h;
    g->mask_y  = (0xFFu << (8 - p->y_bits))  & 0xFFu;   // ColorQuantizer.scala:29-31,42-44
    g->mask_cb = (0xFFu << (8 - p->cb_bits)) & 0xFFu;
    g->mask_cr = (0xFFu << (8 - p->cr_bits)) & 0xFFu;
    return CSIC_OK;
}

// The planar layout of csic.h (see the definition there): where the chroma stage sits decides which counters the output sees.
void planar_layout(const Geometry &g, const csic_params *p, csic_planar_layout *L)
{
    const int64_t n = (int64_t)g.Wo * g.Ho;
    L->y_width = g.Wo; L->y_height = g.Ho;
    if (p->sampling == CSIC_SAMPLING_AVG) {
        L->module_width = g.Wo; L->hold_h = g.h > g.f ? g.h / g.f : 1; L->hold_v = g.v > g.f ? g.v / g.f : 1; L->replay_last = 0;
    } else if (g.s_first || g.f == 1) {                       // the chroma counters run over the (decimated) stream, full width
        L->module_width = g.W; L->hold_h = g.h; L->hold_v = g.v; L->replay_last = 1;
    } else {                                                  // chroma before the decimator: image coordinates, every f-th kept
        L->module_width = g.Wo; L->hold_h = g.h > g.f ? g.h / g.f : 1; L->hold_v = 1; L->replay_last = 1;
    }
    const int64_t Wm = L->module_width, rows = (n + Wm - 1) / Wm;           // chroma rows, the last one possibly partial
    L->chroma_width = (int32_t)((Wm + L->hold_h - 1) / L->hold_h);
    L->chroma_height = (int32_t)((rows + L->hold_v - 1) / L->hold_v);
    // samples that exist: full sample rows, and what the last chroma row holds if it is a sample row
    const int64_t last_len = n - (rows - 1) * Wm, last_is_sample_row = ((rows - 1) % L->hold_v) == 0;
    L->chroma_samples = (int64_t)(L->chroma_height - (last_is_sample_row ? 1 : 0)) * L->chroma_width +
                        (last_is_sample_row ? (last_len + L->hold_h - 1) / L->hold_h : 0);
    auto up = [](int64_t x) { return (x + 255) & ~(int64_t)255; };
    const int64_t plane = (int64_t)L->chroma_width * L->chroma_height;
    L->y_offset = 0;
    L->cb_offset = up(n);
    L->cr_offset = L->cb_offset + up(plane);
    L->frame_bytes = L->cr_offset + up(plane);
    L->payload_bytes = n + 2 * L->chroma_samples;
}

void magic_div(uint32_t d, uint32_t *m, uint32_t *k)
{
    uint32_t l = 0;
    while ((1ull << l) < d) ++l;
    *k = 31 + l;
    *m = (uint32_t)(((1ull << *k) + d - 1) / d);
}

} // namespace csic

using namespace csic;

extern "C" {

int csic_abi_version(void) { return CSIC_ABI_VERSION; }

int csic_params_default(csic_params *p, int32_t width, int32_t height)
{
    if (!p) return set_error(CSIC_EINVAL_NULL, "params is NULL");
    std::memset(p, 0, sizeof *p);
    p->width = width; p->height = height;
    p->chroma_a = 4; p->chroma_b = 4;             // ImageProcessorModel.scala:38-39
    p->y_bits = p->cb_bits = p->cr_bits = 8;
    p->factor = 1;
    p->op[0] = CSIC_OP_CHROMA; p->op[1] = CSIC_OP_SPATIAL; p->op[2] = CSIC_OP_QUANT;
    p->rounding = CSIC_ROUND_FLOOR_HW;
    p->sampling = CSIC_SAMPLING_HOLD_DECIMATE;
    p->in_format = p->out_format = CSIC_FMT_ARGB8888;
    p->strict_divisible = 0;
    clear_error();
    return CSIC_OK;
}

int csic_validate(const csic_params *p)
{
    int st = validate_impl(p);
    if (st == CSIC_OK) clear_error();
    return st;
}

int csic_out_dims(const csic_params *p, int32_t *out_width, int32_t *out_height)
{
    if (!out_width || !out_height) return set_error(CSIC_EINVAL_NULL, "output pointer is NULL");
    Geometry g;
    int st = derive_geometry(p, &g);
    if (st != CSIC_OK) return st;
    *out_width = g.Wo; *out_height = g.Ho;
    clear_error();
    return CSIC_OK;
}

int csic_algorithmic_bytes(const csic_params *p, int64_t *bytes)
{
    if (!bytes) return set_error(CSIC_EINVAL_NULL, "bytes is NULL");
    Geometry g;
    int st = derive_geometry(p, &g);
    if (st != CSIC_OK) return st;
    // every byte of each input row that holds a surviving pixel + the output;
    // with AVG sampling every input row is live: A_avg = 4*W*H + 4*Wo*Ho
    // planar output: the same input bytes, and the payload the format stores instead of 4 bytes per output pixel
    int64_t out_bytes = 4ll * g.Wo * g.Ho;
    if (p->out_format == CSIC_FMT_PLANAR) {
        csic_planar_layout L;
        planar_layout(g, p, &L);
        out_bytes = L.payload_bytes;
    }
    if (p->sampling == CSIC_SAMPLING_AVG) *bytes = 4ll * g.W * g.H + out_bytes;
    else                                  *bytes = 4ll * g.W * g.Ho + out_bytes;
    clear_error();
    return CSIC_OK;
}

int csic_planar_layout_of(const csic_params *p, csic_planar_layout *layout)
{
    if (!layout) return set_error(CSIC_EINVAL_NULL, "layout is NULL");
    Geometry g;
    int st = derive_geometry(p, &g);
    if (st != CSIC_OK) return st;
    std::memset(layout, 0, sizeof *layout);
    planar_layout(g, p, layout);
    clear_error();
    return CSIC_OK;
}

// alignment unit of independent row blocks (see csic_stripe_rows)
static int stripe_unit(const Geometry &g, const csic_params *p, int64_t *L)
{
    if (!g.s_first || p->sampling == CSIC_SAMPLING_AVG) {
        *L = (g.v > g.f) ? g.v : g.f;
        return CSIC_OK;
    }
    if (g.W % g.f != 0)
        return set_error(CSIC_EINVAL_STRIPE,
                         "spatial-before-chroma with width %% factor != 0 cannot be row-striped independently");
    *L = (int64_t)g.v * g.f * g.f;
    return CSIC_OK;
}

int csic_stripe_rows(const csic_params *p, int32_t nranks, int32_t rank,
                     int32_t *row0, int32_t *nrows, int32_t *out_row0, int32_t *out_nrows)
{
    if (!row0 || !nrows || !out_row0 || !out_nrows) return set_error(CSIC_EINVAL_NULL, "output pointer is NULL");
    Geometry g;
    int st = derive_geometry(p, &g);
    if (st != CSIC_OK) return st;
    if (nranks <= 0 || rank < 0 || rank >= nranks)
        return set_error(CSIC_EINVAL_STRIPE, "bad rank %d of %d", rank, nranks);
    // Chroma before spatial (and the AVG extension): chroma indices are image coordinates, so a stripe that
    // starts on a row that is both a vertical chroma (block) row and a decimation (pooling) row depends on
    // nothing above it: L = lcm(v, f) = max(v, f).  Spatial before chroma: chroma runs on the decimated
    // stream with its column counter modulo the FULL width W (ImageCompressorTop.scala:52-58): one chroma
    // row = W decimated pixels = f decimated rows = f*f input rows, and only when f divides W.
    int64_t L;
    st = stripe_unit(g, p, &L);
    if (st != CSIC_OK) return st;
    const int64_t blocks = (g.H + L - 1) / L;
    const int64_t b0 = blocks * rank / nranks, b1 = blocks * (rank + 1) / nranks;
    int64_t r0 = b0 * L, r1 = b1 * L;
    if (r0 > g.H) r0 = g.H;
    if (r1 > g.H) r1 = g.H;
    *row0 = (int32_t)r0; *nrows = (int32_t)(r1 - r0);
    *out_row0 = (int32_t)(r0 / g.f);                         // r0 is a multiple of f
    *out_nrows = (int32_t)(((r1 - r0) + g.f - 1) / g.f);
    clear_error();
    return CSIC_OK;
}

int csic_stripe_halo(const csic_params *p, int32_t nranks, int32_t rank, const int32_t *row_splits,
                     int32_t *proc_row0, int32_t *proc_nrows, int32_t *halo_above, int32_t *tail_below,
                     int32_t *out_row0, int32_t *out_nrows)
{
    if (!row_splits || !proc_row0 || !proc_nrows || !halo_above || !tail_below || !out_row0 || !out_nrows)
        return set_error(CSIC_EINVAL_NULL, "argument is NULL");
    Geometry g;
    int st = derive_geometry(p, &g);
    if (st != CSIC_OK) return st;
    if (nranks <= 0 || rank < 0 || rank >= nranks) return set_error(CSIC_EINVAL_STRIPE, "bad rank %d of %d", rank, nranks);
    if (row_splits[0] != 0 || row_splits[nranks] != g.H)
        return set_error(CSIC_EINVAL_STRIPE, "row_splits must start at 0 and end at height %d", g.H);
    for (int r = 0; r < nranks; ++r)
        if (row_splits[r + 1] < row_splits[r]) return set_error(CSIC_EINVAL_STRIPE, "row_splits must be non-decreasing");
    int64_t L;
    st = stripe_unit(g, p, &L);
    if (st != CSIC_OK) return st;
    // aligned-down boundary in front of every stripe; the last one is the true end of the frame
    auto lo = [&](int r) -> int64_t { return r >= nranks ? g.H : (row_splits[r] / L) * L; };
    for (int r = 1; r < nranks; ++r)          // the halo of rank r must live entirely on rank r-1
        if (lo(r) < row_splits[r - 1])
            return set_error(CSIC_EINVAL_STRIPE, "stripe %d (%d rows) is shorter than the %lld-row halo rank %d needs",
                             r - 1, row_splits[r] - row_splits[r - 1], (long long)(row_splits[r] - lo(r)), r);
    const int64_t a = lo(rank), b = lo(rank + 1);
    *proc_row0 = (int32_t)a;
    *proc_nrows = (int32_t)(b - a);
    *halo_above = (int32_t)(row_splits[rank] - a);
    *tail_below = (int32_t)(row_splits[rank + 1] - b);         // 0 on the last rank (b = H)
    *out_row0 = (int32_t)(a / g.f);
    *out_nrows = (int32_t)((b - a + g.f - 1) / g.f);
    clear_error();
    return CSIC_OK;
}

const char *csic_strerror(int status)
{
    switch (status) {
    case CSIC_OK: return "ok";
    case CSIC_EINVAL_NULL: return "null argument";
    case CSIC_EINVAL_DIMS: return "invalid image dimensions";
    case CSIC_EINVAL_FACTOR: return "invalid spatial factor";
    case CSIC_EINVAL_CHROMA_A: return "invalid chroma parameter a";
    case CSIC_EINVAL_CHROMA_B: return "invalid chroma parameter b";
    case CSIC_EINVAL_BITS: return "invalid quantiser bit depth";
    case CSIC_EINVAL_OP_PERMUTATION: return "operations are not a permutation";
    case CSIC_EINVAL_ROUNDING: return "invalid rounding mode";
    case CSIC_EINVAL_FORMAT: return "invalid pixel format";
    case CSIC_EINVAL_NOT_DIVISIBLE: return "dimensions not divisible by factor";
    case CSIC_EINVAL_SAMPLING: return "invalid sampling mode";
    case CSIC_EINVAL_STRIPE: return "invalid row-stripe request";
    case CSIC_EINVAL_SIZE: return "buffer size mismatch";
    case CSIC_ENODEVICE: return "no HIP device";
    case CSIC_EHIP: return "HIP runtime error";
    case CSIC_ENOMEM: return "out of memory";
    case CSIC_ECAPTURE: return "the stream is capturing and this operation cannot be captured";
    case CSIC_EIO: return "file I/O error";
    case CSIC_EFORMAT: return "bad or unsupported PNG";
    default: return "unknown status";
    }
}

const char *csic_last_error(void) { return g_last_error; }

} // extern "C"
