// csic_files.hip -- PNG files in, PNG files out: the frame pipeline with the codec on worker threads.
//
// The reference handles one image at a time on one thread: readImage -> per-pixel poke ... peek -> writeImage
// (ImageProcessorModel.scala:14-52, ImageCompressorTopApp.scala:39-41,133-144).  Once the kernel runs at the HBM roofline the
// PNG codec is everything: one host thread decodes a 4K frame at ~160 Mpixel/s (round 2, zlib; ~330 with the reader's own
// inflate, csic_inflate.cpp) and encodes at ~50, the kernel moves 4.8 Tpixel/s (profiles/r02_host_io.json).  So the step either side of the path is a pool: `decode_threads` workers inflate
// files STRAIGHT INTO pinned frame slots and launch the fused kernel on the slot's own stream (zero-copy: it reads the pinned
// frame over PCIe and writes the pinned result, dead rows never cross the bus), `encode_threads` workers wait for a slot's
// event and deflate its result to the output file.  Slots are the bounded queue between the two pools: a decoder that finds
// no free slot waits for an encoder to release one.  Files are independent, so frames finish in whatever order the workers
// allow; every output file is byte for byte what the serial path (csic_png_read_argb -> csic_pipeline_* -> csic_png_write_argb,
// one thread) writes for the same input, because codec, level and pixels are the same.
#include <atomic>
#include <chrono>
#include <condition_variable>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <deque>
#include <mutex>
#include <new>
#include <string>
#include <thread>
#include <vector>

#include <sys/mman.h>

#include "csic_hip_common.h"
#include "csic_trace.h"

using namespace csic;

namespace {

using Clock = std::chrono::steady_clock;
inline double secs(Clock::time_point a, Clock::time_point b) { return std::chrono::duration<double>(b - a).count(); }

struct Slot {
    uint32_t *h_in = nullptr, *h_out = nullptr;
    void *registered = nullptr;                   // base of the one registered allocation that holds both (else: two hipHostMalloc blocks)
    hipStream_t stream = nullptr;
    hipEvent_t done = nullptr;
};

// Pinned memory for one slot.  hipHostMalloc pins 4 KiB page by page under a lock: 5.7 ms for a 35 MB slot on an idle host, 49 ms
// each when 16 threads ask at once, and hipHostFree is no better (77 ms each) -- the decoders of a batch used to wait 0.06-0.1 s
// for their slots.  Memory that is already there in 2 MiB pages (aligned_alloc + MADV_HUGEPAGE + first touch) registers in
// 0.1 ms and the whole sequence takes 2.8 ms alone, 30 ms wall for 16 at once (tools/probe_pin.py, profiles/r03_probe_pin.log).
// Registered memory has the same address on the device (checked), so the kernels take it like hipHostMalloc's.
constexpr size_t kHugePage = (size_t)2 << 20;

bool register_slot_memory(size_t in_bytes, size_t out_bytes, Slot &s)
{
    if (std::getenv("CSIC_FILES_NO_REGISTER")) return false;                        // (tests: the hipHostMalloc way)
    const size_t in_room = (in_bytes + 4095) & ~(size_t)4095, total = (in_room + out_bytes + kHugePage - 1) & ~(kHugePage - 1);
    void *p = std::aligned_alloc(kHugePage, total);
    if (!p) return false;
    (void)madvise(p, total, MADV_HUGEPAGE);
    std::memset(p, 0, total);                     // first touch: the pages exist, as huge pages where the host allows, before they are pinned
    void *d = nullptr;
    if (hipHostRegister(p, total, hipHostRegisterMapped) != hipSuccess) { (void)hipGetLastError(); std::free(p); return false; }
    if (hipHostGetDevicePointer(&d, p, 0) != hipSuccess || d != p) { (void)hipGetLastError(); (void)hipHostUnregister(p); std::free(p); return false; }
    s.registered = p;
    s.h_in = static_cast<uint32_t *>(p);
    s.h_out = reinterpret_cast<uint32_t *>(static_cast<unsigned char *>(p) + in_room);
    return true;
}

struct Shared {
    csic_plan *plan = nullptr;
    int device = 0;
    size_t in_px = 0, out_px = 0, final_px = 0;
    int32_t final_w = 0, final_h = 0, out_w = 0, out_h = 0, level = 6;
    const char *const *in_paths = nullptr;
    const char *const *out_paths = nullptr;
    int32_t n = 0;
    std::vector<Slot> slots;

    std::mutex mu;
    std::condition_variable cv_free, cv_work;
    std::deque<int> free_slots;
    int acquired = 0;                             // files whose decoder has got its slot: at n, no slot will be asked for again
    int created = 0;                              // slots allocated so far (each by the decoder thread that first needed one)
    std::deque<std::pair<int, int>> inflight;     // (slot, file index), oldest first
    int decoders_running = 0;
    std::atomic<int> next_file{0};
    std::atomic<bool> failed{false};
    int status = CSIC_OK;
    std::string message;
    std::mutex launch_mu;                         // a plan is not thread-safe: launches take turns (the enqueue is microseconds)
    // accounting (under mu)
    double decode_s = 0, encode_s = 0, gpu_wait_s = 0, slot_wait_s = 0;
    int64_t done_files = 0;
    int max_inflight = 0;
};

void fail(Shared &sh, int status, const std::string &msg)
{
    std::lock_guard<std::mutex> lk(sh.mu);
    if (!sh.failed.exchange(true)) { sh.status = status; sh.message = msg; }
    sh.cv_free.notify_all();
    sh.cv_work.notify_all();
}

bool make_slot(Shared &sh, Slot &s)
{
    hipError_t e = hipSuccess;
    if (!register_slot_memory(sh.in_px * 4, sh.out_px * 4, s)) {                    // (a host without hipHostRegister: the slower way)
        e = hipHostMalloc(reinterpret_cast<void **>(&s.h_in), sh.in_px * 4, hipHostMallocMapped);
        if (e == hipSuccess) e = hipHostMalloc(reinterpret_cast<void **>(&s.h_out), sh.out_px * 4, hipHostMallocMapped);
    }
    if (e == hipSuccess) e = hipStreamCreateWithFlags(&s.stream, hipStreamNonBlocking);
    if (e == hipSuccess) e = hipEventCreateWithFlags(&s.done, hipEventDisableTiming);
    if (e != hipSuccess) {
        (void)hipGetLastError();
        fail(sh, e == hipErrorOutOfMemory ? CSIC_ENOMEM : CSIC_EHIP, std::string("frame-slot allocation failed: ") + hipGetErrorString(e));
        return false;
    }
    return true;
}

void free_slot(Slot &s)                           // the slot is idle: its event has been waited for (or it never launched)
{
    if (s.done) (void)hipEventDestroy(s.done);
    if (s.stream) (void)hipStreamDestroy(s.stream);
    if (s.registered) {
        (void)hipHostUnregister(s.registered);
        std::free(s.registered);
    } else {
        if (s.h_in) (void)hipHostFree(s.h_in);
        if (s.h_out) (void)hipHostFree(s.h_out);
    }
    s = Slot();
}

// A worker must never let an exception leave its thread (std::terminate would take the host process -- a JVM, a Python
// interpreter -- with it): std::string building, vector growth and deque pushes can all throw std::bad_alloc.  The thread
// functions below run their loops inside try / catch and turn whatever escapes into the batch's failure.  "out of memory" is
// short enough for std::string's in-place buffer, so reporting it cannot allocate.
void fail_nothrow(Shared &sh) noexcept
{
    try { fail(sh, CSIC_ENOMEM, "out of memory"); }
    catch (...) { sh.failed.store(true); sh.cv_free.notify_all(); sh.cv_work.notify_all(); }
}

void decoder_loop(Shared &sh, double &t_dec, double &t_wait)
{
    if (hipSetDevice(sh.device) != hipSuccess) { fail(sh, CSIC_EHIP, "decoder thread: cannot make the plan's device current"); }
    while (!sh.failed.load()) {
        const int i = sh.next_file.fetch_add(1);
        if (i >= sh.n) break;
        int slot = -1;
        {
            const auto w0 = Clock::now();
            std::unique_lock<std::mutex> lk(sh.mu);
            bool fresh = false;
            if (sh.free_slots.empty() && sh.created < (int)sh.slots.size()) {
                slot = sh.created++;                                               // mine to allocate, outside the lock
                fresh = true;
            } else {
                sh.cv_free.wait(lk, [&] { return !sh.free_slots.empty() || sh.failed.load(); });
                if (sh.failed.load()) break;
                slot = sh.free_slots.front();
                sh.free_slots.pop_front();
            }
            ++sh.acquired;                                                         // counted only once the slot is in hand
            lk.unlock();
            if (fresh && !make_slot(sh, sh.slots[slot])) break;
            t_wait += secs(w0, Clock::now());
        }
        Slot &s = sh.slots[slot];
        const auto d0 = Clock::now();
        int st;
        {
            trace::Range r("csic:decode_png");
            st = csic_png_read_argb(sh.in_paths[i], s.h_in, sh.in_px);             // straight into pinned memory
        }
        t_dec += secs(d0, Clock::now());
        if (st != CSIC_OK) { fail(sh, st, std::string("file ") + std::to_string(i) + ": " + csic_last_error()); break; }
        {
            std::lock_guard<std::mutex> lk(sh.launch_mu);
            trace::Range r("csic:launch_kernel_zero_copy");
            st = launch_on_stream(sh.plan, s.h_in, s.h_out, 1, s.stream);
            if (st == CSIC_OK && hipEventRecord(s.done, s.stream) != hipSuccess) st = set_error(CSIC_EHIP, "hipEventRecord failed");
        }
        if (st != CSIC_OK) { fail(sh, st, std::string("file ") + std::to_string(i) + ": " + csic_last_error()); break; }
        {
            std::lock_guard<std::mutex> lk(sh.mu);
            sh.inflight.emplace_back(slot, i);
            if ((int)sh.inflight.size() > sh.max_inflight) sh.max_inflight = (int)sh.inflight.size();
        }
        sh.cv_work.notify_one();
    }
}

void decoder(Shared &sh) noexcept
{
    double t_dec = 0, t_wait = 0;
    try { decoder_loop(sh, t_dec, t_wait); }
    catch (...) { fail_nothrow(sh); }
    // whatever happened above, this decoder is no longer running: the encoders' exit condition depends on the count
    std::lock_guard<std::mutex> lk(sh.mu);
    sh.decode_s += t_dec;
    sh.slot_wait_s += t_wait;
    if (--sh.decoders_running == 0) sh.cv_work.notify_all();
}

void encoder_loop(Shared &sh, double &t_enc, double &t_gpu, int64_t &files)
{
    if (hipSetDevice(sh.device) != hipSuccess) { fail(sh, CSIC_EHIP, "encoder thread: cannot make the plan's device current"); }
    std::vector<uint32_t> cropped;
    for (;;) {
        int slot = -1, i = -1;
        {
            std::unique_lock<std::mutex> lk(sh.mu);
            sh.cv_work.wait(lk, [&] { return !sh.inflight.empty() || sh.decoders_running == 0 || sh.failed.load(); });
            if (sh.inflight.empty()) break;                                        // producers are done (or failed) and nothing is queued
            slot = sh.inflight.front().first;
            i = sh.inflight.front().second;
            sh.inflight.pop_front();
        }
        Slot &s = sh.slots[slot];
        const auto g0 = Clock::now();
        hipError_t e;
        {
            trace::Range r("csic:wait_gpu");
            e = hipEventSynchronize(s.done);                                       // (also on the failure path: the slot must be idle before it is freed)
        }
        t_gpu += secs(g0, Clock::now());
        if (e != hipSuccess) fail(sh, CSIC_EHIP, std::string("file ") + std::to_string(i) + ": hipEventSynchronize failed: " + hipGetErrorString(e));
        if (!sh.failed.load()) {
            const auto e0 = Clock::now();
            const uint32_t *src = s.h_out;
            if (sh.final_w != sh.out_w || sh.final_h != sh.out_h) {
                // what the reference's collector keeps: the first final_w * final_h pixels of the output STREAM, laid out final_w
                // per row; pixels it never got stay magenta (ImageCompressorTopApp.scala:108-124, :133-142)
                cropped.assign(sh.final_px, 0xFFFF00FFu);
                const size_t keep = sh.out_px < sh.final_px ? sh.out_px : sh.final_px;
                std::memcpy(cropped.data(), s.h_out, keep * 4);
                src = cropped.data();
            }
            trace::Range r("csic:encode_png");
            const int st = png_write_argb_threads(sh.out_paths[i], src, sh.final_w, sh.final_h, sh.level, 1);   // the pool is the parallelism
            t_enc += secs(e0, Clock::now());
            if (st != CSIC_OK) fail(sh, st, std::string("file ") + std::to_string(i) + ": " + csic_last_error());
            else ++files;
        }
        bool retire;
        {
            std::lock_guard<std::mutex> lk(sh.mu);
            retire = sh.acquired >= sh.n;                                          // every file has its slot: this one is not needed again
            if (!retire) sh.free_slots.push_back(slot);
        }
        // un-pinning a slot takes a millisecond or two; done here it runs beside the other threads' last frames instead of
        // 40-odd times in a row after the join (0.1 s of a 0.4 s batch)
        if (retire) free_slot(s); else sh.cv_free.notify_one();
    }
}

void encoder(Shared &sh) noexcept
{
    double t_enc = 0, t_gpu = 0;
    int64_t files = 0;
    // An exception that ends the loop early may leave queued frames behind: nobody encodes them (the batch has failed), and the
    // caller synchronises every slot's stream before freeing it (release()), so no slot is freed under a running kernel.
    try { encoder_loop(sh, t_enc, t_gpu, files); }
    catch (...) { fail_nothrow(sh); }
    std::lock_guard<std::mutex> lk(sh.mu);
    sh.encode_s += t_enc;
    sh.gpu_wait_s += t_gpu;
    sh.done_files += files;
}

// width and height from the first 33 bytes (signature + IHDR); anything odd is left to csic_png_info's full parse for its message
bool peek_png_dims(const char *path, int32_t *w, int32_t *h)
{
    unsigned char b[33];
    FILE *fp = std::fopen(path, "rb");
    if (!fp) return false;
    const size_t got = std::fread(b, 1, sizeof b, fp);
    std::fclose(fp);
    static const unsigned char sig[8] = {0x89, 'P', 'N', 'G', 0x0D, 0x0A, 0x1A, 0x0A};
    if (got != sizeof b || std::memcmp(b, sig, 8) != 0 || std::memcmp(b + 12, "IHDR", 4) != 0) return false;
    const uint32_t ww = ((uint32_t)b[16] << 24) | ((uint32_t)b[17] << 16) | ((uint32_t)b[18] << 8) | b[19];
    const uint32_t hh = ((uint32_t)b[20] << 24) | ((uint32_t)b[21] << 16) | ((uint32_t)b[22] << 8) | b[23];
    if (ww == 0 || hh == 0 || ww > 0x7FFFFFFFu || hh > 0x7FFFFFFFu) return false;
    *w = (int32_t)ww; *h = (int32_t)hh;
    return true;
}

} // namespace

extern "C" {

int csic_process_png_files(csic_plan *plan, const char *const *in_paths, const char *const *out_paths, int32_t nfiles,
                           int32_t decode_threads, int32_t encode_threads, int32_t png_level, int32_t final_width, int32_t final_height,
                           csic_files_stats *stats)
{
    if (!plan || !in_paths || !out_paths) return set_error(CSIC_EINVAL_NULL, "argument is NULL");
    if (nfiles < 1) return set_error(CSIC_EINVAL_SIZE, "nfiles must be positive. Got %d", nfiles);
    for (int i = 0; i < nfiles; ++i)
        if (!in_paths[i] || !out_paths[i]) return set_error(CSIC_EINVAL_NULL, "file %d: path is NULL", i);
    if (png_level < 0 || png_level > 9) return set_error(CSIC_EINVAL_SIZE, "png_level must be in 0..9. Got %d", png_level);
    if (plan_params(plan).out_format == CSIC_FMT_PLANAR)
        return set_error(CSIC_EINVAL_FORMAT, "the file pools write packed pixels: the plan's out_format must not be CSIC_FMT_PLANAR");
    Shared sh;
    sh.plan = plan;
    sh.device = plan_device(plan);
    plan_sizes(plan, &sh.in_px, &sh.out_px);
    plan_out_dims(plan, &sh.out_w, &sh.out_h);
    sh.final_w = final_width > 0 ? final_width : sh.out_w;
    sh.final_h = final_height > 0 ? final_height : sh.out_h;
    sh.final_px = (size_t)sh.final_w * (size_t)sh.final_h;
    // the collector's image is never larger than the stream it collects from -- (W / f) x (H / f) against ceil sizes,
    // ImageCompressorTopApp.scala:44-45 -- so anything far beyond the plan's output is a caller's mistake, refused before a
    // worker tries to allocate it
    if (final_width < 0 || final_height < 0 || sh.final_px >= ((size_t)1 << 31) || sh.final_px > 4 * sh.out_px + ((size_t)1 << 20))
        return set_error(CSIC_EINVAL_SIZE, "final size %dx%d is out of proportion to the plan's %dx%d output", sh.final_w, sh.final_h, sh.out_w, sh.out_h);
    sh.level = png_level;
    sh.in_paths = in_paths; sh.out_paths = out_paths; sh.n = nfiles;
    // the files must be frames of this plan: checked up front so that a wrong batch fails before any thread starts
    for (int i = 0; i < nfiles; ++i) {
        int32_t w = 0, h = 0;
        if (!peek_png_dims(in_paths[i], &w, &h)) {
            const int st = csic_png_info(in_paths[i], &w, &h);
            if (st != CSIC_OK) return st;
        }
        if ((size_t)w * (size_t)h != sh.in_px || w != plan_width(plan))
            return set_error(CSIC_EINVAL_SIZE, "%s is %dx%d, the plan processes %dx%zu frames", in_paths[i], w, h, plan_width(plan),
                             sh.in_px / (size_t)plan_width(plan));
    }
    // Defaults from the CPU time the process may really use (affinity mask and cgroup quota, host_cpu_budget) -- not from the
    // CPUs the machine shows: a GPU box shows 256 and grants 16.  Twice the budget, because a worker also waits (file reads,
    // slot hand-offs, the GPU): measured on that box, 24 + 8 and 32 + 16 threads finish cfg 5 in 0.29-0.30 s, 16 + 8 in 0.33 s
    // and 12 + 4 in 0.40 s (profiles/r03_probe_files_wall.log).  Two thirds decode, one third encode -- a 4K frame costs 45-49 ms
    // to decode and 23 ms to encode.  At most 32 + 16 as before; the numbers chosen are reported in csic_files_stats.
    const int budget = host_cpu_budget();
    int total = 2 * budget;
    if (total > 48) total = 48;
    if (total < 2) total = 2;
    int defD = (2 * total + 2) / 3, defE = total - defD;
    if (defD > 32) defD = 32;
    if (defE > 16) defE = 16;
    if (defE < 1) defE = 1;
    int D = decode_threads > 0 ? decode_threads : defD;
    int E = encode_threads > 0 ? encode_threads : defE;
    if (D > nfiles) D = nfiles;
    if (E > nfiles) E = nfiles;
    if (D > 256) D = 256;
    if (E > 256) E = 256;
    // slots: one per worker plus two in flight on the GPU, within 8 GiB of pinned memory
    const size_t slot_bytes = (sh.in_px + sh.out_px) * 4;
    size_t S = (size_t)D + (size_t)E + 2;
    const size_t cap = ((size_t)8 << 30) / (slot_bytes ? slot_bytes : 1);
    if (S > cap) S = cap < 2 ? 2 : cap;
    if (S > (size_t)nfiles + 1) S = (size_t)nfiles + 1;
    if ((size_t)D > S) D = (int)S;

    CSIC_DEVICE_SCOPE(sh.device);
    try { sh.slots.resize(S); } catch (const std::bad_alloc &) { return set_error(CSIC_ENOMEM, "out of host memory"); }
    auto release = [&] {
        for (auto &s : sh.slots) {                // what the encoders have not retired already (all of them after a failure)
            if (s.stream) (void)hipStreamSynchronize(s.stream);
            free_slot(s);
        }
    };
    sh.decoders_running = D;                      // (slots are created by the decoder threads, when first needed)

    const auto t0 = Clock::now();
    std::vector<std::thread> pool;
    try {
        for (int k = 0; k < D; ++k) pool.emplace_back(decoder, std::ref(sh));
        for (int k = 0; k < E; ++k) pool.emplace_back(encoder, std::ref(sh));
    } catch (...) {
        // could not start every worker: the ones that exist must still come to an end
        {
            std::lock_guard<std::mutex> lk(sh.mu);
            const int started_dec = (int)pool.size() < D ? (int)pool.size() : D;
            sh.decoders_running -= D - started_dec;
        }
        fail(sh, CSIC_ENOMEM, "could not start the worker threads");
    }
    for (auto &t : pool) t.join();
    const double wall = secs(t0, Clock::now());
    release();
    if (stats) {
        stats->frames = sh.done_files;
        stats->wall_s = wall;
        stats->decode_s = sh.decode_s; stats->encode_s = sh.encode_s; stats->gpu_wait_s = sh.gpu_wait_s; stats->slot_wait_s = sh.slot_wait_s;
        stats->decode_threads = D; stats->encode_threads = E; stats->slots = (int32_t)sh.created; stats->max_in_flight = sh.max_inflight;
        stats->in_pixels = (int64_t)sh.in_px * sh.done_files; stats->out_pixels = (int64_t)sh.final_px * sh.done_files;
    }
    if (sh.failed.load()) return set_error(sh.status, "%s", sh.message.c_str());
    clear_error();
    return CSIC_OK;
}

} // extern "C"
