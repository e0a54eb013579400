// csic_pipeline.hip -- host frames in, host frames out, overlapped: the step either side of the hot path
// (SURVEY.md 8f rank 1).  The reference decodes a PNG into per-pixel ints, feeds the DUT and collects the
// output pixels (ImageProcessorModel.scala:14-52, ImageCompressorTopApp.scala:39-41,76-144); on the GPU
// the equivalent is  decode -> pinned staging -> H2D || kernel || D2H -> pinned staging -> encode.
//
// A pipeline owns `depth` slots.  Each slot has pinned host input/output buffers (hipHostMalloc), device
// input/output buffers and its own HIP stream; a frame's H2D copy, kernel and D2H copy are ordered on its
// slot's stream, and different slots overlap on the copy engines and the CUs.  The producer writes (or
// decodes) straight into the pinned input buffer it acquired, so no extra host copy is needed.
#include <new>
#include <vector>

#include "csic_hip_common.h"
#include "csic_trace.h"

struct csic_pipeline {
    struct Slot {
        uint32_t *h_in = nullptr, *h_out = nullptr;
        void *d_in = nullptr, *d_out = nullptr;
        hipStream_t stream = nullptr;
        hipEvent_t done = nullptr;
        int64_t ticket = -1;
        bool in_flight = false;
    };
    csic_plan *plan = nullptr;
    int device = 0;
    size_t in_px = 0, out_px = 0;         // 4-byte words per frame; a CSIC_FMT_PLANAR plan's output is its frame_bytes / 4
    std::vector<Slot> slots;
    int mode = CSIC_PIPELINE_ZERO_COPY;   // measured 2.4x faster than staged copies on the headline shape
    int64_t next_ticket = 0;
    int head = 0;        // next slot to acquire
    int tail = 0;        // oldest submitted, not yet collected
    int acquired = -1;   // slot handed out by acquire_input and not yet submitted
    int pending = 0;     // submitted and not collected
};

using namespace csic;

static void free_slots(csic_pipeline *pp)
{
    for (auto &s : pp->slots) {
        if (s.stream) (void)hipStreamSynchronize(s.stream);
        if (s.done) (void)hipEventDestroy(s.done);
        if (s.stream) (void)hipStreamDestroy(s.stream);
        if (s.h_in) (void)hipHostFree(s.h_in);
        if (s.h_out) (void)hipHostFree(s.h_out);
        if (s.d_in) (void)hipFree(s.d_in);
        if (s.d_out) (void)hipFree(s.d_out);
    }
    pp->slots.clear();
}

extern "C" {

int csic_pipeline_create(csic_plan *plan, int32_t depth, csic_pipeline **out)
{
    if (!plan || !out) return set_error(CSIC_EINVAL_NULL, "argument is NULL");
    *out = nullptr;
    if (depth < 1 || depth > 64) return set_error(CSIC_EINVAL_SIZE, "pipeline depth must be in 1..64. Got %d", depth);
    csic_pipeline *pp = new (std::nothrow) csic_pipeline();
    if (!pp) return set_error(CSIC_ENOMEM, "out of host memory");
    pp->plan = plan;
    pp->device = plan_device(plan);
    plan_sizes(plan, &pp->in_px, &pp->out_px);
    DeviceGuard guard(pp->device);
    if (guard.status() != 0) {
        const int dev = pp->device;
        delete pp;
        return set_error(CSIC_EHIP, "cannot make device %d current: %s", dev, hipGetErrorString((hipError_t)guard.status()));
    }
    try { pp->slots.resize(depth); } catch (const std::bad_alloc &) { delete pp; return set_error(CSIC_ENOMEM, "out of host memory"); }
    hipError_t e = hipSuccess;
    for (auto &s : pp->slots) {
        if (e == hipSuccess) e = hipHostMalloc(reinterpret_cast<void **>(&s.h_in), pp->in_px * 4, hipHostMallocMapped);
        if (e == hipSuccess) e = hipHostMalloc(reinterpret_cast<void **>(&s.h_out), pp->out_px * 4, hipHostMallocMapped);
        if (e == hipSuccess) e = hipMalloc(&s.d_in, pp->in_px * 4);
        if (e == hipSuccess) e = hipMalloc(&s.d_out, pp->out_px * 4);
        if (e == hipSuccess) e = hipStreamCreateWithFlags(&s.stream, hipStreamNonBlocking);
        if (e == hipSuccess) e = hipEventCreateWithFlags(&s.done, hipEventDisableTiming);
    }
    if (e != hipSuccess) {
        free_slots(pp);
        delete pp;
        return set_error(e == hipErrorOutOfMemory ? CSIC_ENOMEM : CSIC_EHIP, "pipeline allocation failed: %s", hipGetErrorString(e));
    }
    *out = pp;
    clear_error();
    return CSIC_OK;
}

int csic_pipeline_destroy(csic_pipeline *pp)
{
    if (!pp) return CSIC_OK;
    {
        DeviceGuard guard(pp->device);
        if (guard.status() == 0) free_slots(pp);
    }
    delete pp;
    return CSIC_OK;
}

int csic_pipeline_acquire_input(csic_pipeline *pp, uint32_t **host_in)
{
    if (!pp || !host_in) return set_error(CSIC_EINVAL_NULL, "argument is NULL");
    if (pp->acquired >= 0) return set_error(CSIC_EINVAL_SIZE, "an input buffer is already acquired; submit it first");
    csic_pipeline::Slot &s = pp->slots[pp->head];
    if (s.in_flight)
        return set_error(CSIC_EINVAL_SIZE, "all %zu pipeline slots hold uncollected frames; collect one first", pp->slots.size());
    pp->acquired = pp->head;
    *host_in = s.h_in;
    clear_error();
    return CSIC_OK;
}

int csic_pipeline_submit(csic_pipeline *pp, int64_t *ticket)
{
    if (!pp) return set_error(CSIC_EINVAL_NULL, "pipeline is NULL");
    if (pp->acquired < 0) return set_error(CSIC_EINVAL_SIZE, "no acquired input buffer to submit");
    CSIC_DEVICE_SCOPE(pp->device);
    int st;
    csic_pipeline::Slot &s = pp->slots[pp->acquired];
    if (pp->mode == CSIC_PIPELINE_ZERO_COPY) {
        // The kernel streams the pinned host frame over PCIe itself and writes the result straight back:
        // input rows that hold no surviving pixel (r % f != 0) never cross the bus, and the read and
        // write directions are busy at the same time inside one launch.
        trace::Range r("csic:launch_kernel_zero_copy");
        st = launch_on_stream(pp->plan, s.h_in, s.h_out, 1, s.stream);
        if (st != CSIC_OK) return st;
    } else {
        { trace::Range r("csic:enqueue_h2d"); HIP_TRY(hipMemcpyAsync(s.d_in, s.h_in, pp->in_px * 4, hipMemcpyHostToDevice, s.stream)); }
        {
            trace::Range r("csic:launch_kernel");
            st = launch_on_stream(pp->plan, s.d_in, s.d_out, 1, s.stream);
            if (st != CSIC_OK) return st;
        }
        { trace::Range r("csic:enqueue_d2h"); HIP_TRY(hipMemcpyAsync(s.h_out, s.d_out, pp->out_px * 4, hipMemcpyDeviceToHost, s.stream)); }
    }
    HIP_TRY(hipEventRecord(s.done, s.stream));
    s.ticket = pp->next_ticket++;
    s.in_flight = true;
    if (ticket) *ticket = s.ticket;
    pp->acquired = -1;
    pp->head = (pp->head + 1) % (int)pp->slots.size();
    pp->pending += 1;
    clear_error();
    return CSIC_OK;
}

int csic_pipeline_collect(csic_pipeline *pp, const uint32_t **host_out, int64_t *ticket)
{
    if (!pp || !host_out) return set_error(CSIC_EINVAL_NULL, "argument is NULL");
    if (pp->pending == 0) return set_error(CSIC_EINVAL_SIZE, "no submitted frame to collect");
    CSIC_DEVICE_SCOPE(pp->device);
    csic_pipeline::Slot &s = pp->slots[pp->tail];
    { trace::Range r("csic:wait_gpu"); HIP_TRY(hipEventSynchronize(s.done)); }
    *host_out = s.h_out;
    if (ticket) *ticket = s.ticket;
    s.in_flight = false;          // the output buffer stays valid until this slot is submitted again
    pp->tail = (pp->tail + 1) % (int)pp->slots.size();
    pp->pending -= 1;
    clear_error();
    return CSIC_OK;
}

int csic_pipeline_pending(const csic_pipeline *pp) { return pp ? pp->pending : 0; }

int csic_pipeline_set_mode(csic_pipeline *pp, int32_t mode)
{
    if (!pp) return set_error(CSIC_EINVAL_NULL, "pipeline is NULL");
    if (mode != CSIC_PIPELINE_STAGED && mode != CSIC_PIPELINE_ZERO_COPY)
        return set_error(CSIC_EINVAL_SIZE, "unknown pipeline mode %d", mode);
    if (pp->pending || pp->acquired >= 0) return set_error(CSIC_EINVAL_SIZE, "cannot change mode with frames in flight");
    pp->mode = mode;
    clear_error();
    return CSIC_OK;
}

} // extern "C"
