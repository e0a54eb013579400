// csic_multi.hip -- one frame row-striped over several devices from ONE process.
// The reference is single-device; bench.py and the Python driver use one process per GPU
// (torch.distributed), this is the same partition for hosts that own all GPUs in one process (a JVM, a
// C++ service).  Stripes come from csic_stripe_rows (aligned, independent images): there is no halo and
// no peer traffic; every device runs the ordinary fused kernel on its stripe, on its own stream.
#include <new>
#include <vector>

#include "csic_hip_common.h"

struct csic_multi {
    struct Part {
        int device = 0;
        int32_t row0 = 0, nrows = 0, out_row0 = 0, out_nrows = 0;
        csic_plan *plan = nullptr;       // null for an empty stripe
        hipStream_t stream = nullptr;
        void *d_in = nullptr, *d_out = nullptr;   // process_host staging (lazy)
    };
    csic_params params;
    int32_t W = 0, Wo = 0, H = 0, Ho = 0;
    std::vector<Part> parts;
};

using namespace csic;

static void multi_free(csic_multi *m)
{
    for (auto &p : m->parts) {
        if (!p.plan && !p.stream && !p.d_in && !p.d_out) continue;      // never created (or an invalid device)
        DeviceGuard guard(p.device);
        if (guard.status() != 0) continue;
        if (p.stream) { (void)hipStreamSynchronize(p.stream); (void)hipStreamDestroy(p.stream); }
        if (p.d_in) (void)hipFree(p.d_in);
        if (p.d_out) (void)hipFree(p.d_out);
        csic_plan_destroy(p.plan);
    }
    delete m;
}

extern "C" {

int csic_multi_create(const csic_params *p, const int32_t *devices, int32_t ndev, csic_multi **out)
{
    if (!p || !devices || !out) return set_error(CSIC_EINVAL_NULL, "argument is NULL");
    *out = nullptr;
    if (ndev < 1 || ndev > 64) return set_error(CSIC_EINVAL_STRIPE, "ndev must be in 1..64. Got %d", ndev);
    int st = csic_validate(p);
    if (st != CSIC_OK) return st;
    if (p->out_format == CSIC_FMT_PLANAR)
        return set_error(CSIC_EINVAL_FORMAT, "csic_multi_* gathers packed row stripes: out_format must not be CSIC_FMT_PLANAR");
    csic_multi *m = new (std::nothrow) csic_multi();
    if (!m) return set_error(CSIC_ENOMEM, "out of host memory");
    m->params = *p;
    m->W = p->width; m->H = p->height;
    st = csic_out_dims(p, &m->Wo, &m->Ho);
    try { m->parts.resize(ndev); } catch (const std::bad_alloc &) { delete m; return set_error(CSIC_ENOMEM, "out of host memory"); }
    for (int i = 0; i < ndev && st == CSIC_OK; ++i) {
        csic_multi::Part &q = m->parts[i];
        q.device = devices[i];
        st = csic_stripe_rows(p, ndev, i, &q.row0, &q.nrows, &q.out_row0, &q.out_nrows);
        if (st != CSIC_OK || q.nrows == 0) continue;
        csic_params sp = *p;
        sp.height = q.nrows;
        st = csic_plan_create(&sp, q.device, &q.plan);
        if (st != CSIC_OK) break;
        DeviceGuard guard(q.device);
        hipError_t e = (hipError_t)guard.status();
        if (e == hipSuccess) e = hipStreamCreateWithFlags(&q.stream, hipStreamNonBlocking);
        if (e != hipSuccess) st = set_error(CSIC_EHIP, "stream creation on device %d failed: %s", q.device, hipGetErrorString(e));
    }
    if (st != CSIC_OK) { multi_free(m); return st; }
    *out = m;
    clear_error();
    return CSIC_OK;
}

int csic_multi_destroy(csic_multi *m)
{
    if (m) multi_free(m);
    return CSIC_OK;
}

int csic_multi_count(const csic_multi *m) { return m ? (int)m->parts.size() : 0; }

int csic_multi_stripe(const csic_multi *m, int32_t idx, int32_t *device, int32_t *row0, int32_t *nrows,
                      int32_t *out_row0, int32_t *out_nrows)
{
    if (!m || !device || !row0 || !nrows || !out_row0 || !out_nrows) return set_error(CSIC_EINVAL_NULL, "argument is NULL");
    if (idx < 0 || idx >= (int)m->parts.size()) return set_error(CSIC_EINVAL_STRIPE, "stripe index %d out of range", idx);
    const csic_multi::Part &q = m->parts[idx];
    *device = q.device; *row0 = q.row0; *nrows = q.nrows; *out_row0 = q.out_row0; *out_nrows = q.out_nrows;
    clear_error();
    return CSIC_OK;
}

int csic_multi_process_device(csic_multi *m, const void *const *d_in, void *const *d_out)
{
    if (!m || !d_in || !d_out) return set_error(CSIC_EINVAL_NULL, "argument is NULL");
    for (size_t i = 0; i < m->parts.size(); ++i) {
        csic_multi::Part &q = m->parts[i];
        if (!q.plan) continue;
        int st = launch_on_stream(q.plan, d_in[i], d_out[i], 1, q.stream);     // sets the device itself
        if (st != CSIC_OK) return st;
    }
    clear_error();
    return CSIC_OK;
}

int csic_multi_synchronize(csic_multi *m)
{
    if (!m) return set_error(CSIC_EINVAL_NULL, "multi is NULL");
    for (auto &q : m->parts) {
        if (!q.stream) continue;
        CSIC_DEVICE_SCOPE(q.device);
        HIP_TRY(hipStreamSynchronize(q.stream));
    }
    clear_error();
    return CSIC_OK;
}

int csic_multi_process_host(csic_multi *m, const uint32_t *in, size_t in_px, uint32_t *out, size_t out_px)
{
    if (!m || !in || !out) return set_error(CSIC_EINVAL_NULL, "argument is NULL");
    if (in_px != (size_t)m->W * m->H || out_px != (size_t)m->Wo * m->Ho)
        return set_error(CSIC_EINVAL_SIZE, "expected %zu input and %zu output pixels, got %zu and %zu",
                         (size_t)m->W * m->H, (size_t)m->Wo * m->Ho, in_px, out_px);
    for (auto &q : m->parts) {                       // scatter + launch + gather, all asynchronous per device
        if (!q.plan) continue;
        CSIC_DEVICE_SCOPE(q.device);
        const size_t ib = (size_t)q.nrows * m->W * 4, ob = (size_t)q.out_nrows * m->Wo * 4;
        if (!q.d_in) HIP_TRY(hipMalloc(&q.d_in, ib));
        if (!q.d_out) HIP_TRY(hipMalloc(&q.d_out, ob));
        HIP_TRY(hipMemcpyAsync(q.d_in, in + (size_t)q.row0 * m->W, ib, hipMemcpyHostToDevice, q.stream));
        int st = launch_on_stream(q.plan, q.d_in, q.d_out, 1, q.stream);
        if (st != CSIC_OK) return st;
        HIP_TRY(hipMemcpyAsync(out + (size_t)q.out_row0 * m->Wo, q.d_out, ob, hipMemcpyDeviceToHost, q.stream));
    }
    return csic_multi_synchronize(m);
}

} // extern "C"
