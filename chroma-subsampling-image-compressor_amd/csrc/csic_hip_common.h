// csic_hip_common.h -- shared by the HIP translation units (csic_kernels.hip, csic_pipeline.hip,
// csic_multi.hip, csic_graph.hip): error macro, the HIP instantiation of the device guard and the launch
// descriptor that csic_kernels.hip prepares for the other units.
#pragma once
#include <hip/hip_runtime.h>

#include "csic_device_guard.h"
#include "csic_internal.h"

namespace csic {

#define HIP_TRY(expr)                                                                            \
    do {                                                                                         \
        hipError_t e_ = (expr);                                                                  \
        if (e_ != hipSuccess)                                                                    \
            return ::csic::set_error(CSIC_EHIP, "%s failed: %s", #expr, hipGetErrorString(e_));  \
    } while (0)

struct HipDeviceApi {
    static int get(int *d) { return (int)hipGetDevice(d); }
    static int set(int d) { return (int)hipSetDevice(d); }
};
using DeviceGuard = BasicDeviceGuard<HipDeviceApi>;

// `CSIC_DEVICE_SCOPE(dev);` at the top of an entry point: `dev` is current until the scope ends, then the
// caller's device is current again.  Returns CSIC_EHIP from the enclosing function if the switch fails.
#define CSIC_DEVICE_SCOPE(dev)                                                                              \
    ::csic::DeviceGuard csic_device_guard_(dev);                                                            \
    if (csic_device_guard_.status() != 0)                                                                   \
        return ::csic::set_error(CSIC_EHIP, "cannot make device %d current: %s", (int)(dev),                \
                                 hipGetErrorString((hipError_t)csic_device_guard_.status()))

// ---- kernel arguments (kernarg segment -> SGPRs) ----------------------------------------------------
struct KArgs {
    const uint32_t *in;
    uint32_t *out;
    int32_t W, H, Wo, Ho;
    int32_t last_sample_col;
    uint32_t my, mcb, mcr;
    int32_t f, hmask, vmask, s_first;   // hmask = h-1, vmask = v-1 (generic kernel; vmask also k_dec SROWS)
    int32_t sc_shift, bc_row_off, bc_col_in;   // k_dec SROWS: log2 f; held-sample decimated row offset / input column
    int32_t edge_y0;                    // k_avg: first row of edge blocks along grid y (in what used to be padding: the size stays 152)
    int64_t in_frame_px, out_frame_px;  // batch strides (grid z = frame)
    int32_t bdx, bdy, row_step;         // block width/height and gridDim.y * bdy, passed explicitly (see pin_args)
    int32_t ip, op;                     // row pitch of the input / output frame in pixels (>= W / Wo; == when packed)
    uint32_t mW, mWo, kW, kWo;          // k_generic: exact n / W and n / Wo for n < 2^31 as (n * m) >> k (see magic_div)
    const uint32_t *const *in_tab;      // frame-table mode (CSIC_FRAME_GRAPH_FUSED): frame z reads in_tab[z] and writes out_tab[z];
    uint32_t *const *out_tab;           //   null = frames lie back to back behind `in` / `out`
};

// The direct-dispatch engine copies sizeof(KArgs) bytes into raw kernarg memory (csic_graph.hip): every translation unit must
// see this one layout -- there is exactly one copy of this header (tests/test_bench_contract.py checks that, too).
static_assert(sizeof(KArgs) == 152 && alignof(KArgs) == 8, "KArgs layout changed: check the kernarg blocks built in csic_graph.hip");

using KernelFn = void (*)(KArgs);

// One fully resolved launch of the fused kernel: what hipLaunchKernel / hipGraphAddKernelNode need.
struct LaunchDesc {
    KernelFn fn;
    dim3 grid, block;
    KArgs args;
};

// csic_kernels.hip: validates the call, resolves kernel + geometry for `nframes` frames (<= 65535, the grid z
// limit) and fills *d.  Does not touch the device.
int prepare_launch(const csic_plan *pl, const void *d_in, void *d_out, int nframes, int32_t in_pitch, int32_t out_pitch,
                   LaunchDesc *d);
// The same for `nframes` frames in SEPARATE buffers named by device-resident pointer tables (one launch, frame index on
// grid z): `align_bits` is the bitwise OR of every frame pointer in the tables (the vector kernels need all of them 16-byte aligned).
int prepare_launch_table(const csic_plan *pl, const void *const *d_in_tab, void *const *d_out_tab, uintptr_t align_bits, int nframes,
                         LaunchDesc *d);
int enqueue(const LaunchDesc &d, hipStream_t stream);
int launch_on_stream(csic_plan *pl, const void *d_in, void *d_out, int nframes, hipStream_t stream);
int plan_device(const csic_plan *pl);
const csic_params &plan_params(const csic_plan *pl);
const Geometry &plan_geometry(const csic_plan *pl);
int plan_variant(const csic_plan *pl);                       // CSIC_TUNE_VARIANT
bool plan_nontemporal(const csic_plan *pl);                  // CSIC_TUNE_NONTEMPORAL
int plan_block_threads(const csic_plan *pl);                 // CSIC_TUNE_BLOCK_THREADS (0 = the library's choice)
void fill_base_args(const Geometry &g, int32_t ip, int32_t op, KArgs *a);
// csic_planar.hip: out_format = CSIC_FMT_PLANAR (forward: packed input -> planar frame buffers; name of the kernel a plan takes)
// csic_planar.hip: the second kernel argument of the planar kernels, and a prepared planar launch (as LaunchDesc for the packed ones)
struct PExtra {
    uint8_t *planar;              // forward: destination frame buffers, frame_bytes apart; reconstruct: source
    const uint64_t *planar_tab;   // forward, frame-table mode: device-resident table of the frames' planar buffers (else NULL)
    uint32_t *packed;             // reconstruct: destination (n pixels per frame, back to back)
    int64_t cb_off, cr_off, frame_bytes, n;
    int32_t Wm, Wc, lhe, lve, replay_last;      // module width, samples per chroma row, log2 hold_h / hold_v
    uint32_t mWm, kWm;                          // exact j / Wm (magic_div)
    int32_t T;                                  // threads per block
};
struct PlanarLaunchDesc {
    void (*fn)(KArgs, PExtra);
    dim3 grid, block;
    KArgs args;
    PExtra extra;
};
int planar_forward(const csic_plan *pl, const void *d_in, void *d_planar, int nframes, hipStream_t stream);
// `nframes` (<= 65535) frames in SEPARATE buffers named by device-resident pointer tables: resolves kernel + geometry (no device work)
int planar_prepare_table(const csic_plan *pl, const void *const *d_in_tab, void *const *d_planar_tab, uintptr_t align_bits, int nframes,
                         PlanarLaunchDesc *d);
int planar_enqueue(const PlanarLaunchDesc &d, hipStream_t stream);
void planar_kernel_name(const csic_plan *pl, char *buf, size_t len);
int planar_avg_geometry(const csic_plan *pl, int nframes, LaunchDesc *d, bool *tile);   // k_avg's geometry for a planar AVG plan
void plan_sizes(const csic_plan *pl, size_t *in_px, size_t *out_px);
int32_t plan_width(const csic_plan *pl);
void plan_out_dims(const csic_plan *pl, int32_t *wo, int32_t *ho);
int64_t plan_algorithmic_bytes(const csic_plan *pl);          // per frame (csic_algorithmic_bytes of the plan's parameters)

} // namespace csic
