// csic_graph.hip -- BASELINE.json configs[4]: "hipGraph-captured per-frame launch".
//
// One kernel node per frame, built with the explicit graph API (no stream capture, no helper streams):
// the launch descriptor of every frame comes from the same prepare_launch() that csic_process_device uses,
// so a replayed node is bit-for-bit the eager launch.  Frames are independent images (the reference builds
// a fresh DUT per image, ImageCompressorTopApp.scala:53-68), so the graph keeps only `branches` chains of
// dependencies: with one chain every node waits for its predecessor's completion signal (the dependent-kernel
// boundary, ~1.7 us -- half of a 4K sf=4 frame's 4.2 us), with B chains the runtime overlaps B frames.
#include <new>
#include <vector>

#include "csic_hip_common.h"

struct csic_frame_graph {
    int device = 0;
    int32_t nframes = 0, branches = 0;
    hipGraph_t graph = nullptr;
    hipGraphExec_t exec = nullptr;
};

using namespace csic;

static void graph_free(csic_frame_graph *g)
{
    if (g->exec) (void)hipGraphExecDestroy(g->exec);
    if (g->graph) (void)hipGraphDestroy(g->graph);
    delete g;
}

extern "C" {

int csic_frame_graph_create(csic_plan *plan, const void *const *d_in, void *const *d_out, int32_t nframes,
                            int32_t branches, csic_frame_graph **out)
{
    if (!out) return set_error(CSIC_EINVAL_NULL, "out is NULL");
    *out = nullptr;
    if (!plan || !d_in || !d_out) return set_error(CSIC_EINVAL_NULL, "argument is NULL");
    if (nframes < 1 || nframes > 65536) return set_error(CSIC_EINVAL_SIZE, "nframes must be in 1..65536. Got %d", nframes);
    if (branches <= 0) branches = CSIC_FRAME_GRAPH_DEFAULT_BRANCHES;
    if (branches > nframes) branches = nframes;
    CSIC_DEVICE_SCOPE(plan_device(plan));

    csic_frame_graph *g = new (std::nothrow) csic_frame_graph();
    if (!g) return set_error(CSIC_ENOMEM, "out of host memory");
    g->device = plan_device(plan);
    g->nframes = nframes;
    g->branches = branches;
    std::vector<hipGraphNode_t> nodes;
    try { nodes.resize(nframes); } catch (const std::bad_alloc &) { graph_free(g); return set_error(CSIC_ENOMEM, "out of host memory"); }

    hipError_t e = hipGraphCreate(&g->graph, 0);
    int st = CSIC_OK;
    for (int k = 0; k < nframes && e == hipSuccess; ++k) {
        LaunchDesc d;
        st = prepare_launch(plan, d_in[k], d_out[k], 1, 0, 0, &d);
        if (st != CSIC_OK) break;
        void *params[1] = {&d.args};                       // copied by hipGraphAddKernelNode
        hipKernelNodeParams np;
        np.func = reinterpret_cast<void *>(d.fn);
        np.gridDim = d.grid;
        np.blockDim = d.block;
        np.sharedMemBytes = 0;
        np.kernelParams = params;
        np.extra = nullptr;
        const hipGraphNode_t *dep = (k >= branches) ? &nodes[k - branches] : nullptr;
        e = hipGraphAddKernelNode(&nodes[k], g->graph, dep, dep ? 1 : 0, &np);
    }
    if (st == CSIC_OK && e == hipSuccess) e = hipGraphInstantiate(&g->exec, g->graph, nullptr, nullptr, 0);
    if (st != CSIC_OK) { graph_free(g); return st; }
    if (e != hipSuccess) {
        graph_free(g);
        return set_error(CSIC_EHIP, "building the frame graph failed: %s", hipGetErrorString(e));
    }
    *out = g;
    clear_error();
    return CSIC_OK;
}

int csic_frame_graph_launch(csic_frame_graph *g, void *hip_stream)
{
    if (!g) return set_error(CSIC_EINVAL_NULL, "graph is NULL");
    CSIC_DEVICE_SCOPE(g->device);
    HIP_TRY(hipGraphLaunch(g->exec, static_cast<hipStream_t>(hip_stream)));
    clear_error();
    return CSIC_OK;
}

int csic_frame_graph_count(const csic_frame_graph *g, int32_t *nframes, int32_t *branches)
{
    if (!g) return set_error(CSIC_EINVAL_NULL, "graph is NULL");
    if (nframes) *nframes = g->nframes;
    if (branches) *branches = g->branches;
    clear_error();
    return CSIC_OK;
}

int csic_frame_graph_destroy(csic_frame_graph *g)
{
    if (!g) return CSIC_OK;
    DeviceGuard guard(g->device);
    graph_free(g);
    return CSIC_OK;
}

} // extern "C"
