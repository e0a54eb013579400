// tools/ubench_overlap.hip -- developer micro-benchmark (not part of the product): how can independent SMALL
// launches of the fused kernel overlap on MI355X?  One cfg 5 frame (3840x2160, sf=4: 10.4 MB) or one strong-
// scaling stripe of cfg 4 (8192x1024, sf=2: 25 MB) is ~1.3-3.1 us of data behind a ~1.7 us dependent-kernel
// boundary.  Candidates, each timed over `reps` rounds of 64 distinct frames (HIP events + host wall clock):
//   eager1      : hipLaunchKernel back to back on one stream (barrier bit on every packet)
//   anyorder    : hipExtLaunchKernel(..., hipExtAnyOrderLaunch) on one stream
//   eagerS      : round robin over S streams
//   cap-any     : stream capture of the anyorder sequence -> graph replay
//   cap-forkS   : stream capture with an S-way fork/join -> graph replay
//   graph-bB    : explicit graph, B independent chains (what csic_frame_graph_create builds)
// usage: ubench_overlap [cfg5|stripe8|stripe16] [reps] [block_threads]
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -Iinclude -I<pkg>/csrc tools/ubench_overlap.hip <pkg>/csrc/csic_host.cpp <pkg>/csrc/csic_png.cpp -lz -o tools/ubench_overlap
#include "csic_kernels.hip"

#include <hip/hip_ext.h>

#include <chrono>
#include <cstdlib>
#include <functional>
#include <string>
#include <vector>

using namespace csic;
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s (line %d)\n", #x, hipGetErrorString(e), __LINE__); exit(1);} } while (0)

int main(int argc, char **argv)
{
    const std::string what = argc > 1 ? argv[1] : "cfg5";
    const int reps = argc > 2 ? atoi(argv[2]) : 30;
    const int tpb = argc > 3 ? atoi(argv[3]) : 0;
    const int N = 64;
    csic_params p;
    if (what == "cfg5") { csic_params_default(&p, 3840, 2160); p.factor = 4; p.y_bits = 3; p.cb_bits = 3; p.cr_bits = 2; }
    else if (what == "stripe16") { csic_params_default(&p, 8192, 512); p.factor = 2; }
    else { csic_params_default(&p, 8192, 1024); p.factor = 2; }
    p.chroma_a = 2; p.chroma_b = 0;
    csic_plan *pl = nullptr;
    if (csic_plan_create(&p, 0, &pl) != CSIC_OK) { printf("plan: %s\n", csic_last_error()); return 1; }
    if (tpb) csic_plan_tune(pl, CSIC_TUNE_BLOCK_THREADS, tpb);
    size_t ipx, opx;
    plan_sizes(pl, &ipx, &opx);
    int64_t alg = 0;
    csic_algorithmic_bytes(&p, &alg);
    uint32_t *din, *dout;
    CK(hipMalloc(&din, ipx * 4 * N));
    CK(hipMalloc(&dout, opx * 4 * N));
    csic_synth_frame_device(din, (int64_t)ipx * N, 0, 20250629u, nullptr);
    CK(hipDeviceSynchronize());
    std::vector<LaunchDesc> d(N);
    for (int k = 0; k < N; ++k)
        if (prepare_launch(pl, din + (size_t)k * ipx, dout + (size_t)k * opx, 1, 0, 0, &d[k]) != CSIC_OK) { printf("prepare: %s\n", csic_last_error()); return 1; }
    printf("%s: %s, %d frames/round, %lld alg bytes/frame, floor %.3f us/frame, grid %ux%u block %ux%u\n", what.c_str(), csic_plan_kernel_name(pl), N,
           (long long)alg, alg / 8e12 * 1e6, d[0].grid.x, d[0].grid.y, d[0].block.x, d[0].block.y);

    const int NS = 16;
    hipStream_t s[NS];
    for (int i = 0; i < NS; ++i) CK(hipStreamCreateWithFlags(&s[i], hipStreamNonBlocking));
    hipEvent_t e0, e1, fork, join[NS];
    CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    CK(hipEventCreateWithFlags(&fork, hipEventDisableTiming));
    for (int i = 0; i < NS; ++i) CK(hipEventCreateWithFlags(&join[i], hipEventDisableTiming));

    auto L = [&](int k, hipStream_t st, int flags) {
        KArgs a = d[k].args;
        void *params[1] = {&a};
        if (flags < 0) CK(hipLaunchKernel((const void *)d[k].fn, d[k].grid, d[k].block, params, 0, st));
        else CK(hipExtLaunchKernel((const void *)d[k].fn, d[k].grid, d[k].block, params, 0, st, nullptr, nullptr, flags));
    };
    auto report = [&](const char *name, const std::function<void()> &round) {
        for (int i = 0; i < 5; ++i) round();
        CK(hipDeviceSynchronize());
        // clock conditioning
        auto t_end = std::chrono::steady_clock::now() + std::chrono::milliseconds(300);
        while (std::chrono::steady_clock::now() < t_end) { for (int i = 0; i < 4; ++i) round(); CK(hipStreamSynchronize(s[0])); }
        CK(hipDeviceSynchronize());
        const auto h0 = std::chrono::steady_clock::now();
        CK(hipEventRecord(e0, s[0]));
        for (int r = 0; r < reps; ++r) round();
        CK(hipEventRecord(e1, s[0]));
        const auto h1 = std::chrono::steady_clock::now();          // host time to ENQUEUE everything
        CK(hipEventSynchronize(e1));
        CK(hipDeviceSynchronize());
        const auto h2 = std::chrono::steady_clock::now();
        float ms = 0;
        CK(hipEventElapsedTime(&ms, e0, e1));
        const double us = ms * 1e3 / (reps * N);
        printf("%-14s %7.3f us/frame (events)  %5.1f%% of 8 TB/s   host enqueue %6.3f us/frame, wall %6.3f us/frame\n", name, us,
               alg / 8e12 * 1e6 / us * 100, std::chrono::duration<double, std::micro>(h1 - h0).count() / (reps * N),
               std::chrono::duration<double, std::micro>(h2 - h0).count() / (reps * N));
        fflush(stdout);
    };

    report("eager1", [&] { for (int k = 0; k < N; ++k) L(k, s[0], -1); });
    report("ext-flags0", [&] { for (int k = 0; k < N; ++k) L(k, s[0], 0); });
    report("anyorder", [&] { L(0, s[0], 0); for (int k = 1; k < N; ++k) L(k, s[0], hipExtAnyOrderLaunch); });
    for (int S : {2, 4, 8, 16}) {
        char nm[32]; snprintf(nm, sizeof nm, "eager%d", S);
        report(nm, [&] {
            CK(hipEventRecord(fork, s[0]));
            for (int i = 1; i < S; ++i) CK(hipStreamWaitEvent(s[i], fork, 0));
            for (int k = 0; k < N; ++k) L(k, s[k % S], -1);
            for (int i = 1; i < S; ++i) { CK(hipEventRecord(join[i], s[i])); CK(hipStreamWaitEvent(s[0], join[i], 0)); }
        });
    }
    // ---- graphs ----
    auto replay = [&](const char *name, hipGraph_t g) {
        hipGraphExec_t ex;
        CK(hipGraphInstantiate(&ex, g, nullptr, nullptr, 0));
        report(name, [&] { CK(hipGraphLaunch(ex, s[0])); });
        CK(hipGraphExecDestroy(ex));
        CK(hipGraphDestroy(g));
    };
    {
        hipGraph_t g;
        CK(hipStreamBeginCapture(s[0], hipStreamCaptureModeThreadLocal));
        for (int k = 0; k < N; ++k) L(k, s[0], -1);
        CK(hipStreamEndCapture(s[0], &g));
        replay("cap-chain", g);
    }
    {
        hipGraph_t g;
        CK(hipStreamBeginCapture(s[0], hipStreamCaptureModeThreadLocal));
        L(0, s[0], 0);
        for (int k = 1; k < N; ++k) L(k, s[0], hipExtAnyOrderLaunch);
        CK(hipStreamEndCapture(s[0], &g));
        replay("cap-any", g);
    }
    for (int S : {2, 4, 8, 16}) {
        hipGraph_t g;
        CK(hipStreamBeginCapture(s[0], hipStreamCaptureModeThreadLocal));
        CK(hipEventRecord(fork, s[0]));
        for (int i = 1; i < S; ++i) CK(hipStreamWaitEvent(s[i], fork, 0));
        for (int k = 0; k < N; ++k) L(k, s[k % S], -1);
        for (int i = 1; i < S; ++i) { CK(hipEventRecord(join[i], s[i])); CK(hipStreamWaitEvent(s[0], join[i], 0)); }
        CK(hipStreamEndCapture(s[0], &g));
        char nm[32]; snprintf(nm, sizeof nm, "cap-fork%d", S);
        replay(nm, g);
    }
    for (int B : {1, 2, 4, 8, 16, 64}) {
        hipGraph_t g;
        CK(hipGraphCreate(&g, 0));
        std::vector<hipGraphNode_t> nodes(N);
        for (int k = 0; k < N; ++k) {
            void *params[1] = {&d[k].args};
            hipKernelNodeParams np{};
            np.func = (void *)d[k].fn; np.gridDim = d[k].grid; np.blockDim = d[k].block; np.kernelParams = params;
            const hipGraphNode_t *dep = k >= B ? &nodes[k - B] : nullptr;
            CK(hipGraphAddKernelNode(&nodes[k], g, dep, dep ? 1 : 0, &np));
        }
        char nm[32]; snprintf(nm, sizeof nm, "graph-b%d", B);
        replay(nm, g);
    }
    // S chain graphs (each on the fast pre-built-packet path) replayed concurrently on S streams, fork/join by events
    for (int S : {2, 3, 4, 6, 8}) {
        std::vector<hipGraphExec_t> ex(S);
        std::vector<hipGraph_t> gs(S);
        for (int i = 0; i < S; ++i) {
            CK(hipGraphCreate(&gs[i], 0));
            hipGraphNode_t prev{};
            bool have = false;
            for (int k = i; k < N; k += S) {
                void *params[1] = {&d[k].args};
                hipKernelNodeParams np{};
                np.func = (void *)d[k].fn; np.gridDim = d[k].grid; np.blockDim = d[k].block; np.kernelParams = params;
                hipGraphNode_t node;
                CK(hipGraphAddKernelNode(&node, gs[i], have ? &prev : nullptr, have ? 1 : 0, &np));
                prev = node; have = true;
            }
            CK(hipGraphInstantiate(&ex[i], gs[i], nullptr, nullptr, 0));
        }
        char nm[32]; snprintf(nm, sizeof nm, "mchain%d", S);
        report(nm, [&] {
            CK(hipEventRecord(fork, s[0]));
            for (int i = 1; i < S; ++i) CK(hipStreamWaitEvent(s[i], fork, 0));
            for (int i = 0; i < S; ++i) CK(hipGraphLaunch(ex[i], s[i]));
            for (int i = 1; i < S; ++i) { CK(hipEventRecord(join[i], s[i])); CK(hipStreamWaitEvent(s[0], join[i], 0)); }
        });
        for (int i = 0; i < S; ++i) { CK(hipGraphExecDestroy(ex[i])); CK(hipGraphDestroy(gs[i])); }
    }
    // one batched launch of the same 64 frames (contiguous): the bound overlap can approach
    {
        LaunchDesc b;
        prepare_launch(pl, din, dout, N, 0, 0, &b);
        report("batched", [&] { KArgs a = b.args; void *params[1] = {&a}; CK(hipLaunchKernel((const void *)b.fn, b.grid, b.block, params, 0, s[0])); });
    }
    return 0;
}
