// graph_test.hip -- the frame-graph C ABI (csic_frame_graph_*) driven from native host code with the HIP runtime, no
// Python in between: what a C++ service would write.  Built with hipcc and run by tests/test_cpp_host.py (-m gpu).
//   * 6 frames in separate hipMalloc'd buffers, reference outputs from csic_process_device
//   * CSIC_FRAME_GRAPH_HIP (default chains) and CSIC_FRAME_GRAPH_DIRECT: launch on a stream with a producer in front
//     (hipMemcpyAsync refills the inputs) and a consumer behind (checksum kernel), three times in a row without a host
//     synchronisation in between; then host-ordered submit / wait
//   * every output frame compared with the reference by checksum (order-sensitive, csic_checksum_device)
#include <hip/hip_runtime.h>

#include <cstdio>
#include <vector>

#include "csic.h"

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { std::printf("FAIL %s: %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)
#define CS(x) do { int s_ = (x); if (s_ != CSIC_OK) { std::printf("FAIL %s: %s\n", #x, csic_last_error()); return 1; } } while (0)

int main()
{
    const int W = 640, H = 64, n = 6;
    csic_params p;
    CS(csic_params_default(&p, W, H));
    p.chroma_a = 2; p.chroma_b = 0; p.factor = 2; p.y_bits = 5; p.cb_bits = 4; p.cr_bits = 4;
    csic_plan *plan = nullptr;
    CS(csic_plan_create(&p, 0, &plan));
    int32_t wo = 0, ho = 0;
    CS(csic_out_dims(&p, &wo, &ho));
    const size_t ipx = (size_t)W * H, opx = (size_t)wo * ho;

    hipStream_t st;
    CK(hipStreamCreateWithFlags(&st, hipStreamNonBlocking));
    std::vector<void *> src(n), in(n), out(n), ref(n);
    std::vector<const void *> cin(n);
    std::vector<uint64_t> want(n);
    for (int k = 0; k < n; ++k) {
        CK(hipMalloc(&src[k], ipx * 4)); CK(hipMalloc(&in[k], ipx * 4)); CK(hipMalloc(&out[k], opx * 4)); CK(hipMalloc(&ref[k], opx * 4));
        cin[k] = in[k];
        CS(csic_synth_frame_device(src[k], (int64_t)ipx, 1000 * k, 7, st));
        CS(csic_process_device(plan, src[k], ref[k], st));
    }
    CK(hipStreamSynchronize(st));
    for (int k = 0; k < n; ++k) CS(csic_checksum_device(ref[k], (int64_t)opx, &want[k], st));

    int fails = 0;
    for (int backend : {CSIC_FRAME_GRAPH_HIP, CSIC_FRAME_GRAPH_DIRECT, CSIC_FRAME_GRAPH_FUSED}) {
        csic_frame_graph *g = nullptr;
        CS(csic_frame_graph_create_ex(plan, cin.data(), out.data(), n, 0, backend, &g));
        int32_t nf = 0, nb = 0;
        CS(csic_frame_graph_count(g, &nf, &nb));
        const int ordered = csic_frame_graph_stream_ordered(g);
        for (int rep = 0; rep < 3; ++rep) {
            for (int k = 0; k < n; ++k) {
                CK(hipMemsetAsync(out[k], 0, opx * 4, st));
                CK(hipMemcpyAsync(in[k], src[(k + rep) % n], ipx * 4, hipMemcpyDeviceToDevice, st));   // producer
            }
            CS(csic_frame_graph_launch(g, st));                                                          // asynchronous when `ordered`
            for (int k = 0; k < n; ++k) {                                                                // consumer, same stream
                uint64_t got = 0;
                CS(csic_checksum_device(out[k], (int64_t)opx, &got, st));
                if (got != want[(k + rep) % n]) { std::printf("FAIL backend %d rep %d frame %d\n", backend, rep, k); ++fails; }
            }
        }
        if (backend == CSIC_FRAME_GRAPH_DIRECT) {
            CK(hipStreamSynchronize(st));
            for (int k = 0; k < n; ++k) { CK(hipMemset(out[k], 0, opx * 4)); CK(hipMemcpy(in[k], src[k], ipx * 4, hipMemcpyDeviceToDevice)); }
            CK(hipDeviceSynchronize());          // host-ordered contract: the inputs are READY before submit (a D2D hipMemcpy returns early)
            int64_t t0 = -1, t1 = -1;
            CS(csic_frame_graph_submit(g, &t0));
            CS(csic_frame_graph_submit(g, &t1));
            CS(csic_frame_graph_wait(g, t1));
            if (t1 != t0 + 1) { std::printf("FAIL tickets %lld %lld\n", (long long)t0, (long long)t1); ++fails; }
            for (int k = 0; k < n; ++k) {
                uint64_t got = 0;
                CS(csic_checksum_device(out[k], (int64_t)opx, &got, st));
                if (got != want[k]) { std::printf("FAIL submit/wait frame %d\n", k); ++fails; }
            }
        }
        std::printf("backend %s: %d frames, %d chain(s)/queue(s), launch %s: ok\n", backend == CSIC_FRAME_GRAPH_HIP ? "HIP" : backend == CSIC_FRAME_GRAPH_DIRECT ? "DIRECT" : "FUSED", nf, nb,
                    ordered == 1 ? "ordered with the stream on the device" : "host-ordered");
        CS(csic_frame_graph_destroy(g));
    }
    CS(csic_plan_destroy(plan));
    std::printf(fails ? "%d check(s) failed\n" : "all checks passed\n", fails);
    return fails ? 1 : 0;
}
