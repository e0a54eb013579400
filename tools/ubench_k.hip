// tools/ubench_k.hip -- developer micro-benchmark (not part of the product): pixels per lane (K) x block size for the headline
// kernel shape k_dec<floor, argb, f=2, hold 1, c>s, K, nt> on 8192x8192, one frame per launch over a ring of 16 frames,
// serial launches on one stream after 400 ms of clock conditioning.  The library ships K = 4.
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -Iinclude -I<pkg>/csrc tools/ubench_k.hip <pkg>/csrc/csic_host.cpp <pkg>/csrc/csic_png.cpp -lz -o tools/ubench_k
#include "csic_kernels.hip"

#include <chrono>
#include <cstdlib>
#include <vector>

using namespace csic;
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1);} } while (0)

template <int K>
static void run(const std::vector<uint32_t *> &in, const std::vector<uint32_t *> &out, int threads, int reps)
{
    const int W = 8192, H = 8192, Wo = 4096, Ho = 4096, nring = (int)in.size();
    const int lanes_x = Wo / K;
    const int bx = lanes_x < threads ? lanes_x : threads, by = threads / bx;
    KArgs a;
    memset(&a, 0, sizeof a);
    a.W = W; a.H = H; a.Wo = Wo; a.Ho = Ho; a.my = a.mcb = a.mcr = 0xFF; a.f = 2; a.ip = W; a.op = Wo;
    a.in_frame_px = (int64_t)W * H; a.out_frame_px = (int64_t)Wo * Ho; a.sc_shift = 1;
    a.bdx = bx; a.bdy = by;
    const dim3 grid(lanes_x / bx, (Ho + by - 1) / by, 1), block(bx, by, 1);
    a.row_step = grid.y * by;
    auto launch = [&](int i) { KArgs b = a; b.in = in[i % nring]; b.out = out[i % nring]; hipLaunchKernelGGL((k_dec<R_FLOOR, F_ARGB, 2, 1, false, K, true>), grid, block, 0, 0, b); };
    auto t_end = std::chrono::steady_clock::now() + std::chrono::milliseconds(400);
    int i = 0;
    while (std::chrono::steady_clock::now() < t_end) { for (int k = 0; k < 64; ++k) launch(i++); CK(hipDeviceSynchronize()); }
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    float best = 1e9f;
    for (int rep = 0; rep < 3; ++rep) {
        CK(hipEventRecord(e0, 0));
        for (int k = 0; k < reps; ++k) launch(k);
        CK(hipEventRecord(e1, 0));
        CK(hipEventSynchronize(e1));
        float ms; CK(hipEventElapsedTime(&ms, e0, e1));
        if (ms / reps < best) best = ms / reps;
    }
    CK(hipGetLastError());
    printf("K=%d  block %3d (%dx%d)  grid %ux%u   %7.3f us/frame  %5.1f%% of 8 TB/s\n", K, threads, bx, by, grid.x, grid.y, best * 1e3,
           201326592.0 / (best * 1e-3) / 8e12 * 100);
    fflush(stdout);
}

int main()
{
    const int nring = 16;
    std::vector<uint32_t *> in(nring), out(nring);
    for (int i = 0; i < nring; ++i) {
        CK(hipMalloc(&in[i], (size_t)8192 * 8192 * 4)); CK(hipMalloc(&out[i], (size_t)4096 * 4096 * 4));
        hipLaunchKernelGGL(k_synth, dim3(8192), dim3(256), 0, 0, in[i], (int64_t)8192 * 8192, (int64_t)i * 8192 * 8192, 20250629u * 0x9E3779B9u);
    }
    CK(hipDeviceSynchronize());
    for (int rep = 0; rep < 2; ++rep)
        for (int threads : {256, 128, 64}) {
            run<1>(in, out, threads, 1000);
            run<2>(in, out, threads, 1000);
            run<4>(in, out, threads, 1000);
            run<8>(in, out, threads, 1000);
        }
    return 0;
}
