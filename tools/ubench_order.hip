// tools/ubench_order.hip -- does the ORDER in which blocks visit a frame matter?  Headline shape (8192x8192, f = 2, K = 4,
// 128-thread blocks, nt loads/stores), 1-D grid, block b -> (row, x-chunk) through different maps, and the row base kept in
// SGPRs (uniform per block) with 32-bit lane offsets.  Packed 8192-pixel rows put live rows 64 KiB apart; padding the rows by
// 256 px is worth 3 % (profiles/r02_probe_pitch.log), so the question is whether a visiting order recovers some of that.
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -Iinclude -I<pkg>/csrc tools/ubench_order.hip <pkg>/csrc/csic_host.cpp <pkg>/csrc/csic_png.cpp -lz -o tools/ubench_order
#include "csic_kernels.hip"

#include <cstdlib>
#include <vector>

using namespace csic;
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1);} } while (0)
__device__ __forceinline__ uint32_t pix(uint32_t px, const KArgs &a)
{
    const ChromaTerm t = chroma_term<R_FLOOR, F_ARGB>(px, a.mcb, a.mcr);
    return finish<F_ARGB>(px, a.my, t);
}

// MAP 0: row-major (x-chunk fastest)            b -> ro = b / XC, xc = b % XC
// MAP 1: strips of R rows, rows fastest          g = b / (XC*R), w = b % (XC*R): xc = w / R, ro = g*R + w % R
// MAP 2: column-major (rows fastest over the whole frame)
// MAP 3: XCD-contiguous: XCD (b & 7) streams its own eighth of the frame in row-major order
template <int BT, int MAP>
__global__ void __launch_bounds__(BT) k_order(KArgs a, int R)
{
    constexpr int K = 4;
    const int XC = a.Wo / (BT * K);
    const int b = blockIdx.x;
    int ro, xc;
    if (MAP == 0) { ro = b / XC; xc = b - ro * XC; }
    else if (MAP == 1) { const int g = b / (XC * R), w = b - g * (XC * R); xc = w / R; ro = g * R + (w - xc * R); }
    else if (MAP == 2) { xc = b / a.Ho; ro = b - xc * a.Ho; }
    else { const int nb = gridDim.x, lb = (b & 7) * (nb >> 3) + (b >> 3); ro = lb / XC; xc = lb - ro * XC; }
    const gin_t rp = (gin_t)(uintptr_t)a.in + (int64_t)(ro * 2) * a.W + xc * (BT * K * 2);     // uniform: SGPR base
    const gout_t op = (gout_t)(uintptr_t)a.out + (int64_t)ro * a.Wo + xc * (BT * K);
    const uint32_t t = threadIdx.x;
    uint32_t px[K];
#pragma unroll
    for (int k = 0; k < K; ++k) px[k] = ld1<true>(rp + (t + k * BT) * 2u);
#pragma unroll
    for (int k = 0; k < K; ++k) st1<true>(op + (t + k * BT), pix(px[k], a));
}

int main(int argc, char **argv)
{
    const int W = 8192, H = 8192, Wo = 4096, Ho = 4096, nring = 12, iters = 120;
    std::vector<uint32_t *> in, out;
    for (int i = 0; i < nring; ++i) {
        uint32_t *a, *b;
        CK(hipMalloc(&a, (size_t)W * H * 4)); CK(hipMalloc(&b, (size_t)Wo * Ho * 4));
        hipLaunchKernelGGL(k_synth, dim3(8192), dim3(256), 0, 0, a, (int64_t)W * H, (int64_t)i * W * H, 20250629u * 0x9E3779B9u);
        in.push_back(a); out.push_back(b);
    }
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    KArgs base; memset(&base, 0, sizeof base);
    base.W = W; base.H = H; base.Wo = Wo; base.Ho = Ho; base.my = base.mcb = base.mcr = 0xFF; base.f = 2; base.ip = W; base.op = Wo;
    CK(hipDeviceSynchronize());
    auto run = [&](const char *name, auto launch) {
        for (int i = 0; i < 20; ++i) launch(i % nring);
        CK(hipDeviceSynchronize());
        float best = 1e9f, sum = 0;
        for (int rep = 0; rep < 3; ++rep) {
            CK(hipEventRecord(e0, 0));
            for (int i = 0; i < iters; ++i) launch(i % nring);
            CK(hipEventRecord(e1, 0));
            CK(hipEventSynchronize(e1));
            float ms; CK(hipEventElapsedTime(&ms, e0, e1));
            ms /= iters; sum += ms; if (ms < best) best = ms;
        }
        CK(hipGetLastError());
        printf("%-52s %8.3f us (best %7.3f)  %5.1f%% of 8TB/s\n", name, sum / 3 * 1e3, best * 1e3, 201326592.0 / (sum / 3 * 1e-3) / 8e12 * 100);
        fflush(stdout);
    };
    auto args = [&](int i) { KArgs a = base; a.in = in[i]; a.out = out[i]; return a; };
    // warm the clocks
    for (int r = 0; r < 40; ++r) for (int i = 0; i < nring; ++i) hipLaunchKernelGGL((k_order<128, 0>), dim3(8 * 4096), dim3(128), 0, 0, args(i), 1);
    CK(hipDeviceSynchronize());
    for (int pass = 0; pass < 2; ++pass) {
        run("128thr row-major (saddr)", [&](int i) { hipLaunchKernelGGL((k_order<128, 0>), dim3(8 * 4096), dim3(128), 0, 0, args(i), 1); });
        for (int R : {2, 4, 8, 16, 32, 64, 256}) {
            char nm[64]; snprintf(nm, sizeof nm, "128thr strips of %d rows, rows fastest", R);
            run(nm, [&](int i) { hipLaunchKernelGGL((k_order<128, 1>), dim3(8 * 4096), dim3(128), 0, 0, args(i), R); });
        }
        run("128thr column-major", [&](int i) { hipLaunchKernelGGL((k_order<128, 2>), dim3(8 * 4096), dim3(128), 0, 0, args(i), 1); });
        run("128thr XCD-contiguous eighths", [&](int i) { hipLaunchKernelGGL((k_order<128, 3>), dim3(8 * 4096), dim3(128), 0, 0, args(i), 1); });
        run("256thr row-major (saddr)", [&](int i) { hipLaunchKernelGGL((k_order<256, 0>), dim3(4 * 4096), dim3(256), 0, 0, args(i), 1); });
        run("256thr strips of 8 rows", [&](int i) { hipLaunchKernelGGL((k_order<256, 1>), dim3(4 * 4096), dim3(256), 0, 0, args(i), 8); });
        run("64thr row-major (saddr)", [&](int i) { hipLaunchKernelGGL((k_order<64, 0>), dim3(16 * 4096), dim3(64), 0, 0, args(i), 1); });
    }
    (void)argc; (void)argv;
    return 0;
}
