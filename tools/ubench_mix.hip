// tools/ubench_mix.hip -- what does HBM deliver on this box for the read:write mixes the csic kernels have?
// One kernel, R 16-byte non-temporal loads and Wr 16-byte non-temporal stores per lane (each block reads one contiguous
// R*T*16-byte piece and writes one contiguous Wr*T*16-byte piece, like k_recon / k_f1flat), 256 MiB on the larger side,
// launches rotating over 8 source and 8 destination buffers so that the 256 MB Infinity Cache cannot serve them:
//   8:8  copy (k_f1flat: 4 B read + 4 B written per pixel)      8:2  the headline (4 + 1)      8:3  planar forward (4 + 1.5)
//   3:8  reconstruct of a 4:2:0 planar frame (1.5 + 4)          0:8  fill                      8:0  read only (sum kept in a lane)
// Prints GB/s of (bytes read + bytes written) per launch, the figure roofline.achieved uses.
// Build here, run on the GPU box:  hipcc -O3 --offload-arch=gfx950 tools/ubench_mix.hip -o tools/_bin/ubench_mix
//                                  gpurun -- 'timeout -k 10 120 tools/_bin/ubench_mix'      (profiles/r04_ubench_mix.log)
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)

template <int R, int Wr, int T>
__global__ void __launch_bounds__(T) k_mix(const uint32_t *src, uint32_t *dst, int64_t steps, uint32_t *sink)
{
    typedef const u32x4 __attribute__((address_space(1))) *vp;
    typedef u32x4 __attribute__((address_space(1))) *wp;
    const int64_t i = (int64_t)blockIdx.x * T + threadIdx.x;
    if (i >= steps) return;
    const int64_t rbase = (int64_t)blockIdx.x * T * R + threadIdx.x, wbase = (int64_t)blockIdx.x * T * Wr + threadIdx.x;
    u32x4 acc = {0, 0, 0, 0};
    u32x4 v[R > 0 ? R : 1];
#pragma unroll
    for (int k = 0; k < R; ++k) v[k] = __builtin_nontemporal_load((vp)(uintptr_t)(src + 4 * (rbase + k * T)));
#pragma unroll
    for (int k = 0; k < R; ++k) acc += v[k];
#pragma unroll
    for (int k = 0; k < Wr; ++k) {
        u32x4 o = acc + (uint32_t)(i + k);                           // every load feeds every store
        __builtin_nontemporal_store(o, (wp)(uintptr_t)(dst + 4 * (wbase + k * T)));
    }
    if (Wr == 0 && acc.x == 0x12345678u && acc.y == 0x9abcdef0u) *sink = acc.z;      // keeps the loads alive
}

template <int R, int Wr, int T>
static int run(const char *what, uint32_t *const *srcs, uint32_t *const *dsts, uint32_t *sink, int64_t big_bytes)
{
    const int big = R > Wr ? R : Wr;
    const int64_t steps = big_bytes / 16 / big;                  // lanes; each moves R + Wr 16-byte words
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    const unsigned grid = (unsigned)((steps + T - 1) / T);
    for (int r = 0; r < 3; ++r) k_mix<R, Wr, T><<<grid, T>>>(srcs[r % 8], dsts[r % 8], steps, sink);
    CK(hipDeviceSynchronize());
    const int reps = 48;
    CK(hipEventRecord(e0));
    for (int r = 0; r < reps; ++r) k_mix<R, Wr, T><<<grid, T>>>(srcs[r % 8], dsts[r % 8], steps, sink);
    CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
    float ms = 0; CK(hipEventElapsedTime(&ms, e0, e1));
    const double bytes = (double)steps * 16 * (R + Wr);
    printf("%-34s R:W = %d:%d  T=%3d  %7.1f us/launch  %7.1f GB/s  (%.3f of 8 TB/s)\n", what, R, Wr, T, ms * 1e3 / reps,
           bytes * reps / (ms * 1e-3) / 1e9, bytes * reps / (ms * 1e-3) / 8e12);
    fflush(stdout);
    return 0;
}

int main()
{
    const int64_t big = 256ll << 20;
    uint32_t *src[8], *dst[8], *sink;
    CK(hipMalloc(&sink, 4));
    for (int k = 0; k < 8; ++k) {
        CK(hipMalloc(&src[k], big)); CK(hipMalloc(&dst[k], big));
        CK(hipMemset(src[k], 0x5a, big)); CK(hipMemset(dst[k], 0, big));
    }
    for (int rep = 0; rep < 2; ++rep) {
        if (run<8, 8, 64>("copy (k_f1flat)", src, dst, sink, big)) return 1;
        if (run<8, 2, 64>("headline 4:1", src, dst, sink, big)) return 1;
        if (run<8, 3, 64>("planar forward 4:1.5", src, dst, sink, big)) return 1;
        if (run<3, 8, 64>("reconstruct 1.5:4", src, dst, sink, big)) return 1;
        if (run<3, 8, 256>("reconstruct 1.5:4", src, dst, sink, big)) return 1;
        if (run<2, 8, 64>("1:4", src, dst, sink, big)) return 1;
        if (run<0, 8, 64>("fill", src, dst, sink, big)) return 1;
        if (run<0, 8, 256>("fill", src, dst, sink, big)) return 1;
        if (run<8, 0, 64>("read only", src, dst, sink, big)) return 1;
        if (run<8, 0, 256>("read only", src, dst, sink, big)) return 1;
    }
    return 0;
}
