// tools/ubench_rows.hip -- what bounds f = 4 / f = 8?  Reads ONLY the live rows (every f-th row) of a frame, three ways, and
// reports the rate over the bytes of those rows (the algorithmic read traffic):
//   dense16 : every lane loads 16 contiguous bytes (all of the row's bytes, the friendliest request shape), nothing written
//   lane4   : the shipped k_dec shape -- 4 loads of 4 bytes per lane at a stride of f*4 bytes across lanes, nothing written
//   lane4+w : the same plus the dense 4-byte output stores (the shipped kernel's memory shape without its arithmetic)
// If dense16 is as slow as lane4, the limit is the DRAM/fabric side of reading every f-th row; if dense16 is fast, the
// lane-strided request shape is.
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 tools/ubench_rows.hip -o tools/ubench_rows
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstdint>
#include <vector>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1);} } while (0)
typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));

__global__ void __launch_bounds__(256) k_fill(uint32_t *p, int64_t n)
{
    const int64_t s = (int64_t)gridDim.x * blockDim.x;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += s) p[i] = (uint32_t)(i * 2654435761u);
}

// grid: (W/4/256, live rows, frames)
__global__ void __launch_bounds__(256) k_dense16(const uint32_t *in, uint32_t *sink, int W, int f, int64_t frame_px)
{
    const uint32_t *rp = in + blockIdx.z * frame_px + (int64_t)(blockIdx.y * f) * W;
    const u32x4 v = __builtin_nontemporal_load(reinterpret_cast<const u32x4 *>(rp) + blockIdx.x * blockDim.x + threadIdx.x);
    if ((v.x ^ v.y ^ v.z ^ v.w) == 0x12345678u) sink[0] = v.x;
}

// grid: (Wo/(256*4), live rows, frames); lane loads 4 px spaced by the block width in the OUTPUT
template <bool WRITE>
__global__ void __launch_bounds__(256) k_lane4(const uint32_t *in, uint32_t *out, uint32_t *sink, int W, int Wo, int f, int64_t frame_px, int64_t oframe_px)
{
    const uint32_t *rp = in + blockIdx.z * frame_px + (int64_t)(blockIdx.y * f) * W;
    const int bd = blockDim.x;
    const int co0 = blockIdx.x * (bd * 4) + threadIdx.x;
    uint32_t px[4];
#pragma unroll
    for (int k = 0; k < 4; ++k) px[k] = __builtin_nontemporal_load(rp + (int64_t)(co0 + k * bd) * f);
    if (WRITE) {
        uint32_t *op = out + blockIdx.z * oframe_px + (int64_t)blockIdx.y * Wo;
#pragma unroll
        for (int k = 0; k < 4; ++k) __builtin_nontemporal_store(px[k] * 3u + 1u, op + co0 + k * bd);
    } else if ((px[0] ^ px[1] ^ px[2] ^ px[3]) == 0x12345678u) sink[0] = px[0];
}

// lane4 loads, then a transposition through LDS so that every lane stores 16 contiguous bytes (a wave writes 1 KiB contiguous
// instead of four separate 256-byte chunks): does the store granularity matter for the read/write mix?
__global__ void __launch_bounds__(256) k_lane4_lds16(const uint32_t *in, uint32_t *out, int W, int Wo, int f, int64_t frame_px, int64_t oframe_px)
{
    __shared__ uint32_t tile[1024];
    const uint32_t *rp = in + blockIdx.z * frame_px + (int64_t)(blockIdx.y * f) * W;
    const int bd = blockDim.x;
    const int co0 = blockIdx.x * (bd * 4) + threadIdx.x;
    uint32_t px[4];
#pragma unroll
    for (int k = 0; k < 4; ++k) px[k] = __builtin_nontemporal_load(rp + (int64_t)(co0 + k * bd) * f);
#pragma unroll
    for (int k = 0; k < 4; ++k) tile[threadIdx.x + k * bd] = px[k] * 3u + 1u;
    __syncthreads();
    const u32x4 v = *reinterpret_cast<const u32x4 *>(&tile[4 * threadIdx.x]);
    uint32_t *op = out + blockIdx.z * oframe_px + (int64_t)blockIdx.y * Wo + blockIdx.x * (bd * 4);
    __builtin_nontemporal_store(v, reinterpret_cast<u32x4 *>(op) + threadIdx.x);
}

// f >= 4: every lane loads the 16 bytes that start at its live pixel (the live pixel + 3 dead ones; at f = 4 the wave's loads are
// dense), 4 such loads per lane spaced by the block width, dense 4-byte stores: the shipped shape with a wider request
__global__ void __launch_bounds__(256) k_lane16_w(const uint32_t *in, uint32_t *out, int W, int Wo, int f, int64_t frame_px, int64_t oframe_px)
{
    const uint32_t *rp = in + blockIdx.z * frame_px + (int64_t)(blockIdx.y * f) * W;
    const int bd = blockDim.x;
    const int co0 = blockIdx.x * (bd * 4) + threadIdx.x;
    u32x4 px[4];
#pragma unroll
    for (int k = 0; k < 4; ++k) px[k] = __builtin_nontemporal_load(reinterpret_cast<const u32x4 *>(rp + (int64_t)(co0 + k * bd) * f));
    uint32_t *op = out + blockIdx.z * oframe_px + (int64_t)blockIdx.y * Wo;
#pragma unroll
    for (int k = 0; k < 4; ++k) __builtin_nontemporal_store(px[k].x * 3u + 1u, op + co0 + k * bd);
}

// load cache policies for the shipped shape (f = 2 only): POL 0 plain, 1 nt, 2 sc1, 3 sc0 sc1, 4 nt sc1, 5 nt sc0 sc1, 6 sc0
template <int POL> __device__ __forceinline__ uint32_t ld_pol(const uint32_t *p)
{
    uint32_t v;
    if (POL == 1) asm volatile("global_load_dword %0, %1, off nt" : "=v"(v) : "v"(p) : "memory");
    else if (POL == 2) asm volatile("global_load_dword %0, %1, off sc1" : "=v"(v) : "v"(p) : "memory");
    else if (POL == 3) asm volatile("global_load_dword %0, %1, off sc0 sc1" : "=v"(v) : "v"(p) : "memory");
    else if (POL == 4) asm volatile("global_load_dword %0, %1, off sc1 nt" : "=v"(v) : "v"(p) : "memory");
    else if (POL == 5) asm volatile("global_load_dword %0, %1, off sc0 sc1 nt" : "=v"(v) : "v"(p) : "memory");
    else if (POL == 6) asm volatile("global_load_dword %0, %1, off sc0" : "=v"(v) : "v"(p) : "memory");
    else asm volatile("global_load_dword %0, %1, off" : "=v"(v) : "v"(p) : "memory");
    return v;
}
template <int POL>
__global__ void __launch_bounds__(256) k_lane4_pol(const uint32_t *in, uint32_t *out, int W, int Wo, int f, int64_t frame_px, int64_t oframe_px)
{
    const uint32_t *rp = in + blockIdx.z * frame_px + (int64_t)(blockIdx.y * f) * W;
    const int bd = blockDim.x;
    const int co0 = blockIdx.x * (bd * 4) + threadIdx.x;
    uint32_t px[4];
#pragma unroll
    for (int k = 0; k < 4; ++k) px[k] = ld_pol<POL>(rp + (int64_t)(co0 + k * bd) * f);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    uint32_t *op = out + blockIdx.z * oframe_px + (int64_t)blockIdx.y * Wo;
#pragma unroll
    for (int k = 0; k < 4; ++k) __builtin_nontemporal_store(px[k] * 3u + 1u, op + co0 + k * bd);
}

int main()
{
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    uint32_t *sink; CK(hipMalloc(&sink, 64));
    auto run = [&](const char *name, double bytes, auto launch) {
        for (int i = 0; i < 10; ++i) launch(i);
        CK(hipDeviceSynchronize());
        float best = 1e9f;
        for (int rep = 0; rep < 3; ++rep) {
            CK(hipEventRecord(e0, 0));
            for (int i = 0; i < 40; ++i) launch(i);
            CK(hipEventRecord(e1, 0));
            CK(hipEventSynchronize(e1));
            float ms; CK(hipEventElapsedTime(&ms, e0, e1));
            if (ms / 40 < best) best = ms / 40;
        }
        CK(hipGetLastError());
        printf("%-58s %9.2f us  %7.1f GB/s  %5.1f%% of 8 TB/s\n", name, best * 1e3, bytes / (best * 1e-3) / 1e9, bytes / (best * 1e-3) / 8e12 * 100);
        fflush(stdout);
    };
    struct Shape { const char *name; int W, H, frames, nring; };
    const Shape shapes[] = {{"8192x8192 x 4 frames", 8192, 8192, 4, 3}, {"3840x2160 x 64 frames", 3840, 2160, 64, 3}, {"512x512 x 1024 frames", 512, 512, 1024, 3}};
    for (const Shape &sh : shapes) {
        const int64_t frame_px = (int64_t)sh.W * sh.H;
        std::vector<uint32_t *> in(sh.nring), out(sh.nring);
        for (int i = 0; i < sh.nring; ++i) {
            CK(hipMalloc(&in[i], frame_px * sh.frames * 4)); CK(hipMalloc(&out[i], frame_px * sh.frames * 4));
            hipLaunchKernelGGL(k_fill, dim3(4096), dim3(256), 0, 0, in[i], frame_px * sh.frames);
        }
        CK(hipDeviceSynchronize());
        for (int f : {1, 2, 4, 8}) {
            const int Wo = sh.W / f, live = sh.H / f;
            auto pick = [](int lanes) { for (int bt : {256, 240, 192, 128, 120, 96, 64, 60, 32, 30, 16}) if (lanes % bt == 0) return bt; return 0; };
            const int bt16 = pick(sh.W / 4), bt4 = (Wo % 4 == 0) ? pick(Wo / 4) : 0;
            if (!bt16) continue;
            const double rd = 4.0 * sh.W * live * sh.frames, wr = 4.0 * Wo * live * sh.frames;
            char nm[128];
            snprintf(nm, sizeof nm, "%s f=%d dense16 (read only)", sh.name, f);
            run(nm, rd, [&](int i) { hipLaunchKernelGGL(k_dense16, dim3(sh.W / 4 / bt16, live, sh.frames), dim3(bt16), 0, 0, in[i % sh.nring], sink, sh.W, f, frame_px); });
            if (bt4) {
                snprintf(nm, sizeof nm, "%s f=%d lane4 (read only)", sh.name, f);
                run(nm, rd, [&](int i) { hipLaunchKernelGGL((k_lane4<false>), dim3(Wo / 4 / bt4, live, sh.frames), dim3(bt4), 0, 0, in[i % sh.nring], out[i % sh.nring], sink, sh.W, Wo, f, frame_px, (int64_t)Wo * live); });
                snprintf(nm, sizeof nm, "%s f=%d lane4 + stores (rate over read+write)", sh.name, f);
                run(nm, rd + wr, [&](int i) { hipLaunchKernelGGL((k_lane4<true>), dim3(Wo / 4 / bt4, live, sh.frames), dim3(bt4), 0, 0, in[i % sh.nring], out[i % sh.nring], sink, sh.W, Wo, f, frame_px, (int64_t)Wo * live); });
                if (f == 2 && sh.W == 8192) {
                    const char *pn[7] = {"plain", "nt", "sc1", "sc0 sc1", "sc1 nt", "sc0 sc1 nt", "sc0"};
                    auto go = [&](int pol, auto kern) { snprintf(nm, sizeof nm, "%s f=2 lane4 loads [%s] + nt stores", sh.name, pn[pol]);
                        run(nm, rd + wr, [&](int i) { hipLaunchKernelGGL(kern, dim3(Wo / 4 / bt4, live, sh.frames), dim3(bt4), 0, 0, in[i % sh.nring], out[i % sh.nring], sh.W, Wo, f, frame_px, (int64_t)Wo * live); }); };
                    go(0, k_lane4_pol<0>); go(1, k_lane4_pol<1>); go(2, k_lane4_pol<2>); go(3, k_lane4_pol<3>); go(4, k_lane4_pol<4>); go(5, k_lane4_pol<5>); go(6, k_lane4_pol<6>);
                }
                if (f >= 4) {
                    snprintf(nm, sizeof nm, "%s f=%d 16-byte loads at the live pixel + 4-byte stores", sh.name, f);
                    run(nm, rd + wr, [&](int i) { hipLaunchKernelGGL(k_lane16_w, dim3(Wo / 4 / bt4, live, sh.frames), dim3(bt4), 0, 0, in[i % sh.nring], out[i % sh.nring], sh.W, Wo, f, frame_px, (int64_t)Wo * live); });
                }
                snprintf(nm, sizeof nm, "%s f=%d lane4 + LDS transpose + 16-byte stores", sh.name, f);
                run(nm, rd + wr, [&](int i) { hipLaunchKernelGGL(k_lane4_lds16, dim3(Wo / 4 / bt4, live, sh.frames), dim3(bt4), 0, 0, in[i % sh.nring], out[i % sh.nring], sh.W, Wo, f, frame_px, (int64_t)Wo * live); });
            }
        }
        for (int i = 0; i < sh.nring; ++i) { CK(hipFree(in[i])); CK(hipFree(out[i])); }
    }
    return 0;
}
