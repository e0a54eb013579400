// tools/ubench.hip -- developer micro-benchmark for the headline shape (8192x8192, f=2).
// Not part of the product: it includes the product's device functions and times experimental
// load/store/grid shapes next to the shipped kernels so that tuning decisions are measured.
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -Iinclude -I<pkg>/csrc tools/ubench.hip <pkg>/csrc/csic_host.cpp -o tools/ubench
#include "csic_kernels.hip"

#include <cstdlib>
#include <vector>

using namespace csic;
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1);} } while (0)
template <bool COMPUTE> __device__ __forceinline__ uint32_t pix(uint32_t px, const KArgs &a)
{
    if (!COMPUTE) return px;
    const ChromaTerm t = chroma_term<R_FLOOR, F_ARGB>(px, a.mcb, a.mcr);
    return finish<F_ARGB>(px, a.my, t);
}

// full copy, 16 B per lane, grid-stride
__global__ void __launch_bounds__(256) k_copy(const u32x4 *in, u32x4 *out, int64_t n)
{
    const int64_t stride = (int64_t)gridDim.x * blockDim.x;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) out[i] = in[i];
}

// E1: lane owns 32 contiguous input bytes (2 x dwordx4) -> 4 output px (1 x dwordx4)
template <bool NTL, bool NTS, bool COMPUTE>
__global__ void __launch_bounds__(256) k_e1(KArgs a)
{
    const int x = blockIdx.x * blockDim.x + threadIdx.x;
    if (x >= a.Wo / 4) return;
    const int step = gridDim.y * blockDim.y;
    for (int ro = blockIdx.y * blockDim.y + threadIdx.y; ro < a.Ho; ro += step) {
        const uint32_t *rp = a.in + (int64_t)(ro * 2) * a.W + x * 8;
        const u32x4 p = ld4<NTL>(rp), q = ld4<NTL>(rp + 4);
        u32x4 o = {pix<COMPUTE>(p.x, a), pix<COMPUTE>(p.z, a), pix<COMPUTE>(q.x, a), pix<COMPUTE>(q.z, a)};
        st4<NTS>(a.out + (int64_t)ro * a.Wo + x * 4, o);
    }
}

// E2: lane loads dwordx4 from two wave-contiguous 1 KiB chunks -> 2 x dwordx2 stores (all dense)
template <bool NTL, bool NTS, bool COMPUTE>
__global__ void __launch_bounds__(256) k_e2(KArgs a)
{
    // block covers 2 * 256 * 4 input px = 2048 input px = 1024 output px
    const int t = threadIdx.x;
    const int step = gridDim.y;
    for (int ro = blockIdx.y; ro < a.Ho; ro += step) {
        const uint32_t *rp = a.in + (int64_t)(ro * 2) * a.W + blockIdx.x * 2048;
        uint32_t *op = a.out + (int64_t)ro * a.Wo + blockIdx.x * 1024;
        const u32x4 p = ld4<NTL>(rp + t * 4), q = ld4<NTL>(rp + 1024 + t * 4);
        u32x2 o0 = {pix<COMPUTE>(p.x, a), pix<COMPUTE>(p.z, a)};
        u32x2 o1 = {pix<COMPUTE>(q.x, a), pix<COMPUTE>(q.z, a)};
        st2<NTS>(op + t * 2, o0);
        st2<NTS>(op + 512 + t * 2, o1);
    }
}

// E3: K dword loads per lane, lane-contiguous in the output (shape of the shipped k_dec)
template <int K, bool NTL, bool NTS, bool COMPUTE>
__global__ void __launch_bounds__(256) k_e3(KArgs a)
{
    const int co0 = blockIdx.x * (256 * K) + threadIdx.x;
    const int step = gridDim.y;
    for (int ro = blockIdx.y; ro < a.Ho; ro += step) {
        const uint32_t *rp = a.in + (int64_t)(ro * 2) * a.W;
        uint32_t *op = a.out + (int64_t)ro * a.Wo;
        uint32_t px[K];
#pragma unroll
        for (int k = 0; k < K; ++k) px[k] = ld1<NTL>(rp + (co0 + k * 256) * 2);
#pragma unroll
        for (int k = 0; k < K; ++k) st1<NTS>(op + co0 + k * 256, pix<COMPUTE>(px[k], a));
    }
}

// E4: like E1 but each lane handles R rows at once (R independent 32-byte loads in flight)
template <int R, bool NTL, bool NTS, bool COMPUTE>
__global__ void __launch_bounds__(256) k_e4(KArgs a)
{
    const int x = blockIdx.x * blockDim.x + threadIdx.x;
    if (x >= a.Wo / 4) return;
    const int step = gridDim.y * R;
    for (int ro = blockIdx.y * R; ro < a.Ho; ro += step) {
        u32x4 p[R], q[R];
#pragma unroll
        for (int r = 0; r < R; ++r) {
            const uint32_t *rp = a.in + (int64_t)((ro + r) * 2) * a.W + x * 8;
            p[r] = ld4<NTL>(rp); q[r] = ld4<NTL>(rp + 4);
        }
#pragma unroll
        for (int r = 0; r < R; ++r) {
            u32x4 o = {pix<COMPUTE>(p[r].x, a), pix<COMPUTE>(p[r].z, a), pix<COMPUTE>(q[r].x, a), pix<COMPUTE>(q[r].z, a)};
            st4<NTS>(a.out + (int64_t)(ro + r) * a.Wo + x * 4, o);
        }
    }
}

__global__ void __launch_bounds__(256) k_copy_nt(const uint32_t *in, uint32_t *out, int64_t n)
{
    const int64_t stride = (int64_t)gridDim.x * blockDim.x;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) st4<true>(out + 4 * i, ld4<true>(in + 4 * i));
}

// E6: E1 with a 1-D grid remapped so that each XCD (blocks b, b+8, ...) streams one contiguous 1/8 of the frame
template <bool NTL, bool NTS, bool COMPUTE>
__global__ void __launch_bounds__(256) k_e6(KArgs a)
{
    const int nb = gridDim.x;                       // 16384 = 4 blocks/row * 4096 rows
    const int b = blockIdx.x;
    const int lb = (b & 7) * (nb >> 3) + (b >> 3);  // logical block id in memory order
    const int ro = lb >> 2, xb = lb & 3;
    const int x = xb * 256 + threadIdx.x;
    const uint32_t *rp = a.in + (int64_t)(ro * 2) * a.W + x * 8;
    const u32x4 p = ld4<NTL>(rp), q = ld4<NTL>(rp + 4);
    u32x4 o = {pix<COMPUTE>(p.x, a), pix<COMPUTE>(p.z, a), pix<COMPUTE>(q.x, a), pix<COMPUTE>(q.z, a)};
    st4<NTS>(a.out + (int64_t)ro * a.Wo + x * 4, o);
}

// E7: E1 with bigger blocks
template <bool NTL, bool NTS, bool COMPUTE>
__global__ void __launch_bounds__(1024) k_e7(KArgs a)
{
    const int x = blockIdx.x * blockDim.x + threadIdx.x;
    if (x >= a.Wo / 4) return;
    const int ro = blockIdx.y;
    const uint32_t *rp = a.in + (int64_t)(ro * 2) * a.W + x * 8;
    const u32x4 p = ld4<NTL>(rp), q = ld4<NTL>(rp + 4);
    u32x4 o = {pix<COMPUTE>(p.x, a), pix<COMPUTE>(p.z, a), pix<COMPUTE>(q.x, a), pix<COMPUTE>(q.z, a)};
    st4<NTS>(a.out + (int64_t)ro * a.Wo + x * 4, o);
}

// E8: f=1 shape (x4 load -> 4 px -> x4 store), full 8192x8192 frame, out buffer = another input-sized buffer
template <bool NTL, bool NTS, int MODE>   // MODE 0 = copy, 1 = full 4:4:4 pipeline, 2 = 4:2:2 (chroma shared by pairs)
__global__ void __launch_bounds__(256) k_e8(KArgs a)
{
    const int x4 = blockIdx.x * blockDim.x + threadIdx.x;
    const int row = blockIdx.y;
    const int64_t base = (int64_t)row * a.W + 4 * x4;
    const u32x4 p = ld4<NTL>(a.in + base);
    u32x4 o;
    if (MODE == 0) o = p;
    else if (MODE == 1) { o.x = pix<true>(p.x, a); o.y = pix<true>(p.y, a); o.z = pix<true>(p.z, a); o.w = pix<true>(p.w, a); }
    else {
        const ChromaTerm t0 = chroma_term<R_FLOOR, F_ARGB>(p.x, a.mcb, a.mcr), t1 = chroma_term<R_FLOOR, F_ARGB>(p.z, a.mcb, a.mcr);
        o.x = finish<F_ARGB>(p.x, a.my, t0); o.y = finish<F_ARGB>(p.y, a.my, t0);
        o.z = finish<F_ARGB>(p.z, a.my, t1); o.w = finish<F_ARGB>(p.w, a.my, t1);
    }
    st4<NTS>(a.out + base, o);
}

// E9: f=1, K x4-groups per lane spaced by the block width (more bytes in flight per wave)
template <int K, bool NT>
__global__ void __launch_bounds__(256) k_e9(KArgs a)
{
    const int x0 = blockIdx.x * (256 * K) + threadIdx.x;
    const int row = blockIdx.y;
    const int64_t base = (int64_t)row * a.W;
    u32x4 p[K];
#pragma unroll
    for (int k = 0; k < K; ++k) p[k] = ld4<NT>(a.in + base + 4 * (x0 + k * 256));
#pragma unroll
    for (int k = 0; k < K; ++k) {
        u32x4 o = {pix<true>(p[k].x, a), pix<true>(p[k].y, a), pix<true>(p[k].z, a), pix<true>(p[k].w, a)};
        st4<NT>(a.out + base + 4 * (x0 + k * 256), o);
    }
}

// ceilings: read-only (sum into one dword per wave so nothing is DCE'd), write-only
__global__ void __launch_bounds__(256) k_read_nt(const uint32_t *in, uint32_t *sink, int64_t n4)
{
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= n4) return;
    const u32x4 v = ld4<true>(in + 4 * i);
    const uint32_t s = v.x ^ v.y ^ v.z ^ v.w;
    if (s == 0x12345678u) sink[0] = s;             // practically never
}
__global__ void __launch_bounds__(256) k_write_nt(uint32_t *out, int64_t n4, uint32_t v)
{
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= n4) return;
    const u32x4 o = {v, v + 1, v + 2, v + 3};
    st4<true>(out + 4 * i, o);
}
// E10: shipped-shape dword kernel but reading CONTIGUOUS rows (row stride W instead of 2W): does skipping cost?
template <int K>
__global__ void __launch_bounds__(256) k_e10(KArgs a, int row_mul)
{
    const int co0 = blockIdx.x * (256 * K) + threadIdx.x;
    const int ro = blockIdx.y;
    const uint32_t *rp = a.in + (int64_t)(ro * row_mul) * a.W;
    uint32_t *op = a.out + (int64_t)ro * a.Wo;
    uint32_t px[K];
#pragma unroll
    for (int k = 0; k < K; ++k) px[k] = ld1<true>(rp + (co0 + k * 256) * 2);
#pragma unroll
    for (int k = 0; k < K; ++k) st1<true>(op + co0 + k * 256, pix<true>(px[k], a));
}

// E11: decimation by F with K loads per lane; LW = load width in dwords (1 or 4; only element 0 is used)
template <int F, int K, int LW, bool NTL, bool NTS>
__global__ void __launch_bounds__(256) k_e11(KArgs a)
{
    const int co0 = blockIdx.x * (256 * K) + threadIdx.x;
    const int ro = blockIdx.y;
    const uint32_t *rp = a.in + (int64_t)(ro * F) * a.W;
    uint32_t *op = a.out + (int64_t)ro * a.Wo;
    uint32_t px[K];
#pragma unroll
    for (int k = 0; k < K; ++k) {
        const int co = co0 + k * 256;
        if (LW == 4) { const u32x4 v = ld4<NTL>(rp + co * F); px[k] = v.x ^ (v.w & 0u); asm volatile("" ::"v"(v.y), "v"(v.z), "v"(v.w)); }
        else px[k] = ld1<NTL>(rp + co * F);
    }
#pragma unroll
    for (int k = 0; k < K; ++k) st1<NTS>(op + co0 + k * 256, pix<true>(px[k], a));
}

// E12: E3-shape kernel (K dword nt loads) with the store cache policy chosen through inline asm
//   POL 0 = nt (builtin)   1 = sc1   2 = sc0 sc1   3 = sc1 nt   4 = sc0 sc1 nt   5 = plain
template <int POL> __device__ __forceinline__ void st_pol(uint32_t *p, uint32_t v)
{
    if (POL == 0) __builtin_nontemporal_store(v, p);
    else if (POL == 1) asm volatile("global_store_dword %0, %1, off sc1" ::"v"(p), "v"(v) : "memory");
    else if (POL == 2) asm volatile("global_store_dword %0, %1, off sc0 sc1" ::"v"(p), "v"(v) : "memory");
    else if (POL == 3) asm volatile("global_store_dword %0, %1, off sc1 nt" ::"v"(p), "v"(v) : "memory");
    else if (POL == 4) asm volatile("global_store_dword %0, %1, off sc0 sc1 nt" ::"v"(p), "v"(v) : "memory");
    else *p = v;
}
template <int K, int POL>
__global__ void __launch_bounds__(256) k_e12(KArgs a)
{
    const int co0 = blockIdx.x * (256 * K) + threadIdx.x;
    const int ro = blockIdx.y;
    const uint32_t *rp = a.in + (int64_t)(ro * 2) * a.W;
    uint32_t *op = a.out + (int64_t)ro * a.Wo;
    uint32_t px[K];
#pragma unroll
    for (int k = 0; k < K; ++k) px[k] = ld1<true>(rp + (co0 + k * 256) * 2);
#pragma unroll
    for (int k = 0; k < K; ++k) st_pol<POL>(op + co0 + k * 256, pix<true>(px[k], a));
}

// E13: memory shape of the AVG f=2 kernel (2 rows x 16 B in, 8 B out per lane) without the arithmetic
__global__ void __launch_bounds__(256) k_e13(KArgs a)
{
    const int x4 = blockIdx.x * 256 + threadIdx.x;
    const int tr = blockIdx.y;
    const u32x4 p = ld4<true>(a.in + (int64_t)(tr * 2) * a.W + 4 * x4);
    const u32x4 q = ld4<true>(a.in + (int64_t)(tr * 2 + 1) * a.W + 4 * x4);
    const u32x2 o = {p.x ^ p.y ^ q.x ^ q.y, p.z ^ p.w ^ q.z ^ q.w};
    st2<true>(a.out + (int64_t)tr * a.Wo + 2 * x4, o);
}

struct Bench {
    int W = 8192, H = 8192, Wo = 4096, Ho = 4096;
    int nring = 6, iters = 60;
    std::vector<uint32_t *> in, out;
    hipEvent_t e0, e1;
    KArgs base;
    void init()
    {
        for (int i = 0; i < nring; ++i) {
            uint32_t *a, *b;
            CK(hipMalloc(&a, (size_t)W * H * 4)); CK(hipMalloc(&b, (size_t)Wo * Ho * 4));
            hipLaunchKernelGGL(k_synth, dim3(8192), dim3(256), 0, 0, a, (int64_t)W * H, (int64_t)i * W * H, 20250629u * 0x9E3779B9u);
            in.push_back(a); out.push_back(b);
        }
        CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
        memset(&base, 0, sizeof base);
        base.W = W; base.H = H; base.Wo = Wo; base.Ho = Ho; base.my = base.mcb = base.mcr = 0xFF;
        base.f = 2; base.bdx = 256; base.bdy = 1; base.row_step = 8192; base.ip = W; base.op = Wo; base.in_frame_px = (int64_t)W * H; base.out_frame_px = (int64_t)Wo * Ho;
        CK(hipDeviceSynchronize());
    }
    template <class F> void run(const char *name, F launch, double bytes = 201326592.0)
    {
        for (int i = 0; i < 10; ++i) launch(i % nring);
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
        printf("%-44s %8.2f us (best %7.2f)  %7.1f GB/s alg  %5.1f%% of 8TB/s\n", name, sum / 3 * 1e3, best * 1e3,
               bytes / (sum / 3 * 1e-3) / 1e9, bytes / (sum / 3 * 1e-3) / 8e12 * 100);
        fflush(stdout);
    }
    KArgs args(int i) { KArgs a = base; a.in = in[i]; a.out = out[i]; return a; }
};

#define L(kern, grid, block) [&](int i) { hipLaunchKernelGGL(kern, grid, block, 0, 0, B.args(i)); }

int main()
{
    Bench B; B.init();
    // ceilings
    B.run("copy 256MiB->256MiB (x4, 2048 blk)", [&](int i) { hipLaunchKernelGGL(k_copy, dim3(2048), dim3(256), 0, 0, (const u32x4 *)B.in[i], (u32x4 *)B.in[(i + 1) % B.nring], (int64_t)B.W * B.H / 4); }, 2.0 * 268435456.0);
    B.run("copy 256MiB->256MiB (x4, 65536 blk)", [&](int i) { hipLaunchKernelGGL(k_copy, dim3(65536), dim3(256), 0, 0, (const u32x4 *)B.in[i], (u32x4 *)B.in[(i + 1) % B.nring], (int64_t)B.W * B.H / 4); }, 2.0 * 268435456.0);
    // shipped kernels (as launched by the library: 256x1 blocks, gy = Ho)
    B.run("shipped k_dec K4        grid 4x4096", L((k_dec<R_FLOOR, F_ARGB, 2, 1, false, 4, false>), dim3(4, 4096), dim3(256)));
    B.run("shipped k_dec2v var1    grid 8x4096", L((k_dec2v<R_FLOOR, F_ARGB, 1, false>), dim3(8, 4096), dim3(256)));
    B.run("shipped k_dec2v var2    grid 4x4096", L((k_dec2v<R_FLOOR, F_ARGB, 2, false>), dim3(4, 4096), dim3(256)));
    // E1: memory-only vs compute, nt flags
    B.run("E1 nocompute            grid 4x4096", L((k_e1<false, false, false>), dim3(4, 4096), dim3(256)));
    B.run("E1 compute              grid 4x4096", L((k_e1<false, false, true>), dim3(4, 4096), dim3(256)));
    B.run("E1 compute ntL          grid 4x4096", L((k_e1<true, false, true>), dim3(4, 4096), dim3(256)));
    B.run("E1 compute ntS          grid 4x4096", L((k_e1<false, true, true>), dim3(4, 4096), dim3(256)));
    B.run("E1 compute ntL ntS      grid 4x4096", L((k_e1<true, true, true>), dim3(4, 4096), dim3(256)));
    B.run("E1 nocompute ntL ntS    grid 4x4096", L((k_e1<true, true, false>), dim3(4, 4096), dim3(256)));
    // E1 grid shapes (row loop)
    B.run("E1 compute              grid 4x2048", L((k_e1<false, false, true>), dim3(4, 2048), dim3(256)));
    B.run("E1 compute              grid 4x1024", L((k_e1<false, false, true>), dim3(4, 1024), dim3(256)));
    B.run("E1 compute              grid 4x512 ", L((k_e1<false, false, true>), dim3(4, 512), dim3(256)));
    B.run("E1 compute ntL ntS      grid 4x512 ", L((k_e1<true, true, true>), dim3(4, 512), dim3(256)));
    B.run("E1 compute blk 64x4     grid 16x1024", L((k_e1<false, false, true>), dim3(16, 1024), dim3(64, 4)));
    // E2 dense two-chunk
    B.run("E2 nocompute            grid 4x4096", L((k_e2<false, false, false>), dim3(4, 4096), dim3(256)));
    B.run("E2 compute              grid 4x4096", L((k_e2<false, false, true>), dim3(4, 4096), dim3(256)));
    B.run("E2 compute ntL ntS      grid 4x4096", L((k_e2<true, true, true>), dim3(4, 4096), dim3(256)));
    // E3 dword loads
    B.run("E3 K2 compute           grid 8x4096", L((k_e3<2, false, false, true>), dim3(8, 4096), dim3(256)));
    B.run("E3 K4 compute           grid 4x4096", L((k_e3<4, false, false, true>), dim3(4, 4096), dim3(256)));
    B.run("E3 K8 compute           grid 2x4096", L((k_e3<8, false, false, true>), dim3(2, 4096), dim3(256)));
    B.run("E3 K16 compute          grid 1x4096", L((k_e3<16, false, false, true>), dim3(1, 4096), dim3(256)));
    B.run("E3 K8 nocompute         grid 2x4096", L((k_e3<8, false, false, false>), dim3(2, 4096), dim3(256)));
    B.run("E3 K8 compute ntL ntS   grid 2x4096", L((k_e3<8, true, true, true>), dim3(2, 4096), dim3(256)));
    // E4 multi-row ILP
    B.run("E4 R2 compute           grid 4x2048", L((k_e4<2, false, false, true>), dim3(4, 2048), dim3(256)));
    B.run("E4 R4 compute           grid 4x1024", L((k_e4<4, false, false, true>), dim3(4, 1024), dim3(256)));
    B.run("E4 R4 compute ntL ntS   grid 4x1024", L((k_e4<4, true, true, true>), dim3(4, 1024), dim3(256)));
    B.run("E4 R4 nocompute         grid 4x1024", L((k_e4<4, false, false, false>), dim3(4, 1024), dim3(256)));
    B.run("E4 R2 compute           grid 4x512 ", L((k_e4<2, false, false, true>), dim3(4, 512), dim3(256)));

    // ---- round 2 of experiments ----
    B.run("copyNT 256MiB->256MiB (x4, 65536 blk)", [&](int i) { hipLaunchKernelGGL(k_copy_nt, dim3(65536), dim3(256), 0, 0, (const uint32_t *)B.in[i], (uint32_t *)B.in[(i + 1) % B.nring], (int64_t)B.W * B.H / 4); }, 2.0 * 268435456.0);
    B.run("E2 compute ntL ntS      grid 4x4096", L((k_e2<true, true, true>), dim3(4, 4096), dim3(256)));
    B.run("E3 K4 compute ntL ntS   grid 4x4096", L((k_e3<4, true, true, true>), dim3(4, 4096), dim3(256)));
    B.run("E4 R2 compute ntL ntS   grid 4x2048", L((k_e4<2, true, true, true>), dim3(4, 2048), dim3(256)));
    B.run("E1 compute ntL ntS blk128 grid 8x4096", L((k_e1<true, true, true>), dim3(8, 4096), dim3(128)));
    B.run("E1 compute ntL ntS blk64  grid 16x4096", L((k_e1<true, true, true>), dim3(16, 4096), dim3(64)));
    B.run("E1 compute ntL ntS blk64x4 grid 16x1024", L((k_e1<true, true, true>), dim3(16, 1024), dim3(64, 4)));
    B.run("E1 compute ntL ntS blk256x1 again", L((k_e1<true, true, true>), dim3(4, 4096), dim3(256)));
    B.run("E6 xcd-remap compute ntL ntS", L((k_e6<true, true, true>), dim3(16384), dim3(256)));
    B.run("E6 xcd-remap compute plain", L((k_e6<false, false, true>), dim3(16384), dim3(256)));
    B.run("E7 512thr compute ntL ntS grid 2x4096", L((k_e7<true, true, true>), dim3(2, 4096), dim3(512)));
    B.run("E7 1024thr compute ntL ntS grid 1x4096", L((k_e7<true, true, true>), dim3(1, 4096), dim3(1024)));

    B.run("shipped k_dec K4 nt     grid 4x4096", L((k_dec<R_FLOOR, F_ARGB, 2, 1, false, 4, true>), dim3(4, 4096), dim3(256)));
    B.run("shipped k_dec2v var2 nt grid 4x4096", L((k_dec2v<R_FLOOR, F_ARGB, 2, true>), dim3(4, 4096), dim3(256)));

    // ---- f = 1 shape: is the 4-px-per-lane pipeline VALU-limited? (out = next ring input buffer)
    auto f1args = [&](int i) { KArgs a = B.base; a.in = B.in[i]; a.out = B.in[(i + 1) % B.nring]; return a; };
    B.run("E8 f1 copy nt          grid 8x8192", [&](int i) { hipLaunchKernelGGL((k_e8<true, true, 0>), dim3(8, 8192), dim3(256), 0, 0, f1args(i)); }, 536870912.0);
    B.run("E8 f1 4:4:4 nt         grid 8x8192", [&](int i) { hipLaunchKernelGGL((k_e8<true, true, 1>), dim3(8, 8192), dim3(256), 0, 0, f1args(i)); }, 536870912.0);
    B.run("E8 f1 4:2:2 nt         grid 8x8192", [&](int i) { hipLaunchKernelGGL((k_e8<true, true, 2>), dim3(8, 8192), dim3(256), 0, 0, f1args(i)); }, 536870912.0);
    B.run("E8 f1 4:4:4 cached     grid 8x8192", [&](int i) { hipLaunchKernelGGL((k_e8<false, false, 1>), dim3(8, 8192), dim3(256), 0, 0, f1args(i)); }, 536870912.0);

    B.run("E9 f1 K2 4:4:4 nt      grid 4x8192", [&](int i) { hipLaunchKernelGGL((k_e9<2, true>), dim3(4, 8192), dim3(256), 0, 0, f1args(i)); }, 536870912.0);
    B.run("E9 f1 K4 4:4:4 nt      grid 2x8192", [&](int i) { hipLaunchKernelGGL((k_e9<4, true>), dim3(2, 8192), dim3(256), 0, 0, f1args(i)); }, 536870912.0);
    {
        KArgs proto = B.base; proto.H = 8192; proto.Ho = 8192; proto.Wo = 8192; proto.op = 8192; proto.f = 1;
        auto sargs = [&](int i) { KArgs a = proto; a.in = B.in[i]; a.out = B.in[(i + 1) % B.nring]; return a; };
        B.run("shipped k_f1x4 444 nt  grid 8x8192", [&](int i) { hipLaunchKernelGGL((k_f1x4<R_FLOOR, F_ARGB, 1, 1, true>), dim3(8, 8192), dim3(256), 0, 0, sargs(i)); }, 536870912.0);
        B.run("shipped k_f1x4 420 nt  grid 8x8192", [&](int i) { hipLaunchKernelGGL((k_f1x4<R_FLOOR, F_ARGB, 2, 2, true>), dim3(8, 8192), dim3(256), 0, 0, sargs(i)); }, 536870912.0);
    }

    B.run("read-only nt 256MiB   65536 blk", [&](int i) { hipLaunchKernelGGL(k_read_nt, dim3(65536), dim3(256), 0, 0, (const uint32_t *)B.in[i], B.out[0], (int64_t)B.W * B.H / 4); }, 268435456.0);
    B.run("write-only nt 256MiB  65536 blk", [&](int i) { hipLaunchKernelGGL(k_write_nt, dim3(65536), dim3(256), 0, 0, B.in[i], (int64_t)B.W * B.H / 4, (uint32_t)i); }, 268435456.0);
    B.run("E10 K4 rows skipped (x2) grid 4x4096", [&](int i) { hipLaunchKernelGGL((k_e10<4>), dim3(4, 4096), dim3(256), 0, 0, B.args(i), 2); });
    B.run("E10 K4 rows contiguous   grid 4x4096", [&](int i) { hipLaunchKernelGGL((k_e10<4>), dim3(4, 4096), dim3(256), 0, 0, B.args(i), 1); });
    for (int i = 0; i < B.nring; ++i) hipLaunchKernelGGL(k_synth, dim3(8192), dim3(256), 0, 0, B.in[i], (int64_t)B.W * B.H, (int64_t)i * B.W * B.H, 20250629u * 0x9E3779B9u);

    {   // f = 8 on 8192x8192: Wo = Ho = 1024; algorithmic bytes = 4*8192*1024 + 4*1024*1024
        KArgs proto = B.base; proto.Wo = 1024; proto.op = 1024; proto.Ho = 1024; proto.f = 8;
        auto a8 = [&](int i) { KArgs a = proto; a.in = B.in[i]; a.out = B.out[i]; return a; };
        const double by8 = 4.0 * 8192 * 1024 + 4.0 * 1024 * 1024;
        B.run("E11 f8 K1 dword nt   grid 4x1024", [&](int i) { hipLaunchKernelGGL((k_e11<8, 1, 1, true, true>), dim3(4, 1024), dim3(256), 0, 0, a8(i)); }, by8);
        B.run("E11 f8 K2 dword nt   grid 2x1024", [&](int i) { hipLaunchKernelGGL((k_e11<8, 2, 1, true, true>), dim3(2, 1024), dim3(256), 0, 0, a8(i)); }, by8);
        B.run("E11 f8 K4 dword nt   grid 1x1024", [&](int i) { hipLaunchKernelGGL((k_e11<8, 4, 1, true, true>), dim3(1, 1024), dim3(256), 0, 0, a8(i)); }, by8);
        B.run("E11 f8 K1 dword cached grid 4x1024", [&](int i) { hipLaunchKernelGGL((k_e11<8, 1, 1, false, true>), dim3(4, 1024), dim3(256), 0, 0, a8(i)); }, by8);
        B.run("E11 f8 K4 dword cached grid 1x1024", [&](int i) { hipLaunchKernelGGL((k_e11<8, 4, 1, false, true>), dim3(1, 1024), dim3(256), 0, 0, a8(i)); }, by8);
        B.run("E11 f8 K1 x4 nt      grid 4x1024", [&](int i) { hipLaunchKernelGGL((k_e11<8, 1, 4, true, true>), dim3(4, 1024), dim3(256), 0, 0, a8(i)); }, by8);
        B.run("E11 f8 K4 x4 nt      grid 1x1024", [&](int i) { hipLaunchKernelGGL((k_e11<8, 4, 4, true, true>), dim3(1, 1024), dim3(256), 0, 0, a8(i)); }, by8);
        B.run("E11 f8 K2 x4 cached  grid 2x1024", [&](int i) { hipLaunchKernelGGL((k_e11<8, 2, 4, false, true>), dim3(2, 1024), dim3(256), 0, 0, a8(i)); }, by8);
        KArgs p4 = B.base; p4.Wo = 2048; p4.op = 2048; p4.Ho = 2048; p4.f = 4;
        auto a4 = [&](int i) { KArgs a = p4; a.in = B.in[i]; a.out = B.out[i]; return a; };
        const double by4 = 4.0 * 8192 * 2048 + 4.0 * 2048 * 2048;
        B.run("E11 f4 K4 dword nt   grid 2x2048", [&](int i) { hipLaunchKernelGGL((k_e11<4, 4, 1, true, true>), dim3(2, 2048), dim3(256), 0, 0, a4(i)); }, by4);
        B.run("E11 f4 K2 dword nt   grid 4x2048", [&](int i) { hipLaunchKernelGGL((k_e11<4, 2, 1, true, true>), dim3(4, 2048), dim3(256), 0, 0, a4(i)); }, by4);
        B.run("E11 f4 K1 dword nt   grid 8x2048", [&](int i) { hipLaunchKernelGGL((k_e11<4, 1, 1, true, true>), dim3(8, 2048), dim3(256), 0, 0, a4(i)); }, by4);
        B.run("E11 f4 K2 x4 nt      grid 4x2048", [&](int i) { hipLaunchKernelGGL((k_e11<4, 2, 4, true, true>), dim3(4, 2048), dim3(256), 0, 0, a4(i)); }, by4);
        B.run("E11 f4 K4 dword cached grid 2x2048", [&](int i) { hipLaunchKernelGGL((k_e11<4, 4, 1, false, true>), dim3(2, 2048), dim3(256), 0, 0, a4(i)); }, by4);
    }

    B.run("E12 store nt           grid 4x4096", L((k_e12<4, 0>), dim3(4, 4096), dim3(256)));
    B.run("E12 store sc1          grid 4x4096", L((k_e12<4, 1>), dim3(4, 4096), dim3(256)));
    B.run("E12 store sc0 sc1      grid 4x4096", L((k_e12<4, 2>), dim3(4, 4096), dim3(256)));
    B.run("E12 store sc1 nt       grid 4x4096", L((k_e12<4, 3>), dim3(4, 4096), dim3(256)));
    B.run("E12 store sc0 sc1 nt   grid 4x4096", L((k_e12<4, 4>), dim3(4, 4096), dim3(256)));
    B.run("E12 store plain        grid 4x4096", L((k_e12<4, 5>), dim3(4, 4096), dim3(256)));
    B.run("E12 store nt (again)   grid 4x4096", L((k_e12<4, 0>), dim3(4, 4096), dim3(256)));

    {
        KArgs proto = B.base; proto.bdx = 256; proto.bdy = 1; proto.row_step = 4096;
        auto aa = [&](int i) { KArgs a = proto; a.in = B.in[i]; a.out = B.out[i]; return a; };
        const double byavg = 4.0 * 8192 * 8192 + 4.0 * 4096 * 4096;
        B.run("E13 avg-f2 memory shape only  grid 8x4096", [&](int i) { hipLaunchKernelGGL(k_e13, dim3(8, 4096), dim3(256), 0, 0, aa(i)); }, byavg);
        B.run("shipped k_avg f2 h2 v2 nt     grid 4x4096", [&](int i) { hipLaunchKernelGGL((k_avg<R_FLOOR, F_ARGB, 2, 2, 2, true>), dim3(4, 4096), dim3(256), 0, 0, aa(i)); }, byavg);
        B.run("shipped k_avg f2 h1 v1 nt     grid 4x4096", [&](int i) { hipLaunchKernelGGL((k_avg<R_FLOOR, F_ARGB, 2, 1, 1, true>), dim3(4, 4096), dim3(256), 0, 0, aa(i)); }, byavg);
        B.run("shipped k_avg f2 h2 v2 ycc nt grid 4x4096", [&](int i) { hipLaunchKernelGGL((k_avg<R_FLOOR, F_YCC, 2, 2, 2, true>), dim3(4, 4096), dim3(256), 0, 0, aa(i)); }, byavg);
    }
    return 0;
}
