// tools/ubench_unaligned.hip -- do 16-byte global loads / stores work, and at what cost, when the address is only 4-byte
// aligned?  (Rows of a W % 4 != 0 frame start at every 4-byte phase; the AVG tile kernel and the planar kernels would like to
// keep their dwordx4 accesses there.)  Prints correctness and GB/s per dword offset 0..3, nt loads + nt stores, 256 MiB.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <vector>
typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)

__global__ void __launch_bounds__(256) k_copy16(const uint32_t *src, uint32_t *dst, int64_t n4)
{
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= n4) return;
    typedef const u32x4 __attribute__((address_space(1))) *vp;
    typedef u32x4 __attribute__((address_space(1))) *wp;
    const u32x4 v = __builtin_nontemporal_load((vp)(uintptr_t)(src + 4 * i));
    __builtin_nontemporal_store(v, (wp)(uintptr_t)(dst + 4 * i));
}
__global__ void __launch_bounds__(256) k_copy4x4(const uint32_t *src, uint32_t *dst, int64_t n4)
{
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= n4) return;
    uint32_t v[4];
#pragma unroll
    for (int k = 0; k < 4; ++k) v[k] = __builtin_nontemporal_load(src + 4 * i + k);
#pragma unroll
    for (int k = 0; k < 4; ++k) __builtin_nontemporal_store(v[k], dst + 4 * i + k);
}
__global__ void k_fill(uint32_t *p, int64_t n) { for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) p[i] = (uint32_t)(i * 2654435761u); }
__global__ void k_cmp(const uint32_t *a, const uint32_t *b, int64_t n, unsigned long long *bad) { unsigned long long c = 0; for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) c += a[i] != b[i]; if (c) atomicAdd(bad, c); }

int main()
{
    const int64_t n = 64ll << 20;            // dwords: 256 MiB
    uint32_t *src, *dst; unsigned long long *bad;
    CK(hipMalloc(&src, (n + 64) * 4)); CK(hipMalloc(&dst, (n + 64) * 4)); CK(hipMalloc(&bad, 8));
    k_fill<<<4096, 256>>>(src, n + 64);
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    for (int kind = 0; kind < 2; ++kind)
        for (int so = 0; so < 4; ++so)
            for (int dof = 0; dof < 4; dof += (so == 0 ? 1 : 4)) {
                const int64_t n4 = n / 4;
                CK(hipMemset(dst, 0, (n + 64) * 4)); CK(hipMemset(bad, 0, 8));
                auto run = [&]() { if (kind == 0) k_copy16<<<(unsigned)((n4 + 255) / 256), 256>>>(src + so, dst + dof, n4); else k_copy4x4<<<(unsigned)((n4 + 255) / 256), 256>>>(src + so, dst + dof, n4); };
                run(); CK(hipDeviceSynchronize());
                k_cmp<<<4096, 256>>>(src + so, dst + dof, n, bad);
                unsigned long long hb = 0; CK(hipMemcpy(&hb, bad, 8, hipMemcpyDeviceToHost));
                CK(hipEventRecord(e0)); for (int r = 0; r < 20; ++r) run(); CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
                float ms = 0; CK(hipEventElapsedTime(&ms, e0, e1));
                printf("%s src+%d dst+%d dwords: mismatches %llu, %.1f GB/s\n", kind == 0 ? "dwordx4 " : "4 x dword", so, dof, hb, 2.0 * n * 4 * 20 / (ms * 1e-3) / 1e9);
                fflush(stdout);
            }
    return 0;
}
