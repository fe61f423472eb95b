// tools/ubench/copy_peak.hip -- which float4 copy kernel reaches the copy ceiling of the box (bfa_profile_copy uses the winner).
// hipcc --offload-arch=gfx950 -O3 -o /tmp/copy_peak tools/ubench/copy_peak.hip && /tmp/copy_peak
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

typedef float f4 __attribute__((ext_vector_type(4)));
#define float4 f4

template <int U, bool NT>
__global__ __launch_bounds__(256) void k_stride(float4 *__restrict__ dst, const float4 *__restrict__ src, size_t n)
{
    const size_t stride = (size_t)gridDim.x * 256;
    size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
    for (; i + (U - 1) * stride < n; i += U * stride) {
        float4 v[U];
#pragma unroll
        for (int u = 0; u < U; ++u) v[u] = NT ? __builtin_nontemporal_load(&src[i + u * stride]) : src[i + u * stride];
#pragma unroll
        for (int u = 0; u < U; ++u) { if (NT) __builtin_nontemporal_store(v[u], &dst[i + u * stride]); else dst[i + u * stride] = v[u]; }
    }
    for (; i < n; i += stride) dst[i] = src[i];
}

// each workgroup owns a contiguous chunk
template <int U, bool NT>
__global__ __launch_bounds__(256) void k_chunk(float4 *__restrict__ dst, const float4 *__restrict__ src, size_t n)
{
    const size_t per = (n + gridDim.x - 1) / gridDim.x;
    const size_t lo = (size_t)blockIdx.x * per, hi = lo + per < n ? lo + per : n;
    size_t i = lo + threadIdx.x;
    for (; i + (U - 1) * 256 < hi; i += U * 256) {
        float4 v[U];
#pragma unroll
        for (int u = 0; u < U; ++u) v[u] = NT ? __builtin_nontemporal_load(&src[i + u * 256]) : src[i + u * 256];
#pragma unroll
        for (int u = 0; u < U; ++u) { if (NT) __builtin_nontemporal_store(v[u], &dst[i + u * 256]); else dst[i + u * 256] = v[u]; }
    }
    for (; i < hi; i += 256) dst[i] = src[i];
}

template <typename F> float best_ms(F launch, int reps = 10)
{
    hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
    float best = 1e9;
    for (int r = 0; r < reps + 2; ++r) {
        hipEventRecord(a); launch(); hipEventRecord(b); hipEventSynchronize(b);
        float ms; hipEventElapsedTime(&ms, a, b);
        if (r >= 2 && ms < best) best = ms;
    }
    return best;
}

int main()
{
    for (size_t bytes : {(size_t)1 << 30, (size_t)3 << 30}) {
        float4 *src, *dst;
        hipMalloc(&src, bytes); hipMalloc(&dst, bytes);
        hipMemset(src, 1, bytes); hipMemset(dst, 0, bytes);
        const size_t n = bytes / 16;
        printf("buffer %zu MiB each (read + write = %.2f GB per launch)\n", bytes >> 20, 2.0 * bytes / 1e9);
        for (int grid : {1024, 2048, 4096, 8192, 16384, 65536}) {
#define RUN(name, K) { float ms = best_ms([&] { hipLaunchKernelGGL(K, dim3(grid), dim3(256), 0, 0, dst, src, n); }); \
                       printf("  %-22s grid %6d  %.3f ms  %.0f GB/s\n", name, grid, ms, 2.0 * bytes / ms / 1e6); }
            RUN("stride U4", (k_stride<4, false>)); RUN("stride U8", (k_stride<8, false>)); RUN("stride U4 nt", (k_stride<4, true>));
            RUN("stride U8 nt", (k_stride<8, true>)); RUN("chunk U4", (k_chunk<4, false>)); RUN("chunk U8", (k_chunk<8, false>));
            RUN("chunk U8 nt", (k_chunk<8, true>)); RUN("stride U1", (k_stride<1, false>)); RUN("stride U2 nt", (k_stride<2, true>));
        }
        { float ms = best_ms([&] { hipMemcpyAsync(dst, src, bytes, hipMemcpyDeviceToDevice, 0); });
          printf("  hipMemcpyAsync D2D                 %.3f ms  %.0f GB/s\n", ms, 2.0 * bytes / ms / 1e6); }
        hipFree(src); hipFree(dst);
    }
    return 0;
}
