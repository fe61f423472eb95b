// tools/ubench/sstore.hip -- do scalar stores (s_store_dwordx4 + s_dcache_wb) work and what do they cost?
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
__global__ __launch_bounds__(64) void k_s(unsigned long long *out, const float *a, int T)
{
    const int lane = threadIdx.x;
    float x = a[lane], y = a[64 + lane];
    unsigned long long *p = out + (size_t)blockIdx.x * T * 6;
    for (int t = 0; t < T; ++t) {
        x += 0.25f; // changes the masks over time
#pragma unroll
        for (int r = 0; r < 3; ++r) {
            unsigned long long mA, mB;
            asm volatile("v_cmp_lt_f32 %0, %1, %2" : "=s"(mA) : "v"(x), "v"(y + (float)r));
            asm volatile("v_cmp_gt_f32 %0, %1, %2" : "=s"(mB) : "v"(x), "v"(y + (float)r));
            typedef unsigned long long u2 __attribute__((ext_vector_type(2)));
            u2 v = {mA, mB};
            unsigned long long *q = p + (size_t)(t * 3 + r) * 2;
            asm volatile("s_store_dwordx4 %0, %1, 0x0" ::"s"(v), "s"(q) : "memory");
        }
    }
    asm volatile("s_dcache_wb\n s_waitcnt lgkmcnt(0)" ::: "memory");
}
__global__ __launch_bounds__(64) void k_v(unsigned *out, const float *a, int T)
{
    const int lane = threadIdx.x;
    float x = a[lane], y = a[64 + lane];
    unsigned *p = out + (size_t)blockIdx.x * (T / 4) * 64;
    unsigned w = 0;
    for (int t = 0; t < T; ++t) {
        x += 0.25f;
#pragma unroll
        for (int r = 0; r < 3; ++r) {
            asm volatile("v_cmp_lt_f32 vcc, %1, %2\n v_addc_co_u32 %0, vcc, %0, %0, vcc" : "+v"(w) : "v"(x), "v"(y + (float)r) : "vcc");
            asm volatile("v_cmp_gt_f32 vcc, %1, %2\n v_addc_co_u32 %0, vcc, %0, %0, vcc" : "+v"(w) : "v"(x), "v"(y + (float)r) : "vcc");
        }
        if ((t & 3) == 3) { p[(t >> 2) * 64 + lane] = w; w = 0; }
    }
}
int main()
{
    const int B = 8192, T = 1000;
    float ha[128];
    for (int i = 0; i < 128; ++i) ha[i] = (i < 64) ? (float)(i % 7) : 100.0f + (float)(i % 5);
    float *a; (void)hipMalloc(&a, sizeof(ha)); (void)hipMemcpy(a, ha, sizeof(ha), hipMemcpyHostToDevice);
    unsigned long long *os; (void)hipMalloc(&os, (size_t)B * T * 6 * 8); (void)hipMemset(os, 0xff, (size_t)B * T * 6 * 8);
    unsigned *ov; (void)hipMalloc(&ov, (size_t)B * (T / 4) * 64 * 4);
    hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    for (int rep = 0; rep < 2; ++rep) {
        (void)hipEventRecord(e0); hipLaunchKernelGGL(k_s, dim3(B), dim3(64), 0, 0, os, a, T); (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
        float ms; (void)hipEventElapsedTime(&ms, e0, e1); printf("scalar-store masks : %.3f ms\n", ms);
        (void)hipEventRecord(e0); hipLaunchKernelGGL(k_v, dim3(B), dim3(64), 0, 0, ov, a, T); (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
        (void)hipEventElapsedTime(&ms, e0, e1); printf("v_addc packed words: %.3f ms\n", ms);
    }
    // verify a few masks on the host
    std::vector<unsigned long long> h((size_t)T * 6);
    (void)hipMemcpy(h.data(), os + (size_t)4097 * T * 6, h.size() * 8, hipMemcpyDeviceToHost);
    int bad = 0;
    for (int t = 0; t < T; ++t) for (int r = 0; r < 3; ++r) {
        unsigned long long mA = 0, mB = 0;
        for (int l = 0; l < 64; ++l) { float x = ha[l] + 0.25f * (t + 1), y = ha[64 + l] + r; if (x < y) mA |= 1ull << l; if (x > y) mB |= 1ull << l; }
        if (h[(t * 3 + r) * 2] != mA || h[(t * 3 + r) * 2 + 1] != mB) ++bad;
    }
    printf("mask verification: %d wrong of %d\n", bad, T * 3);
    return 0;
}
