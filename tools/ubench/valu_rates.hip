// tools/ubench/valu_rates.hip -- measured issue cost (cycles per wave64 instruction per SIMD) of the VALU
// instructions K1 is made of, on the GPU it runs on.  hipcc --offload-arch=gfx950 -O3 valu_rates.hip -o valu_rates
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#include <string>

#define REP8(x) x x x x x x x x
#define ITER 2048

template <int KIND>
__global__ __launch_bounds__(256) void k(float *out, int n)
{
    float a0 = threadIdx.x * 0.001f, a1 = a0 + 1, a2 = a0 + 2, a3 = a0 + 3, a4 = a0 + 4, a5 = a0 + 5, a6 = a0 + 6, a7 = a0 + 7;
    float b = 1.0001f, c = 0.5f;
    unsigned w0 = threadIdx.x, w1 = w0 + 1, w2 = w0 + 2, w3 = w0 + 3;
    unsigned long long m = 0x5555555555555555ull;
    for (int i = 0; i < n; ++i) {
        if (KIND == 0) { // v_fma_f32
            asm volatile("v_fma_f32 %0, %0, %8, %9\n v_fma_f32 %1, %1, %8, %9\n v_fma_f32 %2, %2, %8, %9\n v_fma_f32 %3, %3, %8, %9\n"
                         "v_fma_f32 %4, %4, %8, %9\n v_fma_f32 %5, %5, %8, %9\n v_fma_f32 %6, %6, %8, %9\n v_fma_f32 %7, %7, %8, %9\n"
                         : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(b), "v"(c));
        } else if (KIND == 1) { // v_pk_fma_f32 (4 instr on pairs)
            asm volatile("v_pk_fma_f32 %0, %0, %4, %5\n v_pk_fma_f32 %1, %1, %4, %5\n v_pk_fma_f32 %2, %2, %4, %5\n v_pk_fma_f32 %3, %3, %4, %5\n"
                         "v_pk_fma_f32 %0, %0, %4, %5\n v_pk_fma_f32 %1, %1, %4, %5\n v_pk_fma_f32 %2, %2, %4, %5\n v_pk_fma_f32 %3, %3, %4, %5\n"
                         : "+v"(*(double *)&a0), "+v"(*(double *)&a2), "+v"(*(double *)&a4), "+v"(*(double *)&a6)
                         : "v"(*(double *)&b), "v"(*(double *)&c));
        } else if (KIND == 2) { // v_add_f32
            asm volatile(REP8("v_add_f32 %0, %0, %2\n v_add_f32 %1, %1, %2\n") "" : "+v"(a0), "+v"(a1) : "v"(b));
        } else if (KIND == 3) { // v_max3_f32
            asm volatile("v_max3_f32 %0, %0, %4, %5\n v_max3_f32 %1, %1, %4, %5\n v_max3_f32 %2, %2, %4, %5\n v_max3_f32 %3, %3, %4, %5\n"
                         "v_max3_f32 %0, %0, %4, %5\n v_max3_f32 %1, %1, %4, %5\n v_max3_f32 %2, %2, %4, %5\n v_max3_f32 %3, %3, %4, %5\n"
                         : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3) : "v"(b), "v"(c));
        } else if (KIND == 4) { // v_cndmask with SGPR mask
            asm volatile("v_cndmask_b32_e64 %0, %0, %4, %5\n v_cndmask_b32_e64 %1, %1, %4, %5\n v_cndmask_b32_e64 %2, %2, %4, %5\n v_cndmask_b32_e64 %3, %3, %4, %5\n"
                         "v_cndmask_b32_e64 %0, %0, %4, %5\n v_cndmask_b32_e64 %1, %1, %4, %5\n v_cndmask_b32_e64 %2, %2, %4, %5\n v_cndmask_b32_e64 %3, %3, %4, %5\n"
                         : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3) : "v"(b), "s"(m));
        } else if (KIND == 5) { // v_ldexp_f32
            asm volatile("v_ldexp_f32 %0, %0, %4\n v_ldexp_f32 %1, %1, %4\n v_ldexp_f32 %2, %2, %4\n v_ldexp_f32 %3, %3, %4\n"
                         "v_ldexp_f32 %0, %0, %4\n v_ldexp_f32 %1, %1, %4\n v_ldexp_f32 %2, %2, %4\n v_ldexp_f32 %3, %3, %4\n"
                         : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3) : "v"(w0 & 1));
        } else if (KIND == 6) { // v_rndne + v_cvt_i32
            asm volatile("v_rndne_f32 %0, %0\n v_cvt_i32_f32 %4, %0\n v_rndne_f32 %1, %1\n v_cvt_i32_f32 %5, %1\n"
                         "v_rndne_f32 %2, %2\n v_cvt_i32_f32 %6, %2\n v_rndne_f32 %3, %3\n v_cvt_i32_f32 %7, %3\n"
                         : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "=v"(w0), "=v"(w1), "=v"(w2), "=v"(w3));
        } else if (KIND == 7) { // v_cmp_lt + v_addc (4 pairs)
            asm volatile("v_cmp_lt_f32 vcc, %4, %5\n v_addc_co_u32 %0, vcc, %0, %0, vcc\n v_cmp_lt_f32 vcc, %5, %4\n v_addc_co_u32 %1, vcc, %1, %1, vcc\n"
                         "v_cmp_lt_f32 vcc, %4, %5\n v_addc_co_u32 %2, vcc, %2, %2, vcc\n v_cmp_lt_f32 vcc, %5, %4\n v_addc_co_u32 %3, vcc, %3, %3, vcc\n"
                         : "+v"(w0), "+v"(w1), "+v"(w2), "+v"(w3) : "v"(a0), "v"(a1) : "vcc");
        } else if (KIND == 8) { // v_mov_b32_dpp wave_shr:1
            asm volatile("s_nop 1\n v_mov_b32_dpp %0, %4 wave_shr:1 row_mask:0xf bank_mask:0xf bound_ctrl:1\n v_mov_b32_dpp %1, %5 wave_shr:1 row_mask:0xf bank_mask:0xf bound_ctrl:1\n"
                         "v_mov_b32_dpp %2, %6 wave_shr:1 row_mask:0xf bank_mask:0xf bound_ctrl:1\n v_mov_b32_dpp %3, %7 wave_shr:1 row_mask:0xf bank_mask:0xf bound_ctrl:1\n"
                         "v_mov_b32_dpp %0, %4 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf bound_ctrl:1\n v_mov_b32_dpp %1, %5 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf bound_ctrl:1\n"
                         "v_mov_b32_dpp %2, %6 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf bound_ctrl:1\n v_mov_b32_dpp %3, %7 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf bound_ctrl:1\n"
                         : "=v"(a0), "=v"(a1), "=v"(a2), "=v"(a3) : "v"(a4), "v"(a5), "v"(a6), "v"(a7));
        } else if (KIND == 9) { // v_pk_add_f32
            asm volatile("v_pk_add_f32 %0, %0, %4\n v_pk_add_f32 %1, %1, %4\n v_pk_add_f32 %2, %2, %4\n v_pk_add_f32 %3, %3, %4\n"
                         "v_pk_add_f32 %0, %0, %4\n v_pk_add_f32 %1, %1, %4\n v_pk_add_f32 %2, %2, %4\n v_pk_add_f32 %3, %3, %4\n"
                         : "+v"(*(double *)&a0), "+v"(*(double *)&a2), "+v"(*(double *)&a4), "+v"(*(double *)&a6) : "v"(*(double *)&b));
        } else if (KIND == 10) { // v_readlane_b32
            unsigned s0, s1, s2, s3;
            asm volatile("v_readlane_b32 %0, %4, 0\n v_readlane_b32 %1, %5, 1\n v_readlane_b32 %2, %6, 2\n v_readlane_b32 %3, %7, 3\n"
                         "v_readlane_b32 %0, %4, 4\n v_readlane_b32 %1, %5, 5\n v_readlane_b32 %2, %6, 6\n v_readlane_b32 %3, %7, 7\n"
                         : "=s"(s0), "=s"(s1), "=s"(s2), "=s"(s3) : "v"(a0), "v"(a1), "v"(a2), "v"(a3));
            w0 += s0 + s1 + s2 + s3;
        } else if (KIND == 11) { // v_mul_f64 / v_add_f64
            double d0 = a0, d1 = a1;
            asm volatile(REP8("v_fma_f64 %0, %0, %2, %2\n v_fma_f64 %1, %1, %2, %2\n") "" : "+v"(d0), "+v"(d1) : "v"((double)b));
            a0 = (float)d0; a1 = (float)d1;
        } else if (KIND == 12) { // s_ (SALU) 64-bit shifts
            unsigned long long s = m;
            asm volatile(REP8("s_lshl_b64 %0, %0, 1\n s_lshr_b64 %0, %0, 1\n") "" : "+s"(s));
            m = s;
        }
    }
    out[blockIdx.x * blockDim.x + threadIdx.x] = a0 + a1 + a2 + a3 + a4 + a5 + a6 + a7 + w0 + w1 + w2 + w3 + (float)(m & 1);
}

template <int KIND>
double run(const char *name, int instr_per_iter, int waves_per_simd)
{
    hipDeviceProp_t prop;
    (void)hipGetDeviceProperties(&prop, 0);
    const int cus = prop.multiProcessorCount;
    const int blocks = cus * waves_per_simd; // 256 threads = 4 waves = one per SIMD
    float *out;
    (void)hipMalloc(&out, sizeof(float) * blocks * 256);
    hipEvent_t e0, e1;
    (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    hipLaunchKernelGGL(k<KIND>, dim3(blocks), dim3(256), 0, 0, out, 64);
    (void)hipDeviceSynchronize();
    (void)hipEventRecord(e0);
    hipLaunchKernelGGL(k<KIND>, dim3(blocks), dim3(256), 0, 0, out, ITER);
    (void)hipEventRecord(e1);
    (void)hipEventSynchronize(e1);
    float ms;
    (void)hipEventElapsedTime(&ms, e0, e1);
    const double clk = prop.clockRate * 1e3; // Hz
    const double cycles = ms * 1e-3 * clk;
    const double per = cycles / ((double)ITER * instr_per_iter * waves_per_simd);
    printf("%-28s waves/SIMD=%d  %.2f cycles per wave-instruction per SIMD (at %.0f MHz nominal, %.3f ms)\n", name,
           waves_per_simd, per, clk / 1e6, ms);
    (void)hipFree(out);
    return per;
}

int main()
{
    for (int w : {1, 2, 4, 8}) {
        run<0>("v_fma_f32", 8, w);
        run<1>("v_pk_fma_f32", 8, w);
        run<2>("v_add_f32 (dep pairs)", 16, w);
        run<3>("v_max3_f32", 8, w);
        run<4>("v_cndmask (sgpr mask)", 8, w);
        run<5>("v_ldexp_f32", 8, w);
        run<6>("v_rndne+v_cvt_i32", 8, w);
        run<7>("v_cmp_lt+v_addc", 8, w);
        run<8>("v_mov_dpp", 8, w);
        run<9>("v_pk_add_f32", 8, w);
        run<10>("v_readlane_b32", 8, w);
        run<11>("v_fma_f64", 16, w);
        run<12>("s_lshl/lshr_b64", 16, w);
        printf("\n");
    }
    return 0;
}
