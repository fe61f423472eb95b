// bfa_math.hpp -- float32 numerics that must agree bit-for-bit with the reference's host library.
//
// The reference normalises with torch.nn.functional.log_softmax on a CPU tensor
// (forced_alignment.py:54,560).  On the AVX512 dispatch of torch 2.10 that is: row max, a
// Sleef-u10 expf of (x - max) accumulated in sixteen lane sums (tail columns into the low lanes),
// an xor-butterfly 8/4/2/1 over the sixteen sums, a Sleef-u10 logf of the total and
// out = (x - max) - log(sum).  The two functions below are written from Sleef's published
// algorithm (FMA build) with explicit __builtin_fmaf so that, compiled with -ffp-contract=off,
// every operation is a single IEEE rounding on the GPU exactly as on the host.
#pragma once
#include <hip/hip_runtime.h>

#pragma clang fp contract(off)

namespace bfa {

__device__ __forceinline__ float as_f(int i) { return __builtin_bit_cast(float, i); }
__device__ __forceinline__ int as_i(float f) { return __builtin_bit_cast(int, f); }

// 2^q for q in the normal exponent range
__device__ __forceinline__ float pow2i(int q) { return as_f((q + 0x7f) << 23); }

__device__ __forceinline__ float expf_u10(float d)
{
    const int q = (int)__builtin_rintf(d * 1.442695040888963407359924681001892137426645954152985934135449406931f);
    const float qf = (float)q;
    float s = __builtin_fmaf(qf, -0.693145751953125f, d);
    s = __builtin_fmaf(qf, -1.428606765330187045e-06f, s);
    float u = 0.000198527617612853646278381f;
    u = __builtin_fmaf(u, s, 0.00139304355252534151077271f);
    u = __builtin_fmaf(u, s, 0.00833336077630519866943359f);
    u = __builtin_fmaf(u, s, 0.0416664853692054748535156f);
    u = __builtin_fmaf(u, s, 0.166666671633720397949219f);
    u = __builtin_fmaf(u, s, 0.5f);
    u = 1.0f + __builtin_fmaf(s * s, u, s);
    u = (u * pow2i(q >> 1)) * pow2i(q - (q >> 1));
    if (d < -104.0f) u = 0.0f;
    if (d > 100.0f) u = __builtin_inff();
    return u;
}

// torch.exp on a float32 CPU tensor (forced_alignment.py:503, utils.py:81, core.py:704) is MKL VML's vsExp, whose
// algorithm is not published; measured, it equals the CORRECTLY ROUNDED float32 exponential on 98.9 % of inputs
// (<= 1 ulp on the rest; Sleef's expf_u10 only matches it on 90 %).  So it is restated as a float64 exponential
// (Taylor to r^13 on |r| <= ln2/2, error < 2^-56) rounded once to float32 -- IEEE float64 ops and explicit fma only,
// the same sequence as oracle/bfa_oracle.c::ora_exp_cr, hence the same bits.  Used by the silence-probability,
// confidence and soft-boundary passes (one element per row: the float64 rate does not matter there).
__device__ __forceinline__ float exp_cr(float xf)
{
    const double x = (double)xf;
    if (!(x > -104.0)) return (x != x) ? xf : 0.0f;
    if (x > 89.0) return __builtin_inff();
    const double kd = __builtin_rint(x * 0x1.71547652b82fep+0);
    double r = __builtin_fma(kd, -0x1.62e42fefa38p-1, x);
    r = __builtin_fma(kd, -0x1.ef35793c7673p-45, r);
    double p = 1.0 / 6227020800.0;
    p = __builtin_fma(p, r, 1.0 / 479001600.0);
    p = __builtin_fma(p, r, 1.0 / 39916800.0);
    p = __builtin_fma(p, r, 1.0 / 3628800.0);
    p = __builtin_fma(p, r, 1.0 / 362880.0);
    p = __builtin_fma(p, r, 1.0 / 40320.0);
    p = __builtin_fma(p, r, 1.0 / 5040.0);
    p = __builtin_fma(p, r, 1.0 / 720.0);
    p = __builtin_fma(p, r, 1.0 / 120.0);
    p = __builtin_fma(p, r, 1.0 / 24.0);
    p = __builtin_fma(p, r, 1.0 / 6.0);
    p = __builtin_fma(p, r, 0.5);
    p = __builtin_fma(p, r, 1.0);
    p = __builtin_fma(p, r, 1.0);
    return (float)__builtin_ldexp(p, (int)kd);
}

struct f2 { float x, y; };
__device__ __forceinline__ f2 df_mul_f(f2 a, float b)
{
    f2 r; r.x = a.x * b; r.y = __builtin_fmaf(a.y, b, __builtin_fmaf(a.x, b, -r.x)); return r;
}
__device__ __forceinline__ f2 df_add2_ff(float a, float b)
{
    f2 r; r.x = a + b; const float v = r.x - a; r.y = (a - (r.x - v)) + (b - v); return r;
}
__device__ __forceinline__ f2 df_div(f2 n, f2 d)
{
    const float t = 1.0f / d.x; // correctly rounded (-fhip-fp32-correctly-rounded-divide-sqrt, the hipcc default)
    f2 q; q.x = n.x * t;
    const float u = __builtin_fmaf(t, n.x, -q.x);
    const float w = __builtin_fmaf(-d.y, t, __builtin_fmaf(-d.x, t, 1.0f));
    q.y = __builtin_fmaf(q.x, w, __builtin_fmaf(n.y, t, u));
    return q;
}
__device__ __forceinline__ f2 df_add_22(f2 a, f2 b)
{
    f2 r; r.x = a.x + b.x; r.y = (((a.x - r.x) + b.x) + a.y) + b.y; return r;
}
__device__ __forceinline__ f2 df_add_2f(f2 a, float b)
{
    f2 r; r.x = a.x + b; r.y = ((a.x - r.x) + b) + a.y; return r;
}

// positive normal arguments only: the softmax denominator lies in [1, C]
__device__ __forceinline__ float logf_u10(float d)
{
    const float de = d * (1.0f / 0.75f);
    const float e = (float)(((as_i(de) >> 23) & 0xff) - 127);
    float m = as_f((as_i(d) & 0x007fffff) | 0x3f800000);
    if (m >= 1.5f) m *= 0.5f;
    const f2 ln2 = {0.69314718246459960938f, -1.904654323148236017e-09f};
    f2 s = df_mul_f(ln2, e);
    const f2 x = df_div(df_add2_ff(-1.0f, m), df_add2_ff(1.0f, m));
    const float x2 = x.x * x.x;
    float t = 0.3027294874e+0f;
    t = __builtin_fmaf(t, x2, 0.3996108174e+0f);
    t = __builtin_fmaf(t, x2, 0.6666694880e+0f);
    const f2 xs = {x.x * 2.0f, x.y * 2.0f};
    s = df_add_22(s, xs);
    s = df_add_2f(s, x2 * x.x * t);
    return s.x + s.y;
}

// ---- cross-lane helpers (wave64) -------------------------------------------------------------
template <int CTRL>
__device__ __forceinline__ float dpp_mov(float old, float v)
{
    return as_f(__builtin_amdgcn_update_dpp(as_i(old), as_i(v), CTRL, 0xf, 0xf, false));
}
constexpr int DPP_WAVE_SHR1 = 0x138; // lane l <- lane l-1 across the whole wave64, lane 0 keeps `old`
// lane l <- lane l-1; lane 0 receives 0 (bound_ctrl), which lets the compiler fold the move into the
// consuming VALU instruction as a DPP operand
__device__ __forceinline__ float wave_shr1_zero(float v)
{
    return as_f(__builtin_amdgcn_update_dpp(0, as_i(v), DPP_WAVE_SHR1, 0xf, 0xf, true));
}
constexpr int DPP_ROW_ROR8 = 0x128;  // rotate within each 16-lane row
constexpr int DPP_ROW_ROR4 = 0x124;
constexpr int DPP_ROW_ROR2 = 0x122;
constexpr int DPP_ROW_ROR1 = 0x121;

// sum over the sixteen lanes of a DPP row in torch's butterfly order: v += v[j^8]; ^4; ^2; ^1.
// After each step the vector is periodic, so rotate-by-sh reads the same value as xor-sh.
__device__ __forceinline__ float row16_butterfly_add(float v)
{
    v = v + dpp_mov<DPP_ROW_ROR8>(0.0f, v);
    v = v + dpp_mov<DPP_ROW_ROR4>(0.0f, v);
    v = v + dpp_mov<DPP_ROW_ROR2>(0.0f, v);
    v = v + dpp_mov<DPP_ROW_ROR1>(0.0f, v);
    return v;
}
__device__ __forceinline__ float row16_max(float v)
{
    v = __builtin_fmaxf(v, dpp_mov<DPP_ROW_ROR8>(v, v));
    v = __builtin_fmaxf(v, dpp_mov<DPP_ROW_ROR4>(v, v));
    v = __builtin_fmaxf(v, dpp_mov<DPP_ROW_ROR2>(v, v));
    v = __builtin_fmaxf(v, dpp_mov<DPP_ROW_ROR1>(v, v));
    return v;
}

// ---- sparse readers of the log-prob matrix (confidences, soft boundaries) -------------------------------------------
// RAW = false: the matrix is stored.  RAW = true: `lp` holds the raw logits and `st` the per-row (maximum, log-sum) pairs
// of their log_softmax, written by K1 for every row it prepared: log_prob = (x - max) - logsum, torch's two float32
// subtractions.  Rows K1 never prepared (silence fills, frames beyond the utterance, proportional / empty items) still
// hold the NaN the call initialised the buffer with; the reader then computes that row's statistics itself -- the same
// sixteen-accumulator order (the sequential one below sixteen columns), one lane, slow but rare -- and leaves them for the next reader.
__device__ __noinline__ float2 row_stats_on_demand(const float *row, int C, float *slot)
{
    float mx = row[0];
    for (int c = 1; c < C; ++c) mx = __builtin_fmaxf(mx, row[c]);
    float ls;
    if (C < 16) { // fewer columns than one host vector: torch adds the exponentials one after the other (bfa_softmax.hpp: softmax16)
        float total = expf_u10(row[0] - mx);
        for (int c = 1; c < C; ++c) total = total + expf_u10(row[c] - mx);
        ls = logf_u10(total);
    } else {
        float acc[16]; // (indexed by unrolled constants only: `acc[c & 15]` put the array -- and 2 x C scratch accesses per call -- into private memory)
#pragma unroll
        for (int j = 0; j < 16; ++j) acc[j] = expf_u10(row[j] - mx);
        for (int c0 = 16; c0 < C; c0 += 16) {
#pragma unroll
            for (int j = 0; j < 16; ++j)
                if (c0 + j < C) acc[j] = acc[j] + expf_u10(row[c0 + j] - mx);
        }
#pragma unroll
        for (int j = 0; j < 8; ++j) acc[j] = acc[j] + acc[j + 8];   // xor 8
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[j] = acc[j] + acc[j + 4];   // xor 4
        ls = logf_u10((acc[0] + acc[2]) + (acc[1] + acc[3]));        // xor 2, xor 1
    }
    const float2 r = make_float2(mx, ls);
    *(float2 *)slot = r;
    return r;
}

template <bool RAW>
struct LpView {
    const float *lp; // row 0 of the utterance
    int64_t ld;
    float *st;       // RAW: row statistics of the utterance ([Tmax] pairs)
    int C;
    __device__ __forceinline__ float at(int f, int c) const
    {
        const float x = lp[(int64_t)f * ld + c];
        if (!RAW) return x;
        float2 ms = *(const float2 *)(st + 2 * (int64_t)f);
        if (ms.x != ms.x || ms.y != ms.y) ms = row_stats_on_demand(lp + (int64_t)f * ld, C, st + 2 * (int64_t)f);
        return (x - ms.x) - ms.y;
    }
    // U elements of column c, frames f0 .. f0+U-1 clamped to `fmax`: every load (values and statistics) is issued before
    // anything is looked at, so the 2U of them are in flight together (one element per row: latency is the whole cost)
    template <int U>
    __device__ __forceinline__ void at_n(int f0, int fmax, int c, float (&out)[U]) const
    {
        float2 ms[U];
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const int f = min(f0 + u, fmax);
            out[u] = lp[(int64_t)f * ld + c];
            if (RAW) ms[u] = *(const float2 *)(st + 2 * (int64_t)f);
        }
        if (RAW) {
            bool missing = false;
#pragma unroll
            for (int u = 0; u < U; ++u) missing = missing || (ms[u].x != ms[u].x) || (ms[u].y != ms[u].y);
            if (missing) { // (unrolled: a dynamic index would push ms[] into scratch memory)
#pragma unroll
                for (int u = 0; u < U; ++u) {
                    const int f = min(f0 + u, fmax);
                    if (ms[u].x != ms[u].x || ms[u].y != ms[u].y) ms[u] = row_stats_on_demand(lp + (int64_t)f * ld, C, st + 2 * (int64_t)f);
                }
            }
#pragma unroll
            for (int u = 0; u < U; ++u) out[u] = (out[u] - ms[u].x) - ms[u].y;
        }
    }
};


// torch.Tensor.sum() / .mean() of a STRIDED float32 1-D view on the CPU -- probs[start:end, phoneme] at core.py:711 -- is not
// a running sum: ATen's cascade_sum (SumKernel.cpp: scalar_inner_sum -> row_sum -> multi_row_sum) reads the elements as rows
// of four, keeps four interleaved float32 accumulators, moves them up a level every sixteen rows (level_power = max(4,
// ceil_log2(rows) / 4) = 4 below 2^20 rows), folds the levels, adds the n % 4 tail elements to accumulator 0 and then
// accumulators 1..3 to it.  Restated operation by operation (checked against torch on 4 000 random columns of 1..400 frames:
// equal bits every time, tests/test_oracle_vs_reference_live.py); `get(i)` = element i of the view.
template <class F>
__host__ __device__ inline float cascade_sum_f32(int n, F get)
{
    float a0[4] = {0.0f, 0.0f, 0.0f, 0.0f}, a1[4] = {0.0f, 0.0f, 0.0f, 0.0f}, a2[4] = {0.0f, 0.0f, 0.0f, 0.0f}, a3[4] = {0.0f, 0.0f, 0.0f, 0.0f};
    const int rows = n >> 2, full = rows & ~15;
    int i = 0;
    while (i < full) {
        for (int j = 0; j < 16; ++j, ++i)
            for (int k = 0; k < 4; ++k) a0[k] = a0[k] + get(4 * i + k);
        for (int k = 0; k < 4; ++k) { a1[k] = a1[k] + a0[k]; a0[k] = 0.0f; }
        if ((i & (15 << 4)) != 0) continue;
        for (int k = 0; k < 4; ++k) { a2[k] = a2[k] + a1[k]; a1[k] = 0.0f; }
        if ((i & (15 << 8)) != 0) continue;
        for (int k = 0; k < 4; ++k) { a3[k] = a3[k] + a2[k]; a2[k] = 0.0f; }
    }
    for (; i < rows; ++i)
        for (int k = 0; k < 4; ++k) a0[k] = a0[k] + get(4 * i + k);
    for (int k = 0; k < 4; ++k) { a0[k] = a0[k] + a1[k]; a0[k] = a0[k] + a2[k]; a0[k] = a0[k] + a3[k]; }
    for (int e = 4 * rows; e < n; ++e) a0[0] = a0[0] + get(e);
    a0[0] = a0[0] + a0[1];
    a0[0] = a0[0] + a0[2];
    a0[0] = a0[0] + a0[3];
    return a0[0];
}

} // namespace bfa
