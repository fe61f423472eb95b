// bfa_softmax.hpp -- four posterior rows per wavefront: lane = 16*g + j handles row g and the
// columns j, 16+j, 32+j, ...  This is exactly the lane/accumulator structure of the reference's
// host log_softmax (see bfa_math.hpp), so the result is bit-identical to it.
#pragma once
#include "bfa_math.hpp"
#include "bfa_types.hpp"

#pragma clang fp contract(off)

namespace bfa {

struct RowLane {
    uint32_t valid; // bit k : column 16k+j exists
    uint32_t tmask; // bit k : column 16k+j is a boosted / floored target
    int blank_k;    // k such that column 16k+j is the blank column, or -1
    int narrowC;    // C when C < 16 (fewer columns than one host vector: another summation order, see softmax16), else 0
};

__device__ __forceinline__ RowLane make_rowlane(int nk, int j, int C, int blank, const uint32_t *um)
{
    RowLane rl;
    rl.valid = 0; rl.tmask = 0; rl.blank_k = -1; rl.narrowC = (C < 16) ? C : 0;
    for (int k = 0; k < nk; ++k) {
        const int c = 16 * k + j;
        if (c < C) {
            rl.valid |= 1u << k;
            if (um && ((um[c >> 5] >> (c & 31)) & 1u)) rl.tmask |= 1u << k;
            if (c == blank) rl.blank_k = k;
        }
    }
    return rl;
}

// narrowC = C for rows of fewer than sixteen columns (0 otherwise): the host then adds the C exponentials one after the other
// (a partial vector is reduced element by element: ((e0 + e1) + e2) + ...), not in sixteen accumulators with a butterfly;
// pinned by tests/golden/narrow_cases.npz (reference outputs for C = 3..15).  The lanes j >= C of a row stay out of the
// maximum (`valid`) and the sequential sum never reads them.
template <int NK>
__device__ __forceinline__ void softmax16(float (&x)[NK], uint32_t valid, float *mx_out = nullptr, float *ls_out = nullptr,
                                          int narrowC = 0)
{
    float mx = (valid & 1u) ? x[0] : -__builtin_inff(); // (a lane without a column: only when C < 16)
#pragma unroll
    for (int k = 1; k < NK; ++k)
        if (valid & (1u << k)) mx = __builtin_fmaxf(mx, x[k]);
    mx = row16_max(mx);
    float acc = 0.0f;
#pragma unroll
    for (int k = 0; k < NK; ++k) {
        x[k] = x[k] - mx;
        const float e = expf_u10(x[k]);
        if (k == 0) acc = e;
        else acc = (valid & (1u << k)) ? (acc + e) : acc;
    }
    float total;
    if (narrowC > 0) { // (wave-uniform)
        const int row0 = (int)(threadIdx.x & 63) & ~15;
        total = __shfl(acc, row0);
        for (int c = 1; c < narrowC; ++c) total = total + __shfl(acc, row0 + c);
    } else total = row16_butterfly_add(acc);
    const float ls = logf_u10(total);
#pragma unroll
    for (int k = 0; k < NK; ++k) x[k] = x[k] - ls;
    if (mx_out) { *mx_out = mx; *ls_out = ls; }
}

// silence anchoring (forced_alignment.py:543-561): `cnt` times { x[blank] += 5 ; x = log_softmax(x) }.
// The four rows of a quad may carry different counts: iterate to the wave maximum and mask.
template <int NK>
__device__ __forceinline__ void anchor_rows(float (&x)[NK], const RowLane &rl, int cnt)
{
    int maxcnt = cnt;
    maxcnt = max(maxcnt, __shfl_xor(maxcnt, 16));
    maxcnt = max(maxcnt, __shfl_xor(maxcnt, 32));
    for (int i = 0; i < maxcnt; ++i) {
        float y[NK];
#pragma unroll
        for (int k = 0; k < NK; ++k) y[k] = (k == rl.blank_k) ? x[k] + 5.0f : x[k];
        softmax16<NK>(y, rl.valid, nullptr, nullptr, rl.narrowC);
        if (i < cnt) {
#pragma unroll
            for (int k = 0; k < NK; ++k) x[k] = y[k];
        }
    }
}

// forced_alignment.py:29-83: +5 on target columns, log_softmax, floor target columns at log(min_phoneme_prob)
template <int NK>
__device__ __forceinline__ void boost_floor(float (&x)[NK], const RowLane &rl, bool boost, bool enforce, float min_logp)
{
    if (boost) {
#pragma unroll
        for (int k = 0; k < NK; ++k) x[k] = (rl.tmask & (1u << k)) ? (x[k] + 5.0f) : x[k];
        softmax16<NK>(x, rl.valid, nullptr, nullptr, rl.narrowC);
    }
    if (enforce) {
#pragma unroll
        for (int k = 0; k < NK; ++k) x[k] = ((rl.tmask & (1u << k)) && x[k] < min_logp) ? min_logp : x[k]; // :79-81
    }
}

} // namespace bfa
