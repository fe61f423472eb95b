// bfa_segment.hip -- silence-anchored segmented mode (forced_alignment.py:268-561), planned on the
// device so that bfa_align_batch never synchronises with the host:
//
//   k_silprob      K0: for every candidate utterance (target contains SIL, anchors > 0) the float32
//                  P(SIL) of each boosted/floored row: exp(m[t, SIL])  (:503-504)
//   k_plan_seg     one thread per candidate: sliding-window silence detection (:471-541), matching of
//                  target SIL groups to audio silences (:226-266), segment list + merge (:328-369) and,
//                  per speech segment, sub-silence detection / anchor counts (:415-419), stride and
//                  band (:422-441).  It emits one work item per piece (DP, silence fill, blank fill)
//                  or leaves the utterance on its standard-mode fallback item (:293-325, :427-429).
//
// All control-flow arithmetic is the reference's: python floats are doubles, int() truncates,
// torch.cumsum(float32) is a float64 running sum rounded to float32 at every element.
#include <hip/hip_runtime.h>

#include "bfa_softmax.hpp"

#pragma clang fp contract(off)

namespace bfa {

// =================================================================================================
// K0 : P(SIL) per row of the candidates
// =================================================================================================
template <int NK>
__global__ __launch_bounds__(64) void k_silprob(AlignArgs a)
{
    const int lane = threadIdx.x & 63, g = lane >> 4, j = lane & 15;
    const DevParams &p = a.p;
    const int n_cand = a.B; // (candidates are marked in ucand[]: no list, no counter -- see k_plan)
    const int nchunk = (a.Tmax + 63) / 64;
    const int sil_k = p.sil >> 4, sil_j = p.sil & 15;
    for (int64_t u = blockIdx.x; u < (int64_t)n_cand * nchunk; u += gridDim.x) {
        const int b = (int)(u / nchunk);
        if (!a.ucand[b]) continue;
        const int t0 = (u % nchunk) * 64;
        const int T = a.uT[b];
        if (t0 >= T) continue;
        const RowLane rl = make_rowlane(NK, j, a.C, p.blank, a.umask + (int64_t)b * MASK_WORDS);
        const float *lp = a.logp + (int64_t)b * a.strideB;
        float *ps = a.psil + (int64_t)b * a.Tmax;
        for (int q = 0; q < 16; ++q) {
            const int t = t0 + 4 * q + g;
            const int row = min(t, T - 1);
            float x[NK];
#pragma unroll
            for (int k = 0; k < NK; ++k) x[k] = lp[(int64_t)row * a.strideT + min(16 * k + j, a.C - 1)];
            if (a.row_stats) softmax16<NK>(x, rl.valid, nullptr, nullptr, rl.narrowC); // raw logits in
            boost_floor<NK>(x, rl, p.boost != 0, p.enforce != 0, p.min_logp);
            float v = 0.0f;
#pragma unroll
            for (int k = 0; k < NK; ++k) if (k == sil_k) v = x[k];
            if (j == sil_j && t < T) ps[t] = exp_cr(v);
        }
    }
}

// =================================================================================================
// planner helpers (single thread)
// =================================================================================================
// _detect_silence_segments (:471-541) on a precomputed P(SIL) vector
__device__ int detect_silence(const float *ps, int Tx, float thr, int k, int32_t *out, int cap)
{
    if (Tx < k) return 0; // :499
    if (k < 1) k = 1;
    const int nwin = (k > 1) ? (Tx - k + 1) : Tx;
    double ahi = 0.0, alo = 0.0;
    float cs_lo = 0.0f;
    if (k > 1) for (int i = 0; i < k - 1; ++i) ahi += (double)ps[i];
    int n = 0, start = 0;
    bool in_sil = false;
    for (int i = 0; i < nwin; ++i) {
        float avg;
        if (k > 1) {
            ahi += (double)ps[i + k - 1];
            const float cs_hi = (float)ahi; // cumsum[i+k-1]
            avg = (cs_hi - cs_lo) / (float)k; // :510
            alo += (double)ps[i];
            cs_lo = (float)alo;             // cumsum[i]
        } else avg = ps[i];
        const bool silent = avg >= thr; // :517
        if (silent && !in_sil) { in_sil = true; start = i; }
        else if (!silent && in_sil) {
            in_sil = false;
            int e = i + k - 1; // :530-531
            if (e > Tx) e = Tx;
            if (e - start >= k) { if (n >= cap) return -1; out[2 * n] = start; out[2 * n + 1] = e; ++n; }
        }
    }
    if (in_sil && Tx - start >= k) { if (n >= cap) return -1; out[2 * n] = start; out[2 * n + 1] = Tx; ++n; } // :536-539
    return n;
}

// The same by the whole wavefront (all lanes follow the same control flow; `ps` in global memory or LDS).  The
// running float64 sum of torch.cumsum stays serial (its additions are not reorderable), but it is walked once per
// call -- cs[i] = float32(cumsum[i]) goes to LDS -- and the window averages, the threshold test and the run
// bookkeeping work on 64 windows at a time (ballot + bit scans).
__device__ int detect_silence_w(const float *ps, int Tx, float thr, int k, int32_t *out, int cap, float *cs, int lane)
{
    if (Tx < k) return 0; // :499
    if (k < 1) k = 1;
    const int nwin = (k > 1) ? (Tx - k + 1) : Tx;
    if (k > 1) {
        // cumsum[i], float64 accumulate (:508): 64 values per LDS round trip, handed from lane to lane with readlane
        // so that no memory latency sits in the serial chain.  Lane l adds the values j <= l of the slice (in order;
        // adding 0.0 for the others changes nothing), so it ends with cumsum[base + l]; lane 63's sum carries over.
        double carry = 0.0;
        float xn = (lane < Tx) ? ps[lane] : 0.0f; // (values past Tx add 0.0)
        for (int base = 0; base < Tx; base += 64) {
            const float x = xn;
            xn = (base + 64 + lane < Tx) ? ps[base + 64 + lane] : 0.0f; // the next slice is in flight during this one's chain
            double acc = carry;
#ifdef BFA_EXP_PLAN_SCAN // (timing experiment: the prefix sums of a slice as a log-step scan -- NOT torch's order of additions)
            {
                double v = (double)x;
#pragma unroll
                for (int d = 1; d < 64; d <<= 1) { const double o = __shfl_up(v, d); if (lane >= d) v += o; }
                acc = carry + v;
            }
#else
            // Step j: the lanes >= j add x[j].  The lane set shrinks by one lane per step, so it is kept in EXEC itself and
            // shifted by the scalar unit; a step is three vector instructions (two broadcasts, one add).  Written out by
            // the compiler from `acc += (lane >= j) ? x[j] : 0` it was eight: the 64 lane masks did not fit the scalar
            // registers and came back through v_readlane, two per step, followed by two selects.
            {
                const long long xb = __builtin_bit_cast(long long, (double)x); // converted once per lane, not once per step
                const int xlo = (int)(xb & 0xffffffffll), xhi = (int)(xb >> 32);
                // (generated: tools/gen_cs_chain.py)  The broadcasts of step j + 5 are issued ahead of the add of step j, into six rotating
                // SGPR pairs: written right before its add, a v_readlane result reached the add ~10 cycles late at every step (the
                // planner's phase clocks, profiles/r05_plan_stamps.txt: 55 cycles per step of four instructions).  Same additions, same order.
                asm volatile(
                    "v_readlane_b32 s20, %[xlo], 0\n\tv_readlane_b32 s21, %[xhi], 0\n\t"
                    "v_readlane_b32 s22, %[xlo], 1\n\tv_readlane_b32 s23, %[xhi], 1\n\t"
                    "v_readlane_b32 s24, %[xlo], 2\n\tv_readlane_b32 s25, %[xhi], 2\n\t"
                    "v_readlane_b32 s26, %[xlo], 3\n\tv_readlane_b32 s27, %[xhi], 3\n\t"
                    "v_readlane_b32 s28, %[xlo], 4\n\tv_readlane_b32 s29, %[xhi], 4\n\t"
                    "v_readlane_b32 s30, %[xlo], 5\n\tv_readlane_b32 s31, %[xhi], 5\n\tv_add_f64 %[acc], %[acc], s[20:21]\n\t"
                    "v_readlane_b32 s20, %[xlo], 6\n\tv_readlane_b32 s21, %[xhi], 6\n\ts_lshl_b64 exec, exec, 1\n\tv_add_f64 %[acc], %[acc], s[22:23]\n\t"
                    "v_readlane_b32 s22, %[xlo], 7\n\tv_readlane_b32 s23, %[xhi], 7\n\ts_lshl_b64 exec, exec, 1\n\tv_add_f64 %[acc], %[acc], s[24:25]\n\t"
                    "v_readlane_b32 s24, %[xlo], 8\n\tv_readlane_b32 s25, %[xhi], 8\n\ts_lshl_b64 exec, exec, 1\n\tv_add_f64 %[acc], %[acc], s[26:27]\n\t"
                    "v_readlane_b32 s26, %[xlo], 9\n\tv_readlane_b32 s27, %[xhi], 9\n\ts_lshl_b64 exec, exec, 1\n\tv_add_f64 %[acc], %[acc], s[28:29]\n\t"
                    "v_readlane_b32 s28, %[xlo], 10\n\tv_readlane_b32 s29, %[xhi], 10\n\ts_lshl_b64 exec, exec, 1\n\tv_add_f64 %[acc], %[acc], s[30:31]\n\t"
                    "v_readlane_b32 s30, %[xlo], 11\n\tv_readlane_b32 s31, %[xhi], 11\n\ts_lshl_b64 exec, exec, 1\n\tv_add_f64 %[acc], %[acc], s[20:21]\n\t"
                    "v_readlane_b32 s20, %[xlo], 12\n\tv_readlane_b32 s21, %[xhi], 12\n\ts_lshl_b64 exec, exec, 1\n\tv_add_f64 %[acc], %[acc], s[22:23]\n\t"
                    "v_readlane_b32 s22, %[xlo], 13\n\tv_readlane_b32 s23, %[xhi], 13\n\ts_lshl_b64 exec, exec, 1\n\tv_add_f64 %[acc], %[acc], s[24:25]\n\t"
                    "v_readlane_b32 s24, %[xlo], 14\n\tv_readlane_b32 s25, %[xhi], 14\n\ts_lshl_b64 exec, exec, 1\n\tv_add_f64 %[acc], %[acc], s[26:27]\n\t"
                    "v_readlane_b32 s26, %[xlo], 15\n\tv_readlane_b32 s27, %[xhi], 15\n\ts_lshl_b64 exec, exec, 1\n\tv_add_f64 %[acc], %[acc], s[28:29]\n\t"
                    "v_readlane_b32 s28, %[xlo], 16\n\tv_readlane_b32 s29, %[xhi], 16\n\ts_lshl_b64 exec, exec, 1\n\tv_add_f64 %[acc], %[acc], s[30:31]\n\t"
                    "v_readlane_b32 s30, %[xlo], 17\n\tv_readlane_b32 s31, %[xhi], 17\n\ts_lshl_b64 exec, exec, 1\n\tv_add_f64 %[acc], %[acc], s[20:21]\n\t"
                    "v_readlane_b32 s20, %[xlo], 18\n\tv_readlane_b32 s21, %[xhi], 18\n\ts_lshl_b64 exec, exec, 1\n\tv_add_f64 %[acc], %[acc], s[22:23]\n\t"
                    "v_readlane_b32 s22, %[xlo], 19\n\tv_readlane_b32 s23, %[xhi], 19\n\ts_lshl_b64 exec, exec, 1\n\tv_add_f64 %[acc], %[acc], s[24:25]\n\t"
                    "v_readlane_b32 s24, %[xlo], 20\n\tv_readlane_b32 s25, %[xhi], 20\n\ts_lshl_b64 exec, exec, 1\n\tv_add_f64 %[acc], %[acc], s[26:27]\n\t"
                    "v_readlane_b32 s26, %[xlo], 21\n\tv_readlane_b32 s27, %[xhi], 21\n\ts_lshl_b64 exec, exec, 1\n\tv_add_f64 %[acc], %[acc], s[28:29]\n\t"
                    "v_readlane_b32 s28, %[xlo], 22\n\tv_readlane_b32 s29, %[xhi], 22\n\ts_lshl_b64 exec, exec, 1\n\tv_add_f64 %[acc], %[acc], s[30:31]\n\t"
                    "v_readlane_b32 s30, %[xlo], 23\n\tv_readlane_b32 s31, %[xhi], 23\n\ts_lshl_b64 exec, exec, 1\n\tv_add_f64 %[acc], %[acc], s[20:21]\n\t"
                    "v_readlane_b32 s20, %[xlo], 24\n\tv_readlane_b32 s21, %[xhi], 24\n\ts_lshl_b64 exec, exec, 1\n\tv_add_f64 %[acc], %[acc], s[22:23]\n\t"
                    "v_readlane_b32 s22, %[xlo], 25\n\tv_readlane_b32 s23, %[xhi], 25\n\ts_lshl_b64 exec, exec, 1\n\tv_add_f64 %[acc], %[acc], s[24:25]\n\t"
                    "v_readlane_b32 s24, %[xlo], 26\n\tv_readlane_b32 s25, %[xhi], 26\n\ts_lshl_b64 exec, exec, 1\n\tv_add_f64 %[acc], %[acc], s[26:27]\n\t"
                    "v_readlane_b32 s26, %[xlo], 27\n\tv_readlane_b32 s27, %[xhi], 27\n\ts_lshl_b64 exec, exec, 1\n\tv_add_f64 %[acc], %[acc], s[28:29]\n\t"
                    "v_readlane_b32 s28, %[xlo], 28\n\tv_readlane_b32 s29, %[xhi], 28\n\ts_lshl_b64 exec, exec, 1\n\tv_add_f64 %[acc], %[acc], s[30:31]\n\t"
                    "v_readlane_b32 s30, %[xlo], 29\n\tv_readlane_b32 s31, %[xhi], 29\n\ts_lshl_b64 exec, exec, 1\n\tv_add_f64 %[acc], %[acc], s[20:21]\n\t"
                    "v_readlane_b32 s20, %[xlo], 30\n\tv_readlane_b32 s21, %[xhi], 30\n\ts_lshl_b64 exec, exec, 1\n\tv_add_f64 %[acc], %[acc], s[22:23]\n\t"
                    "v_readlane_b32 s22, %[xlo], 31\n\tv_readlane_b32 s23, %[xhi], 31\n\ts_lshl_b64 exec, exec, 1\n\tv_add_f64 %[acc], %[acc], s[24:25]\n\t"
                    "v_readlane_b32 s24, %[xlo], 32\n\tv_readlane_b32 s25, %[xhi], 32\n\ts_lshl_b64 exec, exec, 1\n\tv_add_f64 %[acc], %[acc], s[26:27]\n\t"
                    "v_readlane_b32 s26, %[xlo], 33\n\tv_readlane_b32 s27, %[xhi], 33\n\ts_lshl_b64 exec, exec, 1\n\tv_add_f64 %[acc], %[acc], s[28:29]\n\t"
                    "v_readlane_b32 s28, %[xlo], 34\n\tv_readlane_b32 s29, %[xhi], 34\n\ts_lshl_b64 exec, exec, 1\n\tv_add_f64 %[acc], %[acc], s[30:31]\n\t"
                    "v_readlane_b32 s30, %[xlo], 35\n\tv_readlane_b32 s31, %[xhi], 35\n\ts_lshl_b64 exec, exec, 1\n\tv_add_f64 %[acc], %[acc], s[20:21]\n\t"
                    "v_readlane_b32 s20, %[xlo], 36\n\tv_readlane_b32 s21, %[xhi], 36\n\ts_lshl_b64 exec, exec, 1\n\tv_add_f64 %[acc], %[acc], s[22:23]\n\t"
                    "v_readlane_b32 s22, %[xlo], 37\n\tv_readlane_b32 s23, %[xhi], 37\n\ts_lshl_b64 exec, exec, 1\n\tv_add_f64 %[acc], %[acc], s[24:25]\n\t"
                    "v_readlane_b32 s24, %[xlo], 38\n\tv_readlane_b32 s25, %[xhi], 38\n\ts_lshl_b64 exec, exec, 1\n\tv_add_f64 %[acc], %[acc], s[26:27]\n\t"
                    "v_readlane_b32 s26, %[xlo], 39\n\tv_readlane_b32 s27, %[xhi], 39\n\ts_lshl_b64 exec, exec, 1\n\tv_add_f64 %[acc], %[acc], s[28:29]\n\t"
                    "v_readlane_b32 s28, %[xlo], 40\n\tv_readlane_b32 s29, %[xhi], 40\n\ts_lshl_b64 exec, exec, 1\n\tv_add_f64 %[acc], %[acc], s[30:31]\n\t"
                    "v_readlane_b32 s30, %[xlo], 41\n\tv_readlane_b32 s31, %[xhi], 41\n\ts_lshl_b64 exec, exec, 1\n\tv_add_f64 %[acc], %[acc], s[20:21]\n\t"
                    "v_readlane_b32 s20, %[xlo], 42\n\tv_readlane_b32 s21, %[xhi], 42\n\ts_lshl_b64 exec, exec, 1\n\tv_add_f64 %[acc], %[acc], s[22:23]\n\t"
                    "v_readlane_b32 s22, %[xlo], 43\n\tv_readlane_b32 s23, %[xhi], 43\n\ts_lshl_b64 exec, exec, 1\n\tv_add_f64 %[acc], %[acc], s[24:25]\n\t"
                    "v_readlane_b32 s24, %[xlo], 44\n\tv_readlane_b32 s25, %[xhi], 44\n\ts_lshl_b64 exec, exec, 1\n\tv_add_f64 %[acc], %[acc], s[26:27]\n\t"
                    "v_readlane_b32 s26, %[xlo], 45\n\tv_readlane_b32 s27, %[xhi], 45\n\ts_lshl_b64 exec, exec, 1\n\tv_add_f64 %[acc], %[acc], s[28:29]\n\t"
                    "v_readlane_b32 s28, %[xlo], 46\n\tv_readlane_b32 s29, %[xhi], 46\n\ts_lshl_b64 exec, exec, 1\n\tv_add_f64 %[acc], %[acc], s[30:31]\n\t"
                    "v_readlane_b32 s30, %[xlo], 47\n\tv_readlane_b32 s31, %[xhi], 47\n\ts_lshl_b64 exec, exec, 1\n\tv_add_f64 %[acc], %[acc], s[20:21]\n\t"
                    "v_readlane_b32 s20, %[xlo], 48\n\tv_readlane_b32 s21, %[xhi], 48\n\ts_lshl_b64 exec, exec, 1\n\tv_add_f64 %[acc], %[acc], s[22:23]\n\t"
                    "v_readlane_b32 s22, %[xlo], 49\n\tv_readlane_b32 s23, %[xhi], 49\n\ts_lshl_b64 exec, exec, 1\n\tv_add_f64 %[acc], %[acc], s[24:25]\n\t"
                    "v_readlane_b32 s24, %[xlo], 50\n\tv_readlane_b32 s25, %[xhi], 50\n\ts_lshl_b64 exec, exec, 1\n\tv_add_f64 %[acc], %[acc], s[26:27]\n\t"
                    "v_readlane_b32 s26, %[xlo], 51\n\tv_readlane_b32 s27, %[xhi], 51\n\ts_lshl_b64 exec, exec, 1\n\tv_add_f64 %[acc], %[acc], s[28:29]\n\t"
                    "v_readlane_b32 s28, %[xlo], 52\n\tv_readlane_b32 s29, %[xhi], 52\n\ts_lshl_b64 exec, exec, 1\n\tv_add_f64 %[acc], %[acc], s[30:31]\n\t"
                    "v_readlane_b32 s30, %[xlo], 53\n\tv_readlane_b32 s31, %[xhi], 53\n\ts_lshl_b64 exec, exec, 1\n\tv_add_f64 %[acc], %[acc], s[20:21]\n\t"
                    "v_readlane_b32 s20, %[xlo], 54\n\tv_readlane_b32 s21, %[xhi], 54\n\ts_lshl_b64 exec, exec, 1\n\tv_add_f64 %[acc], %[acc], s[22:23]\n\t"
                    "v_readlane_b32 s22, %[xlo], 55\n\tv_readlane_b32 s23, %[xhi], 55\n\ts_lshl_b64 exec, exec, 1\n\tv_add_f64 %[acc], %[acc], s[24:25]\n\t"
                    "v_readlane_b32 s24, %[xlo], 56\n\tv_readlane_b32 s25, %[xhi], 56\n\ts_lshl_b64 exec, exec, 1\n\tv_add_f64 %[acc], %[acc], s[26:27]\n\t"
                    "v_readlane_b32 s26, %[xlo], 57\n\tv_readlane_b32 s27, %[xhi], 57\n\ts_lshl_b64 exec, exec, 1\n\tv_add_f64 %[acc], %[acc], s[28:29]\n\t"
                    "v_readlane_b32 s28, %[xlo], 58\n\tv_readlane_b32 s29, %[xhi], 58\n\ts_lshl_b64 exec, exec, 1\n\tv_add_f64 %[acc], %[acc], s[30:31]\n\t"
                    "v_readlane_b32 s30, %[xlo], 59\n\tv_readlane_b32 s31, %[xhi], 59\n\ts_lshl_b64 exec, exec, 1\n\tv_add_f64 %[acc], %[acc], s[20:21]\n\t"
                    "v_readlane_b32 s20, %[xlo], 60\n\tv_readlane_b32 s21, %[xhi], 60\n\ts_lshl_b64 exec, exec, 1\n\tv_add_f64 %[acc], %[acc], s[22:23]\n\t"
                    "v_readlane_b32 s22, %[xlo], 61\n\tv_readlane_b32 s23, %[xhi], 61\n\ts_lshl_b64 exec, exec, 1\n\tv_add_f64 %[acc], %[acc], s[24:25]\n\t"
                    "v_readlane_b32 s24, %[xlo], 62\n\tv_readlane_b32 s25, %[xhi], 62\n\ts_lshl_b64 exec, exec, 1\n\tv_add_f64 %[acc], %[acc], s[26:27]\n\t"
                    "v_readlane_b32 s26, %[xlo], 63\n\tv_readlane_b32 s27, %[xhi], 63\n\ts_lshl_b64 exec, exec, 1\n\tv_add_f64 %[acc], %[acc], s[28:29]\n\t"
                    "s_lshl_b64 exec, exec, 1\n\tv_add_f64 %[acc], %[acc], s[30:31]\n\t"
                    "s_lshl_b64 exec, exec, 1\n\tv_add_f64 %[acc], %[acc], s[20:21]\n\t"
                    "s_lshl_b64 exec, exec, 1\n\tv_add_f64 %[acc], %[acc], s[22:23]\n\t"
                    "s_lshl_b64 exec, exec, 1\n\tv_add_f64 %[acc], %[acc], s[24:25]\n\t"
                    "s_lshl_b64 exec, exec, 1\n\tv_add_f64 %[acc], %[acc], s[26:27]\n\t"
                    "s_mov_b64 exec, -1"
                    : [acc] "+v"(acc)
                    : [xlo] "v"(xlo), [xhi] "v"(xhi)
                    : "scc", "s20", "s21", "s22", "s23", "s24", "s25", "s26", "s27", "s28", "s29", "s30", "s31");
            }
#endif
            if (base + lane < Tx) cs[base + lane] = (float)acc;
            const long long bits = __builtin_bit_cast(long long, acc);
            const unsigned lo = (unsigned)__builtin_amdgcn_readlane((int)(bits & 0xffffffffll), 63);
            const unsigned hi = (unsigned)__builtin_amdgcn_readlane((int)(bits >> 32), 63);
            carry = __builtin_bit_cast(double, (long long)(((unsigned long long)hi << 32) | lo));
        }
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    }
    int n = 0, start = 0;
    bool in_sil = false;
    for (int base = 0; base < nwin; base += 64) {
        const int i = base + lane;
        const bool valid = i < nwin;
        float avg = 0.0f;
        if (valid) {
            if (k > 1) avg = (cs[i + k - 1] - (i > 0 ? cs[i - 1] : 0.0f)) / (float)k; // :510
            else avg = ps[i];
        }
        const int nv = min(64, nwin - base);
        const unsigned long long vmask = (nv == 64) ? ~0ull : ((1ull << nv) - 1ull);
        const unsigned long long bits = __ballot(valid && avg >= thr) & vmask; // :517
        int pos = 0;
        while (pos < nv) {
            const unsigned long long rest = (in_sil ? (~bits & vmask) : bits) >> pos; // next change of state
            if (rest == 0ull) break;
            const int j = pos + __builtin_ctzll(rest);
            if (!in_sil) { in_sil = true; start = base + j; }
            else {
                in_sil = false;
                int e = base + j + k - 1; // :530-531
                if (e > Tx) e = Tx;
                if (e - start >= k) { if (n >= cap) return -1; if (lane == 0) { out[2 * n] = start; out[2 * n + 1] = e; } ++n; }
            }
            pos = j + 1;
        }
    }
    if (in_sil && Tx - start >= k) { if (n >= cap) return -1; if (lane == 0) { out[2 * n] = start; out[2 * n + 1] = Tx; } ++n; } // :536-539
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    return n;
}

struct SegRec { int32_t a0, a1, t0, t1, is_sil; };

// the reference's serial control flow for one candidate utterance (run by one lane; `ps` may point into LDS)
// planner scratch: target SIL groups, audio silences, sub-silences of a piece, matches, segment records
struct PlanScratch {
    int32_t *groups, *aud, *sub, *match;
    SegRec *segs;
    int aud_cap;
    int groups_cap; // cooperative mode: pairs the LDS group / match arrays hold (the segment records: 2 groups_cap + 4)
    float *cs; // LDS, cooperative mode only
    int32_t *pbk; // LDS, cooperative mode only: the length bucket of every emitted piece (-1: not a DP piece)
    int32_t *gsub; // cooperative mode: the utterance's global sub-silence scratch, for a piece whose runs overflow the LDS array
    int gsub_cap;
};

// COOP: executed by every lane of the wavefront with identical control flow (all decisions depend on wave-uniform
// data); scratch lives in LDS, the heavy loops are lane-parallel and global side effects come from lane 0.
// !COOP: executed by lane 0 alone (utterances too large for the LDS arrays).
// Returns false (cooperative mode only, nothing written yet) when the audio silences do not fit the LDS array: the
// caller then plans the utterance with the global scratch, which holds the true worst case.
template <bool COOP>
__device__ bool plan_candidate(const AlignArgs &a, int b, const float *ps, const PlanScratch &sc, int lane)
{
    const bool writer = !COOP || lane == 0;
    auto silences = [&](const float *x, int Tx, float thr, int k, int32_t *out, int cap) {
        if (COOP) return detect_silence_w(x, Tx, thr, k, out, cap, sc.cs, lane);
        return detect_silence(x, Tx, thr, k, out, cap);
    };
    const DevParams &p = a.p;
    const int T = a.uT[b], S = a.uS[b];
#ifdef BFA_PLAN_STAMPS // (measurement builds, tools/plan_stamps.py: s_memrealtime at the phase boundaries, in the utterance's global scratch)
    unsigned long long *const gst = reinterpret_cast<unsigned long long *>(a.seg_scratch + (int64_t)b * a.seg_scratch_per_utt) + 1;
    auto stamp = [&](int k) { if (COOP && lane == 0) gst[k] = __builtin_amdgcn_s_memrealtime(); };
#else
    auto stamp = [&](int) {};
#endif
    stamp(1);
    const int32_t *tok = a.tokens + (int64_t)b * a.Smax;
    const int fallback_mode = -1 - a.umode[b];
    int32_t *groups = sc.groups, *aud = sc.aud, *sub = sc.sub, *match = sc.match;
    SegRec *segs = sc.segs;
    const int aud_cap = sc.aud_cap;
    int32_t *const pbk = sc.pbk;

    bool ok = true;
    // ---- _find_target_sil_groups :203-224
    int ng = 0;
    if (COOP) { // 64 targets per load (a serial walk over global memory pays one memory round trip per target)
        bool in_g = false;
        int st = 0;
        for (int base = 0; base < S; base += 64) {
            const int nv = min(64, S - base);
            const unsigned long long vmask = (nv == 64) ? ~0ull : ((1ull << nv) - 1ull);
            const unsigned long long bits = __ballot(base + lane < S && tok[min(base + lane, S - 1)] == p.sil) & vmask;
            int pos = 0;
            while (pos < nv) {
                const unsigned long long rest = (in_g ? (~bits & vmask) : bits) >> pos; // next change of state
                if (rest == 0ull) break;
                const int j = pos + __builtin_ctzll(rest);
                if (!in_g) { in_g = true; st = base + j; }
                else {
                    in_g = false;
                    if (ng + 1 >= sc.groups_cap) return false; // more SIL groups than the LDS arrays hold (nothing written yet): the global scratch
                    groups[2 * ng] = st; groups[2 * ng + 1] = base + j; ++ng;
                }
                pos = j + 1;
            }
        }
        if (in_g) {
            if (ng + 1 >= sc.groups_cap) return false;
            groups[2 * ng] = st; groups[2 * ng + 1] = S; ++ng;
        }
    } else {
        for (int i = 0; i < S;) {
            if (tok[i] == p.sil) { const int st = i; while (i < S && tok[i] == p.sil) ++i; groups[2 * ng] = st; groups[2 * ng + 1] = i; ++ng; }
            else ++i;
        }
    }
    if (ng == 0) ok = false; // :293-295
    stamp(2);
    int mf = p.anchors, na = 0;
    if (ok) { // :296-308
        na = silences(ps, T, 0.9f, mf, aud, aud_cap);
        if (na == 0 && S > 200) {
            double nt = 1.0 - (0.09 * (double)mf);
            if (nt < 0.05) nt = 0.05;
            na = silences(ps, T, (float)nt, mf, aud, aud_cap);
        }
        if (na == 0 && S > 200 && mf > 3) { mf = 3; na = silences(ps, T, 0.9f, mf, aud, aud_cap); }
        // More silence runs than the scratch holds (runs only need one silent window followed by a non-silent one, so
        // there can be about nwin / 2 of them -- P(SIL) oscillating around the threshold at every frame).  LDS scratch:
        // the caller repeats the plan with the global scratch (sized for that worst case).  Global scratch (cannot
        // happen with the documented workspace size): BFA_ITEM_TOO_LARGE, never a silently different alignment.
        if (na < 0) {
            if (COOP) return false;
            if (writer) { a.status[b] = BFA_ITEM_TOO_LARGE; a.items[b].kind = ITEM_FILL_BLANK; a.umode[b] = BFA_MODE_SEGMENTED; }
            return true;
        }
        if (na <= 0) ok = false; // :315-320
    }
    stamp(3);
    // ---- _match_silences :226-266
    int nm = 0;
    if (ok) {
        // COOP: the relative position of every audio silence once, one per lane (two float64 divisions each; the serial
        // loop below would evaluate them for every (group, silence) pair it visits) -- parked in the sub-silence scratch,
        // which is not in use before the pieces are emitted
        double *apos = reinterpret_cast<double *>(sub);
        if (COOP) {
            for (int ai = lane; ai < na; ai += 64) apos[ai] = (double)(aud[2 * ai] + aud[2 * ai + 1]) / 2.0 / (double)T;
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
            __builtin_amdgcn_wave_barrier();
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
        }
        int audio_idx = 0;
        for (int gi = 0; gi < ng; ++gi) {
            const double tp = (double)(groups[2 * gi] + groups[2 * gi + 1]) / 2.0 / (double)S;
            int best = -1;
            double bd = __builtin_inf();
            for (int ai = audio_idx; ai < na; ++ai) {
                const double ap = COOP ? apos[ai] : (double)(aud[2 * ai] + aud[2 * ai + 1]) / 2.0 / (double)T;
                const double d = __builtin_fabs(tp - ap);
                if (d < bd) { bd = d; best = ai; }
                else if (d > bd) break;
            }
            if (best >= 0 && bd < 0.3) { match[2 * nm] = gi; match[2 * nm + 1] = best; ++nm; audio_idx = best + 1; }
        }
        if (nm == 0) ok = false; // :324-325
    }
    // ---- segment list :328-354 and merge :357-369
    int ns = 0;
    if (ok) {
        int pa = 0, pt = 0;
        for (int i = 0; i < nm; ++i) {
            const int tg0 = groups[2 * match[2 * i]], tg1 = groups[2 * match[2 * i] + 1];
            const int as0 = aud[2 * match[2 * i + 1]], as1 = aud[2 * match[2 * i + 1] + 1];
            if (pa < as0 && pt < tg0) segs[ns++] = SegRec{pa, as0, pt, tg0, 0};
            else if (pa < as0) segs[ns++] = SegRec{pa, as0, pt, pt, 0};
            segs[ns++] = SegRec{as0, as1, tg0, tg1, 1};
            pa = as1; pt = tg1;
        }
        if (pa < T && pt < S) segs[ns++] = SegRec{pa, T, pt, S, 0};
        else if (pa < T) segs[ns++] = SegRec{pa, T, pt, pt, 0};
        int nmrg = 0;
        for (int i = 0; i < ns; ++i) {
            const SegRec s = segs[i];
            const int nf = s.a1 - s.a0, np = s.t1 - s.t0;
            if (!s.is_sil && np > 0 && nf < 20 && nmrg > 0) { // min_speech_frames = 20
                segs[nmrg - 1].a1 = s.a1; segs[nmrg - 1].t1 = s.t1; segs[nmrg - 1].is_sil = 0;
            } else segs[nmrg++] = s;
        }
        ns = nmrg;
    }
    // ---- validation pass: every speech segment must fit (:427-429), count the pieces
    int npieces = 0;
    int too_large = 0;
    if (ok) {
        for (int i = 0; i < ns; ++i) {
            const SegRec s = segs[i];
            const int n = s.a1 - s.a0;
            if (n <= 0) continue;
            ++npieces;
            const int nt = s.t1 - s.t0;
            if (s.is_sil || nt == 0) continue;
            const int psx = max(0, s.a0 - 3), pex = min(T, s.a1 + 3);
            const int Ts = pex - psx;
            int stride = 4;
            if ((double)(stride * nt + 1) > (double)Ts * 0.9) stride = 3;
            if ((double)(stride * nt + 1) > (double)Ts * 0.8) stride = 2;
            const int L = stride * nt + 1;
            if ((double)L > (double)Ts * 1.2) { ok = false; break; }
            if (L > BIG_MAX_L) too_large = 1;
        }
        if (npieces == 0) ok = false; // :454-455
    }
    stamp(4);
    if (!ok) { // keep the standard-mode fallback item -- which may be the "audio too short" error (:161-165)
        if (writer) {
            if (fallback_mode == BFA_FALLBACK_TOO_SHORT) { a.status[b] = BFA_ITEM_TOO_SHORT; a.umode[b] = BFA_MODE_EMPTY; }
            else a.umode[b] = fallback_mode;
        }
        return true;
    }
    if (too_large) { if (writer) { a.status[b] = BFA_ITEM_TOO_LARGE; a.items[b].kind = ITEM_FILL_BLANK; a.umode[b] = BFA_MODE_SEGMENTED; } return true; }
    // pieces are CONCATENATED (:453-467): audio silences may overlap, so a piece's output position is the
    // running length, not its audio position; the result is truncated / blank-padded to T frames
    npieces += 1; // room for the blank tail
    int base = writer ? atomicAdd(&a.counters[0], npieces) : 0;
    if (COOP) base = __builtin_amdgcn_readfirstlane(base);
    if (base + npieces > a.item_cap) { // cannot happen with the documented workspace size
        if (writer) { a.status[b] = BFA_ITEM_TOO_LARGE; a.items[b].kind = ITEM_FILL_BLANK; a.umode[b] = BFA_MODE_SEGMENTED; }
        return true;
    }
    stamp(5);
    // ---- emit one item per piece (:377-451)
    int64_t bp_off = (int64_t)b * a.bp_per_utt;
    int anch_used = 0;
    uint8_t *apool = a.anchor + (int64_t)b * a.anchor_per_utt;
    int slot = base;
    int w = 0; // frames written so far
    for (int i = 0; i < ns; ++i) {
        const SegRec s = segs[i];
        const int n = s.a1 - s.a0;
        if (n <= 0) continue;
        Item it;
        it.kind = ITEM_NONE; it.utt = b; it.row0 = s.a0; it.Ts = n; it.tok0 = s.t0; it.nt = s.t1 - s.t0; it.stride = 0;
        it.L = 0; it.bw = 0; it.out0 = s.a0; it.nout = n; it.pad_left = 0; it.final_state = FINAL_NOT_COMPUTED; it.anch_off = -1; it.win = 0; it.split = 0; it.xw = XW_FAST; it.t_tail = 0;
        it.bp_off = bp_off;
        if (s.is_sil) it.kind = ITEM_FILL_SIL;          // :382-397
        else if (it.nt == 0) it.kind = ITEM_FILL_BLANK; // :409-412
        else {
            const int psx = max(0, s.a0 - 3), pex = min(T, s.a1 + 3); // :401-403
            const int Ts = pex - psx;
            int stride = 4; // :423-426
            if ((double)(stride * it.nt + 1) > (double)Ts * 0.9) stride = 3;
            if ((double)(stride * it.nt + 1) > (double)Ts * 0.8) stride = 2;
            const int L = stride * it.nt + 1;
            it.kind = ITEM_DP; it.row0 = psx; it.Ts = Ts; it.stride = stride; it.L = L;
            it.bw = (L > 60) ? ((L / 3 > 30) ? L / 3 : 30) : 0; // :441
            it.pad_left = s.a0 - psx;
            {   // backpointers of the piece.  The wide classes (R >= 4) are aligned by two consumer waves (k_dp5_any, round 6), whose
                // layout -- 2 R lane masks per frame -- is that of the full layout with ALL 64 lanes, 16-byte aligned; the
                // narrow ones keep the dwords of the lanes that own a state.  (bfa_workspace_bytes bounds either: it counts 64
                // lanes per quad of every piece.)
                const int Rp = r_class_for_L(L);
                const int64_t quads = (Ts + 3) / 4;
                if (Rp >= 4) bp_off += (quads * bp_words_for_R(Rp) * 64 + quads + 3) & ~(int64_t)3;
                else bp_off = (bp_off + bp_dwords(Ts, L) + 3) & ~(int64_t)3;
            }
            // sub-silences of the padded slice get +5 on blank and a re-normalisation, once per
            // (possibly overlapping) detected segment (:415-419, :543-561)
            int nsub = silences(ps + psx, Ts, 0.8f, mf, sub, aud_cap);
            const int32_t *subr = sub;
            if (COOP && nsub < 0) { // more runs than the LDS array holds: lane 0 detects them again into the global scratch
                int n2 = 0;
                if (lane == 0) n2 = detect_silence(ps + psx, Ts, 0.8f, mf, sc.gsub, sc.gsub_cap);
                nsub = __builtin_amdgcn_readfirstlane(n2);
                __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
                __builtin_amdgcn_wave_barrier();
                __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
                subr = sc.gsub;
            }
            // scratch or anchor pool exhausted (cannot happen with the documented workspace size): flag the utterance
            if ((nsub < 0 || (nsub > 0 && anch_used + Ts > a.anchor_per_utt)) && writer) a.status[b] = BFA_ITEM_TOO_LARGE;
            if (nsub > 0 && anch_used + Ts <= a.anchor_per_utt) {
                uint8_t *ac = apool + anch_used;
                if (COOP) { // one frame per lane: how many detected segments cover it
                    for (int f = lane; f < Ts; f += 64) {
                        int cnt = 0;
                        for (int q = 0; q < nsub; ++q) cnt += (f >= subr[2 * q] && f < subr[2 * q + 1]) ? 1 : 0;
                        ac[f] = (uint8_t)cnt;
                    }
                } else {
                    for (int f = 0; f < Ts; ++f) ac[f] = 0;
                    for (int q = 0; q < nsub; ++q)
                        for (int f = subr[2 * q]; f < subr[2 * q + 1]; ++f) ac[f] = (uint8_t)(ac[f] + 1);
                }
                it.anch_off = anch_used;
                anch_used += Ts;
            }
        }
        it.out0 = w;
        it.nout = min(n, max(0, T - w));
        w += n;
        // k_dp4_any starts the long pieces first (AlignArgs::piece_list: one list per length bucket).  COOP: the bucket of
        // every piece is parked in LDS and the lists are appended to after the loop -- ONE returning atomic per bucket and
        // utterance, all of them in flight together, instead of one memory round trip per piece in the serial chain
        const int bk = (it.kind == ITEM_DP) ? piece_bucket(it.Ts, a.Tmax) : -1;
        if (COOP) { if (lane == 0) pbk[slot - base] = bk; }
        if (writer) {
            a.items[slot] = it;
            if (!COOP && bk >= 0) a.piece_list[(int64_t)bk * a.item_cap + atomicAdd(&a.counters[PIECE_CNT0 + bk], 1)] = slot;
        }
        ++slot;
    }
    stamp(6);
    if (COOP) {
        const int nemit = slot - base; // (<= 2 * groups_cap + 1 entries of pbk)
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
        for (int j0 = 0; j0 < nemit; j0 += 64) {
            const int j = j0 + lane;
            const int mybk = (j < nemit) ? pbk[j] : -1;
            int cnt_mine = 0;   // lane k: how many pieces of this chunk go to bucket k
            int rank = 0;       // this lane's piece: its position among the chunk's pieces of the same bucket
#pragma unroll
            for (int k = 0; k < PIECE_BUCKETS; ++k) {
                const unsigned long long m = __ballot(mybk == k);
                if (lane == k) cnt_mine = __builtin_popcountll(m);
                if (mybk == k) rank = __builtin_popcountll(m & ((1ull << lane) - 1ull));
            }
            int start_mine = 0;
            if (lane < PIECE_BUCKETS && cnt_mine > 0) start_mine = atomicAdd(&a.counters[PIECE_CNT0 + lane], cnt_mine);
            const int st = __shfl(start_mine, mybk < 0 ? 0 : mybk);
            if (mybk >= 0) a.piece_list[(int64_t)mybk * a.item_cap + st + rank] = base + j;
        }
    }
    {   // :461-464 pad with blank / -1 up to T (also fills unused reserved slots)
        Item it;
        it.kind = (w < T) ? ITEM_FILL_BLANK : ITEM_NONE; it.utt = b; it.row0 = 0; it.Ts = 0; it.tok0 = 0; it.nt = 0;
        it.stride = 0; it.L = 0; it.bw = 0; it.out0 = min(w, T); it.nout = max(0, T - w); it.pad_left = 0;
        it.final_state = 0; it.anch_off = -1; it.win = 0; it.split = 0; it.xw = XW_FAST; it.t_tail = 0; it.bp_off = 0;
        while (slot < base + npieces) { if (writer) a.items[slot] = it; ++slot; }
    }
    if (writer) {
        a.items[b].kind = ITEM_NONE; // the standard-mode fallback item is not needed
        a.umode[b] = BFA_MODE_SEGMENTED;
    }
    stamp(7);
#ifdef BFA_PLAN_STAMPS
    if (COOP && lane == 0) { gst[8] = ((unsigned long long)(unsigned)npieces << 32) | (unsigned)na; gst[9] = ((unsigned long long)(unsigned)blockIdx.x << 32) | (unsigned)T; gst[-1] = 0x5a5a1234c3c3abcdull; gst[10] = 0x3c3c4321a5a5dcbaull; }
#endif
    return true;
}

// one wavefront per candidate, planning with LDS scratch (the planner walks its run / group / segment arrays several
// times with serial, dependent accesses -- from global memory that latency was the whole kernel time).  Utterances whose
// worst-case counts exceed the LDS arrays use the global scratch instead.
#ifndef BFA_PLAN_WGS
#define BFA_PLAN_WGS 2048 // workgroups of k_plan_seg = eight per CU (compile-time A/B knob; profiles/r05_plan_grid_ab.txt)
#endif
constexpr int PLAN_LDS_FRAMES = 2048;
constexpr int PLAN_LDS_SILS = 192;  // audio silences / sub-silences (pairs)
constexpr int PLAN_LDS_GROUPS = 64; // target SIL groups / matches (pairs)

// `lds_frames` (a multiple of 64, <= PLAN_LDS_FRAMES): frames of the staged cumulative sums -- sized by the batch's Tmax,
// so that a batch of short utterances keeps more planners resident per CU (they are latency chains, LDS is their limit).
// `sils_cap` / `groups_cap` (<= PLAN_LDS_SILS / PLAN_LDS_GROUPS): the run and group arrays, sized by the launcher from the
// batch's Tmax / Smax as well.  With the fixed maxima a planner held 9.7 KB and the sixteen planners of a CU's share of a
// 4096-utterance batch held ALL of its LDS: a K1 launch of the other head that ran beside them got no workgroup in until
// they left (profiles/r05_stagger_timeline.txt: k_dp4_any 390 us alone, 770-880 us beside a planner).
__global__ __launch_bounds__(64) void k_plan_seg(AlignArgs a, int lds_frames, int sils_cap, int groups_cap)
{
    extern __shared__ __attribute__((aligned(16))) float plan_dyn[];
    float *scs = plan_dyn; // float32(cumsum) of the vector under the sliding window (detect_silence_w)
    float *sps = plan_dyn + lds_frames; // P(SIL) of the utterance: every pass of the planner reads it from here (round 5; until
                                        // then each pass, the top-level one and one per piece, began with a memory round trip)
    int32_t *s_aud = (int32_t *)(plan_dyn + 2 * lds_frames);  // [2 * sils_cap]   (8-byte aligned: lds_frames is a multiple of 64)
    int32_t *s_sub = s_aud + 2 * sils_cap;                // [2 * sils_cap]   (doubles as sils_cap float64; sils_cap is even)
    int32_t *s_groups = s_sub + 2 * sils_cap;             // [2 * groups_cap]
    int32_t *s_match = s_groups + 2 * groups_cap;         // [2 * groups_cap]
    SegRec *s_segs = (SegRec *)(s_match + 2 * groups_cap); // [2 * groups_cap + 4]
    int32_t *s_pbk = (int32_t *)(s_segs + 2 * groups_cap + 4); // [2 * groups_cap + 4]
    const int lane = threadIdx.x & 63;
    // (wave priority 2 / 3 -- the DP consumers that run beside a planner are at 3 -- measured: real text one call at a time
    // 1.709 -> 1.744 / 1.741 ms, profiles/r05_plan_prio_ab.txt)
    for (int b = blockIdx.x; b < a.B; b += gridDim.x) {
        if (!a.ucand[b]) continue; // (wave-uniform: one utterance per wavefront)
        const int T = a.uT[b], S = a.uS[b];
        const float *ps = a.psil + (int64_t)b * a.Tmax;
        // budget of the cooperative path: T / min_k + 1 silence runs (min_k = anchors, or 3 on the S > 200 retry) -- real
        // posteriors stay far below it; the true worst case is ~nwin / 2 overlapping runs, which overflows the scratch and
        // is reported per utterance (see plan_candidate).  SIL groups alternate with other tokens: (S + 1) / 2 at most
        // Round 6: the cooperative path is tried whenever the frames fit; the run / group arrays are checked WHILE planning
        // (plan_candidate returns false before anything is written).  Until then the test was the worst case -- T / min_k + 2
        // silence runs, (S + 1) / 2 + 1 groups against arrays of 128 / 64 -- which sent every utterance of more than 1 260 frames or
        // 125 targets to the one-lane path below: 445 us for ONE 20-s utterance (profiles/r06_latency_realtext_timeline_b1_before.txt),
        // a quarter of the C5 proxy's utterances, 1.6-2.2 ms of planner per call; real posteriors hold a handful of either.
        (void)S;
        const bool coop = T <= lds_frames;
        __builtin_amdgcn_wave_barrier(); // the previous candidate's readers are done
        PlanScratch sc;
        bool planned = false;
        int32_t *const gscr = a.seg_scratch + (int64_t)b * a.seg_scratch_per_utt; // [groups | aud | sub | match | segs]
        if (coop) { // wave-uniform
            // P(SIL) itself stays in global memory: every pass over it is a 64-wide coalesced read one slice ahead of
            // the serial sum, and without a second staged vector all 16 planners of a CU's share of a 4096-utterance
            // batch are resident at once (LDS was the limit: 12 KB each at T = 1000, 2.5 rounds of ~70 us)
            sc.groups = s_groups; sc.aud = s_aud; sc.sub = s_sub; sc.match = s_match; sc.segs = s_segs;
#ifdef BFA_PLAN_STAMPS
            if (lane == 0) (reinterpret_cast<unsigned long long *>(gscr) + 1)[0] = __builtin_amdgcn_s_memrealtime();
#endif
            sc.aud_cap = sils_cap; sc.groups_cap = groups_cap; sc.cs = scs; sc.pbk = s_pbk;
#pragma unroll 8
            for (int i = lane; i < T; i += 64) sps[i] = ps[i];
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
            __builtin_amdgcn_wave_barrier();
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
            sc.gsub = gscr + 2 * (a.Smax + 2) + 2 * (a.Tmax + 2); sc.gsub_cap = a.Tmax + 2;
            planned = plan_candidate<true>(a, b, sps, sc, lane);
        }
        if (!planned && lane == 0) { // too large for the LDS arrays (by the admission test, or found out while planning)
            int32_t *scr = gscr;
            sc.groups = scr; scr += 2 * (a.Smax + 2);
            sc.aud = scr; scr += 2 * (a.Tmax + 2);
            sc.sub = scr; scr += 2 * (a.Tmax + 2);
            sc.match = scr; scr += 2 * (a.Smax + 2);
            sc.segs = (SegRec *)scr;
            sc.aud_cap = a.Tmax + 2; sc.groups_cap = a.Smax + 2; sc.cs = nullptr; sc.gsub = nullptr; sc.gsub_cap = 0; sc.pbk = nullptr;
            (void)plan_candidate<false>(a, b, ps, sc, lane);
        }
    }
}

} // namespace bfa

extern "C" int bfa_launch_silprob3_nk2(const bfa::AlignArgs *args, hipStream_t stream);
extern "C" int bfa_launch_silprob3_nk5(const bfa::AlignArgs *args, hipStream_t stream);

extern "C" void bfa_launch_segment_plan(const bfa::AlignArgs *args, hipStream_t stream)
{
    using namespace bfa;
    const AlignArgs &a = *args;
    const int nk = (a.C + 15) / 16;
    const int grid = 2048;
    // the two head widths take the sixteen-rows-per-pass block (bfa_dp3.inc)
    if (a.C == 17 && bfa_launch_silprob3_nk2(&a, stream)) { /* launched */ }
    else if (a.C == 67 && bfa_launch_silprob3_nk5(&a, stream)) { /* launched */ }
    else if (nk <= 2) hipLaunchKernelGGL(k_silprob<2>, dim3(grid), dim3(64), 0, stream, a);
    else if (nk <= 5) hipLaunchKernelGGL(k_silprob<5>, dim3(grid), dim3(64), 0, stream, a);
    else hipLaunchKernelGGL(k_silprob<8>, dim3(grid), dim3(64), 0, stream, a);
    // (cutting the batch into slices so that the planner of one slice runs beside the row pass of the next was measured
    // and lost: the row pass slows down by more than the planner hides, 1.12 -> 1.17-1.23 ms per step)
    const int lds_frames = a.Tmax >= PLAN_LDS_FRAMES ? PLAN_LDS_FRAMES : ((a.Tmax + 63) / 64) * 64;
    // the run / group arrays by the batch's shape (the admission test of the cooperative path, with Tmax / Smax for T / S)
    const int anchors = a.p.anchors > 0 ? a.p.anchors : 1;
    const int min_k = (a.Smax > 200 && anchors > 3) ? 3 : anchors;
    int sils_cap = lds_frames / min_k + 2, groups_cap = (a.Smax + 1) / 2 + 1;
    sils_cap = sils_cap > PLAN_LDS_SILS ? PLAN_LDS_SILS : ((sils_cap + 1) & ~1);
    groups_cap = groups_cap > PLAN_LDS_GROUPS ? PLAN_LDS_GROUPS : ((groups_cap + 1) & ~1);
    if (sils_cap < 8) sils_cap = 8;
    if (groups_cap < 4) groups_cap = 4;
    const size_t lds = 2 * (size_t)lds_frames * sizeof(float) + (size_t)(4 * sils_cap + 4 * groups_cap + 2 * groups_cap + 4) * sizeof(int32_t) +
                       (size_t)(2 * groups_cap + 4) * sizeof(SegRec);
    // Eight planners per CU, not one per utterance: a resident planner is a latency chain that holds 56 vector registers
    // and 11 KB of LDS, and sixteen of them per CU left the kernels of the other head (or of the next call) that run beside
    // them a third of their occupancy -- realtext one call at a time 1.90 / 1.87 / 1.84 / 1.79 ms with 16 / 12 / 8 / 4 per CU
    // on the planner of rounds 3-4; with P(SIL) staged in LDS and the batched list appends 8 per CU is as fast as 16 for
    // one head alone and 2 % faster with three calls in flight (profiles/r05_plan_grid_ab.txt)
    constexpr int plan_grid = BFA_PLAN_WGS;
    hipLaunchKernelGGL(k_plan_seg, dim3(a.B < plan_grid ? a.B : plan_grid), dim3(64), lds, stream, a, lds_frames, sils_cap, groups_cap);
}
