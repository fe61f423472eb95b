// bfa_assort.hpp -- assort_frames (forced_alignment.py:777-834) for one utterance by one wavefront, forward over the
// framewise arrays: used by the K3a kernel (k_assort) and by K2 for the items it finishes itself (bfa_backtrace.hip).
#pragma once
#include <hip/hip_runtime.h>

#include "bfa_math.hpp"
#include "bfa_types.hpp"

namespace bfa {

// U = 64-frame slices in flight (8 in the K3a kernel; 1 where the pass is a rare path and registers are short)
template <int U = 8>
__device__ __forceinline__ void assort_utterance(const AlignArgs &a, int b, int lane)
{
    const DevParams &p = a.p;
    int32_t *ph = a.frame_ph + (int64_t)b * a.Tmax;
    int32_t *ix = a.frame_idx + (int64_t)b * a.Tmax;
    const int T = a.uT[b];
    const int st = a.status[b];
    const bool none = (st != BFA_ITEM_OK) || (a.uS[b] == 0 && !p.simple);
    const int Tr = none ? 0 : T; // frames that take part in the run-length encoding
    for (int t = Tr + lane; t < a.Tmax; t += 64) { ph[t] = p.blank; ix[t] = -1; }
    bfa_segment *out = a.segs + (int64_t)b * a.seg_cap;
    int count = 0;
    int run_start = 0, run_ph = 0, run_ix = 0; // the open run (wave-uniform)
    // the framewise arrays are read eight 64-frame slices at a time: the slices do not depend on each other,
    // only the run bookkeeping does, and one load round trip per slice would be the whole kernel time
    for (int base0 = 0; base0 < Tr; base0 += 64 * U) {
        int vph[U], vix[U];
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const int t = base0 + 64 * u + lane;
            vph[u] = (t < Tr) ? ph[t] : 0;
            vix[u] = (t < Tr) ? ix[t] : 0;
        }
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const int base = base0 + 64 * u;
            if (base >= Tr) break; // wave-uniform
            const int t = base + lane;
            const bool in = t < Tr;
            const int cph = vph[u], cix = vix[u];
            int pph = __builtin_amdgcn_update_dpp(0, cph, DPP_WAVE_SHR1, 0xf, 0xf, true); // lane l <- lane l-1
            int pix = __builtin_amdgcn_update_dpp(0, cix, DPP_WAVE_SHR1, 0xf, 0xf, true);
            if (lane == 0) { pph = run_ph; pix = run_ix; }
            const bool is_start = in && (t == 0 || cph != pph || cix != pix); // :798-801
            const unsigned long long m = __ballot(is_start);
            // a start at t>0 closes the run that began at the previous start (shuffles stay convergent)
            const unsigned long long below = m & ((1ull << lane) - 1ull);
            const int src = below ? (63 - __builtin_clzll(below)) : 0;
            const int ps = below ? (base + src) : run_start;
            const int pp = pph, pi = pix; // every frame of the closing run carries its (phoneme, index)
            const bool closes = is_start && t > 0;
            bool emit = false;
            if (closes) {
                const int len = t - ps;
                if (pp == p.blank) emit = (!p.ignore_noise) && (len > p.max_blanks); // :819-827
                else emit = true;                                                    // :830-831
            }
            const unsigned long long em = __ballot(emit);
            if (emit) {
                const int slot = count + __builtin_popcountll(em & ((1ull << lane) - 1ull));
                if (slot < a.seg_cap) { bfa_segment sg; sg.phoneme = pp; sg.start = ps; sg.end = t; sg.target_idx = pi; out[slot] = sg; }
            }
            count += __builtin_popcountll(em);
            const int last = m ? (63 - __builtin_clzll(m)) : 0;
            const int lph = __builtin_amdgcn_readlane(cph, last), lix = __builtin_amdgcn_readlane(cix, last);
            if (m) { run_start = base + last; run_ph = lph; run_ix = lix; }
        }
    }
    if (Tr > 0) { // close the final run
        const int len = Tr - run_start;
        bool emit;
        if (run_ph == p.blank) emit = (!p.ignore_noise) && (len > p.max_blanks);
        else emit = true;
        if (emit) {
            if (lane == 0 && count < a.seg_cap) { bfa_segment sg; sg.phoneme = run_ph; sg.start = run_start; sg.end = Tr; sg.target_idx = run_ix; out[count] = sg; }
            count += 1;
        }
    }
    if (lane == 0) {
        if (count > a.seg_cap) { a.seg_count[b] = a.seg_cap; if (st == BFA_ITEM_OK) a.status[b] = BFA_ITEM_SEG_OVERFLOW; }
        else a.seg_count[b] = count;
        if (a.mode) { const int md = a.umode[b]; a.mode[b] = md < 0 ? (-1 - md) : md; }
    }
}

} // namespace bfa
