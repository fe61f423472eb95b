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
    const int n_cand = a.counters[1];
    const int nchunk = (a.Tmax + 63) / 64;
    const int sil_k = p.sil >> 4, sil_j = p.sil & 15;
    for (int u = blockIdx.x; u < n_cand * nchunk; u += gridDim.x) {
        const int b = a.cand[u / nchunk];
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
            boost_floor<NK>(x, rl, p.boost != 0, p.enforce != 0);
            float v = 0.0f;
#pragma unroll
            for (int k = 0; k < NK; ++k) if (k == sil_k) v = x[k];
            if (j == sil_j && t < T) ps[t] = expf_u10(v);
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

struct SegRec { int32_t a0, a1, t0, t1, is_sil; };

__global__ void k_plan_seg(AlignArgs a)
{
    const DevParams &p = a.p;
    const int n_cand = a.counters[1];
    for (int ci = blockIdx.x * blockDim.x + threadIdx.x; ci < n_cand; ci += gridDim.x * blockDim.x) {
        const int b = a.cand[ci];
        const int T = a.uT[b], S = a.uS[b];
        const int32_t *tok = a.tokens + (int64_t)b * a.Smax;
        const float *ps = a.psil + (int64_t)b * a.Tmax;
        const int fallback_mode = -1 - a.umode[b];
        // scratch carve
        int32_t *scr = a.seg_scratch + (int64_t)b * a.seg_scratch_per_utt;
        int32_t *groups = scr; scr += 2 * (a.Smax + 2);
        int32_t *aud = scr; scr += 2 * (a.Tmax + 2);
        int32_t *sub = scr; scr += 2 * (a.Tmax + 2);
        int32_t *match = scr; scr += 2 * (a.Smax + 2);
        SegRec *segs = (SegRec *)scr;
        const int aud_cap = a.Tmax + 2;

        bool ok = true;
        // ---- _find_target_sil_groups :203-224
        int ng = 0;
        for (int i = 0; i < S;) {
            if (tok[i] == p.sil) { const int st = i; while (i < S && tok[i] == p.sil) ++i; groups[2 * ng] = st; groups[2 * ng + 1] = i; ++ng; }
            else ++i;
        }
        if (ng == 0) ok = false; // :293-295
        int mf = p.anchors, na = 0;
        if (ok) { // :296-308
            na = detect_silence(ps, T, 0.9f, mf, aud, aud_cap);
            if (na == 0 && S > 200) {
                double nt = 1.0 - (0.09 * (double)mf);
                if (nt < 0.05) nt = 0.05;
                na = detect_silence(ps, T, (float)nt, mf, aud, aud_cap);
            }
            if (na == 0 && S > 200 && mf > 3) { mf = 3; na = detect_silence(ps, T, 0.9f, mf, aud, aud_cap); }
            if (na <= 0) ok = false; // :315-320
        }
        // ---- _match_silences :226-266
        int nm = 0;
        if (ok) {
            int audio_idx = 0;
            for (int gi = 0; gi < ng; ++gi) {
                const double tp = (double)(groups[2 * gi] + groups[2 * gi + 1]) / 2.0 / (double)S;
                int best = -1;
                double bd = __builtin_inf();
                for (int ai = audio_idx; ai < na; ++ai) {
                    const double ap = (double)(aud[2 * ai] + aud[2 * ai + 1]) / 2.0 / (double)T;
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
        if (!ok) { a.umode[b] = fallback_mode; continue; }
        if (too_large) { a.status[b] = BFA_ITEM_TOO_LARGE; a.items[b].kind = ITEM_FILL_BLANK; a.umode[b] = BFA_MODE_SEGMENTED; continue; }
        // pieces are CONCATENATED (:453-467): audio silences may overlap, so a piece's output position is the
        // running length, not its audio position; the result is truncated / blank-padded to T frames
        npieces += 1; // room for the blank tail
        const int base = atomicAdd(&a.counters[0], npieces);
        if (base + npieces > a.item_cap) { // cannot happen with the documented workspace size
            a.status[b] = BFA_ITEM_TOO_LARGE; a.items[b].kind = ITEM_FILL_BLANK; a.umode[b] = BFA_MODE_SEGMENTED; continue;
        }
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
            it.L = 0; it.bw = 0; it.out0 = s.a0; it.nout = n; it.pad_left = 0; it.final_state = FINAL_NOT_COMPUTED; it.anch_off = -1; it.win = 0; it.pad_ = 0;
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
                bp_off += bp_dwords(Ts, L);
                // sub-silences of the padded slice get +5 on blank and a re-normalisation, once per
                // (possibly overlapping) detected segment (:415-419, :543-561)
                const int nsub = detect_silence(ps + psx, Ts, 0.8f, mf, sub, aud_cap);
                if (nsub > 0 && anch_used + Ts <= a.anchor_per_utt) {
                    uint8_t *ac = apool + anch_used;
                    for (int f = 0; f < Ts; ++f) ac[f] = 0;
                    for (int q = 0; q < nsub; ++q)
                        for (int f = sub[2 * q]; f < sub[2 * q + 1]; ++f) ac[f] = (uint8_t)(ac[f] + 1);
                    it.anch_off = anch_used;
                    anch_used += Ts;
                }
            }
            it.out0 = w;
            it.nout = min(n, max(0, T - w));
            w += n;
            a.items[slot++] = it;
        }
        {   // :461-464 pad with blank / -1 up to T (also fills unused reserved slots)
            Item it;
            it.kind = (w < T) ? ITEM_FILL_BLANK : ITEM_NONE; it.utt = b; it.row0 = 0; it.Ts = 0; it.tok0 = 0; it.nt = 0;
            it.stride = 0; it.L = 0; it.bw = 0; it.out0 = min(w, T); it.nout = max(0, T - w); it.pad_left = 0;
            it.final_state = 0; it.anch_off = -1; it.win = 0; it.pad_ = 0; it.bp_off = 0;
            while (slot < base + npieces) a.items[slot++] = it;
        }
        a.items[b].kind = ITEM_NONE; // the standard-mode fallback item is not needed
        a.umode[b] = BFA_MODE_SEGMENTED;
    }
}

} // namespace bfa

extern "C" void bfa_launch_segment_plan(const bfa::AlignArgs *args, hipStream_t stream)
{
    using namespace bfa;
    const AlignArgs &a = *args;
    const int nk = (a.C + 15) / 16;
    const int grid = 2048;
    if (nk <= 2) hipLaunchKernelGGL(k_silprob<2>, dim3(grid), dim3(64), 0, stream, a);
    else if (nk <= 5) hipLaunchKernelGGL(k_silprob<5>, dim3(grid), dim3(64), 0, stream, a);
    else hipLaunchKernelGGL(k_silprob<8>, dim3(grid), dim3(64), 0, stream, a);
    hipLaunchKernelGGL(k_plan_seg, dim3((a.B + 63) / 64 < 1024 ? (a.B + 63) / 64 : 1024), dim3(64), 0, stream, a);
}
