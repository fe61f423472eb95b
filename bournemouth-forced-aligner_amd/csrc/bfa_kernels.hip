// bfa_kernels.hip -- hand-written gfx950 (CDNA4, wave64) kernels of the forced-alignment core.
//
//   k_plan        per-utterance planning  (forced_alignment.py:87-199 mode selection, :29-83 target mask)
//   k_dp<NK>      K1: fused boost + log_softmax + floor + banded CTC Viterbi forward, one DP per
//                 wavefront, CTC states in registers (R per lane), 2-bit backpointers to HBM
//                 (forced_alignment.py:563-653)
//   k_backtrace   K2: walks the packed backpointers from the final state K1 chose and writes the
//                 framewise assignment (forced_alignment.py:686-700)
//   k_assort      K3a: run-length encoding of the framewise assignment (forced_alignment.py:777-834)
//   k_conf        K3b: confidence pass with the reference's in-place aliasing (utils.py:70-113)
//
// No MFMA anywhere: this is a memory-streaming scan with ~13 float ops per CTC state per frame.
#include <hip/hip_runtime.h>

#include "bfa_math.hpp"
#include "bfa_types.hpp"

#pragma clang fp contract(off)

namespace bfa {

// LDS written by some lanes of a wavefront and read by others of the SAME wavefront: the hardware
// executes one wave's LDS operations in order, this only stops the compiler from reordering.
__device__ __forceinline__ void wave_lds_sync()
{
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}

// =================================================================================================
// k_plan : one thread per utterance
// =================================================================================================
__device__ __forceinline__ int band_standard(int L) { return (L > 60) ? ((L / 4 > 20) ? L / 4 : 20) : 0; } // :190, :976

__global__ void k_plan(AlignArgs a)
{
    const int b = blockIdx.x * blockDim.x + threadIdx.x;
    if (b == 0) a.counters[0] = a.B; // one item slot per utterance; the segmented planner appends more
    if (b >= a.B) return;
    const DevParams &p = a.p;
    const int Traw = a.T_len ? a.T_len[b] : a.Tmax;
    int T = Traw;
    if (T > a.Tmax) T = a.Tmax; // python slicing clamps (forced_alignment.py:887)
    if (T < 0) T = 0;
    int S = a.S_len[b];
    if (S > a.Smax) S = a.Smax;
    if (S < 0) S = 0;
    a.uT[b] = T;
    a.uS[b] = S;

    uint32_t m[MASK_WORDS];
#pragma unroll
    for (int w = 0; w < MASK_WORDS; ++w) m[w] = 0u;
    int status = BFA_ITEM_OK;
    bool has_sil = false;
    const int32_t *tk = a.tokens + (int64_t)b * a.Smax;
    for (int j = 0; j < S; ++j) {
        const int t = tk[j];
        if (t < 0 || t >= a.C) { status = BFA_ITEM_BAD_TOKEN; continue; }
        if (t == p.sil) has_sil = true;
        if (t == p.blank) continue; // :45
        m[t >> 5] |= 1u << (t & 31);
    }
#pragma unroll
    for (int w = 0; w < MASK_WORDS; ++w) a.umask[(int64_t)b * MASK_WORDS + w] = m[w];

    Item it;
    it.kind = ITEM_NONE; it.utt = b; it.row0 = 0; it.Ts = T; it.tok0 = 0; it.nt = S; it.stride = 0; it.L = 0;
    it.bw = 0; it.out0 = 0; it.nout = T; it.pad_left = 0; it.final_state = 0; it.anchored = 0;
    it.bp_off = (int64_t)b * a.bp_per_utt;
    int mode = BFA_MODE_EMPTY;

    if (status != BFA_ITEM_OK) {
        it.kind = ITEM_FILL_BLANK;
    } else if (S == 0) { // :894-897 (and :112-118)
        it.kind = ITEM_FILL_BLANK;
        mode = BFA_MODE_EMPTY;
    } else if (p.simple) { // forced_alignment.py:963-976, lengths are 0-dim int64 tensors -> float32 compares
        int stride = 4;
        if ((float)(stride * S + 1) > (float)Traw * 0.9f) stride = 3;
        if ((float)(stride * S + 1) > (float)Traw * 0.8f) stride = 2;
        it.stride = stride;
        it.L = stride * S + 1;
        it.bw = band_standard(it.L);
        mode = BFA_MODE_STANDARD;
        if (T < 1) { status = BFA_ITEM_TOO_SHORT; it.kind = ITEM_FILL_BLANK; }
        else it.kind = ITEM_DP;
    } else {
        const bool seg_candidate = (p.anchors > 0 && p.sil >= 0 && has_sil);
        // standard mode (also the fallback of the segmented attempt)
        int stride = 4; // :153-157
        if (stride * S + 1 > T) stride = 3;
        if (stride * S + 1 > T) stride = 2;
        if (stride * S + 1 > T) stride = 1;
        const int L = stride * S + 1;
        it.stride = stride;
        it.L = L;
        if (L > T) {
            if (T < S) { status = BFA_ITEM_TOO_SHORT; it.kind = ITEM_FILL_BLANK; } // :161-165
            else { it.kind = ITEM_FILL_PROP; mode = BFA_MODE_PROPORTIONAL; }      // :166-176
        } else {
            it.kind = ITEM_DP;
            it.bw = band_standard(L);
            mode = BFA_MODE_STANDARD;
        }
        if (seg_candidate) mode = -1 - mode; // k_plan_segmented decides (it may keep this fallback)
    }
    if (it.kind == ITEM_DP && r_class_for_L(it.L) == 0) {
        status = BFA_ITEM_TOO_LARGE; // TODO(big-L workgroup kernel)
        it.kind = ITEM_FILL_BLANK;
    }
    a.items[b] = it;
    a.status[b] = status;
    a.umode[b] = mode;
    a.seg_count[b] = 0;
}

// =================================================================================================
// softmax of four posterior rows at once: lane = 16*g + j handles row g, columns j, 16+j, 32+j, ...
// =================================================================================================
struct RowLane {
    uint32_t valid; // bit k : column 16k+j exists
    uint32_t tmask; // bit k : column 16k+j is a boosted / floored target
    int blank_k;    // k such that column 16k+j is the blank column, or -1
};

template <int NK>
__device__ __forceinline__ void softmax16(float (&x)[NK], uint32_t valid)
{
    float mx = x[0];
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
        else if (valid & (1u << k)) acc = acc + e;
    }
    const float ls = logf_u10(row16_butterfly_add(acc));
#pragma unroll
    for (int k = 0; k < NK; ++k) x[k] = x[k] - ls;
}

// forced_alignment.py:29-83 (+ :543-561 when anchor_cnt > 0) applied to the register-resident quad
template <int NK>
__device__ __forceinline__ void prepare_quad(float (&x)[NK], const RowLane &rl, bool boost, bool enforce,
                                             bool anchored, int anchor_cnt)
{
    if (boost) {
#pragma unroll
        for (int k = 0; k < NK; ++k)
            if (rl.tmask & (1u << k)) x[k] = x[k] + 5.0f;
        softmax16<NK>(x, rl.valid);
    }
    if (enforce) {
#pragma unroll
        for (int k = 0; k < NK; ++k)
            if ((rl.tmask & (1u << k)) && x[k] < MIN_LOGP) x[k] = MIN_LOGP;
    }
    if (anchored) {
        // silence anchoring of this row (segmented mode only): counts differ between the four rows of
        // the quad, so run the maximum count on the whole wave and keep only the wanted iterations
        int maxcnt = anchor_cnt;
        maxcnt = max(maxcnt, __shfl_xor(maxcnt, 16));
        maxcnt = max(maxcnt, __shfl_xor(maxcnt, 32));
        for (int i = 0; i < maxcnt; ++i) {
            float y[NK];
#pragma unroll
            for (int k = 0; k < NK; ++k) y[k] = (k == rl.blank_k) ? x[k] + 5.0f : x[k];
            softmax16<NK>(y, rl.valid);
            if (i < anchor_cnt) {
#pragma unroll
                for (int k = 0; k < NK; ++k) x[k] = y[k];
            }
        }
    }
}

// =================================================================================================
// K1 : banded CTC Viterbi forward, one DP per wavefront
// =================================================================================================
__device__ __forceinline__ int path_col(int s, int stride, int nt, const int32_t *tok, int blank)
{
    if (s < 1) return blank;
    const int q = (s - 1) / stride;
    if ((s - 1) - q * stride != 0 || q >= nt) return blank;
    return tok[q];
}

template <int R, int NK>
__device__ __forceinline__ int dp_item(const AlignArgs &a, const Item &it, float *sm)
{
    constexpr int LDW = 16 * NK;   // floats per staged row
    constexpr int W = (R + 3) / 4; // backpointer dwords per lane per 4 frames
    const int lane = threadIdx.x & 63;
    const int g = lane >> 4, j = lane & 15;
    const DevParams &p = a.p;
    const int b = it.utt;
    const int Ts = it.Ts, L = it.L, nt = it.nt, stride = it.stride;
    const float *lp = a.logp + (int64_t)b * a.strideB + (int64_t)it.row0 * a.strideT;
    const int32_t *tok = a.tokens + (int64_t)b * a.Smax + it.tok0;
    const bool anchored = it.anchored != 0;
    const uint8_t *anch = a.anchor + (int64_t)b * a.Tmax + it.row0;
    const bool boost = p.boost && !p.simple, enforce = p.enforce && !p.simple;

    // ---- per-lane constants of the softmax role
    RowLane rl;
    rl.valid = 0; rl.tmask = 0; rl.blank_k = -1;
    {
        const uint32_t *um = a.umask + (int64_t)b * MASK_WORDS;
#pragma unroll
        for (int k = 0; k < NK; ++k) {
            const int c = 16 * k + j;
            if (c < a.C) {
                rl.valid |= 1u << k;
                if ((um[c >> 5] >> (c & 31)) & 1u) rl.tmask |= 1u << k;
                if (c == p.blank) rl.blank_k = k;
            }
        }
    }
    // ---- per-lane constants of the DP role: states s = lane*R + r
    int col[R];
    uint32_t skipm = 0; // bit r : can_skip[s]  (forced_alignment.py:603-605)
    float fs[R];
#pragma unroll
    for (int r = 0; r < R; ++r) {
        const int s = lane * R + r;
        col[r] = (s < L) ? path_col(s, stride, nt, tok, p.blank) : p.blank;
        const int c2 = path_col(s - 2, stride, nt, tok, p.blank);
        if (s >= 2 && s < L && col[r] != c2) skipm |= 1u << r;
        fs[r] = (float)s;
    }
    float dp[R];
#pragma unroll
    for (int r = 0; r < R; ++r) dp[r] = NEG; // :582

    const bool use_band = (it.bw > 0 && Ts > 1 && L > 1); // :586
    const double pace = use_band ? (double)(L - 1) / (double)(Ts - 1) : 0.0;
    const float pace32 = use_band ? (float)(L - 1) / (float)(Ts - 1) : 0.0f;
    const double bwd = (double)it.bw;
    const float bwf = (float)it.bw;

    uint32_t *bp = a.bp + it.bp_off;
    const int nq = (Ts + 3) >> 2;

    float x[NK], xn[NK];
    auto load_quad = [&](float(&dst)[NK], int q) {
        int row = 4 * q + g;
        if (row > Ts - 1) row = Ts - 1;
        const float *rp = lp + (int64_t)row * a.strideT;
#pragma unroll
        for (int k = 0; k < NK; ++k) {
            const int c = 16 * k + j;
            dst[k] = (c < a.C) ? rp[c] : 0.0f;
        }
    };
    load_quad(x, 0);
#pragma unroll
    for (int k = 0; k < NK; ++k) xn[k] = x[k];

    for (int q = 0; q < nq; ++q) {
        if (q + 1 < nq) load_quad(xn, q + 1); // prefetch while this quad is normalised and consumed
        int acnt = 0;
        if (anchored) {
            int row = 4 * q + g;
            if (row > Ts - 1) row = Ts - 1;
            acnt = anch[row];
        }
        prepare_quad<NK>(x, rl, boost, enforce, anchored, acnt);
#pragma unroll
        for (int k = 0; k < NK; ++k) sm[g * LDW + 16 * k + j] = x[k];
        wave_lds_sync();

        uint32_t word[W];
#pragma unroll
        for (int w = 0; w < W; ++w) word[w] = 0u;
#pragma unroll
        for (int f = 0; f < 4; ++f) {
            const int t = 4 * q + f;
            if (t < Ts) {
                float e[R];
#pragma unroll
                for (int r = 0; r < R; ++r) e[r] = sm[f * LDW + col[r]];
                if (t == 0) { // :594-596
#pragma unroll
                    for (int r = 0; r < R; ++r) {
                        const int s = lane * R + r;
                        if (s == 0) dp[r] = e[r];
                        if (s == 1 && L > 1) dp[r] = e[r];
                    }
                } else {
                    // values of the two states left of this lane's first state
                    const float l1 = dpp_mov<DPP_WAVE_SHR1>(NEG, dp[R - 1]);
                    float l2;
                    if constexpr (R >= 2) l2 = dpp_mov<DPP_WAVE_SHR1>(NEG, dp[R >= 2 ? R - 2 : 0]);
                    else l2 = dpp_mov<DPP_WAVE_SHR1>(NEG, l1);
                    float lo = 0.0f, hi = 0.0f;
                    if (use_band) { // :650-653
                        if (p.simple) { const float c = (float)t * pace32; lo = c - bwf; hi = c + bwf; }
                        else { const double c = (double)t * pace; lo = (float)(c - bwd); hi = (float)(c + bwd); }
                    }
                    float nd[R];
#pragma unroll
                    for (int r = 0; r < R; ++r) {
                        const float p1 = (r >= 1) ? dp[(r >= 1) ? r - 1 : 0] : l1;
                        const float p2 = (r >= 2) ? dp[(r >= 2) ? r - 2 : 0] : ((r == 1) ? l1 : l2);
                        const float c0 = dp[r] + e[r];
                        float c1 = p1 + e[r];
                        float c2 = p2 + e[r];
                        if (r == 0 && lane == 0) c1 = NEG;  // s == 0 has no advance  (:616-617)
                        if (!(skipm & (1u << r))) c2 = NEG; // (:620-625, :642)
                        const float best = __builtin_fmaxf(__builtin_fmaxf(c0, c1), c2);
                        const uint32_t k = (c0 == best) ? 0u : ((c1 == best) ? 1u : 2u); // first maximum (:645)
                        word[r >> 2] |= k << (8 * f + 2 * (r & 3));
                        float v = best;
                        if (use_band && (fs[r] < lo || fs[r] > hi)) v = NEG;
                        nd[r] = v;
                    }
#pragma unroll
                    for (int r = 0; r < R; ++r) dp[r] = nd[r];
                }
            }
        }
#pragma unroll
        for (int w = 0; w < W; ++w) bp[((int64_t)q * W + w) * 64 + lane] = word[w];
        wave_lds_sync(); // every lane is done reading sm before the next quad overwrites it
#pragma unroll
        for (int k = 0; k < NK; ++k) x[k] = xn[k];
    }

    // ---- final state (forced_alignment.py:656-682)
    int f;
    {
        int rm = -1;                  // rightmost state with dp > NEG
        float bv = 0.0f; int bi = -1; // best among dp > NEG (first maximum)
        float av = 0.0f; int ai = -1; // best among all (first maximum)
        float vL1 = NEG, vL2 = NEG;   // dp[L-1], dp[L-2]  (only "<= NEG" is tested, see below)
#pragma unroll
        for (int r = 0; r < R; ++r) {
            const int s = lane * R + r;
            if (s < L) {
                if (dp[r] > NEG) { rm = s; if (bi < 0 || dp[r] > bv) { bv = dp[r]; bi = s; } }
                if (ai < 0 || dp[r] > av) { av = dp[r]; ai = s; }
                if (s == L - 1) vL1 = dp[r];
                if (s == L - 2) vL2 = dp[r];
            }
        }
#pragma unroll
        for (int off = 32; off >= 1; off >>= 1) {
            rm = max(rm, __shfl_xor(rm, off));
            // a value below NEG is replaced by NEG here; both are "<= NEG", the only test made
            vL1 = __builtin_fmaxf(vL1, __shfl_xor(vL1, off));
            vL2 = __builtin_fmaxf(vL2, __shfl_xor(vL2, off));
            const float obv = __shfl_xor(bv, off); const int obi = __shfl_xor(bi, off);
            if (obi >= 0 && (bi < 0 || obv > bv || (obv == bv && obi < bi))) { bv = obv; bi = obi; }
            const float oav = __shfl_xor(av, off); const int oai = __shfl_xor(ai, off);
            if (oai >= 0 && (ai < 0 || oav > av || (oav == av && oai < ai))) { av = oav; ai = oai; }
        }
        if (!p.truly_forced) {
            f = (bi >= 0) ? bi : ai;
        } else {
            f = L - 1;
            float v = vL1;
            if (v <= NEG && L >= 2) { f = L - 2; v = vL2; }
            if (v <= NEG) f = (rm >= 0) ? rm : (L - 1);
        }
    }
    return f;
}

// One kernel per (R, NK): the register budget of a wave is set by its own R, so short CTC paths keep
// the occupancy that hides HBM latency.  Every launch walks the whole item list and takes the items
// of its class; the host launches only the classes the shapes allow.
template <int R, int NK>
__global__ __launch_bounds__(64) void k_dp(AlignArgs a)
{
    __shared__ float sm[4 * 16 * NK];
    const int n_items = a.counters[0];
    for (int i = blockIdx.x; i < n_items; i += gridDim.x) {
        const Item it = a.items[i];
        if (it.kind != ITEM_DP || r_class_for_L(it.L) != R) continue;
        const int f = dp_item<R, NK>(a, it, sm);
        if ((threadIdx.x & 63) == 0) a.items[i].final_state = f;
    }
}

// =================================================================================================
// K2 : backtrace + frame outputs.  One wavefront per item.
// =================================================================================================
__global__ __launch_bounds__(64) void k_backtrace(AlignArgs a)
{
    constexpr int CH = 64; // frames per LDS chunk (16 quads)
    __shared__ uint32_t sbp[16 * 4 * 64]; // up to W=4 dwords per lane per quad
    __shared__ int sst[CH];
    const int lane = threadIdx.x & 63;
    const DevParams &p = a.p;
    const int n_items = a.counters[0];
    for (int i = blockIdx.x; i < n_items; i += gridDim.x) {
        const Item it = a.items[i];
        const int b = it.utt;
        int32_t *oph = a.frame_ph + (int64_t)b * a.Tmax;
        int32_t *oid = a.frame_idx + (int64_t)b * a.Tmax;
        const int32_t *tok = a.tokens + (int64_t)b * a.Smax + it.tok0;
        if (it.kind == ITEM_FILL_BLANK) {
            for (int t = lane; t < it.nout; t += 64) { oph[it.out0 + t] = p.blank; oid[it.out0 + t] = -1; }
            continue;
        }
        if (it.kind == ITEM_FILL_PROP) { // forced_alignment.py:170-172
            for (int t = lane; t < it.nout; t += 64) {
                const int fi = (int)(((int64_t)t * it.nt) / it.nout);
                oph[it.out0 + t] = tok[fi]; oid[it.out0 + t] = it.tok0 + fi;
            }
            continue;
        }
        if (it.kind == ITEM_FILL_SIL) { // forced_alignment.py:382-397
            const double fps = (it.nt > 0) ? (double)it.nout / (double)it.nt : 0.0;
            for (int t = lane; t < it.nout; t += 64) {
                int id = -1;
                if (it.nt > 0) { // the k with int(k*fps) <= t < int((k+1)*fps); the ranges are disjoint
                    const int k = (int)((double)t / fps);
                    for (int kk = max(0, k - 1); kk <= min(it.nt - 1, k + 1); ++kk) {
                        const int f0 = (int)((double)kk * fps), f1 = (int)((double)(kk + 1) * fps);
                        if (t >= f0 && t < f1) id = it.tok0 + kk;
                    }
                }
                oph[it.out0 + t] = p.sil; oid[it.out0 + t] = id;
            }
            continue;
        }
        if (it.kind != ITEM_DP) continue;

        const int R = r_class_for_L(it.L);
        const int W = bp_words_for_R(R);
        const int Ts = it.Ts, L = it.L;
        const uint32_t *bp = a.bp + it.bp_off;
        int s = it.final_state;
        int sl = s / R, sr = s - sl * R; // (lane, slot) of the current state
        const int nchunks = (Ts + CH - 1) / CH;
        for (int c = nchunks - 1; c >= 0; --c) {
            const int t0 = c * CH;
            const int t1 = min(Ts, t0 + CH);
            const int q0 = t0 >> 2, q1 = (t1 + 3) >> 2;
            const int ndw = (q1 - q0) * W * 64;
            wave_lds_sync();
            for (int d = lane; d < ndw; d += 64) sbp[d] = bp[(int64_t)q0 * W * 64 + d];
            wave_lds_sync();
            if (lane == 0) {
                for (int t = t1 - 1; t >= t0; --t) {
                    sst[t - t0] = s;
                    if (t > 0) { // state[t-1] = bp[t][state[t]]  (:691-692)
                        const uint32_t wd = sbp[(((t >> 2) - q0) * W + (sr >> 2)) * 64 + sl];
                        const int k = (int)((wd >> (8 * (t & 3) + 2 * (sr & 3))) & 3u);
                        s -= k; sr -= k;
                        while (sr < 0) { sr += R; sl -= 1; } // k = 2 crosses two lanes when R == 1
                        if (s < 0) { s += L; sl = s / R; sr = s - sl * R; } // python negative-index wrap
                    }
                }
            }
            wave_lds_sync();
            // s/sl/sr were advanced by lane 0 only: broadcast for the next chunk
            s = __builtin_amdgcn_readfirstlane(s);
            sl = __builtin_amdgcn_readfirstlane(sl);
            sr = __builtin_amdgcn_readfirstlane(sr);
            const int t = t0 + lane;
            if (t < t1) {
                const int o = t - it.pad_left; // :447-448 trim the boundary padding
                if (o >= 0 && o < it.nout) {
                    const int st = sst[lane];
                    int ph = p.blank, id = -1;
                    if (st >= 1) {
                        const int q = (st - 1) / it.stride;
                        if ((st - 1) - q * it.stride == 0 && q < it.nt) { ph = tok[q]; id = it.tok0 + q; }
                    }
                    oph[it.out0 + o] = ph;
                    oid[it.out0 + o] = id;
                }
            }
        }
    }
}

// =================================================================================================
// K3a : assort_frames (forced_alignment.py:777-834), one wavefront per utterance
// =================================================================================================
__global__ __launch_bounds__(64) void k_assort(AlignArgs a)
{
    const int lane = threadIdx.x & 63;
    const DevParams &p = a.p;
    for (int b = blockIdx.x; b < a.B; b += gridDim.x) {
        int32_t *ph = a.frame_ph + (int64_t)b * a.Tmax;
        int32_t *ix = a.frame_idx + (int64_t)b * a.Tmax;
        const int T = a.uT[b];
        const int st = a.status[b];
        const bool none = (st != BFA_ITEM_OK) || a.uS[b] == 0;
        const int Tr = none ? 0 : T; // frames that take part in the run-length encoding
        for (int t = Tr + lane; t < a.Tmax; t += 64) { ph[t] = p.blank; ix[t] = -1; }
        bfa_segment *out = a.segs + (int64_t)b * a.seg_cap;
        int count = 0;
        int run_start = 0, run_ph = 0, run_ix = 0; // the open run (wave-uniform)
        for (int base = 0; base < Tr; base += 64) {
            const int t = base + lane;
            const bool in = t < Tr;
            const int cph = in ? ph[t] : 0, cix = in ? ix[t] : 0;
            int pph = __shfl_up(cph, 1), pix = __shfl_up(cix, 1);
            if (lane == 0) { pph = run_ph; pix = run_ix; }
            const bool is_start = in && (t == 0 || cph != pph || cix != pix); // :798-801
            const unsigned long long m = __ballot(is_start);
            // a start at t>0 closes the run that began at the previous start (shuffles stay convergent)
            const unsigned long long below = m & ((1ull << lane) - 1ull);
            const int src = below ? (63 - __builtin_clzll(below)) : 0;
            const int sph = __shfl(cph, src), six = __shfl(cix, src);
            const int ps = below ? (base + src) : run_start;
            const int pp = below ? sph : run_ph;
            const int pi = below ? six : run_ix;
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
            const int lph = __shfl(cph, last), lix = __shfl(cix, last);
            if (m) { run_start = base + last; run_ph = lph; run_ix = lix; }
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
}

// =================================================================================================
// K3b : _calculate_confidences (utils.py:70-113), one wavefront per utterance, one lane per tuple
// =================================================================================================
constexpr int CONF_SERIAL_CAP = 2048;

__global__ __launch_bounds__(64) void k_conf(ConfArgs a)
{
    __shared__ float mval[CONF_SERIAL_CAP];
    __shared__ uint8_t mflag[CONF_SERIAL_CAP];
    const int lane = threadIdx.x & 63;
    for (int b = blockIdx.x; b < a.B; b += gridDim.x) {
        const float *lp = a.logp + (int64_t)b * a.strideB;
        const bfa_segment *sg = a.segs + (int64_t)b * a.seg_cap;
        float *cf = a.conf + (int64_t)b * a.seg_cap;
        int T = a.T_rows ? a.T_rows[b] : a.Tmax;
        if (T > a.Tmax) T = a.Tmax;
        int n = a.seg_count[b];
        if (n > a.seg_cap) n = a.seg_cap;
        int bad = 0;
        // ---- does any tuple read a cell that an earlier tuple has written through its 0-dim view?
        int alias = 0;
        for (int i = lane; i < n; i += 64) {
            const int ph = sg[i].phoneme;
            const int s = max(0, sg[i].start), e = min(T, sg[i].end);
            for (int k = 0; k < i; ++k) {
                if (sg[k].phoneme != ph) continue;
                const int sk = max(0, sg[k].start);
                if (sk == s || (sk >= s && sk < e)) { alias = 1; break; }
            }
        }
        alias = __any(alias);
        if (!alias) {
            for (int i = lane; i < n; i += 64) {
                const int ph = sg[i].phoneme;
                const int s = max(0, sg[i].start), e = min(T, sg[i].end); // :86-87
                if (s >= T || ph < 0 || ph >= a.C) { bad = 1; cf[i] = 0.0f; continue; } // IndexError at :89
                float c = expf_u10(lp[(int64_t)s * a.strideT + ph]);
                if (s < e) {
                    const float half = c / 2.0f; // :95 (a fresh tensor: stays constant)
                    int good = 1;
                    float mx = 0.0f;
                    for (int f = s + 1; f < e; ++f) {
                        const float v = expf_u10(lp[(int64_t)f * a.strideT + ph]);
                        mx = (f == s + 1) ? v : __builtin_fmaxf(mx, v);
                        if (v > half || v > 0.1f) { c = c + v; good++; } // :101-103
                    }
                    if (good > 1) {
                        c = c / (float)good; // :105 -- this also lands in probs[start, ph] ...
                        const float m2 = __builtin_fmaxf(c, mx); // ... so :107 sees the mean at `start`
                        if (c < m2 / 2.0f) c = m2;
                    }
                }
                cf[i] = c;
            }
        } else if (n <= CONF_SERIAL_CAP) {
            // exact serial replay with the write-through cells (rare: overlapping tuples of one phoneme)
            for (int i = lane; i < n; i += 64) mflag[i] = 0;
            wave_lds_sync();
            if (lane == 0) {
                auto prob = [&](int f, int ph, int upto) -> float {
                    for (int k = upto; k >= 0; --k)
                        if (mflag[k] && sg[k].phoneme == ph && max(0, sg[k].start) == f) return mval[k];
                    return expf_u10(lp[(int64_t)f * a.strideT + ph]);
                };
                for (int i = 0; i < n; ++i) {
                    const int ph = sg[i].phoneme;
                    const int s = max(0, sg[i].start), e = min(T, sg[i].end);
                    if (s >= T || ph < 0 || ph >= a.C) { bad = 1; cf[i] = 0.0f; continue; }
                    float c = prob(s, ph, i - 1);
                    if (s < e) {
                        const float half = c / 2.0f;
                        int good = 1;
                        for (int f = s + 1; f < e; ++f) {
                            const float v = prob(f, ph, i);
                            if (v > half || v > 0.1f) { c = c + v; good++; mval[i] = c; mflag[i] = 1; }
                        }
                        if (good > 1) {
                            c = c / (float)good; mval[i] = c; mflag[i] = 1;
                            float mx = prob(s, ph, i);
                            for (int f = s + 1; f < e; ++f) mx = __builtin_fmaxf(mx, prob(f, ph, i));
                            if (c < mx / 2.0f) c = mx;
                        }
                    }
                    cf[i] = c;
                }
            }
            wave_lds_sync();
        } else {
            bad = 1;
        }
        bad = __any(bad);
        if (lane == 0 && a.status) a.status[b] = bad ? BFA_ITEM_BAD_TOKEN : BFA_ITEM_OK;
    }
}

// =================================================================================================
// F.log_softmax(dim=-1) of [rows, C] (core.py:898-899), four rows per wavefront
// =================================================================================================
template <int NK>
__global__ __launch_bounds__(256) void k_log_softmax(const float *in, int64_t ld_in, float *out, int64_t ld_out,
                                                      int64_t rows, int C)
{
    const int lane = threadIdx.x & 63, g = lane >> 4, j = lane & 15;
    const int64_t wave = (int64_t)blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
    const int64_t nwaves = (int64_t)gridDim.x * (blockDim.x >> 6);
    uint32_t valid = 0;
#pragma unroll
    for (int k = 0; k < NK; ++k)
        if (16 * k + j < C) valid |= 1u << k;
    for (int64_t r0 = wave * 4; r0 < rows; r0 += nwaves * 4) {
        int64_t row = r0 + g;
        const bool live = row < rows;
        if (!live) row = rows - 1;
        float x[NK];
#pragma unroll
        for (int k = 0; k < NK; ++k) x[k] = (valid & (1u << k)) ? in[row * ld_in + 16 * k + j] : 0.0f;
        softmax16<NK>(x, valid);
        if (live) {
#pragma unroll
            for (int k = 0; k < NK; ++k)
                if (valid & (1u << k)) out[row * ld_out + 16 * k + j] = x[k];
        }
    }
}

} // namespace bfa

// -------------------------------------------------------------------------------------------------
// launchers used by bfa_capi.cpp
// -------------------------------------------------------------------------------------------------
extern "C" int bfa_launch_align(const bfa::AlignArgs *args, int dp_grid, void *stream_, void *ev0, void *ev1)
{
    using namespace bfa;
    hipStream_t stream = (hipStream_t)stream_;
    const AlignArgs &a = *args;
    const int nk = (a.C + 15) / 16;
    hipLaunchKernelGGL(k_plan, dim3((a.B + 127) / 128), dim3(128), 0, stream, a);
    if (ev0) (void)hipEventRecord((hipEvent_t)ev0, stream);
    // classes of CTC path length that can occur: L <= 4*Smax+1 (and L <= 1.2*Tmax in segmented mode)
    const int rmax = r_class_for_L(4 * a.Smax + 1) ? r_class_for_L(4 * a.Smax + 1) : MAX_R;
#define BFA_LAUNCH_DP(R_)                                                                                          \
    if ((R_) <= rmax) {                                                                                           \
        if (nk <= 2) hipLaunchKernelGGL((k_dp<R_, 2>), dim3(dp_grid), dim3(64), 0, stream, a);                    \
        else if (nk <= 5) hipLaunchKernelGGL((k_dp<R_, 5>), dim3(dp_grid), dim3(64), 0, stream, a);               \
        else hipLaunchKernelGGL((k_dp<R_, 8>), dim3(dp_grid), dim3(64), 0, stream, a);                            \
    }
    BFA_LAUNCH_DP(1) BFA_LAUNCH_DP(2) BFA_LAUNCH_DP(3) BFA_LAUNCH_DP(4)
    BFA_LAUNCH_DP(6) BFA_LAUNCH_DP(8) BFA_LAUNCH_DP(12) BFA_LAUNCH_DP(16)
#undef BFA_LAUNCH_DP
    if (ev1) (void)hipEventRecord((hipEvent_t)ev1, stream);
    hipLaunchKernelGGL(k_backtrace, dim3(dp_grid), dim3(64), 0, stream, a);
    hipLaunchKernelGGL(k_assort, dim3(a.B < 65536 ? a.B : 65536), dim3(64), 0, stream, a);
    return (int)hipGetLastError();
}

extern "C" int bfa_launch_conf(const bfa::ConfArgs *args, void *stream_)
{
    using namespace bfa;
    hipStream_t stream = (hipStream_t)stream_;
    hipLaunchKernelGGL(k_conf, dim3(args->B < 65536 ? args->B : 65536), dim3(64), 0, stream, *args);
    return (int)hipGetLastError();
}

extern "C" int bfa_launch_log_softmax(const float *in, int64_t ld_in, float *out, int64_t ld_out, int64_t rows,
                                      int C, void *stream_)
{
    using namespace bfa;
    hipStream_t stream = (hipStream_t)stream_;
    const int nk = (C + 15) / 16;
    int64_t blocks = (rows + 15) / 16;
    if (blocks > 8192) blocks = 8192;
    if (blocks < 1) blocks = 1;
    if (nk <= 2) hipLaunchKernelGGL(k_log_softmax<2>, dim3((unsigned)blocks), dim3(256), 0, stream, in, ld_in, out, ld_out, rows, C);
    else if (nk <= 5) hipLaunchKernelGGL(k_log_softmax<5>, dim3((unsigned)blocks), dim3(256), 0, stream, in, ld_in, out, ld_out, rows, C);
    else hipLaunchKernelGGL(k_log_softmax<8>, dim3((unsigned)blocks), dim3(256), 0, stream, in, ld_in, out, ld_out, rows, C);
    return (int)hipGetLastError();
}
