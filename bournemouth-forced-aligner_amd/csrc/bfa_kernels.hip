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
#include <cstdlib>

#include "bfa_assort.hpp"
#include "bfa_plan.inc"
#include "bfa_softmax.hpp"

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
// k_plan : sixteen lanes per utterance (bfa_plan.inc)
// =================================================================================================
__global__ __launch_bounds__(256) void k_plan(AlignArgs a)
{
    const int gid = blockIdx.x * blockDim.x + threadIdx.x;
    const int b = gid >> 4, sub = gid & 15;
    // One item slot per utterance; the segmented planner appends more.  The counters are zeroed HERE: nothing in this kernel
    // touches them (round 6: the candidates of the silence-anchored mode are marked in ucand[], not appended to a list through
    // an atomic counter), the first atomics on them belong to later kernels -- so no call needs a memset dispatch ahead of its
    // first kernel (until now the silence-anchored mode did: one 64-byte fill per head on a critical path of ~10 short launches).
    if (gid == 0) {
        a.counters[0] = a.B;
        for (int k = 1; k < 16; ++k) a.counters[k] = 0;
    }
    if (b >= a.B) return; // whole 16-lane groups
    plan_utterance(a, b, sub);
}

// =================================================================================================
// k_order : the utterance slots of a mixed-length call by decreasing cost, for k_mix: workgroup w of that kernel takes
// order[w], so the dispatcher starts the longest chains first and fills in with the short ones as workgroups retire.
// One workgroup: counting sort over the 256 cost buckets k_plan left in mix_key (the order inside a bucket does not matter).
// =================================================================================================
__global__ __launch_bounds__(1024) void k_order(AlignArgs a)
{
    __shared__ int cnt[256];
    const int tid = threadIdx.x;
    if (tid < 256) cnt[tid] = 0;
    __syncthreads();
    for (int b = tid; b < a.B; b += 1024) atomicAdd(&cnt[a.mix_key[b]], 1);
    __syncthreads();
    if (tid < 64) { // exclusive prefix sum over the buckets: four per lane, then a wave scan
        const int c0 = cnt[4 * tid], c1 = cnt[4 * tid + 1], c2 = cnt[4 * tid + 2], c3 = cnt[4 * tid + 3];
        const int mine = c0 + c1 + c2 + c3;
        int inc = mine;
#pragma unroll
        for (int off = 1; off < 64; off <<= 1) { const int o = __shfl_up(inc, off); if (tid >= off) inc += o; }
        const int base = inc - mine;
        cnt[4 * tid] = base; cnt[4 * tid + 1] = base + c0; cnt[4 * tid + 2] = base + c0 + c1; cnt[4 * tid + 3] = base + c0 + c1 + c2;
    }
    __syncthreads();
    for (int b = tid; b < a.B; b += 1024) a.mix_order[atomicAdd(&cnt[a.mix_key[b]], 1)] = b;
}

// =================================================================================================
// K3a : assort_frames (forced_alignment.py:777-834), one wavefront per utterance
// =================================================================================================
__global__ __launch_bounds__(64) void k_assort(AlignArgs a)
{
    const int lane = threadIdx.x & 63;
    for (int b = blockIdx.x; b < a.B; b += gridDim.x) assort_utterance<8>(a, b, lane);
}

// =================================================================================================
// K3b : _calculate_confidences (utils.py:70-113), one wavefront per utterance, one lane per tuple
// =================================================================================================
constexpr int CONF_SERIAL_CAP = 2048;

template <bool RAW>
__global__ __launch_bounds__(64) void k_conf(ConfArgs a)
{
    __shared__ float mval[CONF_SERIAL_CAP];
    __shared__ uint8_t mflag[CONF_SERIAL_CAP];
    const int lane = threadIdx.x & 63;
    for (int b = blockIdx.x; b < a.B; b += gridDim.x) {
        const LpView<RAW> lp{a.logp + (int64_t)b * a.strideB, a.strideT,
                             RAW ? a.row_stats + 2 * (int64_t)b * a.Tmax : nullptr, a.C};
        const bfa_segment *sg = a.segs + (int64_t)b * a.seg_cap;
        float *cf = a.conf + (int64_t)b * a.seg_cap;
        int T = a.T_rows ? a.T_rows[b] : a.Tmax;
        if (T > a.Tmax) T = a.Tmax;
        int n = a.seg_count[b];
        if (n > a.seg_cap) n = a.seg_cap;
        for (int i = (n < 0 ? 0 : n) + lane; i < a.seg_cap; i += 64) cf[i] = 0.0f; // the caller's buffer needs no fill ahead of the call
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
                float c = exp_cr(lp.at(s, ph));
                if (s < e) {
                    const float half = c / 2.0f; // :95 (a fresh tensor: stays constant)
                    int good = 1;
                    float mx = 0.0f;
                    // the lane's frames are one element per 268-byte row: eight independent (clamped) loads in
                    // flight per lane, or the longest tuple of the wave pays one full memory latency per frame
                    constexpr int U = 8;
                    for (int f0 = s + 1; f0 < e; f0 += U) {
                        float x[U];
                        lp.template at_n<U>(f0, e - 1, ph, x);
#pragma unroll
                        for (int u = 0; u < U; ++u) {
                            if (f0 + u < e) {
                                const float v = exp_cr(x[u]);
                                mx = (f0 + u == s + 1) ? v : __builtin_fmaxf(mx, v);
                                if (v > half || v > 0.1f) { c = c + v; good++; } // :101-103
                            }
                        }
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
                    return exp_cr(lp.at(f, ph));
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
        softmax16<NK>(x, valid, nullptr, nullptr, C < 16 ? C : 0);
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
extern "C" void bfa_launch_dp_nk2(const bfa::AlignArgs *args, unsigned class_mask, int mode, int grid, bfa::LaunchFan *fan);
extern "C" void bfa_launch_dp_redo_nk2(const bfa::AlignArgs *args, unsigned class_mask, int mode, hipStream_t stream);
extern "C" void bfa_launch_dp_nk5(const bfa::AlignArgs *args, unsigned class_mask, int mode, int grid, bfa::LaunchFan *fan);
extern "C" void bfa_launch_dp_redo_nk5(const bfa::AlignArgs *args, unsigned class_mask, int mode, hipStream_t stream);
extern "C" void bfa_launch_dp_nk8(const bfa::AlignArgs *args, unsigned class_mask, int mode, int grid, bfa::LaunchFan *fan);
extern "C" void bfa_launch_dp_redo_nk8(const bfa::AlignArgs *args, unsigned class_mask, int mode, hipStream_t stream);
extern "C" void bfa_launch_dp_big_nk2(const bfa::AlignArgs *args, int grid, hipStream_t stream, int huge);
extern "C" void bfa_launch_dp_big_nk5(const bfa::AlignArgs *args, int grid, hipStream_t stream, int huge);
extern "C" void bfa_launch_dp_big_nk8(const bfa::AlignArgs *args, int grid, hipStream_t stream, int huge);
extern "C" void bfa_launch_backtrace(const bfa::AlignArgs *args, int grid, hipStream_t stream, int wide);
extern "C" void bfa_launch_backtrace_sel(const bfa::AlignArgs *args, int sel, int fused, int grid, hipStream_t stream, int wide);
extern "C" void bfa_launch_segment_plan(const bfa::AlignArgs *args, hipStream_t stream);
extern "C" void bfa_k1_one_nk2(const bfa::AlignArgs *a, int RW, hipStream_t s);
extern "C" void bfa_k1_one_nk5(const bfa::AlignArgs *a, int RW, hipStream_t s);
extern "C" void bfa_k1_mix_nk2(const bfa::AlignArgs *a, const int32_t *order, hipStream_t s);
extern "C" void bfa_k1_mix_nk5(const bfa::AlignArgs *a, const int32_t *order, hipStream_t s);

namespace bfa {

// Which kernels a call takes, from its shapes and the caller's hints alone (host arithmetic; bfa_call_path reports it):
// the per-class kernels side by side, the one-kernel mixed-length path (k_mix) or the one-kernel path of small single-class
// calls (k_one).
struct CallPlan {
    int mode;            // 0 reference-default flags without silence anchoring, 1 with, 2 other flags (no fused preparation)
    unsigned mask;       // class kernels that may find items: bits 0-6 full layouts, 8-15 fast windows, 20-27 exact windows
    unsigned wmask, wall, xmask;
    bool hinted, seg_possible, one_ok, use_mix, seg_mix;
    int rw1;
};

// (A/B: BFA_NO_WIDE_XWIN=1 keeps the full layout for the wide fallbacks of silence-anchored calls, as before round 6)
static bool seg_wide_xwin()
{
    static const bool on = [] { const char *e = std::getenv("BFA_NO_WIDE_XWIN"); return !(e && e[0] == '1'); }();
    return on;
}

static CallPlan plan_call(int B, int C, int Smax, const DevParams &p, bool frames, int wide_any_max = 512)
{
    // K1 classes that can occur: every CTC path has L <= 4*Smax+1; the caller may narrow this down
    const int Lmax = 4 * Smax + 1;
    unsigned mask = r_class_mask_upto(Lmax);
    const bool hinted = ((unsigned)p.class_mask & 0x0ff0ffffu) != 0u; // a class selection (flag bits alone are not one)
    if (hinted) mask &= (unsigned)p.class_mask & 0xffffu;
    const bool seg_possible = !p.simple && p.anchors > 0 && p.sil >= 0 && !(p.class_mask & BFA_HINT_NO_SILENCE_TARGETS);
    const int mode = (p.boost && p.enforce && !p.simple) ? (seg_possible ? 1 : 0) : 2;
    // sliding-window classes (16-rows-per-pass kernels only: reference-default flags, C = 67 or 17)
    unsigned wmask = 0, wall = 0;
    if (mode != 2 && (C == 67 || C == 17) && Lmax > 60) {
        wmask = 0xafu; // Rw in {1,2,3,4,6,8} (bit Rw-1); classes no utterance can use cost one empty launch each
        const int top = win_class_for(Lmax, (Lmax / 4 > 20) ? Lmax / 4 : 20, 1);
        for (int rw = 8; rw >= 1; --rw) // drop the classes above the one the longest possible path would take
            if (top > 0 && rw > top) wmask &= ~(1u << (rw - 1));
        wall = wmask;
        if (hinted) wmask &= (p.class_mask >> 8);
    }
    // (the window result is exact because emissions are <= 0; a floor above log(1) = 0 would break that argument)
    if (!(p.min_logp <= 0.0f)) { wmask = 0; wall = 0; }
    // Silence-anchored mode: no sliding window.  Only the few utterances whose segmented attempt fails would use it, and
    // the window consumers in the mode's merged K1 kernel cost every PIECE 30 spilled VGPRs (one 256-byte scratch store per
    // item and spilled register: 0.29 GB of DRAM writes per 4096-utterance launch; realtext 1.54 -> 1.49 ms per step).
    if (seg_possible) wmask = 0;
    mask |= wmask << 8;
    // Exact-window classes (bfa_dp3.inc: DpCoreW<.., EX>): banded standard-mode items the fast window is not tried on -- too
    // many frames or tokens for a result that only stands above the sentinel -- compute their in-band states only, too,
    // instead of all L; bit 8: the exact window also reruns the fast windows that end at the sentinel.  Same conditions as
    // the fast window (its exactness argument needs emissions <= 0 as well).
    unsigned xmask = 0;
    if (wall && mode == 0) xmask = (hinted ? (wall & ((unsigned)p.class_mask >> 20)) : wall) | XWIN_REDO;
    // Mixed-length calls (the caller has not promised uniform lengths): ONE kernel aligns and walks every utterance of the
    // narrow classes (bfa_dp4.inc: k_mix) -- the exact window for every stride >= 3 window item of the classes Rw <= 4 (a hinted
    // fast-window class stands for its exact twin), the full layout R <= 4 -- in longest-first order (k_order).
    // (a small call the hint puts in ONE fast-window class is k_one's, see below)
    const unsigned hw1 = (p.class_mask >> 8) & 0xffu;
    const int rw1 = (hw1 == 1u) ? 1 : (hw1 == 2u) ? 2 : (hw1 == 4u) ? 3 : 0;
    const bool one_ok = !seg_possible && mode == 0 && (C == 67 || C == 17) && rw1 > 0 && (p.class_mask & 0x7fu) == 0 &&
                        ((p.class_mask >> 20) & 0xffu) == 0 && ((wmask >> (rw1 > 0 ? rw1 - 1 : 0)) & 1u) && B <= ONE_MAX_BATCH && Lmax <= 256 &&
                        frames;
    const bool use_mix = mode == 0 && !seg_possible && (C == 67 || C == 17) && wall != 0 && B >= MIX_MIN_BATCH && !one_ok &&
                         !(p.class_mask & BFA_HINT_UNIFORM_LENGTHS) && frames;
    if (use_mix) {
        const unsigned narrow_x = hinted ? (wall & (((unsigned)p.class_mask >> 20) | ((unsigned)p.class_mask >> 8)) & 0xfu) : (wall & 0xfu);
        xmask |= narrow_x | XWIN_MIX;
    }
    // Silence-anchored mixed-length calls (round 6): an utterance whose anchoring FAILS is aligned in the standard mode
    // (forced_alignment.py:131-141) -- with soft posteriors that is every utterance of the call (no window average reaches 0.9),
    // and until now each took the full state layout of its class, one kernel per class (C5 proxy: 8.3 ms at peak 3 against 5.1
    // at peak 9, profiles/r06_c5proxy_p*.json).  k_plan marks the banded stride >= 3 fallbacks of the window classes Rw <= 4 as
    // exact-window items (the planner drops the slot when the anchoring succeeds) and k_mix aligns and walks them, longest
    // first, exactly as in a standard-mode mixed call.  The fast window stays out of this mode (wmask = 0).
    bool seg_mix = false;
    // (full batches only: a small call keeps its fallbacks in the one wide launch, k_dp5_any -- behind it on the caller's stream
    // k_mix doubled the DP phase of the reference's sixteen-utterance chunk, 1.02 -> 1.22 ms)
    if (mode == 1 && (C == 67 || C == 17) && Lmax > 60 && p.min_logp <= 0.0f && B > wide_any_max &&
        !(p.class_mask & BFA_HINT_UNIFORM_LENGTHS) && frames) {
        unsigned w4 = 0xfu;
        const int top = win_class_for(Lmax, (Lmax / 4 > 20) ? Lmax / 4 : 20, 1);
        for (int rw = 4; rw >= 1; --rw) if (top > 0 && rw > top) w4 &= ~(1u << (rw - 1));
        xmask = w4 | XWIN_MIX;
        seg_mix = w4 != 0u;
        if (!seg_mix) xmask = 0;
    }
    // ... and the fallbacks of the WIDE window classes Rw = 6 / 8 -- the longest utterances of a call: 30-s segments with 100-150
    // targets, whose full layout is the class R = 8 / 12 / 16 -- as exact-window items of their class kernels (k_dp4x<6 / 8>: the
    // in-band states only, half of the full layout's, and the closed-form dead tails): C5 proxy 4.65 -> 4.43 ms at peak 9, 5.10 ->
    // 4.40 at peak 3 (profiles/r06_wide_xwin_ab.txt).  Full batches only: in a small call the two class kernels are two more
    // streams, and the runtime put each on the OTHER head's hardware queue -- the reference's sixteen-utterance chunk 0.97 -> 1.69 ms
    // although the chain itself got shorter (k_dp4x<6> 341 + 50 us against k_dp5_any's 470; same file).  Only where the full-layout
    // classes such an utterance would take can occur at all (Rw 6: R >= 8, Rw 8: R >= 12; `mask` holds the hint).
    if (mode == 1 && (C == 67 || C == 17) && Lmax > 60 && p.min_logp <= 0.0f && frames && B > wide_any_max && seg_wide_xwin()) {
        unsigned w68 = 0;
        const int top = win_class_for(Lmax, (Lmax / 4 > 20) ? Lmax / 4 : 20, 1);
        if (top >= 6 && (mask & (16u | 32u | 64u))) w68 |= 0x20u;
        if (top >= 8 && (mask & (32u | 64u))) w68 |= 0x80u;
        xmask |= w68;
    }
    mask |= (xmask & 0xafu) << 20;
    CallPlan c;
    c.mode = mode; c.mask = mask; c.wmask = wmask; c.wall = wall; c.xmask = xmask; c.hinted = hinted;
    c.seg_possible = seg_possible; c.one_ok = one_ok && (xmask & 0xafu) == 0; c.use_mix = use_mix; c.rw1 = rw1; c.seg_mix = seg_mix;
    return c;
}

} // namespace bfa

// 0: per-class kernels, 1: k_mix (+ class kernels for what it has no body for), 2: k_one
extern "C" int bfa_call_path_impl(int B, int C, int Smax, const bfa::DevParams *p)
{
    const bfa::CallPlan c = bfa::plan_call(B, C, Smax, *p, true);
    return c.one_ok ? 2 : (c.use_mix ? 1 : 0);
}

extern "C" int bfa_launch_align(const bfa::AlignArgs *args, int dp_grid, void *stream_, void *ev0, void *ev1,
                                void **aux_streams, void **aux_events, void **fork_event, int (*ensure_aux)(void *), void *ctx,
                                int aux_first, int aux_count)
{
    using namespace bfa;
    hipStream_t stream = (hipStream_t)stream_;
    AlignArgs a = *args;
    const DevParams &p = a.p;
    const int nk = (a.C + 15) / 16;
    const CallPlan cp = plan_call(a.B, a.C, a.Smax, p, a.frame_ph && a.frame_idx, a.wide_any_max);
    const int Lmax = 4 * a.Smax + 1;
    unsigned mask = cp.mask;
    const unsigned wmask = cp.wmask, xmask = cp.xmask;
    const bool seg_possible = cp.seg_possible, use_mix = cp.use_mix;
    const int mode = cp.mode, rw1 = cp.rw1;
    a.p.win_mask = wmask;
    a.p.xwin_mask = xmask | ((xmask & XWIN_REDO) ? (p.xwin_mask & XWIN_ROUTE) : 0u); // (routing: the caller's bit, when reruns by the exact window exist)
    // One item per utterance (no silence-anchored pieces) on the 16-rows-per-pass kernels: K2 walks each full-layout
    // class right behind its K1 kernel on that kernel's stream, the window classes after the sentinel reruns, and
    // emits the run-length tuples during the walk -- no K3a launch.
    const bool fused_k2 = !seg_possible && mode == 0 && (a.C == 67 || a.C == 17);
    const bool per_class_k2 = fused_k2; // (measured for the silence-anchored mode as well: every extra K2 launch re-scans
                                        // the long item list of that mode, 1.11 -> 1.18 ms; it keeps the single walk after the join)
    a.k2_sel = K2_ALL;
    a.k2_fused_rle = fused_k2 ? 1 : 0;
    a.k2_per_class = per_class_k2 ? 1 : 0;
    a.xcd_contig = (mode == 0 && (p.class_mask & BFA_HINT_UNIFORM_LENGTHS)) ? 1 : 0;
    // Small batches of ONE sliding-window class (by the caller's hint): plan + window DP + rerun + walk in one kernel, one
    // workgroup per utterance (bfa_dp4.inc: k_one).  The serial chain of the DP is all that is left of the call.
    if (cp.one_ok) {
        a.p.xwin_mask = 0; // (k_one reruns its own window failures with the full layout)
        if (ev0) (void)hipEventRecord((hipEvent_t)ev0, stream);
        if (a.C == 67) bfa_k1_one_nk5(&a, rw1, stream); else bfa_k1_one_nk2(&a, rw1, stream);
        if (ev1) (void)hipEventRecord((hipEvent_t)ev1, stream);
        return (int)hipGetLastError();
    }
    a.mix_exact_only = cp.seg_mix ? 1 : 0;
    hipLaunchKernelGGL(k_plan, dim3((a.B + 15) / 16), dim3(256), 0, stream, a);
    if (seg_possible) bfa_launch_segment_plan(&a, stream);
    if (ev0) (void)hipEventRecord((hipEvent_t)ev0, stream);
    if (cp.seg_mix || use_mix) hipLaunchKernelGGL(k_order, dim3(1), dim3(1024), 0, stream, a);
    if (use_mix) {
        // the narrow classes are k_mix's: the class kernels below only see what it does not take
        mask &= ~(7u | (0xfu << 20));
    }
    // one kernel per class; with more than one class to launch they run side by side on the auxiliary streams
    // (silence-anchored mode on the two head widths: the narrow classes are one launch on the caller's stream, bfa_dp3.inc)
    const bool merged_narrow = mode == 1 && (a.C == 67 || a.C == 17);
    const unsigned kmask = merged_narrow ? (mask & ~3u) : mask;
    const int n_kernels = __builtin_popcount(kmask & 0xf7fu) + __builtin_popcount(xmask & 0xafu) + ((merged_narrow && (mask & 3u)) ? 1 : 0) + (Lmax > 1024 ? 1 : 0) + ((use_mix || cp.seg_mix) ? 1 : 0);
    LaunchFan fan;
    fan.main_stream = stream;
    fan.aux = (hipStream_t *)aux_streams + aux_first;   // (the caller's share of the handle's streams: bfa_align_heads gives each head its own)
    fan.joined = (hipEvent_t *)aux_events + aux_first;
    // (the handle creates its auxiliary streams when a call first has kernels to run side by side: bfa_capi.cpp ensure_aux)
    fan.naux = (n_kernels > 1 && aux_streams && ensure_aux) ? ensure_aux(ctx) : 0;
    fan.naux = fan.naux - aux_first < aux_count ? fan.naux - aux_first : aux_count;
    if (fan.naux < 0) fan.naux = 0;
    fan.forked = fork_event ? (hipEvent_t)*fork_event : nullptr;
    fan.used = 0;
    // several K1 kernels side by side: the window items are walked on the window kernels' stream as soon as those are
    // done (with ONE kernel -- the headline batch -- that would only add a launch to the serial chain)
    a.k2_windows = (per_class_k2 && fan.naux > 0 && (wmask != 0 || (xmask & 0xafu) != 0)) ? 1 : 0;
    if (Lmax > 1024) { // paths of more than 1024 states can occur: the workgroup-wide kernel takes them
        const int big_grid = a.B < 1024 ? a.B : 1024;
        hipStream_t bs = fan.pick();
        for (int huge = 0; huge <= (Lmax > BIG1_MAX_L ? 1 : 0); ++huge) { // (paths beyond 8 192 states: the sixteen-wave kernel, behind the first)
            if (nk <= 2) bfa_launch_dp_big_nk2(&a, big_grid, bs, huge);
            else if (nk <= 5) bfa_launch_dp_big_nk5(&a, big_grid, bs, huge);
            else bfa_launch_dp_big_nk8(&a, big_grid, bs, huge);
        }
        if (per_class_k2) bfa_launch_backtrace_sel(&a, K2_BIG, fused_k2 ? 1 : 0, dp_grid, bs, 2);
    }
    // the fallbacks of a silence-anchored full batch: on the head's third auxiliary stream, beside the pieces (caller's stream) --
    // forked here, ahead of the caller's-stream launches; k_mix itself goes behind the wide exact-window kernels of that lane
    if (cp.seg_mix) (void)fan.lane(2);
    if (nk <= 2) bfa_launch_dp_nk2(&a, mask, mode, dp_grid, &fan);
    else if (nk <= 5) bfa_launch_dp_nk5(&a, mask, mode, dp_grid, &fan);
    else bfa_launch_dp_nk8(&a, mask, mode, dp_grid, &fan);
    if (cp.seg_mix) {
        hipStream_t ms = fan.lane(2);
        if (a.C == 67) bfa_k1_mix_nk5(&a, a.mix_order, ms); else bfa_k1_mix_nk2(&a, a.mix_order, ms);
    }
    if (use_mix) { // on the caller's stream, after the forks (the other classes' kernels run beside it)
        if (a.C == 67) bfa_k1_mix_nk5(&a, a.mix_order, stream); else bfa_k1_mix_nk2(&a, a.mix_order, stream);
    }
    fan.join();
    const bool redo_done = a.k2_windows && (a.C == 67 || a.C == 17); // (launched behind the window kernels on their stream, bfa_dp3.inc)
    if (redo_done) { /* nothing */ }
    else if (nk <= 2) bfa_launch_dp_redo_nk2(&a, mask, mode, stream);
    else if (nk <= 5) bfa_launch_dp_redo_nk5(&a, mask, mode, stream);
    else bfa_launch_dp_redo_nk8(&a, mask, mode, stream);
    if (ev1) (void)hipEventRecord((hipEvent_t)ev1, stream);
    // which of the two walk kernels can find work: narrow (Rw <= 4 / R <= 4, and every non-DP item), wide (Rw 6 / 8, R >= 6,
    // paths beyond 1024 states) -- from the classes that can occur (`mask`) and, for window reruns, the widest path
    const bool any_wide = (mask & (0x78u | (0xa0u << 8) | (0xa0u << 20))) != 0 || Lmax > 256;
    if (per_class_k2) {
        // window / rerun items, fills, items nobody took: wide only if a window item (or its full-layout rerun) can be
        const bool rest_wide = (mask & ((0xa0u << 8) | (0xa0u << 20))) != 0 || Lmax > 256; // (also wide items whose K1 class the hint left out: they are reported as BAD_HINT by the wide walk)
        bfa_launch_backtrace_sel(&a, a.k2_windows ? K2_REST_NOWIN : K2_REST, fused_k2 ? 1 : 0, dp_grid, stream, rest_wide ? 3 : 1);
        if (!fused_k2) hipLaunchKernelGGL(k_assort, dim3(a.B < 65536 ? a.B : 65536), dim3(64), 0, stream, a);
    } else {
        bfa_launch_backtrace(&a, dp_grid, stream, any_wide ? 3 : 1);
        hipLaunchKernelGGL(k_assort, dim3(a.B < 65536 ? a.B : 65536), dim3(64), 0, stream, a);
    }
    return (int)hipGetLastError();
}

extern "C" int bfa_launch_prepare_nk2(const bfa::AlignArgs *args, float *out, int64_t oB, int64_t oT, hipStream_t stream);
extern "C" int bfa_launch_prepare_nk5(const bfa::AlignArgs *args, float *out, int64_t oB, int64_t oT, hipStream_t stream);

// emission preparation only (forced_alignment.py:121-129): k_plan for the target masks, then the block kernel
extern "C" int bfa_launch_prepare(const bfa::AlignArgs *args, float *out, int64_t oB, int64_t oT, void *stream_)
{
    using namespace bfa;
    hipStream_t stream = (hipStream_t)stream_;
    const AlignArgs &a = *args;
    hipLaunchKernelGGL(k_plan, dim3((a.B + 15) / 16), dim3(256), 0, stream, a); // (zeroes the counters itself)
    int done = 0;
    if (a.C <= 32) done = bfa_launch_prepare_nk2(&a, out, oB, oT, stream);
    else if (a.C <= 80) done = bfa_launch_prepare_nk5(&a, out, oB, oT, stream);
    if (!done) return -1;
    return (int)hipGetLastError();
}

extern "C" int bfa_launch_conf(const bfa::ConfArgs *args, void *stream_)
{
    using namespace bfa;
    hipStream_t stream = (hipStream_t)stream_;
    if (args->row_stats) hipLaunchKernelGGL(k_conf<true>, dim3(args->B < 65536 ? args->B : 65536), dim3(64), 0, stream, *args);
    else hipLaunchKernelGGL(k_conf<false>, dim3(args->B < 65536 ? args->B : 65536), dim3(64), 0, stream, *args);
    return (int)hipGetLastError();
}

extern "C" int bfa_launch_log_softmax3_nk2(const float *in, int64_t ld_in, float *out, int64_t ld_out, int64_t rows, int C, hipStream_t stream);
extern "C" int bfa_launch_log_softmax3_nk5(const float *in, int64_t ld_in, float *out, int64_t ld_out, int64_t rows, int C, hipStream_t stream);

extern "C" int bfa_launch_log_softmax(const float *in, int64_t ld_in, float *out, int64_t ld_out, int64_t rows,
                                      int C, void *stream_)
{
    using namespace bfa;
    hipStream_t stream = (hipStream_t)stream_;
    // the two head widths take the sixteen-rows-per-pass block of the alignment producer (bfa_dp3.inc)
    if (C == 17 && bfa_launch_log_softmax3_nk2(in, ld_in, out, ld_out, rows, C, stream)) return (int)hipGetLastError();
    if (C == 67 && bfa_launch_log_softmax3_nk5(in, ld_in, out, ld_out, rows, C, stream)) return (int)hipGetLastError();
    const int nk = (C + 15) / 16;
    int64_t blocks = (rows + 15) / 16;
    if (blocks > 8192) blocks = 8192;
    if (blocks < 1) blocks = 1;
    if (nk <= 2) hipLaunchKernelGGL(k_log_softmax<2>, dim3((unsigned)blocks), dim3(256), 0, stream, in, ld_in, out, ld_out, rows, C);
    else if (nk <= 5) hipLaunchKernelGGL(k_log_softmax<5>, dim3((unsigned)blocks), dim3(256), 0, stream, in, ld_in, out, ld_out, rows, C);
    else hipLaunchKernelGGL(k_log_softmax<8>, dim3((unsigned)blocks), dim3(256), 0, stream, in, ld_in, out, ld_out, rows, C);
    return (int)hipGetLastError();
}
