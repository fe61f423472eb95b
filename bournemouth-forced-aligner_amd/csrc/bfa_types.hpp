// bfa_types.hpp -- device-side records shared by the kernels and the C-ABI layer.
#pragma once
#include <stdint.h>

#include "../../include/bfa.h"

namespace bfa {

constexpr float NEG = -1000.0f;                   // forced_alignment.py:23 `_neg_inf` (finite on purpose)
constexpr float MIN_LOGP = -18.420680999755859375f; // float32 log(1e-8), forced_alignment.py:70 (the default of bfa_params.min_log_prob)
constexpr int MAX_C = 128;                        // target-column mask is 4 x u32
constexpr int MASK_WORDS = MAX_C / 32;
constexpr int MAX_R = 16;                         // CTC states per lane in the wave kernel -> L <= 1024
constexpr int MAX_L_BIG = 16384;                  // workgroup fallback kernel

// what one DP / fill work item covers
enum ItemKind : int32_t {
    ITEM_NONE = 0,
    ITEM_DP = 1,          // banded Viterbi over rows [row0,row0+Ts), writes frames [out0,out0+nout)
    ITEM_FILL_BLANK = 2,  // blank / -1
    ITEM_FILL_PROP = 3,   // proportional assignment forced_alignment.py:170-172
    ITEM_FILL_SIL = 4,    // silence segment, forced_alignment.py:382-397
    ITEM_DONE = 5         // k_mix has aligned AND walked the utterance (nothing left for the walk kernels)
};

struct Item {
    int32_t kind;
    int32_t utt;       // utterance index b
    int32_t row0;      // first posterior row of the DP (padded_start)
    int32_t Ts;        // rows in the DP
    int32_t tok0;      // first target index covered
    int32_t nt;        // number of targets covered
    int32_t stride;    // CTC expansion stride 1..4
    int32_t L;         // stride*nt+1
    int32_t bw;        // Sakoe-Chiba half width, 0 = no band
    int32_t out0;      // first output frame (audio_start)
    int32_t nout;      // output frames
    int32_t pad_left;  // out0 - row0
    int32_t final_state;
    int32_t anch_off;  // >= 0: offset of this DP's per-row silence-anchor counts in the utterance's pool; -1: none
    int64_t bp_off;    // dword offset of this item's backpointer block in the workspace
    int32_t win;       // > 0: K1 used the sliding in-band state window with `win` states per lane (bp: window layout)
    int32_t split;     // 2: K1 split the full layout over two consumer waves (bp: per-frame lane masks, bfa_dp5.inc)
    // Window items only.  XW_FAST: the fast window consumer (its result stands when the final score is above the sentinel);
    // XW_EXACT: the EXACT window consumer takes the item -- every window state is computed exactly as in the reference and
    // the codes of the states outside the window follow from a per-frame side band (see DpCoreW<.., EX>), so the result
    // stands in every regime; XW_REDO: a fast window ended at the sentinel, the exact consumer reruns the item;
    // XW_REDONE: that rerun is done (walked with the rest after the join).
    int32_t xw;
    int32_t t_tail;    // exact-window items: > 0 = K1 left the serial loop at this frame (the scores were dead): k_dp4x_tail writes the rest
};
constexpr int XW_FAST = 0, XW_EXACT = 1, XW_REDO = 2, XW_REDONE = 3;
constexpr int PIECE_BUCKETS = 12, PIECE_CNT0 = 4; // AlignArgs::piece_list: length buckets, their counters in counters[4 .. 15]
__host__ __device__ inline int piece_bucket(int Ts, int Tmax)
{
    const int b = (int)(((int64_t)Ts * PIECE_BUCKETS) / (Tmax + 8));
    return b < 0 ? 0 : (b >= PIECE_BUCKETS ? PIECE_BUCKETS - 1 : b);
}
constexpr uint32_t XWIN_REDO = 0x100u, XWIN_MIX = 0x200u;
// XWIN_ROUTE (round 6): the caller's last calls lost most of their fast windows to the sentinel (bfa_capi.cpp: window routing):
// k_plan hands every fast-window item with a stride >= 3 to the exact rerun kernels at once (XW_REDO), no fast attempt.
constexpr uint32_t XWIN_ROUTE = 0x400u;
// counters[] slots of the exact reruns, standard mode only (the silence-anchored mode keeps its piece counts there):
// items the exact rerun kernels aligned / how many of them ended ABOVE the sentinel (the fast window would have done)
constexpr int CNT_XDONE = 4, CNT_XALIVE = 5;
constexpr int ONE_MAX_BATCH = 1024; // largest single-class call that takes the one-kernel path (k_one; 8192: 0.62 ms against 0.32 + 0.06 for the headline batch)
#ifndef BFA_MIX_MIN_BATCH
#define BFA_MIX_MIN_BATCH 2
#endif
constexpr int MIX_MIN_BATCH = BFA_MIX_MIN_BATCH; // a lone utterance keeps the per-class kernels (its dead tail runs segment-parallel there); 64 until the
                                                 // end of round 4: calls of 4-48 sentences took 1.3-1.7 x as long through the class kernels (profiles/r04_latency_mixed.txt)

// An item record every lane of a wave has loaded from the same address is wave-uniform, but the compiler cannot know (the
// load is a vector load whenever the kernel also writes the item list): everything derived from it -- frame counts, loop
// bounds, the walk's state and frame masks -- then lives in vector registers and is computed by the VALU with exec-mask
// branches.  Passing the fields through v_readfirstlane once puts them, and what follows from them, on the scalar unit.
#ifndef BFA_UNIFORM_ITEM
#define BFA_UNIFORM_ITEM 1 // (A/B: 0 = the records as loaded)
#endif
__device__ __forceinline__ Item uniform_item(const Item &v)
{
#if !BFA_UNIFORM_ITEM
    return v;
#endif
    auto u = [](int x) { return __builtin_amdgcn_readfirstlane(x); };
    Item it;
    it.kind = u(v.kind); it.utt = u(v.utt); it.row0 = u(v.row0); it.Ts = u(v.Ts); it.tok0 = u(v.tok0); it.nt = u(v.nt);
    it.stride = u(v.stride); it.L = u(v.L); it.bw = u(v.bw); it.out0 = u(v.out0); it.nout = u(v.nout);
    it.pad_left = u(v.pad_left); it.final_state = u(v.final_state); it.anch_off = u(v.anch_off);
    it.bp_off = ((int64_t)u((int)(v.bp_off >> 32)) << 32) | (uint32_t)u((int)(v.bp_off & 0xffffffffll));
    it.win = u(v.win); it.split = u(v.split); it.xw = u(v.xw); it.t_tail = u(v.t_tail);
    return it;
}

// one wavefront's candidates for the final-state rule (forced_alignment.py:656-682) when a DP is spread over
// several wavefronts (k_dp_big, k_dp5): rightmost / best state above the sentinel, best of all, dp[L-1], dp[L-2]
struct BigFinal {
    int rm; float bv; int bi; float av; int ai; float vL1, vL2;
};

struct DevParams {
    int32_t blank, sil, anchors, ignore_noise, truly_forced, boost, enforce, simple, max_blanks;
    uint32_t class_mask; // K1 classes to launch (host hint), 0 = derive from the shapes
    uint32_t win_mask;   // sliding-window classes the planner may use (bit Rw-1), set by bfa_launch_align
    uint32_t xwin_mask;  // exact-window classes (bit Rw-1): banded items the fast window is not tried on (long, many tokens);
                         // bit 8: the exact window reruns fast windows that ended at the sentinel; bit 9 (XWIN_MIX): a mixed-length
                         // call -- every window item with stride >= 3 of these classes is an exact-window item (k_mix takes them)
    int32_t win_max_tokens; // no window attempt for utterances with more tokens
    int32_t win_max_frames; // ... or more frames
    float min_logp;         // floor of the target columns: float32 log(min_phoneme_prob), forced_alignment.py:70
};

// everything the kernels of one bfa_align_batch call need; passed by value
struct AlignArgs {
    const float *logp;
    int64_t strideB, strideT;
    // != nullptr: `logp` holds RAW LOGITS (core.py:897); every kernel that prepares rows first applies
    // F.log_softmax(dim=-1) (core.py:898-899) and K1 writes that row's (maximum, log-sum) here, [B, Tmax] pairs, so
    // that the later sparse readers (confidences, soft boundaries) reconstitute log_prob = (x - max) - logsum with the
    // very same two float32 subtractions (no separate log-softmax pass over the logits)
    float *row_stats;
    // Silence-anchored mode on the 16-rows-per-pass kernels: the P(SIL) pass (K0) leaves the (maximum, log-sum) of the
    // BOOSTED rows of every candidate utterance here ([B, Tmax] pairs; nullptr = not in use) and K1's row preparation
    // reuses them instead of evaluating the softmax of the same rows a second time.  ucand[b] = 1 marks the candidates.
    float *row_stats2;
    uint8_t *ucand;
    int32_t B, Tmax, C, Smax;
    const int32_t *T_len, *tokens, *S_len;
    DevParams p;
    int32_t *frame_ph, *frame_idx;
    bfa_segment *segs;
    int32_t seg_cap;
    int32_t *seg_count, *status, *mode;
    // workspace carve
    Item *items;
    int32_t item_cap;
    int32_t *counters;      // [0] = number of items, [1] = number of segmented-mode candidates
    uint32_t *umask;        // [B][MASK_WORDS] target-column bit mask
    int32_t *uT, *uS;       // [B] clamped lengths
    int32_t *umode;         // [B] decode mode (negative while the segmented planner is undecided)
    uint8_t *anchor;        // [B][anchor_per_utt] per-DP silence-anchor counts (segmented mode)
    int32_t anchor_per_utt;
    float *psil;            // [B][Tmax] exp(m[:,SIL]) of the boosted/floored rows (segmented mode)
    int32_t *cand;          // [B] utterances whose target contains SIL (candidates for the segmented mode)
    int32_t *seg_scratch;   // [B][seg_scratch_per_utt] planner scratch
    int32_t seg_scratch_per_utt;
    uint32_t *bp;           // backpointer pool
    int64_t bp_cap;         // dwords
    int64_t bp_per_utt;     // dwords reserved for the one-DP-per-utterance case
    int32_t k2_sel;         // K2: which items this launch takes (K2_ALL / K2_FULL + R / K2_BIG / K2_REST)
    int32_t k2_fused_rle;   // K2: every utterance is one item and the walk also emits its tuples (no K3a launch)
    int32_t k2_per_class;   // launcher: one K2 launch per full-layout class behind its K1 kernel(s), K2_REST after the join
    int32_t xcd_contig;     // K1 (one item per workgroup): slot = xcd_eighth(workgroup id) (BFA_HINT_UNIFORM_LENGTHS)
    int32_t k2_windows;     // launcher: the window items are walked behind the window kernels on their stream (K2_WIN)
    uint8_t *mix_key;       // [B] mixed-length calls (DevParams::xwin_mask & XWIN_MIX): 255 - cost bucket of the utterance (k_plan)
    int32_t *mix_order;     // [B] ... the utterance slots by decreasing cost (k_order), the work list of k_mix
    // silence-anchored mode: the DP pieces by length, longest first, for k_dp4_any.  k_plan_seg appends a piece's item index to
    // the list of its length bucket (counters[PIECE_CNT0 + bucket] entries in piece_list[bucket * item_cap ..]); workgroup
    // B + k of k_dp4_any takes the k-th entry counting from the longest bucket down (bfa_dp4.inc: any_item)
    int32_t *piece_list;
    // window routing (bfa_capi.cpp): 8 ints of host-mapped memory the last walk kernel of a standard-mode call leaves this
    // call's window statistics in ({tag, fast windows that gave up, exact reruns, of those alive, B, routed}); nullptr = off
    int32_t *hist;
    int32_t hist_tag;
    int32_t wide_any_max;   // silence-anchored mode: calls of at most this many utterances take the wide classes as ONE launch, slots included (k_dp5_any)
    int32_t mix_exact_only; // k_mix in a silence-anchored call: the exact-window utterance slots (standard-mode fallbacks) only
    int32_t pieces_merged;  // launcher: the pieces of the wide classes are k_dp5_any's, the class kernels take the utterance slots only
};
constexpr int K2_ALL = 0, K2_REST = 1, K2_BIG = 2, K2_WIN = 3, K2_REST_NOWIN = 4, K2_XWIN = 5, K2_FULL = 16; // K2_FULL + R, R in {2,3,4,6,8,12,16}
constexpr int K2_NARROW = 64; // | (class bits of the merged narrow full-layout classes << 8): K2_WIN + those classes (k_dp4w_any)

struct ConfArgs {
    const float *logp;
    float *row_stats; // != nullptr: logp is raw logits, see AlignArgs::row_stats (rows without statistics get them here)
    int64_t strideB, strideT;
    int32_t B, Tmax, C;
    const int32_t *T_rows;
    const bfa_segment *segs;
    int32_t seg_cap;
    const int32_t *seg_count;
    float *conf;
    int32_t *status;
};

// states-per-lane class of a CTC path of L states (0 = too long for the wave kernel)
__host__ __device__ inline int r_class_for_L(int L)
{
    const int r = (L + 63) / 64;
    if (r <= 2) return 2;
    if (r <= 3) return 3;
    if (r <= 4) return 4;
    if (r <= 6) return 6;
    if (r <= 8) return 8;
    if (r <= 12) return 12;
    if (r <= 16) return 16;
    return 0;
}
// CTC paths longer than 16 states x 64 lanes are split over the wavefronts of one workgroup (k_dp_big): wave w owns
// states [1024 w, 1024 w + 1024), 16 per lane; backpointers row-major, [frame][ceil(L/16)] dwords, dword g = states
// 16g..16g+15, 2 bits each (LSB first).
constexpr int BFA_FALLBACK_TOO_SHORT = 4; // planner-internal umode: the standard-mode fallback of a segmented candidate is the T < S error
constexpr int FINAL_NOT_COMPUTED = -2; // Item::final_state until a K1 kernel has taken the item (K2 reports the rest)
constexpr int BIG_WAVES = 8;                  // k_dp_big: paths of 1 025 .. 8 192 states, 16 per lane
constexpr int BIG1_MAX_L = 1024 * BIG_WAVES;
// ... and of 8 193 .. 32 768 states (round 5): sixteen wavefronts (a full workgroup) of 32 states per lane, two backpointer
// dwords per lane and frame, the staged rows in dynamic LDS
constexpr int BIG2_WAVES = 16, BIG2_R = 32;
constexpr int BIG_MAX_L = 64 * BIG2_R * BIG2_WAVES;
__host__ __device__ inline unsigned r_class_bit(int R)
{
    switch (R) {
    case 2: return 1u; case 3: return 2u; case 4: return 4u; case 6: return 8u;
    case 8: return 16u; case 12: return 32u; case 16: return 64u; default: return 0u;
    }
}
// ---- sliding in-band window (K1 consumer, standard-mode items with an active band) ----------------
// The reference masks every state outside [t*pace - bw, t*pace + bw] to -1000 after each frame
// (forced_alignment.py:650-653).  With emissions <= 0, whenever the final score is above the sentinel the traced
// path never leaves the band, so only the in-band states need computing: a window of 64*Rw consecutive states
// that slides with the band.  Rw = states per lane of the window, 0 = not applicable (keep the full layout).
// Sliding-window consumer (DpCoreW): 64*Rw states follow the band.  A lane's backpointer dword covers
// win_frames_per_word(Rw) frames (2 bits x Rw slots per frame) and the window only moves between dwords, so it must
// hold the band of all those frames: the 2*bw+1 in-band states + the band's advance over the dword (pace <= 1) +
// the two below-band predecessor states + the lane granularity of the move + 1 of slack.
// The window result is only valid while path scores stay above the sentinel -1000 (otherwise the item is redone with
// the full layout, which costs more than never trying): scores are sums of per-frame log-probabilities, so long
// utterances cross it even with good posteriors.  Past WIN_MAX_FRAMES frames the planner does not try the window.
// Workgroup id -> utterance slot with every XCD on ONE contiguous eighth of the batch (the dispatcher places workgroup w on
// XCD w % 8, so with slot = w every XCD streams every eighth utterance).  Used when the caller says the utterances of the
// call have about the same number of frames (BFA_HINT_UNIFORM_LENGTHS -> AlignArgs::xcd_contig): headline K1 0.3215 -> 0.3127 ms
// in three interleaved pairs; without that promise a call ordered by length would put all the long utterances on one XCD
// (C4 7.4 -> 11.7 ms), and chunked forms (4 ... 256 utterances per XCD turn) stay within the +-3 % that the placement of the
// posterior buffer decides anyway (profiles/r03_placement.txt).  Bijective on [0, n) for any n.
__host__ __device__ inline int xcd_eighth(int w, int n)
{
    const int q = n >> 3, r = n & 7, x = w & 7;
    return x * q + (x < r ? x : r) + (w >> 3);
}

// (band / window changes laid out as cold branches were measured slower for the headline K1, 0.321 -> 0.341 ms)
#define BFA_UNLIKELY(x) (x)
constexpr int WIN_MAX_FRAMES = 1536;
// ... and every token costs the path at least one frame in a blank state; with a model that does not put noise between
// phonemes that is about -15 per token after the boost, so past ~64 tokens the -1000 line is usually crossed too.
// bfa_params.window_max_tokens > 0 overrides this limit (e.g. a large value for posteriors known to be CTC-like).
constexpr int WIN_MAX_TOKENS = 64;
// Window backpointers as lane masks written by the scalar unit: the consumer spends one v_cmp per code bit (the result
// lands in an SGPR pair and leaves through s_store_dwordx4) instead of v_cmp + v_addc into per-lane packed dwords, and K2
// reads its frame's masks straight from memory (no LDS staging).  (false: the packed-dword form, Rw <= 4 only.)
constexpr bool WIN_SSTORE = true;
__host__ __device__ inline int win_frames_per_word(int rw) { return rw == 1 ? 16 : (rw == 2 ? 8 : 4); }
__host__ __device__ inline int win_class_for(int L, int bw, int Ts, int max_frames = WIN_MAX_FRAMES)
{
    if (bw <= 0 || Ts > max_frames) return 0;
    const int rfull = r_class_for_L(L);
    if (rfull == 0) return 0;
    constexpr int classes[6] = {1, 2, 3, 4, 6, 8}; // window states-per-lane classes
    for (int k = 0; k < 6; ++k) {
        const int rw = classes[k];
        if (rw >= rfull) break;
        if (2 * bw + 1 + win_frames_per_word(rw) + 2 + rw + 1 <= 64 * rw) return rw;
    }
    return 0;
}
// window class bits in the class mask: bit 8 + (Rw-1), Rw in {1,2,3,4,6,8}
__host__ __device__ inline unsigned win_class_bit(int rw) { return (rw >= 1 && rw <= 8) ? (1u << (7 + rw)) : 0u; }

// all classes up to and including the class of L
__host__ __device__ inline unsigned r_class_mask_upto(int L)
{
    const int R = r_class_for_L(L);
    if (R == 0) return 127u;
    const unsigned bit = r_class_bit(R);
    return bit | (bit - 1u);
}
__host__ __device__ inline int bp_words_for_R(int R) { return (R + 3) / 4; } // dwords per lane per 4 frames

// Backpointer block of one DP: [quad][w][lane] dwords, lane < nl = ceil(L/R) (only lanes that own a
// state are stored).  Dword w of lane l holds, for the four frames of the quad, the 2-bit codes of the
// lane's register slots 4w..4w+3:  bits [8*f + 2*(r&3), +2) = k(frame 4q+f, state l*R + r).
__host__ __device__ inline int bp_lanes(int L, int R) { return (L + R - 1) / R; }
__host__ __device__ inline int64_t bp_dwords(int Ts, int L)
{
    const int R = r_class_for_L(L);
    const int64_t quads = (Ts + 3) / 4;
    if (R > 0) return quads * bp_words_for_R(R) * bp_lanes(L, R) + quads; // + room for the window layout's per-quad base
    return (int64_t)Ts * ((L + 15) / 16); // big-L kernel: 2 bits per state, row-major
}

// K1 launches one kernel per register class; the classes work on disjoint items, so on a mixed-length batch they
// run side by side on the handle's auxiliary streams (forked from and joined back into the caller's stream).
#ifdef __HIPCC__
struct LaunchFan {
    hipStream_t main_stream;
    hipStream_t *aux;   // may be null
    hipEvent_t *joined; // one per aux stream
    hipEvent_t forked;
    int naux, used;
    unsigned touched = 0; // aux streams that have been forked from the caller's stream in this call
    // aux stream k (forked from the caller's stream at its first use)
    hipStream_t lane(int k)
    {
        if (naux <= 0) return main_stream;
        k %= naux;
        if (touched == 0) (void)hipEventRecord(forked, main_stream);
        if (!(touched & (1u << k))) { (void)hipStreamWaitEvent(aux[k], forked, 0); touched |= 1u << k; }
        return aux[k];
    }
    hipStream_t pick() { return lane(used++); }
    void join()
    {
        for (int k = 0; k < naux; ++k) {
            if (!(touched & (1u << k))) continue;
            (void)hipEventRecord(joined[k], aux[k]);
            (void)hipStreamWaitEvent(main_stream, joined[k], 0);
        }
        touched = 0;
        used = 0;
    }
};
#endif

} // namespace bfa
