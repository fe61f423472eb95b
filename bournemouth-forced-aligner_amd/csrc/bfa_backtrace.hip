// bfa_backtrace.hip -- K2: backtrace over the packed backpointers + framewise outputs
// (forced_alignment.py:686-700), plus the non-DP fills (proportional :170-172, silence :382-397).
//
// One wavefront per work item.  The serial dependence state[t-1] = bp[t][state[t]] is only a real
// dependence at frames where the path moves (k != 0): about L of the T frames.  So instead of one
// step per frame, all 64 lanes look at 64 consecutive frames at once for the CURRENT state, a ballot
// finds the latest frame with a move, every frame above it takes the current state, the state is
// updated and the search continues below that frame: ~ (#moves + T/64) wave steps per utterance.
//
// Every wave of the launch is resident at once and a walk step is a short dependent chain
// (LDS gather -> extract -> ballot -> readlane), so the kernel time is (steps x step latency) of one
// utterance: the walk is instantiated per backpointer layout (states per lane, window or full) so that the
// per-step address and shift arithmetic folds into a handful of instructions.
#include <hip/hip_runtime.h>

#include "bfa_assort.hpp"
#include "bfa_math.hpp"
#include "bfa_types.hpp"

namespace bfa {

__device__ __forceinline__ void wave_sync_lds()
{
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}

// assort_frames (forced_alignment.py:777-834) DURING the walk (utterances that are one DP item): the walk visits the
// chunks from the last frame down, so the runs come out in reverse.  A frame that ends a run (its (phoneme, index) pair
// differs from the frame above it, or it is the last frame) tells where the run ABOVE it starts; that run's end is the
// next run end above, or -- past the top of the chunk -- the end carried over from the chunks already walked.  Tuples
// go to the back of the utterance's segment array (slot cap-1-k for the k-th tuple found) and are moved to the front
// by finish(); if there are more than seg_cap of them the forward pass (assort_utterance) redoes the utterance, which
// also sets the overflow status exactly as the K3a kernel does.
struct RleRev {
    bfa_segment *out;
    int cap, count;
    int c_end;          // end (exclusive) of the run that reaches down into the chunk being walked
    int n_ph, n_id;     // pair of the lowest frame walked so far
    int blank, ignore_noise, max_blanks, Tr;
    bool overflow;
    __device__ __forceinline__ void init(const AlignArgs &a, int b, int Tr_)
    {
        out = a.segs + (int64_t)b * a.seg_cap; cap = a.seg_cap; count = 0; c_end = Tr_; n_ph = 0; n_id = 0;
        blank = a.p.blank; ignore_noise = a.p.ignore_noise; max_blanks = a.p.max_blanks; Tr = Tr_; overflow = false;
    }
    __device__ __forceinline__ bool emits(int ph, int len) const
    {
        return (ph == blank) ? (!ignore_noise && len > max_blanks) : true; // :819-831
    }
    // frames [t0, t0+n) of the utterance, lane l <-> frame t0+l, (ph, id) valid for l < n
    __device__ __forceinline__ void chunk(int ph, int id, int t0, int n, int lane)
    {
        int uph = __builtin_amdgcn_update_dpp(0, ph, 0x130, 0xf, 0xf, true); // wave_shl:1, lane l <- lane l+1
        int uid = __builtin_amdgcn_update_dpp(0, id, 0x130, 0xf, 0xf, true);
        if (lane == n - 1) { uph = n_ph; uid = n_id; }
        const int t = t0 + lane;
        const bool in = lane < n;
        const bool is_end = in && ((t == Tr - 1) || ph != uph || id != uid);
        const unsigned long long E = __ballot(is_end);
        // the run above an end frame: [t+1, next end above) with the pair of frame t+1; none above the last frame
        const unsigned long long above = (lane >= 63) ? 0ull : (E >> (lane + 1));
        const int r_end = above ? (t + 1 + __builtin_ctzll(above) + 1) : c_end;
        const bool has_run = is_end && (t + 1 < Tr);
        const bool emit = has_run && emits(uph, r_end - (t + 1));
        const unsigned long long em = __ballot(emit);
        if (emit) {
            const unsigned long long higher = (lane >= 63) ? 0ull : (em >> (lane + 1));
            const int k = count + __builtin_popcountll(higher);
            if (k < cap) { bfa_segment sg; sg.phoneme = uph; sg.start = t + 1; sg.end = r_end; sg.target_idx = uid; out[cap - 1 - k] = sg; }
        }
        count += __builtin_popcountll(em);
        if (E) c_end = t0 + __builtin_ctzll(E) + 1;
        n_ph = __builtin_amdgcn_readlane(ph, 0);
        n_id = __builtin_amdgcn_readlane(id, 0);
    }
    // after frame 0 has been walked: the run that starts at frame 0, then the tuples move to the front
    __device__ __forceinline__ void finish(const AlignArgs &a, int b, int lane)
    {
        if (Tr > 0 && emits(n_ph, c_end)) {
            if (lane == 0 && count < cap) { bfa_segment sg; sg.phoneme = n_ph; sg.start = 0; sg.end = c_end; sg.target_idx = n_id; out[cap - 1 - count] = sg; }
            count += 1;
        }
        if (count > cap) { // more runs than seg_cap: the forward pass keeps the FIRST seg_cap and reports the overflow
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
            __builtin_amdgcn_wave_barrier();
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
            assort_utterance<1>(a, b, lane);
            return;
        }
        // out[cap-count .. cap) holds the tuples last-to-first: tuple j (front order) sits at cap-count+ (count-1-j) ...
        // written as slot cap-1-k for the k-th found = the k-th from the END, i.e. front index j = count-1-k lives at
        // cap-1-k = cap-count+j: already in front order inside the back block; move the block down by cap-count.
        const int shift = cap - count;
        if (shift > 0) {
            for (int j0 = 0; j0 < count; j0 += 64) { // ascending: a batch's sources lie above everything written so far
                const int j = j0 + lane;
                bfa_segment sg;
                if (j < count) sg = out[shift + j];
                __builtin_amdgcn_wave_barrier();
                if (j < count) out[j] = sg;
                __builtin_amdgcn_wave_barrier();
            }
        }
        if (lane == 0) {
            a.seg_count[b] = count;
            if (a.mode) { const int md = a.umode[b]; a.mode[b] = md < 0 ? (-1 - md) : md; }
        }
    }
};

// Backpointer layouts (written by DpCore / DpCoreW in bfa_dp3.inc and k_dp in bfa_dp.inc):
//   full   : rows of 4 frames, W = ceil(R/4) dwords per lane and row, lanes < nl = ceil(L/R); dword w of lane l
//            holds slots 4w..4w+3 (Rsub of them), pair (frame f, slot r) at bits 2*(4*Rsub-1-(f*Rsub+(r&3))).
//   window : [nrows] window bases, then rows of FPW = win_frames_per_word(R) frames, one dword per lane, 64 lanes;
//            lane l, slot r of a row holds state base[row] + l*R + r, pair (f, r) at bits 2*(FPW*R-1-(f*R+r)).
// code = (A<<1)|B with A = (c0 < best), B = (c1 < best): k = A ? (B ? 2 : 1) : 0 = max(code,1) - 1.
template <int R, bool WIN>
__device__ __forceinline__ void walk_item(const AlignArgs &a, const Item &it, uint32_t *sbp, int32_t *stok, int lane, RleRev &rle, bool do_rle)
{
    constexpr int W = WIN ? 1 : (R + 3) / 4;
    constexpr int FPW = WIN ? (R == 1 ? 16 : R == 2 ? 8 : 4) : 4;
    constexpr int FSH = (FPW == 16) ? 4 : (FPW == 8) ? 3 : 2;
    constexpr int CQ = WIN ? 64 / FPW : 16 / W; // rows per chunk
    constexpr int CF = CQ * FPW;                // frames per chunk (<= 64)
    constexpr int NDW = WIN ? CQ : 16;          // dwords per lane and chunk
    constexpr int NPRE = 3;                     // chunks in flight in registers
    const DevParams &p = a.p;
    const int b = it.utt;
    int32_t *oph = a.frame_ph + (int64_t)b * a.Tmax;
    int32_t *oid = a.frame_idx + (int64_t)b * a.Tmax;
    const int32_t *tok = a.tokens + (int64_t)b * a.Smax + it.tok0;
    const int Ts = it.Ts, L = it.L;
    const int nrows = (Ts + FPW - 1) >> FSH;
    const int nl = WIN ? 64 : bp_lanes(L, R);
    const uint32_t *bp_base = a.bp + it.bp_off;
    const uint32_t *bp = bp_base + (WIN ? nrows : 0);
    const int inv_stride = 65536 / it.stride + 1; // (d * inv) >> 16 == d / stride for d < 1100, stride <= 4
    int s = it.final_state;                       // wave-uniform walk state
    int sl = s / R, sr = s - sl * R;              // full layout: its (lane, register slot)
    const int nchunks = (Ts + CF - 1) / CF;

    // the item's tokens go to LDS: a global gather per chunk would sit between the backpointer loads in flight
    // and the stores, and waiting for it (vmcnt counts in order) would wait for all of them
    for (int j = lane; j < it.nt; j += 64) stok[j] = tok[j];

    uint32_t pre[NPRE][NDW + 1]; // + the window base of this lane's row (loaded with the chunk, not at its use)
    // Unconditional loads with clamped addresses (a chunk index below 0 re-reads chunk 0, dwords past the chunk
    // re-read its last one; neither is used): predicated loads sit in their own basic blocks, and the wait before
    // the LDS staging then degrades to vmcnt(0), i.e. waits for the chunks that were meant to stay in flight.
    auto fetch = [&](uint32_t (&dst)[NDW + 1], int c) {
        const int q0 = max(c, 0) * CQ;
        const int q1 = min(nrows, q0 + CQ);
        const int ndw = (q1 - q0) * W * nl;
        const uint32_t *src = bp + (int64_t)q0 * W * nl;
#pragma unroll
        for (int d = 0; d < NDW; ++d) dst[d] = src[min(d * 64 + lane, ndw - 1)];
        if (WIN) dst[NDW] = bp_base[min(max(c, 0) * CF + lane, Ts - 1) >> FSH];
        else dst[NDW] = 0u;
    };
#pragma unroll
    for (int k = 0; k < NPRE; ++k) fetch(pre[k], nchunks - 1 - k);

    // one chunk: stage its dwords in LDS, refill the register buffer with the chunk NPRE further down, walk
    auto walk_chunk = [&](uint32_t (&buf)[NDW + 1], int c) {
        const int t0 = c * CF;
        const int t1 = min(Ts, t0 + CF);
        wave_sync_lds();
#pragma unroll
        for (int d = 0; d < NDW; ++d) sbp[d * 64 + lane] = buf[d];
        wave_sync_lds();
        const int wbase = (int)buf[NDW]; // window base of this lane's row
        fetch(buf, c - NPRE);

        const int t = t0 + lane; // this lane's frame
        const int row = (t >> FSH) - (t0 >> FSH);
        const int fw = t & (FPW - 1); // frame within its row
        const bool mine = t < t1;
        const bool can_move = mine && t > 0;
        int my_state = 0;
        int t_hi = t1 - 1; // frames (.., t_hi] still to be labelled in this chunk
        // per-lane constants of the gather
        int rowoff, sh0;
        if (WIN) {
            rowoff = row * 64;
            sh0 = 2 * (FPW * R - 1 - fw * R);
        } else {
            rowoff = row * W * nl;
            sh0 = 0;
        }
        if (WIN) {
            // lean loop: one exit, scalar bookkeeping only (window items never wrap below state 0: their path scores
            // are above the sentinel)
            const int n = t1 - t0;
            unsigned long long todo = (n == 64) ? ~0ull : ((1ull << n) - 1ull); // frames the walk has not passed
            if (t0 == 0) todo &= ~1ull;                                          // frame 0 has no predecessor
            int hi = n - 1;                                                      // lanes (.., hi] still unlabelled
            for (;;) {
                // a frame at which s lies outside the window is below the next move: its code is not used
                const int d = s - wbase;
                const int dc = min(max(d, 0), 64 * R - 1);
                const int xl = dc / R, xr = dc - xl * R;
                const uint32_t wd = sbp[rowoff + xl];
                const uint32_t code = ((unsigned)d < (unsigned)(64 * R)) ? ((wd >> (sh0 - 2 * xr)) & 3u) : 0u;
                const unsigned long long mv = __ballot(code >= 2u) & todo; // (A<<1)|B : k = A ? (B ? 2 : 1) : 0
                const int jl = (mv == 0ull) ? 0 : 63 - __builtin_clzll(mv); // latest frame at which the path moves
                if (lane >= jl && lane <= hi) my_state = s;
                if (mv == 0ull) break; // the path stays in s down to the chunk start
                s -= (int)__builtin_amdgcn_readlane((int)code, jl) - 1;
                hi = jl - 1;
                todo &= (1ull << jl) - 1ull;
            }
        } else
        while (t_hi >= t0) {
            // backpointer code of this lane's frame for the CURRENT state (state[t-1] = s - k at frame t)
            uint32_t code = 0;
            if (WIN) {
                // a frame at which s lies outside the window is below the next move: its code is not used
                const int d = s - wbase;
                const int dc = min(max(d, 0), 64 * R - 1);
                const int xl = dc / R, xr = dc - xl * R;
                const uint32_t wd = sbp[rowoff + xl];
                code = ((unsigned)d < (unsigned)(64 * R)) ? ((wd >> (sh0 - 2 * xr)) & 3u) : 0u;
            } else {
                const int w = sr >> 2;                  // wave-uniform
                const int rsub = min(4, R - 4 * w);     // slots in dword w
                const uint32_t wd = sbp[rowoff + w * nl + sl];
                code = (wd >> (2 * (4 * rsub - 1 - (fw * rsub + (sr & 3))))) & 3u;
            }
            const uint32_t k = (can_move && t <= t_hi) ? (max(code, 1u) - 1u) : 0u;
            const unsigned long long mv = __ballot(k != 0);
            if (mv == 0) { // the path stays in s down to the chunk start
                if (t <= t_hi) my_state = s;
                break;
            }
            const int jl = 63 - __builtin_clzll(mv); // latest frame (lane) at which the path moves
            if (t <= t_hi && lane >= jl) my_state = s;
            const int kk = (int)__builtin_amdgcn_readlane((int)k, jl);
            const bool at_top = (t0 + jl == t_hi);
            s -= kk;
            if (!WIN) {
                sr -= kk;
                while (sr < 0) { sr += R; sl -= 1; }
            }
            if (s < 0) { s += L; sl = s / R; sr = s - sl * R; } // python negative-index wrap (:692)
            t_hi = t0 + jl - 1;
            if (!WIN && at_top && kk == 2 && t_hi >= t0 && s >= 2) {
                // dense descent (paths below the reference's sentinel step down two states per frame, see
                // walk_item_mask): lane t looks up its frame's code at the state the path has if it kept descending
                // by two from the top; the run of "moves by two" below the top is taken in one round
                const int hi_l = t_hi - t0;
                const int sp = s - 2 * (hi_l - lane);
                const bool cand = can_move && lane <= hi_l && sp >= 0;
                const int spc = cand ? sp : 0;
                const int pl = spc / R, pr = spc - pl * R, pw = pr >> 2, prs = min(4, R - 4 * pw);
                const uint32_t wd2 = sbp[rowoff + pw * nl + pl];
                const uint32_t code2 = (wd2 >> (2 * (4 * prs - 1 - (fw * prs + (pr & 3))))) & 3u;
                const bool ok = cand && code2 == 3u; // (A << 1) | B: k = 2
                const unsigned long long oks = __ballot(ok) << (63 - hi_l); // bit 63 = the top frame
                int run = (~oks == 0ull) ? 64 : __builtin_clzll(~oks);
                run = min(run, hi_l + 1);
                if (run > 0) {
                    if (lane <= hi_l && lane > hi_l - run) my_state = sp;
                    s -= 2 * run;
                    if (s < 0) s += L; // the run's last move may leave state 0 / 1 downwards: python negative-index wrap (:692)
                    sl = s / R; sr = s - sl * R;
                    t_hi -= run;
                }
            }
        }
        int ph = p.blank, id = -1;
        if (mine) {
            const int o = t - it.pad_left; // :447-448 trim the boundary padding
            if (o >= 0 && o < it.nout) {
                if (my_state >= 1) {
                    const int q = ((my_state - 1) * inv_stride) >> 16;
                    if ((my_state - 1) - q * it.stride == 0 && q < it.nt) { ph = stok[q]; id = it.tok0 + q; }
                }
                oph[it.out0 + o] = ph;
                oid[it.out0 + o] = id;
            }
        }
        if (do_rle) rle.chunk(ph, id, t0, t1 - t0, lane); // (one-item utterances: pad_left = 0, nout = Ts)
    };
    // the register buffers keep their roles (no copies, which would wait for the loads in flight)
    static_assert(NPRE == 3, "the unrolled group below is written for three buffers");
    int cg = nchunks - 1;
    for (; cg >= NPRE - 1; cg -= NPRE) { // no branches around the chunks of a group (see fetch)
        walk_chunk(pre[0], cg);
        walk_chunk(pre[1], cg - 1);
        walk_chunk(pre[2], cg - 2);
    }
    if (cg >= 0) walk_chunk(pre[0], cg);
    if (cg >= 1) walk_chunk(pre[1], cg - 1);
}

// Lane-mask layout of the window items (WIN_SSTORE): [nrows] window bases (FPW frames per row), padded to 16 bytes, then
// per frame R x {maskA, maskB} (WIN = false would be the same masks without the header and with base 0).  Bit l
// of maskA / maskB of slot r = A / B of state base + l*R + r at that frame (base = 0 without a window; A = c0 < best,
// B = c1 < best; k = A ? (B ? 2 : 1) : 0).  Lane = frame: a lane keeps its frame's 2R masks in registers (loaded ahead,
// coalesced), so a walk step is select / shift / ballot with no memory access.
template <int R, bool WIN, int NC = 1>
__device__ __forceinline__ void walk_item_mask(const AlignArgs &a, const Item &it, uint32_t *sbp, int32_t *stok, int lane, RleRev &rle, bool do_rle)
{
    constexpr int FPW = (R == 1) ? 16 : (R == 2) ? 8 : 4;
    constexpr int FSH = (FPW == 16) ? 4 : (FPW == 8) ? 3 : 2;
    // R <= 4: a lane keeps its frame's 2R masks in registers.  Wider layouts (the wide walk kernel): the chunk's masks are
    // staged in LDS, lane = frame, and a step reads its pair from there -- instead of a chain of R-1 vector selects per
    // 64-bit mask; the registers are free for the next chunk's loads as soon as the chunk is staged.
    constexpr bool STAGE = (R > 4);
    constexpr int NPRE = STAGE ? 2 : 3;                     // chunks in flight in registers (2R qwords per lane each)
    constexpr int NQ = 2 * R;                              // 64-bit masks per frame
    constexpr int LSTR = NQ + 1;                           // qwords per lane in LDS (+1: bank spread)
    unsigned long long *sm64 = reinterpret_cast<unsigned long long *>(sbp);
    const DevParams &p = a.p;
    const int b = it.utt;
    int32_t *oph = a.frame_ph + (int64_t)b * a.Tmax;
    int32_t *oid = a.frame_idx + (int64_t)b * a.Tmax;
    const int32_t *tok = a.tokens + (int64_t)b * a.Smax + it.tok0;
    const int Ts = it.Ts, L = it.L;
    const int nrows = (Ts + FPW - 1) >> FSH;
    const uint32_t *bp_base = a.bp + it.bp_off;
    const unsigned long long *masks = (const unsigned long long *)(bp_base + (WIN ? ((nrows + 3) & ~3) : 0));
    const int inv_stride = 65536 / it.stride + 1; // (d * inv) >> 16 == d / stride for d < 1100, stride <= 4
    int s = it.final_state;                       // wave-uniform walk state
    const int nchunks = (Ts + 63) >> 6;

    wave_sync_lds(); // the previous item's readers of stok are done
    for (int j = lane; j < it.nt && j < 1024; j += 64) stok[j] = tok[j]; // see walk_item
    wave_sync_lds();

    unsigned long long pre[NPRE][NQ];
    int preb[NPRE];
    auto fetch = [&](unsigned long long (&dst)[NQ], int &dstb, int c) { // unconditional, clamped (see walk_item)
        const int t = min(max(c, 0) * 64 + lane, Ts - 1);
        const unsigned long long *src = masks + (int64_t)t * NQ;
#pragma unroll
        for (int q = 0; q < NQ; ++q) dst[q] = src[q];
        dstb = WIN ? (int)bp_base[t >> FSH] : 0;
    };
#pragma unroll
    for (int k = 0; k < NPRE; ++k) fetch(pre[k], preb[k], nchunks - 1 - k);

    auto walk_chunk = [&](unsigned long long (&buf)[NQ], int &bufb, int c) {
        const int t0 = c * 64;
        const int n = min(Ts - t0, 64); // frames in this chunk
        unsigned long long mk[STAGE ? 1 : NQ];
        if (!STAGE) {
#pragma unroll
            for (int q = 0; q < NQ; ++q) mk[q] = buf[q];
        }
        const int wbase = bufb;
        if (STAGE) {
            wave_sync_lds(); // the previous chunk's readers are done
#pragma unroll
            for (int q = 0; q < NQ; ++q) sm64[lane * LSTR + q] = buf[q];
            wave_sync_lds();
        }
        fetch(buf, bufb, c - NPRE);
        const int t = t0 + lane;
        unsigned long long todo = (n == 64) ? ~0ull : ((1ull << n) - 1ull); // frames the walk has not passed
        if (t0 == 0) todo &= ~1ull;                                          // frame 0 has no predecessor
        int hi = n - 1;                                                      // lanes (.., hi] still unlabelled
        int my_state = 0;
        for (;;) {
            // (window) a frame at which s lies outside the window is below the next move: its code is not used
            const int d = s - wbase;
            const int dc = min(max(d, 0), 64 * R - 1);
            int xl, xr; // lane and mask slot of the state
            if (NC == 1) { xl = dc / R; xr = dc - xl * R; }
            else { // the layout split over NC consumer waves (bfa_dp5.inc): wave hh owns 64*RS contiguous states
                constexpr int RS = R / NC;
                const int hh = dc / (64 * RS), rem = dc - hh * 64 * RS;
                xl = rem / RS;
                xr = hh * RS + (rem - xl * RS);
            }
            unsigned long long mA, mB;
            if (!STAGE) {
                mA = mk[0]; mB = mk[1];
#pragma unroll
                for (int r = 1; r < R; ++r) { if (xr == r) { mA = mk[2 * r]; mB = mk[2 * r + 1]; } }
            } else {
                mA = sm64[lane * LSTR + 2 * xr];
                mB = sm64[lane * LSTR + 2 * xr + 1];
            }
            const bool inw = (unsigned)d < (unsigned)(64 * R);
            const unsigned A = inw ? (unsigned)((mA >> xl) & 1ull) : 0u;
            const unsigned long long mv = __ballot(A != 0u) & todo;
            const int jl = (mv == 0ull) ? 0 : 63 - __builtin_clzll(mv); // latest frame at which the path moves
            if (lane >= jl && lane <= hi) my_state = s;
            if (mv == 0ull) break; // the path stays in s down to the chunk start
            const unsigned B = (unsigned)((mB >> xl) & 1ull);
            const int kmove = 1 + (int)__builtin_amdgcn_readlane((int)B, jl); // k = A ? (B ? 2 : 1) : 0
            const bool at_top = (jl == hi);
            s -= kmove;
            if (!WIN && s < 0) s += L;                           // python negative-index wrap (:692)
            hi = jl - 1;
            todo &= (1ull << jl) - 1ull;
            if constexpr (STAGE && !WIN) {
                // Dense descent.  Once the path score has crossed the reference's -1000 sentinel (long utterances: more
                // than ~66 targets) every state is "dead" and the stored backpointers step down two states per frame:
                // the walk then moves at EVERY frame, one ballot round per frame (8 us per 64-frame chunk).  After a
                // move by two at the top frame the lanes test the continuation in one round: lane t assumes the path
                // kept descending by two from the top (state s - 2 (hi - t)) and looks up ITS frame's pair at that state;
                // the run of lanes below the top that all say "moves by two" is taken at once.
                if (at_top && kmove == 2 && hi >= 0 && s >= 0) {
                    const int sp = s - 2 * (hi - lane);
                    const bool cand = lane <= hi && ((todo >> lane) & 1ull) != 0ull && sp >= 0;
                    const int spc = cand ? sp : 0;
                    int pl, pq; // lane and mask slot of the assumed state
                    if (NC == 1) { pl = spc / R; pq = spc - pl * R; }
                    else {
                        constexpr int RS = R / NC;
                        const int ph = spc / (64 * RS), rem = spc - ph * 64 * RS;
                        pl = rem / RS;
                        pq = ph * RS + (rem - pl * RS);
                    }
                    const unsigned long long pA = sm64[lane * LSTR + 2 * pq], pB = sm64[lane * LSTR + 2 * pq + 1];
                    const bool ok = cand && (((pA & pB) >> pl) & 1ull) != 0ull;
                    const unsigned long long oks = __ballot(ok) << (63 - hi); // bit 63 = the top frame
                    int run = (~oks == 0ull) ? 64 : __builtin_clzll(~oks);
                    run = min(run, hi + 1);
                    if (run > 0) {
                        if (lane <= hi && lane > hi - run) my_state = sp;
                        s -= 2 * run;
                        if (s < 0) s += L; // (the run's last move may leave state 0 / 1 downwards: the wrap of :692)
                        hi -= run;
                        todo &= (hi >= 0) ? ((2ull << hi) - 1ull) : 0ull;
                    }
                }
            }
        }
        int ph = p.blank, id = -1;
        if (lane < n) {
            const int o = t - it.pad_left; // :447-448 trim the boundary padding
            if (o >= 0 && o < it.nout) {
                if (my_state >= 1) {
                    const int q = ((my_state - 1) * inv_stride) >> 16;
                    if ((my_state - 1) - q * it.stride == 0 && q < it.nt) { ph = (q < 1024) ? stok[q] : tok[q]; id = it.tok0 + q; }
                }
                oph[it.out0 + o] = ph;
                oid[it.out0 + o] = id;
            }
        }
        if (do_rle) rle.chunk(ph, id, t0, n, lane);
    };
    int cg = nchunks - 1;
    if (NPRE == 3) {
        for (; cg >= 2; cg -= 3) { // no branches around the chunks of a group (see walk_item's fetch)
            walk_chunk(pre[0], preb[0], cg);
            walk_chunk(pre[NPRE > 1 ? 1 : 0], preb[NPRE > 1 ? 1 : 0], cg - 1);
            walk_chunk(pre[NPRE > 2 ? 2 : 0], preb[NPRE > 2 ? 2 : 0], cg - 2);
        }
        if (cg >= 0) walk_chunk(pre[0], preb[0], cg);
        if (cg >= 1) walk_chunk(pre[NPRE > 1 ? 1 : 0], preb[NPRE > 1 ? 1 : 0], cg - 1);
    } else if (NPRE == 2) {
        for (; cg >= 1; cg -= 2) {
            walk_chunk(pre[0], preb[0], cg);
            walk_chunk(pre[NPRE > 1 ? 1 : 0], preb[NPRE > 1 ? 1 : 0], cg - 1);
        }
        if (cg >= 0) walk_chunk(pre[0], preb[0], cg);
    } else {
        for (; cg >= 0; --cg) walk_chunk(pre[0], preb[0], cg);
    }
}

// Paths of more than 1024 states (k_dp_big): backpointers row-major, [frame][ng = ceil(L/16)] dwords, dword g =
// states 16g..16g+15, 2 bits each.  Same ballot-jump walk; every lane keeps the dword of its frame for the current
// state's group and reloads it when the walk enters another group (every <= 16 moves).
__device__ __forceinline__ void walk_item_big(const AlignArgs &a, const Item &it, int32_t *stok, int lane, RleRev &rle, bool do_rle)
{
    const DevParams &p = a.p;
    const int b = it.utt;
    int32_t *oph = a.frame_ph + (int64_t)b * a.Tmax;
    int32_t *oid = a.frame_idx + (int64_t)b * a.Tmax;
    const int32_t *tok = a.tokens + (int64_t)b * a.Smax + it.tok0;
    const int Ts = it.Ts, L = it.L;
    const int ng = (L + 15) >> 4;
    const uint32_t *bp = a.bp + it.bp_off;
    int s = it.final_state;
    const int nchunks = (Ts + 63) >> 6;
    for (int c = nchunks - 1; c >= 0; --c) {
        const int t0 = c * 64;
        const int t1 = min(Ts, t0 + 64);
        const int t = t0 + lane;
        const bool mine = t < t1;
        const bool can_move = mine && t > 0;
        int my_state = 0;
        int t_hi = t1 - 1;
        int g = s >> 4;
        uint32_t wd = mine ? bp[(int64_t)t * ng + g] : 0u;
        while (t_hi >= t0) {
            const uint32_t code = (wd >> (2 * (s & 15))) & 3u;
            const uint32_t k = (can_move && t <= t_hi) ? (max(code, 1u) - 1u) : 0u;
            const unsigned long long mv = __ballot(k != 0);
            if (mv == 0) {
                if (t <= t_hi) my_state = s;
                break;
            }
            const int jl = 63 - __builtin_clzll(mv);
            if (t <= t_hi && lane >= jl) my_state = s;
            s -= (int)__builtin_amdgcn_readlane((int)k, jl);
            if (s < 0) s += L; // python negative-index wrap (:692)
            t_hi = t0 + jl - 1;
            if ((s >> 4) != g) { g = s >> 4; wd = mine ? bp[(int64_t)t * ng + g] : 0u; }
        }
        int ph = p.blank, id = -1;
        if (mine) {
            const int o = t - it.pad_left;
            if (o >= 0 && o < it.nout) {
                if (my_state >= 1) {
                    const int q = (my_state - 1) / it.stride;
                    if ((my_state - 1) - q * it.stride == 0 && q < it.nt) { ph = tok[q]; id = it.tok0 + q; }
                }
                oph[it.out0 + o] = ph;
                oid[it.out0 + o] = id;
            }
        }
        if (do_rle) rle.chunk(ph, id, t0, t1 - t0, lane);
    }
    (void)stok;
}

// Which items a launch takes (AlignArgs::k2_sel): with one K1 kernel per class running side by side, the full-layout
// classes are walked right behind their own K1 kernel on its stream (the walks of the short classes then overlap the
// DPs of the long ones); window items wait for the sentinel reruns, which follow the join.
//   K2_ALL          every item (legacy order: one launch after all K1 kernels)
//   K2_FULL + R     full-layout DPs of class R that a K1 kernel has finished
//   K2_BIG          paths of more than 1024 states
//   K2_REST         everything the two above never take: non-DP items, window items, rerun items, items no K1 kernel took
//   K2_WIN          window items whose window kernel succeeded (launched behind the window kernels on their stream);
//   K2_REST_NOWIN   K2_REST without those
__device__ __forceinline__ bool k2_selected(int sel, const Item &it)
{
    if (sel == K2_ALL) return true;
    const bool plain_full = it.kind == ITEM_DP && it.win == 0 && it.final_state != FINAL_NOT_COMPUTED;
    // a window item whose window kernel ended above the sentinel (the others are rerun with the full layout: win < 0)
    const bool window_done = it.kind == ITEM_DP && it.win > 0 && it.final_state != FINAL_NOT_COMPUTED;
    if (sel == K2_WIN) return window_done;
    if ((sel & 0xff) == K2_NARROW) // the merged narrow kernel's items: its window classes and the full-layout classes of its mask
        return window_done || (plain_full && it.L <= 256 && (((unsigned)sel >> 8) & r_class_bit(r_class_for_L(it.L))) != 0u);
    if (sel == K2_REST_NOWIN) return !plain_full && !window_done;
    if (sel == K2_REST) return !plain_full;
    if (!plain_full) return false;
    if (sel == K2_BIG) return it.L > 1024;
    return it.L <= 1024 && r_class_for_L(it.L) == (sel - K2_FULL);
}

__device__ __forceinline__ void wave_sync_global()
{
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
}

// Two builds of the walk kernel share the items: the NARROW one (window Rw <= 4, full layout R <= 4: the classes a big
// batch of short utterances lands in) keeps 4 waves per SIMD so that a 4096-utterance batch is resident at once; the
// WIDE one (Rw = 6 / 8, R >= 6, paths beyond 1024 states: long utterances, few of them) may use the whole register
// file -- its staged walks hold two 64-frame chunks of 2R lane masks per lane, and with the 128-register budget they
// spilled into scratch memory inside the step loop (0.65 ms for the R = 8 class of the mixed-length shard).
__device__ __forceinline__ bool k2_is_wide(const Item &it)
{
    if (it.kind != ITEM_DP) return false;
    if (it.win > 0) return it.win > 4;
    return it.L > 1024 || r_class_for_L(it.L) >= 6;
}

template <bool WIDE>
__device__ __forceinline__ void backtrace_body(const AlignArgs &a, uint32_t *sbp, int32_t *stok)
{
    const int lane = threadIdx.x & 63;
    const DevParams &p = a.p;
    const int n_items = a.counters[0];
    // the walk is one short dependent chain per step; beside the K1 kernels of other classes (whose DP waves run at
    // raised priority) it would otherwise only get the issue slots they leave
    __builtin_amdgcn_s_setprio(3);
    const bool fused = a.k2_fused_rle != 0; // every utterance is ONE item: this kernel also produces its tuples (K3a)
    for (int i = blockIdx.x; i < n_items; i += gridDim.x) {
        const Item it = a.items[i];
        if (!k2_selected(a.k2_sel, it)) continue;
        if (k2_is_wide(it) != WIDE) continue;
        const int b = it.utt;
        int32_t *oph = a.frame_ph + (int64_t)b * a.Tmax;
        int32_t *oid = a.frame_idx + (int64_t)b * a.Tmax;
        const int32_t *tok = a.tokens + (int64_t)b * a.Smax + it.tok0;
        bool filled = false;
        if (it.kind == ITEM_FILL_BLANK) {
            for (int t = lane; t < it.nout; t += 64) { oph[it.out0 + t] = p.blank; oid[it.out0 + t] = -1; }
            filled = true;
        } else if (it.kind == ITEM_FILL_PROP) { // forced_alignment.py:170-172
            for (int t = lane; t < it.nout; t += 64) {
                const int fi = (int)(((int64_t)t * it.nt) / it.nout);
                oph[it.out0 + t] = tok[fi]; oid[it.out0 + t] = it.tok0 + fi;
            }
            filled = true;
        } else if (it.kind == ITEM_FILL_SIL) { // forced_alignment.py:382-397
            // frames per SIL token of the WHOLE silence segment (it.Ts frames); only the first it.nout of them are
            // written when the concatenation is cut off at T (:465-467)
            const double fps = (it.nt > 0) ? (double)it.Ts / (double)it.nt : 0.0;
            for (int t = lane; t < it.nout; t += 64) {
                int id = -1;
                if (it.nt > 0) {
                    // the k with int(k*fps) <= t < int((k+1)*fps), i.e. k*fps < t+1 <= (k+1)*fps: k = ceil((t+1)/fps) - 1
                    // (ranges are disjoint; with more SIL tokens than frames most of them are empty); the candidates
                    // around that estimate are tested with the reference's own double products
                    const int k = (int)__builtin_ceil((double)(t + 1) / fps) - 1;
                    for (int kk = max(0, k - 1); kk <= min(it.nt - 1, k + 1); ++kk) {
                        const int f0 = (int)((double)kk * fps), f1 = (int)((double)(kk + 1) * fps);
                        if (t >= f0 && t < f1) id = it.tok0 + kk;
                    }
                }
                oph[it.out0 + t] = p.sil; oid[it.out0 + t] = id;
            }
            filled = true;
        } else if (it.kind != ITEM_DP) {
            continue;
        } else if (it.final_state < 0) { // no K1 kernel took this item: its class was missing from the caller's class hint
            for (int t = lane; t < it.nout; t += 64) { oph[it.out0 + t] = p.blank; oid[it.out0 + t] = -1; }
            if (lane == 0 && a.status[b] == BFA_ITEM_OK) a.status[b] = BFA_ITEM_BAD_HINT;
            filled = true;
        }
        if (filled) {
            if (fused) { wave_sync_global(); assort_utterance<1>(a, b, lane); } // the forward pass over what was just written
            continue;
        }
        RleRev rle;
        rle.init(a, b, it.Ts);
        const bool do_rle = fused;
        if (it.win > 0) {
            if constexpr (!WIDE) {
                switch (it.win) {
                case 1: if (WIN_SSTORE) walk_item_mask<1, true>(a, it, sbp, stok, lane, rle, do_rle); else walk_item<1, true>(a, it, sbp, stok, lane, rle, do_rle); break;
                case 2: if (WIN_SSTORE) walk_item_mask<2, true>(a, it, sbp, stok, lane, rle, do_rle); else walk_item<2, true>(a, it, sbp, stok, lane, rle, do_rle); break;
                case 3: if (WIN_SSTORE) walk_item_mask<3, true>(a, it, sbp, stok, lane, rle, do_rle); else walk_item<3, true>(a, it, sbp, stok, lane, rle, do_rle); break;
                default: if (WIN_SSTORE) walk_item_mask<4, true>(a, it, sbp, stok, lane, rle, do_rle); else walk_item<4, true>(a, it, sbp, stok, lane, rle, do_rle); break;
                }
            } else {
                if (it.win == 6) { if (WIN_SSTORE) walk_item_mask<6, true>(a, it, sbp, stok, lane, rle, do_rle); }
                else { if (WIN_SSTORE) walk_item_mask<8, true>(a, it, sbp, stok, lane, rle, do_rle); }
            }
        } else {
            if constexpr (!WIDE) {
                if (it.split == 2) walk_item_mask<4, false, 2>(a, it, sbp, stok, lane, rle, do_rle); // (BFA_SPLIT_R4 builds)
                else switch (r_class_for_L(it.L)) {
                case 2: walk_item<2, false>(a, it, sbp, stok, lane, rle, do_rle); break;
                case 3: walk_item<3, false>(a, it, sbp, stok, lane, rle, do_rle); break;
                default: walk_item<4, false>(a, it, sbp, stok, lane, rle, do_rle); break;
                }
            } else if (it.split == 2) { // K1 split the DP over two consumer waves: per-frame lane masks (bfa_dp5.inc)
                switch (r_class_for_L(it.L)) {
                case 6: walk_item_mask<6, false, 2>(a, it, sbp, stok, lane, rle, do_rle); break;
                case 8: walk_item_mask<8, false, 2>(a, it, sbp, stok, lane, rle, do_rle); break;
                case 12: walk_item_mask<12, false, 2>(a, it, sbp, stok, lane, rle, do_rle); break;
                default: walk_item_mask<16, false, 2>(a, it, sbp, stok, lane, rle, do_rle); break;
                }
            } else if (it.split == 4) { // four consumer waves (BFA_NC8 = 4)
                if (r_class_for_L(it.L) == 8) walk_item_mask<8, false, 4>(a, it, sbp, stok, lane, rle, do_rle);
                else walk_item_mask<16, false, 4>(a, it, sbp, stok, lane, rle, do_rle);
            } else {
                switch (it.L > 1024 ? 0 : r_class_for_L(it.L)) {
                case 6: walk_item<6, false>(a, it, sbp, stok, lane, rle, do_rle); break;
                case 8: walk_item<8, false>(a, it, sbp, stok, lane, rle, do_rle); break;
                case 12: walk_item<12, false>(a, it, sbp, stok, lane, rle, do_rle); break;
                case 16: walk_item<16, false>(a, it, sbp, stok, lane, rle, do_rle); break;
                default: walk_item_big(a, it, stok, lane, rle, do_rle); break;
                }
            }
        }
        if (fused) {
            rle.finish(a, b, lane);
            for (int t = it.Ts + lane; t < a.Tmax; t += 64) { oph[t] = p.blank; oid[t] = -1; } // beyond the utterance
        }
    }
}

__global__ __launch_bounds__(64) __attribute__((amdgpu_waves_per_eu(4, 8))) void k_backtrace(AlignArgs a)
{
    __shared__ __attribute__((aligned(16))) uint32_t sbp[18 * 64]; // one chunk: <= 1024 dwords (dword layouts) / 9 qwords per lane (staged masks)
    __shared__ int32_t stok[1024];    // the item's tokens (nt <= L <= 1024)
    backtrace_body<false>(a, sbp, stok);
}

__global__ __launch_bounds__(64) void k_backtrace_wide(AlignArgs a)
{
    __shared__ __attribute__((aligned(16))) uint32_t sbp[2 * 33 * 64]; // a chunk's 2R lane masks per frame, R <= 16 (+1 qword: bank spread)
    __shared__ int32_t stok[1024];
    backtrace_body<true>(a, sbp, stok);
}

} // namespace bfa

// `wide` : bit 0 = the narrow kernel may find items, bit 1 = the wide one may (see k2_is_wide)
extern "C" void bfa_launch_backtrace(const bfa::AlignArgs *args, int grid, hipStream_t stream, int wide)
{
    if (wide & 1) hipLaunchKernelGGL(bfa::k_backtrace, dim3(grid), dim3(64), 0, stream, *args);
    if (wide & 2) hipLaunchKernelGGL(bfa::k_backtrace_wide, dim3(grid), dim3(64), 0, stream, *args);
}

// one selection of items (see k2_selected); `fused` = also produce the tuples (no K3a launch follows)
extern "C" void bfa_launch_backtrace_sel(const bfa::AlignArgs *args, int sel, int fused, int grid, hipStream_t stream, int wide)
{
    bfa::AlignArgs a = *args;
    a.k2_sel = sel;
    a.k2_fused_rle = fused;
    bfa_launch_backtrace(&a, grid, stream, wide);
}
