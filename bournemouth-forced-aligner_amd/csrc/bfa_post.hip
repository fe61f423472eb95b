// bfa_post.hip -- the post-DP boundary stages of extract_timestamps_from_segment_batch
// (core.py:925-931): ensure_target_coverage with ensure_completeness=False (core.py:488-513,660)
// and extend_soft_boundaries_func (core.py:682-809).  One wavefront per utterance, one lane per
// tuple; each of the four extension passes only reads neighbour fields that the same pass does not
// write, so a pass is data-parallel and passes are separated by a wave-level LDS sync.
#include <hip/hip_runtime.h>

#include "bfa_math.hpp"
#include "bfa_types.hpp"

#pragma clang fp contract(off)

namespace bfa {


__device__ __forceinline__ void post_sync()
{
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}

// -DBFA_POST_TIMES (A/B builds, tools/post_stamps.py): cycles per phase of k_postconf, summed over the utterances of a launch
#ifdef BFA_POST_TIMES
__device__ unsigned long long g_post_times[16];
#define POST_STAMP(k) do { const long long t1_ = clock64(); if (threadIdx.x == 0) atomicAdd(&g_post_times[k], (unsigned long long)(t1_ - t0_)); t0_ = clock64(); } while (0)
#else
#define POST_STAMP(k) do { } while (0)
#endif

struct PostArgs {
    const float *logp;
    float *row_stats; // != nullptr: logp is raw logits (AlignArgs::row_stats)
    int64_t strideB, strideT;
    int32_t B, Tmax, C;
    const int32_t *S_len;
    bfa_segment *segs;
    int32_t seg_cap;
    int32_t *seg_count;
    int32_t extend;
    double th1, th2;
};

// the four passes for tuple i on an array `t` (LDS or global)
template <typename Arr, bool RAW>
__device__ __forceinline__ void extend_pass(int pass, int i, int n, Arr &t, const LpView<RAW> &lp, int Tpad, int C,
                                            double th1, double th2, const double *mean)
{
    const int ph = t[i].phoneme, s = t[i].start, e = t[i].end;
    if (s >= Tpad || ph >= C) return; // :719,:740,:760,:784
    auto P = [&](int f) -> double { return (double)exp_cr(lp.at(f, ph)); };
    const int d = e - s;
    if (pass == 1) { // :717-735
        int min_start = (int)((double)s - (double)d * 10.0);
        if (min_start < 0) min_start = 0;
        if (i > 0) { int v = t[i - 1].end + 10; if (v > s) v = s; if (v > min_start) min_start = v; }
        double thr = mean[i] * th1; if (thr > th1) thr = th1;
        int ns = s;
        for (int f = s - 1; f >= min_start; --f) { if (P(f) >= thr) ns = f; else break; }
        t[i].start = ns;
    } else if (pass == 2) { // :738-755
        int max_end = (int)((double)e + (double)d * 10.0);
        if (max_end > Tpad) max_end = Tpad;
        if (i + 1 < n) { int v = t[i + 1].start - 10; if (v > e) v = e; if (v < max_end) max_end = v; }
        double thr = mean[i] * th1; if (thr > th1) thr = th1;
        int ne = e;
        for (int f = e; f < max_end; ++f) { if (P(f) >= thr) ne = f + 1; else break; }
        t[i].end = ne;
    } else if (pass == 3) { // :758-778
        int min_start = 0;
        if (i > 0) min_start = t[i - 1].end;
        if (s <= min_start) return;
        int ns = s;
        for (int f = s - 1; f >= min_start; --f) { if (P(f) >= th2) ns = f; else break; }
        t[i].start = ns;
    } else { // :782-805
        int max_end = (int)((double)e + (double)d * 10.0);
        if (max_end > Tpad) max_end = Tpad;
        if (i + 1 < n) { const int v = t[i + 1].start; if (v < max_end) max_end = v; }
        int ne = e;
        for (int f = e; f < max_end; ++f) { if (P(f) >= th2) ne = f + 1; else break; }
        t[i].end = ne;
    }
}

template <bool RAW>
__device__ __forceinline__ double seg_mean(const bfa_segment &g, const LpView<RAW> &lp, int Tpad, int C)
{
    // :709-714 ; float32 mean of the strided column slice: torch's cascade sum (bfa_math.hpp: cascade_sum_f32), then / n in float32
    if (g.start < Tpad && g.phoneme < C && g.start < g.end) {
        const int ee = g.end > Tpad ? Tpad : g.end;
        const int n = ee - g.start;
        constexpr int U = 8; // independent (clamped) loads in flight: one element per 268-byte row (two rows of four)
        float x[U];
        int have = -1; // first element of the cached group
        const float s = cascade_sum_f32(n, [&](int e) -> float {
            const int g0 = e & ~(U - 1);
            if (g0 != have) { lp.template at_n<U>(g.start + g0, ee - 1, g.phoneme, x); have = g0; }
            float v = x[0];
#pragma unroll
            for (int u = 1; u < U; ++u) if ((e & (U - 1)) == u) v = x[u];
            return exp_cr(v);
        });
        return (double)(s / (float)n);
    }
    return 0.001;
}

// BIG = false: the utterance's tuples and their means live in LDS (24 bytes per tuple slot: seg_cap <= POST_LDS_CAP).
// BIG = true (more slots than that -- paths of thousands of tokens, or seg_cap = Tmax + 1 of a long recording): the tuples
// stay where they are, in the caller's array, and only the MEANS go through LDS, POST_CHUNK tuples at a time.  The means
// belong to the tuples as the DP left them (core.py:709-714 runs before the passes) and passes 1 and 2 both need them, so
// the chunks are interleaved: means and pass 1 of chunk k+1, THEN pass 2 of chunk k -- pass 1 only writes starts and reads
// the end of the tuple before (which pass 2 of that chunk has not touched yet), pass 2 only writes ends and reads the start
// of the tuple after (whose pass 1 is done).  Passes 3 and 4 need no means: plain sweeps.
constexpr int POST_LDS_CAP = 6500;
constexpr int POST_CHUNK = 2048;

template <bool RAW, bool BIG>
__global__ __launch_bounds__(64) void k_postprocess(PostArgs a)
{
    extern __shared__ __attribute__((aligned(16))) unsigned char dyn_lds[];
    double *smean = (double *)dyn_lds;                                   // [seg_cap] (BIG: [2][POST_CHUNK])
    const int lane = threadIdx.x & 63;
    const double th1 = a.th1, th2 = a.th2;
    for (int b = blockIdx.x; b < a.B; b += gridDim.x) {
        bfa_segment *sg = a.segs + (int64_t)b * a.seg_cap;
        bfa_segment *st = BIG ? sg : (bfa_segment *)(dyn_lds + (size_t)a.seg_cap * 8);  // [seg_cap]
        const LpView<RAW> lp{a.logp + (int64_t)b * a.strideB, a.strideT,
                             RAW ? a.row_stats + 2 * (int64_t)b * a.Tmax : nullptr, a.C};
        int n = a.seg_count[b];
        if (n > a.seg_cap) n = a.seg_cap;
        const int S = a.S_len[b];
        // ---- ensure_target_coverage (default): drop idx == -1 or idx >= S (core.py:488-513)
        // (BIG compacts in place: slot m + rank <= base + lane, and the stores of a batch depend on its loads)
        int m = 0;
        for (int base = 0; base < n; base += 64) {
            const int i = base + lane;
            bfa_segment g;
            bool keep = false;
            if (i < n) { g = sg[i]; keep = (g.target_idx != -1 && g.target_idx < S); }
            const unsigned long long km = __ballot(keep);
            if (keep) st[m + __builtin_popcountll(km & ((1ull << lane) - 1ull))] = g;
            m += __builtin_popcountll(km);
        }
        post_sync();
        // stable sort by start (core.py:660); assort output is already ordered, so this is a check
        int unsorted = 0;
        for (int i = lane; i + 1 < m; i += 64) if (st[i].start > st[i + 1].start) unsorted = 1;
        if (__any(unsorted)) {
            if (lane == 0) {
                for (int i = 1; i < m; ++i) {
                    const bfa_segment key = st[i];
                    int j = i - 1;
                    while (j >= 0 && st[j].start > key.start) { st[j + 1] = st[j]; --j; }
                    st[j + 1] = key;
                }
            }
            post_sync();
        }
        if (a.extend) {
            if constexpr (!BIG) {
                for (int i = lane; i < m; i += 64) smean[i] = seg_mean(st[i], lp, a.Tmax, a.C);
                post_sync();
                for (int pass = 1; pass <= 4; ++pass) {
                    for (int i = lane; i < m; i += 64) extend_pass(pass, i, m, st, lp, a.Tmax, a.C, th1, th2, smean);
                    post_sync();
                }
            } else {
                const int n_chunks = (m + POST_CHUNK - 1) / POST_CHUNK;
                for (int k = 0; k <= n_chunks; ++k) {
                    if (k < n_chunks) { // means + pass 1 of chunk k
                        const int c0 = k * POST_CHUNK, c1 = min(m, c0 + POST_CHUNK);
                        double *mk = smean + (k & 1) * POST_CHUNK - c0; // indexed by the tuple
                        for (int i = c0 + lane; i < c1; i += 64) mk[i] = seg_mean(st[i], lp, a.Tmax, a.C);
                        post_sync();
                        for (int i = c0 + lane; i < c1; i += 64) extend_pass(1, i, m, st, lp, a.Tmax, a.C, th1, th2, mk);
                        post_sync();
                    }
                    if (k > 0) { // pass 2 of chunk k - 1
                        const int c0 = (k - 1) * POST_CHUNK, c1 = min(m, c0 + POST_CHUNK);
                        const double *mk = smean + ((k - 1) & 1) * POST_CHUNK - c0;
                        for (int i = c0 + lane; i < c1; i += 64) extend_pass(2, i, m, st, lp, a.Tmax, a.C, th1, th2, mk);
                        post_sync();
                    }
                }
                for (int pass = 3; pass <= 4; ++pass) {
                    for (int i = lane; i < m; i += 64) extend_pass(pass, i, m, st, lp, a.Tmax, a.C, th1, th2, smean);
                    post_sync();
                }
            }
        }
        if constexpr (!BIG) for (int i = lane; i < m; i += 64) sg[i] = st[i];
        if (lane == 0) a.seg_count[b] = m;
        post_sync();
    }
}


// =====================================================================================================================
// k_postconf: the post-DP stages AND the confidence pass of one utterance with the probabilities they look at STAGED IN
// LDS.  Both stages read exp(log_prob[f, phoneme]) for the frames of a tuple and a few frames around it -- one element
// per 268-byte row.  Written tuple-per-lane against global memory (k_postprocess, k_conf) every probe of the four
// extension passes is a dependent load, and a step costs a full memory round trip: 0.19 + 0.10 ms per 4096-utterance
// batch with 12-40-frame silence tuples (profiles/r03_realtext_timeline_before.txt), a latency chain of ~100 round trips
// per wavefront.  Here the wave first loads, for every tuple i, the cells of column phoneme_i over the frames
// [start_i - K, end_i + K) -- a flat list of cells over all tuples, 64 lanes x 8 independent loads at a time, four or five
// round trips for the whole utterance -- converts them once (exp_cr, all lanes busy) and leaves the float32
// probabilities in LDS; the passes, the segment means and the confidences then run tuple-per-lane at LDS speed.  A
// probe outside a tuple's staged window (an extension running further than K frames) falls back to the global read, so
// the result does not depend on K.  Arithmetic and order of operations are those of k_postprocess / k_conf.
// =====================================================================================================================
struct PostConfArgs {
    const float *logp;
    float *row_stats;
    int64_t strideB, strideT;
    int32_t B, Tmax, C;
    const int32_t *S_len;   // do_post
    bfa_segment *segs;
    int32_t seg_cap;
    int32_t *seg_count;
    int32_t do_post, extend, do_conf;
    double th1, th2;
    const int32_t *T_rows;  // do_conf (nullptr: Tmax)
    float *conf;
    int32_t *status;
    int32_t cap_cells;      // floats of LDS for the staged probabilities
};

#ifndef BFA_STAGE_K
#define BFA_STAGE_K 4
#endif
#ifndef BFA_POST_NW4_MAX_BATCH
#define BFA_POST_NW4_MAX_BATCH 512
#endif
constexpr int STAGE_K = BFA_STAGE_K;
#ifndef BFA_POST_WIDE_REACH
#define BFA_POST_WIDE_REACH 192
#endif
constexpr int POST_WIDE_REACH = BFA_POST_WIDE_REACH; // frames a wide window reaches beyond its tuple (k_postconf, rounds 1 / 2) // frames staged on either side of a tuple (probes beyond fall back to memory)

template <bool RAW>
struct StagedProb {
    const LpView<RAW> &lp;
    const float *sp;
    const int32_t *off, *wlo, *whi;
    // exp(log_prob[f, ph]) of tuple i's column: from the staged window, else from memory
    __device__ __forceinline__ float at(int i, int f, int ph) const
    {
        const int lo = wlo[i];
        if (f >= lo && f < whi[i]) return sp[off[i] + (f - lo)];
        return exp_cr(lp.at(f, ph));
    }
};

// NW wavefronts per utterance (round 6).  One wave per utterance is a single instruction stream of ~10^5 cycles for a 30-s
// segment with 150 tuples: in a call of the reference's sixteen utterances the two heads' k_postconf launches were 0.25-0.29 ms
// of a 1.15-ms call, every SIMD but sixteen idle (profiles/r06_latency_realtext_timeline_b16_after.txt).  With NW = 4 the
// staging (cells, exponentials), the means, the passes and the confidences are shared out over 256 lanes, the prefix sums of the
// window layout stay with wave 0, and the wave-level syncs become workgroup barriers.  Full batches keep NW = 1.
template <bool RAW, int NW>
__global__ __launch_bounds__(64 * NW) __attribute__((amdgpu_waves_per_eu(4))) void k_postconf(PostConfArgs a)
{
    __shared__ int s_flag, s_tot, s_m;
    constexpr int NT = 64 * NW;
    const int tid = threadIdx.x, wave = (int)(threadIdx.x >> 6);
    auto post_sync = [&]() { if (NW == 1) bfa::post_sync(); else __syncthreads(); };
    // does the predicate hold for any tuple of the utterance? (all threads call this)
    auto any_wg = [&](int x) -> bool {
        if (NW == 1) return __any(x) != 0;
        if (tid == 0) s_flag = 0;
        __syncthreads();
        if (x) s_flag = 1;
        __syncthreads();
        const bool r = s_flag != 0;
        __syncthreads();
        return r;
    };
    extern __shared__ __attribute__((aligned(16))) unsigned char dyn_lds[];
    // [seg_cap] doubles | [seg_cap] tuples | 4 x [seg_cap] ints | [cap_cells] floats
    double *smean = (double *)dyn_lds;
    bfa_segment *st = (bfa_segment *)(dyn_lds + (size_t)a.seg_cap * 8);
    int32_t *off = (int32_t *)(dyn_lds + (size_t)a.seg_cap * (8 + sizeof(bfa_segment)));
    int32_t *wlo = off + a.seg_cap, *whi = wlo + a.seg_cap, *wfl = whi + a.seg_cap;
    float *sp = (float *)(wfl + a.seg_cap);
    const int lane = threadIdx.x & 63;
    const double th1 = a.th1, th2 = a.th2;
    for (int b = blockIdx.x; b < a.B; b += gridDim.x) {
        bfa_segment *sg = a.segs + (int64_t)b * a.seg_cap;
        const LpView<RAW> lp{a.logp + (int64_t)b * a.strideB, a.strideT,
                             RAW ? a.row_stats + 2 * (int64_t)b * a.Tmax : nullptr, a.C};
#ifdef BFA_POST_TIMES
        long long t0_ = clock64();
#endif
        int n = a.seg_count[b];
        if (n > a.seg_cap) n = a.seg_cap;
        int m = n;
        post_sync(); // the previous utterance's readers of the LDS arrays are done
        if (a.do_post) {
            // ---- ensure_target_coverage (default): drop idx == -1 or idx >= S (core.py:488-513)
            const int S = a.S_len[b];
            m = 0;
            if (wave == 0) {
                for (int base = 0; base < n; base += 64) {
                    const int i = base + lane;
                    bfa_segment g;
                    bool keep = false;
                    if (i < n) { g = sg[i]; keep = (g.target_idx != -1 && g.target_idx < S); }
                    const unsigned long long km = __ballot(keep);
                    if (keep) st[m + __builtin_popcountll(km & ((1ull << lane) - 1ull))] = g;
                    m += __builtin_popcountll(km);
                }
                if (NW > 1 && lane == 0) s_m = m;
            }
            post_sync();
            if (NW > 1) m = s_m;
            // stable sort by start (core.py:660); assort output is already ordered, so this is a check
            int unsorted = 0;
            for (int i = tid; i + 1 < m; i += NT) if (st[i].start > st[i + 1].start) unsorted = 1;
            if (any_wg(unsorted)) {
                if (tid == 0) {
                    for (int i = 1; i < m; ++i) {
                        const bfa_segment key = st[i];
                        int j = i - 1;
                        while (j >= 0 && st[j].start > key.start) { st[j + 1] = st[j]; --j; }
                        st[j + 1] = key;
                    }
                }
                post_sync();
            }
        } else {
            for (int i = tid; i < m; i += NT) st[i] = sg[i];
            post_sync();
        }
        POST_STAMP(0);
        const int Tpad = a.Tmax;                      // the soft-boundary stage works on the padded rows (core.py:705)
        int Tc = a.T_rows ? a.T_rows[b] : a.Tmax;     // the confidence pass on log_probs.shape[0] as the caller passes it
        if (Tc > a.Tmax) Tc = a.Tmax;
        const bool staging = (a.do_post && a.extend) || a.do_conf;
        const int K = (a.do_post && a.extend) ? STAGE_K : 0;
        // ---- the windows.  mode 0: [start - K, end + K) clipped to the rows, K = 0 without the extension passes.  mode 1 (the
        // left walks of passes 1 and 3, see below): a tuple whose flag bit 0 is set reaches up to W frames further to the left,
        // but never beyond the tuple before it (the passes cannot: core.py:724-728,765); the LAST tuple with bit 1 also to the
        // right (pass 2 moves no other tuple's end: core.py:745-747).  mode 2 (the right walks of pass 4 and the confidences):
        // from the tuple's start as it is now to end + K, with bit 1 up to W frames further but never beyond the start of the
        // tuple after it as it is now (core.py:791-793).  Tuples of an alignment are disjoint runs: every mode needs at most
        // Tmax + 2 K seg_cap cells.  Returns the number of cells.
        auto assign_windows = [&](int mode, int W) -> int {
            int tot = 0;
            if (NW > 1 && wave != 0) { __syncthreads(); return s_tot; } // (the running offsets are one wave's)
            for (int base = 0; base < m; base += 64) {
                const int i = base + lane;
                int len = 0, lo = 0;
                if (i < m) {
                    const bfa_segment g = st[i];
                    const int s0 = max(0, g.start), e0 = min(a.Tmax, g.end);
                    if (g.phoneme >= 0 && g.phoneme < a.C && s0 < a.Tmax) {
                        const int e1 = max(e0, s0 + 1);
                        int xl = (mode == 2) ? 0 : K, xr = K;
                        if (mode != 0) {
                            const int fl = wfl[i];
                            if (mode == 1 && (fl & 1)) { const int limL = (i > 0) ? min(s0, max(0, st[i - 1].end)) : 0; xl = max(K, min(W, s0 - limL)); }
                            if ((fl & 2) && (mode == 2 || i == m - 1)) {
                                const int limR = (i + 1 < m) ? max(e1, min(a.Tmax, st[i + 1].start)) : a.Tmax;
                                xr = max(K, min(W, limR - e1));
                            }
                        }
                        lo = max(0, s0 - xl);
                        const int hi = min(a.Tmax, e1 + xr);
                        len = hi - lo;
                    }
                }
                // exclusive prefix sum of len over the 64 lanes
                int incl = len;
#pragma unroll
                for (int d = 1; d < 64; d <<= 1) { const int v = __shfl_up(incl, d); if (lane >= d) incl += v; }
                if (i < m) { off[i] = tot + incl - len; wlo[i] = lo; whi[i] = lo + len; }
                tot += __shfl(incl, 63);
            }
            if (NW > 1) { if (lane == 0) s_tot = tot; __syncthreads(); }
            return tot;
        };
        // ---- stage: cell c belongs to the tuple i with off[i] <= c < off[i] + len_i (binary search), frame wlo[i] + (c - off[i])
        auto stage_cells = [&](int total) {
            constexpr int U = 8;
            // steps of the search below: the same for every lane and cell, so that the U searches of a lane run side by side
            // (one LDS round trip per step for all of them; with a trip count of its own per cell they ran one behind the other)
            int nstep = 0;
            while ((1 << nstep) < m) ++nstep;
            for (int c0 = wave * 64 * U; c0 < total; c0 += NT * U) {
                float x[U];
                float2 ms[U];
                {
                    int own[U];
#pragma unroll
                    for (int u = 0; u < U; ++u) own[u] = 0;
                    // last tuple with off <= c (empty windows share their successor's offset): binary lifting over [0, m)
                    for (int st_ = nstep - 1; st_ >= 0; --st_) {
#pragma unroll
                        for (int u = 0; u < U; ++u) {
                            const int cand = own[u] + (1 << st_);
                            if (cand < m && off[cand] <= min(c0 + u * 64 + lane, total - 1)) own[u] = cand;
                        }
                    }
                    // ALL loads of the round first -- 2 U of them in flight -- and only then the test for rows without statistics:
                    // with the test (and its call, which stores the row's statistics) behind each pair of loads the next pair
                    // could not be moved above it, and a round of U cells was U memory round trips one behind the other: 65 % of
                    // the wave's cycles on the real-text batch (tools/post_stamps.py, profiles/r06_postconf_stamps.txt)
#pragma unroll
                    for (int u = 0; u < U; ++u) {
                        const int f = wlo[own[u]] + (min(c0 + u * 64 + lane, total - 1) - off[own[u]]);
                        x[u] = lp.lp[(int64_t)f * lp.ld + st[own[u]].phoneme];
                        if (RAW) ms[u] = *(const float2 *)(lp.st + 2 * (int64_t)f);
                        else ms[u] = make_float2(0.0f, 0.0f); // (unused)
                    }
                }
                if (RAW) {
                    bool need = false;
#pragma unroll
                    for (int u = 0; u < U; ++u) need = need || (ms[u].x != ms[u].x || ms[u].y != ms[u].y);
                    if (__any(need)) {
                        // padded rows beyond the ones K0 left statistics for (rare): the round cell by cell, nothing kept in
                        // registers across the one-lane-per-row softmax calls
#pragma unroll 1
                        for (int u = 0; u < U; ++u) {
                            const int c = c0 + u * 64 + lane;
                            if (c >= total) break;
                            int o = 0;
                            for (int st_ = nstep - 1; st_ >= 0; --st_) { const int cand = o + (1 << st_); if (cand < m && off[cand] <= c) o = cand; }
                            const int f = wlo[o] + (c - off[o]);
                            sp[c] = exp_cr(lp.at(f, st[o].phoneme));
                        }
                        continue;
                    }
                }
#pragma unroll
                for (int u = 0; u < U; ++u) {
                    const float v = RAW ? ((x[u] - ms[u].x) - ms[u].y) : x[u];
                    if (c0 + u * 64 + lane < total) sp[min(c0 + u * 64 + lane, total - 1)] = exp_cr(v);
                }
            }
        };
        const StagedProb<RAW> pr{lp, sp, off, wlo, whi};
        const bool ext = a.do_post && a.extend;
        int total = 0, wide = 0;
        // Three rounds, ONE copy of the staging code (as a helper used from three places it was inlined three times: 150 vector
        // registers instead of 101, three waves per SIMD instead of four, the real-text step 7 % slower on one box --
        // profiles/r06_vs_r05.txt).  Round 0: windows of +-K frames, then the means and the wide test; round 1 (only when a walk
        // can leave its window): windows that reach to the left, then passes 1-3; round 2: to the right, then pass 4.
#pragma unroll 1
        for (int round = 0; round < 3; ++round) {
            if (staging && (round == 0 || wide)) {
                if (round) post_sync();
                // (POST_WIDE_REACH frames beyond the +-K margin at most: the walks of more than 32 frames are the whole wave's, 64
                // probes per step, and go on through memory at the same price; staging a window's full reach made every padded row
                // up to Tmax a cell of the last tuple -- rows whose statistics nobody has, eleven one-lane-per-row softmax calls per
                // wave on the C5 proxy at peak 3, profiles/r06_postconf_pmc_c5p3.txt)
                int W = POST_WIDE_REACH, tot = 0;
                for (;;) { // the widest reach whose cells fit the LDS (a probe beyond it reads memory)
                    tot = assign_windows(round, W);
                    if (tot <= a.cap_cells || W <= 1 || round == 0) break;
                    W >>= 1;
                    post_sync();
                }
                if (tot > a.cap_cells) { // more cells than the LDS holds (overlapping caller-made tuples): nothing is staged
                    post_sync();
                    for (int i = tid; i < m; i += NT) whi[i] = wlo[i];
                    tot = 0;
                }
                total = tot;
                post_sync();
                POST_STAMP(1);
                stage_cells(total);
                post_sync();
                POST_STAMP(2);
            }
            if (!ext) break;
            if (round == 0) {
                // ---- segment means (core.py:709-714): torch's float32 cascade sum of the strided slice, / n in float32
                for (int i = tid; i < m; i += NT) {
                    const bfa_segment g = st[i];
                    double mean = 0.001;
                    if (g.start < Tpad && g.phoneme < a.C && g.start < g.end) {
                        const int ee = g.end > Tpad ? Tpad : g.end;
                        const int n = ee - g.start;
                        const float sum = cascade_sum_f32(n, [&](int e) -> float { return pr.at(i, g.start + e, g.phoneme); });
                        mean = (double)(sum / (float)n);
                    }
                    smean[i] = mean;
                }
                post_sync();
                POST_STAMP(3);
                // ---- Can a walk of the passes below leave its tuple's staged window?  Only if every staged margin cell on that side
                // is at or above the smallest threshold any pass compares it with (the walks stop at the first cell below theirs) and
                // the window does not already reach the neighbour.  On sharp posteriors no tuple qualifies and this costs 2 K LDS
                // reads per tuple.  On soft ones (logits N(0,1) + 3 on the planted class: every class keeps ~1e-2, above both
                // thresholds) EVERY walk runs to its limit -- through memory that was one dependent 64-byte-sector read, a row
                // statistics read and a float64 exponential per frame and lane: 12.5 ms per head on the C5 proxy against 0.8 ms on
                // sharp posteriors (profiles/r06_softness_before.jsonl).  Such tuples get windows that reach as far as their passes
                // can (the gap to the neighbouring tuple): once to the LEFT for passes 1-3, once to the RIGHT -- from the starts as
                // pass 3 leaves them -- for pass 4 and the confidences, so that either round needs no more cells than the first (a
                // gap is staged for one neighbour at a time); the cells of the earlier rounds come from the L2, and the walks run at
                // LDS speed.  The result does not depend on any of this (StagedProb::at).
                if (total > 0) {
                    int anyw = 0;
                    for (int i = tid; i < m; i += NT) {
                        const bfa_segment g = st[i];
                        int fl = 0;
                        const int lo = wlo[i], hi = whi[i];
                        if (hi > lo) {
                            const int s0 = max(0, g.start), e1 = max(min(a.Tmax, g.end), s0 + 1);
                            double tmin = smean[i] * th1; if (tmin > th1) tmin = th1; if (th2 < tmin) tmin = th2;
                            const int limL = (i > 0) ? min(s0, max(0, st[i - 1].end)) : 0;
                            const int limR = (i + 1 < m) ? max(e1, min(a.Tmax, st[i + 1].start)) : a.Tmax;
                            if (lo > limL && s0 > lo) {
                                bool all = true;
                                for (int f = lo; f < s0; ++f) all = all && ((double)sp[off[i] + (f - lo)] >= tmin);
                                if (all) fl |= 1;
                            }
                            if (hi < limR && hi > e1) {
                                bool all = true;
                                for (int f = e1; f < hi; ++f) all = all && ((double)sp[off[i] + (f - lo)] >= tmin);
                                if (all) fl |= 2;
                            }
                        }
                        wfl[i] = fl;
                        anyw |= fl;
                    }
                    wide = any_wg(anyw) ? 1 : 0;
                }
                POST_STAMP(4);
            } else {
                // ---- the four passes (core.py:717-805); a pass only reads neighbour fields it does not write
                const int p_last = round == 1 ? 3 : 4;
                for (int pass = round == 1 ? 1 : 4; pass <= p_last; ++pass) {
                    // A walk is a scan for the first frame below the threshold, left of the start or right of the end, up to a
                    // limit.  Tuple per lane it is one dependent LDS probe per frame: fine for the usual few frames, but on soft
                    // posteriors -- or behind an utterance that ended at the sentinel, whose tuples can lie a thousand frames
                    // apart -- single lanes walked ~10^3 frames while their wave waited (C5 proxy at peak 3: k_postconf 1.7 + 3.1
                    // of a 7.9-ms call; the first sweep, profiles/r06_c5proxy_p3_timeline_before.txt, has 12.5 ms per head).  Walks of more than 32 frames are taken by the
                    // whole wave, 64 frames per step: one probe per lane, one ballot, the first failing lane ends the walk.
                    for (int base = wave * 64; base < m; base += NT) {
                        const int i = base + lane;
                        bool act = i < m;
                        int ph = 0, s = 0, e = 0;
                        if (act) { ph = st[i].phoneme; s = st[i].start; e = st[i].end; if (s >= Tpad || ph >= a.C) act = false; } // :719,:740,:760,:784
                        const int d = e - s;
                        int dir = (pass == 1 || pass == 3) ? -1 : 1; // -1: probes s - 1 down to `lim` (inclusive); +1: e up to `lim` (exclusive)
                        int lim = 0;
                        double thr = th2;
                        if (act) {
                            if (pass == 1) { // :717-735
                                int min_start = (int)((double)s - (double)d * 10.0);
                                if (min_start < 0) min_start = 0;
                                if (i > 0) { int v = st[i - 1].end + 10; if (v > s) v = s; if (v > min_start) min_start = v; }
                                thr = smean[i] * th1; if (thr > th1) thr = th1;
                                lim = min_start;
                            } else if (pass == 2) { // :738-755
                                int max_end = (int)((double)e + (double)d * 10.0);
                                if (max_end > Tpad) max_end = Tpad;
                                if (i + 1 < m) { int v = st[i + 1].start - 10; if (v > e) v = e; if (v < max_end) max_end = v; }
                                thr = smean[i] * th1; if (thr > th1) thr = th1;
                                lim = max_end;
                            } else if (pass == 3) { // :758-778
                                int min_start = 0;
                                if (i > 0) min_start = st[i - 1].end;
                                if (s <= min_start) act = false;
                                lim = min_start;
                            } else { // :782-805
                                int max_end = (int)((double)e + (double)d * 10.0);
                                if (max_end > Tpad) max_end = Tpad;
                                if (i + 1 < m) { const int v = st[i + 1].start; if (v < max_end) max_end = v; }
                                lim = max_end;
                            }
                        }
                        const int f0 = dir < 0 ? s - 1 : e;                       // first frame probed
                        const int len = act ? (dir < 0 ? f0 - lim + 1 : lim - f0) : 0; // frames the walk may probe
                        int res = dir < 0 ? s : e;
                        const bool lng = len > 32;
                        if (act && !lng) {
                            if (dir < 0) { for (int f = f0; f >= lim; --f) { if ((double)pr.at(i, f, ph) >= thr) res = f; else break; } }
                            else { for (int f = f0; f < lim; ++f) { if ((double)pr.at(i, f, ph) >= thr) res = f + 1; else break; } }
                        }
                        unsigned long long pend = __ballot(act && lng);
                        while (pend) {
                            const int src = __builtin_ctzll(pend);
                            pend &= pend - 1ull;
                            const int ci = __shfl(i, src), cph = __shfl(ph, src), cf0 = __shfl(f0, src), cn = __shfl(len, src), clim = __shfl(lim, src);
                            const long long tb = __builtin_bit_cast(long long, thr);
                            const unsigned tlo = (unsigned)__shfl((int)(tb & 0xffffffffll), src), thi = (unsigned)__shfl((int)(tb >> 32), src);
                            const double cthr = __builtin_bit_cast(double, (long long)(((unsigned long long)thi << 32) | tlo));
                            int r = clim; // nothing fails: the walk ends at its limit (left: start = lim, right: end = lim)
                            for (int c0 = 0; c0 < cn; c0 += 64) {
                                const int k = c0 + lane;
                                bool fail = false;
                                if (k < cn) fail = !((double)pr.at(ci, dir < 0 ? cf0 - k : cf0 + k, cph) >= cthr);
                                const unsigned long long fm = __ballot(fail);
                                if (fm) {
                                    const int kf = c0 + __builtin_ctzll(fm);
                                    r = dir < 0 ? cf0 - kf + 1 : cf0 + kf; // left: the frame behind the first failing one; right: the failing frame
                                    break;
                                }
                            }
                            if (lane == src) res = r;
                        }
                        if (act) { if (dir < 0) st[i].start = res; else st[i].end = res; }
                    }
                    post_sync();
                }
                POST_STAMP(5);
            }
        }
        if (a.do_post) {
            for (int i = tid; i < m; i += NT) sg[i] = st[i];
            if (tid == 0) a.seg_count[b] = m;
        }
        if (a.do_conf) {
            // ---- _calculate_confidences (utils.py:70-113) on the tuples as they are now
            float *cf = a.conf + (int64_t)b * a.seg_cap;
            for (int i = (m < 0 ? 0 : m) + tid; i < a.seg_cap; i += NT) cf[i] = 0.0f; // the caller's buffer needs no fill ahead of the call
            int bad = 0;
            // does any tuple read a cell that an earlier tuple has written through its 0-dim view?
            // (with the starts in non-decreasing order -- what an alignment's tuples are -- an earlier tuple can only share a later
            // one's cells if it STARTS on the same frame: the tuples right before i with that start are all there is to look at;
            // the all-pairs test, 11 % of the wave's cycles on 150-tuple utterances, is left for tuples in another order)
            int alias = 0, unord = 0;
            for (int i = tid; i + 1 < m; i += NT) if (max(0, st[i].start) > max(0, st[i + 1].start)) unord = 1;
            if (!any_wg(unord)) {
                for (int i = tid; i < m; i += NT) {
                    const int ph = st[i].phoneme;
                    const int s = max(0, st[i].start);
                    for (int k = i - 1; k >= 0 && max(0, st[k].start) == s; --k)
                        if (st[k].phoneme == ph) { alias = 1; break; }
                }
            } else {
                for (int i = tid; i < m; i += NT) {
                    const int ph = st[i].phoneme;
                    const int s = max(0, st[i].start), e = min(Tc, st[i].end);
                    for (int k = 0; k < i; ++k) {
                        if (st[k].phoneme != ph) continue;
                        const int sk = max(0, st[k].start);
                        if (sk == s || (sk >= s && sk < e)) { alias = 1; break; }
                    }
                }
            }
            POST_STAMP(6);
            alias = any_wg(alias) ? 1 : 0;
            if (!alias) {
                // Tuple per lane; a tuple of more than CONF_COOP frames is taken by the whole wave -- 64 probes at a time, then the
                // sum in frame order over the 64 values (the reference's float32 additions, :101-103, in its order).  On soft
                // posteriors the extension passes leave tuples of hundreds of frames, longer than their staged windows: one memory
                // probe per frame and lane, 41 % of the wave's cycles on the C5 proxy at peak 3 (profiles/r06_postconf_stamps.txt).
                constexpr int CONF_COOP = 96;
                for (int base = wave * 64; base < m; base += NT) {
                    const int i = base + lane;
                    bool ok = false;
                    int ph = 0, s = 0, e = 0;
                    float c = 0.0f;
                    if (i < m) {
                        ph = st[i].phoneme;
                        s = max(0, st[i].start); e = min(Tc, st[i].end); // :86-87
                        if (s >= Tc || ph < 0 || ph >= a.C) bad = 1; // IndexError at :89
                        else { ok = true; c = pr.at(i, s, ph); }
                    }
                    const bool lng = ok && (e - s) > CONF_COOP;
                    if (ok && !lng && s < e) {
                        const float half = c / 2.0f; // :95 (a fresh tensor: stays constant)
                        int good = 1;
                        float mx = 0.0f;
                        for (int f = s + 1; f < e; ++f) {
                            const float v = pr.at(i, f, ph);
                            mx = (f == s + 1) ? v : __builtin_fmaxf(mx, v);
                            if (v > half || v > 0.1f) { c = c + v; good++; } // :101-103
                        }
                        if (good > 1) {
                            c = c / (float)good; // :105 -- this also lands in probs[start, ph] ...
                            const float m2 = __builtin_fmaxf(c, mx); // ... so :107 sees the mean at `start`
                            if (c < m2 / 2.0f) c = m2;
                        }
                    }
                    unsigned long long pend = __ballot(lng);
                    while (pend) {
                        const int src = __builtin_ctzll(pend);
                        pend &= pend - 1ull;
                        // (wave-uniform values in scalar registers: the loops below are scalar loops, a value of lane j is one v_readlane)
                        const int ci = __builtin_amdgcn_readlane(i, src), cph = __builtin_amdgcn_readlane(ph, src);
                        const int cs = __builtin_amdgcn_readlane(s, src), ce = __builtin_amdgcn_readlane(e, src);
                        float cc = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, c), src));
                        const float half = cc / 2.0f;
                        int good = 1;
                        float mx = 0.0f;
                        for (int f0 = cs + 1; f0 < ce; f0 += 64) {
                            const int f = f0 + lane;
                            const float v = (f < ce) ? pr.at(ci, f, cph) : 0.0f;
                            const int nv = min(64, ce - f0);
                            for (int j = 0; j < nv; ++j) {
                                const float vj = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, v), j));
                                mx = (f0 + j == cs + 1) ? vj : __builtin_fmaxf(mx, vj);
                                if (vj > half || vj > 0.1f) { cc = cc + vj; good++; }
                            }
                        }
                        if (good > 1) {
                            cc = cc / (float)good;
                            const float m2 = __builtin_fmaxf(cc, mx);
                            if (cc < m2 / 2.0f) cc = m2;
                        }
                        if (lane == src) c = cc;
                    }
                    if (i < m) cf[i] = ok ? c : 0.0f;
                }
            } else if ((size_t)m * 5 <= (size_t)a.cap_cells * 4) {
                // exact serial replay with the write-through cells (rare: overlapping tuples of one phoneme); the staged
                // probabilities are not needed any more: their LDS holds the replay's cells
                post_sync();
                float *mval = sp;
                uint8_t *mflag = (uint8_t *)(sp + m);
                for (int i = tid; i < m; i += NT) mflag[i] = 0;
                post_sync();
                if (tid == 0) {
                    auto prob = [&](int f, int ph, int upto) -> float {
                        for (int k = upto; k >= 0; --k)
                            if (mflag[k] && st[k].phoneme == ph && max(0, st[k].start) == f) return mval[k];
                        return exp_cr(lp.at(f, ph));
                    };
                    for (int i = 0; i < m; ++i) {
                        const int ph = st[i].phoneme;
                        const int s = max(0, st[i].start), e = min(Tc, st[i].end);
                        if (s >= Tc || ph < 0 || ph >= a.C) { bad = 1; cf[i] = 0.0f; continue; }
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
                post_sync();
            } else {
                bad = 1;
            }
            POST_STAMP(7);
            bad = any_wg(bad) ? 1 : 0;
            if (tid == 0 && a.status) a.status[b] = bad ? BFA_ITEM_BAD_TOKEN : BFA_ITEM_OK;
        }
    }
}

// bytes of dynamic LDS k_postconf needs for these shapes, or 0 when it does not fit (the caller then takes the
// tuple-per-lane kernels): tuples of one alignment are disjoint runs, so Tmax + 2 K seg_cap cells hold every window
static size_t postconf_lds(int Tmax, int seg_cap, int *cap_cells)
{
    const size_t cells = (size_t)Tmax + 2 * (size_t)bfa::STAGE_K * (size_t)seg_cap + 64;
    const size_t bytes = (size_t)seg_cap * (8 + sizeof(bfa_segment) + 16) + cells * 4;
    if (bytes > 60 * 1024) return 0;
    *cap_cells = (int)cells;
    return bytes;
}

} // namespace bfa

extern "C" int bfa_launch_postprocess(const float *logp, float *row_stats, int64_t strideB, int64_t strideT, int B, int Tmax, int C,
                                      const int32_t *S_len, bfa_segment *segs, int seg_cap, int32_t *seg_count,
                                      int extend, double th1, double th2, void *stream_)
{
    using namespace bfa;
    PostArgs a;
    a.logp = logp; a.row_stats = row_stats; a.strideB = strideB; a.strideT = strideT; a.B = B; a.Tmax = Tmax; a.C = C; a.S_len = S_len;
    a.segs = segs; a.seg_cap = seg_cap; a.seg_count = seg_count; a.extend = extend; a.th1 = th1; a.th2 = th2;
    const unsigned grid = B < 65536 ? B : 65536;
    if (seg_cap > POST_LDS_CAP) { // tuples in place, means through LDS in chunks (k_postprocess<.., BIG>)
        const size_t lds = 2 * (size_t)POST_CHUNK * 8;
        if (row_stats) hipLaunchKernelGGL((k_postprocess<true, true>), dim3(grid), dim3(64), lds, (hipStream_t)stream_, a);
        else hipLaunchKernelGGL((k_postprocess<false, true>), dim3(grid), dim3(64), lds, (hipStream_t)stream_, a);
        return (int)hipGetLastError();
    }
    const size_t lds = (size_t)seg_cap * (8 + sizeof(bfa_segment));
    if (lds > 48 * 1024) { // beyond the default dynamic-LDS limit (seg_cap = Tmax + 1 with ignore_noise = False)
        (void)hipFuncSetAttribute((const void *)k_postprocess<true, false>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        (void)hipFuncSetAttribute((const void *)k_postprocess<false, false>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    }
    if (row_stats) hipLaunchKernelGGL((k_postprocess<true, false>), dim3(grid), dim3(64), lds, (hipStream_t)stream_, a);
    else hipLaunchKernelGGL((k_postprocess<false, false>), dim3(grid), dim3(64), lds, (hipStream_t)stream_, a);
    return (int)hipGetLastError();
}

#ifdef BFA_POST_TIMES
extern "C" __attribute__((visibility("default"))) int bfa_dbg_post_times(unsigned long long *out, int reset)
{
    int rc = (int)hipMemcpyFromSymbol(out, HIP_SYMBOL(bfa::g_post_times), sizeof(unsigned long long) * 16);
    if (reset) { unsigned long long z[16] = {}; rc |= (int)hipMemcpyToSymbol(HIP_SYMBOL(bfa::g_post_times), z, sizeof(z)); }
    return rc;
}
#endif

// postprocess and / or confidences of a batch through k_postconf; returns -1 when the shapes do not fit its LDS staging
extern "C" int bfa_launch_postconf(const float *logp, float *row_stats, int64_t strideB, int64_t strideT, int B, int Tmax, int C,
                                   const int32_t *S_len, bfa_segment *segs, int seg_cap, int32_t *seg_count, int do_post,
                                   int extend, double th1, double th2, int do_conf, const int32_t *T_rows, float *conf,
                                   int32_t *status, void *stream_)
{
    using namespace bfa;
    int cap_cells = 0;
    const size_t lds = postconf_lds(Tmax, seg_cap, &cap_cells);
    if (lds == 0) return -1;
    PostConfArgs a;
    a.logp = logp; a.row_stats = row_stats; a.strideB = strideB; a.strideT = strideT; a.B = B; a.Tmax = Tmax; a.C = C;
    a.S_len = S_len; a.segs = segs; a.seg_cap = seg_cap; a.seg_count = seg_count; a.do_post = do_post; a.extend = extend;
    a.do_conf = do_conf; a.th1 = th1; a.th2 = th2; a.T_rows = T_rows; a.conf = conf; a.status = status; a.cap_cells = cap_cells;
    if (lds > 48 * 1024) {
        (void)hipFuncSetAttribute((const void *)k_postconf<true, 1>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        (void)hipFuncSetAttribute((const void *)k_postconf<false, 1>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    }
    const int grid = B < 65536 ? B : 65536; // (fewer workgroups were measured: profiles/r05_postconf_clip_grid_ab.txt)
    // four waves per utterance while that still leaves the machine room (256 CUs x ~2 such workgroups), one otherwise
    if (B <= BFA_POST_NW4_MAX_BATCH) {
        if (lds > 48 * 1024) {
            (void)hipFuncSetAttribute((const void *)k_postconf<true, 4>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
            (void)hipFuncSetAttribute((const void *)k_postconf<false, 4>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        }
        if (row_stats) hipLaunchKernelGGL((k_postconf<true, 4>), dim3(grid), dim3(256), lds, (hipStream_t)stream_, a);
        else hipLaunchKernelGGL((k_postconf<false, 4>), dim3(grid), dim3(256), lds, (hipStream_t)stream_, a);
        return (int)hipGetLastError();
    }
    if (row_stats) hipLaunchKernelGGL((k_postconf<true, 1>), dim3(grid), dim3(64), lds, (hipStream_t)stream_, a);
    else hipLaunchKernelGGL((k_postconf<false, 1>), dim3(grid), dim3(64), lds, (hipStream_t)stream_, a);
    return (int)hipGetLastError();
}
