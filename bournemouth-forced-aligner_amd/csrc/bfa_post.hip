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
    // :709-714 ; float32 mean of the column slice (accumulated in double, rounded once)
    if (g.start < Tpad && g.phoneme < C && g.start < g.end) {
        const int ee = g.end > Tpad ? Tpad : g.end;
        double acc = 0.0;
        constexpr int U = 8; // independent (clamped) loads in flight: one element per 268-byte row
        for (int f0 = g.start; f0 < ee; f0 += U) {
            float x[U];
            lp.template at_n<U>(f0, ee - 1, g.phoneme, x);
#pragma unroll
            for (int u = 0; u < U; ++u) if (f0 + u < ee) acc += (double)exp_cr(x[u]);
        }
        return (double)(float)(acc / (double)(ee - g.start));
    }
    return 0.001;
}

template <bool RAW>
__global__ __launch_bounds__(64) void k_postprocess(PostArgs a)
{
    extern __shared__ __attribute__((aligned(16))) unsigned char dyn_lds[];
    double *smean = (double *)dyn_lds;                                   // [seg_cap]
    bfa_segment *st = (bfa_segment *)(dyn_lds + (size_t)a.seg_cap * 8);  // [seg_cap]
    const int lane = threadIdx.x & 63;
    const double th1 = a.th1, th2 = a.th2;
    for (int b = blockIdx.x; b < a.B; b += gridDim.x) {
        bfa_segment *sg = a.segs + (int64_t)b * a.seg_cap;
        const LpView<RAW> lp{a.logp + (int64_t)b * a.strideB, a.strideT,
                             RAW ? a.row_stats + 2 * (int64_t)b * a.Tmax : nullptr, a.C};
        int n = a.seg_count[b];
        if (n > a.seg_cap) n = a.seg_cap;
        const int S = a.S_len[b];
        // ---- ensure_target_coverage (default): drop idx == -1 or idx >= S (core.py:488-513)
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
            for (int i = lane; i < m; i += 64) smean[i] = seg_mean(st[i], lp, a.Tmax, a.C);
            post_sync();
            for (int pass = 1; pass <= 4; ++pass) {
                for (int i = lane; i < m; i += 64) extend_pass(pass, i, m, st, lp, a.Tmax, a.C, th1, th2, smean);
                post_sync();
            }
        }
        for (int i = lane; i < m; i += 64) sg[i] = st[i];
        if (lane == 0) a.seg_count[b] = m;
        post_sync();
    }
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
    const size_t lds = (size_t)seg_cap * (8 + sizeof(bfa_segment));
    if (lds > 48 * 1024) { // beyond the default dynamic-LDS limit (seg_cap = Tmax + 1 with ignore_noise = False)
        (void)hipFuncSetAttribute((const void *)k_postprocess<true>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        (void)hipFuncSetAttribute((const void *)k_postprocess<false>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    }
    if (row_stats) hipLaunchKernelGGL(k_postprocess<true>, dim3(B < 65536 ? B : 65536), dim3(64), lds, (hipStream_t)stream_, a);
    else hipLaunchKernelGGL(k_postprocess<false>, dim3(B < 65536 ? B : 65536), dim3(64), lds, (hipStream_t)stream_, a);
    return (int)hipGetLastError();
}
