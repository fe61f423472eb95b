// bfa_capi.cpp -- the C-ABI of include/bfa.h on top of the gfx950 kernels in bfa_kernels.hip.
// Host-side work here is argument checking and carving the caller's workspace; every per-utterance
// decision (mode, stride, band, segmentation) is taken on the device, so a call never synchronises.
#include <hip/hip_runtime.h>

#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <new>
#include <string>
#include <utility>
#include <vector>

#include "bfa_types.hpp"

extern "C" int bfa_launch_align(const bfa::AlignArgs *args, int dp_grid, void *stream, void *ev0, void *ev1,
                                void **aux_streams, void **aux_events, void **fork_event, int (*ensure_aux)(void *), void *ctx,
                                int aux_first, int aux_count);
extern "C" int bfa_launch_conf(const bfa::ConfArgs *args, void *stream);
extern "C" int bfa_launch_prepare(const bfa::AlignArgs *args, float *out, int64_t oB, int64_t oT, void *stream);
extern "C" int bfa_launch_log_softmax(const float *in, int64_t ld_in, float *out, int64_t ld_out, int64_t rows,
                                      int C, void *stream);
extern "C" int bfa_launch_stitch(const float *win, int B, int NW, int F, int C, const float *weights, int total_frames,
                                 float *out, int64_t oB, int64_t oT, void *stream);
extern "C" int bfa_launch_postprocess(const float *logp, float *row_stats, int64_t strideB, int64_t strideT, int B, int Tmax, int C,
                                      const int32_t *S_len, bfa_segment *segs, int seg_cap, int32_t *seg_count,
                                      int extend, double th1, double th2, void *stream);
// the same stages and / or the confidence pass with the probabilities staged in LDS (bfa_post.hip); -1: shapes do not fit
extern "C" int bfa_launch_postconf(const float *logp, float *row_stats, int64_t strideB, int64_t strideT, int B, int Tmax, int C,
                                   const int32_t *S_len, bfa_segment *segs, int seg_cap, int32_t *seg_count, int do_post,
                                   int extend, double th1, double th2, int do_conf, const int32_t *T_rows, float *conf,
                                   int32_t *status, void *stream);
extern "C" int bfa_launch_pack(const int32_t *segs, int seg_cap, const int32_t *seg_count, const float *conf, const int32_t *gidx,
                               int gidx_base, int n, int n_cap, int tuple_cap, int32_t *out, void *stream);
extern "C" int bfa_launch_pack16(const int32_t *segs, int seg_cap, const int32_t *seg_count, int n, int n_cap, int tuple_cap,
                                 int32_t *out, void *stream);
extern "C" int bfa_launch_index_records(const int32_t *rec, int world, int64_t words, int n_max, int n_total, int32_t *owner,
                                        int32_t *offset, int32_t *count, void *stream);
extern "C" int bfa_call_path_impl(int B, int C, int Smax, const bfa::DevParams *p);
extern "C" int bfa_launch_copy(void *dst, const void *src, size_t bytes, void *stream);
static bool staged_post() { return true; } // k_postconf whenever the shapes fit its LDS budget (the tuple-per-lane kernels otherwise)

struct bfa_context {
    int device;
    int num_cu;
    std::string err;
    int profile = 0;       // 0 off, n >= 1: bracket K1 of every n-th bfa_align_batch call
    int profile_tick = 0;
    std::vector<std::pair<hipEvent_t, hipEvent_t>> events; // recorded K1 brackets
    std::vector<std::pair<hipEvent_t, hipEvent_t>> pool;   // reusable pairs
    // K1 class kernels of one call run side by side on these (forked from / joined into the caller's stream)
    static constexpr int NAUX = 6;
    hipStream_t aux[NAUX] = {};
    hipEvent_t aux_done[NAUX] = {};
    hipEvent_t forked = nullptr, forked_b = nullptr; // (_b: the second head of a bfa_align_heads call)
    int naux = 0;
    // bfa_align_heads: heads after the first are enqueued on this stream (forked from / joined into the caller's)
    hipStream_t head_stream = nullptr;
    hipEvent_t head_fork = nullptr, head_join = nullptr;
    // ... or, when both exist, head k on pair[k % 2]: two streams of the caller's (normal) priority created one right after
    // the other, so that the runtime's round-robin puts them on two DIFFERENT hardware queues whatever their number (four by
    // default) -- the heads then share the machine from the start (see bfa_align_heads)
    hipStream_t pair[2] = {nullptr, nullptr};
    hipEvent_t pair_join[2] = {nullptr, nullptr};
    int calls_in_flight = 0; // BFA_OPT_CALLS_IN_FLIGHT
    // The streams above are created when a call first needs them (ensure_aux / ensure_head_streams): a process that only
    // ever aligns single-class batches -- the reference's 16-utterance chunks -- never pays for nine streams and their events
    // (bfa_create 22 ms -> see profiles/r06_cold_start.json).
    bool aux_tried = false, heads_tried = false;
    // Window routing (BFA_OPT_WINDOW_ROUTING; standard mode on the head widths).  The fast sliding window only stands when an
    // utterance ends above the reference's -1000 sentinel; a caller whose posteriors lose ~1 log-unit per frame loses it on
    // every utterance of ~1000 frames, and every call would pay a fast attempt plus the exact rerun.  The last walk kernel of
    // a call leaves the call's window statistics in host-mapped memory (hist[slot], slot 0: C = 67, 1: C = 17); the next calls
    // read whatever has landed -- no synchronisation -- and switch: most fast windows gave up -> exact window at once
    // (XWIN_ROUTE); most exact reruns of a routed call ended above the sentinel -> fast windows again.
    int wide_any_max = 512;       // BFA_OPT_WIDE_ANY_MAX_BATCH (< 0: the class kernels of rounds 2-5 for slots AND pieces)
    int routing = 1;              // 0 never, 1 by history, 2 always exact-first
    int32_t *hist = nullptr;      // [2][8] host-mapped
    bool hist_tried = false;
    int route_state[2] = {0, 0};  // what the next call does
    int hist_seen[2] = {0, 0};    // tag of the statistics the state already reflects
    int hist_tag[2] = {0, 0};     // tag of the last call issued
};

namespace {

inline size_t align_up(size_t x, size_t a) { return (x + a - 1) / a * a; }

struct Carve {
    size_t off = 0;
    char *base;
    explicit Carve(void *b) : base((char *)b) {}
    template <typename T> T *take(size_t n)
    {
        off = align_up(off, 256);
        T *p = base ? (T *)(base + off) : nullptr;
        off += n * sizeof(T);
        return p;
    }
};

bfa::DevParams to_dev_params(const bfa_params *params, bool has_T_len)
{
    bfa::DevParams p;
    std::memset(&p, 0, sizeof(p));
    p.blank = params->blank_id; p.sil = params->silence_id; p.anchors = params->silence_anchors;
    p.ignore_noise = params->ignore_noise; p.truly_forced = params->truly_forced;
    p.boost = params->boost_targets; p.enforce = params->enforce_minimum; p.simple = params->simple;
    p.max_blanks = params->max_blanks > 0 ? params->max_blanks : 10;
    p.class_mask = (uint32_t)params->class_mask; p.win_mask = 0;
    if (!has_T_len) p.class_mask |= (uint32_t)BFA_HINT_UNIFORM_LENGTHS; // every utterance has Tmax frames
    p.win_max_tokens = params->window_max_tokens > 0 ? params->window_max_tokens : bfa::WIN_MAX_TOKENS;
    p.win_max_frames = params->window_max_frames > 0 ? params->window_max_frames : bfa::WIN_MAX_FRAMES;
    p.min_logp = params->has_min_log_prob ? params->min_log_prob : bfa::MIN_LOGP;
    return p;
}

bool segmented_possible(const bfa_params *p)
{
    return !p->simple && p->silence_anchors > 0 && p->silence_id >= 0 &&
           !((uint32_t)p->class_mask & (uint32_t)BFA_HINT_NO_SILENCE_TARGETS);
}

// shape-only bounds shared by bfa_workspace_bytes and bfa_align_batch
struct Layout {
    int item_cap;
    int64_t bp_per_utt; // dwords
};

Layout layout_for(int B, int Tmax, int Smax, const bfa_params *p)
{
    Layout l;
    const bool seg = segmented_possible(p);
    const int Lmax = 4 * (Smax > 0 ? Smax : 1) + 1;
    const int nseg_max = seg ? (Smax + 4) : 0; // speech + silence pieces of one utterance + blank tail
    l.item_cap = B + (seg ? B * nseg_max : 0);
    const int R = bfa::r_class_for_L(Lmax);
    const int64_t quads = (Tmax + 3) / 4 + (seg ? 3 * (int64_t)(Smax / 2 + 2) : 0);
    if (R > 0) l.bp_per_utt = ((quads * bfa::bp_words_for_R(R) * 64 + quads + 4) + 3) & ~(int64_t)3; // 16-byte multiples
    else l.bp_per_utt = ((quads * 4 * ((Lmax + 15) / 16) + 4) + 3) & ~(int64_t)3;
    return l;
}

size_t carve_all(Carve &c, int B, int Tmax, int Smax, int C, const bfa_params *p, const Layout &l, bfa::AlignArgs *a,
                 bool need_frames)
{
    const bool seg = segmented_possible(p);
    auto items = c.take<bfa::Item>((size_t)l.item_cap);
    auto counters = c.take<int32_t>(16);
    auto umask = c.take<uint32_t>((size_t)B * bfa::MASK_WORDS);
    auto uT = c.take<int32_t>((size_t)B);
    auto uS = c.take<int32_t>((size_t)B);
    auto umode = c.take<int32_t>((size_t)B);
    const int anchor_per_utt = Tmax + 6 * (Smax / 2 + 2);
    const int scratch_per_utt = 2 * (Smax + 2) + 4 * (Tmax + 2) + 2 * (Smax + 2) + 5 * (2 * Smax + 6);
    auto anchor = c.take<uint8_t>(seg ? (size_t)B * anchor_per_utt : 1);
    auto psil = c.take<float>(seg ? (size_t)B * Tmax : 1);
    auto cand = c.take<int32_t>((size_t)B);
    // K0's statistics of the boosted rows (reused by K1): only on the 16-rows-per-pass widths with the default flags
    const bool reuse = seg && (C == 67 || C == 17) && p->boost_targets && p->enforce_minimum && !p->simple;
    auto row_stats2 = c.take<float>(reuse ? (size_t)B * Tmax * 2 : 1);
    auto ucand = c.take<uint8_t>((size_t)B);
    auto seg_scratch = c.take<int32_t>(seg ? (size_t)B * scratch_per_utt : 1);
    int32_t *fph = nullptr, *fidx = nullptr;
    if (need_frames) {
        fph = c.take<int32_t>((size_t)B * Tmax);
        fidx = c.take<int32_t>((size_t)B * Tmax);
    }
    auto bp = c.take<uint32_t>((size_t)B * (size_t)l.bp_per_utt);
    auto mix_key = c.take<uint8_t>((size_t)B);   // mixed-length calls: cost bucket per utterance (k_plan) ...
    auto mix_order = c.take<int32_t>((size_t)B); // ... and the utterance slots by decreasing cost (k_order -> k_mix)
    auto piece_list = c.take<int32_t>(seg ? (size_t)bfa::PIECE_BUCKETS * (size_t)l.item_cap : 1); // silence-anchored mode: the pieces by length bucket
    if (a) {
        a->mix_key = mix_key; a->mix_order = mix_order; a->piece_list = piece_list;
        a->items = items; a->item_cap = l.item_cap; a->counters = counters; a->umask = umask; a->uT = uT; a->uS = uS;
        a->umode = umode; a->anchor = anchor; a->anchor_per_utt = anchor_per_utt; a->psil = psil; a->cand = cand;
        a->row_stats2 = reuse ? row_stats2 : nullptr; a->ucand = ucand;
        a->seg_scratch = seg_scratch; a->seg_scratch_per_utt = scratch_per_utt; a->bp = bp; a->bp_cap = (int64_t)B * l.bp_per_utt;
        a->bp_per_utt = l.bp_per_utt;
        if (need_frames) { a->frame_ph = fph; a->frame_idx = fidx; }
    }
    (void)Smax;
    return align_up(c.off, 256);
}

// the handle's device is made current for the duration of a call (the caller may have another one current)
struct DeviceGuard {
    int prev = -1;
    explicit DeviceGuard(bfa_handle h)
    {
        int cur = -1;
        if (h && hipGetDevice(&cur) == hipSuccess && cur != h->device) { prev = cur; (void)hipSetDevice(h->device); }
    }
    ~DeviceGuard() { if (prev >= 0) (void)hipSetDevice(prev); }
};

int fail(bfa_handle h, int code, const char *msg)
{
    if (h) h->err = msg;
    return code;
}

// Auxiliary streams of a handle, created when a call first has more than one K1 kernel to run side by side (best effort:
// without them the class kernels simply run one after the other).  The handle's device is current (DeviceGuard).
int ensure_aux(void *ctx)
{
    bfa_context *h = (bfa_context *)ctx;
    if (h->aux_tried) return h->forked ? h->naux : 0;
    h->aux_tried = true;
    bool ok = hipEventCreateWithFlags(&h->forked, hipEventDisableTiming) == hipSuccess &&
              hipEventCreateWithFlags(&h->forked_b, hipEventDisableTiming) == hipSuccess;
    if (!ok) h->forked = nullptr;
    // Created in the order 0, 3, 1, 4, 2, 5: the runtime deals streams to its hardware queues (four by default) in turn, and
    // the two halves (bfa_align_heads: one per head) should not start on the same queue -- nor, when the heads' own streams
    // were created just before (ensure_head_streams), on the other head's (best effort: the mapping is the runtime's).
    static const int order[bfa_context::NAUX] = {0, 3, 1, 4, 2, 5};
    int made = 0;
    // (Auxiliary streams of another PRIORITY get hardware queues of their own -- four more -- and were measured: the C5 proxy
    // 4.69 -> 4.44 ms (high) / 5.13 -> 4.13 at peak 3 (low), but the sixteen-utterance chunk 0.97 -> 1.30 / 1.27 ms and the
    // headline batch 0.404 -> 0.430 / 0.437: profiles/r06_aux_priority_ab.txt.  Normal priority stays.)
    for (int j = 0; ok && j < bfa_context::NAUX; ++j) {
        const int k = order[j];
        ok = hipStreamCreateWithFlags(&h->aux[k], hipStreamNonBlocking) == hipSuccess &&
             hipEventCreateWithFlags(&h->aux_done[k], hipEventDisableTiming) == hipSuccess;
        if (ok) ++made;
    }
    h->naux = made == bfa_context::NAUX ? made : 0;
    return h->forked ? h->naux : 0;
}

// Streams of bfa_align_heads, created by its first call with more than one head.
void ensure_head_streams(bfa_context *h)
{
    if (h->heads_tried) return;
    h->heads_tried = true;
    // pair[]: two streams of the caller's (normal) priority created one right after the other, so that the runtime's
    // round-robin puts them on two DIFFERENT hardware queues whatever their number (four by default)
#ifndef BFA_NO_PAIR_STREAMS // (A/B of the stream -> queue mapping with several calls in flight)
    if (hipStreamCreateWithFlags(&h->pair[0], hipStreamNonBlocking) != hipSuccess) h->pair[0] = nullptr;
#endif
    if (h->pair[0] && hipStreamCreateWithFlags(&h->pair[1], hipStreamNonBlocking) != hipSuccess) h->pair[1] = nullptr;
    for (int k = 0; k < 2; ++k)
        if (h->pair[k] && hipEventCreateWithFlags(&h->pair_join[k], hipEventDisableTiming) != hipSuccess) {
            (void)hipStreamDestroy(h->pair[k]);
            h->pair[k] = nullptr;
        }
    // The runtime maps the streams of one priority onto a few hardware queues (four by default), and a queue runs its
    // kernels in order: created like the auxiliary streams, this stream landed on the CALLER's queue and the heads
    // ran one behind the other (profiles/r03_realtext_timeline_before.txt).  A stream of another priority gets a queue
    // of its own; the later heads are the narrow ones (group head: C = 17), which fill in beside the phoneme head.
    // Lowest priority (measured against normal / high: 2.37 vs 2.42 / 2.40 ms one step in flight).
    int prio_lo = 0, prio_hi = 0;
    (void)hipDeviceGetStreamPriorityRange(&prio_lo, &prio_hi); // (numerically: lowest priority = largest value)
    if (hipStreamCreateWithPriority(&h->head_stream, hipStreamNonBlocking, prio_lo) != hipSuccess) h->head_stream = nullptr;
    if (hipEventCreateWithFlags(&h->head_fork, hipEventDisableTiming) != hipSuccess ||
        hipEventCreateWithFlags(&h->head_join, hipEventDisableTiming) != hipSuccess) {
        if (h->head_stream) (void)hipStreamDestroy(h->head_stream);
        h->head_stream = nullptr;
        if (h->head_fork) { (void)hipEventDestroy(h->head_fork); h->head_fork = nullptr; }
    }
}

} // namespace

extern "C" {

const char *bfa_version(void) { return "bfa-hip 0.1.0 (gfx950)"; }
int bfa_abi_version(void) { return BFA_ABI_VERSION; }

void bfa_params_default(bfa_params *p, int blank_id, int silence_id)
{
    std::memset(p, 0, sizeof(*p));
    p->blank_id = blank_id;
    p->silence_id = silence_id;
    p->silence_anchors = 10;
    p->ignore_noise = 1;
    p->truly_forced = 1;
    p->boost_targets = 1;
    p->enforce_minimum = 1;
    p->simple = 0;
    p->max_blanks = 10;
}

int bfa_create(bfa_handle *out, int device)
{
    if (!out) return BFA_ERR_INVALID_ARGUMENT;
    *out = nullptr;
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess || n <= 0) return BFA_ERR_NO_DEVICE; // no CPU fallback
    if (device < 0) {
        if (hipGetDevice(&device) != hipSuccess) return BFA_ERR_NO_DEVICE;
    }
    if (device >= n) return BFA_ERR_INVALID_ARGUMENT;
    int num_cu = 0; // (one attribute, not hipGetDeviceProperties: that call fills ~1 KB of fields through dozens of queries)
    if (hipDeviceGetAttribute(&num_cu, hipDeviceAttributeMultiprocessorCount, device) != hipSuccess) return BFA_ERR_NO_DEVICE;
    bfa_context *h = new (std::nothrow) bfa_context();
    if (!h) return BFA_ERR_INVALID_ARGUMENT;
    h->device = device;
    h->num_cu = num_cu;
    *out = h;
    return BFA_OK;
}

int bfa_profile_enable(bfa_handle h, int on)
{
    if (!h) return BFA_ERR_INVALID_ARGUMENT;
    h->profile = on > 0 ? on : 0;
    h->profile_tick = 0;
    return BFA_OK;
}

int bfa_profile_collect(bfa_handle h, float *out_ms_host, int cap)
{
    if (!h || (!out_ms_host && cap > 0)) return BFA_ERR_INVALID_ARGUMENT;
    int n = 0;
    for (auto &pr : h->events) {
        float ms = 0.0f;
        (void)hipEventSynchronize(pr.second);
        (void)hipEventElapsedTime(&ms, pr.first, pr.second);
        if (n < cap) out_ms_host[n++] = ms;
        h->pool.push_back(pr);
    }
    h->events.clear();
    return n;
}

int bfa_profile_collect_spans(bfa_handle h, void *base_event, float *out_start_ms_host, float *out_end_ms_host, int cap)
{
    if (!h || !base_event || ((!out_start_ms_host || !out_end_ms_host) && cap > 0)) return BFA_ERR_INVALID_ARGUMENT;
    int n = 0;
    for (auto &pr : h->events) {
        float t0 = 0.0f, t1 = 0.0f;
        (void)hipEventSynchronize(pr.second);
        (void)hipEventElapsedTime(&t0, (hipEvent_t)base_event, pr.first);
        (void)hipEventElapsedTime(&t1, (hipEvent_t)base_event, pr.second);
        if (n < cap) { out_start_ms_host[n] = t0; out_end_ms_host[n] = t1; ++n; }
        h->pool.push_back(pr);
    }
    h->events.clear();
    return n;
}

int bfa_destroy(bfa_handle h)
{
    if (h) {
        for (auto &pr : h->events) { (void)hipEventDestroy(pr.first); (void)hipEventDestroy(pr.second); }
        for (auto &pr : h->pool) { (void)hipEventDestroy(pr.first); (void)hipEventDestroy(pr.second); }
        for (int k = 0; k < bfa_context::NAUX; ++k) {
            if (h->aux[k]) (void)hipStreamDestroy(h->aux[k]);
            if (h->aux_done[k]) (void)hipEventDestroy(h->aux_done[k]);
        }
        if (h->forked) (void)hipEventDestroy(h->forked);
        if (h->forked_b) (void)hipEventDestroy(h->forked_b);
        for (int k = 0; k < 2; ++k) {
            if (h->pair[k]) (void)hipStreamDestroy(h->pair[k]);
            if (h->pair_join[k]) (void)hipEventDestroy(h->pair_join[k]);
        }
        if (h->head_stream) (void)hipStreamDestroy(h->head_stream);
        if (h->head_fork) (void)hipEventDestroy(h->head_fork);
        if (h->head_join) (void)hipEventDestroy(h->head_join);
        if (h->hist) (void)hipHostFree(h->hist);
    }
    delete h;
    return BFA_OK;
}

const char *bfa_last_error(bfa_handle h) { return h ? h->err.c_str() : "null handle"; }

size_t bfa_workspace_bytes(int B, int Tmax, int Smax, int C, const bfa_params *p)
{
    if (B <= 0 || Tmax <= 0 || Smax <= 0 || !p) return 0;
    const Layout l = layout_for(B, Tmax, Smax, p);
    Carve c(nullptr);
    return carve_all(c, B, Tmax, Smax, C, p, l, nullptr, true) + 256;
}

int bfa_call_counters(bfa_handle h, const void *workspace, int B, int Tmax, int Smax, int C, const bfa_params *p,
                      int32_t *out_host, void *stream)
{
    if (!h || !workspace || !p || !out_host || B <= 0 || Tmax <= 0 || Smax <= 0) return BFA_ERR_INVALID_ARGUMENT;
    DeviceGuard guard(h);
    const Layout l = layout_for(B, Tmax, Smax, p);
    const uintptr_t aligned = ((uintptr_t)workspace + 255) & ~(uintptr_t)255;
    Carve c((void *)aligned);
    bfa::AlignArgs a;
    std::memset(&a, 0, sizeof(a));
    (void)carve_all(c, B, Tmax, Smax, C, p, l, &a, false);
    if (hipMemcpyAsync(out_host, a.counters, 16 * sizeof(int32_t), hipMemcpyDeviceToHost, (hipStream_t)stream) != hipSuccess ||
        hipStreamSynchronize((hipStream_t)stream) != hipSuccess)
        return fail(h, BFA_ERR_LAUNCH, "copy of the call counters failed");
    return BFA_OK;
}

} // extern "C"

static int align_impl(bfa_handle h, const float *logp, float *row_stats, int64_t strideB, int64_t strideT, int B, int Tmax, int C,
                    const int32_t *T_len, const int32_t *tokens, const int32_t *S_len, int Smax,
                    const bfa_params *params, int32_t *out_frame_phoneme, int32_t *out_frame_idx,
                    bfa_segment *out_segs, int seg_cap, int32_t *out_seg_count, int32_t *out_status,
                    int32_t *out_mode, void *workspace, size_t workspace_bytes, void *stream, int aux_set = -1)
{
    if (!h) return BFA_ERR_INVALID_ARGUMENT;
    DeviceGuard guard(h);
    if (!logp || !tokens || !S_len || !params || !out_segs || !out_seg_count || !out_status || !workspace)
        return fail(h, BFA_ERR_INVALID_ARGUMENT, "null pointer argument");
    if (B <= 0 || Tmax <= 0 || Smax <= 0 || seg_cap <= 0) return fail(h, BFA_ERR_INVALID_ARGUMENT, "non-positive size");
    if (C < 2 || C > bfa::MAX_C) return fail(h, BFA_ERR_UNSUPPORTED, "C must be in [2,128]");
    if (params->blank_id < 0 || params->blank_id >= C)
        return fail(h, BFA_ERR_INVALID_ARGUMENT, "Blank ID not set"); // forced_alignment.py:104-105
    if (strideT < C) return fail(h, BFA_ERR_INVALID_ARGUMENT, "strideT < C");
    // grids over (utterance, 64-frame segment) units must fit a launch (the exact window's tail kernel, bfa_dp4.inc); such a
    // batch would hold > 10^11 posterior rows
    if ((int64_t)B * ((Tmax + 63) / 64) > 0x7fffffffll) return fail(h, BFA_ERR_UNSUPPORTED, "B x Tmax / 64 exceeds a launch grid");
    if ((out_frame_phoneme == nullptr) != (out_frame_idx == nullptr))
        return fail(h, BFA_ERR_INVALID_ARGUMENT, "frame outputs must both be given or both be NULL");

    const bool need_frames = (out_frame_phoneme == nullptr);
    const Layout l = layout_for(B, Tmax, Smax, params);
    bfa::AlignArgs a;
    std::memset(&a, 0, sizeof(a));
    const uintptr_t basep = (uintptr_t)workspace;
    const uintptr_t aligned = (basep + 255) & ~(uintptr_t)255;
    Carve c((void *)aligned);
    a.frame_ph = out_frame_phoneme;
    a.frame_idx = out_frame_idx;
    const size_t need = carve_all(c, B, Tmax, Smax, C, params, l, &a, need_frames) + (aligned - basep);
    if (need > workspace_bytes) return fail(h, BFA_ERR_WORKSPACE_TOO_SMALL, "workspace too small");

    a.logp = logp; a.row_stats = row_stats; a.strideB = strideB; a.strideT = strideT;
    if (row_stats) {
        // rows no kernel prepares (silence fills, frames beyond an utterance, items without a DP) keep this NaN and get
        // their statistics from the first sparse reader that needs them (bfa_math.hpp: row_stats_on_demand).  In the
        // silence-anchored mode on the head widths K0 prepares EVERY row of EVERY utterance (a.row_stats2: k_silprob3) and
        // marks the frames beyond an utterance itself: no 8 B x B x Tmax fill ahead of the call (32 MB per head for the
        // 4096 x 1000 batch -- and, on the runtime's default four hardware queues, a fill kernel that sat 0.28 ms behind the
        // other head's K0: profiles/r04_realtext_timeline_q4_before.txt)
        if (!a.row_stats2 &&
            hipMemsetAsync(row_stats, 0xFF, (size_t)B * (size_t)Tmax * 2 * sizeof(float), (hipStream_t)stream) != hipSuccess)
            return fail(h, BFA_ERR_LAUNCH, "memset of row_stats failed");
    }
    a.B = B; a.Tmax = Tmax; a.C = C; a.Smax = Smax;
    a.T_len = T_len; a.tokens = tokens; a.S_len = S_len;
    a.p = to_dev_params(params, T_len != nullptr);
    if (a.p.min_logp != a.p.min_logp) return fail(h, BFA_ERR_INVALID_ARGUMENT, "min_log_prob is NaN");
    a.segs = out_segs; a.seg_cap = seg_cap; a.seg_count = out_seg_count; a.status = out_status; a.mode = out_mode;

    // window routing: standard-mode calls on the head widths (the only ones with fast windows)
    if ((C == 67 || C == 17) && !segmented_possible(params) && params->boost_targets && params->enforce_minimum && !params->simple) {
        const int slot = C == 67 ? 0 : 1;
        if (!h->hist_tried) {
            h->hist_tried = true;
            void *hp = nullptr;
            if (h->routing == 1 && hipHostMalloc(&hp, 2 * 8 * sizeof(int32_t), hipHostMallocMapped) == hipSuccess) {
                std::memset(hp, 0, 2 * 8 * sizeof(int32_t));
                h->hist = (int32_t *)hp;
            }
        }
        if (h->routing == 1 && h->hist) {
            volatile int32_t *v = h->hist + 8 * slot;
            const int tag = v[0];
            if (tag != h->hist_seen[slot]) { // statistics of a call this state has not seen yet
                const int gave_up = v[1], xdone = v[2], xalive = v[3], nB = v[4], routed = v[5];
                if (v[0] == tag && nB > 0) {
                    // A rerun kernel behind the fast windows is a second serial chain however few items it holds: 139 of 4096
                    // items redone cost a headline-shaped call 0.13 ms (0.56 against 0.43), the exact window for ALL of them
                    // 0.04 (profiles/r06_softness_after.jsonl) -- so one item in 64 is enough to switch, and the way back
                    // wants 99 in 100 of a routed call's items above the sentinel
                    if (!routed && 64 * gave_up >= nB) h->route_state[slot] = 1;
                    else if (routed && xdone > 0 && 100 * (int64_t)xalive >= 99 * (int64_t)xdone) h->route_state[slot] = 0;
                    h->hist_seen[slot] = tag;
                }
            }
            a.hist = h->hist + 8 * slot; // (the host pointer of mapped memory is valid on the device)
            a.hist_tag = ++h->hist_tag[slot];
            if (a.hist_tag == 0) a.hist_tag = ++h->hist_tag[slot];
        }
        const bool route = h->routing == 2 || (h->routing == 1 && h->route_state[slot] != 0);
        if (route) a.p.xwin_mask |= bfa::XWIN_ROUTE; // (bfa_launch_align keeps the bit)
    }
    // one wavefront per work item; surplus items are taken by the blocks' stride loops
    int grid = l.item_cap < 16384 ? l.item_cap : 16384;
    void *ev0 = nullptr, *ev1 = nullptr;
    if (h->profile > 0 && (h->profile_tick++ % h->profile) == 0) {
        std::pair<hipEvent_t, hipEvent_t> pr;
        if (!h->pool.empty()) { pr = h->pool.back(); h->pool.pop_back(); }
        else { (void)hipEventCreate(&pr.first); (void)hipEventCreate(&pr.second); }
        h->events.push_back(pr);
        ev0 = (void *)pr.first; ev1 = (void *)pr.second;
    }
    a.wide_any_max = h->wide_any_max;
    a.pieces_merged = h->wide_any_max >= 0 ? 1 : 0;
    // the heads of one bfa_align_heads call fan out over DIFFERENT halves of the handle's auxiliary streams (aux_set 0 / 1:
    // three streams each -- the class kernels are laid out on three, bfa_dp3.inc launch3): on the same ones the class kernels
    // of the second head queued behind the first head's (profiles/r06_latency_realtext_timeline_b16_before.txt)
    const int half = bfa_context::NAUX / 2;
    const int rc = bfa_launch_align(&a, grid, stream, ev0, ev1, (void **)h->aux, (void **)h->aux_done,
                                    (void **)((aux_set > 0 && (aux_set & 1)) ? &h->forked_b : &h->forked), ensure_aux, (void *)h, aux_set < 0 ? 0 : half * (aux_set & 1), aux_set < 0 ? bfa_context::NAUX : half);
    if (rc != 0) return fail(h, BFA_ERR_LAUNCH, hipGetErrorString((hipError_t)rc));
    return BFA_OK;
}

extern "C" {

int bfa_align_batch(bfa_handle h, const float *logp, int64_t strideB, int64_t strideT, int B, int Tmax, int C,
                    const int32_t *T_len, const int32_t *tokens, const int32_t *S_len, int Smax,
                    const bfa_params *params, int32_t *out_frame_phoneme, int32_t *out_frame_idx,
                    bfa_segment *out_segs, int seg_cap, int32_t *out_seg_count, int32_t *out_status,
                    int32_t *out_mode, void *workspace, size_t workspace_bytes, void *stream)
{
    return align_impl(h, logp, nullptr, strideB, strideT, B, Tmax, C, T_len, tokens, S_len, Smax, params,
                      out_frame_phoneme, out_frame_idx, out_segs, seg_cap, out_seg_count, out_status, out_mode, workspace,
                      workspace_bytes, stream);
}

int bfa_align_heads(bfa_handle h, const bfa_head *heads, int n_heads, int B, int Tmax, const int32_t *T_len,
                    const int32_t *S_len, void *stream)
{
    if (!h) return BFA_ERR_INVALID_ARGUMENT;
    DeviceGuard guard(h);
    if (!heads || n_heads <= 0 || n_heads > 8) return fail(h, BFA_ERR_INVALID_ARGUMENT, "bad head list");
    for (int k = 0; k < n_heads; ++k)
        if (!heads[k].logits || !heads[k].out_row_stats) return fail(h, BFA_ERR_INVALID_ARGUMENT, "head without logits / out_row_stats");
    // Two heads side by side.  Round 3 put the later heads on ONE side stream of the lowest priority (a stream of another
    // priority gets a hardware queue of its own) and head 0 on the caller's: with the runtime's default four queues the
    // low-priority queue is not served while the other has workgroups waiting -- head 1's first kernel sat 0.27 ms behind
    // head 0's K0 (profiles/r04_realtext_timeline_q4_before.txt) and a step took 2.1 ms against 1.7 with eight queues.
    // Now head k runs on pair[k % 2], two streams of the caller's priority on two different queues; the caller's stream
    // waits for both: 2.09 -> 1.75 ms one call at a time on four queues.  With several calls in flight (three handles) the
    // six extra streams collide on the queues and the side stream is the better layout (1.46 against 1.66 ms per step):
    // BFA_OPT_CALLS_IN_FLIGHT tells which caller this is.
    if (n_heads > 1) ensure_head_streams(h);
    const bool paired = n_heads > 1 && !h->calls_in_flight && h->pair[0] && h->pair[1] && h->head_fork;
    const bool side = !paired && n_heads > 1 && h->head_stream != nullptr;
    if (paired) {
        (void)hipEventRecord(h->head_fork, (hipStream_t)stream);
        (void)hipStreamWaitEvent(h->pair[0], h->head_fork, 0);
        (void)hipStreamWaitEvent(h->pair[1], h->head_fork, 0);
    }
    if (side) {
        (void)hipEventRecord(h->head_fork, (hipStream_t)stream);
        (void)hipStreamWaitEvent(h->head_stream, h->head_fork, 0);
    }
    int rc = BFA_OK;
    // the later heads first; head 0 (the wide phoneme head) last: its kernels are the long ones, the others fill in beside them
    // (Head by head.  Enqueueing the fronts of all heads -- plan, K0, segment planner -- before any head's class kernels was built
    // and measured: the second head then starts at once instead of behind the first head's class kernel on a shared hardware
    // queue, but the staggered start is the better schedule -- one head's memory-bound K0 beside the other's latency-bound chains:
    // C5 proxy 4.12 against 4.47 ms at peak 9, 3.96 against 4.42 at peak 3, profiles/r06_heads_fronts_first_ab.txt.)
    for (int k = n_heads - 1; k >= 0 && rc == BFA_OK; --k) {
        const bfa_head &hd = heads[k];
        void *st = paired ? (void *)h->pair[k & 1] : ((side && k > 0) ? (void *)h->head_stream : stream);
        rc = align_impl(h, hd.logits, hd.out_row_stats, hd.strideB, hd.strideT, B, Tmax, hd.C, T_len, hd.tokens, S_len,
                        hd.Smax, &hd.params, hd.out_frame_phoneme, hd.out_frame_idx, hd.out_segs, hd.seg_cap,
                        hd.out_seg_count, hd.out_status, hd.out_mode, hd.workspace, hd.workspace_bytes, st, n_heads > 1 ? k : -1);
        // core.py:925-937 for this head on ITS stream: coverage + soft boundaries, then the confidences of the final tuples
        // -- in ONE kernel when both are asked for and the shapes fit its LDS staging (bfa_post.hip: k_postconf)
        if (rc == BFA_OK && hd.postprocess && hd.out_conf && staged_post() && hd.seg_cap <= 6500) {
            const double th1 = std::pow(10.0, -3.0), th2 = std::pow(10.0, -(double)hd.boundary_softness);
            const int lrc = bfa_launch_postconf(hd.logits, hd.out_row_stats, hd.strideB, hd.strideT, B, Tmax, hd.C, S_len, hd.out_segs,
                                                hd.seg_cap, hd.out_seg_count, 1, hd.extend, th1, th2, 1, nullptr, hd.out_conf,
                                                hd.out_conf_status, st);
            if (lrc == 0) continue;
            if (lrc > 0) { rc = fail(h, BFA_ERR_LAUNCH, hipGetErrorString((hipError_t)lrc)); break; }
        }
        if (rc == BFA_OK && hd.postprocess)
            rc = bfa_postprocess(h, hd.logits, hd.out_row_stats, hd.strideB, hd.strideT, B, Tmax, hd.C, S_len, hd.out_segs,
                                 hd.seg_cap, hd.out_seg_count, hd.extend, hd.boundary_softness, st);
        if (rc == BFA_OK && hd.out_conf)
            rc = bfa_confidences(h, hd.logits, hd.out_row_stats, hd.strideB, hd.strideT, B, Tmax, hd.C, nullptr, hd.out_segs,
                                 hd.seg_cap, hd.out_seg_count, hd.out_conf, hd.out_conf_status, st);
    }
    if (paired) {
        for (int k = 0; k < 2; ++k) {
            (void)hipEventRecord(h->pair_join[k], h->pair[k]);
            (void)hipStreamWaitEvent((hipStream_t)stream, h->pair_join[k], 0);
        }
    }
    if (side) {
        (void)hipEventRecord(h->head_join, h->head_stream);
        (void)hipStreamWaitEvent((hipStream_t)stream, h->head_join, 0);
    }
    return rc;
}

int bfa_set_option(bfa_handle h, int option, int value)
{
    if (!h) return BFA_ERR_INVALID_ARGUMENT;
    if (option == BFA_OPT_CALLS_IN_FLIGHT) { h->calls_in_flight = value != 0; return BFA_OK; }
    if (option == BFA_OPT_WIDE_ANY_MAX_BATCH) { h->wide_any_max = value; return BFA_OK; }
    if (option == BFA_OPT_PRECREATE_STREAMS) { // (the streams a handle otherwise creates when a call first needs them)
        DeviceGuard guard(h);
        if (value & 1) (void)ensure_aux((void *)h);
        if (value & 2) ensure_head_streams(h);
        return BFA_OK;
    }
    if (option == BFA_OPT_WINDOW_ROUTING) {
        if (value < 0 || value > 2) return fail(h, BFA_ERR_INVALID_ARGUMENT, "window routing: 0 never, 1 by history, 2 always");
        h->routing = value; h->route_state[0] = h->route_state[1] = 0;
        return BFA_OK;
    }
    return fail(h, BFA_ERR_INVALID_ARGUMENT, "unknown option");
}

int bfa_prepare_emissions(bfa_handle h, const float *logp, int64_t strideB, int64_t strideT, int B, int Tmax, int C,
                          const int32_t *T_len, const int32_t *tokens, const int32_t *S_len, int Smax,
                          const bfa_params *params, float *out, int64_t out_strideB, int64_t out_strideT,
                          void *workspace, size_t workspace_bytes, void *stream)
{
    if (!h) return BFA_ERR_INVALID_ARGUMENT;
    DeviceGuard guard(h);
    if (!logp || !tokens || !S_len || !params || !out || !workspace) return fail(h, BFA_ERR_INVALID_ARGUMENT, "null pointer argument");
    if (B <= 0 || Tmax <= 0 || Smax <= 0) return fail(h, BFA_ERR_INVALID_ARGUMENT, "non-positive size");
    if (C != 67 && C != 17) return fail(h, BFA_ERR_UNSUPPORTED, "bfa_prepare_emissions supports C = 67 and C = 17");
    const Layout l = layout_for(B, Tmax, Smax, params);
    bfa::AlignArgs a;
    std::memset(&a, 0, sizeof(a));
    const uintptr_t basep = (uintptr_t)workspace;
    const uintptr_t aligned = (basep + 255) & ~(uintptr_t)255;
    Carve c((void *)aligned);
    const size_t need = carve_all(c, B, Tmax, Smax, C, params, l, &a, true) + (aligned - basep);
    if (need > workspace_bytes) return fail(h, BFA_ERR_WORKSPACE_TOO_SMALL, "workspace too small");
    a.logp = logp; a.strideB = strideB; a.strideT = strideT;
    a.B = B; a.Tmax = Tmax; a.C = C; a.Smax = Smax;
    a.T_len = T_len; a.tokens = tokens; a.S_len = S_len;
    a.p.blank = params->blank_id; a.p.sil = params->silence_id; a.p.anchors = 0;
    a.p.boost = params->boost_targets; a.p.enforce = params->enforce_minimum; a.p.simple = 1; a.p.max_blanks = 10;
    a.p.class_mask = 0; a.p.win_mask = 0; a.p.win_max_tokens = 0; a.p.win_max_frames = 0;
    a.p.min_logp = params->has_min_log_prob ? params->min_log_prob : bfa::MIN_LOGP;
    if (a.p.min_logp != a.p.min_logp) return fail(h, BFA_ERR_INVALID_ARGUMENT, "min_log_prob is NaN"); // (as in align_core)
    // k_plan also writes seg_count/status: point them at scratch
    a.seg_count = a.uS; a.status = a.umode; a.seg_cap = 1;
    a.seg_count = (int32_t *)a.frame_ph; a.status = (int32_t *)a.frame_idx;
    const int rc = bfa_launch_prepare(&a, out, out_strideB, out_strideT, stream);
    if (rc != 0) return fail(h, BFA_ERR_LAUNCH, rc < 0 ? "unsupported width" : hipGetErrorString((hipError_t)rc));
    return BFA_OK;
}

int bfa_confidences(bfa_handle h, const float *logp, float *row_stats, int64_t strideB, int64_t strideT, int B, int Tmax, int C,
                    const int32_t *T_rows, const bfa_segment *segs, int seg_cap, const int32_t *seg_count,
                    float *out_conf, int32_t *out_item_status, void *stream)
{
    if (!h) return BFA_ERR_INVALID_ARGUMENT;
    DeviceGuard guard(h);
    if (!logp || !segs || !seg_count || !out_conf) return fail(h, BFA_ERR_INVALID_ARGUMENT, "null pointer argument");
    if (B <= 0 || Tmax <= 0 || C <= 0 || seg_cap <= 0) return fail(h, BFA_ERR_INVALID_ARGUMENT, "non-positive size");
#ifndef BFA_CONF_TUPLE_PER_LANE // (A/B: the confidences alone through the tuple-per-lane kernel)
    if (staged_post()) { // the probabilities staged in LDS (k_postconf); falls through when the shapes do not fit
        const int lrc = bfa_launch_postconf(logp, row_stats, strideB, strideT, B, Tmax, C, nullptr, const_cast<bfa_segment *>(segs),
                                            seg_cap, const_cast<int32_t *>(seg_count), 0, 0, 0.0, 0.0, 1, T_rows, out_conf,
                                            out_item_status, stream);
        if (lrc == 0) return BFA_OK;
        if (lrc > 0) return fail(h, BFA_ERR_LAUNCH, hipGetErrorString((hipError_t)lrc));
    }
#endif
    bfa::ConfArgs a;
    a.logp = logp; a.row_stats = row_stats; a.strideB = strideB; a.strideT = strideT; a.B = B; a.Tmax = Tmax; a.C = C; a.T_rows = T_rows;
    a.segs = segs; a.seg_cap = seg_cap; a.seg_count = seg_count; a.conf = out_conf; a.status = out_item_status;
    const int rc = bfa_launch_conf(&a, stream);
    if (rc != 0) return fail(h, BFA_ERR_LAUNCH, hipGetErrorString((hipError_t)rc));
    return BFA_OK;
}

int bfa_postprocess(bfa_handle h, const float *logp, float *row_stats, int64_t strideB, int64_t strideT, int B, int Tmax, int C,
                    const int32_t *S_len, bfa_segment *segs, int seg_cap, int32_t *seg_count, int extend,
                    int boundary_softness, void *stream)
{
    if (!h) return BFA_ERR_INVALID_ARGUMENT;
    DeviceGuard guard(h);
    if (!logp || !S_len || !segs || !seg_count) return fail(h, BFA_ERR_INVALID_ARGUMENT, "null pointer argument");
    if (B <= 0 || Tmax <= 0 || C <= 0 || seg_cap <= 0) return fail(h, BFA_ERR_INVALID_ARGUMENT, "non-positive size");
    // core.py:699-701 : python `10.0 ** -n` is libm pow on doubles
    const double th1 = std::pow(10.0, -3.0), th2 = std::pow(10.0, -(double)boundary_softness);
    if (staged_post()) {
        const int lrc = bfa_launch_postconf(logp, row_stats, strideB, strideT, B, Tmax, C, S_len, segs, seg_cap, seg_count, 1, extend,
                                            th1, th2, 0, nullptr, nullptr, nullptr, stream);
        if (lrc == 0) return BFA_OK;
        if (lrc > 0) return fail(h, BFA_ERR_LAUNCH, hipGetErrorString((hipError_t)lrc));
    }
    const int rc = bfa_launch_postprocess(logp, row_stats, strideB, strideT, B, Tmax, C, S_len, segs, seg_cap, seg_count, extend,
                                          th1, th2, stream);
    if (rc != 0) return fail(h, BFA_ERR_LAUNCH, hipGetErrorString((hipError_t)rc));
    return BFA_OK;
}

int bfa_log_softmax(bfa_handle h, const float *logits, int64_t ld_in, float *out, int64_t ld_out, int64_t rows,
                    int C, void *stream)
{
    if (!h) return BFA_ERR_INVALID_ARGUMENT;
    DeviceGuard guard(h);
    if (!logits || !out || rows < 0) return fail(h, BFA_ERR_INVALID_ARGUMENT, "bad argument");
    if (C < 2 || C > bfa::MAX_C) return fail(h, BFA_ERR_UNSUPPORTED, "C must be in [2,128]");
    if (rows == 0) return BFA_OK;
    const int rc = bfa_launch_log_softmax(logits, ld_in, out, ld_out, rows, C, stream);
    if (rc != 0) return fail(h, BFA_ERR_LAUNCH, hipGetErrorString((hipError_t)rc));
    return BFA_OK;
}

int bfa_call_path(int B, int Tmax, int Smax, int C, const bfa_params *params, int has_T_len)
{
    (void)Tmax;
    if (!params || B <= 0 || Smax <= 0 || C < 2) return BFA_ERR_INVALID_ARGUMENT;
    const bfa::DevParams p = to_dev_params(params, has_T_len != 0);
    return bfa_call_path_impl(B, C, Smax, &p);
}

int bfa_profile_copy(bfa_handle h, void *dst, const void *src, size_t bytes, void *stream)
{
    if (!h) return BFA_ERR_INVALID_ARGUMENT;
    DeviceGuard guard(h);
    if (!dst || !src || (bytes & 15) || ((uintptr_t)dst & 15) || ((uintptr_t)src & 15))
        return fail(h, BFA_ERR_INVALID_ARGUMENT, "dst / src / bytes must be 16-byte multiples");
    const int rc = bfa_launch_copy(dst, src, bytes, stream);
    if (rc != 0) return fail(h, BFA_ERR_LAUNCH, hipGetErrorString((hipError_t)rc));
    return BFA_OK;
}

int bfa_pack_results(bfa_handle h, const bfa_segment *segs, int seg_cap, const int32_t *seg_count, const float *conf,
                     const int32_t *global_index, int gidx_base, int n, int n_cap, int tuple_cap, int32_t *out, void *stream)
{
    if (!h) return BFA_ERR_INVALID_ARGUMENT;
    DeviceGuard guard(h);
    if (!out || n < 0 || n_cap < n || tuple_cap < 0 || seg_cap < 0) return fail(h, BFA_ERR_INVALID_ARGUMENT, "bad argument");
    if (n > 0 && (!segs || !seg_count)) return fail(h, BFA_ERR_INVALID_ARGUMENT, "null pointer argument");
    if (((uintptr_t)out & 15) || ((uintptr_t)segs & 15)) return fail(h, BFA_ERR_INVALID_ARGUMENT, "segs / out must be 16-byte aligned");
    const int rc = bfa_launch_pack((const int32_t *)segs, seg_cap, seg_count, conf, global_index, gidx_base, n, n_cap, tuple_cap, out, stream);
    if (rc != 0) return fail(h, BFA_ERR_LAUNCH, hipGetErrorString((hipError_t)rc));
    return BFA_OK;
}

int bfa_pack_results16(bfa_handle h, const bfa_segment *segs, int seg_cap, const int32_t *seg_count, int n, int n_cap,
                       int tuple_cap, int32_t *out, void *stream)
{
    if (!h) return BFA_ERR_INVALID_ARGUMENT;
    DeviceGuard guard(h);
    if (!out || n < 0 || n_cap < n || tuple_cap < 0 || seg_cap < 0) return fail(h, BFA_ERR_INVALID_ARGUMENT, "bad argument");
    if (n > 0 && (!segs || !seg_count)) return fail(h, BFA_ERR_INVALID_ARGUMENT, "null pointer argument");
    if (((uintptr_t)out & 15) || ((uintptr_t)segs & 15)) return fail(h, BFA_ERR_INVALID_ARGUMENT, "segs / out must be 16-byte aligned");
    const int rc = bfa_launch_pack16((const int32_t *)segs, seg_cap, seg_count, n, n_cap, tuple_cap, out, stream);
    if (rc != 0) return fail(h, BFA_ERR_LAUNCH, hipGetErrorString((hipError_t)rc));
    return BFA_OK;
}

int bfa_index_records(bfa_handle h, const int32_t *records, int world, int64_t words, int n_max, int n_total,
                      int32_t *out_owner, int32_t *out_offset, int32_t *out_count, void *stream)
{
    if (!h) return BFA_ERR_INVALID_ARGUMENT;
    DeviceGuard guard(h);
    if (!records || !out_owner || !out_offset || !out_count || world <= 0 || world > 65535 || words < 8 || n_total < 0 || n_max < 0)
        return fail(h, BFA_ERR_INVALID_ARGUMENT, "bad argument");
    const int rc = bfa_launch_index_records(records, world, words, n_max, n_total, out_owner, out_offset, out_count, stream);
    if (rc != 0) return fail(h, BFA_ERR_LAUNCH, hipGetErrorString((hipError_t)rc));
    return BFA_OK;
}

int bfa_stitch_windows(bfa_handle h, const float *window_logits, int B, int NW, int F, int C, const float *weights,
                       int total_frames, float *out, int64_t out_strideB, int64_t out_strideT, void *stream)
{
    if (!h) return BFA_ERR_INVALID_ARGUMENT;
    DeviceGuard guard(h);
    if (!window_logits || !weights || !out || B < 0 || NW < 0 || F <= 0 || C <= 0 || total_frames < 0 ||
        out_strideT < C)
        return fail(h, BFA_ERR_INVALID_ARGUMENT, "bad argument");
    const int stride = F / 2;
    // the reference raises a shape error when a window other than the last does not fit (windowing.py:144-149)
    if (NW >= 2 && (int64_t)(NW - 2) * stride + F > total_frames)
        return fail(h, BFA_ERR_INVALID_ARGUMENT, "a full window does not fit into total_frames");
    if (NW > 0 && total_frames > 0 && (int64_t)(NW - 1) * stride >= total_frames)
        return fail(h, BFA_ERR_INVALID_ARGUMENT, "the last window starts beyond total_frames");
    if (B == 0 || total_frames == 0) return BFA_OK;
    const int rc = bfa_launch_stitch(window_logits, B, NW, F, C, weights, total_frames, out, out_strideB, out_strideT, stream);
    if (rc != 0) return fail(h, BFA_ERR_LAUNCH, hipGetErrorString((hipError_t)rc));
    return BFA_OK;
}

} // extern "C"
