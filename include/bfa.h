/*
 * include/bfa.h -- C-ABI of the MI355X-native forced-alignment core (libbfa_hip.so).
 *
 * Drop-in boundary for ONE hot path of tabahi/bournemouth-forced-aligner: the Viterbi forced
 * alignment over the [T x C] phoneme-posterior matrix and the per-phoneme confidence pass.
 * The reference has no FFI for this path -- it is a Python class boundary:
 *
 *   bournemouth_aligner/core.py:23       from .forced_alignment import AlignmentUtils
 *   bournemouth_aligner/core.py:252-257  _setup_decoders() builds alignment_utils_p / _g
 *   bournemouth_aligner/core.py:902-922  .decode_alignments(...)          -> bfa_align_batch
 *   bournemouth_aligner/core.py:1028     .decode_alignments_simple(...)   -> bfa_align_batch (params.simple=1)
 *   bournemouth_aligner/core.py:936-937  utils._calculate_confidences     -> bfa_confidences
 *   bournemouth_aligner/core.py:925-931  ensure_target_coverage (default) +
 *                                        extend_soft_boundaries_func      -> bfa_postprocess
 *
 * so these entry points are what a ctypes binding inside the reference's forced_alignment.py
 * would call (INTEGRATION.md shows that stub).  Plain pointers and sizes only; no torch types.
 *
 * Conventions
 *   - every data pointer is a DEVICE pointer (HBM) unless the name ends in _host;
 *   - the caller owns every buffer, including outputs and the workspace (size it with
 *     bfa_workspace_bytes); the library allocates nothing per call and never frees caller memory;
 *   - calls are stream-ordered and return without synchronising; buffers must stay alive until
 *     the stream has been synchronised.  Inside a call the alignment kernels of different length
 *     classes may run side by side on streams owned by the handle; they are forked from and joined
 *     back into `stream` with events, so the ordering the caller sees is that of `stream` alone, and a call can be
 *     stream-captured into a hipGraph (nothing in it synchronises or touches the host);
 *   - return value: BFA_OK or a negative bfa_status for call-level failures (bad argument,
 *     launch failure).  Per-utterance outcomes go to out_status[B] (see BFA_ITEM_*);
 *   - one handle per GPU per host thread; no hidden global state.
 */
#ifndef BFA_H
#define BFA_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif
/* the library is built with -fvisibility=hidden and an export map (csrc/bfa_exports.map): exactly the functions declared in
 * this header are visible */
#if defined(__GNUC__)
#pragma GCC visibility push(default)
#endif

#define BFA_ABI_VERSION 6 /* v6: bfa_call_counters (what a call's items did: fast windows redone, exact-first routing); v5: bfa_pack_results / bfa_pack_words / bfa_index_records (packed result records); v4: bfa_set_option; v3: bfa_params.min_log_prob (ViterbiDecoder.min_phoneme_prob), bfa_set_tail_stream removed */

typedef struct bfa_context *bfa_handle;

typedef enum {
    BFA_OK = 0,
    BFA_ERR_INVALID_ARGUMENT = -1,
    BFA_ERR_NO_DEVICE = -2,
    BFA_ERR_LAUNCH = -3,
    BFA_ERR_WORKSPACE_TOO_SMALL = -4,
    BFA_ERR_UNSUPPORTED = -5
} bfa_status;

/* per-utterance status written to out_status[b] */
#define BFA_ITEM_OK 0
#define BFA_ITEM_TOO_SHORT 1   /* T < S : reference raises ValueError("Audio too short to align ...")
                                  (forced_alignment.py:161-165) and aborts the whole call */
#define BFA_ITEM_BAD_TOKEN 2   /* token id outside [0,C) : reference raises IndexError */
#define BFA_ITEM_TOO_LARGE 3   /* CTC path longer than this build supports (32 768 states = 8 191 phonemes in one DP at stride 4) */
#define BFA_ITEM_SEG_OVERFLOW 4 /* more runs than seg_cap: frame outputs are valid, segments truncated */
#define BFA_ITEM_BAD_HINT 5     /* the class_mask hint excluded what this utterance needs: BFA_HINT_NO_SILENCE_TARGETS
                                   although the target contains silence_id, or a K1 class bit that is missing */

#define BFA_HINT_NO_SILENCE_TARGETS (1 << 16)
#define BFA_HINT_UNIFORM_LENGTHS (1 << 17)

/* per-utterance decode mode written to out_mode[b] */
#define BFA_MODE_EMPTY 0        /* S == 0 -> no segments (forced_alignment.py:894-897) */
#define BFA_MODE_SEGMENTED 1    /* silence-anchored segmented Viterbi (forced_alignment.py:268-469) */
#define BFA_MODE_STANDARD 2     /* one banded Viterbi (forced_alignment.py:153-193) */
#define BFA_MODE_PROPORTIONAL 3 /* S >= T-? proportional assignment, no DP (forced_alignment.py:166-176) */

/* AlignmentUtils(...) constructor fields + decode_alignments(...) flags
 * (forced_alignment.py:841-853, 856-858) */
typedef struct {
    int32_t blank_id;        /* CTC blank ("noise"): 66 for the ph66 head, 16 for the group head */
    int32_t silence_id;      /* SIL: 0 ; negative = None */
    int32_t silence_anchors; /* default 10 ; 0 disables the segmented mode */
    int32_t ignore_noise;    /* default 1 */
    int32_t truly_forced;    /* default 1 (enforce_all_targets) */
    int32_t boost_targets;   /* default 1 : +5.0 on target columns, then log_softmax */
    int32_t enforce_minimum; /* default 1 : floor target columns at log(1e-8) */
    int32_t simple;          /* 1 = decode_alignments_simple semantics (forced_alignment.py:932-987) */
    int32_t max_blanks;      /* assort_frames(max_blanks=10) */
    /* ---- optional host hints (ABI v2 names; v1 carried them as reserved[0..2]).  0 = library default. ---- */
    int32_t class_mask;        /* which K1 kernel classes are worth launching (0 = derive everything from the tensor
                                  shapes):
                                  bits 0-6   full-layout states-per-lane classes {2,3,4,6,8,12,16};
                                  bits 8-15  sliding-window ("fast window") classes Rw in {1,2,3,4,6,8} at bit 7+Rw: in-band
                                             states only; the result stands while the path score stays above the -1000
                                             sentinel, otherwise the utterance is redone (no hint bit needed for that);
                                  bit 16     BFA_HINT_NO_SILENCE_TARGETS: no target contains silence_id, so the
                                             silence-anchored planning kernels are not launched;
                                  bit 17     BFA_HINT_UNIFORM_LENGTHS: the utterances of this call have about the same
                                             number of frames.  Results are the same either way, but it SELECTS KERNELS:
                                             with it, each class runs as its own kernel and each XCD takes one contiguous
                                             eighth of the batch (right for the headline batch: 0.33 + 0.06 ms); without it a
                                             call of two utterances or more on the head widths (C = 67 / 17, default flags,
                                             no silence anchoring) is taken as a MIXED-LENGTH call: one kernel aligns and walks
                                             every utterance, longest first (right for T ~ U[200, 3000]: 1.1 against 1.6 ms per
                                             4096 utterances in round 4, 0.9 ms since round 5; on a uniform batch it costs ~10 %: the headline batch
                                             0.45 against 0.41 ms per step).  A caller whose
                                             lengths live on the device and are known to be uniform should set it; T_len ==
                                             NULL (every utterance has Tmax frames) sets it implicitly;
                                  bits 20-27 exact-window classes Rw in {1,2,3,4,6,8} at bit 19+Rw: in-band states computed
                                             exactly in every regime (utterances with more frames / tokens than the fast
                                             window is tried on, stride >= 3; in a mixed-length call every stride >= 3
                                             window item of the classes Rw <= 4 -- there a fast-window bit stands for its
                                             exact twin).
                                  "Hinted" = any of bits 0-15 / 20-27 set.  A small call (B <= 1024, 4 * Smax + 1 <= 256) whose
                                  hint names exactly ONE fast-window class Rw <= 3 and nothing else runs as one kernel per
                                  utterance (plan + DP + rerun + walk).  bfa_call_path reports which layout a call takes.
                                  A hint that excludes what an utterance needs is reported as BFA_ITEM_BAD_HINT. */
    int32_t window_max_tokens; /* 0 = 64.  K1's sliding-window variant is exact only while the path score stays above
                                  the reference's -1000 sentinel; otherwise the utterance is redone with the full state
                                  layout.  Every token costs the path a frame in a blank state, so utterances with more
                                  tokens than this are not tried in the window at all. */
    int32_t window_max_frames; /* 0 = 1536: likewise for long utterances (scores are sums of per-frame log-probs). */
    /* ---- ViterbiDecoder(min_phoneme_prob=1e-8) (forced_alignment.py:16-20): the floor _enforce_minimum_probabilities
     * puts under the target columns is torch.log(torch.tensor(min_phoneme_prob)) in float32 (forced_alignment.py:70).
     * The caller passes that float32 LOGARITHM (its own torch computes it, so the bits are the reference's);
     * has_min_log_prob = 0 selects the default log(1e-8) = -18.420681f. ---- */
    int32_t has_min_log_prob;
    float min_log_prob;
} bfa_params;

/* one aligned run: assort_frames tuple (phoneme_id, start_frame, end_frame, target_seq_idx),
 * end exclusive (forced_alignment.py:827,831) */
typedef struct {
    int32_t phoneme;
    int32_t start;
    int32_t end;
    int32_t target_idx;
} bfa_segment;

const char *bfa_version(void);
int bfa_abi_version(void);

/* device < 0 : current HIP device.  Fails with BFA_ERR_NO_DEVICE when no GPU is present --
 * there is no CPU fallback behind this ABI. */
int bfa_create(bfa_handle *out, int device);
int bfa_destroy(bfa_handle h);
const char *bfa_last_error(bfa_handle h);
void bfa_params_default(bfa_params *p, int blank_id, int silence_id);

/* bytes of device scratch bfa_align_batch needs for these shapes (shape-only upper bound, so no
 * host knowledge of the per-utterance lengths is required) */
size_t bfa_workspace_bytes(int B, int Tmax, int Smax, int C, const bfa_params *p);

/* Diagnostics of the LAST bfa_align_batch / bfa_align_heads call that used `workspace` (same shapes and params as that call):
 * copies sixteen int32 device counters to `out_host` and waits for `stream`.  [0] work items (utterances + silence-anchored
 * pieces); [2] sliding-window items that gave up (dead or doomed at the reference's -1000 sentinel) and were redone with the
 * full layout; [3] the same, redone with the exact window; standard mode: [4] items the exact rerun kernels aligned (those
 * of [3] plus the ones BFA_OPT_WINDOW_ROUTING handed over at once), [5] how many of them ended above the sentinel;
 * silence-anchored mode: [4..15] piece counts per length bucket.  Not on any hot path. */
int bfa_call_counters(bfa_handle h, const void *workspace, int B, int Tmax, int Smax, int C, const bfa_params *p,
                      int32_t *out_host, void *stream);

/* Which launch layout bfa_align_batch takes for these shapes and hints (host arithmetic only, no device needed):
 * 0 = one kernel per class side by side, 1 = the one-kernel mixed-length path (plus class kernels for the wide classes),
 * 2 = the one-kernel path of small single-class calls.  has_T_len = 0 means T_len == NULL. */
#define BFA_PATH_CLASS_KERNELS 0
#define BFA_PATH_MIXED 1
#define BFA_PATH_ONE_KERNEL 2
int bfa_call_path(int B, int Tmax, int Smax, int C, const bfa_params *p, int has_T_len);

/*
 * AlignmentUtils.decode_alignments / decode_alignments_simple for a whole batch.
 *   logp        [B,Tmax,C] float32, element (b,t,c) at logp[b*strideB + t*strideT + c]
 *   T_len       [B] int32 pred_lens (clamped to Tmax like the reference's slicing), NULL = Tmax
 *   tokens      [B,Smax] int32 target ids (padding beyond S_len[b] is ignored)
 *   S_len       [B] int32 true_seqs_lens
 * outputs (any of frame/mode pointers may be NULL)
 *   out_frame_phoneme / out_frame_idx  [B,Tmax] int32 framewise assignment (blank / -1 beyond T_len)
 *   out_segs    [B,seg_cap] bfa_segment ; out_seg_count [B] ; out_status [B] ; out_mode [B]
 */
int bfa_align_batch(bfa_handle h, const float *logp, int64_t strideB, int64_t strideT, int B, int Tmax, int C,
                    const int32_t *T_len, const int32_t *tokens, const int32_t *S_len, int Smax,
                    const bfa_params *params, int32_t *out_frame_phoneme, int32_t *out_frame_idx,
                    bfa_segment *out_segs, int seg_cap, int32_t *out_seg_count, int32_t *out_status,
                    int32_t *out_mode, void *workspace, size_t workspace_bytes, void *stream);

/*
 * The prepared emissions the DP consumes -- the reference's `modified_log_probs`
 * (forced_alignment.py:121-129: _boost_target_phonemes + _enforce_minimum_probabilities) -- written to
 * out[b*out_strideB + t*out_strideT + c] for t < T_len[b].  Bit-identical to the reference's float32 values.
 * Same workspace as bfa_align_batch.  C = 67 (ph66 head) and C = 17 (group head).
 */
int bfa_prepare_emissions(bfa_handle h, const float *logp, int64_t strideB, int64_t strideT, int B, int Tmax, int C,
                          const int32_t *T_len, const int32_t *tokens, const int32_t *S_len, int Smax,
                          const bfa_params *params, float *out, int64_t out_strideB, int64_t out_strideT,
                          void *workspace, size_t workspace_bytes, void *stream);

/*
 * utils._calculate_confidences (utils.py:70-113) for a whole batch, including the reference's
 * in-place aliasing of probs[start, phoneme].  logp is the ORIGINAL (un-boosted) matrix; T_rows[b]
 * is log_probs.shape[0] of item b as the reference passes it (the padded Tmax, core.py:936).
 *   row_stats  NULL: logp holds log-probabilities.  Otherwise logp holds the RAW LOGITS the matrix was made from and
 *              row_stats the [B,Tmax,2] row statistics bfa_align_heads wrote for them (see there; rows the
 *              alignment never prepared get their statistics filled in here, so the buffer is read-write).
 *   segs [B,seg_cap] ; seg_count [B] ; out_conf [B,seg_cap] float32 ;
 *   out_item_status [B] (BFA_ITEM_BAD_TOKEN where the reference would raise IndexError)
 */
int bfa_confidences(bfa_handle h, const float *logp, float *row_stats, int64_t strideB, int64_t strideT, int B, int Tmax, int C,
                    const int32_t *T_rows, const bfa_segment *segs, int seg_cap, const int32_t *seg_count,
                    float *out_conf, int32_t *out_item_status, void *stream);

/*
 * Post-DP boundary stages of extract_timestamps_from_segment_batch (core.py:925-931), in place on
 * segs/seg_count: ensure_target_coverage with ensure_completeness=False (drop target_idx -1 / >= S,
 * stable sort by start; core.py:488-513,660) then, if extend != 0, extend_soft_boundaries_func
 * (core.py:682-809) over the padded Tmax rows.  row_stats as in bfa_confidences.
 */
int bfa_postprocess(bfa_handle h, const float *logp, float *row_stats, int64_t strideB, int64_t strideT, int B, int Tmax, int C,
                    const int32_t *S_len, bfa_segment *segs, int seg_cap, int32_t *seg_count, int extend,
                    int boundary_softness, void *stream);

/*
 * The front of extract_timestamps_from_segment_batch (core.py:897-922) for all heads of the model in ONE call: raw
 * logits in, F.log_softmax(dim=2) (core.py:898-899) fused into the alignment kernels' row preparation, the phoneme
 * head (C = 67) and the group head (C = 17) aligned from the same call.  No log-prob
 * matrix is ever written: K1 normalises each 16-row block in registers (bit-identical to torch's CPU log_softmax),
 * goes on to boost / floor / DP with it, and leaves the row's (maximum, log-sum) pair in out_row_stats, from which
 * bfa_confidences / bfa_postprocess reconstitute the few log-probs they touch as (x - max) - logsum -- the same two
 * float32 subtractions.  Heads after the first run on a stream of the handle, forked from and joined into `stream`
 * (together with their optional post-DP stages, see the last fields of bfa_head).
 * T_len / S_len are shared by the heads (core.py:874-876); everything else is per head.
 */
typedef struct {
    const float *logits;        /* [B,Tmax,C] raw model outputs, element (b,t,c) at b*strideB + t*strideT + c */
    int64_t strideB, strideT;
    int32_t C;
    int32_t Smax;
    const int32_t *tokens;      /* [B,Smax] target ids of this head */
    bfa_params params;          /* blank / silence ids of this head, flags, hints */
    float *out_row_stats;       /* [B,Tmax,2] float32: (row maximum, log of the row's exp-sum) of log_softmax */
    int32_t *out_frame_phoneme; /* [B,Tmax] or NULL (both or neither) */
    int32_t *out_frame_idx;
    bfa_segment *out_segs;      /* [B,seg_cap] */
    int32_t seg_cap;
    int32_t *out_seg_count, *out_status, *out_mode; /* [B] ; out_mode may be NULL */
    void *workspace;            /* bfa_workspace_bytes(B, Tmax, Smax, C, &params); one per head */
    size_t workspace_bytes;
    /* ---- optional post-DP stages of this head (core.py:925-937), enqueued right behind its alignment on the stream the
     * head runs on, so that the stages of one head overlap the alignment of the others:
     *   postprocess != 0 : bfa_postprocess(extend, boundary_softness) in place on out_segs / out_seg_count;
     *   out_conf != NULL : bfa_confidences on the resulting tuples -> out_conf [B,seg_cap] float32, out_conf_status [B]
     *                      (may be NULL); the padded Tmax rows count as in core.py:936.  All zero = alignment only. ---- */
    int32_t postprocess, extend, boundary_softness;
    float *out_conf;
    int32_t *out_conf_status;
} bfa_head;

int bfa_align_heads(bfa_handle h, const bfa_head *heads, int n_heads, int B, int Tmax, const int32_t *T_len,
                    const int32_t *S_len, void *stream);

/*
 * Handle options (none changes results).
 *   BFA_OPT_CALLS_IN_FLIGHT  0 (default): the caller issues one bfa_align_heads call at a time, like the reference's loop
 *       (core.py:897-937) -- the heads of a call run on two library streams of the caller's priority, i.e. on two
 *       hardware queues even with the runtime's default of four, and share the machine from the first kernel on.
 *       1: the caller keeps several calls in flight on several handles / streams (BatchesInFlight): head 0 stays on the
 *       caller's stream, the others go to one low-priority side stream -- fewer streams per call, the calls overlap each other.
 *   BFA_OPT_WINDOW_ROUTING  1 (default): standard-mode calls on the head widths go by the handle's recent calls.  The fast
 *       sliding window of K1 only stands when an utterance ends above the reference's -1000 sentinel
 *       (forced_alignment.py:23,656-682); posteriors that lose ~1 log-unit per frame end every utterance of ~1000 frames below
 *       it, and each call would pay a fast attempt plus the rerun by the exact window.  The last kernel of a call leaves its
 *       window statistics in host-mapped memory of the handle; the following calls read what has landed (no
 *       synchronisation, possibly a call or two late) and, when one in 64 or more of a call's fast windows gave up (the
 *       rerun behind them is a second serial chain however few they are), hand every window item to the exact window at once
 *       -- until 99 in 100 of a routed call's items end above the sentinel again.  0: never (a fast attempt first, always), 2: always the exact window first.  Results are identical in all three.
 *   BFA_OPT_WIDE_ANY_MAX_BATCH  512 (default; 256 until the end of round 6, profiles/r06_layout_threshold_ab.txt): the silence-anchored mode aligns the PIECES of the wide CTC-path classes (more
 *       than 192 states) in one kernel with two consumer waves per piece, longest first; calls of at most this many utterances
 *       hand it their utterance slots (standard-mode fallbacks) as well instead of a kernel per class -- on an empty machine
 *       the launches are the cost, on a full one the registers of the widest class are.  < 0: a kernel per class and item
 *       kind for both (the layout of rounds 2-5, kept for A/B).
 *   BFA_OPT_PRECREATE_STREAMS  value bit 0: create the handle's auxiliary streams now, bit 1: the streams of bfa_align_heads.  A
 *       handle creates them when a call first needs them (bfa_create 22 ms -> 0.02 ms); a caller that keeps several calls in
 *       flight on several handles wants them up front: their existence changes how the runtime spreads the CALLER's streams
 *       over its hardware queues (four headline batches in flight: 0.40 against 0.43 ms per step).
 */
#define BFA_OPT_CALLS_IN_FLIGHT 1
#define BFA_OPT_WINDOW_ROUTING 2
#define BFA_OPT_WIDE_ANY_MAX_BATCH 3
#define BFA_OPT_PRECREATE_STREAMS 4
int bfa_set_option(bfa_handle h, int option, int value);

/*
 * Measurement hooks (bench.py): with on = n >= 1, every n-th bfa_align_batch call brackets its K1 launches
 * (the banded-Viterbi forward kernel) with a pair of HIP events on the caller's stream (n > 1 samples);
 * on = 0 switches it off.
 * bfa_profile_collect synchronises on the recorded events, writes up to `cap` K1 durations in
 * milliseconds (oldest first), clears the list and returns how many were written.
 */
int bfa_profile_enable(bfa_handle h, int on);
int bfa_profile_collect(bfa_handle h, float *out_ms_host, int cap);
/*
 * The same brackets as intervals: start / end of every recorded K1 bracket in milliseconds after `base_event` (a
 * timing-enabled hipEvent_t the caller recorded on this device before the calls).  With several calls in flight on
 * different streams (one handle each) the K1 launches of different calls overlap; the union of the intervals of all
 * handles is the time the kernel was running at all, which is what a throughput figure has to be priced against.
 */
int bfa_profile_collect_spans(bfa_handle h, void *base_event, float *out_start_ms_host, float *out_end_ms_host, int cap);
/*
 * The copy ceiling of this GPU, for pricing the roofline against what the memory system delivers rather than the 8 TB/s of
 * the data sheet: one launch of a float4 copy kernel dst[i] = src[i] over `bytes` (a multiple of
 * 16; 2 x bytes of HBM traffic) on `stream`.  The caller times it (events on that stream).
 */
int bfa_profile_copy(bfa_handle h, void *dst, const void *src, size_t bytes, void *stream);

/*
 * Window stitching, the step in front of the path: stich_window_predictions
 * (bournemouth_aligner/cupe2i/windowing.py:103-173, called at core.py:422-438): cosine-weighted overlap-add of
 * per-window outputs window_logits [B, NW, F, C] into out [B, total_frames, C] with the caller's row / batch
 * strides (in floats; out_strideT >= C, e.g. padded rows).  weights [F] (device) = cos(linspace(-pi/2, pi/2, F)) as the
 * caller's torch computes it; total_frames as in windowing.py:121-126.  Bit-identical to the reference.
 */
int bfa_stitch_windows(bfa_handle h, const float *window_logits, int B, int NW, int F, int C, const float *weights,
                       int total_frames, float *out, int64_t out_strideB, int64_t out_strideT, void *stream);

/* F.log_softmax(dim=-1) of raw logits [rows,C] (core.py:898-899), torch-CPU-exact numerics */
int bfa_log_softmax(bfa_handle h, const float *logits, int64_t ld_in, float *out, int64_t ld_out, int64_t rows,
                    int C, void *stream);

/*
 * Packed result records (ABI v5).  The tuples of a call sit in a padded [n, seg_cap] array; what LEAVES the GPU -- the copy
 * to the host behind AlignmentUtils.decode_alignments' list of lists (forced_alignment.py:871,908), the final gather of a
 * batch sharded over several GPUs ("one collective at the end ... of per-utterance result records")
 * -- needs the valid tuples only.  bfa_pack_results writes them back to back (CSR) into ONE caller-owned int32 buffer `out`
 * of bfa_pack_words(n_cap, tuple_cap, conf != NULL) words, in one kernel launch and without host knowledge of the counts:
 *
 *   word 0 n | 1 total tuples (<= tuple_cap) | 2 n_cap | 3 tuple_cap | 4 has_conf | 5 overflow (tuples cut at tuple_cap) | 6-7 zero
 *   gidx  [n_cap]  global index of utterance j: global_index[j], or gidx_base + j when global_index is NULL; -1 for j >= n
 *   count [n_cap]  tuples of utterance j (min(seg_count[j], seg_cap))
 *   offset[n_cap]  exclusive prefix sum of count
 *   tuples[tuple_cap] bfa_segment (16-byte aligned), utterance after utterance
 *   conf  [tuple_cap] float32, only when conf != NULL
 * (each of the three [n_cap] tables starts at a multiple of four words.)  n_cap >= n and tuple_cap are the CALLER's bounds:
 * ranks that exchange records of equal size agree on them from shapes they all know (the largest shard, the largest sum of
 * target lengths), so the exchange needs no size round trip.
 */
int64_t bfa_pack_words(int n_cap, int64_t tuple_cap, int has_conf);
int bfa_pack_results(bfa_handle h, const bfa_segment *segs, int seg_cap, const int32_t *seg_count, const float *conf,
                     const int32_t *global_index, int gidx_base, int n, int n_cap, int tuple_cap, int32_t *out, void *stream);
/*
 * The same record with 8-byte tuples -- phoneme, start, end as uint16, target_idx as int16 -- for the copy to the host behind
 * decode_alignments' list of lists, where the bytes of the tuples ARE the call's host-side cost.  Only when Tmax, C and Smax
 * fit 16 bits (the caller checks: BFA_ERR_INVALID_ARGUMENT otherwise is NOT reported, the fields are simply truncated).  No
 * confidences, gidx = j.  bfa_pack16_words(n_cap, tuple_cap) words; header word 6 = 1; tuples at the same offset, 2 words each.
 */
int64_t bfa_pack16_words(int n_cap, int64_t tuple_cap);
int bfa_pack_results16(bfa_handle h, const bfa_segment *segs, int seg_cap, const int32_t *seg_count, int n, int n_cap,
                       int tuple_cap, int32_t *out, void *stream);
/*
 * The receiving side: `records` = world records of `words` int32 each, as bfa_pack_results wrote them (one per rank, e.g. the
 * output of a gather).  For every global utterance index g < n_total named by a record: owner[g] = which record,
 * offset[g] = its first tuple inside that record's tuple section, count[g].  Entries no record names are left untouched.
 */
int bfa_index_records(bfa_handle h, const int32_t *records, int world, int64_t words, int n_max, int n_total,
                      int32_t *out_owner, int32_t *out_offset, int32_t *out_count, void *stream);

#if defined(__GNUC__)
#pragma GCC visibility pop
#endif
#ifdef __cplusplus
}
#endif
#endif /* BFA_H */
