/*
 * oracle/bfa_oracle.h -- TEST INFRASTRUCTURE ONLY.
 *
 * Scalar CPU restatement of the reference hot path
 *   /root/reference/bournemouth_aligner/forced_alignment.py   (ViterbiDecoder, AlignmentUtils)
 *   /root/reference/bournemouth_aligner/utils.py:70-149        (_calculate_confidences, convert_to_ms)
 *   /root/reference/bournemouth_aligner/core.py:462-809        (ensure_target_coverage default path,
 *                                                               extend_soft_boundaries_func)
 *
 * Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may load this library.
 * The product (bournemouth-forced-aligner_amd/) never links, imports or calls it.
 *
 * Parity pin: checked bit-exact against golden vectors generated from the reference itself
 * (tests/golden/make_golden.py, run in the build container where /root/reference exists).
 */
#ifndef BFA_ORACLE_H
#define BFA_ORACLE_H
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define ORA_OK 0
#define ORA_ERR_TOO_SHORT 1 /* forced_alignment.py:161-165 ValueError */
#define ORA_ERR_ALLOC 2
#define ORA_ERR_ARG 3       /* an index the reference would raise IndexError on */

/* decode modes reported by ora_decode_forced */
#define ORA_MODE_EMPTY 0        /* S == 0                       forced_alignment.py:112-118 */
#define ORA_MODE_SEGMENTED 1    /* silence-anchored             forced_alignment.py:133-145 */
#define ORA_MODE_STANDARD 2     /* single banded Viterbi        forced_alignment.py:153-193 */
#define ORA_MODE_PROPORTIONAL 3 /* T == S.. no DP               forced_alignment.py:166-176 */

typedef struct {
    int32_t blank_id;
    int32_t silence_id;      /* < 0 : None */
    int32_t silence_anchors; /* 0 disables segmented mode (forced_alignment.py:902) */
    int32_t ignore_noise;
    int32_t truly_forced;
    int32_t boost_targets;
    int32_t enforce_minimum;
    float min_log_prob;      /* float32 torch.log(torch.tensor(min_phoneme_prob)), forced_alignment.py:70 (default log 1e-8) */
} ora_params;

/* torch CPU (AVX512 dispatch) numerics, restated: Sleef expf/logf u10 and the
 * vectorised log_softmax reduction order. */
float ora_expf_u10(float d);
float ora_strided_mean(const float *lp, long ldT, int s, int e, int ph); /* torch's float32 .mean() of exp(lp)[s:e, ph] (core.py:711) */
float ora_exp_cr(float x); /* restatement of torch.exp (float32 CPU): float64 exp rounded once */
void ora_exp_cr_arr(const float *x, float *y, long n);
float ora_logf_u10(float d);
void ora_expf_u10_arr(const float *x, float *y, long n);
void ora_logf_u10_arr(const float *x, float *y, long n);
void ora_log_softmax_rows(const float *x, long ldx, float *out, long ldo, long T, int C);

/* forced_alignment.py:563-703 */
int ora_viterbi(const float *lp, long ldT, int T, int C, const int32_t *path, const int32_t *pidx, int L,
                int band_width, int truly_forced, int blank, int pace_f32, int32_t *frame_ph,
                int32_t *frame_idx, int32_t *states_out, float *final_dp_out);

/* the same, also handing out the codes K[T][L] and the first all-dead frame (tests of the dead-tail closed form) */
int ora_viterbi_trace(const float *lp, long ldT, int T, int C, const int32_t *path, const int32_t *pidx, int L,
                      int band_width, int truly_forced, int blank, int pace_f32, int32_t *frame_ph,
                      int32_t *frame_idx, int32_t *states_out, float *final_dp_out, uint8_t *K_out, int32_t *t_dead_out);
/* closed form of the codes in the dead sentinel regime (derived from forced_alignment.py:608-653; see the .c) */
int ora_dead_tail_codes(const float *lp, long ldT, int T, int C, const int32_t *path, int L, int band_width,
                        int pace_f32, int t_from, uint8_t *K_out);

/* the banded recurrence on a sliding window of 64 RW states + the closed form outside it (see the .c) */
int ora_window_codes(const float *lp, long ldT, int T, int C, const int32_t *path, int L, int band_width, int RW,
                     uint8_t *K_out, float *final_dp_out, int32_t *min_margin_out);

/* forced_alignment.py:29-83 (boost + log_softmax + floor); out is contiguous [T,C] */
int ora_prepare_emissions(const float *lp, long ldT, int T, int C, const int32_t *seq, int S,
                          const ora_params *p, float *out);

/* forced_alignment.py:471-541 ; segs = (start,end) pairs, returns count (or -1 if cap exceeded) */
int ora_detect_silence(const float *x, long ld, int Tx, int C, int sil, double thr, int k, int32_t *segs,
                       int cap);

/* forced_alignment.py:87-199 ; modified_out (nullable) receives the [T,C] matrix after boost/floor */
int ora_decode_forced(const float *lp, long ldT, int T, int C, const int32_t *seq, int S,
                      const ora_params *p, int32_t *frame_ph, int32_t *frame_idx, int32_t *mode_out,
                      float *modified_out);

/* forced_alignment.py:777-834 ; out4 = (phoneme,start,end,target_idx) x cap ; returns count, -1 on overflow */
int ora_assort_frames(const int32_t *frame_ph, const int32_t *frame_idx, int n, int blank, int ignore_noise,
                      int max_blanks, int32_t *out4, int cap);

/* forced_alignment.py:856-910 ; seg_out [B,seg_cap,4], seg_count [B], status [B] */
int ora_decode_alignments(const float *lp, long ldB, long ldT, int B, int Tmax, int C, const int32_t *T_len,
                          const int32_t *tokens, int Smax, const int32_t *S_len, const ora_params *p,
                          int32_t *frame_ph, int32_t *frame_idx, int32_t *seg_out, int seg_cap,
                          int32_t *seg_count, int32_t *status, int32_t *mode);

/* forced_alignment.py:932-987 */
int ora_decode_alignments_simple(const float *lp, long ldB, long ldT, int B, int Tmax, int C,
                                 const int32_t *T_len, const int32_t *tokens, int Smax, const int32_t *S_len,
                                 const ora_params *p, int32_t *frame_ph, int32_t *frame_idx, int32_t *seg_out,
                                 int seg_cap, int32_t *seg_count, int32_t *status);

/* forced_alignment.py:767-773 */
double ora_alignment_score(const float *lp, long ldT, int T, int C, const int32_t *frame_ph);

/* utils.py:70-113 ; segs = (phoneme,start,end) triples (stride seg_stride int32s) ; conf[n] */
int ora_confidences(const float *lp, long ldT, int T, int C, const int32_t *segs, int seg_stride, int n,
                    float *conf, int32_t *start_out, int32_t *end_out);

/* core.py:462-679 with ensure_completeness=False: drop tuples whose target idx is -1 / >= S,
 * stable sort by start. in/out4 (ph,start,end,idx). returns new count. */
int ora_ensure_target_coverage_default(int32_t *seg4, int n, int S);

/* core.py:682-809 ; seg4 (ph,start,end,idx) updated in place ; lp is the padded [Tpad,C] item */
int ora_extend_soft_boundaries(const float *lp, long ldT, int Tpad, int C, int32_t *seg4, int n,
                               int boundary_softness);

/* utils.py:115-149 with spectral_length a 0-dim int64 tensor (core.py:941) */
void ora_convert_to_ms(const int32_t *seg4, int n, int spectral_len, double start_offset, double wav_len,
                       double sample_rate, float *start_ms, float *end_ms);

/* cupe2i/windowing.py:103-173 stich_window_predictions: cosine-weighted overlap-add of per-window outputs.
 * win [B][NW][F][C] float32 ; weights[F] = cos(linspace(-pi/2, pi/2, F)) as computed by the caller ;
 * out [B][total_frames][ld_out >= C] (columns >= C untouched).  Returns 0, or ORA_ERR_ARG when a window that is not
 * the last one does not fit into total_frames (the reference raises a shape error there). */
int ora_stitch_windows(const float *win, int B, int NW, int F, int C, const float *weights, int total_frames,
                       float *out, long ld_out);

#ifdef __cplusplus
}
#endif
#endif
