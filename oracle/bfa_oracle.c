/*
 * oracle/bfa_oracle.c -- TEST INFRASTRUCTURE ONLY (see bfa_oracle.h).
 *
 * Plain-C scalar restatement of the reference's forced-alignment hot path. Every function cites
 * the reference lines it follows (paths relative to /root/reference/bournemouth_aligner/).
 * Build: gcc -O2 -ffp-contract=off (no fast-math: every float op below is one IEEE rounding,
 * fused multiply-adds are explicit fmaf calls).
 */
#include "bfa_oracle.h"

#include <math.h>
#include <stdlib.h>
#include <string.h>

#define NEGF (-1000.0f)              /* forced_alignment.py:23  _neg_inf */
/* (the default floor log(1e-8) = -18.420681f is the CALLER's ora_params.min_log_prob: oracle.py MIN_LOGP) */

/* ------------------------------------------------------------------------------------------
 * torch-CPU numerics.  F.log_softmax(dim=-1) on a float32 CPU tensor (torch 2.10, AVX512
 * dispatch) is: max, Sleef_expf16_u10(x-max) summed in sixteen lane-accumulators (tail
 * elements into the low lanes), xor-butterfly 8,4,2,1, Sleef_logf16_u10 of the sum, and
 * out = (x - max) - log(sum).  Verified bit-exact on 2.7M elements (C in 16..128).
 * ---------------------------------------------------------------------------------------- */
static inline float i2f(int32_t i) { float f; memcpy(&f, &i, 4); return f; }
static inline int32_t f2i(float f) { int32_t i; memcpy(&i, &f, 4); return i; }
static inline float pow2if(int q) { return i2f((int32_t)(q + 0x7f) << 23); }

float ora_expf_u10(float d)
{
    int q = (int)rintf(d * 1.442695040888963407359924681001892137426645954152985934135449406931f);
    float s, u;
    s = fmaf((float)q, -0.693145751953125f, d);
    s = fmaf((float)q, -1.428606765330187045e-06f, s);
    u = 0.000198527617612853646278381f;
    u = fmaf(u, s, 0.00139304355252534151077271f);
    u = fmaf(u, s, 0.00833336077630519866943359f);
    u = fmaf(u, s, 0.0416664853692054748535156f);
    u = fmaf(u, s, 0.166666671633720397949219f);
    u = fmaf(u, s, 0.5f);
    u = 1.0f + fmaf(s * s, u, s);
    u = u * pow2if(q >> 1) * pow2if(q - (q >> 1));
    if (d < -104.0f) u = 0.0f;
    if (d > 100.0f) u = INFINITY;
    return u;
}

/* ------------------------------------------------------------------------------------------
 * torch.exp on a float32 CPU tensor (forced_alignment.py:503, utils.py:81, core.py:704) goes through MKL VML
 * (vsExp, high-accuracy mode), whose algorithm is not published.  Measured on 5 M log-probabilities in the build
 * container: VML equals the CORRECTLY ROUNDED float32 exponential on 98.86 % of the inputs (never more than 1 ulp
 * away), Sleef's expf_u10 on 90.4 %.  So torch.exp is restated as: exp in float64 (error < 2^-56, far below the
 * float32 rounding step) rounded once to float32.  Every operation is an IEEE float64 op / explicit fma, so the HIP
 * kernels (bfa_math.hpp::exp_cr) produce the same bits.
 * ---------------------------------------------------------------------------------------- */
float ora_exp_cr(float xf)
{
    double x = (double)xf;
    if (!(x > -104.0)) return (x != x) ? xf : 0.0f;  /* below 2^-150: rounds to +0 (NaN passes through) */
    if (x > 89.0) return INFINITY;                   /* above FLT_MAX */
    double kd = rint(x * 0x1.71547652b82fep+0);      /* x / ln 2, nearest-even */
    double r = fma(kd, -0x1.62e42fefa38p-1, x);      /* ln2 hi: 42 significant bits, kd * hi is exact */
    r = fma(kd, -0x1.ef35793c7673p-45, r);           /* ln2 lo */
    double p = 1.0 / 6227020800.0;                   /* Taylor to r^13 / 13! : |r| <= 0.3466 -> rel. error < 2^-57 */
    p = fma(p, r, 1.0 / 479001600.0);
    p = fma(p, r, 1.0 / 39916800.0);
    p = fma(p, r, 1.0 / 3628800.0);
    p = fma(p, r, 1.0 / 362880.0);
    p = fma(p, r, 1.0 / 40320.0);
    p = fma(p, r, 1.0 / 5040.0);
    p = fma(p, r, 1.0 / 720.0);
    p = fma(p, r, 1.0 / 120.0);
    p = fma(p, r, 1.0 / 24.0);
    p = fma(p, r, 1.0 / 6.0);
    p = fma(p, r, 0.5);
    p = fma(p, r, 1.0);
    p = fma(p, r, 1.0);
    return (float)ldexp(p, (int)kd);
}
void ora_exp_cr_arr(const float *x, float *y, long n) { for (long i = 0; i < n; i++) y[i] = ora_exp_cr(x[i]); }

typedef struct { float x, y; } f2_t;
static inline f2_t df_mul_f(f2_t a, float b) { f2_t r; r.x = a.x * b; r.y = fmaf(a.y, b, fmaf(a.x, b, -r.x)); return r; }
static inline f2_t df_add2_ff(float a, float b) { f2_t r; r.x = a + b; float v = r.x - a; r.y = (a - (r.x - v)) + (b - v); return r; }
static inline f2_t df_div(f2_t n, f2_t d)
{
    float t = 1.0f / d.x;
    f2_t q;
    q.x = n.x * t;
    float u = fmaf(t, n.x, -q.x);
    float w = fmaf(-d.y, t, fmaf(-d.x, t, 1.0f));
    q.y = fmaf(q.x, w, fmaf(n.y, t, u));
    return q;
}
static inline f2_t df_add_22(f2_t a, f2_t b) { f2_t r; r.x = a.x + b.x; r.y = (((a.x - r.x) + b.x) + a.y) + b.y; return r; }
static inline f2_t df_add_2f(f2_t a, float b) { f2_t r; r.x = a.x + b; r.y = ((a.x - r.x) + b) + a.y; return r; }

/* positive normal inputs only (the log_softmax denominator is in [1, C]) */
float ora_logf_u10(float d)
{
    float de = d * (1.0f / 0.75f);
    float e = (float)(((f2i(de) >> 23) & 0xff) - 127);                 /* vgetexp */
    float m = i2f((f2i(d) & 0x007fffff) | 0x3f800000);                 /* vgetmant, [0.75,1.5) */
    if (m >= 1.5f) m *= 0.5f;
    f2_t ln2 = { 0.69314718246459960938f, -1.904654323148236017e-09f };
    f2_t s = df_mul_f(ln2, e);
    f2_t x = df_div(df_add2_ff(-1.0f, m), df_add2_ff(1.0f, m));
    float x2 = x.x * x.x;
    float t = 0.3027294874e+0f;
    t = fmaf(t, x2, 0.3996108174e+0f);
    t = fmaf(t, x2, 0.6666694880e+0f);
    f2_t xs = { x.x * 2.0f, x.y * 2.0f };
    s = df_add_22(s, xs);
    s = df_add_2f(s, x2 * x.x * t);
    return s.x + s.y;
}

void ora_expf_u10_arr(const float *x, float *y, long n) { for (long i = 0; i < n; i++) y[i] = ora_expf_u10(x[i]); }
void ora_logf_u10_arr(const float *x, float *y, long n) { for (long i = 0; i < n; i++) y[i] = ora_logf_u10(x[i]); }

static void log_softmax_row(const float *x, float *out, int C)
{
    float mx = x[0];
    for (int c = 1; c < C; c++) if (x[c] > mx) mx = x[c];
    float sum;
    if (C >= 16) {
        float acc[16];
        int nfull = C / 16;
        for (int j = 0; j < 16; j++) acc[j] = ora_expf_u10(x[j] - mx);
        for (int k = 1; k < nfull; k++)
            for (int j = 0; j < 16; j++) acc[j] = acc[j] + ora_expf_u10(x[k * 16 + j] - mx);
        for (int j = 0; j < C - nfull * 16; j++) acc[j] = acc[j] + ora_expf_u10(x[nfull * 16 + j] - mx);
        for (int sh = 8; sh >= 1; sh >>= 1) {
            float nxt[16];
            for (int j = 0; j < 16; j++) nxt[j] = acc[j] + acc[j ^ sh];
            memcpy(acc, nxt, sizeof acc);
        }
        sum = acc[0];
    } else { /* fewer elements than one vector: sequential */
        sum = ora_expf_u10(x[0] - mx);
        for (int c = 1; c < C; c++) sum = sum + ora_expf_u10(x[c] - mx);
    }
    float ls = ora_logf_u10(sum);
    for (int c = 0; c < C; c++) out[c] = (x[c] - mx) - ls;
}

void ora_log_softmax_rows(const float *x, long ldx, float *out, long ldo, long T, int C)
{
    float tmp[1024];
    for (long t = 0; t < T; t++) {
        if (C <= 1024) { /* allow in-place */
            memcpy(tmp, x + t * ldx, sizeof(float) * (size_t)C);
            log_softmax_row(tmp, out + t * ldo, C);
        } else {
            log_softmax_row(x + t * ldx, out + t * ldo, C);
        }
    }
}

/* ------------------------------------------------------------------------------------------
 * _viterbi_decode, forced_alignment.py:563-703
 * ---------------------------------------------------------------------------------------- */
static int viterbi_impl(const float *lp, long ldT, int T, int C, const int32_t *path, const int32_t *pidx, int L,
                        int band_width, int truly_forced, int blank, int pace_f32, int32_t *frame_ph,
                        int32_t *frame_idx, int32_t *states_out, float *final_dp_out, uint8_t *K_out, int32_t *t_dead_out)
{
    if (T <= 0 || L <= 0) return ORA_ERR_ARG;
    if (blank < 0 || blank >= C) return ORA_ERR_ARG;
    for (int s = 0; s < L; s++) if (path[s] < 0 || path[s] >= C) return ORA_ERR_ARG;

    float *dp = (float *)malloc(sizeof(float) * (size_t)L * 2);
    uint8_t *K = (uint8_t *)malloc((size_t)T * (size_t)L);
    uint8_t *can_skip = (uint8_t *)malloc((size_t)L);
    int32_t *st = (int32_t *)malloc(sizeof(int32_t) * (size_t)T);
    if (!dp || !K || !can_skip || !st) { free(dp); free(K); free(can_skip); free(st); return ORA_ERR_ALLOC; }
    float *cur = dp, *nxt = dp + L;

    /* :582,594-596 */
    for (int s = 0; s < L; s++) cur[s] = NEGF;
    cur[0] = lp[blank];
    if (L > 1) cur[1] = lp[path[1]];
    /* :599-605 */
    for (int s = 0; s < L; s++) can_skip[s] = (s >= 2 && path[s] != path[s - 2]) ? 1 : 0;
    /* :586-591 */
    int use_band = (band_width > 0 && T > 1 && L > 1);
    double pace = use_band ? (double)(L - 1) / (double)(T - 1) : 0.0;
    float pace32 = use_band ? (float)(L - 1) / (float)(T - 1) : 0.0f; /* decode_alignments_simple: 0-dim tensors */
    int t_dead = -1;

    for (int t = 1; t < T; t++) { /* :608-653 */
        const float *row = lp + (long)t * ldT;
        uint8_t *Kt = K + (size_t)t * (size_t)L;
        for (int s = 0; s < L; s++) {
            float e = row[path[s]];
            float c0 = cur[s] + e;
            float c1 = (s >= 1) ? (cur[s - 1] + e) : NEGF;
            float c2 = (s >= 2 && can_skip[s]) ? (cur[s - 2] + e) : NEGF;
            int k = 0; float best = c0;                  /* torch.argmax: first maximal index */
            if (c1 > best) { k = 1; best = c1; }
            if (c2 > best) { k = 2; best = c2; }
            nxt[s] = best;
            Kt[s] = (uint8_t)k;
        }
        if (use_band) { /* :650-653, backpointers are not masked */
            float lo, hi;
            if (pace_f32) {
                float center = (float)t * pace32;
                lo = center - (float)band_width;
                hi = center + (float)band_width;
            } else {
                double center = (double)t * pace;
                lo = (float)(center - (double)band_width);
                hi = (float)(center + (double)band_width);
            }
            for (int s = 0; s < L; s++) {
                float fs = (float)s;
                if (fs < lo || fs > hi) nxt[s] = NEGF;
            }
        }
        if (t_dead < 0) { /* first frame after which every state holds <= -1000 (then they all stay there: emissions <= 0) */
            int alive = 0;
            for (int s = 0; s < L; s++) if (nxt[s] > NEGF) { alive = 1; break; }
            if (!alive) t_dead = t;
        }
        float *tmp = cur; cur = nxt; nxt = tmp;
    }
    if (t_dead_out) *t_dead_out = (t_dead < 0) ? T : t_dead;
    if (K_out) { memset(K_out, 0, (size_t)L); memcpy(K_out + L, K + L, (size_t)(T - 1) * (size_t)L); }

    int f;
    if (!truly_forced) { /* :656-666 */
        int any = 0; f = 0; float bestv = 0.0f;
        for (int s = 0; s < L; s++) if (cur[s] > NEGF) { if (!any || cur[s] > bestv) { bestv = cur[s]; f = s; any = 1; } }
        if (!any) { f = 0; bestv = cur[0]; for (int s = 1; s < L; s++) if (cur[s] > bestv) { bestv = cur[s]; f = s; } }
    } else { /* :668-682 */
        f = L - 1;
        if (cur[f] <= NEGF && L >= 2) f = L - 2;
        if (cur[f] <= NEGF) {
            int found = -1;
            for (int s = 0; s < L; s++) if (cur[s] > NEGF) found = s;
            f = (found >= 0) ? found : (L - 1);
        }
    }
    if (final_dp_out) memcpy(final_dp_out, cur, sizeof(float) * (size_t)L);

    /* :686-692 -- backpointers[t] = s - k may be negative and wraps like a Python index */
    st[T - 1] = f;
    for (int t = T - 2; t >= 0; t--) {
        int s1 = st[t + 1];
        int b = s1 - (int)K[(size_t)(t + 1) * (size_t)L + (size_t)s1];
        if (b < 0) b += L;
        st[t] = b;
    }
    for (int t = 0; t < T; t++) { /* :695-700 */
        frame_ph[t] = path[st[t]];
        frame_idx[t] = pidx ? pidx[st[t]] : -1;
        if (states_out) states_out[t] = st[t];
    }
    free(dp); free(K); free(can_skip); free(st);
    return ORA_OK;
}

int ora_viterbi(const float *lp, long ldT, int T, int C, const int32_t *path, const int32_t *pidx, int L,
                int band_width, int truly_forced, int blank, int pace_f32, int32_t *frame_ph,
                int32_t *frame_idx, int32_t *states_out, float *final_dp_out)
{
    return viterbi_impl(lp, ldT, T, C, path, pidx, L, band_width, truly_forced, blank, pace_f32, frame_ph, frame_idx,
                        states_out, final_dp_out, NULL, NULL);
}

/* the same recurrence, additionally handing out the codes K[T][L] (k = 0 stay / 1 advance / 2 skip; row 0 is zero) and
 * t_dead = the first frame after which every state is at or below the -1000 sentinel (T if that never happens) */
int ora_viterbi_trace(const float *lp, long ldT, int T, int C, const int32_t *path, const int32_t *pidx, int L,
                      int band_width, int truly_forced, int blank, int pace_f32, int32_t *frame_ph,
                      int32_t *frame_idx, int32_t *states_out, float *final_dp_out, uint8_t *K_out, int32_t *t_dead_out)
{
    return viterbi_impl(lp, ldT, T, C, path, pidx, L, band_width, truly_forced, blank, pace_f32, frame_ph, frame_idx,
                        states_out, final_dp_out, K_out, t_dead_out);
}

/* ------------------------------------------------------------------------------------------
 * The "dead sentinel regime" of _viterbi_decode in closed form (a DERIVED property of forced_alignment.py:608-653,
 * proved against viterbi_impl by tests/test_dead_tail.py; the HIP tail kernel implements exactly this).
 *
 * Premises: emissions <= 0 (log-probabilities), every state <= -1000 after some frame t_dead, and the CTC path is such
 * that a state that can skip never follows one that can (stride 2 / 4 paths whose tokens differ from the blank id:
 * can_skip is false for every even state).  Then for every frame t >= t_dead + 2
 *     dp[t][s] = -1000                                      s cannot skip (c2 is the constant -1000 and wins or ties)
 *     dp[t][s] = in_band(t, s) ? f32(-1000 + e[t][s]) : -1000   s can skip (c1 = f32(dp[t-1][s-1] + e) with dp[t-1][s-1] = -1000)
 * i.e. a function of frame t alone, and the code of state s at frame t + 1 follows from dp[t][s], e[t+1][s]:
 *     s cannot skip:  c2 = -1000 is the maximum; k = 0 if f32(-1000 + e) == -1000, else 2 (state 0: 1 -- c1 is the constant)
 *     s can skip   :  c1 = f32(-1000 + e) is the maximum; k = 0 if f32(dp[t][s] + e) == c1, else 1
 * K_out rows [t_from, T) are written.  Returns ORA_ERR_ARG when the path does not have the structure above.
 * ---------------------------------------------------------------------------------------- */
int ora_dead_tail_codes(const float *lp, long ldT, int T, int C, const int32_t *path, int L, int band_width,
                        int pace_f32, int t_from, uint8_t *K_out)
{
    if (T <= 1 || L <= 0 || t_from < 2 || t_from > T) return ORA_ERR_ARG;
    for (int s = 0; s < L; s++) if (path[s] < 0 || path[s] >= C) return ORA_ERR_ARG;
    uint8_t *skip = (uint8_t *)malloc((size_t)L);
    if (!skip) return ORA_ERR_ALLOC;
    for (int s = 0; s < L; s++) skip[s] = (s >= 2 && path[s] != path[s - 2]) ? 1 : 0;
    for (int s = 1; s < L; s++) if (skip[s] && skip[s - 1]) { free(skip); return ORA_ERR_ARG; }
    const int use_band = (band_width > 0 && T > 1 && L > 1);
    const double pace = use_band ? (double)(L - 1) / (double)(T - 1) : 0.0;
    const float pace32 = use_band ? (float)(L - 1) / (float)(T - 1) : 0.0f;
    for (int t = t_from; t < T; t++) {
        const float *row = lp + (long)t * ldT, *prow = lp + (long)(t - 1) * ldT;
        float lo = 0.0f, hi = 0.0f; /* band of frame t-1 */
        if (use_band) {
            if (pace_f32) { float c = (float)(t - 1) * pace32; lo = c - (float)band_width; hi = c + (float)band_width; }
            else { double c = (double)(t - 1) * pace; lo = (float)(c - (double)band_width); hi = (float)(c + (double)band_width); }
        }
        uint8_t *Kt = K_out + (size_t)t * (size_t)L;
        for (int s = 0; s < L; s++) {
            const float e = row[path[s]];
            if (e > 0.0f) { free(skip); return ORA_ERR_ARG; }
            const float x = NEGF + e;
            if (!skip[s]) {
                Kt[s] = (x < NEGF) ? (uint8_t)(s >= 1 ? 2 : 1) : 0;
            } else {
                const int inb = !use_band || !((float)s < lo || (float)s > hi);
                const float d = inb ? (NEGF + prow[path[s]]) : NEGF;
                const float y = d + e;
                Kt[s] = (y < x) ? 1 : 0;
            }
        }
    }
    free(skip);
    return ORA_OK;
}

/* ------------------------------------------------------------------------------------------
 * The banded recurrence restricted to a sliding window of 64 * RW states (a DERIVED property of
 * forced_alignment.py:608-653, proved against viterbi_impl by tests/test_dead_tail.py; the HIP "exact window" consumer
 * and the walk's out-of-window rule implement exactly this).
 *
 * Premises: emissions <= 0, an active band (bw > 0, T > 1, L > 1) and L <= T (pace <= 1).  The window [base, base + 64 RW)
 * moves up in steps of RW states at the start of every row of FPW frames (FPW = 16 / 8 / 4 for RW = 1 / 2 / >= 3) until
 * base + RW > lo - 2, lo = the lower band limit of the row's first frame.  Inside the window the recurrence runs as in the
 * reference with every state outside it taken as exactly -1000 (true: such states are out of band in the previous frame).
 * Outside the window all three predecessors of a state are out of band, hence exactly -1000, in ANY regime:
 *     s cannot skip:  c2 (or, for s < 2, the masked constants) = -1000 is the maximum; k = 0 if f32(-1000 + e) == -1000,
 *                     else 2 (state 0: 1)
 *     s can skip   :  c0 = c1 = c2 = f32(-1000 + e): k = 0
 * K_out [T][L] receives every code, final_dp_out [L] the scores after the last frame.
 * ---------------------------------------------------------------------------------------- */
int ora_window_codes(const float *lp, long ldT, int T, int C, const int32_t *path, int L, int band_width, int RW,
                     uint8_t *K_out, float *final_dp_out, int32_t *min_margin_out)
{
    if (T <= 1 || L <= 1 || band_width <= 0 || L > T || RW < 1) return ORA_ERR_ARG;
    for (int s = 0; s < L; s++) if (path[s] < 0 || path[s] >= C) return ORA_ERR_ARG;
    const int FPW = (RW == 1) ? 16 : (RW == 2) ? 8 : 4, WN = 64 * RW;
    float *cur = (float *)malloc(sizeof(float) * (size_t)L * 2);
    uint8_t *skip = (uint8_t *)malloc((size_t)L);
    if (!cur || !skip) { free(cur); free(skip); return ORA_ERR_ALLOC; }
    float *nxt = cur + L;
    for (int s = 0; s < L; s++) { cur[s] = NEGF; skip[s] = (s >= 2 && path[s] != path[s - 2]) ? 1 : 0; }
    cur[0] = lp[path[0]];
    cur[1] = lp[path[1]];
    memset(K_out, 0, (size_t)L);
    const double pace = (double)(L - 1) / (double)(T - 1);
    int base = 0, margin = 1 << 30;
    for (int t = 1; t < T; t++) {
        const float *row = lp + (long)t * ldT;
        const float lo = (float)((double)t * pace - (double)band_width), hi = (float)((double)t * pace + (double)band_width);
        if (t % FPW == 0) { /* the window moves between rows only */
            const int lo4 = (int)ceilf(lo);
            while (base + RW <= lo4 - 2) base += RW;
        }
        uint8_t *Kt = K_out + (size_t)t * (size_t)L;
        for (int s = 0; s < L; s++) {
            const float e = row[path[s]];
            if (e > 0.0f) { free(cur); free(skip); return ORA_ERR_ARG; }
            if (s < base || s >= base + WN) { /* outside the window: closed form, score exactly -1000 */
                const float x = NEGF + e;
                Kt[s] = skip[s] ? 0 : ((x < NEGF) ? (uint8_t)(s >= 1 ? 2 : 1) : 0);
                nxt[s] = NEGF;
                continue;
            }
            const float d0 = cur[s];
            const float d1 = (s - 1 >= base) ? cur[s - 1] : NEGF; /* (below the window: -1000; cur[] of states that left it is stale) */
            const float d2 = (s - 2 >= base) ? cur[s - 2] : NEGF;
            const float c0 = d0 + e;
            const float c1 = (s >= 1) ? (d1 + e) : NEGF;
            const float c2 = (s >= 2 && skip[s]) ? (d2 + e) : NEGF;
            int k = 0; float best = c0;
            if (c1 > best) { k = 1; best = c1; }
            if (c2 > best) { k = 2; best = c2; }
            Kt[s] = (uint8_t)k;
            nxt[s] = ((float)s < lo || (float)s > hi) ? NEGF : best;
        }
        { /* how far the in-band states (and the two above them) stay from the window's top: the kernel's safety margin */
            const int ihi = (int)floorf(hi);
            const int m = (base + WN - 1) - ((ihi < L - 1 ? ihi : L - 1) + 2);
            if (ihi + 2 <= L - 1 + 2 && m < margin) margin = m;
        }
        for (int s = base + WN; s < L; s++) nxt[s] = NEGF;
        float *tmp = cur; cur = nxt; nxt = tmp;
        /* states entering the window at the next move must read as -1000: they do (set above); states below `base` keep
         * stale values in cur[] but are never read again (d1 / d2 guard) */
    }
    for (int s = 0; s < L; s++) if (s < base) cur[s] = NEGF;
    if (final_dp_out) memcpy(final_dp_out, cur, sizeof(float) * (size_t)L);
    if (min_margin_out) *min_margin_out = margin;
    free(cur < nxt ? cur : nxt); free(skip);
    return ORA_OK;
}

/* ------------------------------------------------------------------------------------------
 * _boost_target_phonemes + _enforce_minimum_probabilities, forced_alignment.py:29-83
 * ---------------------------------------------------------------------------------------- */
static int target_mask(const int32_t *seq, int S, int C, int blank, uint8_t *mask)
{
    memset(mask, 0, (size_t)C);
    for (int j = 0; j < S; j++) {
        int p = seq[j];
        if (p == blank || p == -100) continue;   /* :45-46 */
        if (p < 0) return ORA_ERR_ARG;           /* would wrap as a Python index; not supported */
        if (p < C) mask[p] = 1;                  /* :49 */
    }
    return ORA_OK;
}

int ora_prepare_emissions(const float *lp, long ldT, int T, int C, const int32_t *seq, int S,
                          const ora_params *p, float *out)
{
    uint8_t *mask = (uint8_t *)malloc((size_t)C);
    if (!mask) return ORA_ERR_ALLOC;
    int rc = target_mask(seq, S, C, p->blank_id, mask);
    if (rc) { free(mask); return rc; }
    for (int t = 0; t < T; t++) memcpy(out + (long)t * C, lp + (long)t * ldT, sizeof(float) * (size_t)C);
    if (p->boost_targets) { /* :41-54 */
        for (int t = 0; t < T; t++) {
            float *r = out + (long)t * C;
            for (int c = 0; c < C; c++) if (mask[c]) r[c] = r[c] + 5.0f;
        }
        ora_log_softmax_rows(out, C, out, C, T, C);
    }
    if (p->enforce_minimum) { /* :69-81 */
        for (int t = 0; t < T; t++) {
            float *r = out + (long)t * C;
            for (int c = 0; c < C; c++) if (mask[c] && r[c] < p->min_log_prob) r[c] = p->min_log_prob; /* :79-81 */
        }
    }
    free(mask);
    return ORA_OK;
}

/* ------------------------------------------------------------------------------------------
 * _detect_silence_segments, forced_alignment.py:471-541
 * torch.exp -> ora_exp_cr (correctly rounded; MKL-VML's own vsExp equals it on 98.9 % of inputs, <= 1 ulp on the rest),
 * torch.cumsum(float32) == float64 running sum rounded to float32 at each element.
 * ---------------------------------------------------------------------------------------- */
int ora_detect_silence(const float *x, long ld, int Tx, int C, int sil, double thr, int k, int32_t *segs,
                       int cap)
{
    if (sil < 0 || sil >= C) return 0; /* :497 */
    if (Tx < k) return 0;              /* :499 */
    if (k < 1) k = 1;
    float *cs = (float *)malloc(sizeof(float) * (size_t)(Tx + 1));
    if (!cs) return -1;
    double acc = 0.0;
    for (int i = 0; i < Tx; i++) {
        float p = ora_exp_cr(x[(long)i * ld + sil]);
        if (k > 1) { acc += (double)p; cs[i] = (float)acc; } else cs[i] = p;
    }
    int nwin = (k > 1) ? (Tx - k + 1) : Tx;
    float thr32 = (float)thr;
    int n = 0, in_sil = 0, start = 0;
    for (int i = 0; i < nwin; i++) {
        float avg;
        if (k > 1) {
            float lo = (i > 0) ? cs[i - 1] : 0.0f;
            avg = (cs[i + k - 1] - lo) / (float)k; /* :510 */
        } else avg = cs[i];
        int silent = avg >= thr32; /* :517 */
        if (silent && !in_sil) { in_sil = 1; start = i; }
        else if (!silent && in_sil) {
            in_sil = 0;
            int e = i + k - 1; if (e > Tx) e = Tx; /* :530-531 */
            if (e - start >= k) { if (n >= cap) { free(cs); return -1; } segs[2 * n] = start; segs[2 * n + 1] = e; n++; }
        }
    }
    if (in_sil) { /* :536-539 */
        if (Tx - start >= k) { if (n >= cap) { free(cs); return -1; } segs[2 * n] = start; segs[2 * n + 1] = Tx; n++; }
    }
    free(cs);
    return n;
}

/* build ctc_path / ctc_path_true_idx, forced_alignment.py:181-186 / 433-438 / 970-973 */
static void build_path(const int32_t *seq, int S, int idx0, int stride, int blank, int32_t *path, int32_t *pidx)
{
    int L = stride * S + 1;
    for (int s = 0; s < L; s++) { path[s] = blank; pidx[s] = -1; }
    for (int j = 0; j < S; j++) { path[1 + j * stride] = seq[j]; pidx[1 + j * stride] = idx0 + j; }
}

/* ------------------------------------------------------------------------------------------
 * _segmented_viterbi_decode, forced_alignment.py:268-469. returns 1 = result of T frames,
 * 0 = "[], []" (caller falls back), <0 = -error
 * ---------------------------------------------------------------------------------------- */
typedef struct { int a0, a1, t0, t1, is_sil; } seg_t;

static int segmented(const float *m, int T, int C, const int32_t *seq, int S, const ora_params *p,
                     int32_t *fph, int32_t *fidx)
{
    const int blank = p->blank_id, sil = p->silence_id;
    const int boundary_pad = 3, min_speech_frames = 20;
    int rc = 0;
    int32_t *groups = (int32_t *)malloc(sizeof(int32_t) * 2 * (size_t)(S + 1));
    int32_t *aud = (int32_t *)malloc(sizeof(int32_t) * 2 * (size_t)(T + 1));
    int32_t *sub = (int32_t *)malloc(sizeof(int32_t) * 2 * (size_t)(T + 8));
    int32_t *match = (int32_t *)malloc(sizeof(int32_t) * 2 * (size_t)(S + 1));
    seg_t *segs = (seg_t *)malloc(sizeof(seg_t) * (size_t)(2 * S + 4));
    float *x = NULL; int32_t *path = NULL, *pidx = NULL, *sph = NULL, *sidx = NULL;
    if (!groups || !aud || !sub || !match || !segs) { rc = -ORA_ERR_ALLOC; goto done; }

    /* _find_target_sil_groups :203-224 */
    int ng = 0;
    for (int i = 0; i < S;) {
        if (seq[i] == sil) { int st = i; while (i < S && seq[i] == sil) i++; groups[2 * ng] = st; groups[2 * ng + 1] = i; ng++; }
        else i++;
    }
    if (ng == 0) { rc = 0; goto done; } /* :293-295 */

    int mf = p->silence_anchors; /* :296-308 */
    int na = ora_detect_silence(m, C, T, C, sil, 0.9, mf, aud, T + 1);
    if (na < 0) { rc = -ORA_ERR_ALLOC; goto done; }
    if (na == 0 && S > 200) {
        double nt = 1.0 - (0.09 * (double)mf);
        if (nt < 0.05) nt = 0.05;
        na = ora_detect_silence(m, C, T, C, sil, nt, mf, aud, T + 1);
    }
    if (na == 0 && S > 200 && mf > 3) {
        mf = 3;
        na = ora_detect_silence(m, C, T, C, sil, 0.9, mf, aud, T + 1);
    }
    if (na <= 0) { rc = 0; goto done; } /* :315-320 */

    /* _match_silences :226-266 */
    int nm = 0;
    {
        int audio_idx = 0;
        for (int g = 0; g < ng; g++) {
            double tp = (double)(groups[2 * g] + groups[2 * g + 1]) / 2.0 / (double)S;
            int best = -1; double bd = INFINITY;
            for (int ai = audio_idx; ai < na; ai++) {
                double ap = (double)(aud[2 * ai] + aud[2 * ai + 1]) / 2.0 / (double)T;
                double d = fabs(tp - ap);
                if (d < bd) { bd = d; best = ai; }
                else if (d > bd) break;
            }
            if (best >= 0 && bd < 0.3) { match[2 * nm] = g; match[2 * nm + 1] = best; nm++; audio_idx = best + 1; }
        }
    }
    if (nm == 0) { rc = 0; goto done; } /* :324-325 */

    /* :328-354 */
    int ns = 0;
    {
        int pa = 0, pt = 0;
        for (int i = 0; i < nm; i++) {
            int tg0 = groups[2 * match[2 * i]], tg1 = groups[2 * match[2 * i] + 1];
            int as0 = aud[2 * match[2 * i + 1]], as1 = aud[2 * match[2 * i + 1] + 1];
            if (pa < as0 && pt < tg0) { seg_t s = { pa, as0, pt, tg0, 0 }; segs[ns++] = s; }
            else if (pa < as0) { seg_t s = { pa, as0, pt, pt, 0 }; segs[ns++] = s; }
            { seg_t s = { as0, as1, tg0, tg1, 1 }; segs[ns++] = s; }
            pa = as1; pt = tg1;
        }
        if (pa < T && pt < S) { seg_t s = { pa, T, pt, S, 0 }; segs[ns++] = s; }
        else if (pa < T) { seg_t s = { pa, T, pt, pt, 0 }; segs[ns++] = s; }
    }
    /* :357-369 merge short speech segments into the previous segment */
    {
        int nmrg = 0;
        for (int i = 0; i < ns; i++) {
            seg_t s = segs[i];
            int nf = s.a1 - s.a0, np = s.t1 - s.t0;
            if (!s.is_sil && np > 0 && nf < min_speech_frames && nmrg > 0) {
                seg_t *pv = &segs[nmrg - 1];
                pv->a1 = s.a1; pv->t1 = s.t1; pv->is_sil = 0;
            } else segs[nmrg++] = s;
        }
        ns = nmrg;
    }

    x = (float *)malloc(sizeof(float) * (size_t)(T + 8) * (size_t)C);
    path = (int32_t *)malloc(sizeof(int32_t) * (size_t)(4 * S + 2));
    pidx = (int32_t *)malloc(sizeof(int32_t) * (size_t)(4 * S + 2));
    sph = (int32_t *)malloc(sizeof(int32_t) * (size_t)(T + 8));
    sidx = (int32_t *)malloc(sizeof(int32_t) * (size_t)(T + 8));
    if (!x || !path || !pidx || !sph || !sidx) { rc = -ORA_ERR_ALLOC; goto done; }

    /* :377-451 -- concatenation goes straight into fph/fidx, truncated at T (:465-467) */
    long w = 0;
    int any_piece = 0;
    for (int i = 0; i < ns; i++) {
        seg_t s = segs[i];
        int n = s.a1 - s.a0;
        if (n <= 0) continue;
        any_piece = 1;
        if (s.is_sil) { /* :382-397 */
            int nsil = s.t1 - s.t0;
            for (int f = 0; f < n; f++) { sph[f] = sil; sidx[f] = -1; }
            if (nsil > 0) {
                double fps = (double)n / (double)nsil;
                for (int k = 0; k < nsil; k++) {
                    int f0 = (int)((double)k * fps), f1 = (int)((double)(k + 1) * fps);
                    if (f1 > n) f1 = n;
                    for (int f = f0; f < f1; f++) sidx[f] = s.t0 + k;
                }
            }
            for (int f = 0; f < n && w < T; f++, w++) { fph[w] = sph[f]; fidx[w] = sidx[f]; }
        } else {
            int nt = s.t1 - s.t0;
            if (nt == 0) { /* :409-412 */
                for (int f = 0; f < n && w < T; f++, w++) { fph[w] = blank; fidx[w] = -1; }
                continue;
            }
            int ps = s.a0 - boundary_pad; if (ps < 0) ps = 0;     /* :401-403 */
            int pe = s.a1 + boundary_pad; if (pe > T) pe = T;
            int pad_left = s.a0 - ps;
            int Ts = pe - ps;
            memcpy(x, m + (long)ps * C, sizeof(float) * (size_t)Ts * (size_t)C);
            /* :415-419 , _anchor_silence_in_log_probs :543-561 */
            int nsub = ora_detect_silence(x, C, Ts, C, sil, 0.8, mf, sub, T + 8);
            if (nsub < 0) { rc = -ORA_ERR_ALLOC; goto done; }
            for (int q = 0; q < nsub; q++) {
                int b0 = sub[2 * q], b1 = sub[2 * q + 1];
                for (int f = b0; f < b1; f++) x[(long)f * C + blank] = x[(long)f * C + blank] + 5.0f;
                ora_log_softmax_rows(x + (long)b0 * C, C, x + (long)b0 * C, C, b1 - b0, C);
            }
            /* :422-429 */
            int stride = 4;
            if ((double)(stride * nt + 1) > (double)Ts * 0.9) stride = 3;
            if ((double)(stride * nt + 1) > (double)Ts * 0.8) stride = 2;
            int L = stride * nt + 1;
            if ((double)L > (double)Ts * 1.2) { rc = 0; goto done; }
            build_path(seq + s.t0, nt, s.t0, stride, blank, path, pidx);
            int bw = (L > 60) ? ((L / 3 > 30) ? L / 3 : 30) : 0; /* :441 */
            int vr = ora_viterbi(x, C, Ts, C, path, pidx, L, bw, p->truly_forced, blank, 0, sph, sidx, NULL, NULL);
            if (vr) { rc = -vr; goto done; }
            for (int f = 0; f < n && w < T; f++, w++) { fph[w] = sph[pad_left + f]; fidx[w] = sidx[pad_left + f]; }
        }
    }
    if (!any_piece) { rc = 0; goto done; } /* :454-455 */
    for (; w < T; w++) { fph[w] = blank; fidx[w] = -1; } /* :461-464 */
    rc = 1;
done:
    free(groups); free(aud); free(sub); free(match); free(segs);
    free(x); free(path); free(pidx); free(sph); free(sidx);
    return rc;
}

/* ------------------------------------------------------------------------------------------
 * decode_with_forced_alignment, forced_alignment.py:87-199
 * ---------------------------------------------------------------------------------------- */
int ora_decode_forced(const float *lp, long ldT, int T, int C, const int32_t *seq, int S,
                      const ora_params *p, int32_t *frame_ph, int32_t *frame_idx, int32_t *mode_out,
                      float *modified_out)
{
    const int blank = p->blank_id;
    if (S == 0) { /* :112-118 */
        for (int t = 0; t < T; t++) { frame_ph[t] = blank; frame_idx[t] = -1; }
        if (mode_out) *mode_out = ORA_MODE_EMPTY;
        return ORA_OK;
    }
    float *m = modified_out ? modified_out : (float *)malloc(sizeof(float) * (size_t)(T > 0 ? T : 1) * (size_t)C);
    if (!m) return ORA_ERR_ALLOC;
    int rc = ora_prepare_emissions(lp, ldT, T, C, seq, S, p, m); /* :121-129 */
    if (rc) goto out;

    if (p->silence_anchors > 0 && p->silence_id >= 0) { /* :133-145 */
        int sr = segmented(m, T, C, seq, S, p, frame_ph, frame_idx);
        if (sr < 0) { rc = -sr; goto out; }
        if (sr == 1) { if (mode_out) *mode_out = ORA_MODE_SEGMENTED; rc = ORA_OK; goto out; }
    }
    {
        int stride = 4; /* :153-157 */
        if (stride * S + 1 > T) stride = 3;
        if (stride * S + 1 > T) stride = 2;
        if (stride * S + 1 > T) stride = 1;
        int L = stride * S + 1;
        if (L > T) {
            if (T < S) { rc = ORA_ERR_TOO_SHORT; goto out; } /* :161-165 */
            for (int t = 0; t < T; t++) { /* :170-172 */
                int fi = (int)(((long)t * (long)S) / (long)T);
                frame_ph[t] = seq[fi]; frame_idx[t] = fi;
            }
            if (mode_out) *mode_out = ORA_MODE_PROPORTIONAL;
            rc = ORA_OK; goto out;
        }
        int32_t *path = (int32_t *)malloc(sizeof(int32_t) * (size_t)L * 2);
        if (!path) { rc = ORA_ERR_ALLOC; goto out; }
        build_path(seq, S, 0, stride, blank, path, path + L);
        int bw = (L > 60) ? ((L / 4 > 20) ? L / 4 : 20) : 0; /* :190 */
        rc = ora_viterbi(m, C, T, C, path, path + L, L, bw, p->truly_forced, blank, 0, frame_ph, frame_idx, NULL, NULL);
        free(path);
        if (mode_out) *mode_out = ORA_MODE_STANDARD;
    }
out:
    if (!modified_out) free(m);
    return rc;
}

/* ------------------------------------------------------------------------------------------
 * assort_frames, forced_alignment.py:777-834
 * ---------------------------------------------------------------------------------------- */
int ora_assort_frames(const int32_t *ph, const int32_t *idx, int n, int blank, int ignore_noise,
                      int max_blanks, int32_t *out4, int cap)
{
    int cnt = 0;
    int i = 0;
    while (i < n) {
        int j = i + 1;
        while (j < n && ph[j] == ph[i] && idx[j] == idx[i]) j++; /* :798-801 */
        int sp = ph[i], si = idx[i];
        if (si == -1) for (int q = i; q < j; q++) if (idx[q] != -1) { si = idx[q]; break; } /* :812-816 (no-op: run is constant) */
        int emit = 0;
        if (sp == blank) { /* :819-827 */
            if (!ignore_noise && (j - i) > max_blanks) emit = 1;
        } else emit = 1; /* :830-831 */
        if (emit) {
            if (cnt >= cap) return -1;
            out4[4 * cnt + 0] = sp; out4[4 * cnt + 1] = i; out4[4 * cnt + 2] = j; out4[4 * cnt + 3] = si;
            cnt++;
        }
        i = j;
    }
    return cnt;
}

/* ------------------------------------------------------------------------------------------
 * AlignmentUtils.decode_alignments, forced_alignment.py:856-910
 * ---------------------------------------------------------------------------------------- */
int ora_decode_alignments(const float *lp, long ldB, long ldT, int B, int Tmax, int C, const int32_t *T_len,
                          const int32_t *tokens, int Smax, const int32_t *S_len, const ora_params *p,
                          int32_t *frame_ph, int32_t *frame_idx, int32_t *seg_out, int seg_cap,
                          int32_t *seg_count, int32_t *status, int32_t *mode)
{
    int worst = ORA_OK;
    for (int b = 0; b < B; b++) {
        int T = T_len ? T_len[b] : Tmax; if (T > Tmax) T = Tmax; if (T < 0) T = 0; /* slicing clamps :887 */
        int S = S_len[b]; if (S > Smax) S = Smax; if (S < 0) S = 0;
        int32_t *fp = frame_ph + (long)b * Tmax, *fi = frame_idx + (long)b * Tmax;
        int32_t md = ORA_MODE_EMPTY; int rc = ORA_OK;
        seg_count[b] = 0;
        if (S == 0) { /* :894-897 : empty tensors -> assort_frames([]) == [] */
            for (int t = 0; t < Tmax; t++) { fp[t] = p->blank_id; fi[t] = -1; }
        } else {
            rc = ora_decode_forced(lp + (long)b * ldB, ldT, T, C, tokens + (long)b * Smax, S, p, fp, fi, &md, NULL);
            if (rc == ORA_OK) {
                int n = ora_assort_frames(fp, fi, T, p->blank_id, p->ignore_noise, 10, seg_out + (long)b * seg_cap * 4, seg_cap);
                if (n < 0) { rc = ORA_ERR_ARG; n = 0; }
                seg_count[b] = n;
                for (int t = T; t < Tmax; t++) { fp[t] = p->blank_id; fi[t] = -1; }
            }
        }
        if (status) status[b] = rc;
        if (mode) mode[b] = md;
        if (rc != ORA_OK && worst == ORA_OK) worst = rc;
    }
    return worst;
}

/* ------------------------------------------------------------------------------------------
 * decode_alignments_simple, forced_alignment.py:932-987 with pred_lens / true_seqs_lens given as
 * int64 tensors (core.py:1025-1033): T*0.9 and the band centre are float32 arithmetic.
 * ---------------------------------------------------------------------------------------- */
int ora_decode_alignments_simple(const float *lp, long ldB, long ldT, int B, int Tmax, int C,
                                 const int32_t *T_len, const int32_t *tokens, int Smax, const int32_t *S_len,
                                 const ora_params *p, int32_t *frame_ph, int32_t *frame_idx, int32_t *seg_out,
                                 int seg_cap, int32_t *seg_count, int32_t *status)
{
    int worst = ORA_OK;
    for (int b = 0; b < B; b++) {
        int Traw = T_len[b];
        int T = Traw; if (T > Tmax) T = Tmax;
        int S = S_len[b]; if (S > Smax) S = Smax;
        int32_t *fp = frame_ph + (long)b * Tmax, *fi = frame_idx + (long)b * Tmax;
        int rc;
        seg_count[b] = 0;
        int stride = 4; /* :963-968 : int64 tensor > (int64 tensor * python float -> float32 tensor) */
        if ((float)(stride * S + 1) > (float)Traw * 0.9f) stride = 3;
        if ((float)(stride * S + 1) > (float)Traw * 0.8f) stride = 2;
        int L = stride * S + 1;
        int32_t *path = (int32_t *)malloc(sizeof(int32_t) * (size_t)L * 2);
        if (!path) return ORA_ERR_ALLOC;
        build_path(tokens + (long)b * Smax, S, 0, stride, p->blank_id, path, path + L);
        int bw = (L > 60) ? ((L / 4 > 20) ? L / 4 : 20) : 0; /* :976 */
        rc = ora_viterbi(lp + (long)b * ldB, ldT, T, C, path, path + L, L, bw, p->truly_forced, p->blank_id, 1, fp, fi, NULL, NULL);
        free(path);
        if (rc == ORA_OK) {
            int n = ora_assort_frames(fp, fi, T, p->blank_id, p->ignore_noise, 10, seg_out + (long)b * seg_cap * 4, seg_cap);
            if (n < 0) { rc = ORA_ERR_ARG; n = 0; }
            seg_count[b] = n;
            for (int t = T; t < Tmax; t++) { fp[t] = p->blank_id; fi[t] = -1; }
        }
        if (status) status[b] = rc;
        if (rc != ORA_OK && worst == ORA_OK) worst = rc;
    }
    return worst;
}

/* _calculate_alignment_score, forced_alignment.py:767-773 (python float accumulation) */
double ora_alignment_score(const float *lp, long ldT, int T, int C, const int32_t *frame_ph)
{
    double total = 0.0;
    for (int t = 0; t < T; t++) if (frame_ph[t] < C) total += (double)lp[(long)t * ldT + frame_ph[t]];
    return total;
}

/* ------------------------------------------------------------------------------------------
 * _calculate_confidences, utils.py:70-113.  `avg_confidence = probs[start, ph]` is a 0-dim VIEW:
 * `+=` and `/=` write through into probs[start, ph]; later reads (the max at :107 and later
 * tuples) see the mutated cell.  Mutated cells are kept in an override list.
 * ---------------------------------------------------------------------------------------- */
typedef struct { int f, ph; float v; } ovr_t;

static float prob_at(const float *lp, long ldT, int f, int ph, const ovr_t *ov, int nov)
{
    for (int i = nov - 1; i >= 0; i--) if (ov[i].f == f && ov[i].ph == ph) return ov[i].v;
    return ora_exp_cr(lp[(long)f * ldT + ph]);
}

static void set_ovr(ovr_t *ov, int *nov, int f, int ph, float v)
{
    for (int i = 0; i < *nov; i++) if (ov[i].f == f && ov[i].ph == ph) { ov[i].v = v; return; }
    ov[*nov].f = f; ov[*nov].ph = ph; ov[*nov].v = v; (*nov)++;
}

int ora_confidences(const float *lp, long ldT, int T, int C, const int32_t *segs, int seg_stride, int n,
                    float *conf, int32_t *start_out, int32_t *end_out)
{
    ovr_t *ov = (ovr_t *)malloc(sizeof(ovr_t) * (size_t)(n + 1));
    if (!ov) return ORA_ERR_ALLOC;
    int nov = 0, rc = ORA_OK;
    for (int i = 0; i < n; i++) {
        int ph = segs[(long)i * seg_stride + 0];
        int s = segs[(long)i * seg_stride + 1], e = segs[(long)i * seg_stride + 2];
        if (s < 0) s = 0;       /* :86 */
        if (e > T) e = T;       /* :87 */
        if (s >= T || ph < 0 || ph >= C) { rc = ORA_ERR_ARG; conf[i] = 0.0f; if (start_out) start_out[i] = s; if (end_out) end_out[i] = e; continue; } /* IndexError at :89 */
        float c = prob_at(lp, ldT, s, ph, ov, nov);
        if (s < e) { /* :93 */
            float half = c / 2.0f; /* fresh tensor :95 */
            int good = 1;
            for (int f = s + 1; f < e; f++) {
                float v = prob_at(lp, ldT, f, ph, ov, nov);
                if (v > half || v > 0.1f) { c = c + v; good++; set_ovr(ov, &nov, s, ph, c); } /* :101-103 */
            }
            if (good > 1) {
                c = c / (float)good; set_ovr(ov, &nov, s, ph, c); /* :105 */
                float mx = prob_at(lp, ldT, s, ph, ov, nov);
                for (int f = s + 1; f < e; f++) { float v = prob_at(lp, ldT, f, ph, ov, nov); if (v > mx) mx = v; }
                if (c < mx / 2.0f) c = mx; /* :108-109 rebinding, no write-through */
            }
        }
        conf[i] = c;
        if (start_out) start_out[i] = s;
        if (end_out) end_out[i] = e;
    }
    free(ov);
    return rc;
}

/* ------------------------------------------------------------------------------------------
 * ensure_target_coverage with ensure_completeness=False, core.py:462-679: drops tuples whose
 * target index is -1 or >= S (:488-513), stable sort by start (:660).
 * ---------------------------------------------------------------------------------------- */
int ora_ensure_target_coverage_default(int32_t *seg4, int n, int S)
{
    /* python: invalid_target_indices is a SET of index VALUES; every tuple whose idx is in it is dropped */
    int m = 0;
    for (int i = 0; i < n; i++) {
        int idx = seg4[4 * i + 3];
        if (idx < S && idx != -1) { if (m != i) memcpy(seg4 + 4 * m, seg4 + 4 * i, 16); m++; }
    }
    /* stable insertion sort by start */
    for (int i = 1; i < m; i++) {
        int32_t key[4]; memcpy(key, seg4 + 4 * i, 16);
        int j = i - 1;
        while (j >= 0 && seg4[4 * j + 1] > key[1]) { memcpy(seg4 + 4 * (j + 1), seg4 + 4 * j, 16); j--; }
        memcpy(seg4 + 4 * (j + 1), key, 16);
    }
    return m;
}

/* ------------------------------------------------------------------------------------------
 * extend_soft_boundaries_func, core.py:682-809.  probs[f,ph].item() is a python float (the
 * float32 value widened), thresholds are python doubles.
 * ---------------------------------------------------------------------------------------- */
/* torch.Tensor.sum() of a STRIDED float32 1-D view on the CPU (core.py:711: probs[start:end, ph].mean()) -- ATen SumKernel.cpp,
 * cascade_sum -> scalar_inner_sum -> row_sum -> multi_row_sum<ilp_factor = 4>: the elements as rows of four, four interleaved
 * float32 accumulators, moved up a level every 16 rows (level_power = max(4, ceil_log2(rows) / 4) = 4 below 2^20 rows; four
 * levels), levels folded in order, the n % 4 tail added to accumulator 0, then accumulators 1..3.  Pinned against torch
 * itself: tests/test_oracle_vs_reference_live.py::test_strided_mean_is_torchs (equal bits on every random column). */
static float ora_cascade_sum_col(const float *lp, long ldT, int s, int e, int ph)
{
    float a[4][4] = {{0}};
    const int n = e - s, rows = n / 4, full = rows & ~15;
#define EL(i) ora_exp_cr(lp[(long)(s + (i)) * ldT + ph])
    int i = 0;
    while (i < full) {
        for (int j = 0; j < 16; j++, i++)
            for (int k = 0; k < 4; k++) a[0][k] = a[0][k] + EL(4 * i + k);
        for (int j = 1; j < 4; j++) {
            for (int k = 0; k < 4; k++) { a[j][k] = a[j][k] + a[j - 1][k]; a[j - 1][k] = 0.0f; }
            if ((i & (15 << (4 * j))) != 0) break;
        }
    }
    for (; i < rows; i++)
        for (int k = 0; k < 4; k++) a[0][k] = a[0][k] + EL(4 * i + k);
    for (int j = 1; j < 4; j++)
        for (int k = 0; k < 4; k++) a[0][k] = a[0][k] + a[j][k];
    for (int el = 4 * rows; el < n; el++) a[0][0] = a[0][0] + EL(el);
    for (int k = 1; k < 4; k++) a[0][0] = a[0][0] + a[0][k];
#undef EL
    return a[0][0];
}

/* exposed for the pin test: the mean of lp's column slice as ora_extend_soft_boundaries computes it */
float ora_strided_mean(const float *lp, long ldT, int s, int e, int ph)
{
    return ora_cascade_sum_col(lp, ldT, s, e, ph) / (float)(e - s);
}

int ora_extend_soft_boundaries(const float *lp, long ldT, int Tpad, int C, int32_t *seg4, int n,
                               int boundary_softness)
{
    const double max_ext = 10.0;
    const double th1 = pow(10.0, -3.0);                       /* :699-700 : 10 ** -max(2, 7-4) */
    const double th2 = pow(10.0, -(double)boundary_softness); /* :701 */
    double *mean = (double *)malloc(sizeof(double) * (size_t)(n + 1));
    if (!mean) return ORA_ERR_ALLOC;
#define P(f, ph) ((double)ora_exp_cr(lp[(long)(f) * ldT + (ph)]))
    for (int i = 0; i < n; i++) { /* :709-714 ; tensor.mean() of float32 -> see note below */
        int ph = seg4[4 * i], s = seg4[4 * i + 1], e = seg4[4 * i + 2];
        if (s < Tpad && ph < C && s < e) {
            /* torch .mean() of the strided float32 column probs[s:e, ph]: ATen's cascade sum (ora_cascade_sum_f32), then
             * sum / n in float32; .item() widens the float32 */
            int ee = e > Tpad ? Tpad : e;
            mean[i] = (double)(ora_cascade_sum_col(lp, ldT, s, ee, ph) / (float)(ee - s));
        } else mean[i] = 0.001;
    }
    for (int i = 0; i < n; i++) { /* pass 1 :717-735 */
        int ph = seg4[4 * i], s = seg4[4 * i + 1], e = seg4[4 * i + 2];
        if (s >= Tpad || ph >= C) continue;
        int d = e - s;
        int min_start = (int)((double)s - (double)d * max_ext); if (min_start < 0) min_start = 0;
        if (i > 0) { int a = seg4[4 * (i - 1) + 2] + 10; if (a > s) a = s; if (a > min_start) min_start = a; }
        double thr = mean[i] * th1; if (thr > th1) thr = th1;
        int ns = s;
        for (int f = s - 1; f >= min_start; f--) { if (P(f, ph) >= thr) ns = f; else break; }
        seg4[4 * i + 1] = ns;
    }
    for (int i = 0; i < n; i++) { /* pass 2 :738-755 */
        int ph = seg4[4 * i], s = seg4[4 * i + 1], e = seg4[4 * i + 2];
        if (s >= Tpad || ph >= C) continue;
        int d = e - s;
        int max_end = (int)((double)e + (double)d * max_ext); if (max_end > Tpad) max_end = Tpad;
        if (i + 1 < n) { int a = seg4[4 * (i + 1) + 1] - 10; if (a > e) a = e; if (a < max_end) max_end = a; }
        double thr = mean[i] * th1; if (thr > th1) thr = th1;
        int ne = e;
        for (int f = e; f < max_end; f++) { if (P(f, ph) >= thr) ne = f + 1; else break; }
        seg4[4 * i + 2] = ne;
    }
    for (int i = 0; i < n; i++) { /* pass 3 :758-778 */
        int ph = seg4[4 * i], s = seg4[4 * i + 1];
        if (s >= Tpad || ph >= C) continue;
        int min_start = 0;
        if (i > 0) min_start = seg4[4 * (i - 1) + 2];
        if (s <= min_start) continue;
        int ns = s;
        for (int f = s - 1; f >= min_start; f--) { if (P(f, ph) >= th2) ns = f; else break; }
        seg4[4 * i + 1] = ns;
    }
    for (int i = 0; i < n; i++) { /* pass 4 :782-805 */
        int ph = seg4[4 * i], s = seg4[4 * i + 1], e = seg4[4 * i + 2];
        if (s >= Tpad || ph >= C) continue;
        int d = e - s;
        int max_end = (int)((double)e + (double)d * max_ext); if (max_end > Tpad) max_end = Tpad;
        if (i + 1 < n) { int a = seg4[4 * (i + 1) + 1]; if (a < max_end) max_end = a; }
        int ne = e;
        for (int f = e; f < max_end; f++) { if (P(f, ph) >= th2) ne = f + 1; else break; }
        seg4[4 * i + 2] = ne;
    }
#undef P
    free(mean);
    return ORA_OK;
}

/* ------------------------------------------------------------------------------------------
 * convert_to_ms, utils.py:115-149, called with spectral_length = 0-dim int64 tensor
 * (core.py:939-945): python float / int64 tensor -> float32 tensor arithmetic from there on.
 * ---------------------------------------------------------------------------------------- */
void ora_convert_to_ms(const int32_t *seg4, int n, int spectral_len, double start_offset, double wav_len,
                       double sample_rate, float *start_ms, float *end_ms)
{
    double dur = wav_len / sample_rate;                                            /* :126 python float */
    /* :127 python float / 0-dim int64 tensor -> Tensor.__rtruediv__ = tensor.reciprocal() * other (float32) */
    float dpf = spectral_len > 0 ? (1.0f / (float)spectral_len) * (float)dur : 0.0f;
    for (int i = 0; i < n; i++) {
        float ss = (float)start_offset + ((float)seg4[4 * i + 1] * dpf);           /* :141 */
        float es = (float)start_offset + ((float)seg4[4 * i + 2] * dpf);           /* :142 */
        start_ms[i] = ss * 1000.0f;                                                /* :144 */
        end_ms[i] = es * 1000.0f;
    }
}

/* ------------------------------------------------------------------------------------------
 * stich_window_predictions, cupe2i/windowing.py:103-173.  Float32 throughout, in the reference's order:
 * combined[t] (+)= x * w window by window (ascending), weight_sum[t] (+)= w, then combined / (weight_sum + 1e-8).
 * The last window is cut to the frames that still fit (:152-163); stride_frames = F // 2 (:127).
 * ---------------------------------------------------------------------------------------- */
int ora_stitch_windows(const float *win, int B, int NW, int F, int C, const float *weights, int total_frames,
                       float *out, long ld_out)
{
    if (B < 0 || NW < 0 || F <= 0 || C <= 0 || total_frames < 0 || ld_out < C) return ORA_ERR_ARG;
    const int stride = F / 2;
    for (int i = 0; i + 1 < NW; i++)
        if (i * stride + F > total_frames) return ORA_ERR_ARG; /* :144-149 would raise */
    if (NW > 0 && (NW - 1) * stride >= total_frames && total_frames > 0) return ORA_ERR_ARG; /* negative slice length */
    float *wsum = (float *)malloc(sizeof(float) * (size_t)(total_frames > 0 ? total_frames : 1));
    if (!wsum) return ORA_ERR_ALLOC;
    for (int b = 0; b < B; b++) {
        float *ob = out + (long)b * total_frames * ld_out;
        for (int t = 0; t < total_frames; t++) {
            wsum[t] = 0.0f;
            for (int c = 0; c < C; c++) ob[(long)t * ld_out + c] = 0.0f;
        }
        for (int i = 0; i < NW; i++) {
            const int start = i * stride;
            int nf = F;
            if (i == NW - 1 && start + F > total_frames) nf = total_frames - start;
            const float *wb = win + (((long)b * NW + i) * F) * C;
            for (int f = 0; f < nf; f++) {
                const float w = weights[f];
                float *o = ob + (long)(start + f) * ld_out;
                for (int c = 0; c < C; c++) {
                    const float prod = wb[(long)f * C + c] * w; /* full_slices[:, i] * window_weights */
                    o[c] = o[c] + prod;                          /* combined[...] += */
                }
                wsum[start + f] = wsum[start + f] + w;
            }
        }
        for (int t = 0; t < total_frames; t++) {
            const float den = wsum[t] + 1e-8f; /* python float 1e-8 added to a float32 tensor */
            for (int c = 0; c < C; c++) ob[(long)t * ld_out + c] = ob[(long)t * ld_out + c] / den;
        }
    }
    free(wsum);
    return ORA_OK;
}
