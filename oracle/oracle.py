"""oracle/oracle.py -- TEST INFRASTRUCTURE ONLY.

ctypes binding of oracle/libbfa_oracle.so (the scalar C restatement of the reference hot path,
see bfa_oracle.h).  Importable only from tests/, __graft_entry__.smoke() and bench.py's
cpu_baseline leg; the product package never imports this module.
"""
import ctypes
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_SO = os.path.join(_HERE, "libbfa_oracle.so")

OK, ERR_TOO_SHORT, ERR_ALLOC, ERR_ARG = 0, 1, 2, 3
MODE_EMPTY, MODE_SEGMENTED, MODE_STANDARD, MODE_PROPORTIONAL = 0, 1, 2, 3


class Params(ctypes.Structure):
    _fields_ = [(n, ctypes.c_int32) for n in (
        "blank_id", "silence_id", "silence_anchors", "ignore_noise", "truly_forced",
        "boost_targets", "enforce_minimum")] + [("min_log_prob", ctypes.c_float)]

MIN_LOGP = -18.420680999755859375  # float32 torch.log(torch.tensor(1e-8)), forced_alignment.py:70


def make_params(blank_id, silence_id=0, silence_anchors=10, ignore_noise=True, truly_forced=True,
                boost_targets=True, enforce_minimum=True, min_log_prob=MIN_LOGP):
    """min_log_prob: the float32 logarithm of ViterbiDecoder.min_phoneme_prob as the caller's torch computes it"""
    return Params(int(blank_id), -1 if silence_id is None else int(silence_id), int(silence_anchors),
                  int(bool(ignore_noise)), int(bool(truly_forced)), int(bool(boost_targets)),
                  int(bool(enforce_minimum)), float(min_log_prob))


def build(force=False):
    src = os.path.join(_HERE, "bfa_oracle.c")
    if force or not os.path.exists(_SO) or os.path.getmtime(_SO) < os.path.getmtime(src):
        subprocess.check_call(["make", "-C", _HERE, "-B" if force else "-s", "libbfa_oracle.so"])
    return _SO


_lib = None


def lib():
    global _lib
    if _lib is None:
        build()
        _lib = ctypes.CDLL(_SO)
        _lib.ora_alignment_score.restype = ctypes.c_double
        _lib.ora_expf_u10.restype = ctypes.c_float
        _lib.ora_expf_u10.argtypes = [ctypes.c_float]
        _lib.ora_logf_u10.restype = ctypes.c_float
        _lib.ora_logf_u10.argtypes = [ctypes.c_float]
    return _lib


def _p(a):
    return a.ctypes.data_as(ctypes.c_void_p)


def _f32(a):
    return np.ascontiguousarray(a, dtype=np.float32)


def _i32(a):
    return np.ascontiguousarray(a, dtype=np.int32)


def expf_u10(x):
    x = _f32(x)
    y = np.empty_like(x)
    lib().ora_expf_u10_arr(_p(x), _p(y), ctypes.c_long(x.size))
    return y


def exp_cr(x):
    """the restatement of torch.exp (float32 CPU tensor): float64 exp rounded once to float32"""
    x = _f32(x)
    y = np.empty_like(x)
    lib().ora_exp_cr_arr(_p(x), _p(y), ctypes.c_long(x.size))
    return y


def logf_u10(x):
    x = _f32(x)
    y = np.empty_like(x)
    lib().ora_logf_u10_arr(_p(x), _p(y), ctypes.c_long(x.size))
    return y


def log_softmax_rows(x):
    x = _f32(x)
    T, C = x.shape
    out = np.empty_like(x)
    lib().ora_log_softmax_rows(_p(x), ctypes.c_long(C), _p(out), ctypes.c_long(C), ctypes.c_long(T), C)
    return out


def viterbi(lp, path, pidx, band_width, truly_forced, blank, pace_f32=False):
    lp = _f32(lp)
    T, C = lp.shape
    path = _i32(path)
    pidx = _i32(pidx)
    L = path.shape[0]
    fph = np.empty(T, np.int32)
    fidx = np.empty(T, np.int32)
    states = np.empty(T, np.int32)
    fdp = np.empty(L, np.float32)
    rc = lib().ora_viterbi(_p(lp), ctypes.c_long(C), T, C, _p(path), _p(pidx), L, int(band_width),
                           int(bool(truly_forced)), int(blank), int(bool(pace_f32)), _p(fph), _p(fidx),
                           _p(states), _p(fdp))
    return rc, fph, fidx, states, fdp


def viterbi_trace(lp, path, pidx, band_width, truly_forced, blank, pace_f32=False):
    """viterbi() plus the codes K [T, L] (uint8) and t_dead (first frame after which every state is <= -1000; T if never)"""
    lp = _f32(lp)
    T, C = lp.shape
    path = _i32(path)
    pidx = _i32(pidx)
    L = path.shape[0]
    fph = np.empty(T, np.int32)
    fidx = np.empty(T, np.int32)
    states = np.empty(T, np.int32)
    fdp = np.empty(L, np.float32)
    K = np.empty((T, L), np.uint8)
    td = ctypes.c_int32(0)
    rc = lib().ora_viterbi_trace(_p(lp), ctypes.c_long(C), T, C, _p(path), _p(pidx), L, int(band_width),
                                 int(bool(truly_forced)), int(blank), int(bool(pace_f32)), _p(fph), _p(fidx),
                                 _p(states), _p(fdp), _p(K), ctypes.byref(td))
    return rc, fph, fidx, states, fdp, K, int(td.value)


def dead_tail_codes(lp, path, band_width, t_from, pace_f32=False):
    """closed form of the codes of frames [t_from, T) in the dead sentinel regime (bfa_oracle.c: ora_dead_tail_codes)"""
    lp = _f32(lp)
    T, C = lp.shape
    path = _i32(path)
    L = path.shape[0]
    K = np.zeros((T, L), np.uint8)
    rc = lib().ora_dead_tail_codes(_p(lp), ctypes.c_long(C), T, C, _p(path), L, int(band_width), int(bool(pace_f32)),
                                   int(t_from), _p(K))
    return rc, K


def window_codes(lp, path, band_width, RW):
    """codes of the full DP from the sliding-window recurrence + the out-of-window closed form (ora_window_codes)"""
    lp = _f32(lp)
    T, C = lp.shape
    path = _i32(path)
    L = path.shape[0]
    K = np.zeros((T, L), np.uint8)
    fdp = np.empty(L, np.float32)
    mg = ctypes.c_int32(0)
    rc = lib().ora_window_codes(_p(lp), ctypes.c_long(C), T, C, _p(path), L, int(band_width), int(RW), _p(K), _p(fdp),
                                ctypes.byref(mg))
    return rc, K, fdp, int(mg.value)


def win_class_for(L, bw):
    """bfa_types.hpp: win_class_for without the frame limit (states per lane of the narrowest window that holds the band)"""
    if bw <= 0:
        return 0
    r = (L + 63) // 64
    rfull = next((c for c in (2, 3, 4, 6, 8, 12, 16) if r <= c), 0)
    for rw in (1, 2, 3, 4, 6, 8):
        if rfull == 0 or rw >= rfull:
            break
        fpw = 16 if rw == 1 else 8 if rw == 2 else 4
        if 2 * bw + 1 + fpw + 2 + rw + 1 <= 64 * rw:
            return rw
    return 0


def prepare_emissions(lp, seq, params):
    lp = _f32(lp)
    T, C = lp.shape
    seq = _i32(seq)
    out = np.empty((T, C), np.float32)
    rc = lib().ora_prepare_emissions(_p(lp), ctypes.c_long(C), T, C, _p(seq), seq.shape[0],
                                     ctypes.byref(params), _p(out))
    return rc, out


def detect_silence(x, sil, thr, k):
    x = _f32(x)
    T, C = x.shape
    segs = np.empty((T + 2, 2), np.int32)
    n = lib().ora_detect_silence(_p(x), ctypes.c_long(C), T, C, int(sil), ctypes.c_double(thr), int(k),
                                 _p(segs), T + 2)
    return [tuple(int(v) for v in segs[i]) for i in range(max(n, 0))]


def decode_forced(lp, seq, params, want_modified=False):
    lp = _f32(lp)
    T, C = lp.shape
    seq = _i32(seq)
    fph = np.empty(T, np.int32)
    fidx = np.empty(T, np.int32)
    mode = ctypes.c_int32(-1)
    mod = np.empty((max(T, 1), C), np.float32) if want_modified else None
    rc = lib().ora_decode_forced(_p(lp), ctypes.c_long(C), T, C, _p(seq), seq.shape[0], ctypes.byref(params),
                                 _p(fph), _p(fidx), ctypes.byref(mode), _p(mod) if want_modified else None)
    return rc, fph, fidx, mode.value, mod


def assort_frames(fph, fidx, blank, ignore_noise=True, max_blanks=10):
    fph = _i32(fph)
    fidx = _i32(fidx)
    n = fph.shape[0]
    out = np.empty((n + 1, 4), np.int32)
    cnt = lib().ora_assort_frames(_p(fph), _p(fidx), n, int(blank), int(bool(ignore_noise)), int(max_blanks),
                                  _p(out), n + 1)
    return [tuple(int(v) for v in out[i]) for i in range(cnt)]


def decode_alignments(lp, tokens, T_len, S_len, params, seg_cap=None, simple=False):
    """Batch entry point.  lp [B,Tmax,C]; tokens [B,Smax].  Returns dict of numpy arrays."""
    lp = _f32(lp)
    B, Tmax, C = lp.shape
    tokens = _i32(tokens).reshape(B, -1)
    Smax = tokens.shape[1]
    T_len = _i32(T_len)
    S_len = _i32(S_len)
    if seg_cap is None:
        seg_cap = Tmax + 1
    fph = np.empty((B, Tmax), np.int32)
    fidx = np.empty((B, Tmax), np.int32)
    seg = np.zeros((B, seg_cap, 4), np.int32)
    cnt = np.zeros(B, np.int32)
    status = np.zeros(B, np.int32)
    mode = np.full(B, -1, np.int32)
    if simple:
        rc = lib().ora_decode_alignments_simple(
            _p(lp), ctypes.c_long(Tmax * C), ctypes.c_long(C), B, Tmax, C, _p(T_len), _p(tokens), Smax,
            _p(S_len), ctypes.byref(params), _p(fph), _p(fidx), _p(seg), seg_cap, _p(cnt), _p(status))
    else:
        rc = lib().ora_decode_alignments(
            _p(lp), ctypes.c_long(Tmax * C), ctypes.c_long(C), B, Tmax, C, _p(T_len), _p(tokens), Smax,
            _p(S_len), ctypes.byref(params), _p(fph), _p(fidx), _p(seg), seg_cap, _p(cnt), _p(status), _p(mode))
    return dict(rc=rc, frame_ph=fph, frame_idx=fidx, seg=seg, seg_count=cnt, status=status, mode=mode)


def segments_as_lists(res):
    out = []
    for b in range(res["seg"].shape[0]):
        out.append([tuple(int(v) for v in res["seg"][b, i]) for i in range(int(res["seg_count"][b]))])
    return out


def alignment_score(lp, frame_ph):
    lp = _f32(lp)
    T, C = lp.shape
    frame_ph = _i32(frame_ph)
    return float(lib().ora_alignment_score(_p(lp), ctypes.c_long(C), T, C, _p(frame_ph)))


def confidences(lp, segs):
    """segs: iterable of tuples starting (phoneme, start, end, ...).  Returns (rc, conf, start, end)."""
    lp = _f32(lp)
    T, C = lp.shape
    arr = _i32([[s[0], s[1], s[2]] for s in segs]).reshape(-1, 3)
    n = arr.shape[0]
    conf = np.zeros(n, np.float32)
    st = np.zeros(n, np.int32)
    en = np.zeros(n, np.int32)
    rc = lib().ora_confidences(_p(lp), ctypes.c_long(C), T, C, _p(arr), 3, n, _p(conf), _p(st), _p(en))
    return rc, conf, st, en


def ensure_target_coverage_default(segs, S):
    arr = _i32([list(s[:4]) for s in segs]).reshape(-1, 4).copy()
    m = lib().ora_ensure_target_coverage_default(_p(arr), arr.shape[0], int(S))
    return [tuple(int(v) for v in arr[i]) for i in range(m)]


def strided_mean(lp, s, e, ph):
    """float32 mean of exp(lp)[s:e, ph] as torch computes it for that strided view (core.py:711)"""
    lp = _f32(lp)
    fn = lib().ora_strided_mean
    fn.restype = ctypes.c_float
    return np.float32(fn(_p(lp), ctypes.c_long(lp.shape[1]), int(s), int(e), int(ph)))


def extend_soft_boundaries(lp_padded, segs, boundary_softness=3):
    lp = _f32(lp_padded)
    T, C = lp.shape
    arr = _i32([list(s[:4]) for s in segs]).reshape(-1, 4).copy()
    rc = lib().ora_extend_soft_boundaries(_p(lp), ctypes.c_long(C), T, C, _p(arr), arr.shape[0],
                                          int(boundary_softness))
    assert rc == 0
    return [tuple(int(v) for v in arr[i]) for i in range(arr.shape[0])]


def convert_to_ms(segs, spectral_len, start_offset, wav_len, sample_rate):
    arr = _i32([list(s[:4]) for s in segs]).reshape(-1, 4)
    n = arr.shape[0]
    a = np.zeros(n, np.float32)
    b = np.zeros(n, np.float32)
    lib().ora_convert_to_ms(_p(arr), n, int(spectral_len), ctypes.c_double(start_offset),
                            ctypes.c_double(wav_len), ctypes.c_double(sample_rate), _p(a), _p(b))
    return a, b


def stitch_total_frames(original_audio_length, cnn_output_size, sample_rate=16000, window_size_ms=160, stride_ms=80):
    """cupe2i/windowing.py:121-126"""
    window_size_samples = int(window_size_ms * sample_rate / 1000)
    stride_samples = int(stride_ms * sample_rate / 1000)
    num_windows_total = ((original_audio_length - window_size_samples) // stride_samples) + 1
    return (num_windows_total * cnn_output_size) // 2


def stitch_windows(window_logits, weights, total_frames, ld_out=None):
    """cupe2i/windowing.py:103-173 given the cosine weights and the number of output frames."""
    w = _f32(window_logits)
    B, NW, F, C = w.shape
    ld = C if ld_out is None else int(ld_out)
    out = np.zeros((B, max(int(total_frames), 0), ld), np.float32)
    rc = lib().ora_stitch_windows(_p(w), B, NW, F, C, _p(_f32(weights)), int(total_frames), _p(out), ctypes.c_long(ld))
    return rc, out
