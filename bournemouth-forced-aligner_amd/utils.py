"""Host-side mirror of bournemouth_aligner/utils.py:70-149 (`_calculate_confidences`, `convert_to_ms`).

The confidence pass runs in the k_conf kernel (bfa_confidences); `convert_to_ms` is trivial host
arithmetic and reproduces the reference's float32 tensor arithmetic.
"""
import ctypes

import numpy as np
import torch

from . import _lib
from .forced_alignment import _as_i32, _device_of


def calculate_confidences_batch(log_probs, segs, seg_count, T_rows=None, row_stats=None, handle_slot=0):
    """Batch form: log_probs [B,T,C] (device), segs int32 [B,seg_cap,4], seg_count int32 [B].
    Returns (conf float32 [B,seg_cap], status int32 [B]) as device tensors; no synchronisation.
    With `row_stats` ([B,T,2] from bfa_align_heads) `log_probs` holds the RAW LOGITS instead.
    `handle_slot`: which of the process's library handles to use (the pass itself is stream-ordered on the current
    stream and keeps no state in the handle)."""
    dev = _device_of(log_probs)
    lp = log_probs.to(device=dev, dtype=torch.float32)
    if lp.stride(2) != 1:
        lp = lp.contiguous()
    B, Tmax, C = lp.shape
    segs = segs.to(device=dev, dtype=torch.int32).contiguous()
    seg_count = _as_i32(seg_count, dev)
    T_rows = _as_i32(T_rows, dev)
    seg_cap = segs.shape[1]
    conf = torch.empty((B, seg_cap), dtype=torch.float32, device=dev)   # (the kernel writes every entry: zeros beyond the count)
    status = torch.empty((B,), dtype=torch.int32, device=dev)
    L = _lib.lib()
    h = _lib.handle(dev.index if dev.index is not None else torch.cuda.current_device(), handle_slot)
    stream = torch.cuda.current_stream(dev).cuda_stream
    with torch.cuda.device(dev):
        rc = L.bfa_confidences(h, lp.data_ptr(), row_stats.data_ptr() if row_stats is not None else None,
                               lp.stride(0), lp.stride(1), B, Tmax, C,
                               T_rows.data_ptr() if T_rows is not None else None, segs.data_ptr(), seg_cap,
                               seg_count.data_ptr(), conf.data_ptr(), status.data_ptr(), stream)
    _lib.check(rc, h, "bfa_confidences")
    return conf, status


def _calculate_confidences(log_probs, framestamps):
    """utils.py:70-113.  log_probs [T, C]; framestamps: list of
    (phoneme_id, start_frame, end_frame, target_seq_idx, is_estimated).
    Returns list of (phoneme_id, start, end, target_seq_idx, is_estimated, avg_confidence)."""
    n = len(framestamps)
    if n == 0:
        return []
    T = log_probs.shape[0]
    arr = np.zeros((1, n, 4), np.int32)
    for i, fs in enumerate(framestamps):
        ph, s, e, idx, est = fs[0], int(fs[1]), int(fs[2]), fs[3], fs[4]
        s = max(0, s)
        e = min(T, e)
        if s >= T:  # utils.py:88-91
            if est:
                raise ValueError(
                    f"Invalid frame range for estimated timestamp: start_frame={s}, end_frame={e}, "
                    f"log_probs shape={tuple(log_probs.shape)}, is_estimated={est}, phoneme_id={ph}")
            raise IndexError(f"index {s} is out of bounds for dimension 0 with size {T}")
        arr[0, i] = (int(ph), s, e, int(idx))
    dev = _device_of(log_probs)
    conf, status = calculate_confidences_batch(log_probs.unsqueeze(0), torch.from_numpy(arr).to(dev),
                                               torch.tensor([n], dtype=torch.int32, device=dev))
    if int(status.cpu()[0]) != _lib.ITEM_OK:
        raise IndexError("phoneme id or start frame out of range in _calculate_confidences")
    c = conf[0, :n].cpu().numpy()
    out = []
    for i, fs in enumerate(framestamps):
        out.append((fs[0], int(arr[0, i, 1]), int(arr[0, i, 2]), fs[3], fs[4], float(c[i])))
    return out


def convert_to_ms(framestamps, spectral_length, start_offset_time, wav_len, sample_rate):
    """utils.py:115-149.  The reference is called with `spectral_length` a 0-dim int64 tensor
    (core.py:939-945), which makes everything after `duration_in_seconds` float32 tensor arithmetic;
    with a python int it is float64 arithmetic.  Both are reproduced; values are returned as python
    floats."""
    tensor_mode = isinstance(spectral_length, torch.Tensor)
    sl = int(spectral_length)
    duration_in_seconds = float(wav_len) / float(sample_rate)
    f32 = np.float32
    if tensor_mode:
        # python float / int64 tensor dispatches to Tensor.__rtruediv__ = tensor.reciprocal() * other
        dpf = (f32(1) / f32(sl)) * f32(duration_in_seconds) if sl > 0 else f32(0)
    else:
        dpf = duration_in_seconds / sl if sl > 0 else 0
    out = []
    for tup in framestamps:
        if len(tup) == 6:
            phoneme_id, start_frame, end_frame, target_seq_idx, is_estimated, avg_confidence = tup
        else:
            phoneme_id, start_frame, end_frame = tup[:3]
            target_seq_idx = tup[3] if len(tup) > 3 else -1
            is_estimated = tup[4] if len(tup) > 4 else False
            avg_confidence = tup[5] if len(tup) > 5 else 0.0
        if tensor_mode:
            off = f32(start_offset_time)
            start_ms = float((off + f32(start_frame) * dpf) * f32(1000))
            end_ms = float((off + f32(end_frame) * dpf) * f32(1000))
        else:
            start_ms = (start_offset_time + (start_frame * dpf)) * 1000
            end_ms = (start_offset_time + (end_frame * dpf)) * 1000
        out.append((phoneme_id, start_frame, end_frame, target_seq_idx, is_estimated, avg_confidence,
                    start_ms, end_ms))
    return out


def log_softmax(logits):
    """F.log_softmax(logits, dim=-1) (core.py:898-899) with the reference's CPU numerics, on the GPU."""
    dev = _device_of(logits)
    x = logits.to(device=dev, dtype=torch.float32).contiguous()
    C = x.shape[-1]
    rows = x.numel() // C
    out = torch.empty_like(x)
    L = _lib.lib()
    h = _lib.handle(dev.index if dev.index is not None else torch.cuda.current_device())
    with torch.cuda.device(dev):
        rc = L.bfa_log_softmax(h, x.data_ptr(), C, out.data_ptr(), C, rows, C,
                               torch.cuda.current_stream(dev).cuda_stream)
    _lib.check(rc, h, "bfa_log_softmax")
    return out


def postprocess_batch(log_probs, S_len, segs, seg_count, extend=True, boundary_softness=3, row_stats=None):
    """core.py:925-931 on the device, in place on `segs` / `seg_count`: ensure_target_coverage with
    ensure_completeness=False (drops target_idx -1 / >= S, stable sort by start) and, if `extend`,
    extend_soft_boundaries_func over the padded rows (bfa_postprocess).  With `row_stats` ([B,T,2] from
    bfa_align_heads) `log_probs` holds the RAW LOGITS instead."""
    dev = _device_of(log_probs)
    lp = log_probs.to(device=dev, dtype=torch.float32)
    if lp.stride(2) != 1:
        lp = lp.contiguous()
    B, Tmax, C = lp.shape
    assert segs.is_cuda and segs.dtype == torch.int32 and segs.is_contiguous()
    assert seg_count.is_cuda and seg_count.dtype == torch.int32
    S_len = _as_i32(S_len, dev)
    L = _lib.lib()
    h = _lib.handle(dev.index if dev.index is not None else torch.cuda.current_device())
    with torch.cuda.device(dev):
        rc = L.bfa_postprocess(h, lp.data_ptr(), row_stats.data_ptr() if row_stats is not None else None,
                               lp.stride(0), lp.stride(1), B, Tmax, C, S_len.data_ptr(),
                               segs.data_ptr(), segs.shape[1], seg_count.data_ptr(), int(bool(extend)),
                               int(boundary_softness), torch.cuda.current_stream(dev).cuda_stream)
    _lib.check(rc, h, "bfa_postprocess")
    return segs, seg_count
