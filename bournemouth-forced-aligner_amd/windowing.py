"""Window stitching in front of the alignment path: the mirror of bournemouth_aligner/cupe2i/windowing.py
(`slice_windows` :54-79, `stich_window_predictions` :103-173 -- the reference's spelling is kept), with the
overlap-add on the device (bfa_stitch_windows).  The acoustic model between the two stays on PyTorch-ROCm."""
import math

import torch

from . import _lib


_WEIGHTS = {}


def _weights(F, dev):
    """cos(linspace(-pi/2, pi/2, F)) as float32 on `dev` (cached per frames-per-window and device)."""
    key = (int(F), str(dev))
    w = _WEIGHTS.get(key)
    if w is None:
        w = torch.cos(torch.linspace(-math.pi / 2, math.pi / 2, F)).to(dev)
        _WEIGHTS[key] = w
    return w


def slice_windows(audio_batch, sample_rate=16000, window_size_ms=160, stride_ms=80):
    """windowing.py:54-79 -- a strided view, no arithmetic: [B, 1, n] -> [B, num_windows, window_size]."""
    audio_batch = audio_batch.squeeze(1)
    window_size = int(window_size_ms * sample_rate / 1000)
    stride = int(stride_ms * sample_rate / 1000)
    return audio_batch.unfold(dimension=1, size=window_size, step=stride)


def stitch_total_frames(original_audio_length, cnn_output_size, sample_rate=16000, window_size_ms=160, stride_ms=80):
    """windowing.py:121-126"""
    window_size_samples = int(window_size_ms * sample_rate / 1000)
    stride_samples = int(stride_ms * sample_rate / 1000)
    num_windows_total = ((original_audio_length - window_size_samples) // stride_samples) + 1
    return (num_windows_total * cnn_output_size) // 2


def stich_window_predictions(window_logits, original_audio_length, cnn_output_size, sample_rate=16000,
                             window_size_ms=160, stride_ms=80, row_stride=None):
    """windowing.py:103-173 on the device.  window_logits [B, num_windows, frames_per_window, D] (CUDA/HIP tensor);
    returns [B, total_frames, D].  `row_stride` (floats, >= D) pads the rows of the returned tensor's storage, e.g.
    68 for the 67-class head, so that the alignment kernels read 16-byte aligned rows; the result is then a view."""
    if not (isinstance(window_logits, torch.Tensor) and window_logits.is_cuda):
        raise RuntimeError("stich_window_predictions runs on the GPU only: pass a device tensor (there is no CPU fallback)")
    x = window_logits.to(torch.float32).contiguous()
    B, NW, F, D = x.shape
    total = stitch_total_frames(original_audio_length, cnn_output_size, sample_rate, window_size_ms, stride_ms)
    ld = D if row_stride is None else int(row_stride)
    if ld < D:
        raise ValueError("row_stride must be >= the output dimension")
    dev = x.device
    # the reference's weights (:130), computed by the same torch: cos(linspace(-pi/2, pi/2, F)) in float32
    weights = _weights(F, dev)
    store = torch.zeros((B, max(total, 0), ld), dtype=torch.float32, device=dev) if ld != D else \
        torch.empty((B, max(total, 0), ld), dtype=torch.float32, device=dev)
    L = _lib.lib()
    h = _lib.handle(dev.index if dev.index is not None else torch.cuda.current_device())
    with torch.cuda.device(dev):
        rc = L.bfa_stitch_windows(h, x.data_ptr(), B, NW, F, D, weights.data_ptr(), int(total), store.data_ptr(),
                                  store.stride(0), store.stride(1), torch.cuda.current_stream(dev).cuda_stream)
    _lib.check(rc, h, "bfa_stitch_windows")
    return store[:, :, :D]
