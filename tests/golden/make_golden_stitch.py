#!/usr/bin/env python3
"""Generates tests/golden/stitch_cases.npz from the reference's own stich_window_predictions
(bournemouth_aligner/cupe2i/windowing.py:103-173), loaded by path (the module imports only torch and math).
Run in the build container (needs /root/reference); the .npz is what travels."""
import importlib.util
import os

import numpy as np
import torch

REF = "/root/reference/bournemouth_aligner/cupe2i/windowing.py"
spec = importlib.util.spec_from_file_location("ref_windowing", REF)
mod = importlib.util.module_from_spec(spec)
spec.loader.exec_module(mod)

rng = np.random.default_rng(2024)
out = {}
cases = [
    # (B, NW, F, C, audio_len, sample_rate, window_ms, stride_ms)
    (2, 5, 10, 67, 16000 * 480 // 1000, 16000, 160, 80),    # NW = num_windows_total: the last window is cut in half
    (1, 1, 10, 17, 16000 * 160 // 1000, 16000, 160, 80),    # a single window
    (3, 12, 10, 67, 16000 * 1040 // 1000, 16000, 160, 80),
    (2, 7, 9, 17, 16000 * 640 // 1000, 16000, 160, 80),     # odd frames per window: three windows overlap
    (1, 24, 10, 5, 16000 * 2000 // 1000, 16000, 160, 80),
    (2, 4, 10, 67, 16000 * 480 // 1000, 16000, 160, 80),    # fewer windows than the audio has: uncovered tail frames
]
for k, (B, NW, F, C, alen, sr, wms, sms) in enumerate(cases):
    x = torch.from_numpy(rng.normal(0, 3, size=(B, NW, F, C)).astype(np.float32))
    y = mod.stich_window_predictions(x, original_audio_length=alen, cnn_output_size=F, sample_rate=sr,
                                     window_size_ms=wms, stride_ms=sms)
    w = torch.cos(torch.linspace(-np.pi / 2, np.pi / 2, F))   # :130 (math.pi there)
    out[f"c{k}_x"] = x.numpy()
    out[f"c{k}_y"] = y.numpy()
    out[f"c{k}_w"] = w.numpy()
    out[f"c{k}_cfg"] = np.array([alen, F, sr, wms, sms], np.int64)
out["n"] = np.array(len(cases))
path = os.path.join(os.path.dirname(os.path.abspath(__file__)), "stitch_cases.npz")
np.savez_compressed(path, **out)
print("wrote", path, os.path.getsize(path))
