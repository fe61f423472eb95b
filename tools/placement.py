#!/usr/bin/env python3
"""tools/placement.py -- K1's duration as a function of WHERE the posterior buffer lives: N headline batches allocated the
way bench.py's generator leaves them, then the same data copied into fresh allocations; every buffer through every one of
D decoders (own library handle and workspace each); mean K1 duration (HIP events inside the library) per (decoder, buffer)."""
import ctypes, os, sys
import numpy as np
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from tools.synth import synth_batch
from bournemouth_forced_aligner_amd import AlignmentUtils, _lib

dev = torch.device("cuda", 0)
B, T, S, C = 4096, 1000, 40, 67
N = int(sys.argv[1]) if len(sys.argv) > 1 else 5
D = 2
lib = _lib.lib()
T_len = torch.full((B,), T, dtype=torch.int32, device=dev)
S_len = torch.full((B,), S, dtype=torch.int32, device=dev)
decs = []
for k in range(D):
    au = AlignmentUtils(C - 1, 0)
    au.viterbi_decoder.handle_slot = k
    decs.append(au)
hint = decs[0].viterbi_decoder.class_mask_hint([T] * B, [S] * B, has_sil=False, n_classes=C)


def k1(au, lp, tk, n=12):
    h = _lib.handle(0, au.viterbi_decoder.handle_slot)
    for _ in range(3):
        au.decode_alignments_device(lp, tk, T_len, S_len, class_mask=hint)
    torch.cuda.synchronize()
    lib.bfa_profile_enable(h, 1)
    for _ in range(n):
        au.decode_alignments_device(lp, tk, T_len, S_len, class_mask=hint)
    torch.cuda.synchronize()
    lib.bfa_profile_enable(h, 0)
    buf = (ctypes.c_float * n)()
    m = lib.bfa_profile_collect(h, buf, n)
    return float(np.mean([buf[i] for i in range(m)]))


def stream_ms(lp, n=10):
    """a plain sequential read of the same buffer (torch.sum), ms per pass"""
    for _ in range(2):
        lp.sum()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize()
    e0.record()
    for _ in range(n):
        lp.sum()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n


def report(tag, bufs):
    for j, (lp, tk) in enumerate(bufs):
        row = " ".join(f"{k1(au, lp, tk):.4f}" for au in decs)
        print(f"{tag} buffer {j} at 0x{lp.data_ptr():x}: K1 ms per decoder: {row}; torch.sum {stream_ms(lp):.4f} ms", flush=True)


bufs = [synth_batch(B, T, S, C, 1003 + 1000 * i, dev) for i in range(N)]
print(f"memory: allocated {torch.cuda.memory_allocated() / 2**30:.2f} GiB, reserved {torch.cuda.memory_reserved() / 2**30:.2f} GiB")
report("generator", bufs)
report("generator (again)", bufs[:2])
fresh = []
for lp, tk in bufs:
    c = torch.empty_like(lp)
    c.copy_(lp)
    fresh.append((c, tk))
report("copy (torch.empty_like)", fresh)
del fresh
torch.cuda.empty_cache()
fresh = []
for lp, tk in bufs:
    c = torch.empty_like(lp)
    c.copy_(lp)
    fresh.append((c, tk))
report("copy after empty_cache", fresh)

# ---- physically contiguous allocations (hipExtMallocWithFlags, hipDeviceMallocContiguous)
hip = ctypes.CDLL("libamdhip64.so")
hip.hipExtMallocWithFlags.argtypes = [ctypes.POINTER(ctypes.c_void_p), ctypes.c_size_t, ctypes.c_uint]


class _Raw:
    def __init__(self, ptr, shape):
        self.__cuda_array_interface__ = {"shape": tuple(shape), "typestr": "<f4", "data": (ptr, False), "version": 2}


del fresh
torch.cuda.empty_cache()
contig = []
for flag, name in ((0x4, "contiguous"), (0x0, "default hipMalloc")):
    for lp, tk in bufs[:4]:
        p = ctypes.c_void_p()
        rc = hip.hipExtMallocWithFlags(ctypes.byref(p), lp.numel() * 4, flag)
        if rc != 0:
            print(f"hipExtMallocWithFlags({name}) failed: {rc}")
            break
        t = torch.as_tensor(_Raw(p.value, lp.shape), device=dev)
        t.copy_(lp)
        contig.append((t, tk))
        print(f"{name} buffer at 0x{p.value:x}: K1 ms per decoder: " + " ".join(f"{k1(au, t, tk):.4f}" for au in decs) + f"; torch.sum {stream_ms(t):.4f} ms", flush=True)
