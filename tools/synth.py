"""tools/synth.py -- synthetic planted-path posteriors for bench.py and the tests (inputs only).

Two families:

* `synth_batch` / `synth_ragged`: torch-generator based (round 1), one tensor per call.
* `c4_*`: a COUNTER-BASED generator: every random draw is splitmix64 of a key built from
  (seed, global utterance index, frame, column), so an utterance's posteriors depend only on its global index --
  not on the rank that synthesises it, the shard it lands in, or the batch it is padded into.  That is what lets
  BASELINE.json configs[3] (B = 32768, T in [200,3000], S = T // 25, seed 1004) be sharded over N ranks with every
  rank synthesising only its own utterances, and lets rank 0 re-synthesise a parity sample after the gather.
"""
import numpy as np
import torch

_M64 = 1 << 64


def _s64(v):
    """python int -> the signed 64-bit value with the same bit pattern"""
    v %= _M64
    return v - _M64 if v >= (1 << 63) else v


def mix64(x):
    """splitmix64 finaliser on an int64 tensor (wrapping arithmetic, logical shifts)."""
    x = x + _s64(0x9E3779B97F4A7C15)
    x = (x ^ ((x >> 30) & ((1 << 34) - 1))) * _s64(0xBF58476D1CE4E5B9)
    x = (x ^ ((x >> 27) & ((1 << 37) - 1))) * _s64(0x94D049BB133111EB)
    return x ^ ((x >> 31) & ((1 << 33) - 1))


def _mix64_np(x):
    x = np.asarray(x, np.uint64)
    with np.errstate(over="ignore"):
        x = x + np.uint64(0x9E3779B97F4A7C15)
        x = (x ^ (x >> np.uint64(30))) * np.uint64(0xBF58476D1CE4E5B9)
        x = (x ^ (x >> np.uint64(27))) * np.uint64(0x94D049BB133111EB)
        return x ^ (x >> np.uint64(31))


def _u01(h):
    """53-bit uniform in [0,1) (float64) from a hash"""
    return ((h >> 11) & ((1 << 53) - 1)).to(torch.float64) * (1.0 / (1 << 53))


# key streams (disjoint by the tag in the top bits of the per-seed base)
_TAG_LEN, _TAG_TOK, _TAG_CUT, _TAG_N1, _TAG_N2 = 1, 2, 3, 4, 5


def _base(seed, tag):
    return _s64(int(_mix64_np(np.uint64((int(seed) * 8 + tag) & (_M64 - 1)))))


def c4_lengths(n_total, seed=1004, Tlo=200, Thi=3000):
    """(T[n_total], S[n_total]) int64 numpy: T uniform on Tlo..Thi, S = max(1, T // 25).  Host only, identical on every
    rank."""
    idx = np.arange(n_total, dtype=np.uint64)
    h = _mix64_np(np.uint64(_base(seed, _TAG_LEN) % _M64) + idx)
    T = (Tlo + (h >> np.uint64(11)) % np.uint64(Thi - Tlo + 1)).astype(np.int64)
    S = np.maximum(1, T // 25)
    return T, S


def c4_utterances(gidx, T_len, S_len, C, seed, device, Tpad=None, Spad=None, peak=9.0, sigma=1.0):
    """Posteriors of the utterances with GLOBAL indices `gidx` (lengths T_len / S_len from c4_lengths), padded to
    [n, Tpad, C] / [n, Spad].  Planted path: tokens uniform on 1..C-2 (no SIL, no blank), random monotone segmentation
    with >= 2 frames per token, logits = N(0,1) + peak * onehot(planted), log_probs = log_softmax(logits); rows beyond
    T are a softmax of plain noise (never read by the aligner).  Returns (lp f32 [n,Tpad,C], tokens i32 [n,Spad])."""
    gidx = torch.as_tensor(np.asarray(gidx, np.int64), device=device)
    Tl = torch.as_tensor(np.asarray(T_len, np.int64), device=device)
    Sl = torch.as_tensor(np.asarray(S_len, np.int64), device=device)
    n = gidx.numel()
    Tpad = int(Tpad or int(Tl.max()))
    Spad = int(Spad or max(1, int(Sl.max())))
    blank = C - 1
    # tokens
    j = torch.arange(Spad, device=device, dtype=torch.int64).unsqueeze(0)
    htok = mix64(_base(seed, _TAG_TOK) + gidx.unsqueeze(1) * 4096 + j)
    toks = 1 + ((htok >> 1) & ((1 << 62) - 1)) % (C - 2)
    # segmentation: 2S sorted cuts in [0, extra], gaps between them are the blank / token-extension runs
    extra = (Tl - 2 * Sl).clamp(min=0).unsqueeze(1)
    k = torch.arange(2 * Spad, device=device, dtype=torch.int64).unsqueeze(0)
    u = _u01(mix64(_base(seed, _TAG_CUT) + gidx.unsqueeze(1) * 8192 + k))
    cuts = torch.floor(u * (extra + 1).to(torch.float64)).to(torch.int64)
    cuts = torch.where(k < 2 * Sl.unsqueeze(1), cuts, extra.expand(-1, 2 * Spad))
    cuts, _ = torch.sort(cuts, dim=1)
    zeros = torch.zeros((n, 1), dtype=torch.int64, device=device)
    sizes = torch.diff(torch.cat([zeros, cuts, extra], dim=1), dim=1)  # gap,tok,gap,tok,...,gap
    tokslot = torch.arange(Spad, device=device).unsqueeze(0) < Sl.unsqueeze(1)
    sizes[:, 1::2] += 2 * tokslot.to(torch.int64)
    ends = torch.cumsum(sizes, dim=1)
    t = torch.arange(Tpad, device=device, dtype=torch.int64).unsqueeze(0).expand(n, Tpad).contiguous()
    slot = torch.searchsorted(ends, t, right=True)
    is_tok = ((slot % 2) == 1) & (t < Tl.unsqueeze(1))
    tok_idx = torch.clamp((slot - 1) // 2, 0, Spad - 1)
    planted = torch.where(is_tok, torch.gather(toks, 1, tok_idx), torch.full_like(slot, blank))
    # noise: Box-Muller in float64 on two hashed uniforms per element
    key = ((gidx.view(n, 1, 1) * 4096 + t.view(n, Tpad, 1)) * 128 +
           torch.arange(C, device=device, dtype=torch.int64).view(1, 1, C))
    u1 = _u01(mix64(_base(seed, _TAG_N1) + key)).clamp_(min=2.0 ** -53)
    u2 = _u01(mix64(_base(seed, _TAG_N2) + key))
    logits = (torch.sqrt(-2.0 * torch.log(u1)) * torch.cos(2.0 * np.pi * u2)).to(torch.float32)
    del key, u1, u2
    if sigma != 1.0:
        logits *= sigma
    logits.scatter_add_(2, planted.unsqueeze(-1), torch.full((n, Tpad, 1), peak, device=device))
    lp = torch.log_softmax(logits, dim=-1)
    return lp, toks.to(torch.int32)


def input_checksum(lp, T_len):
    """Per-utterance int64 sum of the float32 bit patterns of rows t < T (a regenerated utterance must reproduce it)."""
    B, Tmax, C = lp.shape
    Tl = torch.as_tensor(np.asarray(T_len, np.int64), device=lp.device) if not isinstance(T_len, torch.Tensor) \
        else T_len.to(device=lp.device, dtype=torch.int64)
    valid = (torch.arange(Tmax, device=lp.device).unsqueeze(0) < Tl.unsqueeze(1)).unsqueeze(-1)
    bits = lp.contiguous().view(torch.int32).to(torch.int64)
    return (bits * valid).sum(dim=(1, 2))


def synth_batch(B, T, S, C, seed, device, peak=9.0, sigma=1.0):
    """Planted-path posteriors: tokens iid uniform on 1..C-2 (no SIL, no blank),
    random monotone segmentation with >= 2 frames per token, logits = N(0,1) + peak*onehot(planted),
    log_probs = log_softmax(logits).  Generated on the device with a seeded torch generator."""
    g = torch.Generator(device=device)
    g.manual_seed(seed)
    blank = C - 1
    toks = torch.randint(1, C - 1, (B, S), generator=g, device=device)
    extra = T - 2 * S
    assert extra >= 0
    cuts, _ = torch.sort(torch.randint(0, extra + 1, (B, 2 * S), generator=g, device=device), dim=1)
    zeros = torch.zeros((B, 1), dtype=cuts.dtype, device=device)
    full = torch.full((B, 1), extra, dtype=cuts.dtype, device=device)
    sizes = torch.diff(torch.cat([zeros, cuts, full], dim=1), dim=1)  # [B, 2S+1] gap,tok,gap,tok,...,gap
    sizes[:, 1::2] += 2
    ends = torch.cumsum(sizes, dim=1)  # slot k covers [ends[k-1], ends[k])
    t = torch.arange(T, device=device).unsqueeze(0).expand(B, T).contiguous()
    slot = torch.searchsorted(ends, t, right=True)  # [B,T] in 0..2S
    is_tok = (slot % 2) == 1
    tok_idx = torch.clamp((slot - 1) // 2, 0, S - 1)
    planted = torch.where(is_tok, torch.gather(toks, 1, tok_idx), torch.full_like(slot, blank))
    logits = torch.randn((B, T, C), generator=g, device=device, dtype=torch.float32)
    if sigma != 1.0:
        logits *= sigma
    logits.scatter_add_(2, planted.unsqueeze(-1), torch.full((B, T, 1), peak, device=device))
    lp = torch.log_softmax(logits, dim=-1)
    return lp, toks.to(torch.int32)


def synth_ragged(B, Tlo, Thi, C, seed, device, peak=9.0, sigma=1.0, tok_div=25):
    """BASELINE.json configs[3] shape: T ~ U{Tlo..Thi}, S = max(1, T // 25), padded to Thi / max S.  Every
    utterance gets its own planted path over its own T frames and S tokens (same construction as synth_batch)."""
    gc = torch.Generator(device="cpu")
    gc.manual_seed(seed)
    T_len = torch.randint(Tlo, Thi + 1, (B,), generator=gc)
    S_len = torch.clamp(T_len // tok_div, min=1)
    Tmax, Smax = int(T_len.max()), int(S_len.max())
    g = torch.Generator(device=device)
    g.manual_seed(seed)
    blank = C - 1
    Td, Sd = T_len.to(device), S_len.to(device)
    toks = torch.randint(1, C - 1, (B, Smax), generator=g, device=device)
    extra = (Td - 2 * Sd).unsqueeze(1)  # [B,1] frames not forced to a token
    u = torch.rand((B, 2 * Smax), generator=g, device=device)
    cuts = torch.floor(u * (extra + 1).to(u.dtype)).to(torch.int64)
    k = torch.arange(2 * Smax, device=device).unsqueeze(0)
    cuts = torch.where(k < 2 * Sd.unsqueeze(1), cuts, extra.expand(-1, 2 * Smax))  # unused slots get no frames
    cuts, _ = torch.sort(cuts, dim=1)
    zeros = torch.zeros((B, 1), dtype=cuts.dtype, device=device)
    sizes = torch.diff(torch.cat([zeros, cuts, extra], dim=1), dim=1)  # [B, 2Smax+1] gap,tok,gap,tok,...,gap
    tokslot = torch.arange(Smax, device=device).unsqueeze(0) < Sd.unsqueeze(1)
    sizes[:, 1::2] += 2 * tokslot.to(sizes.dtype)
    ends = torch.cumsum(sizes, dim=1)
    t = torch.arange(Tmax, device=device).unsqueeze(0).expand(B, Tmax).contiguous()
    slot = torch.searchsorted(ends, t, right=True)
    is_tok = ((slot % 2) == 1) & (t < Td.unsqueeze(1))
    tok_idx = torch.clamp((slot - 1) // 2, 0, Smax - 1)
    planted = torch.where(is_tok, torch.gather(toks, 1, tok_idx), torch.full_like(slot, blank))
    logits = torch.randn((B, Tmax, C), generator=g, device=device, dtype=torch.float32)
    if sigma != 1.0:
        logits *= sigma
    logits.scatter_add_(2, planted.unsqueeze(-1), torch.full((B, Tmax, 1), peak, device=device))
    lp = torch.log_softmax(logits, dim=-1)
    return lp, toks.to(torch.int32), T_len.to(torch.int32), S_len.to(torch.int32)


def group_lut(device=None):
    """phoneme id -> group id of the synthetic "real text" workload: SIL (0) -> silence group 0, phonemes 1..65 ->
    groups 1..15, blank 66 -> blank group 16 (the shape of the reference's phoneme_id_to_group_id, core.py:868-871)."""
    lut = torch.tensor([0] + [1 + p % 15 for p in range(1, 66)] + [16], dtype=torch.int64)
    return lut if device is None else lut.to(device)


def synth_realtext(B, T, S, seed, device, sil_rate=1.0 / 12, sil_len=(12, 40), peak=9.0, gpeak=7.0, sigma=1.0):
    """What real transcripts give the aligner (SURVEY.md section 8(d) "-sil" variant, core.py:897-922): RAW logits of both
    heads (ph66: C = 67, blank 66; groups: C = 17, blank 16), targets with SIL (id 0) at ~`sil_rate` of the positions
    (punctuation -> SIL, ph66_phonemeizer.py:185-199) and a planted silence of sil_len frames for each, >= 2 frames per
    other token, blank elsewhere; logits = N(0,1) + peak * onehot(planted).  The group targets are the phoneme targets
    through `group_lut`.  Returns (logits_p [B,T,67], logits_g [B,T,17], tokens [B,S] i32, group_tokens [B,S] i32)."""
    g = torch.Generator(device=device)
    g.manual_seed(seed)
    blank = 66
    toks = torch.randint(1, blank, (B, S), generator=g, device=device)
    is_sil = torch.rand((B, S), generator=g, device=device) < sil_rate
    toks = torch.where(is_sil, torch.zeros_like(toks), toks)
    sil_frames = torch.randint(sil_len[0], sil_len[1] + 1, (B, S), generator=g, device=device)
    want = torch.where(is_sil, sil_frames, torch.full_like(sil_frames, 2))
    extra = (T - want.sum(dim=1, keepdim=True))
    assert int(extra.min()) >= 0, "the planted tokens do not fit into T frames"
    u = torch.rand((B, 2 * S), generator=g, device=device)
    cuts, _ = torch.sort(torch.floor(u * (extra + 1).to(u.dtype)).to(torch.int64), dim=1)
    zeros = torch.zeros((B, 1), dtype=torch.int64, device=device)
    sizes = torch.diff(torch.cat([zeros, cuts, extra], dim=1), dim=1)  # gap,tok,gap,tok,...,gap
    sizes[:, 1::2] += want
    ends = torch.cumsum(sizes, dim=1)
    t = torch.arange(T, device=device).unsqueeze(0).expand(B, T).contiguous()
    slot = torch.searchsorted(ends, t, right=True)
    is_tok = (slot % 2) == 1
    tok_idx = torch.clamp((slot - 1) // 2, 0, S - 1)
    planted = torch.where(is_tok, torch.gather(toks, 1, tok_idx), torch.full_like(slot, blank))
    lut = group_lut(device)
    logits_p = torch.randn((B, T, 67), generator=g, device=device, dtype=torch.float32)
    logits_g = torch.randn((B, T, 17), generator=g, device=device, dtype=torch.float32)
    if sigma != 1.0:
        logits_p *= sigma
        logits_g *= sigma
    logits_p.scatter_add_(2, planted.unsqueeze(-1), torch.full((B, T, 1), peak, device=device))
    logits_g.scatter_add_(2, lut[planted].unsqueeze(-1), torch.full((B, T, 1), gpeak, device=device))
    return logits_p, logits_g, toks.to(torch.int32), lut[toks].to(torch.int32)


def synth_realtext_ragged(B, Tlo, Thi, tok_div, seed, device, sil_rate=1.0 / 40, sil_len=(6, 30), peak=9.0, gpeak=7.0,
                          sigma=1.0):
    """The C5 proxy (BASELINE.json configs[4] without the model): what synth_realtext makes, with per-utterance lengths
    T ~ U{Tlo..Thi} (the reference cuts audio into segments of at most 30 s = 1 870 frames, README.md:1231) and
    S = max(1, T // tok_div) targets; SIL at `sil_rate` of the positions with a planted silence of sil_len frames each (the
    reference's committed LJSpeech outputs: 2 SIL in 110 phonemes, 64-182 ms; examples/samples/LJSpeech/LJ001-0001.vs.json),
    >= 2 frames per other token.  Padded to [B, Thi', *] / [B, Smax].  Returns (logits_p, logits_g, tokens, group_tokens,
    T_len i32 cpu, S_len i32 cpu)."""
    gc = torch.Generator(device="cpu")
    gc.manual_seed(seed)
    T_len = torch.randint(Tlo, Thi + 1, (B,), generator=gc)
    S_len = torch.clamp(T_len // tok_div, min=1)
    Tmax, Smax = int(T_len.max()), int(S_len.max())
    g = torch.Generator(device=device)
    g.manual_seed(seed)
    blank = 66
    Td, Sd = T_len.to(device).unsqueeze(1), S_len.to(device).unsqueeze(1)
    toks = torch.randint(1, blank, (B, Smax), generator=g, device=device)
    is_sil = torch.rand((B, Smax), generator=g, device=device) < sil_rate
    toks = torch.where(is_sil, torch.zeros_like(toks), toks)
    tokslot = torch.arange(Smax, device=device).unsqueeze(0) < Sd
    sil_frames = torch.randint(sil_len[0], sil_len[1] + 1, (B, Smax), generator=g, device=device)
    want = torch.where(is_sil, sil_frames, torch.full_like(sil_frames, 2)) * tokslot
    # (an utterance whose planted silences do not fit keeps 2 frames for them as well)
    over = want.sum(dim=1, keepdim=True) > Td
    want = torch.where(over, 2 * tokslot.to(want.dtype), want)
    extra = (Td - want.sum(dim=1, keepdim=True)).clamp(min=0)
    u = torch.rand((B, 2 * Smax), generator=g, device=device)
    cuts = torch.floor(u * (extra + 1).to(u.dtype)).to(torch.int64)
    k = torch.arange(2 * Smax, device=device).unsqueeze(0)
    cuts = torch.where(k < 2 * Sd, cuts, extra.expand(-1, 2 * Smax))
    cuts, _ = torch.sort(cuts, dim=1)
    zeros = torch.zeros((B, 1), dtype=torch.int64, device=device)
    sizes = torch.diff(torch.cat([zeros, cuts, extra], dim=1), dim=1)
    sizes[:, 1::2] += want
    ends = torch.cumsum(sizes, dim=1)
    t = torch.arange(Tmax, device=device).unsqueeze(0).expand(B, Tmax).contiguous()
    slot = torch.searchsorted(ends, t, right=True)
    is_tok = ((slot % 2) == 1) & (t < Td)
    tok_idx = torch.clamp((slot - 1) // 2, 0, Smax - 1)
    planted = torch.where(is_tok, torch.gather(toks, 1, tok_idx), torch.full_like(slot, blank))
    lut = group_lut(device)
    logits_p = torch.randn((B, Tmax, 67), generator=g, device=device, dtype=torch.float32)
    logits_g = torch.randn((B, Tmax, 17), generator=g, device=device, dtype=torch.float32)
    if sigma != 1.0:
        logits_p *= sigma
        logits_g *= sigma
    logits_p.scatter_add_(2, planted.unsqueeze(-1), torch.full((B, Tmax, 1), peak, device=device))
    logits_g.scatter_add_(2, lut[planted].unsqueeze(-1), torch.full((B, Tmax, 1), gpeak, device=device))
    return logits_p, logits_g, toks.to(torch.int32), lut[toks].to(torch.int32), T_len.to(torch.int32), S_len.to(torch.int32)
