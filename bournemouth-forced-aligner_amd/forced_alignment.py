"""Host-side mirror of bournemouth_aligner/forced_alignment.py (reference lines cited per method).

Same class names, constructor arguments, method names, return types and error behaviour as the
reference's `ViterbiDecoder` / `AlignmentUtils`; the work itself runs in the gfx950 kernels behind
the C-ABI of include/bfa.h.  Posteriors stay device-resident: `log_probs` is expected to be a
float32 tensor on the GPU (a CPU tensor is uploaded, that is plumbing, not a fallback -- without a
GPU every entry point raises).
"""
import collections.abc
import ctypes

import numpy as np
import torch

from . import _lib


def _device_of(t):
    if t.is_cuda:
        return t.device
    if not torch.cuda.is_available():
        raise RuntimeError("bournemouth-forced-aligner_amd needs an AMD GPU (HIP): no device is visible and "
                           "there is no CPU implementation of the alignment path")
    return torch.device("cuda", torch.cuda.current_device())


class HostLens:
    """A length vector known on the host, converted ONCE (int64 numpy for the class hint, int32 device tensor for the
    kernels): both heads of a call, the post-DP stages and the result shaping share it instead of converting the caller's
    list of B python ints every time (0.15 ms per conversion at B = 4096)."""
    __slots__ = ("host", "_dev")

    def __init__(self, x):
        if isinstance(x, torch.Tensor):
            x = x.cpu().numpy()
        self.host = np.ascontiguousarray(np.asarray(x, dtype=np.int64).reshape(-1))
        self._dev = {}

    def __len__(self):
        return self.host.shape[0]

    def __iter__(self):
        return iter(self.host.tolist())

    def __getitem__(self, i):
        v = self.host[i]
        return int(v) if np.ndim(v) == 0 else v.tolist()

    def dev(self, device):
        t = self._dev.get(device)
        if t is None:
            t = self._dev[device] = torch.from_numpy(self.host.astype(np.int32)).to(device, non_blocking=True)
        return t


def _as_i32(x, device, n=None):
    if x is None:
        return None
    if isinstance(x, HostLens):
        return x.dev(device)
    if not isinstance(x, torch.Tensor):
        x = torch.as_tensor(np.asarray(x))
    return x.to(device=device, dtype=torch.int32, non_blocking=True).contiguous()


def _host_array(x):
    """x as a numpy array if it lives on the host (list / ndarray / CPU tensor), else None (no device sync)."""
    if x is None:
        return None
    if isinstance(x, HostLens):
        return x.host
    if isinstance(x, torch.Tensor):
        return x.numpy() if x.device.type == "cpu" else None
    try:
        return np.asarray(x)
    except Exception:
        return None


def _host_ints(x):
    a = _host_array(x)
    return None if a is None else a.astype(np.int64).reshape(-1)


class _Workspace:
    """Caller-owned scratch for bfa_align_batch, cached per shape."""

    def __init__(self):
        self.buf = None

    def get(self, nbytes, device):
        if self.buf is None or self.buf.numel() < nbytes or self.buf.device != device:
            self.buf = torch.empty(int(nbytes), dtype=torch.uint8, device=device)
        return self.buf


_ROW4 = np.dtype([("phoneme", "<i4"), ("start", "<i4"), ("end", "<i4"), ("target_idx", "<i4")])
_ROW4_16 = np.dtype([("phoneme", "<u2"), ("start", "<u2"), ("end", "<u2"), ("target_idx", "<i2")])  # bfa_pack_results16


class LazyRowLists(collections.abc.Sequence):
    """list[B]-like result of a batch: element b is the list of utterance b's tuples (forced_alignment.py:871,908), built
    from ONE packed structured array the first time it is asked for and kept -- `x[b]`, iteration, `len`, slices, `==` with
    lists of lists all behave like the reference's list of lists; `list(x)` / `x.tolist()` give the plain list.  Building
    163 840 tuples costs CPython 14 ms on the headline batch whoever does it; a caller that looks at some utterances, or
    hands the batch on, does not pay for the rest."""
    __slots__ = ("_rec", "_off", "_rows")

    def __init__(self, records, counts):
        self._rec = records
        self._off = np.concatenate([[0], np.cumsum(np.asarray(counts, np.int64))])
        self._rows = [None] * (len(self._off) - 1)

    def __len__(self):
        return len(self._rows)

    def _row(self, b):
        r = self._rows[b]
        if r is None:
            r = self._rows[b] = self._rec[self._off[b]:self._off[b + 1]].tolist()
        return r

    def __getitem__(self, b):
        if isinstance(b, slice):
            return [self._row(i) for i in range(*b.indices(len(self._rows)))]
        n = len(self._rows)
        if b < -n or b >= n:
            raise IndexError("list index out of range")
        return self._row(b % n if n else 0)

    def tolist(self):
        """the plain list[B] of lists of tuples, every row built in ONE numpy pass (rows_as_tuple_lists)"""
        if any(r is None for r in self._rows):
            built = rows_as_tuple_lists(self._rec, np.diff(self._off))
            self._rows = [r if r is not None else nb for r, nb in zip(self._rows, built)]
        return list(self._rows)

    def __iter__(self):
        return iter(self.tolist()) if all(r is None for r in self._rows) else (self._row(b) for b in range(len(self._rows)))

    def __eq__(self, other):
        if isinstance(other, (list, tuple, LazyRowLists)):
            return len(other) == len(self) and all(a == b for a, b in zip(self.tolist(), other))
        return NotImplemented

    __hash__ = None

    def __repr__(self):
        return repr(self.tolist())


class LazyRowDicts(collections.abc.Sequence):
    """list[B]-like of per-utterance dicts {head: rows} (core.py:958-964: `phoneme_timestamps`, `group_timestamps`) over
    LazyRowLists: the dict of utterance b, with that utterance's tuples, is built when it is asked for and kept."""
    __slots__ = ("_heads", "_n", "_dicts")

    def __init__(self, heads, n):
        self._heads = heads  # {key: LazyRowLists}
        self._n = n
        self._dicts = [None] * n

    def __len__(self):
        return self._n

    def _one(self, b):
        d = self._dicts[b]
        if d is None:
            d = self._dicts[b] = {k: v[b] for k, v in self._heads.items()}
        return d

    def __getitem__(self, b):
        if isinstance(b, slice):
            return [self._one(i) for i in range(*b.indices(self._n))]
        if b < -self._n or b >= self._n:
            raise IndexError("list index out of range")
        return self._one(b % self._n if self._n else 0)

    def tolist(self):
        if all(d is None for d in self._dicts):   # nothing looked at yet: every head in one numpy pass, one dict literal per utterance
            keys = list(self._heads)
            cols = [self._heads[k].tolist() for k in keys]
            if len(keys) == 2:
                k0, k1 = keys
                self._dicts = [{k0: x, k1: y} for x, y in zip(cols[0], cols[1])]
            else:
                self._dicts = [dict(zip(keys, rows)) for rows in zip(*cols)]
            return list(self._dicts)
        for v in self._heads.values():
            v.tolist()  # every row of a head in one numpy pass
        return [self._one(b) for b in range(self._n)]

    def __iter__(self):
        return iter(self.tolist())

    def __eq__(self, other):
        if isinstance(other, (list, tuple, LazyRowDicts)):
            return len(other) == self._n and all(a == b for a, b in zip(self.tolist(), other))
        return NotImplemented

    __hash__ = None

    def __repr__(self):
        return repr(self.tolist())


def rows_as_tuple_lists(records, counts):
    """`records`: a flat numpy STRUCTURED array, one record per valid row of the batch in utterance order; `counts` [B].
    Returns list[B] of list[tuple]: numpy turns a record into a tuple of python scalars in C, the flat list is then cut
    per utterance.  The cyclic garbage collector is paused meanwhile: the result is ~10^5 tuples, none of them part of a
    cycle, and every generation-0 pass over them (one per 700 allocations) only costs time."""
    import gc
    paused = gc.isenabled()
    if paused:
        gc.disable()
    try:
        flat = records.tolist()
        out, lo = [], 0
        for hi in np.cumsum(counts).tolist():
            out.append(flat[lo:hi])
            lo = hi
        return out
    finally:
        if paused:
            gc.enable()


class AlignmentResult:
    """Device-resident result of one batch (no host synchronisation has happened yet)."""

    def __init__(self, segs, seg_count, status, mode, frame_ph, frame_idx, T_len, S_len):
        self.segs = segs            # int32 [B, seg_cap, 4]  (phoneme, start, end, target_idx)
        self.seg_count = seg_count  # int32 [B]
        self.status = status        # int32 [B]  BFA_ITEM_*
        self.mode = mode            # int32 [B]  BFA_MODE_*
        self.frame_phonemes = frame_ph    # int32 [B, Tmax]
        self.frame_phonemes_idx = frame_idx
        self.T_len = T_len
        self.S_len = S_len
        self.conf = None         # float32 [B, seg_cap] / int32 [B]: only when the call also ran the post-DP stages
        self.conf_status = None  # (align_heads(post=...))
        self.postprocessed = False  # coverage / soft boundaries already ran in place on segs / seg_count (they are not idempotent)

    def raise_for_status(self):
        """Reproduce the reference's exceptions (they abort the whole call)."""
        st = self.status.cpu().numpy()
        if (st == _lib.ITEM_OK).all():
            return
        T = self.T_len.cpu().numpy() if self.T_len is not None else None
        S = self.S_len.cpu().numpy()
        for b in range(st.shape[0]):
            if st[b] == _lib.ITEM_TOO_SHORT:  # forced_alignment.py:161-165
                nf = int(T[b]) if T is not None else self.frame_phonemes.shape[1]
                raise ValueError(
                    f"Audio too short to align: {int(S[b])} phonemes cannot be fit into "
                    f"{nf} frames (need at least 1 frame per phoneme).")
            if st[b] == _lib.ITEM_BAD_TOKEN:
                raise IndexError(f"target phoneme id out of range for item {b} (index out of range in log_probs)")
            if st[b] == _lib.ITEM_TOO_LARGE:
                raise RuntimeError(f"item {b}: CTC path longer than this build supports")
            if st[b] == _lib.ITEM_SEG_OVERFLOW:
                raise RuntimeError(f"item {b}: more aligned runs than seg_cap")
            if st[b] == _lib.ITEM_BAD_HINT:
                raise RuntimeError(f"item {b}: the class_mask hint excludes what this utterance needs (a K1 class bit is "
                                   f"missing, or no-silence was promised although the target contains the silence id)")

    def call_counters(self):
        """Diagnostics (bfa_call_counters; synchronises): what the items of the call that produced this result did --
        {"items", "redone_full", "redone_exact" (fast windows that gave up), "exact_done" / "exact_alive" (standard mode: items the exact rerun kernels aligned / of those above the sentinel)}.  Only valid until the decoder's next call (the counters
        live in its workspace)."""
        B, Tmax, Smax, C, params, ws, dev = self._call
        L = _lib.lib()
        h = _lib.handle(dev.index if dev.index is not None else torch.cuda.current_device())  # (any handle: the counters live in the workspace)
        out = (ctypes.c_int32 * 16)()
        with torch.cuda.device(dev):
            rc = L.bfa_call_counters(h, ws.data_ptr(), B, Tmax, Smax, C, ctypes.byref(params), out,
                                     torch.cuda.current_stream(dev).cuda_stream)
        _lib.check(rc, h, "bfa_call_counters")
        return {"items": int(out[0]), "redone_full": int(out[2]), "redone_exact": int(out[3]), "exact_done": int(out[4]),
                "exact_alive": int(out[5]), "raw": [int(v) for v in out]}

    def to_lists(self, check_status=False):
        """list[B] of list[(phoneme_id, start_frame, end_frame, target_seq_idx)] (forced_alignment.py:871).
        ONE kernel packs the valid rows back to back (bfa_pack_results), ONE pinned copy brings the record to the host, one
        flat list of tuples is cut per utterance (a per-utterance `.tolist()` loop cost 76 ms on the 4096-utterance headline
        batch against 0.35 ms of device time).  `check_status`: the per-utterance status rides along in the same round of
        copies and raise_for_status() runs on it -- one synchronisation per decode_alignments call instead of two."""
        from .sharding import pack_layout, pack_results
        n, cap = int(self.segs.shape[0]), int(self.segs.shape[1])
        bound = n * cap
        # frame counts, ids and target indices that fit 16 bits (any real batch): 8 bytes per tuple instead of 16 -- the copy
        # of the tuples is most of what the call costs the host beyond the device step (2.6 MB for the headline batch)
        # (phoneme ids are < C <= 128: the library refuses wider posteriors, bfa_capi.cpp align_impl)
        small = self.frame_phonemes.shape[1] < 65536 and cap < 32768
        if small:
            L = _lib.lib()
            dev = self.segs.device
            words = int(L.bfa_pack16_words(n, bound))
            rec = torch.empty((words,), dtype=torch.int32, device=dev)
            h = _lib.handle(dev.index if dev.index is not None else torch.cuda.current_device())
            with torch.cuda.device(dev):
                rc = L.bfa_pack_results16(h, self.segs.data_ptr(), cap, self.seg_count.data_ptr(), n, n, bound, rec.data_ptr(),
                                          torch.cuda.current_stream(dev).cuda_stream)
            _lib.check(rc, h, "bfa_pack_results16")
            lay = pack_layout(n, 0, False)
            lay["words"] = words
            tw, row = 2, _ROW4_16   # words per tuple, record type
        else:
            rec = pack_results(self.segs, self.seg_count, None, None, n, bound)
            lay = pack_layout(n, bound, False)
            tw, row = 4, _ROW4
        st = torch.cuda.current_stream(rec.device)
        status_h = None
        if check_status:
            status_h = torch.empty((n,), dtype=torch.int32, pin_memory=True)
            status_h.copy_(self.status, non_blocking=True)
        if lay["words"] * 4 <= (8 << 20):   # one copy of the whole record into pinned memory, one synchronisation
            host = torch.empty((lay["words"],), dtype=torch.int32, pin_memory=True)
            host.copy_(rec, non_blocking=True)
            st.synchronize()
            if status_h is not None and bool((status_h != 0).any()):
                self.raise_for_status()
            head = host.numpy()
            total = int(head[1])
            if int(head[5]) != 0:   # header word 5: the record was cut (bfa.h, bfa_pack_results)
                raise RuntimeError("bfa_pack_results: the packed record overflowed its bound")
            packed = head[lay["tuples"]:lay["tuples"] + tw * total]
        else:                               # the count table first, then exactly the packed tuples
            head = rec[:lay["tuples"]].cpu().numpy()
            if status_h is not None and bool((status_h != 0).any()):
                self.raise_for_status()
            total = int(head[1])
            if int(head[5]) != 0:
                raise RuntimeError("bfa_pack_results: the packed record overflowed its bound")
            host = torch.empty((tw * total,), dtype=torch.int32, pin_memory=True)
            host.copy_(rec[lay["tuples"]:lay["tuples"] + tw * total], non_blocking=True)
            st.synchronize()
            packed = host.numpy()
        cnt = head[lay["count"]:lay["count"] + n]
        self._host_record = host  # (the numpy views above live in this pinned block)
        return LazyRowLists(packed.view(row).reshape(-1), cnt)


class ViterbiDecoder:
    """Mirror of forced_alignment.py:11-834 (constructor :16-23)."""

    def __init__(self, blank_id, silence_id, silence_anchors=3, min_phoneme_prob=1e-8, ignore_noise=True,
                 truly_forced=False):
        self.blank_id = blank_id
        self.silence_id = silence_id
        self.silence_anchors = silence_anchors
        self.min_phoneme_prob = min_phoneme_prob  # forced_alignment.py:20; its float32 log is the floor (bfa_params.min_log_prob)
        self.ignore_noise = ignore_noise
        self.truly_forced = truly_forced
        self._neg_inf = -1000.0
        self._ws = _Workspace()
        # K1's sliding-window variant is only tried up to this many tokens (None = the library's 64); raise it for
        # posteriors that are known to keep path scores above the -1000 sentinel (bfa_params.window_max_tokens)
        self.window_max_tokens = None
        self.window_max_frames = None   # likewise for the frame limit (bfa_params.window_max_frames)
        self.handle_slot = 0            # which of the process's handles this decoder uses (see _lib.handle)
        self._hint_cache = {}           # hint_and_path results by length vector
        self._dev_hint = {}             # ... and for lengths / targets that live on the device, by tensor identity
        self.hint_from_device_lengths = True  # see _hint_of_device_lengths

    def set_blank_id(self, blank_id):
        """forced_alignment.py:25-27"""
        self.blank_id = blank_id

    # ---- parameter block for the C-ABI
    def _params(self, boost_targets, enforce_minimum, anchor_pauses, simple=False, max_blanks=10):
        if self.blank_id is None:
            raise ValueError("Blank ID not set. Call set_blank_id first.")  # forced_alignment.py:104-105
        p = _lib.BfaParams()
        # forced_alignment.py:70: the floor is torch.log(torch.tensor(min_phoneme_prob)) -- float32, computed by the
        # caller's own torch so that the kernels compare against the very bits the reference compares against
        min_log = float(torch.log(torch.tensor(self.min_phoneme_prob, dtype=torch.float32)))
        if min_log != min_log:  # NaN (negative probability): `x < nan` is never true, the reference floors nothing
            enforce_minimum = False
        else:
            p.has_min_log_prob = 1
            p.min_log_prob = min_log
        p.blank_id = int(self.blank_id)
        p.silence_id = -1 if self.silence_id is None else int(self.silence_id)
        p.silence_anchors = int(self.silence_anchors) if anchor_pauses else 0
        p.ignore_noise = int(bool(self.ignore_noise))
        p.truly_forced = int(bool(self.truly_forced))
        p.boost_targets = int(bool(boost_targets))
        p.enforce_minimum = int(bool(enforce_minimum))
        p.simple = int(bool(simple))
        p.max_blanks = int(max_blanks)
        return p

    @staticmethod
    def _r_class(L):
        r = (L + 63) // 64
        for i, c in enumerate((2, 3, 4, 6, 8, 12, 16)):
            if r <= c:
                return i
        return None

    _WIN_MAX_FRAMES = 1536  # bfa_types.hpp WIN_MAX_FRAMES
    _WIN_MAX_TOKENS = 64    # bfa_types.hpp WIN_MAX_TOKENS (bfa_params.window_max_tokens / `window_max_tokens` overrides)

    @classmethod
    def _win_class(cls, L, T=0):
        """Sliding-window class (states per lane) the planner picks for a standard-mode DP of L states over T
        frames, or 0 (bfa_types.hpp win_class_for)."""
        bw = max(L // 4, 20) if L > 60 else 0
        full = cls._r_class(L)
        if bw <= 0 or full is None or T > cls._WIN_MAX_FRAMES:
            return 0
        rfull = (2, 3, 4, 6, 8, 12, 16)[full]
        for rw in (1, 2, 3, 4, 6, 8):
            if rw >= rfull:
                break
            frames_per_word = {1: 16, 2: 8}.get(rw, 4)
            if 2 * bw + 1 + frames_per_word + 2 + rw + 1 <= 64 * rw:
                return rw
        return 0

    def class_mask_hint(self, T_lens, S_lens, has_sil, anchor_pauses=True, simple=False, n_classes=None,
                        boost_targets=True, enforce_minimum=True):
        """bfa_params.class_mask for a batch, from HOST copies of its lengths (see hint_and_path for the rules)."""
        return self.hint_and_path(T_lens, S_lens, has_sil, anchor_pauses, simple, n_classes, boost_targets, enforce_minimum)[0]

    def hint_and_path(self, T_lens, S_lens, has_sil, anchor_pauses=True, simple=False, n_classes=None,
                      boost_targets=True, enforce_minimum=True, Smax=None):
        """Cached front of _hint_and_path: the hint of a batch depends on its (T, S) pairs only, and callers align batch
        after batch of the same shape -- the numpy passes over 4096 lengths cost 0.3 ms per call, as much as the device
        step.  All-equal lengths are reduced to one pair; other length vectors are looked up by their bytes."""
        T = np.ascontiguousarray(np.asarray(list(T_lens) if not isinstance(T_lens, np.ndarray) else T_lens, dtype=np.int64).reshape(-1))
        S = np.ascontiguousarray(np.asarray(list(S_lens) if not isinstance(S_lens, np.ndarray) else S_lens, dtype=np.int64).reshape(-1))
        if T.size and T.size == S.size and int(T.min()) == int(T.max()) and int(S.min()) == int(S.max()):
            lens_key = ("eq", T.size, int(T[0]), int(S[0]))
        else:
            lens_key = (T.tobytes(), S.tobytes())
        key = (lens_key, bool(has_sil), bool(anchor_pauses), bool(simple), n_classes, bool(boost_targets), bool(enforce_minimum), Smax,
               self.silence_anchors, self.silence_id, self.min_phoneme_prob, self.window_max_tokens, self.window_max_frames)
        hit = self._hint_cache.get(key)
        if hit is None:
            hit = self._hint_and_path(T, S, has_sil, anchor_pauses, simple, n_classes, boost_targets, enforce_minimum, Smax)
            if len(self._hint_cache) >= 16:
                self._hint_cache.pop(next(iter(self._hint_cache)))
            self._hint_cache[key] = hit
        return hit

    def _hint_and_path(self, T_lens, S_lens, has_sil, anchor_pauses=True, simple=False, n_classes=None,
                       boost_targets=True, enforce_minimum=True, Smax=None):
        """(class_mask, path): the hint and the launch layout it is written for (_lib.PATH_*; the library's own decision
        for the same shapes is bfa_call_path -- tests/test_host_and_abi.py holds the two against each other).  `Smax`: the
        padded target width the call will pass (default: the longest target).
        Optional host-side hint for bfa_params.class_mask: the K1 kernel classes that occur in this batch,
        from HOST copies of the lengths.  Bits 0-6: full-layout states-per-lane classes {2,3,4,6,8,12,16};
        bits 8-15: sliding-window classes Rw in {1,2,3,4,6,8} at bit 7+Rw (used for standard-mode DPs whose band is narrower than
        the path, with the reference-default flags on the 67- / 17-class heads; pass `n_classes`); bits 20-27: the EXACT window
        of class Rw at bit 19+Rw (utterances with more frames / tokens than the fast window is tried on).
        `has_sil` says whether any target may contain the silence id (then the segmented mode can create
        shorter DPs, and every full class up to the largest is kept; otherwise bit 16 tells the library to
        skip the silence planning, and a target that does contain it is reported as ITEM_BAD_HINT).  Bit 17 (a speed hint) says the
        utterances have about the same number of frames.  Without a hint the library launches every class the tensor shapes allow."""
        T = np.asarray(list(T_lens) if not isinstance(T_lens, np.ndarray) else T_lens, dtype=np.int64).reshape(-1)
        S = np.asarray(list(S_lens) if not isinstance(S_lens, np.ndarray) else S_lens, dtype=np.int64).reshape(-1)
        if T.size == 0:
            return 0, _lib.PATH_CLASS_KERNELS
        path = _lib.PATH_CLASS_KERNELS
        Lmax_call = 4 * int(Smax if Smax is not None else max(1, int(S.max()))) + 1   # the launcher's bound (bfa_kernels.hip plan_call)
        window_ok = (n_classes in (67, 17)) and boost_targets and enforce_minimum and not simple
        # a floor above log(1) = 0 (min_phoneme_prob > 1) breaks the window's exactness argument: the library then runs
        # every item with the full layout (bfa_launch_align), so the hint must name the full classes
        if window_ok and not (float(torch.log(torch.tensor(self.min_phoneme_prob, dtype=torch.float32))) <= 0.0):
            window_ok = False
        classes = np.array((2, 3, 4, 6, 8, 12, 16))
        if simple:  # forced_alignment.py:963-968 (float32 compares); S = 0 still runs a DP over the single blank state
            f32 = np.float32
            stride = np.full(T.shape, 4, np.int64)
            stride[(4 * S + 1).astype(f32) > T.astype(f32) * f32(0.9)] = 3
            stride[(stride * S + 1).astype(f32) > T.astype(f32) * f32(0.8)] = 2
            L = stride * S + 1
            is_dp = T >= 1
        else:       # forced_alignment.py:153-176
            stride = np.full(T.shape, 4, np.int64)
            for s2 in (3, 2, 1):
                stride[stride * S + 1 > T] = s2
            L = stride * S + 1
            is_dp = (S > 0) & (L <= T)
        ci = np.searchsorted(classes, (L + 63) // 64)            # index of the full-layout class, 7 = beyond 1024 states
        mask = 0
        rw = np.zeros(T.shape, np.int64)
        if window_ok:                                            # bfa_types.hpp win_class_for
            bw = np.where(L > 60, np.maximum(L // 4, 20), 0)
            rfull = classes[np.minimum(ci, 6)]
            for r in (8, 6, 4, 3, 2, 1):
                fpw = {1: 16, 2: 8}.get(r, 4)
                fits = (2 * bw + 1 + fpw + 2 + r + 1 <= 64 * r) & (r < rfull)
                rw[fits] = r
            max_tok = self.window_max_tokens if self.window_max_tokens else self._WIN_MAX_TOKENS
            max_frm = self.window_max_frames if self.window_max_frames else self._WIN_MAX_FRAMES
            rw[(bw <= 0) | (ci > 6) | ~is_dp] = 0
            # too long / too many tokens for the fast window (its result only stands above the sentinel): the EXACT window
            # of the same class, bits 20-27 (strides >= 3, standard mode only: bfa_plan.inc)
            exact = (rw > 0) & ((T > max_frm) | (S > max_tok))
            no_seg = not (has_sil and anchor_pauses and self.silence_anchors > 0)
            # a mixed-length call (bfa_launch_align: no promise of uniform lengths, two utterances or more): every stride >= 3
            # window item of the classes Rw <= 4 is an exact-window item (k_mix aligns and walks it in one workgroup)
            # (not a small call whose DPs all sit in ONE fast-window class of at most 256 states: that one is k_one's)
            dp_rw = rw[is_dp]
            # (the library takes such a call as ONE kernel up to _lib.ONE_MAX_BATCH utterances and 4 * Smax + 1 <= 256 states; the
            # hint asks for it below ONE_HINT_MAX_BATCH, where it measured faster than k_mix, profiles/r04_latency_mixed.txt)
            single = (dp_rw.size > 0 and int(dp_rw.min()) == int(dp_rw.max()) and 1 <= int(dp_rw.max()) <= 3
                      and not bool(exact.any()) and Lmax_call <= 256 and no_seg)
            one_class = single and T.size < _lib.ONE_HINT_MAX_BATCH
            mixed = no_seg and T.size >= _lib.MIX_MIN_BATCH and not self._uniform(T) and not one_class and Lmax_call > 60
            if mixed:
                exact = exact | ((rw > 0) & (rw <= 4) & (stride >= 3))
                path = _lib.PATH_MIXED
            elif single and T.size <= _lib.ONE_MAX_BATCH:  # (a uniform single-class call up to ONE_MAX_BATCH, e.g. BASELINE configs[1])
                path = _lib.PATH_ONE_KERNEL
            rx = np.where(exact & (stride >= 3) & no_seg, rw, 0)
            rw[exact] = 0
            for r in np.unique(rx[rx > 0]):
                mask |= 1 << (19 + int(r))
        else:
            rx = rw
        for r in np.unique(rw[rw > 0]):
            mask |= 1 << (7 + int(r))                            # (the rare sentinel rerun needs no hint bit)
        for c in np.unique(ci[is_dp & (rw == 0) & (rx == 0) & (ci <= 6)]):
            mask |= 1 << int(c)
        if has_sil and anchor_pauses and self.silence_anchors > 0 and not simple and (S > 0).any():
            top = int(np.minimum(np.searchsorted(classes, (4 * S[S > 0] + 1 + 63) // 64), 6).max())
            mask |= (1 << (top + 1)) - 1                         # speech segments can be any shorter class
        if not has_sil and mask:
            mask |= _lib.HINT_NO_SILENCE_TARGETS  # the silence-anchored planning kernels are not launched
        if mask and self._uniform(T):
            mask |= _lib.HINT_UNIFORM_LENGTHS     # about the same number of frames everywhere: one contiguous eighth of the batch per XCD
        if not mask:  # no class of the hint occurs (paths beyond 1024 states, no DP at all): the call goes out unhinted, and
            # without the no-silence bit the library has to assume silence anchoring whenever the decoder allows it
            seg = anchor_pauses and self.silence_anchors > 0 and self.silence_id is not None and not simple
            path = _lib.PATH_CLASS_KERNELS if (seg or not window_ok or Lmax_call <= 60 or T.size < _lib.MIX_MIN_BATCH) else _lib.PATH_MIXED
        return mask, path

    def _hint_of_device_lengths(self, pred_lens, true_seqs_lens, true_seqs, B, Tmax, toks, sil, hint_kw):
        """The class hint for lengths / targets that live on the DEVICE: without it the library has to assume silence
        anchoring and every class (0.75 against 0.40 ms on the headline batch).  The first call with a given set of
        tensors reads them once (one small copy to the host, i.e. one synchronisation); the hint is then kept by tensor
        IDENTITY and version counter, so a caller that aligns the same lengths again -- or batch after batch through the
        same length tensors without writing to them -- pays nothing.  `hint_from_device_lengths = False` switches this
        off (the call then never touches the host)."""
        import weakref
        ts = [t for t in (pred_lens, true_seqs_lens, true_seqs) if isinstance(t, torch.Tensor)]
        # Everything hint_and_path keys on that is not a tensor rides in the key (a decoder whose anchors / floor / window
        # limits change must not replay the old class bits).  Targets that are NOT a tensor (host list / ndarray) are not
        # identified by anything in the key: such a call is never served from the cache (its has_sil is recomputed below).
        cacheable = isinstance(true_seqs, torch.Tensor)
        key = tuple((id(t), t._version, t.data_ptr()) for t in ts) + tuple(sorted(hint_kw.items())) + \
            (B, Tmax, sil, self.silence_anchors, self.min_phoneme_prob, self.window_max_tokens, self.window_max_frames)
        hit = self._dev_hint.get(key) if cacheable else None
        if hit is not None and all(r() is t for r, t in zip(hit[1], ts)):
            return hit[0]
        Th = np.full(B, Tmax, np.int64) if pred_lens is None else _host_ints(pred_lens.cpu() if isinstance(pred_lens, torch.Tensor) else pred_lens)
        Sh = _host_ints(true_seqs_lens.cpu())
        if len(Th) != B or len(Sh) != B:
            return 0
        valid = torch.arange(toks.shape[1], device=toks.device)[None, :] < true_seqs_lens.to(toks.device)[:, None]
        has_sil = bool(((toks == sil) & valid).any())
        mask = self.hint_and_path(np.clip(Th, 0, Tmax), np.clip(Sh, 0, toks.shape[1]), has_sil, **hint_kw)[0]
        if cacheable:
            if len(self._dev_hint) >= 8:
                self._dev_hint.pop(next(iter(self._dev_hint)))
            self._dev_hint[key] = (mask, [weakref.ref(t) for t in ts])
        return mask

    def forget_device_hints(self):
        """Drop the hints kept for device-resident length / target tensors (a caller that rewrites such a tensor behind
        torch's version counter -- `.data`, an external kernel -- and sees BFA_ITEM_BAD_HINT calls this and aligns again)."""
        self._dev_hint.clear()

    @staticmethod
    def _uniform(T):
        return T.size >= 64 and int(T.max() - T.min()) * 8 <= int(T.max())

    def _prepare_call(self, log_probs, true_seqs, pred_lens, true_seqs_lens, boost_targets=True, enforce_minimum=True,
                      anchor_pauses=True, simple=False, seg_cap=None, max_blanks=10, class_mask=0):
        """Everything one bfa_align_batch / bfa_head needs: device views of the inputs, the parameter block (with the
        class hint derived from host-resident lengths), the output tensors and the workspace."""
        if log_probs.dim() != 3:
            raise ValueError("log_probs must be [B, T, C]")
        dev = _device_of(log_probs)
        lp = log_probs.to(device=dev, dtype=torch.float32)
        if lp.stride(2) != 1:
            lp = lp.contiguous()
        B, Tmax, C = lp.shape
        toks = _as_i32(true_seqs, dev)
        if toks.dim() == 1:
            toks = toks.unsqueeze(0)
        Smax = max(1, toks.shape[1])
        if toks.shape[1] == 0:
            toks = torch.zeros((B, 1), dtype=torch.int32, device=dev)
        if class_mask is None:
            class_mask = 0
        elif class_mask == 0:
            # lengths (and tokens) that are still on the host cost nothing to look at: derive the class hint, so
            # that only the K1 classes of this batch are launched (class_mask=None keeps the library's own choice)
            Th, Sh = _host_ints(pred_lens), _host_ints(true_seqs_lens)
            if Th is None and pred_lens is None:
                Th = np.full(B, Tmax, np.int64)
            sil = self.silence_id if self.silence_id is not None else -1
            hint_kw = dict(anchor_pauses=anchor_pauses, simple=simple, n_classes=C, boost_targets=boost_targets,
                           enforce_minimum=enforce_minimum, Smax=Smax)
            if Th is not None and Sh is not None and len(Th) == B and len(Sh) == B:
                tk_host = _host_array(true_seqs)
                has_sil = True if tk_host is None else bool((tk_host == sil).any())
                class_mask = self.hint_and_path(np.clip(Th, 0, Tmax), np.clip(Sh, 0, toks.shape[1]), has_sil, **hint_kw)[0]
            elif self.hint_from_device_lengths and isinstance(true_seqs_lens, torch.Tensor):
                class_mask = self._hint_of_device_lengths(pred_lens, true_seqs_lens, true_seqs, B, Tmax, toks, sil, hint_kw)
        S_len = _as_i32(true_seqs_lens, dev)
        T_len = _as_i32(pred_lens, dev)
        params = self._params(boost_targets, enforce_minimum, anchor_pauses, simple, max_blanks)
        params.class_mask = int(class_mask)
        params.window_max_tokens = int(self.window_max_tokens or 0)
        params.window_max_frames = int(self.window_max_frames or 0)
        if seg_cap is None:
            seg_cap = Smax + 2 if self.ignore_noise else Tmax + 1
        L = _lib.lib()
        nbytes = L.bfa_workspace_bytes(B, Tmax, Smax, C, ctypes.byref(params))
        c = dict(dev=dev, lp=lp, B=B, Tmax=Tmax, C=C, toks=toks, Smax=Smax, S_len=S_len, T_len=T_len, params=params,
                 seg_cap=seg_cap, ws=self._ws.get(nbytes, dev),
                 segs=torch.empty((B, seg_cap, 4), dtype=torch.int32, device=dev),
                 seg_count=torch.empty((B,), dtype=torch.int32, device=dev),
                 status=torch.empty((B,), dtype=torch.int32, device=dev),
                 mode=torch.empty((B,), dtype=torch.int32, device=dev),
                 fph=torch.empty((B, Tmax), dtype=torch.int32, device=dev),
                 fidx=torch.empty((B, Tmax), dtype=torch.int32, device=dev))
        return c

    @staticmethod
    def _result(c):
        res = AlignmentResult(c["segs"], c["seg_count"], c["status"], c["mode"], c["fph"], c["fidx"], c["T_len"], c["S_len"])
        res._keepalive = (c["lp"], c["toks"], c["ws"])
        res._call = (c["B"], c["Tmax"], c["Smax"], c["C"], c["params"], c["ws"], c["dev"])
        return res

    def align_batch(self, log_probs, true_seqs, pred_lens, true_seqs_lens, boost_targets=True, enforce_minimum=True,
                    anchor_pauses=True, simple=False, seg_cap=None, max_blanks=10, class_mask=0):
        """Whole-batch device call (bfa_align_batch).  Returns an AlignmentResult of device tensors; nothing
        is synchronised or copied to the host here."""
        c = self._prepare_call(log_probs, true_seqs, pred_lens, true_seqs_lens, boost_targets, enforce_minimum,
                               anchor_pauses, simple, seg_cap, max_blanks, class_mask)
        dev, lp, T_len = c["dev"], c["lp"], c["T_len"]
        L = _lib.lib()
        h = _lib.handle(dev.index if dev.index is not None else torch.cuda.current_device(), self.handle_slot)
        stream = torch.cuda.current_stream(dev).cuda_stream
        with torch.cuda.device(dev):
            rc = L.bfa_align_batch(h, lp.data_ptr(), lp.stride(0), lp.stride(1), c["B"], c["Tmax"], c["C"],
                                   T_len.data_ptr() if T_len is not None else None, c["toks"].data_ptr(),
                                   c["S_len"].data_ptr(), c["Smax"], ctypes.byref(c["params"]), c["fph"].data_ptr(),
                                   c["fidx"].data_ptr(), c["segs"].data_ptr(), c["seg_cap"], c["seg_count"].data_ptr(),
                                   c["status"].data_ptr(), c["mode"].data_ptr(), c["ws"].data_ptr(), c["ws"].numel(),
                                   stream)
        _lib.check(rc, h, "bfa_align_batch")
        return self._result(c)

    def prepare_emissions(self, log_probs, true_seqs, pred_lens, true_seqs_lens, boost_targets=True, enforce_minimum=True):
        """The reference's `modified_log_probs` (forced_alignment.py:121-129: _boost_target_phonemes then
        _enforce_minimum_probabilities) for a batch, as a device tensor [B,T,C] (rows beyond pred_lens are zero)."""
        dev = _device_of(log_probs)
        lp = log_probs.to(device=dev, dtype=torch.float32)
        if lp.stride(2) != 1:
            lp = lp.contiguous()
        B, Tmax, C = lp.shape
        toks = _as_i32(true_seqs, dev)
        if toks.dim() == 1:
            toks = toks.unsqueeze(0)
        Smax = max(1, toks.shape[1])
        S_len = _as_i32(true_seqs_lens, dev)
        T_len = _as_i32(pred_lens, dev)
        params = self._params(boost_targets, enforce_minimum, False)
        L = _lib.lib()
        h = _lib.handle(dev.index if dev.index is not None else torch.cuda.current_device())
        ws = self._ws.get(L.bfa_workspace_bytes(B, Tmax, Smax, C, ctypes.byref(params)), dev)
        out = torch.zeros_like(lp)
        with torch.cuda.device(dev):
            rc = L.bfa_prepare_emissions(h, lp.data_ptr(), lp.stride(0), lp.stride(1), B, Tmax, C,
                                         T_len.data_ptr() if T_len is not None else None, toks.data_ptr(),
                                         S_len.data_ptr(), Smax, ctypes.byref(params), out.data_ptr(), out.stride(0),
                                         out.stride(1), ws.data_ptr(), ws.numel(),
                                         torch.cuda.current_stream(dev).cuda_stream)
        _lib.check(rc, h, "bfa_prepare_emissions")
        return out

    def decode_with_forced_alignment(self, log_probs, true_sequence, return_scores=False, boost_targets=True,
                                     enforce_minimum=True, anchor_pauses=True, debug=False):
        """forced_alignment.py:87-199 for one utterance: log_probs [T, C], true_sequence [S].
        Returns (frame_phonemes[T], frame_phonemes_idx[T], score-or-None) as int64 tensors."""
        if self.blank_id is None:
            raise ValueError("Blank ID not set. Call set_blank_id first.")
        T = log_probs.shape[0]
        S = int(true_sequence.shape[0])
        dev = _device_of(log_probs)
        if S == 0:  # :112-118
            fp = torch.full((T,), int(self.blank_id), dtype=torch.long, device=dev)
            fi = torch.full((T,), -1, dtype=torch.long, device=dev)
            score = log_probs[:, self.blank_id].sum() if return_scores else None
            return fp, fi, score
        res = self.align_batch(log_probs.unsqueeze(0), true_sequence.reshape(1, -1), [T], [S],
                               boost_targets=boost_targets, enforce_minimum=enforce_minimum,
                               anchor_pauses=anchor_pauses and self.silence_id is not None, seg_cap=max(T, 1))
        res.raise_for_status()
        fp = res.frame_phonemes[0, :T].long()
        fi = res.frame_phonemes_idx[0, :T].long()
        score = self._calculate_alignment_score(log_probs, fp) if return_scores else None
        return fp, fi, score

    def _calculate_alignment_score(self, log_probs, frame_phonemes):
        """forced_alignment.py:767-773 (python-float accumulation == float64 sum)."""
        lp = log_probs.to(frame_phonemes.device)
        ok = frame_phonemes < lp.shape[1]
        idx = torch.where(ok, frame_phonemes, torch.zeros_like(frame_phonemes))
        vals = lp[torch.arange(lp.shape[0], device=lp.device), idx].double()
        return float((vals * ok.double()).sum().item())

    def assort_frames(self, frame_phonemes, frame_phonemes_idx, max_blanks=10):
        """forced_alignment.py:777-834 as a standalone call (host arithmetic on integer arrays; the batch path
        runs the same rule in the k_assort kernel)."""
        if len(frame_phonemes) == 0:
            return []
        ph = np.asarray(frame_phonemes.cpu() if isinstance(frame_phonemes, torch.Tensor) else frame_phonemes)
        ix = np.asarray(frame_phonemes_idx.cpu() if isinstance(frame_phonemes_idx, torch.Tensor) else frame_phonemes_idx)
        change = np.ones(len(ph), bool)
        change[1:] = (ph[1:] != ph[:-1]) | (ix[1:] != ix[:-1])
        starts = np.flatnonzero(change)
        ends = np.append(starts[1:], len(ph))
        out = []
        for s, e in zip(starts.tolist(), ends.tolist()):
            p, i = int(ph[s]), int(ix[s])
            if p == self.blank_id:
                if (not self.ignore_noise) and (e - s) > max_blanks:
                    out.append((p, s, e, i))
            else:
                out.append((p, s, e, i))
        return out


class AlignmentUtils:
    """Mirror of forced_alignment.py:836-987."""

    def __init__(self, blank_id, silence_id, silence_anchors=10, ignore_noise=True, truly_forced=True):
        self.blank_id = blank_id
        self.silence_id = silence_id
        self.silence_anchors = silence_anchors
        self.truly_forced = truly_forced
        self.viterbi_decoder = ViterbiDecoder(blank_id, silence_id, silence_anchors=self.silence_anchors,
                                              ignore_noise=ignore_noise, truly_forced=self.truly_forced)

    def decode_alignments_device(self, log_probs, true_seqs, pred_lens, true_seqs_lens, boost_targets=True,
                                 enforce_minimum=True, seg_cap=None, class_mask=0):
        """decode_alignments without the host round trip: returns an AlignmentResult (device tensors)."""
        return self.viterbi_decoder.align_batch(log_probs, true_seqs, pred_lens, true_seqs_lens,
                                                boost_targets=boost_targets, enforce_minimum=enforce_minimum,
                                                anchor_pauses=self.silence_anchors > 0, seg_cap=seg_cap,
                                                class_mask=class_mask)

    def decode_alignments(self, log_probs, true_seqs=None, pred_lens=None, true_seqs_lens=None,
                          forced_alignment=True, boost_targets=True, enforce_minimum=True, debug=False, lazy=False):
        """forced_alignment.py:856-910.  Returns list[B] of list[(phoneme_id, start, end, target_seq_idx)] -- a plain list
        of lists of tuples, like the reference.  `lazy=True` (not in the reference): a LazyRowLists over one packed host
        array instead, which builds an utterance's tuples when it is looked at (the 163 840 tuples of the headline batch
        cost CPython 14 ms, forty times the device step)."""
        if not forced_alignment:
            raise NotImplementedError("free decoding (forced_alignment=False) is outside the accelerated path")
        if (true_seqs is None) or (true_seqs_lens is None):
            raise ValueError("Phoneme sequences and lengths required for forced alignment")  # :878-879
        res = self.decode_alignments_device(log_probs, true_seqs, pred_lens, true_seqs_lens,
                                            boost_targets=boost_targets, enforce_minimum=enforce_minimum)
        out = res.to_lists(check_status=True)  # (the reference's exceptions first, from the same round of copies)
        return out if lazy else out.tolist()

    def decode_alignments_simple(self, log_probs, true_seqs, pred_lens=None, true_seqs_lens=None, lazy=False):
        """forced_alignment.py:932-987 (no boost / floor / anchoring; float32 band arithmetic)."""
        B, Tmax = log_probs.shape[0], log_probs.shape[1]
        if pred_lens is None:
            pred_lens = [Tmax] * B
        if true_seqs_lens is None:
            true_seqs_lens = [true_seqs.shape[1]] * B
        res = self.viterbi_decoder.align_batch(log_probs, true_seqs, pred_lens, true_seqs_lens, boost_targets=False,
                                               enforce_minimum=False, anchor_pauses=False, simple=True)
        res.raise_for_status()
        out = res.to_lists()
        return out if lazy else out.tolist()


def align_heads(utils_list, logits_list, seqs_list, pred_lens, true_seqs_lens, boost_targets=True, enforce_minimum=True,
                seg_cap=None, class_masks=None, post=None):
    """core.py:897-922 from the model's RAW logits for every head in ONE library call (bfa_align_heads): the
    log_softmax of core.py:898-899 is fused into the alignment kernels, no log-prob matrix is written.  `utils_list` are
    the heads' AlignmentUtils (phoneme head first), `logits_list` / `seqs_list` their [B,T,C] logits and [B,S] targets.
    Returns [(AlignmentResult, row_stats [B,T,2]) per head]; later stages take (logits, row_stats) in place of log-probs
    (calculate_confidences_batch / postprocess_batch `row_stats=`).  `class_masks`: per head, the optional
    bfa_params.class_mask hint (ViterbiDecoder.class_mask_hint) for callers whose lengths live on the device; by default
    it is derived from host-resident lengths / targets like in align_batch.  The call runs on the library handle of the
    first head's decoder (`viterbi_decoder.handle_slot`).
    `post` = {"extend": bool, "boundary_softness": int, "confidences": bool}: the post-DP stages of every head
    (core.py:925-937: coverage + soft boundaries in place on the result's segs / seg_count, then the confidences of the final
    tuples) are enqueued by the same library call right behind each head's alignment on the stream that head runs on, so
    that they overlap the other head's alignment; the result then carries `.conf` [B,seg_cap] / `.conf_status` [B]."""
    calls = []
    for k, (au, lg, sq) in enumerate(zip(utils_list, logits_list, seqs_list)):
        vd = au.viterbi_decoder
        c = vd._prepare_call(lg, sq, pred_lens, true_seqs_lens, boost_targets, enforce_minimum,
                             au.silence_anchors > 0, False, seg_cap, 10, class_masks[k] if class_masks else 0)
        c["stats"] = torch.empty((c["B"], c["Tmax"], 2), dtype=torch.float32, device=c["dev"])
        calls.append(c)
    c0 = calls[0]
    for c in calls[1:]:
        if (c["B"], c["Tmax"]) != (c0["B"], c0["Tmax"]) or c["dev"] != c0["dev"]:
            raise ValueError("all heads must share batch size, frame count and device")
    heads = (_lib.BfaHead * len(calls))()
    for k, c in enumerate(calls):
        hd = heads[k]
        hd.logits, hd.strideB, hd.strideT = c["lp"].data_ptr(), c["lp"].stride(0), c["lp"].stride(1)
        hd.C, hd.Smax, hd.tokens, hd.params = c["C"], c["Smax"], c["toks"].data_ptr(), c["params"]
        hd.out_row_stats = c["stats"].data_ptr()
        hd.out_frame_phoneme, hd.out_frame_idx = c["fph"].data_ptr(), c["fidx"].data_ptr()
        hd.out_segs, hd.seg_cap = c["segs"].data_ptr(), c["seg_cap"]
        hd.out_seg_count, hd.out_status, hd.out_mode = c["seg_count"].data_ptr(), c["status"].data_ptr(), c["mode"].data_ptr()
        hd.workspace, hd.workspace_bytes = c["ws"].data_ptr(), c["ws"].numel()
        if post is not None:
            hd.postprocess, hd.extend = 1, int(bool(post.get("extend", True)))
            hd.boundary_softness = int(post.get("boundary_softness", 3))
            if post.get("confidences", True):
                c["conf"] = torch.empty((c["B"], c["seg_cap"]), dtype=torch.float32, device=c["dev"])  # (written in full by the kernel)
                c["cstat"] = torch.empty((c["B"],), dtype=torch.int32, device=c["dev"])
                hd.out_conf, hd.out_conf_status = c["conf"].data_ptr(), c["cstat"].data_ptr()
    dev = c0["dev"]
    L = _lib.lib()
    h = _lib.handle(dev.index if dev.index is not None else torch.cuda.current_device(),
                    utils_list[0].viterbi_decoder.handle_slot)
    T_len = c0["T_len"]
    with torch.cuda.device(dev):
        rc = L.bfa_align_heads(h, heads, len(calls), c0["B"], c0["Tmax"], T_len.data_ptr() if T_len is not None else None,
                               c0["S_len"].data_ptr(), torch.cuda.current_stream(dev).cuda_stream)
    _lib.check(rc, h, "bfa_align_heads")
    out = []
    for c in calls:
        res = ViterbiDecoder._result(c)
        res.conf, res.conf_status = c.get("conf"), c.get("cstat")
        res.postprocessed = post is not None
        out.append((res, c["stats"]))
    return out
