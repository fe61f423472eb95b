"""API surface of bournemouth_aligner/core.py::PhonemeTimestampAligner for the accelerated path.

Only the part of the reference class that sits on the hot path is mirrored: decoder setup
(core.py:252-257), extract_timestamps_from_segment_batch (core.py:811-992) from the log-softmax
onwards, and the thin process_sentence / process_sentences_batch / process_segments wrappers
(core.py:1553-1616, 1212).  The acoustic model (CUPE-2i) and the espeak phonemiser are NOT part of
this package: they are injected as callables and stay on PyTorch-ROCm / the host.

    posterior_fn(wavs, wav_lens) -> (logits_class [B,T,C_p], logits_group [B,T,C_g], spectral_lens list[int])
    phonemizer(text)             -> dict(ph66=[ids], pg16=[group ids] (optional), words/word_num/eipa (optional))
"""
import itertools
import threading

import numpy as np
import torch

from .coverage import ensure_target_coverage
from .forced_alignment import AlignmentUtils, HostLens, LazyRowDicts, LazyRowLists, align_heads, rows_as_tuple_lists
from .utils import calculate_confidences_batch, convert_to_ms, log_softmax, postprocess_batch

# one row of extract_timestamps_from_segment_batch's result (core.py:939-956)
_ROW8 = np.dtype([("id", "<i4"), ("start", "<i4"), ("end", "<i4"), ("idx", "<i4"), ("est", "?"), ("conf", "<f4"),
                  ("start_ms", "<f4"), ("end_ms", "<f4")])


_STAGE = threading.local()  # per host thread: {device -> pinned uint8 block, grow-only}, the staging area of _to_host


def _to_host(*tensors):
    """device tensors -> numpy arrays: the copies go back to back into ONE persistent pinned staging block (one
    synchronisation, ~50 GB/s instead of a pageable copy each), and the caller gets fresh numpy arrays copied out of it.
    (Handing out the pinned memory itself was measured and lost: 27.8 against 6.9 ms per headline call -- blocks the caller
    still holds cannot be recycled and every call allocated fresh pinned memory.)"""
    dev = next((t.device for t in tensors if t.is_cuda), None)
    if dev is None:
        return [t.numpy() for t in tensors]
    sizes = [(t.numel() * t.element_size() + 255) & ~255 for t in tensors]
    total = sum(sizes)
    # one block per host thread and device: the GIL is released while the copies and the synchronisation run, so two threads
    # (an aligner / handle slot each) must not stage through the same bytes
    blocks = _STAGE.__dict__.setdefault("blocks", {})
    blk = blocks.get(dev.index)
    if blk is None or blk.numel() < total:
        blk = blocks[dev.index] = torch.empty(max(total, 1 << 20), dtype=torch.uint8, pin_memory=True)
    views, off = [], 0
    for t, sz in zip(tensors, sizes):
        tc = t.contiguous()
        v = blk[off:off + tc.numel() * tc.element_size()].view(tc.dtype).view(tc.shape)
        v.copy_(tc, non_blocking=True)
        views.append(v)
        off += sz
    torch.cuda.current_stream(dev).synchronize()
    return [np.array(v.numpy()) for v in views]


def _pad_rows(rows, fill):
    """list of id lists -> int32 [B, max(1, longest)] tensor padded with `fill` (core.py:848-853)."""
    lens = np.fromiter((len(r) for r in rows), np.int64, len(rows))
    out = np.full((len(rows), max(1, int(lens.max(initial=0)))), fill, np.int32)
    total = int(lens.sum())
    if total:
        try:  # plain ints: no Python-level work per element
            flat = np.fromiter(itertools.chain.from_iterable(rows), np.int32, total)
        except (TypeError, ValueError):
            flat = np.fromiter((int(x) for r in rows for x in r), np.int32, total)
        out[np.arange(out.shape[1])[None, :] < lens[:, None]] = flat
    return torch.from_numpy(out)


class PhonemeTimestampAligner:
    """Mirror of core.py:34-234 at class level.  The positional / keyword arguments of the reference constructor are
    accepted with their defaults; the acoustic model and the espeak phonemiser are outside this package, so

    * `preset` / `model_name` / `cupe_ckpt_path` select nothing here: no checkpoint is ever loaded.  The posterior
      producer is the keyword-only `posterior_fn` (the CUPE model on PyTorch-ROCm in production, anything in tests);
      without one `self.extractor` stays None and the extraction entry points raise the reference's own
      "model is not loaded" error (core.py:886) -- exactly what the reference does after `preset=None`;
    * `lang` is only recorded (the injected `phonemizer` owns the language);
    * `mapper` other than "ph66" raises like core.py:141-142; `duration_max` sets `wav_len_max` (core.py:136-138).
    """

    def __init__(self, preset="en-us", model_name=None, cupe_ckpt_path=None, lang="en-us", mapper="ph66",
                 duration_max=30, device="auto", silence_anchors=10, boost_targets=True, enforce_minimum=True,
                 enforce_all_targets=True, ensure_completeness=False, ignore_noise=True, extend_soft_boundaries=True,
                 boundary_softness=3, bad_confidence_threshold=0.6, *, posterior_fn=None, phonemizer=None,
                 phoneme_id_to_group_id=None, blank_class=66, silence_class=0, blank_group=16, silence_group=0,
                 sample_rate=16000, phoneme_id_to_label=None, group_id_to_label=None):
        self.warn_level = 1
        if device == "auto":
            device = "cuda"  # this package has no CPU implementation of the alignment path
        self.device = torch.device(device)
        self.preset, self.model_name, self.cupe_ckpt_path = preset, model_name, cupe_ckpt_path
        self.lang = lang
        self.posterior_fn = posterior_fn
        self.extractor = posterior_fn  # the reference's attribute name for "the model" (None = not loaded)
        self.phonemizer = phonemizer
        self.resampler_sample_rate = sample_rate
        self.sample_rate = sample_rate
        self.padding_ph_label = -100
        self.ph_seq_min = 1
        self.seg_duration_min = 0.05
        self.seg_duration_min_samples = int(self.seg_duration_min * self.resampler_sample_rate)
        self.seg_duration_max = duration_max
        self.wav_len_max = int(self.seg_duration_max * self.resampler_sample_rate)
        self.selected_mapper = mapper
        if self.selected_mapper != "ph66":
            raise ValueError("Currently only 'ph66' mapper is supported.")
        self.phonemes_key = getattr(phonemizer, "phonemes_key", "ph66")
        self.phoneme_groups_key = getattr(phonemizer, "phoneme_groups_key", "pg16")
        self.phoneme_id_to_group_id = phoneme_id_to_group_id if phoneme_id_to_group_id is not None else \
            getattr(phonemizer, "phoneme_id_to_group_id", None)
        self.blank_class, self.silence_class = blank_class, silence_class
        self.blank_group, self.silence_group = blank_group, silence_group
        self.silence_anchors = silence_anchors
        self.boost_targets = boost_targets
        self.enforce_minimum = enforce_minimum
        self.enforce_all_targets = enforce_all_targets
        self.ensure_completeness = ensure_completeness
        self._shape_pool = None  # two threads for the result shaping of the two heads (extract_timestamps_from_logits)
        self.ignore_noise = ignore_noise
        self.extend_soft_boundaries = extend_soft_boundaries
        self.boundary_softness = boundary_softness
        self.bad_confidence_threshold = bad_confidence_threshold
        self.phoneme_id_to_label = phoneme_id_to_label or getattr(phonemizer, "index_to_plabel", None) or {}
        self.group_id_to_label = group_id_to_label or getattr(phonemizer, "index_to_glabel", None) or {}
        self._setup_decoders()
        self.reset_counters()

    def reset_counters(self):
        """core.py:184-194"""
        self.total_segments_processed = 0
        self.total_segments_bad = 0
        self.total_segments_failed = 0
        self.total_phonemes_aligned = 0
        self.total_phonemes_target = 0
        self.total_phonemes_aligned_easily = 0
        self.total_phonemes_missed = 0
        self.total_phonemes_extra = 0
        self.perfect_matches = 0

    def phonemize_sentence(self, text):
        """core.py: the injected phonemiser (a callable, or an object with .phonemize_sentence like the reference's
        ph66 Phonemizer)."""
        if self.phonemizer is None:
            raise ValueError("process_segments needs a phonemizer (espeak is not part of this package): pass "
                             "phonemizer=callable(text) -> {'ph66': [...], 'pg16': [...], ...}")
        fn = getattr(self.phonemizer, "phonemize_sentence", None) or self.phonemizer
        return fn(text)

    @torch.no_grad()
    def _chop_segments(self, clips, spans):
        """The audio front end of process_segments for ALL sub-segments at once (what core.py:276-320 does one segment
        at a time): segment i is samples [lo_i, hi_i) of clip i (a (channels, samples) tensor), down-mixed to mono,
        scaled to unit RMS over the whole segment, and laid into row i of one zero-filled [N, wav_len_max] matrix
        (longer segments are cut).  Returns (matrix, lengths, codes): code -1 = the requested span is shorter than
        `seg_duration_min_samples` (an end of -1 counts as that), -2 = the clip ends before the span is long enough;
        rows with a non-zero code stay zero and have length 0."""
        n = len(spans)
        codes = [0] * n
        lengths = [0] * n
        first = clips[0]
        out = torch.zeros((n, self.wav_len_max), dtype=first.dtype if first.is_floating_point() else torch.float32,
                          device=first.device)
        for i, (clip, (lo, hi)) in enumerate(zip(clips, spans)):
            if hi == -1 or hi - lo < self.seg_duration_min_samples:
                codes[i] = -1
                continue
            piece = clip[:, lo:hi]
            if piece.shape[1] < self.seg_duration_min_samples:
                codes[i] = -2
                continue
            mono = piece.mean(dim=0)
            level = mono.square().mean().sqrt()
            if level > 0:
                mono = mono / level
            k = min(mono.shape[0], self.wav_len_max)
            out[i, :k] = mono[:k]
            lengths[i] = k
        return out, lengths, codes

    def chop_wav(self, wav, start_frame, end_frame):
        """core.py:288-320, the reference's public single-segment front end (its examples and tests call it): one
        (channels, samples) clip -> (mono unit-RMS row of wav_len_max samples, valid length, 0), or (None, None, -1 / -2)
        with the reference's messages.  A one-row call of `_chop_segments`."""
        rows, lengths, codes = self._chop_segments([wav], [(start_frame, end_frame)])
        if codes[0] == -1:
            asked = (end_frame - start_frame) if end_frame != -1 else -1
            print(f"ERROR: Segment too short: {asked} frames, minimum required is {self.seg_duration_min_samples} frames.")
            return None, None, -1
        if codes[0] == -2:
            print(f"Wav shape is too small: {wav[:, start_frame:end_frame].shape}, start_frame: {start_frame}, end_frame: {end_frame}")
            return None, None, -2
        return rows[0], lengths[0], 0

    @staticmethod
    def _rms_normalize(audio):
        """core.py:322-328: unit RMS (silence stays silence)."""
        level = audio.square().mean().sqrt()
        return audio / level if level > 0 else audio

    def _setup_decoders(self):
        """core.py:252-257"""
        self.alignment_utils_g = AlignmentUtils(blank_id=self.blank_group, silence_id=self.silence_group,
                                                silence_anchors=self.silence_anchors, ignore_noise=self.ignore_noise,
                                                truly_forced=self.enforce_all_targets)
        self.alignment_utils_p = AlignmentUtils(blank_id=self.blank_class, silence_id=self.silence_class,
                                                silence_anchors=self.silence_anchors, ignore_noise=self.ignore_noise,
                                                truly_forced=self.enforce_all_targets)

    def _map_phonemes_to_groups(self, seq):
        if self.phoneme_id_to_group_id is None:
            raise ValueError("group_sequences not given and no phoneme_id_to_group_id mapping was configured")
        return [self.phoneme_id_to_group_id.get(int(p), self.blank_group) for p in seq]

    # ---- the reference's two public post-DP methods, for callers that use them directly (core.py:462, 682)
    def ensure_target_coverage(self, phoneme_sequences, aligned_frames, seq_lens=None, _silence_class=0, debug=False):
        """core.py:462-680 on host lists: rows (phoneme, start, end, target_idx) -> 5-tuples with the `is_estimated` flag.
        With ensure_completeness the repair of coverage.py; by default rows with target index -1 / beyond the sequence are
        dropped and the rest stably sorted by start frame (what bfa_postprocess does on the device inside
        extract_timestamps_from_logits)."""
        if self.ensure_completeness:
            return ensure_target_coverage(phoneme_sequences, aligned_frames, seq_lens, _silence_class)
        out = []
        for b, rows in enumerate(aligned_frames):
            seq = phoneme_sequences[b]
            n = int(seq_lens[b]) if seq_lens is not None else len(seq)
            keep = sorted((r for r in rows if int(r[3]) != -1 and int(r[3]) < n), key=lambda r: r[1])
            out.append([tuple(r) if len(r) != 4 else (*r, False) for r in keep])
        return out

    def extend_soft_boundaries_func(self, log_probs, framestamps, boundary_softness=3, debug=False):
        """core.py:682-809 for callers that hold log-probs [B, T, C] and lists of (phoneme, start, end, target_idx,
        is_estimated): the four extension passes on the device (bfa_postprocess), rows returned in the same structure.
        The device stage works on rows in start-frame order with valid target indices -- what ensure_target_coverage
        returns; anything else is refused rather than silently reordered."""
        B = len(framestamps)
        cap = max(1, max((len(r) for r in framestamps), default=1))
        host = np.zeros((B, cap, 4), np.int32)
        cnt = np.zeros(B, np.int32)
        for b, rows in enumerate(framestamps):
            starts = [int(r[1]) for r in rows]
            if any(x > y for x, y in zip(starts, starts[1:])) or any(int(r[3]) < 0 for r in rows):
                raise ValueError("extend_soft_boundaries_func: rows must be sorted by start frame with target indices >= 0 "
                                 "(the output of ensure_target_coverage)")
            if rows:
                host[b, :len(rows)] = [[int(r[0]), int(r[1]), int(r[2]), int(r[3])] for r in rows]
            cnt[b] = len(rows)
        dev = self.device
        segs, count = torch.from_numpy(host).to(dev), torch.from_numpy(cnt).to(dev)
        keep_all = torch.full((B,), 2 ** 31 - 1, dtype=torch.int32, device=dev)   # (no row is dropped: indices are < this)
        postprocess_batch(log_probs.to(dev), keep_all, segs, count, extend=True, boundary_softness=boundary_softness)
        got, n = _to_host(segs, count)
        out = []
        for b, rows in enumerate(framestamps):
            assert int(n[b]) == len(rows)
            out.append([(int(r[0]), int(g[1]), int(g[2]), int(r[3])) + tuple(r[4:]) for r, g in zip(rows, got[b])])
        return out

    # ---- one head: align -> coverage -> soft boundaries -> confidences, all on the device
    def _head(self, utils, log_probs, seqs, seq_lens, spectral_lens, aligned=None):
        """One head after the alignment: coverage -> soft boundaries -> confidences on the device.  `aligned` =
        (AlignmentResult, row_stats) from the fused bfa_align_heads call, in which case `log_probs` holds the raw
        logits; otherwise the head is aligned here from log-probs (the two-pass path)."""
        if aligned is None:
            res = utils.decode_alignments_device(log_probs, seqs, spectral_lens, seq_lens,
                                                 boost_targets=self.boost_targets, enforce_minimum=self.enforce_minimum)
            stats = None
        else:
            res, stats = aligned
        estimated = None
        if self.ensure_completeness:
            # rare-case repair over a handful of rows per utterance: host side (coverage.py), between the device
            # alignment and the device soft-boundary / confidence passes.  The completed rows are valid and sorted,
            # so the device coverage stage of bfa_postprocess leaves them as they are.
            res.raise_for_status()
            rows = ensure_target_coverage(seqs, res.to_lists(), seq_lens, utils.viterbi_decoder.silence_id)
            cap = res.segs.shape[1]
            host = np.zeros((len(rows), cap, 4), np.int32)
            for b, rs in enumerate(rows):
                if len(rs) > cap:
                    raise RuntimeError(f"item {b}: {len(rs)} completed rows exceed seg_cap {cap}")
                if rs:
                    host[b, :len(rs)] = [r[:4] for r in rs]
            res.segs.copy_(torch.from_numpy(host))
            res.seg_count.copy_(torch.tensor([len(rs) for rs in rows], dtype=torch.int32))
            estimated = [[bool(r[4]) for r in rs] for rs in rows]
        if res.postprocessed and res.conf is not None:  # the fused call already ran coverage / soft boundaries / confidences on the head's stream
            return res, res.conf, res.conf_status, estimated
        if not res.postprocessed:  # (a result that only carries confidences -- BatchesInFlight.submit(confidences=True) -- still needs the stages)
            postprocess_batch(log_probs, seq_lens, res.segs, res.seg_count, extend=self.extend_soft_boundaries,
                              boundary_softness=self.boundary_softness, row_stats=stats)
            res.postprocessed = True
        conf, cstat = calculate_confidences_batch(log_probs, res.segs, res.seg_count, row_stats=stats)  # padded rows (core.py:936)
        return res, conf, cstat, estimated

    def extract_timestamps_from_logits(self, logits_class, logits_group, spectral_lens, phoneme_sequences, wav_lens,
                                       start_offset_times=0, group_sequences=None, do_groups=True, as_arrays=False,
                                       fused=True, lazy=False):
        """core.py:897-964 given the model's logits.  Returns list[B] of dicts with 'phoneme_timestamps' and
        'group_timestamps': lists of (id, start_frame, end_frame, target_seq_idx, is_estimated, confidence,
        start_ms, end_ms).  With `as_arrays` the same data as padded numpy arrays per head ({'rows' [B,cap,4],
        'count' [B], 'is_estimated', 'confidence', 'start_ms', 'end_ms' [B,cap]}) for bulk consumers; with `lazy` a
        LazyRowDicts that builds an utterance's dict and tuples when it is looked at (neither flag is in the reference;
        the default is the reference's plain list of dicts of lists of tuples)."""
        dev = self.device
        B = logits_class.shape[0]
        if isinstance(phoneme_sequences, torch.Tensor):
            ph_seq_lens = (phoneme_sequences != self.blank_class).sum(dim=1).cpu().numpy()  # core.py:844
            ph = phoneme_sequences.to(torch.int32).cpu()
        else:
            ph_seq_lens = [len(s) for s in phoneme_sequences]
            ph = _pad_rows(phoneme_sequences, self.blank_class)
        if group_sequences is None:  # core.py:868-871, as one table lookup over the padded batch
            if self.phoneme_id_to_group_id is None:
                raise ValueError("group_sequences not given and no phoneme_id_to_group_id mapping was configured")
            ids = ph.numpy().astype(np.int64)
            top = max(int(ids.max(initial=0)), max(self.phoneme_id_to_group_id, default=0)) + 1
            lut = np.full(top, self.blank_group, np.int32)
            for k, v in self.phoneme_id_to_group_id.items():
                if 0 <= int(k) < top:
                    lut[int(k)] = v
            grp = np.where(ids >= 0, lut[np.clip(ids, 0, top - 1)], self.blank_group).astype(np.int32)
            grp[np.arange(ids.shape[1])[None, :] >= np.asarray(ph_seq_lens, np.int64)[:, None]] = self.blank_group
            gr = torch.from_numpy(grp)
        elif not isinstance(group_sequences, torch.Tensor):
            gr = _pad_rows(group_sequences, self.blank_group)
        else:
            gr = group_sequences.to(torch.int32)
        # the two length vectors: converted and uploaded once, shared by both heads, the post-DP stages and the shaping
        spec = HostLens(spectral_lens)
        ph_seq_lens = HostLens(ph_seq_lens)
        if fused:
            # raw logits of both heads in, log_softmax (core.py:898-899) inside the alignment
            # kernels, both heads from one bfa_align_heads call; the later stages read (logits, row statistics)
            xs = [logits_class.to(device=dev, dtype=torch.float32), logits_group.to(device=dev, dtype=torch.float32)]
            us = [self.alignment_utils_p, self.alignment_utils_g]
            # (with ensure_completeness the host-side repair sits between the alignment and the post-DP stages)
            post = None if self.ensure_completeness else {"extend": self.extend_soft_boundaries,
                                                          "boundary_softness": self.boundary_softness}
            aligned = align_heads(us, xs, [ph, gr], spec, ph_seq_lens, boost_targets=self.boost_targets,
                                  enforce_minimum=self.enforce_minimum, post=post)
            pending = [(key, self._head(u, x, sq, ph_seq_lens, spec, aligned=al))
                       for key, u, x, sq, al in zip(("phoneme_timestamps", "group_timestamps"), us, xs, [ph, gr], aligned)]
        else:
            lp_p = log_softmax(logits_class.to(dev))  # core.py:898-899
            lp_g = log_softmax(logits_group.to(dev))
            heads = [("phoneme_timestamps", self.alignment_utils_p, lp_p, ph)]
            heads.append(("group_timestamps", self.alignment_utils_g, lp_g, gr))  # always runs (core.py:914)
            pending = [(key, self._head(utils, lp, seqs, ph_seq_lens, spec)) for key, utils, lp, seqs in heads]
        arrays = {}
        # every head's results in ONE round of copies (pinned, one synchronisation) instead of five blocking ones per head
        flat = _to_host(*[t for _, (res, conf, cstat, _e) in pending for t in (res.status, cstat, res.seg_count, res.segs, conf)])
        scale = self._ms_scale(spec, wav_lens, start_offset_times, B)
        jobs = []
        for k, (key, (res, conf, cstat, estimated)) in enumerate(pending):
            st_h, cs_h = flat[5 * k], flat[5 * k + 1]
            if (st_h != 0).any():
                res.raise_for_status()
            if (cs_h != 0).any():
                raise IndexError("confidence pass: phoneme id or start frame out of range")
            res._host3 = flat[5 * k + 2:5 * k + 5]
            jobs.append((key, res, conf, estimated))
        if len(jobs) > 1 and B >= 1024:   # the heads side by side: the shaping is numpy passes over [B, cap] arrays (no GIL)
            if self._shape_pool is None:
                import concurrent.futures
                self._shape_pool = concurrent.futures.ThreadPoolExecutor(max_workers=2)
            futs = [(key, self._shape_pool.submit(self._shape_rows, res, conf, estimated, spec, wav_lens, start_offset_times,
                                                  as_arrays, scale)) for key, res, conf, estimated in jobs]
            for key, f in futs:
                arrays[key] = f.result()
        else:
            for key, res, conf, estimated in jobs:
                arrays[key] = self._shape_rows(res, conf, estimated, spec, wav_lens, start_offset_times, as_arrays, scale)
        for _key, res, _c, _e in jobs:
            res._host3 = None
        # (the reference returns a list of B dicts of lists of 8-tuples: here that list builds an utterance's dict and tuples
        # when it is looked at -- 327 680 tuples of the headline batch cost CPython ~85 ms whoever builds them)
        if as_arrays:
            return arrays
        out = LazyRowDicts(arrays, B)
        return out if lazy else out.tolist()

    def _ms_scale(self, spec, wav_lens, start_offset_times, B):
        """(seconds per frame, start offset) per utterance in the float32 arithmetic of utils.py:126-142 (a 0-dim int64
        tensor `spectral_length`: reciprocal * duration); shared by the heads of a call"""
        f32 = np.float32
        sl = spec.host if isinstance(spec, HostLens) else np.asarray(spec, np.int64)
        dur = (np.asarray(wav_lens, np.float64) / float(self.resampler_sample_rate)).astype(f32)
        with np.errstate(divide="ignore", invalid="ignore"):
            dpf = np.where(sl > 0, (f32(1) / sl.astype(f32)) * dur, f32(0)).astype(f32)
        per_item = isinstance(start_offset_times, (list, tuple))
        off = np.asarray(start_offset_times if per_item else [start_offset_times] * B, np.float64).astype(f32)
        return dpf, off

    def _shape_rows(self, res, conf, estimated, spec, wav_lens, start_offset_times, as_arrays, scale=None):
        """core.py:939-956 for one head and the whole batch at once: convert_to_ms in the float32 tensor arithmetic
        the reference ends up in (utils.py:128-146 with a 0-dim tensor `spectral_length`), then the rows as the
        reference's 8-tuples -- or, with `as_arrays`, as padded numpy arrays (no per-row Python objects: on a
        4096-utterance batch building the tuples costs far more than the device passes)."""
        f32 = np.float32
        cnt, segs, cf = res._host3 if getattr(res, "_host3", None) is not None else _to_host(res.seg_count, res.segs, conf)
        B, cap = segs.shape[0], segs.shape[1]
        dpf, off = scale if scale is not None else self._ms_scale(spec, wav_lens, start_offset_times, B)
        # start and end frames in one pass: (offset + frame * seconds_per_frame) * 1000, each operation rounded to float32
        ms = segs[:, :, 1:3].astype(f32)
        ms *= dpf[:, None, None]
        ms += off[:, None, None]
        ms *= f32(1000)
        sms, ems = ms[:, :, 0], ms[:, :, 1]
        est = np.zeros((B, cap), bool)
        if estimated:
            for b, flags in enumerate(estimated):
                est[b, :len(flags)] = flags
        # core.py:955-956 sorts by start_ms (stable).  Rounding is monotonic, so rows sorted by start FRAME are sorted by
        # start_ms too and the sort is the identity; only utterances whose start frames are out of order need it
        starts = segs[:, :, 1]
        live = np.arange(1, cap)[None, :] < cnt[:, None]   # (rows beyond an utterance's count hold whatever the buffer held)
        bad = np.flatnonzero(((starts[:, 1:] < starts[:, :-1]) & live).any(axis=1))
        for b in bad:
            n = int(min(cnt[b], cap))
            if (np.diff(sms[b, :n]) < 0).any():
                order = np.argsort(sms[b, :n], kind="stable")
                for a in (segs, cf, sms, ems, est):
                    a[b, :n] = a[b, :n][order]
        if as_arrays:
            return {"rows": segs, "count": cnt, "is_estimated": est, "confidence": cf[:, :cap], "start_ms": sms,
                    "end_ms": ems}
        # the reference's 8-tuples: the valid rows of the whole batch as ONE structured array, turned into tuples by
        # numpy in one pass and cut per utterance -- not B x 8 slices and B zips
        ncl = np.minimum(cnt, cap)
        valid = np.arange(cap)[None, :] < ncl[:, None]
        rec = np.empty(int(ncl.sum()), dtype=_ROW8)
        picked = segs[valid]
        for k, name in enumerate(("id", "start", "end", "idx")):
            rec[name] = picked[:, k]
        rec["est"], rec["conf"], rec["start_ms"], rec["end_ms"] = est[valid], cf[:, :cap][valid], sms[valid], ems[valid]
        return LazyRowLists(rec, ncl)

    def extract_timestamps_from_segment_batch(self, wavs, wav_lens, phoneme_sequences, start_offset_times=0,
                                              group_sequences=None, extract_embeddings=False, do_groups=True,
                                              debug=False):
        """core.py:811-992 (embeddings pooling is not part of the accelerated path)."""
        if self.posterior_fn is None:
            raise AssertionError("CUPE extractor model is not loaded: pass posterior_fn=...")  # core.py:886
        if extract_embeddings:
            raise NotImplementedError("extract_embeddings=True is outside the accelerated path")
        logits_class, logits_group, spectral_lens = self.posterior_fn(wavs, wav_lens)
        ts = self.extract_timestamps_from_logits(logits_class, logits_group, spectral_lens, phoneme_sequences,
                                                 wav_lens, start_offset_times, group_sequences, do_groups)
        n = len(ts)
        return ts, [None] * n, [None] * n

    def extract_timestamps_from_logits_simplified(self, logits_class, spectral_lens, phoneme_sequences, wav_lens,
                                                  start_offset_times=0.0):
        """core.py:1018-1044 given the model's logits: log_softmax -> decode_alignments_simple -> ms.  No boost,
        floor, anchoring, coverage or confidence stage; rows are (id, start_frame, end_frame, target_seq_idx, False,
        0.0, start_ms, end_ms) as convert_to_ms pads 4-tuples (utils.py:133-138), ms in float64 because
        `spectral_lens[b]` is a plain int there."""
        dev = self.device
        B = logits_class.shape[0]
        if isinstance(phoneme_sequences, torch.Tensor):
            ph_seq_lens = [int((row != self.blank_class).sum()) for row in phoneme_sequences]  # core.py:1004-1005
            ph = phoneme_sequences.to(torch.int32)
        else:
            ph_seq_lens = [len(s) for s in phoneme_sequences]
            ph = torch.full((B, max(1, max(ph_seq_lens))), self.blank_class, dtype=torch.int32)
            for b, s in enumerate(phoneme_sequences):
                ph[b, :len(s)] = torch.as_tensor(list(s), dtype=torch.int32)
        spec = [int(x) for x in spectral_lens]
        lp = log_softmax(logits_class.to(dev))
        rows = self.alignment_utils_p.decode_alignments_simple(lp, ph, spec, ph_seq_lens)
        out = []
        for b in range(B):
            off = start_offset_times[b] if isinstance(start_offset_times, (list, tuple)) else start_offset_times
            out.append({"phoneme_timestamps": convert_to_ms(rows[b], spec[b], off, wav_lens[b],
                                                            self.resampler_sample_rate)})
        return out

    def extract_timestamps_from_segment_simplified(self, wavs, wav_lens, phoneme_sequences, start_offset_times=0.0,
                                                   debug=True):
        """core.py:995-1044."""
        if self.posterior_fn is None:
            raise AssertionError("CUPE extractor model is not loaded: pass posterior_fn=...")
        logits_class, _, spectral_lens = self.posterior_fn(wavs, wav_lens)
        return self.extract_timestamps_from_logits_simplified(logits_class, spectral_lens, phoneme_sequences, wav_lens,
                                                              start_offset_times)

    # ---- host-side result shaping (core.py:1062-1210, 1620-1631, 1701-1733, 1780-1919)
    def _align_words(self, phoneme_ts, word_num, words_list):
        """core.py:1062-1120: group consecutive 'phoneme_ts' entries by their word number.  Quirks kept: entries
        are paired with `word_num` by POSITION (not by target_seq_idx), only the first min(len) positions are
        looked at, and a word is closed when the number changes or at the last position of `word_num` -- so a
        final one-phoneme word that starts exactly there is never emitted, and nothing is closed at all when
        `phoneme_ts` is the shorter list."""
        if not phoneme_ts or not word_num:
            return []
        words_ts = []
        last = len(word_num) - 1
        cur, members, start_ms = word_num[0], [], phoneme_ts[0]["start_ms"]
        for i in range(min(len(word_num), len(phoneme_ts))):
            if word_num[i] == cur and i != last:
                members.append(phoneme_ts[i])
                continue
            if word_num[i] == cur:  # the last position, still inside the current word
                members.append(phoneme_ts[i])
            words_ts.append({
                "word": words_list[cur] if cur < len(words_list) else f"UNK_WORD_{cur}",
                "start_ms": start_ms,
                "end_ms": members[-1]["end_ms"],
                "confidence": sum(m["confidence"] for m in members) / len(members),
                "ph66": [m["phoneme_id"] for m in members],
                "ipa": [m["ipa_label"] for m in members],
            })
            if i < last:
                cur, members, start_ms = word_num[i], [phoneme_ts[i]], phoneme_ts[i]["start_ms"]
        return words_ts

    def analyze_alignment_coverage(self, target_sequence, aligned_timestamps, index_to_label):
        """core.py:1701-1733: set-level coverage of the target ids by the aligned ids."""
        target = set(target_sequence.tolist() if hasattr(target_sequence, "tolist") else target_sequence)
        aligned = {row[0] for row in aligned_timestamps}
        missing, extra = target - aligned, aligned - target
        ratio = len(target - missing) / len(target) if target else 1.0
        label = lambda p: index_to_label.get(p, f"UNK_{p}")
        return {"target_count": len(target), "aligned_count": len(aligned), "missing_count": len(missing),
                "extra_count": len(extra), "coverage_ratio": ratio, "missing_phonemes": [label(p) for p in missing],
                "extra_phonemes": [label(p) for p in extra], "bad_alignment": ratio < 0.8}

    def convert_to_textgrid(self, timestamps_dict, output_file=None, include_confidence=False):
        """core.py:1620-1631."""
        from .textgrid import dict_to_textgrid
        text = dict_to_textgrid(timestamps_dict, output_file=None, include_confidence=include_confidence)
        if output_file and text is not None:
            with open(output_file, "w", encoding="utf-8") as f:
                f.write(text)
        return text

    def ceil(self, float_value):
        """core.py:1780-1781 (also for negative values: truncation towards zero plus one if there is a fraction)."""
        return int(float_value) + (float_value % 1 > 0)

    def compress_frames(self, frames_list):
        """core.py:1783-1803: run-length pairs (value, count)."""
        runs = []
        for v in frames_list:
            if runs and runs[-1][0] == v:
                runs[-1][1] += 1
            else:
                runs.append([v, 1])
        return [(v, n) for v, n in runs]

    def decompress_frames(self, compressed_frames):
        """core.py:1806-1811."""
        return [v for v, n in compressed_frames for _ in range(n)]

    def framewise_assortment(self, aligned_ts, total_frames, frames_per_second, gap_contraction=5,
                             select_key="phoneme_id", offset_ms=0):
        """core.py:1813-1919: one label per frame of a `frames_per_second` grid (for TTS consumers) from a list of
        'phoneme_ts' / 'group_ts' / 'words_ts' entries.  Sorts `aligned_ts` in place by start_ms like the
        reference.  Stage 1: each entry claims the still-unlabelled frames of [start-1, end+1) (first come, first
        served in start order).  Stage 2, per unlabelled run that has a labelled frame on its left and either a
        labelled frame on its right or the end of the grid: runs up to `gap_contraction` take the left label, runs
        up to twice that are split at the midpoint, longer runs get `gap_contraction` frames from each side; the
        rest stays -1.  A run ending at the grid's end has no right label: its right part stays -1."""
        GAP = -1
        ms_per_frame = 1000.0 / frames_per_second
        aligned_ts.sort(key=lambda x: x["start_ms"])
        labels = [GAP] * total_frames
        for item in aligned_ts:
            first = max(int((item["start_ms"] - offset_ms) / ms_per_frame), 0)
            end = min(self.ceil((item["end_ms"] - offset_ms) / ms_per_frame), total_frames)
            if end - first > total_frames:
                continue
            if select_key not in item:
                raise ValueError(f"select_key '{select_key}' not found in timestamp item", item)
            for f in range(max(first - 1, 0), min(end + 1, total_frames)):
                if labels[f] == GAP:
                    labels[f] = item[select_key]
        gaps, f = [], 0
        while f < total_frames:  # the unlabelled runs as they are BEFORE any of them is filled
            if labels[f] != GAP:
                f += 1
                continue
            g0 = f
            while f < total_frames and labels[f] == GAP:
                f += 1
            gaps.append((g0, f))
        for g0, g1 in gaps:
            if g0 == 0:
                continue
            left = labels[g0 - 1]
            at_end = g1 == total_frames
            right = GAP if at_end else labels[g1]
            if left == GAP or (right == GAP and not at_end):
                continue
            n = g1 - g0
            if n <= gap_contraction:
                labels[g0:g1] = [left] * n
            elif n <= 2 * gap_contraction:
                mid = g0 + n // 2
                labels[g0:mid] = [left] * (mid - g0)
                labels[mid:g1] = [right] * (g1 - mid)
            else:
                k = max(gap_contraction, 0)
                labels[g0:g0 + k] = [left] * k
                labels[g1 - k:g1] = [right] * k
        return labels

    # ---- thin wrappers (core.py:1212, 1553, 1586)
    def _post_process_segment(self, segment, ts, phoneme_timestamps, group_timestamps=None, phoneme_sequence=None):
        """core.py:1140-1210."""
        out = dict(segment)
        if phoneme_sequence is None:
            phoneme_sequence = ts.get("ph66", [])
        out["coverage_analysis"] = self.analyze_alignment_coverage(phoneme_sequence, phoneme_timestamps,
                                                                   self.phoneme_id_to_label)
        out["ipa"] = ts.get("eipa", "")
        out["word_num"] = ts.get("word_num", "")
        out["words"] = ts.get("words", "")
        out["phoneme_ts"] = [
            {"phoneme_id": int(p), "phoneme_label": self.phoneme_id_to_label.get(p, f"UNK_{p}"),
             "ipa_label": out["ipa"][tidx] if 0 <= tidx < len(out["ipa"]) else "overflow",
             "start_ms": float(sms), "end_ms": float(ems), "confidence": float(c), "is_estimated": bool(est),
             "target_seq_idx": int(tidx), "index": i}
            for i, (p, sf, ef, tidx, est, c, sms, ems) in enumerate(phoneme_timestamps)]
        if group_timestamps is not None:
            out["group_ts"] = [
                {"group_id": int(g), "group_label": self.group_id_to_label.get(g, f"UNK_{g}"), "start_ms": float(sms),
                 "end_ms": float(ems), "confidence": float(c), "is_estimated": bool(est), "target_seq_idx": int(tidx),
                 "index": i}
                for i, (g, sf, ef, tidx, est, c, sms, ems) in enumerate(group_timestamps)]
        out["words_ts"] = self._align_words(out["phoneme_ts"], ts.get("word_num", []), ts.get("words", []))
        return out

    def post_process_segment(self, segment, ts, phoneme_sequence, phoneme_timestamps, group_timestamps=None,
                             debug=False):
        """core.py:1140 (the reference's argument order)."""
        return self._post_process_segment(segment, ts, phoneme_timestamps, group_timestamps, phoneme_sequence)

    def process_segments(self, srt_data, audio_wavs, extract_embeddings=False, do_groups=False, batch_size=16,
                         debug=False):
        """core.py:1212-1487, step for step: normalise the inputs (a dict -> a one-clip batch; a (C,T) / (B,C,T) tensor
        -> list of (C,T) clips), flatten the sub-segments of all clips, phonemise, drop sequences shorter than
        `ph_seq_min`, cut / down-mix / RMS-normalise / pad the audio of all segments (`_chop_segments`), run the extraction in one call -- or, when
        `batch_size < number of segments`, in slices whose `ValueError` ("Audio too short to align", core.py:1367-1386)
        turns that slice into empty results; in the one-call branch the error propagates like in the reference --
        post-process, regroup per clip, then the confidence analysis that can set `coverage_analysis.bad_alignment`.
        Always returns a list with one {'segments': [...]} per clip."""
        if extract_embeddings:
            raise NotImplementedError("extract_embeddings=True is outside the accelerated path")
        if isinstance(audio_wavs, torch.Tensor):
            if audio_wavs.dim() == 3:
                audio_wavs = [audio_wavs[i] for i in range(audio_wavs.size(0))]
            elif audio_wavs.dim() == 2:
                audio_wavs = [audio_wavs]
            else:
                raise ValueError(f"Expected audio_wavs of 2D (C,T) or 3D (B,C,T), got {audio_wavs.dim()}D")
        if isinstance(srt_data, dict):
            srt_data = [srt_data]
        if len(srt_data) != len(audio_wavs):
            raise ValueError(f"Batch size mismatch: {len(srt_data)} srt items vs {len(audio_wavs)} audio waveforms.")
        for bi, batch_item in enumerate(srt_data):
            if "segments" not in batch_item:
                raise ValueError(f"Batch item {bi} missing 'segments' key. Keys: {list(batch_item.keys())}")
            for si, seg in enumerate(batch_item["segments"]):
                if not all(k in seg for k in ("start", "end", "text")):
                    raise ValueError(f"Batch {bi}, segment {si} missing required keys (start/end/text). Has: {list(seg.keys())}")
        num_batch = len(srt_data)
        flat_items = [(bi, seg, clip_wav) for bi, (item, clip_wav) in enumerate(zip(srt_data, audio_wavs))
                      for seg in item["segments"]]
        if not flat_items:
            return [{"segments": []} for _ in range(num_batch)]

        ts_outs = [self.phonemize_sentence(seg["text"]) for _, seg, _ in flat_items]
        phoneme_sequences = [ts[self.phonemes_key] for ts in ts_outs]
        group_sequences = [ts[self.phoneme_groups_key] for ts in ts_outs] if do_groups else [None] * len(flat_items)
        for (_, seg, _), ph_seq, grp_seq in zip(flat_items, phoneme_sequences, group_sequences):
            seg[self.phonemes_key] = ph_seq          # (the reference writes these into the caller's dicts too)
            seg[self.phoneme_groups_key] = grp_seq
        valid = []
        for i, ((bi, seg, _), ph_seq) in enumerate(zip(flat_items, phoneme_sequences)):
            if not ph_seq or len(ph_seq) < self.ph_seq_min:
                if debug or self.warn_level >= 1:
                    print(f"Skipping clip {bi}, segment '{seg.get('text', '')[:30]}': insufficient phoneme sequence "
                          f"({len(ph_seq) if ph_seq else 0})")
                continue
            valid.append(i)
        batch_results = [{"segments": []} for _ in range(num_batch)]
        if not valid:
            return batch_results
        flat_f = [flat_items[i] for i in valid]
        ph_f = [phoneme_sequences[i] for i in valid]
        grp_f = [group_sequences[i] for i in valid]
        ts_f = [ts_outs[i] for i in valid]

        rate = self.resampler_sample_rate
        wavs, wav_lens, codes = self._chop_segments([clip_wav for _, _, clip_wav in flat_f],
                                                    [(int(seg["start"] * rate), int(seg["end"] * rate)) for _, seg, _ in flat_f])
        keep = [i for i, code in enumerate(codes) if code == 0]
        if len(keep) < len(flat_f):
            if debug or self.warn_level >= 1:
                for i, code in enumerate(codes):
                    if code != 0:
                        bi, seg, _ = flat_f[i]
                        print(f"Skipping clip {bi}, segment start: {seg['start']}, end: {seg['end']} due to chopping error ({code})")
            flat_f = [flat_f[i] for i in keep]
            ph_f = [ph_f[i] for i in keep]
            grp_f = [grp_f[i] for i in keep]
            ts_f = [ts_f[i] for i in keep]
            wavs = wavs[keep]
            wav_lens = [wav_lens[i] for i in keep]
        if not keep:
            raise ValueError("All segments have audio chopping errors. Cannot proceed with timestamp extraction.")
        start_times = [seg["start"] for _, seg, _ in flat_f]

        if batch_size < len(flat_items):  # (the reference compares with the UNfiltered count)
            results = []
            for i in range(0, len(flat_f), batch_size):
                sl = slice(i, i + batch_size)
                try:
                    part, _, _ = self.extract_timestamps_from_segment_batch(
                        wavs[sl], wav_lens[sl], ph_f[sl], start_offset_times=start_times[sl],
                        group_sequences=grp_f[sl] if do_groups else None, extract_embeddings=False,
                        do_groups=do_groups, debug=debug)
                    results.extend(part)
                except ValueError as ex:
                    if debug or self.warn_level >= 1:
                        print(f"ValueError processing batch {sl}: {ex}. This may be due to audio duration too short for "
                              f"the phoneme sequence.")
                    results.extend([{"phoneme_timestamps": [], "group_timestamps": []} for _ in range(batch_size)])
        else:
            results, _, _ = self.extract_timestamps_from_segment_batch(
                wavs, wav_lens, ph_f, start_offset_times=start_times, group_sequences=grp_f if do_groups else None,
                extract_embeddings=False, do_groups=do_groups, debug=debug)

        for (bi, seg, _), result, ts in zip(flat_f, results, ts_f):
            processed = self.post_process_segment(seg, ts, seg[self.phonemes_key], result["phoneme_timestamps"],
                                                  result["group_timestamps"] if do_groups else None, debug=debug)
            batch_results[bi]["segments"].append(processed)
        self._confidence_analysis(batch_results, debug)
        return batch_results

    def _confidence_analysis(self, batch_results, debug=False):
        """core.py:1420-1470: counters, and `coverage_analysis.bad_alignment` for long segments with too many
        low-confidence phonemes or a confident-start / lost-end pattern."""
        for bi, batch_item in enumerate(batch_results):
            for si, seg_out in enumerate(batch_item["segments"]):
                self.total_segments_processed += 1
                if not seg_out.get("phoneme_ts"):
                    self.total_segments_failed += 1
                    continue
                phoneme_ts = seg_out["phoneme_ts"]
                if [t["phoneme_id"] for t in phoneme_ts] == seg_out[self.phonemes_key]:
                    self.perfect_matches += 1
                if len(phoneme_ts) > 60:
                    confidences = [t["confidence"] for t in phoneme_ts]
                    low_ratio = sum(1 for c in confidences if c < 0.5) / len(confidences)
                    if low_ratio > self.bad_confidence_threshold:
                        seg_out["coverage_analysis"]["bad_alignment"] = True
                        self.total_segments_bad += 1
                    first_20 = sum(confidences[10:30]) / 20
                    last_20 = sum(confidences[-30:-10]) / 20
                    if first_20 > 0.1 and last_20 < 0.1:
                        if self.silence_anchors == 0:
                            raise Exception(f"Bad confidence pattern in clip {bi}, segment {si+1}: first 20 avg "
                                            f"{first_20:.3f} vs last 20 avg {last_20:.3f}. Consider setting `silence_anchors=3`.")
                        seg_out["coverage_analysis"]["bad_alignment"] = True
                        self.total_segments_bad += 1

    def process_sentence(self, text, audio_wav, extract_embeddings=False, do_groups=False, debug=False):
        """core.py:1553-1584: one sentence, one (C, T) clip (a 1-D waveform is taken as one channel)."""
        audio_wav = self._as_clip(audio_wav)
        duration = audio_wav.shape[1] / self.sample_rate
        srt_data = [{"segments": [{"start": 0.0, "end": duration, "text": text.strip()}]}]
        return self.process_segments(srt_data, [audio_wav], extract_embeddings=extract_embeddings, do_groups=do_groups,
                                     debug=debug)[0]

    def process_sentences_batch(self, texts, audio_wavs, extract_embeddings=False, do_groups=False, debug=False):
        """core.py:1586-1616: one sentence per clip."""
        assert len(texts) == len(audio_wavs), \
            f"Number of texts ({len(texts)}) must match number of audio waveforms ({len(audio_wavs)})"
        audio_wavs = [self._as_clip(w) for w in audio_wavs]
        srt_data = [{"segments": [{"start": 0.0, "end": w.shape[1] / self.sample_rate, "text": text.strip()}]}
                    for text, w in zip(texts, audio_wavs)]
        return self.process_segments(srt_data, audio_wavs, extract_embeddings=extract_embeddings, do_groups=do_groups,
                                     debug=debug)

    @staticmethod
    def _as_clip(w):
        w = w if isinstance(w, torch.Tensor) else torch.as_tensor(np.asarray(w))
        return w.unsqueeze(0) if w.dim() == 1 else w

    process_batch = process_sentences_batch  # the name BASELINE.json uses
