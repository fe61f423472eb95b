"""Several alignment batches in flight on one GPU.

One `AlignmentUtils.decode_alignments_device` call is planning + K1 (the banded forward pass, which keeps the machine
busy) followed by a tail of latency chains (rerun launch, backtrace, run-length encoding) that leave it mostly idle, and
K1's own ramp-down is serial as well.  A caller with a stream of batches -- the reference's `process_sentences_batch`
loop, a data-loader feeding posteriors -- gets ~15 % more throughput by letting consecutive batches overlap:
`BatchesInFlight` owns n decoders (each with its own HIP stream, library handle, workspace and output tensors) and hands
the batches to them in turn.  Results are identical to plain calls; what changes is only WHEN they are complete: a
result is complete on the stream of its slot (`result.wait()` makes the current stream wait for it, `synchronize()` waits
on the host).  Replaces nothing in the reference -- its batch loop is sequential (forced_alignment.py:885-905).
"""
import torch

from .forced_alignment import AlignmentUtils
from .utils import calculate_confidences_batch


class BatchesInFlight:
    def __init__(self, blank_id, silence_id, n=3, device=None, first_handle_slot=0, wait_for_caller=True,
                 **alignment_utils_kwargs):
        """wait_for_caller: a slot's stream first waits for the caller's current stream (needed when the caller's
        stream has just produced the inputs; False when they are known to be complete, e.g. resident benchmark data)."""
        assert n >= 1
        self.wait_for_caller = wait_for_caller
        self.device = torch.device("cuda", torch.cuda.current_device()) if device is None else torch.device(device)
        self.decoders = [AlignmentUtils(blank_id, silence_id, **alignment_utils_kwargs) for _ in range(n)]
        for k, d in enumerate(self.decoders):
            d.viterbi_decoder.handle_slot = first_handle_slot + k
        # one batch in flight: the caller's current stream, exactly the plain call
        self.streams = [torch.cuda.Stream(device=self.device) for _ in range(n)] if n > 1 else [None]
        if n > 1:  # several calls in flight: the handles' auxiliary streams up front (_lib.precreate_streams says why)
            from . import _lib
            idx = self.device.index if self.device.index is not None else torch.cuda.current_device()
            for d in self.decoders:
                _lib.precreate_streams(idx, d.viterbi_decoder.handle_slot)
        self._next = 0

    @property
    def handle_slots(self):
        return [d.viterbi_decoder.handle_slot for d in self.decoders]

    def _one(self, k, log_probs, true_seqs, pred_lens, true_seqs_lens, confidences, kwargs):
        res = self.decoders[k].decode_alignments_device(log_probs, true_seqs, pred_lens, true_seqs_lens, **kwargs)
        if confidences:  # utils._calculate_confidences of the aligned tuples (core.py:936-937), right behind the alignment
            res.conf, res.conf_status = calculate_confidences_batch(
                log_probs, res.segs, res.seg_count, T_rows=pred_lens,
                handle_slot=self.decoders[k].viterbi_decoder.handle_slot)
        return res

    def submit(self, log_probs, true_seqs, pred_lens, true_seqs_lens, confidences=False, **kwargs):
        """Enqueue one batch on the next slot (arguments of AlignmentUtils.decode_alignments_device).  The inputs must
        already be valid on the device when this is called from the caller's stream: the slot's stream first waits for
        the caller's current stream.  `confidences`: also enqueue the confidence pass of the aligned tuples on the slot's
        stream (`.conf` [B, seg_cap], `.conf_status` [B]).  Returns the AlignmentResult with `.stream` (None = the current
        stream) and `.wait()`."""
        k = self._next
        self._next = (k + 1) % len(self.decoders)
        st = self.streams[k]
        if st is None:
            res = self._one(k, log_probs, true_seqs, pred_lens, true_seqs_lens, confidences, kwargs)
            res.stream = None
            res.wait = lambda: None
            return res
        if self.wait_for_caller:
            st.wait_stream(torch.cuda.current_stream(self.device))
        with torch.cuda.stream(st):
            res = self._one(k, log_probs, true_seqs, pred_lens, true_seqs_lens, confidences, kwargs)
        res.stream = st
        res.wait = lambda s=st: torch.cuda.current_stream(self.device).wait_stream(s)
        return res

    def synchronize(self):
        for st in self.streams:
            if st is not None:
                st.synchronize()
