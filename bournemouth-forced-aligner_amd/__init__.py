"""MI355X-native forced-alignment core: drop-in for the Viterbi / confidence path of
tabahi/bournemouth-forced-aligner (bournemouth_aligner/forced_alignment.py, utils.py:70-149).

Python host code -> thin C-ABI (include/bfa.h, ctypes) -> hand-written gfx950 HIP kernels.
There is no CPU implementation in this package: without the HIP library or without a GPU the
alignment entry points raise.
"""
from .forced_alignment import AlignmentUtils, ViterbiDecoder  # noqa: F401
from .utils import (_calculate_confidences, convert_to_ms, calculate_confidences_batch, log_softmax,  # noqa: F401
                    postprocess_batch)
from .core import PhonemeTimestampAligner  # noqa: F401
from .inflight import BatchesInFlight  # noqa: F401
from .windowing import slice_windows, stich_window_predictions, stitch_total_frames  # noqa: F401
from .textgrid import dict_to_textgrid, dict_to_textgrid_with_confidence  # noqa: F401

__all__ = ["AlignmentUtils", "ViterbiDecoder", "_calculate_confidences", "convert_to_ms",
           "calculate_confidences_batch", "log_softmax", "postprocess_batch", "PhonemeTimestampAligner", "BatchesInFlight",
           "slice_windows", "stich_window_predictions", "stitch_total_frames", "dict_to_textgrid",
           "dict_to_textgrid_with_confidence"]
__version__ = "0.1.0"
