"""tests/refload.py -- loads the REFERENCE's hot-path modules by file path (build container only).

`import bournemouth_aligner` fails here (torchaudio / phonemizer absent), but forced_alignment.py
and utils.py import only torch, so they load through importlib.  Used by tests/golden/make_golden.py
(fixture generation) and by the live oracle-vs-reference fuzz tests, which skip when
/root/reference is absent (it never exists on the GPU box).
"""
import importlib.util
import os
import sys
import types

REF_ROOT = "/root/reference"
REF_PKG = os.path.join(REF_ROOT, "bournemouth_aligner")


def available():
    return os.path.isfile(os.path.join(REF_PKG, "forced_alignment.py"))


def _load(name, path):
    spec = importlib.util.spec_from_file_location(name, path)
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod


_cache = {}


def forced_alignment():
    if "fa" not in _cache:
        _cache["fa"] = _load("_ref_forced_alignment", os.path.join(REF_PKG, "forced_alignment.py"))
    return _cache["fa"]


def utils():
    if "ut" not in _cache:
        _cache["ut"] = _load("_ref_utils", os.path.join(REF_PKG, "utils.py"))
    return _cache["ut"]


def core_aligner(**kwargs):
    """Level-2 harness: construct the reference PhonemeTimestampAligner(preset=None) with stub
    torchaudio / phonemizer modules so that its post-DP stages (ensure_target_coverage,
    extend_soft_boundaries_func, extract_timestamps_from_segment_batch) can be driven with
    synthetic logits.  No model is loaded (core.py:159)."""
    if "core" not in _cache:
        for name in ("torchaudio", "torchaudio.transforms", "phonemizer", "phonemizer.backend",
                     "phonemizer.separator", "phonemizer.backend.espeak", "phonemizer.backend.espeak.wrapper",
                     "librosa", "huggingface_hub_stub"):
            if name not in sys.modules:
                sys.modules[name] = types.ModuleType(name)
        ta = sys.modules["torchaudio"]
        ta.transforms = sys.modules["torchaudio.transforms"]

        class _Resample:
            def __init__(self, *a, **k):
                pass

            def __call__(self, x):
                return x
        ta.transforms.Resample = _Resample
        ta.load = lambda *a, **k: (_ for _ in ()).throw(RuntimeError("stub"))

        class _Backend:
            def __init__(self, *a, **k):
                pass

            def phonemize(self, texts, **k):
                return ["" for _ in texts]
        sys.modules["phonemizer.backend"].EspeakBackend = _Backend
        sys.modules["phonemizer.backend"].BACKENDS = {}

        class _Sep:
            def __init__(self, *a, **k):
                pass
        sys.modules["phonemizer.separator"].Separator = _Sep
        sys.modules["phonemizer"].backend = sys.modules["phonemizer.backend"]
        sys.modules["phonemizer"].separator = sys.modules["phonemizer.separator"]
        if REF_ROOT not in sys.path:
            sys.path.insert(0, REF_ROOT)
        import bournemouth_aligner.core as core  # noqa
        _cache["core"] = core
    core = _cache["core"]
    args = dict(preset=None, device="cpu")
    args.update(kwargs)
    return core.PhonemeTimestampAligner(**args)
