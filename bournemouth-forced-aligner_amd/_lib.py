"""ctypes binding of libbfa_hip.so (include/bfa.h).  Fails loudly when the library is missing:
there is deliberately no fallback implementation."""
import ctypes
import os
import subprocess

_HERE = os.path.dirname(os.path.abspath(__file__))
# BFA_HIP_LIBRARY selects another build of the same ABI (kernel experiments under tools/ubench)
SO_PATH = os.environ.get("BFA_HIP_LIBRARY") or os.path.join(_HERE, "libbfa_hip.so")

BFA_OK = 0
BFA_ERR_INVALID_ARGUMENT, BFA_ERR_NO_DEVICE, BFA_ERR_LAUNCH = -1, -2, -3
BFA_ERR_WORKSPACE_TOO_SMALL, BFA_ERR_UNSUPPORTED = -4, -5
ITEM_OK, ITEM_TOO_SHORT, ITEM_BAD_TOKEN, ITEM_TOO_LARGE, ITEM_SEG_OVERFLOW, ITEM_BAD_HINT = 0, 1, 2, 3, 4, 5
HINT_NO_SILENCE_TARGETS = 1 << 16
HINT_UNIFORM_LENGTHS = 1 << 17
OPT_PRECREATE_STREAMS = 4  # bfa_set_option: create the handle's auxiliary / head streams now instead of at first need
OPT_WIDE_ANY_MAX_BATCH = 3  # bfa_set_option: silence-anchored calls up to this many utterances take the wide classes as one launch
OPT_WINDOW_ROUTING = 2  # bfa_set_option: 0 fast window first always, 1 by the handle's history (default), 2 exact window first always
OPT_CALLS_IN_FLIGHT = 1  # bfa_set_option: the caller keeps several bfa_align_heads calls in flight (BatchesInFlight)
MIX_MIN_BATCH = 2   # bfa_types.hpp: calls of at least this many utterances with non-uniform lengths take the one-kernel mixed path (k_mix)
ONE_HINT_MAX_BATCH = 64  # forced_alignment.hint_and_path asks for the one-kernel path below this many utterances
ONE_MAX_BATCH = 1024  # bfa_types.hpp: calls of at most this many utterances in ONE fast-window class take the one-kernel path of small calls (k_one)
PATH_CLASS_KERNELS, PATH_MIXED, PATH_ONE_KERNEL = 0, 1, 2  # bfa_call_path
MODE_EMPTY, MODE_SEGMENTED, MODE_STANDARD, MODE_PROPORTIONAL = 0, 1, 2, 3

EXPORTS = ["bfa_version", "bfa_abi_version", "bfa_create", "bfa_destroy", "bfa_last_error",
           "bfa_params_default", "bfa_workspace_bytes", "bfa_align_batch", "bfa_confidences",
           "bfa_postprocess", "bfa_log_softmax", "bfa_profile_enable", "bfa_profile_collect",
           "bfa_prepare_emissions", "bfa_stitch_windows", "bfa_align_heads", "bfa_profile_collect_spans", "bfa_set_option",
           "bfa_pack_words", "bfa_pack_results", "bfa_index_records", "bfa_call_path", "bfa_profile_copy",
           "bfa_pack16_words", "bfa_pack_results16", "bfa_call_counters"]


class BfaParams(ctypes.Structure):
    _fields_ = [(n, ctypes.c_int32) for n in (
        "blank_id", "silence_id", "silence_anchors", "ignore_noise", "truly_forced", "boost_targets",
        "enforce_minimum", "simple", "max_blanks", "class_mask", "window_max_tokens", "window_max_frames",
        "has_min_log_prob")] + [("min_log_prob", ctypes.c_float)]


class BfaHead(ctypes.Structure):
    """include/bfa.h: bfa_head (one head of bfa_align_heads)"""
    _fields_ = [("logits", ctypes.c_void_p), ("strideB", ctypes.c_int64), ("strideT", ctypes.c_int64),
                ("C", ctypes.c_int32), ("Smax", ctypes.c_int32), ("tokens", ctypes.c_void_p), ("params", BfaParams),
                ("out_row_stats", ctypes.c_void_p), ("out_frame_phoneme", ctypes.c_void_p),
                ("out_frame_idx", ctypes.c_void_p), ("out_segs", ctypes.c_void_p), ("seg_cap", ctypes.c_int32),
                ("out_seg_count", ctypes.c_void_p), ("out_status", ctypes.c_void_p), ("out_mode", ctypes.c_void_p),
                ("workspace", ctypes.c_void_p), ("workspace_bytes", ctypes.c_size_t),
                ("postprocess", ctypes.c_int32), ("extend", ctypes.c_int32), ("boundary_softness", ctypes.c_int32),
                ("out_conf", ctypes.c_void_p), ("out_conf_status", ctypes.c_void_p)]


class BfaSegment(ctypes.Structure):
    _fields_ = [("phoneme", ctypes.c_int32), ("start", ctypes.c_int32), ("end", ctypes.c_int32),
                ("target_idx", ctypes.c_int32)]


def build(force=False):
    """Compile the gfx950 library in-tree (hipcc cross-compiles without a GPU)."""
    src_dir = os.path.join(_HERE, "csrc")
    args = ["make", "-j%d" % max(1, min(8, os.cpu_count() or 1)), "-C", src_dir]
    if force:
        args.append("-B")
    subprocess.check_call(args)
    return SO_PATH


_lib = None


def lib():
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(SO_PATH):
        raise RuntimeError(
            f"HIP extension {SO_PATH} is missing. Build it with `python __graft_entry__.py` (or "
            f"`make -C {os.path.join(_HERE, 'csrc')}`). There is no CPU fallback for the alignment path.")
    L = ctypes.CDLL(SO_PATH)
    vp, i32, i64, sz = ctypes.c_void_p, ctypes.c_int, ctypes.c_int64, ctypes.c_size_t
    L.bfa_version.restype = ctypes.c_char_p
    L.bfa_abi_version.restype = ctypes.c_int
    L.bfa_create.argtypes = [ctypes.POINTER(vp), i32]
    L.bfa_destroy.argtypes = [vp]
    L.bfa_last_error.argtypes = [vp]
    L.bfa_last_error.restype = ctypes.c_char_p
    L.bfa_params_default.argtypes = [ctypes.POINTER(BfaParams), i32, i32]
    L.bfa_params_default.restype = None
    L.bfa_workspace_bytes.argtypes = [i32, i32, i32, i32, ctypes.POINTER(BfaParams)]
    L.bfa_workspace_bytes.restype = sz
    L.bfa_align_batch.argtypes = [vp, vp, i64, i64, i32, i32, i32, vp, vp, vp, i32, ctypes.POINTER(BfaParams),
                                  vp, vp, vp, i32, vp, vp, vp, vp, sz, vp]
    L.bfa_prepare_emissions.argtypes = [vp, vp, i64, i64, i32, i32, i32, vp, vp, vp, i32, ctypes.POINTER(BfaParams),
                                        vp, i64, i64, vp, sz, vp]
    L.bfa_confidences.argtypes = [vp, vp, vp, i64, i64, i32, i32, i32, vp, vp, i32, vp, vp, vp, vp]
    L.bfa_postprocess.argtypes = [vp, vp, vp, i64, i64, i32, i32, i32, vp, vp, i32, vp, i32, i32, vp]
    L.bfa_align_heads.argtypes = [vp, ctypes.POINTER(BfaHead), i32, i32, i32, vp, vp, vp]
    L.bfa_log_softmax.argtypes = [vp, vp, i64, vp, i64, i64, i32, vp]
    L.bfa_stitch_windows.argtypes = [vp, vp, i32, i32, i32, i32, vp, i32, vp, i64, i64, vp]
    L.bfa_profile_enable.argtypes = [vp, i32]
    L.bfa_set_option.argtypes = [vp, i32, i32]
    L.bfa_profile_collect_spans.argtypes = [vp, vp, vp, vp, i32]
    L.bfa_profile_collect.argtypes = [vp, ctypes.POINTER(ctypes.c_float), i32]
    L.bfa_call_path.argtypes = [i32, i32, i32, i32, ctypes.POINTER(BfaParams), i32]
    L.bfa_profile_copy.argtypes = [vp, vp, vp, sz, vp]
    L.bfa_pack_words.argtypes = [i32, i64, i32]
    L.bfa_pack_words.restype = i64
    L.bfa_pack_results.argtypes = [vp, vp, i32, vp, vp, vp, i32, i32, i32, i32, vp, vp]
    L.bfa_pack16_words.argtypes = [i32, i64]
    L.bfa_pack16_words.restype = i64
    L.bfa_pack_results16.argtypes = [vp, vp, i32, vp, i32, i32, i32, vp, vp]
    L.bfa_call_counters.argtypes = [vp, vp, i32, i32, i32, i32, ctypes.POINTER(BfaParams), ctypes.POINTER(ctypes.c_int32), vp]
    L.bfa_index_records.argtypes = [vp, vp, i32, i64, i32, i32, vp, vp, vp, vp]
    _lib = L
    return L


_handles = {}


def handle(device_index, slot=0):
    """One bfa handle per GPU per process -- plus one per extra `slot` for callers that keep several calls in flight on
    different streams (a handle owns the auxiliary streams its class kernels fan out to; calls that share a handle
    queue behind each other there)."""
    key = (device_index, slot)
    if key not in _handles:
        h = ctypes.c_void_p()
        rc = lib().bfa_create(ctypes.byref(h), int(device_index))
        if rc != BFA_OK:
            raise RuntimeError(f"bfa_create(device={device_index}) failed with status {rc} "
                               f"(no GPU visible? this package has no CPU path)")
        _handles[key] = h
    return _handles[key]


def set_calls_in_flight(device_index, slot, on):
    """BFA_OPT_CALLS_IN_FLIGHT on the handle of (device, slot): the caller keeps several bfa_align_heads calls in flight
    (one handle / stream each), see include/bfa.h."""
    h = handle(device_index, slot)
    check(lib().bfa_set_option(h, OPT_CALLS_IN_FLIGHT, 1 if on else 0), h, "bfa_set_option")
    # (no precreate_streams here: three bfa_align_heads calls in flight measured SLOWER with the auxiliary streams in existence,
    # 1.53 against 1.43 ms per real-text step on one box -- the mapping of streams onto hardware queues is the runtime's)


def precreate_streams(device_index, slot, which=1):
    """BFA_OPT_PRECREATE_STREAMS: create the handle's auxiliary streams (1), head streams (2) or both (3) now.  A handle creates
    them when a call first needs them (a fresh process then aligns its first chunk 22 ms sooner); callers that keep several
    calls in flight on several handles create them up front: with them in existence the runtime spreads the callers' own
    streams over its hardware queues differently, and four headline batches in flight run 6 % faster (0.40 against 0.43 ms per
    step on one box, profiles/r06_precreate_ab.txt)."""
    h = handle(device_index, slot)
    check(lib().bfa_set_option(h, OPT_PRECREATE_STREAMS, int(which)), h, "bfa_set_option")


def set_window_routing(device_index, slot, value):
    """BFA_OPT_WINDOW_ROUTING on the handle of (device, slot): 0 never, 1 by history (default), 2 always exact-first."""
    h = handle(device_index, slot)
    check(lib().bfa_set_option(h, OPT_WINDOW_ROUTING, int(value)), h, "bfa_set_option")


def check(rc, h, what):
    if rc != BFA_OK:
        msg = lib().bfa_last_error(h)
        raise RuntimeError(f"{what} failed: status {rc}: {msg.decode() if msg else ''}")
