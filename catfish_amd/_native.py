"""ctypes binding of the C ABI declared in include/catfish_hip.h.

The product path has NO CPU fallback: if the HIP library has not been built
(``python -c "import __graft_entry__ as g; g.build()"`` or
``python -m catfish_amd.build``) importing a symbol from here raises.
"""
from __future__ import annotations

import ctypes as C
import os

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
DEFAULT_LIB_PATH = os.path.join(_HERE, "csrc", "libcatfish_hip.so")


def debug_knobs_on():
    """``CATFISH_DEBUG_KNOBS=1``: the ONE switch behind which the A/B knobs of tools/ and tests live -- the library's
    ``CATFISH_*`` kernel-selection variables (csrc/catfish_hip.hip ``cf_knob``) and ``CATFISH_HIP_LIB`` here.  Without it a
    stray environment variable changes nothing; with it every knob that takes effect is named on stderr."""
    return os.environ.get("CATFISH_DEBUG_KNOBS", "0") not in ("", "0")


def _lib_path():
    override = os.environ.get("CATFISH_HIP_LIB")
    if override and debug_knobs_on():
        import sys
        sys.stderr.write("catfish_amd: debug knob CATFISH_HIP_LIB=%s is active (NOT the in-tree library)\n" % override)
        return override
    return DEFAULT_LIB_PATH


LIB_PATH = DEFAULT_LIB_PATH
CF_ABI_VERSION = 6            # include/catfish_hip.h

CF_OK = 0
CF_ERR_INVALID = -1
CF_ERR_HIP = -2
CF_ERR_NOMEM = -3
CF_ERR_IO = -4
CF_WINDOW = 35
CF_PROF_SLOTS = 12
PRECISIONS = {"fp32": 0, "bf16x3": 1, "bf16": 2}

_f32p = C.POINTER(C.c_float)


class cf_hparams(C.Structure):
    _fields_ = [("layer_size", C.c_int32), ("n_layers", C.c_int32),
                ("layer_size_res", C.c_int32), ("n_layers_res", C.c_int32),
                ("window", C.c_int32), ("bn_epsilon", C.c_float),
                ("max_windows_per_pass", C.c_int64), ("n_streams", C.c_int32), ("precision", C.c_int32), ("fuse_layers", C.c_int32)]


class cf_conv_bn(C.Structure):
    _fields_ = [("kernel", _f32p), ("bias", _f32p), ("gamma", _f32p), ("beta", _f32p),
                ("moving_mean", _f32p), ("moving_variance", _f32p),
                ("ksize", C.c_int32), ("cin", C.c_int32)]


class cf_gru_dir(C.Structure):
    _fields_ = [("gates_kernel", _f32p), ("gates_bias", _f32p),
                ("candidate_kernel", _f32p), ("candidate_bias", _f32p),
                ("cin", C.c_int32)]


class cf_weights(C.Structure):
    _fields_ = [("conv", C.POINTER(cf_conv_bn)), ("gru", C.POINTER(cf_gru_dir)),
                ("dense_kernel", _f32p), ("dense_bias", _f32p)]


class NativeLibraryMissing(RuntimeError):
    pass


class CatfishHipError(RuntimeError):
    def __init__(self, code, msg):
        RuntimeError.__init__(self, "catfish_hip error %d: %s" % (code, msg))
        self.code = code


_lib = None

# name -> (restype, argtypes); every symbol include/catfish_hip.h declares
SYMBOLS = {
    "cf_model_create": (C.c_int, [C.POINTER(cf_weights), C.POINTER(cf_hparams), C.c_int, C.POINTER(C.c_void_p)]),
    "cf_model_destroy": (None, [C.c_void_p]),
    "cf_infer": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int64, C.c_void_p, C.c_void_p]),
    "cf_infer_host": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int64, C.c_void_p]),
    "cf_infer_logits": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int64, C.c_void_p, C.c_void_p, C.c_void_p]),
    "cf_infer_host_logits": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int64, C.c_void_p, C.c_void_p]),
    "cf_check_error": (C.c_int, [C.c_void_p]),
    "cf_clear_error": (C.c_int, [C.c_void_p]),
    "cf_launch_regimes": (C.c_int, [C.c_void_p, C.POINTER(C.c_int64)]),
    "cf_postprocess": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int64, C.c_int64,
                                 C.c_float, C.c_int32, C.c_void_p, C.c_void_p]),
    "cf_spans": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int64, C.c_int64, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]),
    "cf_normalize": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int64, C.c_void_p, C.c_void_p]),
    "cf_chunks_from_spans": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int64, C.c_int64,
                                       C.c_void_p, C.c_void_p, C.c_void_p, C.c_int64, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int64]),
    "cf_load_npy_int16": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int64, C.c_void_p, C.c_int64, C.c_void_p, C.c_void_p, C.c_int32]),
    "cf_stat_files": (C.c_int, [C.c_char_p, C.c_void_p, C.c_void_p, C.c_int64, C.c_void_p, C.c_int32]),
    "cf_postprocess_spans": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int64, C.c_int64, C.c_float, C.c_int32,
                                       C.c_void_p, C.c_int64, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]),
    "cf_crc32c": (C.c_uint32, [C.c_void_p, C.c_int64, C.c_uint32]),
    "cf_listing_open": (C.c_int, [C.c_char_p, C.POINTER(C.c_void_p), C.POINTER(C.c_int64), C.POINTER(C.c_uint64)]),
    "cf_listing_from_names": (C.c_int, [C.c_char_p, C.c_void_p, C.c_int64, C.c_int64, C.POINTER(C.c_void_p), C.POINTER(C.c_uint64)]),
    "cf_listing_sizes": (C.c_int, [C.c_void_p, C.c_int64, C.c_int64, C.c_void_p, C.c_int32]),
    "cf_listing_names": (C.c_int, [C.c_void_p, C.c_int64, C.c_int64, C.c_void_p, C.c_int64, C.c_void_p, C.POINTER(C.c_int64)]),
    "cf_listing_close": (None, [C.c_void_p]),
    "cf_listing_load_npy_int16": (C.c_int, [C.c_void_p, C.c_int64, C.c_int64, C.c_void_p, C.c_int64, C.c_void_p, C.c_void_p, C.c_int32]),
    "cf_listing_split_npy_int16": (C.c_int, [C.c_void_p, C.c_int64, C.c_int64, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p,
                                             C.c_void_p, C.c_char_p, C.c_char_p, C.c_int32, C.c_void_p]),
    "cf_chunks_json": (C.c_int64, [C.c_void_p, C.c_void_p, C.c_int64, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p,
                                   C.c_void_p, C.c_int64]),
    "cf_gru_pack_map": (C.c_int, [C.c_int32, C.c_int32, C.c_void_p, C.c_void_p, C.c_int64, C.POINTER(C.c_int64)]),
    "cf_gru_train_forward": (C.c_int, [C.c_void_p, C.c_int32, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int64,
                                       C.c_void_p]),
    "cf_gru_train_backward": (C.c_int, [C.c_void_p, C.c_int32, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p,
                                        C.c_void_p, C.c_void_p, C.c_void_p, C.c_int64, C.c_void_p]),
    "cf_gru_train_forward_dropout": (C.c_int, [C.c_void_p, C.c_int32, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int64,
                                               C.c_void_p, C.c_float, C.c_uint32, C.c_int32, C.c_void_p, C.c_void_p]),
    "cf_gru_train_backward_dropout": (C.c_int, [C.c_void_p, C.c_int32, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p,
                                                C.c_void_p, C.c_void_p, C.c_void_p, C.c_int64, C.c_float, C.c_uint32, C.c_int32,
                                                C.c_void_p, C.c_void_p]),
    "cf_gru_anysize_train_forward": (C.c_int, [C.c_void_p, C.c_int32, C.c_int32, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p,
                                               C.c_void_p, C.c_int64, C.c_void_p]),
    "cf_gru_anysize_train_backward": (C.c_int, [C.c_void_p, C.c_int32, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p,
                                                C.c_int64, C.c_void_p]),
    "cf_dropout_scale": (C.c_int, [C.c_void_p, C.c_float, C.c_uint32, C.c_int32, C.c_void_p, C.c_int64, C.c_void_p, C.c_void_p]),
    "cf_gru_wgrad_workspace_floats": (C.c_int64, [C.c_void_p, C.c_int32, C.c_int64]),
    "cf_gru_train_wgrad": (C.c_int, [C.c_void_p, C.c_int32, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int64,
                                     C.c_void_p, C.c_int64, C.c_void_p, C.c_void_p]),
    "cf_res_train_param_floats": (C.c_int64, [C.c_int32]),
    "cf_res_train_workspace_floats": (C.c_int64, [C.c_int32, C.c_int64]),
    "cf_res_train_forward": (C.c_int, [C.c_void_p, C.c_int32, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int64, C.c_void_p]),
    "cf_res_train_backward": (C.c_int, [C.c_void_p, C.c_int32, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int64,
                                        C.c_void_p, C.c_int64, C.c_void_p]),
    "cf_train_head_workspace_floats": (C.c_int64, [C.c_void_p, C.c_int64]),
    "cf_train_head": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int64, C.c_void_p, C.c_void_p,
                                C.c_void_p, C.c_int64, C.c_void_p, C.c_void_p, C.c_void_p]),
    "cf_opt_step": (C.c_int, [C.c_void_p, C.c_int32, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int64, C.c_float,
                              C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int64, C.c_void_p]),
    "cf_profile_enable": (C.c_int, [C.c_void_p, C.c_int]),
    "cf_profile_reset": (C.c_int, [C.c_void_p]),
    "cf_profile_read": (C.c_int, [C.c_void_p, C.POINTER(C.c_double), C.POINTER(C.c_int64)]),
    "cf_profile_slot_name": (C.c_char_p, [C.c_int]),
    "cf_debug_stage": (C.c_int, [C.c_void_p, C.c_int, C.c_int64, C.c_void_p]),
    "cf_device_identity": (C.c_int, [C.c_int, C.c_char_p, C.c_int64, C.c_char_p, C.c_int64]),
    "cf_workspace_bytes": (C.c_int64, [C.c_void_p]),
    "cf_last_error": (C.c_char_p, []),
    "cf_version": (C.c_char_p, []),
    "cf_abi_version": (C.c_int, []),
}


def _preload_torch_hip_runtime():
    """One process must hold ONE HIP/ROCr runtime.

    PyTorch-ROCm wheels bundle their own libamdhip64.so (soname libamdhip64.so.7).  If our
    library pulled in the system copy first, torch would later load its bundled copy next to
    it and the second ROCr instance finds no GPU ("No HIP GPUs are available").  Loading
    torch's copy first makes the dynamic linker satisfy our NEEDED libamdhip64.so.7 with
    that same object (soname match).  Without torch installed the system runtime is used.
    """
    try:
        import torch
    except ImportError:
        return
    cand = os.path.join(os.path.dirname(torch.__file__), "lib", "libamdhip64.so")
    if os.path.exists(cand):
        C.CDLL(cand, mode=C.RTLD_GLOBAL)


def lib():
    """Load (once) and return the C-ABI library; raise loudly when it is absent."""
    global _lib, LIB_PATH
    if _lib is None:
        path = _lib_path()
        if not os.path.exists(path):
            raise NativeLibraryMissing(
                "%s not found: the HIP extension has not been built. Run "
                "`python -m catfish_amd.build` (needs hipcc). There is no CPU fallback." % path)
        _preload_torch_hip_runtime()
        handle = C.CDLL(path)
        try:
            abi = getattr(handle, "cf_abi_version")
        except AttributeError:
            abi = None
        built = int(abi()) if abi is not None else None
        if built != CF_ABI_VERSION:
            raise NativeLibraryMissing("%s was built for ABI %s, this binding needs %d: rebuild it "
                                       "(`python -m catfish_amd.build --force`)" % (path, built, CF_ABI_VERSION))
        for name, (res, args) in SYMBOLS.items():
            fn = getattr(handle, name)
            fn.restype = res
            fn.argtypes = args
        _lib, LIB_PATH = handle, path
    return _lib


def check(rc):
    if rc != CF_OK:
        msg = lib().cf_last_error().decode("utf-8", "replace")
        if rc == CF_ERR_INVALID:
            raise ValueError("catfish_hip: " + msg)
        if rc == CF_ERR_IO:
            raise OSError("catfish_hip: " + msg)
        raise CatfishHipError(rc, msg)


def _as_f32(a):
    arr = np.ascontiguousarray(np.asarray(a, dtype=np.float32))
    return arr, arr.ctypes.data_as(_f32p)


def conv_name(j):
    return "conv1d" if j == 0 else "conv1d_%d" % j


def bn_name(j):
    return "batch_normalization" if j == 0 else "batch_normalization_%d" % j


def gru_prefix(layer, direction):
    return "stack_bidirectional_rnn/cell_%d/bidirectional_rnn/%s/gru_cell" % (layer, direction)


def build_weight_structs(weights, n_layers, n_layers_res):
    """dict {TF variable name: array} -> (cf_weights, keepalive list).

    Variable names are the ones TF auto-assigns while RNN.__init__ builds the
    graph (reference rnn_class.py:37-39, resnet_class.py:17-25).
    """
    keep = []

    def ptr(name):
        if name not in weights:
            raise ValueError("checkpoint is missing tensor %r" % name)
        arr, p = _as_f32(weights[name])
        keep.append(arr)
        return arr, p

    convs = (cf_conv_bn * max(1, 4 * n_layers_res))()
    for j in range(4 * n_layers_res):
        k_arr, k_ptr = ptr(conv_name(j) + "/kernel")
        if k_arr.ndim != 3:
            raise ValueError("%s/kernel must be [K, Cin, Cout]" % conv_name(j))
        c = convs[j]
        c.kernel = k_ptr
        c.bias = ptr(conv_name(j) + "/bias")[1]
        c.gamma = ptr(bn_name(j) + "/gamma")[1]
        c.beta = ptr(bn_name(j) + "/beta")[1]
        c.moving_mean = ptr(bn_name(j) + "/moving_mean")[1]
        c.moving_variance = ptr(bn_name(j) + "/moving_variance")[1]
        c.ksize = k_arr.shape[0]
        c.cin = k_arr.shape[1]
    grus = (cf_gru_dir * (2 * n_layers))()
    for layer in range(n_layers):
        for d, dname in enumerate(("fw", "bw")):
            p = gru_prefix(layer, dname)
            gk_arr, gk_ptr = ptr(p + "/gates/kernel")
            ck_arr, ck_ptr = ptr(p + "/candidate/kernel")
            g = grus[2 * layer + d]
            g.gates_kernel = gk_ptr
            g.gates_bias = ptr(p + "/gates/bias")[1]
            g.candidate_kernel = ck_ptr
            g.candidate_bias = ptr(p + "/candidate/bias")[1]
            h = ck_arr.shape[1]
            if gk_arr.shape != (ck_arr.shape[0], 2 * h):
                raise ValueError("%s: gates/candidate kernel shapes disagree" % p)
            g.cin = ck_arr.shape[0] - h
    w = cf_weights()
    w.conv = convs
    w.gru = grus
    w.dense_kernel = ptr("final_fully_connected/kernel")[1]
    w.dense_bias = ptr("final_fully_connected/bias")[1]
    keep.extend([convs, grus])
    return w, keep
