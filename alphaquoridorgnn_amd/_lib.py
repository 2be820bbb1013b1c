"""ctypes binding of libaqgnn_hip.so (C ABI in include/aqgnn.h).

The HIP library IS the product path.  There is no CPU fallback: if the shared object is missing or a call
fails, this module raises -- loudly -- instead of computing anything on the host.
"""
import ctypes
import os

import torch

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("AQG_LIB_PATH", os.path.join(_HERE, "libaqgnn_hip.so"))  # override: diagnostic builds only
MAX_LEGAL = 136
GNN_EXACT_F32 = 1        # AQG_GNN_EXACT_F32 (include/aqgnn.h)
GNN_RANGE_PROVEN = 2     # AQG_GNN_RANGE_PROVEN
GNN_PROVEN_MAX_WALLS = 16
ABI_VERSION = 10
TRAIN_PART_FLOATS = 2 * 128 * 128 + 128 * 6 + 3 * 128    # AQG_TRAIN_PART_FLOATS, per position of the batch

_c = ctypes
_vp, _i32, _f32 = _c.c_void_p, _c.c_int32, _c.c_float


class EngineStruct(_c.Structure):
    """Mirror of `struct aqg_engine` (include/aqgnn.h)."""
    _fields_ = (
        [(n, _i32) for n in ("board_size", "num_walls", "plies_for_draw", "num_games", "quota", "sims", "node_cap",
                             "max_plies", "prior_mode", "fake_bias", "gnn_flags")]
        + [("c_puct", _f32), ("temperature", _f32)]
        + [(n, _vp) for n in ("node_rec",
                              "node_count", "root_state", "path", "path_len",
                              "leaf_flag", "leaf_state",
                              "game_active", "slot_game", "game_plies", "game_result", "game_done", "game_slot", "game_first_move",
                              "legal_order", "legal_count", "pooled", "policy", "value",
                              "hist_state72", "hist_visits", "hist_action",
                              "counters", "stat_leaf_evals", "stat_terminal_sims", "packed_weights", "gnn_workspace",
                              "eval_cache_keys", "eval_cache_rows", "eval_cache_slot", "eval_mask", "stat_cache_hits", "eval_list", "eval_count")]
        + [("eval_cache_log2", _i32)]
    )


class TrainStruct(_c.Structure):
    """Mirror of `struct aqg_train` (include/aqgnn.h)."""
    _fields_ = (
        [(n, _i32) for n in ("board_size", "batch", "policy_size", "step")]
        + [(n, _f32) for n in ("lr", "beta1", "beta2", "eps")]
        + [(n, _vp * 14) for n in ("params", "grads", "adam_m", "adam_v")]
        + [(n, _vp) for n in ("h1", "h2", "h3", "zbuf", "dh", "g", "dg", "hp", "hv", "dhp", "dhv",
                              "lg", "pol", "vp", "val", "loss", "part")]
    )


SIGNATURES = {
    "aqg_abi_version": (_c.c_int, []),
    "aqg_last_error": (_c.c_char_p, []),
    "aqg_set_option": (_c.c_int, [_c.c_char_p, _c.c_int]),
    "aqg_debug_poison_lds": (_c.c_int, [_vp]),
    "aqg_debug_trace": (_c.c_int, [_vp, _c.c_uint]),
    "aqg_profile_collect": (_c.c_int, [_c.POINTER(_c.c_double), _c.POINTER(_c.c_longlong), _c.POINTER(_c.c_longlong), _c.c_int]),
    "aqg_legal_actions": (_c.c_int, [_c.c_int, _vp, _c.c_int, _vp, _vp, _vp, _vp]),
    "aqg_state_next": (_c.c_int, [_c.c_int, _vp, _vp, _c.c_int, _vp, _vp]),
    "aqg_state_status": (_c.c_int, [_c.c_int, _vp, _c.c_int, _c.c_int, _vp, _vp]),
    "aqg_gcn_packed_floats": (_c.c_size_t, [_c.c_int]),
    "aqg_gcn_pack_weights_host": (_c.c_int, [_c.c_int, _c.POINTER(_vp), _vp]),
    "aqg_gcn_forward_boards": (_c.c_int, [_c.c_int, _vp, _c.c_int, _c.c_int, _vp, _vp, _vp, _vp, _vp, _vp, _c.c_int, _vp]),
    "aqg_gcn_forward_boards_guarded": (_c.c_int, [_c.c_int, _vp, _c.c_int, _c.c_int, _vp, _vp, _vp, _vp, _vp, _vp, _c.c_int, _vp, _vp]),
    "aqg_gcn_boards_any_workspace_floats": (_c.c_size_t, [_c.c_int, _c.c_int]),
    "aqg_gcn_forward_boards_any": (_c.c_int, [_c.c_int, _vp, _c.c_int, _c.c_int, _vp, _vp, _c.c_size_t, _vp, _vp, _vp, _vp, _vp, _c.c_int, _vp]),
    "aqg_gcn_forward_graph": (_c.c_int, [_c.c_int, _c.c_int, _vp, _c.c_int, _vp, _vp, _vp, _vp, _c.c_int, _vp, _vp,
                                         _vp, _vp, _vp, _vp, _vp, _vp, _vp]),
    "aqg_engine_reset": (_c.c_int, [_c.POINTER(EngineStruct), _vp]),
    "aqg_engine_clear_eval_cache": (_c.c_int, [_c.POINTER(EngineStruct), _vp]),
    "aqg_engine_move": (_c.c_int, [_c.POINTER(EngineStruct), _vp, _vp]),
    "aqg_engine_search": (_c.c_int, [_c.POINTER(EngineStruct), _vp, _vp]),
    "aqg_engine_begin_move": (_c.c_int, [_c.POINTER(EngineStruct), _vp]),
    "aqg_engine_step": (_c.c_int, [_c.POINTER(EngineStruct), _c.c_int, _c.c_int, _vp]),
    "aqg_engine_finish_move": (_c.c_int, [_c.POINTER(EngineStruct), _vp, _vp]),
    "aqg_engine_set_roots": (_c.c_int, [_c.POINTER(EngineStruct), _vp, _vp]),
    "aqg_engine_root_visits": (_c.c_int, [_c.POINTER(EngineStruct), _vp, _vp, _vp, _vp]),
    "aqg_gcn_train_step": (_c.c_int, [_c.POINTER(TrainStruct), _vp, _vp, _vp, _c.c_int, _vp]),
    "aqg_gcn_train_steps": (_c.c_int, [_c.POINTER(TrainStruct), _vp, _vp, _vp, _vp, _c.c_longlong, _vp, _vp]),
    "aqg_gcn_train_fallbacks": (_c.c_longlong, [_c.c_int]),
    "aqg_host_legal_actions": (_c.c_int, [_c.c_int, _vp, _vp]),
    "aqg_host_next": (_c.c_int, [_c.c_int, _vp, _c.c_int, _vp]),
    "aqg_host_shortest_path": (_c.c_int, [_c.c_int, _vp]),
    "aqg_host_heuristic_eval": (_c.c_double, [_c.c_int, _vp, _c.c_int]),
    "aqg_host_alpha_beta_action": (_c.c_int, [_c.c_int, _vp, _c.c_int, _c.c_int, _c.c_int]),
}

_lib = None


class HipLibraryError(RuntimeError):
    pass


def load():
    """Load libaqgnn_hip.so; raises HipLibraryError if it is missing (build with __graft_entry__.build())."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise HipLibraryError(
            f"{LIB_PATH} not found: the HIP extension is required (no CPU fallback). "
            "Build it with `python -c 'import __graft_entry__ as g; g.build()'` or alphaquoridorgnn_amd/csrc/build.sh")
    lib = ctypes.CDLL(LIB_PATH)
    for name, (res, args) in SIGNATURES.items():
        fn = getattr(lib, name)          # AttributeError here == header/library mismatch
        fn.restype = res
        fn.argtypes = args
    if lib.aqg_abi_version() != ABI_VERSION:
        raise HipLibraryError(f"ABI mismatch: library {lib.aqg_abi_version()} vs binding {ABI_VERSION}")
    _lib = lib
    return lib


def check(rc, what):
    if rc != 0:
        msg = load().aqg_last_error().decode(errors="replace")
        raise HipLibraryError(f"{what} failed: {msg}")


def ptr(t):
    """Device pointer of a torch tensor (None -> NULL)."""
    if t is None:
        return None
    assert t.is_cuda and t.is_contiguous(), "HIP entry points take contiguous device tensors"
    return ctypes.c_void_p(t.data_ptr())


def stream_ptr(device=None):
    return ctypes.c_void_p(torch.cuda.current_stream(device).cuda_stream)


def require_gpu(device=None):
    if not torch.cuda.is_available():
        raise HipLibraryError("no MI355X visible (torch.cuda.is_available() is False): the hot path runs on the GPU only")
    return torch.device("cuda", torch.cuda.current_device()) if device is None else torch.device(device)


def set_option(name, value):
    check(load().aqg_set_option(name.encode(), int(value)), f"aqg_set_option({name})")


def profile_collect(reset=False):
    """(total_ms, launches, boards) of the profiled trunk launches so far (see aqg_profile_collect)."""
    ms, n, b = _c.c_double(0), _c.c_longlong(0), _c.c_longlong(0)
    check(load().aqg_profile_collect(_c.byref(ms), _c.byref(n), _c.byref(b), 1 if reset else 0), "aqg_profile_collect")
    return ms.value, n.value, b.value


def poison_lds(device=None):
    """Fill every CU's LDS with NaN patterns (tests: makes LDS read-before-write deterministic)."""
    check(load().aqg_debug_poison_lds(stream_ptr(device)), "aqg_debug_poison_lds")
