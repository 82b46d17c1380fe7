"""ctypes binding of libcaro_hip.so (declared in include/caro_hip.h).

There is no CPU fallback: if the shared library is missing or cannot be loaded
this module raises, and every engine entry point fails loudly when the process
has no GPU (CARO_E_NODEV).  The host-only helpers (rules on single states,
noise spec) work without a GPU; they are compiled from the same caro_rules.h as
the kernels.
"""
import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "libcaro_hip.so")
if os.environ.get("CARO_HIP_LIB"):  # kernel experiments: another build of the same library
    LIB_PATH = os.environ["CARO_HIP_LIB"]

GAME_CONNECT4, GAME_MNK = 0, 1


class CaroError(RuntimeError):
    pass


class CaroConfig(C.Structure):
    _fields_ = [
        ("game_kind", C.c_int32), ("n", C.c_int32), ("k", C.c_int32), ("n_games", C.c_int32),
        ("n_stores", C.c_int32), ("n_nets", C.c_int32), ("max_batch", C.c_int32), ("node_cap", C.c_int32),
        ("steps_before_tau_0", C.c_int32), ("first_player_mode", C.c_int32),
        ("c_puct", C.c_float), ("alpha", C.c_double), ("explore", C.c_double),
        ("seed", C.c_uint64), ("uid_base", C.c_uint64), ("uid_stride", C.c_uint64),
        ("device_id", C.c_int32), ("evict", C.c_int32), ("stagger", C.c_int32), ("stagger_recycle", C.c_int32),
        ("games_limit", C.c_int64),
    ]


_P = C.c_void_p
_SIGNATURES = {
    # name: (restype, argtypes)
    "caro_last_error": (C.c_char_p, []),
    "caro_version": (C.c_int, []),
    "caro_key_words": (C.c_int, [C.c_int, C.c_int]),
    "caro_action_space": (C.c_int, [C.c_int, C.c_int]),
    "caro_obs_cells": (C.c_int, [C.c_int, C.c_int]),
    "caro_host_initial": (C.c_int, [C.c_int, C.c_int, C.c_int, _P]),
    "caro_host_move": (C.c_int, [C.c_int, C.c_int, C.c_int, _P, C.c_int, C.c_int, _P]),
    "caro_host_legal": (C.c_int, [C.c_int, C.c_int, C.c_int, _P, _P]),
    "caro_host_encode": (C.c_int, [C.c_int, C.c_int, C.c_int, _P, C.c_int, _P]),
    "caro_host_noise_row": (C.c_int, [C.c_uint64, C.c_uint64, C.c_uint32, C.c_uint32, C.c_int, C.c_double, _P]),
    "caro_host_move_uniform": (C.c_double, [C.c_uint64, C.c_uint64, C.c_uint32]),
    "caro_rules_move_batch": (C.c_int, [C.c_int, C.c_int, C.c_int, C.c_int64, _P, _P, _P, _P, _P, _P]),
    "caro_rules_legal_batch": (C.c_int, [C.c_int, C.c_int, C.c_int, C.c_int64, _P, _P, _P]),
    "caro_rules_encode_batch": (C.c_int, [C.c_int, C.c_int, C.c_int, C.c_int64, _P, _P, _P, _P]),
    "caro_noise_batch": (C.c_int, [C.c_uint64, C.c_int64, C.c_int, C.c_double, _P, _P, _P, _P, _P]),
    "caro_engine_create": (C.c_int, [C.POINTER(CaroConfig), C.POINTER(_P)]),
    "caro_engine_destroy": (None, [_P]),
    "caro_engine_restart": (C.c_int, [_P, C.POINTER(CaroConfig), _P]),
    "caro_reset_games": (C.c_int, [_P, _P, _P]),
    "caro_set_roots": (C.c_int, [_P, _P, _P, _P]),
    "caro_select": (C.c_int, [_P, C.c_int, C.c_int, _P, _P, _P, _P]),
    "caro_leaf_counts": (C.c_int, [_P, _P, _P]),
    "caro_leaf_counts_dev": (C.c_int, [_P, _P]),
    "caro_net_packed_size": (C.c_int64, [C.c_int, C.c_int, C.c_int]),
    "caro_net_create": (C.c_int, [C.c_int, C.c_int, C.c_int, C.c_float, _P, C.c_int64, C.c_int, _P]),
    "caro_net_destroy": (None, [_P]),
    "caro_net_enable_winograd": (C.c_int, [_P, _P, C.c_int64]),
    "caro_net_winograd2d_size": (C.c_int, []),
    "caro_net_winograd2d_supported": (C.c_int, [C.c_int, C.c_int]),
    "caro_net_enable_winograd2d": (C.c_int, [_P, _P, C.c_int64]),
    "caro_net_split_bf16_size": (C.c_int64, []),
    "caro_net_enable_split_bf16": (C.c_int, [_P, _P, C.c_int64]),
    "caro_net_boards_per_workgroup": (C.c_int, [_P]),
    "caro_net_stream_evictions": (C.c_int64, [_P]),
    "caro_net_forward": (C.c_int, [_P, _P, _P, C.c_int, C.c_int64, _P, _P, _P]),
    "caro_stream_create_partition": (C.c_int, [C.c_int, C.c_int, C.c_int, _P]),
    "caro_stream_destroy": (C.c_int, [_P]),
    "caro_search_batch": (C.c_int, [_P, _P, _P, C.c_int, C.c_int, _P, _P, _P, _P, _P, _P]),
    "caro_search_move": (C.c_int, [_P, _P, _P, C.c_int, C.c_int] + [_P] * 10),
    "caro_net_forward_pair": (C.c_int, [_P, _P, _P, _P, C.c_int64, _P, _P, _P]),
    "caro_net_forward_pair_at": (C.c_int, [_P, _P, _P, _P, C.c_int64, C.c_int64, _P, _P, _P]),
    "caro_net_forward_slots": (C.c_int, [_P, _P, _P, _P, _P, C.c_int, C.c_int, _P, _P, _P]),
    "caro_net_forward_slot_list": (C.c_int, [_P, _P, _P, _P, _P, _P, C.c_int, C.c_int, _P, _P, _P]),
    "caro_net_create_hash": (C.c_int, [C.c_int, C.c_int, C.c_int, C.c_uint64, C.c_int, _P]),
    "caro_net_forward_stamped": (C.c_int, [_P, _P, _P, C.c_int, C.c_int64, _P, _P, _P, _P]),
    "caro_net_debug_stamps": (C.c_int, [_P, _P]),
    "caro_get_descent": (C.c_int, [_P, C.c_int, C.c_int, _P, _P, _P, _P, _P, _P]),
    "caro_select_cancel": (C.c_int, [_P]),
    "caro_expand_backup": (C.c_int, [_P, _P, _P, _P]),
    "caro_policy": (C.c_int, [_P, _P, _P, _P]),
    "caro_step": (C.c_int, [_P, _P, _P, _P, _P, _P]),
    "caro_drain_tuples": (C.c_int, [_P, C.c_int64, _P, _P, _P, _P, _P, C.c_int, _P, _P, _P]),
    "caro_drain_tuples_begin": (C.c_int, [_P, C.c_int64, _P, _P, _P, _P, _P, C.c_int, _P]),
    "caro_drain_tuples_end": (C.c_int, [_P, _P, _P]),
    "caro_search_staggered": (C.c_int, [_P, _P, _P, C.c_int, C.c_int, _P, _P, _P, _P, _P]),
    "caro_drain_parked_begin": (C.c_int, [_P, C.c_int64, _P, _P, _P, _P, _P, _P]),
    "caro_counters": (C.c_int, [_P, _P, _P]),
    "caro_live_games": (C.c_int, [_P, _P, _P]),
    "caro_pending_leaves": (C.c_int, [_P, _P, _P]),
    "caro_debug_stamps": (C.c_int, [_P, C.c_int]),
    "caro_debug_read": (C.c_int, [_P, _P, C.c_int64, _P]),
    "caro_debug_sqrt_check": (C.c_int, [C.c_uint32, _P]),
    "caro_profile_enable": (C.c_int, [_P, C.c_int]),
    "caro_profile_read": (C.c_int, [_P, _P, _P, C.c_int]),
    "caro_profile_begin": (C.c_int, [_P, C.c_int, _P]),
    "caro_profile_end": (None, [_P, C.c_int, _P]),
    "caro_tree_sizes": (C.c_int, [_P, _P, _P]),
    "caro_tree_live": (C.c_int, [_P, _P, _P]),
    "caro_lookup_nodes": (C.c_int, [_P, C.c_int64] + [_P] * 10),
    "caro_get_roots": (C.c_int, [_P] * 6),
    "caro_poke_nodes": (C.c_int, [_P, C.c_int64] + [_P] * 9),
    "caro_backup_path": (C.c_int, [_P, C.c_int, C.c_int, C.c_float, C.c_int, C.c_int, _P, _P, _P]),
    "caro_dump_tree": (C.c_int, [_P, C.c_int, C.c_int, C.c_int64] + [_P] * 8),
}
EXPORTS = sorted(_SIGNATURES)

_lib = None


def load():
    """Load libcaro_hip.so; raises CaroError if it is not there (no fallback)."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise CaroError("%s is missing: build it with `python -m caro_ai_amd.build` "
                        "(hipcc --offload-arch=gfx950); there is no CPU fallback" % LIB_PATH)
    try:
        lib = C.CDLL(LIB_PATH)
    except OSError as e:  # pragma: no cover
        raise CaroError("cannot load %s: %s" % (LIB_PATH, e))
    for name, (res, args) in _SIGNATURES.items():
        fn = getattr(lib, name)  # AttributeError = the header and the library disagree
        fn.restype = res
        fn.argtypes = args
    _lib = lib
    return lib


def check(rc):
    if rc != 0:
        msg = load().caro_last_error()
        raise CaroError("libcaro_hip error %d: %s" % (rc, msg.decode() if msg else "?"))
