"""ctypes binding of libaz_amd.so (include/az_amd.h).  No fallback: a missing library is an error."""
import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
SO_PATH = os.path.join(_HERE, "libaz_amd.so")
_LIB = None

AZ_OK, AZ_EINVAL, AZ_EHIP, AZ_ESTATE, AZ_ECAPACITY, AZ_EILLEGAL = 0, -1, -2, -3, -4, -5
GAME_IDS = {"othello": 0, "connect4": 1, "tictactoe": 2}
TIE_LOWEST, TIE_RANDOM = 0, 1
NOISE_OFF, NOISE_PHILOX, NOISE_HASH = 0, 1, 2
EVAL_NET, EVAL_FAKE, EVAL_ROLLOUT = 0, 1, 2


class AzError(RuntimeError):
    pass


class EngineCfg(C.Structure):
    _fields_ = [("game", C.c_int32), ("H", C.c_int32), ("W", C.c_int32), ("n_slots", C.c_int32), ("n_sim", C.c_int32),
                ("dirichlet_alpha", C.c_double), ("dirichlet_epsilon", C.c_double),
                ("temp_max_step", C.c_int32), ("temp_min_step", C.c_int32),
                ("tie_mode", C.c_int32), ("noise_mode", C.c_int32), ("evaluator", C.c_int32), ("seed", C.c_uint32),
                ("node_capacity", C.c_int32), ("max_plies", C.c_int32), ("sample_capacity", C.c_int64)]


class EngineStats(C.Structure):
    _fields_ = [("games_done", C.c_int64), ("samples", C.c_int64), ("net_evals", C.c_int64),
                ("lockstep_iters", C.c_int64), ("plies", C.c_int64), ("max_nodes_used", C.c_int32),
                ("error_flags", C.c_int32), ("graph_replays", C.c_int64), ("max_path_len", C.c_int32), ("reserved", C.c_int32)]


# every symbol include/az_amd.h declares (tests check that the library exports all of them)
SYMBOLS = [
    "az_last_error", "az_version", "az_board_legal_batch", "az_board_play_batch", "az_board_status_batch",
    "az_net_create", "az_net_destroy", "az_net_set_tensor", "az_net_commit", "az_net_set_tensor_device", "az_net_commit_device", "az_net_forward", "az_net_forward_dyn",
    "az_net_action_size",
    "az_net_flops_per_board", "az_net_time_stage", "az_net_stage_kernel", "az_net_profile", "az_net_profiling", "az_net_profile_read", "az_net_profile_overhead", "az_engine_create", "az_engine_destroy", "az_engine_run",
    "az_engine_get_stats", "az_engine_samples", "az_engine_set_roots", "az_engine_search", "az_engine_search_begin", "az_engine_search_end", "az_engine_pair", "az_engine_advance",
    "az_engine_root_children", "az_engine_nodes_used", "az_engine_grow_pools", "az_engine_play", "az_augment_count", "az_augment",
    "az_engine_set_sides", "az_engine_best_moves", "az_engine_baseline_moves", "az_engine_root_status",
    "az_trainer_create", "az_trainer_destroy", "az_trainer_load", "az_trainer_store", "az_trainer_begin", "az_trainer_set_lr",
    "az_trainer_steps", "az_trainer_check", "az_trainer_debug",
]


def lib():
    global _LIB
    if _LIB is not None:
        return _LIB
    if not os.path.exists(SO_PATH):
        raise ImportError(f"{SO_PATH} is missing: build the HIP extension first "
                          f"(python -c 'import __graft_entry__ as g; g.build()'); there is no CPU fallback")
    L = C.CDLL(SO_PATH)
    vp, i32, i64 = C.c_void_p, C.c_int32, C.c_int64
    L.az_last_error.restype = C.c_char_p
    L.az_board_legal_batch.argtypes = [C.c_int, C.c_int, C.c_int, vp, vp, vp, i64, vp, vp]
    L.az_board_play_batch.argtypes = [C.c_int, C.c_int, C.c_int, vp, vp, vp, i64, vp, vp, vp, vp]
    L.az_board_status_batch.argtypes = [C.c_int, C.c_int, C.c_int, vp, vp, i64, vp, vp, vp, vp]
    L.az_net_create.argtypes = [C.c_int, C.c_int, C.c_int, C.c_int, C.POINTER(vp)]
    L.az_net_destroy.argtypes = [vp]
    L.az_net_destroy.restype = None
    L.az_net_set_tensor.argtypes = [vp, C.c_char_p, vp, i64]
    L.az_net_commit.argtypes = [vp, vp]
    L.az_net_set_tensor_device.argtypes = [vp, C.c_char_p, vp, i64, vp]
    L.az_net_commit_device.argtypes = [vp, vp]
    L.az_net_forward.argtypes = [vp, vp, C.c_int, vp, vp, vp]
    L.az_net_forward_dyn.argtypes = [vp, vp, vp, C.c_int, vp, vp, vp]
    L.az_net_action_size.argtypes = [vp]
    L.az_net_flops_per_board.argtypes = [vp]
    L.az_net_flops_per_board.restype = i64
    L.az_net_time_stage.argtypes = [vp, C.c_int, C.c_int, C.c_int, vp, C.POINTER(C.c_float)]
    L.az_net_stage_kernel.argtypes = [vp, C.c_int, C.c_int, C.c_char_p, C.c_int]
    L.az_net_profile.argtypes = [vp, C.c_int]
    L.az_net_profile_read.argtypes = [vp, C.POINTER(C.c_double), C.POINTER(C.c_int64)]
    L.az_net_profile_overhead.argtypes = [vp, C.POINTER(C.c_double)]
    L.az_engine_create.argtypes = [C.POINTER(EngineCfg), vp, vp, C.POINTER(vp)]
    L.az_engine_destroy.argtypes = [vp]
    L.az_engine_destroy.restype = None
    L.az_engine_run.argtypes = [vp, C.c_uint32, i32]
    L.az_engine_get_stats.argtypes = [vp, C.POINTER(EngineStats)]
    L.az_engine_samples.argtypes = [vp, C.POINTER(i64), C.POINTER(vp), C.POINTER(vp), C.POINTER(vp), C.POINTER(vp),
                                    C.POINTER(vp)]
    L.az_engine_set_roots.argtypes = [vp, vp, vp, vp, vp, i32]
    L.az_engine_search.argtypes = [vp, i32]
    L.az_engine_search_begin.argtypes = [vp, i32]
    L.az_engine_search_end.argtypes = [vp]
    L.az_engine_pair.argtypes = [vp, vp]
    L.az_engine_advance.argtypes = [vp]
    L.az_engine_play.argtypes = [vp, vp, i32, vp]
    L.az_engine_set_sides.argtypes = [vp, vp, i32]
    L.az_engine_best_moves.argtypes = [vp, vp]
    L.az_engine_baseline_moves.argtypes = [vp, i32, C.c_uint32, vp]
    L.az_engine_root_status.argtypes = [vp, vp, vp, vp, vp]
    L.az_augment_count.argtypes = [C.c_int, vp, i64, C.POINTER(i64), vp]
    L.az_augment.argtypes = [C.c_int, C.c_int, C.c_int, vp, vp, vp, vp, i64, vp, vp, vp, vp, i64, vp]
    L.az_engine_root_children.argtypes = [vp, i32, vp, vp, vp, vp, C.POINTER(i32), C.POINTER(i32)]
    L.az_engine_nodes_used.argtypes = [vp, i32, C.POINTER(i32)]
    L.az_engine_grow_pools.argtypes = [vp, i32]
    L.az_trainer_create.argtypes = [C.c_int, C.c_int, C.c_int, C.c_int, C.POINTER(vp)]
    L.az_trainer_destroy.argtypes = [vp]
    L.az_trainer_destroy.restype = None
    L.az_trainer_load.argtypes = [vp, C.c_char_p, vp, i64, vp]
    L.az_trainer_store.argtypes = [vp, C.c_char_p, vp, i64, vp]
    L.az_trainer_begin.argtypes = [vp, C.c_float, C.c_float, C.c_float, C.c_float, C.c_uint32, vp]
    L.az_trainer_set_lr.argtypes = [vp, C.c_float, vp]
    L.az_trainer_steps.argtypes = [vp, vp, vp, vp, i64, vp, i32, i32, vp, vp, vp]
    L.az_trainer_check.argtypes = [vp]
    L.az_trainer_debug.argtypes = [vp, C.c_char_p, C.POINTER(vp), C.POINTER(i64)]
    _LIB = L
    return L


# the sources a measurement depends on: counter files of the network kernels stay valid while only the engine or the training step
# changes, and the other way round
CSRC_COMPONENTS = {"net": ("az_net.hip",), "engine": ("az_engine.hip",), "train": ("az_train.hip",)}
_CSRC_SHARED = ("az_device.h", "az_host.h", "Makefile")  # az_common.hip (version number, error string) holds no kernel code


def csrc_tree_hash(component=None):
    """sha256[:16] over the sources libaz_amd.so is built from (component None: the whole csrc tree; "net" / "engine" / "train": that
    file plus the shared headers, the Makefile and include/az_amd.h): counter files under profiles/ carry the hash of the sources they
    were measured on, bench.py drops them when it differs from the running build"""
    import hashlib
    h = hashlib.sha256()
    d = os.path.join(_HERE, "csrc")
    for name in sorted(os.listdir(d)):
        if component is not None and name not in CSRC_COMPONENTS[component] + _CSRC_SHARED:
            continue
        if name.endswith((".hip", ".h")) or name == "Makefile":
            h.update(name.encode())
            h.update(open(os.path.join(d, name), "rb").read())
    h.update(open(os.path.join(os.path.dirname(_HERE), "include", "az_amd.h"), "rb").read())
    return h.hexdigest()[:16]


def check(rc):
    """0 -> ok; otherwise raise like the reference does (ValueError for bad arguments / illegal moves)."""
    if rc == AZ_OK:
        return
    msg = lib().az_last_error().decode(errors="replace")
    if rc in (AZ_EINVAL, AZ_EILLEGAL):
        raise ValueError(msg)
    raise AzError(f"[{rc}] {msg}")
