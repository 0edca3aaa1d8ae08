"""ctypes binding of libc4engine.so -- the ONLY way host Python reaches the HIP engine.

Mirrors include/c4_engine.h one to one.  There is no CPU fallback: if the shared library has not
been built (python -c "import __graft_entry__ as g; g.build()") importing the symbols fails loudly,
and every compute entry point fails with C4_EDEVICE when no gfx950 device is visible.
"""
import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
# C4_ENGINE_LIB: tuning aid -- load another build of the same library (A/B runs of kernel variants in one process tree)
LIB_PATH = os.environ.get("C4_ENGINE_LIB") or os.path.join(_HERE, "libc4engine.so")
HEADER_PATH = os.path.join(os.path.dirname(_HERE), "include", "c4_engine.h")

ABI_VERSION = 4
OK, EINVAL, EDEVICE, ENOMEM, ESTATE, ECAPACITY = 0, -1, -2, -3, -4, -5
RESULT_NONE, RESULT_XWIN, RESULT_DRAW, RESULT_OWIN = -1, 0, 1, 2
EVAL_EXTERNAL_F32, EVAL_EXTERNAL_F64, EVAL_CENTRE = 0, 1, 2
RNG_PHILOX, RNG_TAPE = 0, 1
PLANES_F32, PLANES_F16, PLANES_BF16 = 0, 1, 2
SLOT_ACTIVE, SLOT_PARKED, SLOT_MOVE_DONE = 0, 1, 2


class EngineError(RuntimeError):
    def __init__(self, code, message):
        super().__init__("c4 engine error %d: %s" % (code, message))
        self.code = code


class Config(C.Structure):
    _fields_ = [("abi_version", C.c_int32), ("n_slots", C.c_int32), ("simulations", C.c_int32),
                ("pb_c_base", C.c_int32), ("pb_c_init", C.c_double),
                ("root_dirichlet_alpha", C.c_double), ("root_exploration_fraction", C.c_double),
                ("num_sampling_moves", C.c_int32), ("eval_mode", C.c_int32), ("rng_mode", C.c_int32),
                ("seed", C.c_uint64), ("stop_after_move", C.c_int32), ("games_target", C.c_int64),
                ("record_capacity_games", C.c_int32), ("max_inner_iters", C.c_int32),
                ("planes_dtype", C.c_int32), ("eval_cache_log2_entries", C.c_int32), ("level_budget", C.c_int32), ("time_budget_cycles", C.c_int32), ("reserved", C.c_int32 * 4)]


class Stats(C.Structure):
    _fields_ = [(n, C.c_int64) for n in (
        "simulations", "expansions", "children_created", "terminal_sims", "leaf_evals", "depth_sum",
        "moves", "games_started", "games_finished", "launches", "active_slots", "capped_slots",
        "eval_cache_hits", "eval_cache_probes", "bad_evals", "dropped_games", "speculative_evals")]

    def as_dict(self):
        return {n: int(getattr(self, n)) for n, _ in self._fields_}


class RootResult(C.Structure):
    _fields_ = [("state", C.c_int32), ("move", C.c_int32), ("value", C.c_double),
                ("root_visits", C.c_uint32), ("root_value_sum", C.c_double),
                ("child_visits", C.c_uint32 * 7), ("child_value_sum", C.c_double * 7),
                ("child_status", C.c_int32 * 7), ("root_prior", C.c_double * 7),
                ("values_policy", C.c_double * 7), ("color0", C.c_uint64), ("color1", C.c_uint64),
                ("expansions", C.c_int64), ("simulations", C.c_int64)]


class GameRecord(C.Structure):
    _fields_ = [("game_id", C.c_int64), ("length", C.c_int32), ("result", C.c_int32),
                ("color0", C.c_uint64 * 42), ("color1", C.c_uint64 * 42), ("move", C.c_int32 * 42),
                ("value", C.c_double * 42), ("policy", (C.c_double * 7) * 42)]


class ExportBuffers(C.Structure):
    """c4_export_buffers: device pointers (torch.Tensor.data_ptr()) or None."""
    _fields_ = [(n, C.c_void_p) for n in ("boards_dev", "moves_dev", "values_dev", "policy_dev", "targets_dev",
                                          "game_index_dev", "lengths_dev", "results_dev", "ids_dev")]


class NetDesc(C.Structure):
    _fields_ = [("channels", C.c_int32), ("filters", C.c_int32), ("n_residuals", C.c_int32), ("precision", C.c_int32)] + \
        [(n, C.POINTER(C.c_float)) for n in ("stem_w", "stem_b", "conv_w", "conv_b", "head_w", "head_b",
                                             "vfc_w", "vfc_b", "vout_w", "pfc_w", "pfc_b")] + \
        [("vout_b", C.c_float), ("w1", C.c_float), ("w2", C.c_float), ("reserved2", C.c_float)]


_P = C.POINTER
_u64p, _i32p, _f32p, _f64p, _i64p = _P(C.c_uint64), _P(C.c_int32), _P(C.c_float), _P(C.c_double), _P(C.c_int64)

# name -> (restype, argtypes); every symbol include/c4_engine.h declares
SIGNATURES = {
    "c4_abi_version": (C.c_int, []),
    "c4_engine_create": (C.c_int, [_P(Config), C.c_int, _P(C.c_void_p)]),
    "c4_engine_destroy": (C.c_int, [C.c_void_p]),
    "c4_last_error": (C.c_char_p, [C.c_void_p]),
    "c4_set_stream": (C.c_int, [C.c_void_p, C.c_void_p]),
    "c4_clear_eval_cache": (C.c_int, [C.c_void_p]),
    "c4_reset": (C.c_int, [C.c_void_p, _u64p, _u64p, C.c_int32]),
    "c4_set_tapes": (C.c_int, [C.c_void_p, _f64p, _f64p, C.c_int32]),
    "c4_step": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]),
    "c4_step_range": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int32, C.c_int32, C.c_void_p]),
    "c4_run_centre": (C.c_int, [C.c_void_p, C.c_int32]),
    "c4_leaf_buffers": (C.c_int, [C.c_void_p, _P(C.c_void_p), _P(C.c_void_p), _P(C.c_void_p)]),
    "c4_read_leaves": (C.c_int, [C.c_void_p, _u64p, _u64p, _i32p]),
    "c4_get_stats": (C.c_int, [C.c_void_p, _P(Stats)]),
    "c4_read_roots": (C.c_int, [C.c_void_p, _P(RootResult)]),
    "c4_drain_games": (C.c_int, [C.c_void_p, _P(GameRecord), C.c_int32, _i32p]),
    "c4_finished_games": (C.c_int, [C.c_void_p, _i64p, _i64p]),
    "c4_export_games_dev": (C.c_int, [C.c_void_p, _P(ExportBuffers), C.c_int32, C.c_int64, C.c_void_p, C.c_void_p]),
    "c4_training_tensors_dev": (C.c_int, [C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int64, C.c_int32,
                                          C.c_void_p, C.c_void_p, C.c_void_p]),
    "c4_eval_cache_lookup": (C.c_int, [C.c_void_p, _u64p, _u64p, C.c_int32, _f32p, _f32p, _i32p]),
    "c4_debug_root_noise": (C.c_int, [C.c_int, C.c_uint64, C.c_double, _i64p, _i32p, _i32p, C.c_int32, _f64p, _f64p]),
    "c4_debug_sample_move": (C.c_int, [C.c_int, C.c_uint64, _i64p, _i32p, _f64p, _i32p, _f64p, C.c_int32, _f64p, _i32p]),
    "c4_debug_div_mismatches": (C.c_int, [C.c_int, C.c_int32, C.c_int32, C.c_int64, _i64p]),
    "c4_board_make_move": (C.c_int, [C.c_int, _u64p, _u64p, _i32p, C.c_int32, _u64p, _u64p, _i32p]),
    "c4_board_wins": (C.c_int, [C.c_int, _u64p, C.c_int32, _i32p]),
    "c4_board_valid_mask": (C.c_int, [C.c_int, _u64p, _u64p, C.c_int32, _i32p]),
    "c4_board_planes": (C.c_int, [C.c_int, _u64p, _u64p, C.c_int32, _f32p]),
    "c4_board_fliplr": (C.c_int, [C.c_int, _u64p, _u64p, C.c_int32, _u64p, _u64p]),
    "c4_board_centre_value": (C.c_int, [C.c_int, _u64p, _u64p, C.c_int32, _f64p]),
    "c4_debug_stamps": (C.c_int, [C.c_void_p, _P(C.c_uint64)]),
    "c4_debug_fused_net_stamps": (C.c_int, [C.c_void_p, _P(C.c_uint64)]),
    "c4_debug_latency_stamps": (C.c_int, [C.c_void_p, _P(C.c_uint64)]),
    "c4_net_create": (C.c_int, [C.c_int, _P(NetDesc), _P(C.c_void_p)]),
    "c4_net_destroy": (C.c_int, [C.c_void_p]),
    "c4_net_forward": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int32, C.c_void_p, C.c_void_p]),
    "c4_net_forward_wave": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int32, C.c_void_p, C.c_void_p]),
    "c4_net_last_error": (C.c_char_p, []),
    "c4_selfplay_steps": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int32, C.c_void_p]),
    "c4_net_debug_stamps": (C.c_int, [C.c_void_p, _P(C.c_uint64)]),
    "c4_bn_workspace_floats": (C.c_longlong, [C.c_int, C.c_int]),
    "c4_bn_train_forward": (C.c_int, [C.c_void_p] * 11 + [C.c_int] * 4 + [C.c_float] * 3 + [C.c_void_p]),
    "c4_bn_train_backward": (C.c_int, [C.c_void_p] * 11 + [C.c_int] * 4 + [C.c_float, C.c_void_p]),
    "c4_conv3x3_wrw_workspace_floats": (C.c_longlong, []),
    "c4_conv3x3_wrw": (C.c_int, [C.c_void_p] * 4 + [C.c_int] * 4 + [C.c_void_p]),
}

_lib = None


def _share_hip_runtime_with_torch():
    """PyTorch-ROCm wheels bundle their own libamdhip64.so (SONAME libamdhip64.so.7).  Two HIP
    runtimes in one process fight over the device ("No HIP GPUs are available"), so make sure the
    one torch will use is the one already mapped when libc4engine.so resolves libamdhip64.so.7.
    Only the runtime library is preloaded here -- torch itself is not imported."""
    import importlib.util
    import sys
    if "torch" in sys.modules:
        return  # torch (and its runtime) is already loaded; the SONAME match does the rest
    try:
        spec = importlib.util.find_spec("torch")
    except (ImportError, ValueError):
        spec = None
    if spec is None or not spec.origin:
        return
    cand = os.path.join(os.path.dirname(spec.origin), "lib", "libamdhip64.so")
    if os.path.exists(cand):
        try:
            C.CDLL(cand, mode=C.RTLD_GLOBAL)
        except OSError:
            pass


def load():
    """Load libc4engine.so and bind every symbol.  Raises if the HIP extension is missing."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise ImportError(
            "HIP engine %s is not built. Run `python -c \"import __graft_entry__ as g; g.build()\"` "
            "(hipcc --offload-arch=gfx950). There is no CPU fallback." % LIB_PATH)
    _share_hip_runtime_with_torch()
    L = C.CDLL(LIB_PATH)
    for name, (res, args) in SIGNATURES.items():
        fn = getattr(L, name)  # AttributeError if the library does not export it
        fn.restype = res
        fn.argtypes = args
    if L.c4_abi_version() != ABI_VERSION:
        raise ImportError("libc4engine.so ABI %d != binding %d" % (L.c4_abi_version(), ABI_VERSION))
    _lib = L
    return L


def last_error(handle=None):
    msg = load().c4_last_error(handle)
    return msg.decode("utf-8", "replace") if msg else ""


def check(rc, handle=None):
    if rc != OK:
        raise EngineError(rc, last_error(handle))
