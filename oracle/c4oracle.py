"""ctypes binding for the CPU oracle (oracle/c4_oracle.c).

TEST INFRASTRUCTURE ONLY.  May be imported from tests/, __graft_entry__.smoke() and the
cpu_baseline leg of bench.py -- never from connect4_amd/ (the product).
"""
import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB_PATH = os.path.join(_HERE, "libc4oracle.so")

NONE, XWIN, DRAW, OWIN = -1, 0, 1, 2


def build(force=False):
    src = [os.path.join(_HERE, f) for f in ("c4_oracle.c", "c4_oracle.h")]
    if force or not os.path.exists(_LIB_PATH) or any(
            os.path.getmtime(s) > os.path.getmtime(_LIB_PATH) for s in src if os.path.exists(s)):
        subprocess.check_call(["make", "-s", "-C", _HERE, "libc4oracle.so"])
    return _LIB_PATH


class Board(C.Structure):
    _fields_ = [("color", C.c_uint64 * 2), ("age", C.c_int32), ("result", C.c_int32)]

    @classmethod
    def from_bits(cls, c0, c1):
        b = cls()
        lib().c4o_board_from_bits(C.byref(b), int(c0), int(c1))
        return b

    @classmethod
    def empty(cls):
        b = cls()
        lib().c4o_board_init(C.byref(b))
        return b

    def copy(self):
        b = Board()
        C.memmove(C.byref(b), C.byref(self), C.sizeof(Board))
        return b

    def make_move(self, col):
        return lib().c4o_make_move(C.byref(self), int(col))

    def valid_mask(self):
        return lib().c4o_valid_mask(C.byref(self))

    def planes(self):
        out = np.zeros(126, dtype=np.uint8)
        lib().c4o_planes(C.byref(self), out.ctypes.data_as(C.POINTER(C.c_uint8)))
        return out.reshape(3, 6, 7)

    def key(self):
        return (int(self.color[0]), int(self.color[1]))


class Config(C.Structure):
    _fields_ = [("simulations", C.c_int32), ("pb_c_base", C.c_int32), ("pb_c_init", C.c_double),
                ("root_dirichlet_alpha", C.c_double), ("root_exploration_fraction", C.c_double),
                ("num_sampling_moves", C.c_int32)]


def make_config(simulations, pb_c_base=19652, pb_c_init=1.25, root_dirichlet_alpha=0.0,
                root_exploration_fraction=0.0, num_sampling_moves=0):
    return Config(int(simulations), int(pb_c_base), float(pb_c_init), float(root_dirichlet_alpha),
                  float(root_exploration_fraction), int(num_sampling_moves))


class RootInfo(C.Structure):
    _fields_ = [("root_visits", C.c_uint32), ("root_value_sum", C.c_double),
                ("child_visits", C.c_uint32 * 7), ("child_value_sum", C.c_double * 7),
                ("child_status", C.c_int32 * 7), ("child_value", C.c_double * 7),
                ("values_policy", C.c_double * 7), ("visit_policy", C.c_double * 7),
                ("root_prior", C.c_double * 7), ("best_move", C.c_int32),
                ("n_nodes", C.c_int64), ("n_expansions", C.c_int64),
                ("n_children_created", C.c_int64), ("n_terminal_sims", C.c_int64),
                ("n_evals", C.c_int64), ("depth_sum", C.c_int64), ("depth_max", C.c_int32)]


class MoveRecord(C.Structure):
    _fields_ = [("color0", C.c_uint64), ("color1", C.c_uint64), ("move", C.c_int32),
                ("value", C.c_double), ("policy", C.c_double * 7)]


class TableEntry(C.Structure):
    _fields_ = [("c0", C.c_uint64), ("c1", C.c_uint64), ("value", C.c_float), ("prior", C.c_float * 7)]


class Table(C.Structure):
    _fields_ = [("entries", C.POINTER(TableEntry)), ("n", C.c_int64), ("prior_f32", C.c_int),
                ("misses", C.c_int64)]


EVAL_FN = C.CFUNCTYPE(C.c_int, C.c_void_p, C.POINTER(Board), C.POINTER(C.c_double), C.POINTER(C.c_double))

_lib = None


def lib():
    global _lib
    if _lib is None:
        build()
        L = C.CDLL(_LIB_PATH)
        L.c4o_board_init.argtypes = [C.POINTER(Board)]
        L.c4o_board_from_bits.argtypes = [C.POINTER(Board), C.c_uint64, C.c_uint64]
        L.c4o_board_from_pieces.argtypes = [C.POINTER(Board), C.POINTER(C.c_uint8), C.POINTER(C.c_uint8)]
        L.c4o_wins.argtypes = [C.c_uint64]
        L.c4o_make_move.argtypes = [C.POINTER(Board), C.c_int]
        L.c4o_valid_mask.argtypes = [C.POINTER(Board)]
        L.c4o_planes.argtypes = [C.POINTER(Board), C.POINTER(C.c_uint8)]
        L.c4o_flip_color.argtypes = [C.c_uint64]
        L.c4o_flip_color.restype = C.c_uint64
        L.c4o_fliplr.argtypes = [C.POINTER(Board), C.POINTER(Board)]
        L.c4o_make_random_ips.argtypes = [C.c_int, C.POINTER(C.c_uint64), C.POINTER(C.c_uint64), C.c_int]
        L.c4o_evaluate_centre.argtypes = [C.POINTER(Board)]
        L.c4o_evaluate_centre.restype = C.c_double
        L.c4o_tree_new.argtypes = [C.POINTER(Config), C.POINTER(Board)]
        L.c4o_tree_new.restype = C.c_void_p
        L.c4o_tree_free.argtypes = [C.c_void_p]
        L.c4o_tree_root_request.argtypes = [C.c_void_p, C.POINTER(Board)]
        L.c4o_tree_root_apply.argtypes = [C.c_void_p, C.c_double, C.POINTER(C.c_double), C.c_int,
                                          C.POINTER(C.c_double)]
        L.c4o_tree_select.argtypes = [C.c_void_p, C.POINTER(Board)]
        L.c4o_tree_apply.argtypes = [C.c_void_p, C.c_double, C.POINTER(C.c_double), C.c_int]
        L.c4o_search.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.POINTER(C.c_double)]
        L.c4o_tree_root_info.argtypes = [C.c_void_p, C.POINTER(RootInfo)]
        L.c4o_tree_pick_move.argtypes = [C.c_void_p, C.c_int, C.c_double, C.POINTER(C.c_double)]
        L.c4o_selfplay_game.argtypes = [C.POINTER(Config), C.c_void_p, C.c_void_p,
                                        C.POINTER(C.c_double), C.POINTER(C.c_double),
                                        C.POINTER(MoveRecord), C.POINTER(C.c_int),
                                        C.POINTER(C.c_int64), C.POINTER(C.c_int64), C.POINTER(C.c_int64)]
        L.c4o_set_threads.argtypes = [C.c_int]
        L.c4o_pool_new.argtypes = [C.POINTER(Config), C.c_int, C.c_uint64]
        L.c4o_pool_new.restype = C.c_void_p
        L.c4o_pool_free.argtypes = [C.c_void_p]
        L.c4o_pool_collect.argtypes = [C.c_void_p, C.c_void_p]
        L.c4o_pool_apply.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p]
        L.c4o_pool_stats.argtypes = [C.c_void_p] + [C.POINTER(C.c_int64)] * 5
        L.c4o_replay_new.argtypes = [C.POINTER(Config), C.c_int, C.c_void_p, C.c_void_p]
        L.c4o_replay_new.restype = C.c_void_p
        L.c4o_replay_free.argtypes = [C.c_void_p]
        L.c4o_replay_collect.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]
        L.c4o_replay_apply.argtypes = [C.c_void_p, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p]
        L.c4o_replay_game.argtypes = [C.c_void_p, C.c_int, C.POINTER(MoveRecord), C.POINTER(C.c_int)]
        L.c4o_replay_stats.argtypes = [C.c_void_p] + [C.POINTER(C.c_int64)] * 3
        _lib = L
    return _lib


def wins(stones):
    return bool(lib().c4o_wins(int(stones)))


def from_pieces(o, x):
    o = np.ascontiguousarray(np.asarray(o, dtype=np.uint8).reshape(42))
    x = np.ascontiguousarray(np.asarray(x, dtype=np.uint8).reshape(42))
    b = Board()
    lib().c4o_board_from_pieces(C.byref(b), o.ctypes.data_as(C.POINTER(C.c_uint8)),
                                x.ctypes.data_as(C.POINTER(C.c_uint8)))
    return b


def flip_color(stones):
    return int(lib().c4o_flip_color(int(stones)))


def make_random_ips(plies):
    cap = 7 ** plies
    c0 = (C.c_uint64 * cap)()
    c1 = (C.c_uint64 * cap)()
    n = lib().c4o_make_random_ips(plies, c0, c1, cap)
    return [(int(c0[i]), int(c1[i])) for i in range(n)]


def evaluate_centre(board):
    return float(lib().c4o_evaluate_centre(C.byref(board)))


def _dptr(a):
    return a.ctypes.data_as(C.POINTER(C.c_double)) if a is not None else None


class TableEvaluator:
    """Sorted (c0,c1) -> (value f32, prior f32[7]) table usable as a native evaluator."""

    def __init__(self, keys_c0, keys_c1, values, priors, prior_f32=True):
        order = np.lexsort((np.asarray(keys_c1, dtype=np.uint64), np.asarray(keys_c0, dtype=np.uint64)))
        n = len(order)
        self._entries = (TableEntry * n)()
        for j, i in enumerate(order):
            e = self._entries[j]
            e.c0 = int(keys_c0[i])
            e.c1 = int(keys_c1[i])
            e.value = float(values[i])
            for k in range(7):
                e.prior[k] = float(priors[i][k])
        self.table = Table(C.cast(self._entries, C.POINTER(TableEntry)), n, 1 if prior_f32 else 0, 0)
        self.fn = C.cast(lib().c4o_eval_table, C.c_void_p)
        self.ctx = C.cast(C.pointer(self.table), C.c_void_p)


class CentreEvaluator:
    def __init__(self):
        self.fn = C.cast(lib().c4o_eval_centre_with_prior, C.c_void_p)
        self.ctx = None


class CallbackEvaluator:
    """Wraps a Python callable (c0, c1) -> (value, prior[7], prior_is_f32)."""

    def __init__(self, fn):
        def _cb(ctx, bptr, vptr, pptr):
            b = bptr.contents
            v, p, f32 = fn(int(b.color[0]), int(b.color[1]))
            vptr[0] = float(v)
            for k in range(7):
                pptr[k] = float(p[k])
            return 1 if f32 else 0
        self._keep = EVAL_FN(_cb)
        self.fn = C.cast(self._keep, C.c_void_p)
        self.ctx = None


def search(cfg, board, evaluator, gamma_noise=None):
    """mcts.py:94-121.  Returns RootInfo."""
    L = lib()
    t = L.c4o_tree_new(C.byref(cfg), C.byref(board))
    try:
        noise = None if gamma_noise is None else np.ascontiguousarray(gamma_noise, dtype=np.float64)
        rc = L.c4o_search(t, evaluator.fn, evaluator.ctx, _dptr(noise))
        if rc < 0:
            raise RuntimeError("oracle evaluator failed (rc=%d)" % rc)
        info = RootInfo()
        L.c4o_tree_root_info(t, C.byref(info))
        return info
    finally:
        L.c4o_tree_free(t)


def search_and_pick(cfg, board, evaluator, gamma_noise=None, u=-1.0):
    L = lib()
    t = L.c4o_tree_new(C.byref(cfg), C.byref(board))
    try:
        noise = None if gamma_noise is None else np.ascontiguousarray(gamma_noise, dtype=np.float64)
        rc = L.c4o_search(t, evaluator.fn, evaluator.ctx, _dptr(noise))
        if rc < 0:
            raise RuntimeError("oracle evaluator failed (rc=%d)" % rc)
        info = RootInfo()
        L.c4o_tree_root_info(t, C.byref(info))
        av = C.c_double()
        mv = L.c4o_tree_pick_move(t, board.age, float(u), C.byref(av))
        return info, mv, av.value
    finally:
        L.c4o_tree_free(t)


def selfplay_game(cfg, evaluator, noise_tape=None, u_tape=None):
    """training_game.py:8-19.  Returns dict(moves, boards, values, policies, result, sims, expansions)."""
    L = lib()
    rec = (MoveRecord * 42)()
    res = C.c_int()
    sims, exps, evals = C.c_int64(), C.c_int64(), C.c_int64()
    nt = None if noise_tape is None else np.ascontiguousarray(noise_tape, dtype=np.float64)
    ut = None if u_tape is None else np.ascontiguousarray(u_tape, dtype=np.float64)
    n = L.c4o_selfplay_game(C.byref(cfg), evaluator.fn, evaluator.ctx, _dptr(nt), _dptr(ut), rec,
                            C.byref(res), C.byref(sims), C.byref(exps), C.byref(evals))
    if n < 0:
        raise RuntimeError("oracle selfplay failed (rc=%d)" % n)
    return dict(moves=[rec[i].move for i in range(n)],
                boards=[(int(rec[i].color0), int(rec[i].color1)) for i in range(n)],
                values=[rec[i].value for i in range(n)],
                policies=[list(rec[i].policy) for i in range(n)],
                result=res.value, sims=sims.value, expansions=exps.value, evals=evals.value)


def set_threads(n):
    lib().c4o_set_threads(int(n))


class Pool:
    """Lock-step many-game CPU self-play (cpu_baseline): collect leaf planes -> caller evaluates -> apply."""

    def __init__(self, cfg, n_games, seed=0):
        self.n = n_games
        self._p = lib().c4o_pool_new(C.byref(cfg), n_games, seed)
        self.planes = np.zeros((n_games, 3, 6, 7), dtype=np.uint8)

    def collect(self):
        lib().c4o_pool_collect(self._p, self.planes.ctypes.data)
        return self.planes

    def apply(self, values, priors):
        v = np.ascontiguousarray(values, dtype=np.float32)
        p = np.ascontiguousarray(priors, dtype=np.float32)
        assert v.shape == (self.n,) and p.shape == (self.n, 7)
        lib().c4o_pool_apply(self._p, v.ctypes.data, p.ctypes.data)

    def stats(self):
        vals = [C.c_int64() for _ in range(5)]
        lib().c4o_pool_stats(self._p, *[C.byref(v) for v in vals])
        return dict(zip(("sims", "expansions", "games", "moves", "evals"), [v.value for v in vals]))

    def close(self):
        if self._p:
            lib().c4o_pool_free(self._p)
            self._p = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


class ReplayPool:
    """n tape-driven games of training_game.py:8-19 side by side behind one memoising evaluator table
    (evaluators.py:18-25): collect() -> positions the table lacks -> caller answers -> apply()."""

    def __init__(self, cfg, n_games, noise_tapes=None, u_tapes=None):
        self.n = n_games
        self._noise = None if noise_tapes is None else np.ascontiguousarray(noise_tapes, dtype=np.float64)
        self._u = None if u_tapes is None else np.ascontiguousarray(u_tapes, dtype=np.float64)
        if self._noise is not None:
            assert self._noise.shape == (n_games, 42, 7)
        if self._u is not None:
            assert self._u.shape == (n_games, 42)
        self._cfg = cfg
        self._p = lib().c4o_replay_new(C.byref(cfg), n_games,
                                       None if self._noise is None else self._noise.ctypes.data,
                                       None if self._u is None else self._u.ctypes.data)
        self.c0 = np.zeros(n_games, dtype=np.uint64)
        self.c1 = np.zeros(n_games, dtype=np.uint64)
        self.game_of = np.zeros(n_games, dtype=np.int32)

    def collect(self):
        """Number m of positions wanted; they are c0[:m], c1[:m] (for games game_of[:m]).  0 = every game has ended."""
        return lib().c4o_replay_collect(self._p, self.c0.ctypes.data, self.c1.ctypes.data, self.game_of.ctypes.data)

    def apply(self, m, values, priors):
        v = np.ascontiguousarray(values, dtype=np.float32)
        p = np.ascontiguousarray(priors, dtype=np.float32)
        assert v.shape == (m,) and p.shape == (m, 7)
        lib().c4o_replay_apply(self._p, m, self.game_of.ctypes.data, v.ctypes.data, p.ctypes.data)

    def game(self, i):
        rec = (MoveRecord * 42)()
        res = C.c_int()
        n = lib().c4o_replay_game(self._p, i, rec, C.byref(res))
        if n < 0:
            raise RuntimeError("replay game %d did not finish (rc=%d)" % (i, n))
        return dict(moves=[rec[k].move for k in range(n)],
                    boards=[(int(rec[k].color0), int(rec[k].color1)) for k in range(n)],
                    values=[rec[k].value for k in range(n)],
                    policies=[list(rec[k].policy) for k in range(n)], result=res.value)

    def stats(self):
        vals = [C.c_int64() for _ in range(3)]
        lib().c4o_replay_stats(self._p, *[C.byref(v) for v in vals])
        return dict(zip(("lookups", "hits", "table_entries"), [v.value for v in vals]))

    def close(self):
        if self._p:
            lib().c4o_replay_free(self._p)
            self._p = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass
