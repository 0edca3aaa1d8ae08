"""Thin object wrapper over the C ABI (connect4_amd/_lib.py -> libc4engine.so).

Host plumbing only: numpy for host arrays, torch tensors only as device buffers whose
`data_ptr()` is handed across the ABI.  All search work happens in the HIP kernels.
"""
import ctypes as C

import numpy as np

from . import _lib as L


def _u64(a):
    return np.ascontiguousarray(a, dtype=np.uint64)


def _ptr(a, ctype):
    return a.ctypes.data_as(C.POINTER(ctype))


class Engine:
    """One engine per GPU per process; not thread safe (one host thread drives it)."""

    def __init__(self, n_slots, simulations, pb_c_base=19652, pb_c_init=1.25,
                 root_dirichlet_alpha=0.0, root_exploration_fraction=0.0, num_sampling_moves=0,
                 eval_mode=L.EVAL_EXTERNAL_F32, rng_mode=L.RNG_PHILOX, seed=0, stop_after_move=False,
                 games_target=-1, record_capacity_games=0, max_inner_iters=0,
                 planes_dtype=L.PLANES_F32, eval_cache_log2_entries=0, level_budget=0, time_budget_cycles=0, device=0):
        self._lib = L.load()
        self.cfg = L.Config()
        self.cfg.abi_version = L.ABI_VERSION
        self.cfg.n_slots = int(n_slots)
        self.cfg.simulations = int(simulations)
        self.cfg.pb_c_base = int(pb_c_base)
        self.cfg.pb_c_init = float(pb_c_init)
        self.cfg.root_dirichlet_alpha = float(root_dirichlet_alpha)
        self.cfg.root_exploration_fraction = float(root_exploration_fraction)
        self.cfg.num_sampling_moves = int(num_sampling_moves)
        self.cfg.eval_mode = int(eval_mode)
        self.cfg.rng_mode = int(rng_mode)
        self.cfg.seed = int(seed) & 0xFFFFFFFFFFFFFFFF
        self.cfg.stop_after_move = 1 if stop_after_move else 0
        self.cfg.games_target = int(games_target)
        self.cfg.record_capacity_games = int(record_capacity_games)
        self.cfg.max_inner_iters = int(max_inner_iters)
        self.cfg.planes_dtype = int(planes_dtype)
        self.cfg.eval_cache_log2_entries = int(eval_cache_log2_entries)
        self.cfg.level_budget = int(level_budget)
        self.cfg.time_budget_cycles = int(time_budget_cycles)
        self.n_slots = int(n_slots)
        self.device = int(device)
        self._h = C.c_void_p()
        L.check(self._lib.c4_engine_create(C.byref(self.cfg), self.device, C.byref(self._h)))

    # -- lifecycle ---------------------------------------------------------------------------
    def close(self):
        if getattr(self, "_h", None) is not None and self._h:
            self._lib.c4_engine_destroy(self._h)
            self._h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def __enter__(self):
        return self

    def __exit__(self, *exc):
        self.close()

    def _check(self, rc):
        L.check(rc, self._h)

    def clear_eval_cache(self):
        self._check(self._lib.c4_clear_eval_cache(self._h))

    def set_stream(self, stream_handle):
        self._check(self._lib.c4_set_stream(self._h, C.c_void_p(int(stream_handle))))

    def reset(self, color0=None, color1=None, n_active=None):
        if color0 is None:
            n = self.n_slots if n_active is None else int(n_active)
            self._check(self._lib.c4_reset(self._h, None, None, n))
            return
        c0, c1 = _u64(color0), _u64(color1)
        n = len(c0) if n_active is None else int(n_active)
        assert len(c0) == len(c1) >= n
        self._check(self._lib.c4_reset(self._h, _ptr(c0, C.c_uint64), _ptr(c1, C.c_uint64), n))

    def set_tapes(self, gamma_noise, uniforms):
        """gamma_noise [games][42][7] raw Gamma(alpha,1) draws, uniforms [games][42] (u<0: best_move)."""
        nz = np.ascontiguousarray(gamma_noise, dtype=np.float64)
        u = np.ascontiguousarray(uniforms, dtype=np.float64)
        n = nz.shape[0]
        assert nz.shape == (n, 42, 7) and u.shape == (n, 42)
        self._check(self._lib.c4_set_tapes(self._h, _ptr(nz, C.c_double), _ptr(u, C.c_double), n))

    # -- hot path ----------------------------------------------------------------------------
    def step_ptrs(self, values_ptr, priors_ptr, planes_ptr):
        self._check(self._lib.c4_step(self._h, C.c_void_p(values_ptr or 0), C.c_void_p(priors_ptr or 0),
                                      C.c_void_p(planes_ptr or 0)))

    def step(self, values=None, priors=None, planes=None):
        """values/priors/planes are torch device tensors (or None)."""
        self.step_ptrs(0 if values is None else values.data_ptr(),
                       0 if priors is None else priors.data_ptr(),
                       0 if planes is None else planes.data_ptr())

    def step_range(self, values, priors, planes, slot_lo, slot_count, stream=0):
        self._check(self._lib.c4_step_range(
            self._h, C.c_void_p(0 if values is None else values.data_ptr()),
            C.c_void_p(0 if priors is None else priors.data_ptr()),
            C.c_void_p(0 if planes is None else planes.data_ptr()), int(slot_lo), int(slot_count), C.c_void_p(stream or 0)))

    def run_centre(self, max_launches=64):
        self._check(self._lib.c4_run_centre(self._h, int(max_launches)))

    def leaf_buffers(self):
        a, b, c = C.c_void_p(), C.c_void_p(), C.c_void_p()
        self._check(self._lib.c4_leaf_buffers(self._h, C.byref(a), C.byref(b), C.byref(c)))
        return a.value, b.value, c.value

    def read_leaves(self):
        c0 = np.zeros(self.n_slots, dtype=np.uint64)
        c1 = np.zeros(self.n_slots, dtype=np.uint64)
        has = np.zeros(self.n_slots, dtype=np.int32)
        self._check(self._lib.c4_read_leaves(self._h, _ptr(c0, C.c_uint64), _ptr(c1, C.c_uint64),
                                             _ptr(has, C.c_int32)))
        return c0, c1, has

    # -- read-out ----------------------------------------------------------------------------
    def stats(self):
        s = L.Stats()
        self._check(self._lib.c4_get_stats(self._h, C.byref(s)))
        return s.as_dict()

    def read_roots(self):
        out = (L.RootResult * self.n_slots)()
        self._check(self._lib.c4_read_roots(self._h, out))
        return out

    @property
    def record_capacity(self):
        return int(self.cfg.record_capacity_games or 2 * self.n_slots)

    def finished_games(self):
        """(finished games waiting in the device ring, finished games lost to a full ring)."""
        ready, dropped = C.c_int64(), C.c_int64()
        self._check(self._lib.c4_finished_games(self._h, C.byref(ready), C.byref(dropped)))
        return ready.value, dropped.value

    def drain_games(self, cap=None):
        """Host copy of the finished games (float64 values/policies, for parity tests and GameData)."""
        ready, _ = self.finished_games()
        cap = int(min(cap, ready) if cap is not None else ready)
        if cap <= 0:
            return []
        out = (L.GameRecord * cap)()
        n = C.c_int32()
        self._check(self._lib.c4_drain_games(self._h, out, cap, C.byref(n)))
        return [out[i] for i in range(n.value)]

    _EXPORT_FIELDS = ("boards", "moves", "values", "policy", "targets", "game_index", "lengths", "results", "ids")

    def export_buffers(self, max_games=None, cap_positions=None):
        """Device tensors an export writes into: (dict of tensors, counts int64[2] = {games, positions})."""
        import torch
        max_games = min(int(max_games if max_games is not None else self.record_capacity), self.record_capacity)
        cap_positions = int(cap_positions if cap_positions is not None else 42 * max_games)
        dev = torch.device("cuda", self.device)
        t = dict(boards=torch.empty((cap_positions, 2), dtype=torch.int64, device=dev),
                 moves=torch.empty(cap_positions, dtype=torch.uint8, device=dev),
                 values=torch.empty(cap_positions, dtype=torch.float32, device=dev),
                 policy=torch.empty((cap_positions, 7), dtype=torch.float32, device=dev),
                 targets=torch.empty(cap_positions, dtype=torch.float32, device=dev),
                 game_index=torch.empty(cap_positions, dtype=torch.int32, device=dev),
                 lengths=torch.empty(max_games, dtype=torch.int32, device=dev),
                 results=torch.empty(max_games, dtype=torch.int8, device=dev),
                 ids=torch.empty(max_games, dtype=torch.int64, device=dev))
        return t, torch.zeros(2, dtype=torch.int64, device=dev)

    def export_games_async(self, tensors, counts, stream=0):
        """Queue one device-side export (c4_export_games_dev) behind the launches on `stream`: consumes the
        finished games that fit `tensors`, writes {games, positions} to `counts`.  No host synchronisation."""
        import torch
        if not stream:
            stream = torch.cuda.current_stream(torch.device("cuda", self.device)).cuda_stream
        b = L.ExportBuffers(*[tensors[k].data_ptr() for k in self._EXPORT_FIELDS])
        self._check(self._lib.c4_export_games_dev(self._h, C.byref(b), int(tensors["lengths"].shape[0]),
                                                  int(tensors["moves"].shape[0]), C.c_void_p(counts.data_ptr()),
                                                  C.c_void_p(stream)))

    def export_games(self, max_games=None, cap_positions=None, stream=0):
        """Device-side export: consume up to max_games finished games into torch DEVICE tensors (compact
        record, ~50 B/position) with no per-game host work.  Returns PackedGames (connect4_amd.packed)."""
        from .packed import PackedGames
        t, counts = self.export_buffers(max_games, cap_positions)
        self.export_games_async(t, counts, stream)
        n_games, n_pos = [int(x) for x in counts.tolist()]     # the one synchronisation of the export
        per_pos = ("boards", "moves", "values", "policy", "targets", "game_index")
        return PackedGames(**{k: (v[:n_pos] if k in per_pos else v[:n_games]) for k, v in t.items()})

    def cache_lookup(self, color0, color1):
        """What the evaluation cache answers for the positions: (values f32[n], priors f32[n,7], found bool[n])."""
        c0, c1 = _u64(color0), _u64(color1)
        n = len(c0)
        v = np.zeros(n, dtype=np.float32)
        p = np.zeros((n, 7), dtype=np.float32)
        f = np.zeros(n, dtype=np.int32)
        self._check(self._lib.c4_eval_cache_lookup(self._h, _ptr(c0, C.c_uint64), _ptr(c1, C.c_uint64), n,
                                                   _ptr(v, C.c_float), _ptr(p, C.c_float), _ptr(f, C.c_int32)))
        return v, p, f.astype(bool)


# -- pure board functions executed by the device code ------------------------------------------
def board_make_move(color0, color1, cols, device=0):
    c0, c1 = _u64(color0), _u64(color1)
    col = np.ascontiguousarray(cols, dtype=np.int32)
    o0, o1 = np.zeros_like(c0), np.zeros_like(c1)
    res = np.zeros(len(c0), dtype=np.int32)
    lib = L.load()
    L.check(lib.c4_board_make_move(device, _ptr(c0, C.c_uint64), _ptr(c1, C.c_uint64), _ptr(col, C.c_int32),
                                   len(c0), _ptr(o0, C.c_uint64), _ptr(o1, C.c_uint64), _ptr(res, C.c_int32)))
    return o0, o1, res


def board_wins(stones, device=0):
    s = _u64(stones)
    out = np.zeros(len(s), dtype=np.int32)
    L.check(L.load().c4_board_wins(device, _ptr(s, C.c_uint64), len(s), _ptr(out, C.c_int32)))
    return out


def board_valid_mask(color0, color1, device=0):
    c0, c1 = _u64(color0), _u64(color1)
    out = np.zeros(len(c0), dtype=np.int32)
    L.check(L.load().c4_board_valid_mask(device, _ptr(c0, C.c_uint64), _ptr(c1, C.c_uint64), len(c0),
                                         _ptr(out, C.c_int32)))
    return out


def board_planes(color0, color1, device=0):
    c0, c1 = _u64(color0), _u64(color1)
    out = np.zeros((len(c0), 3, 6, 7), dtype=np.float32)
    L.check(L.load().c4_board_planes(device, _ptr(c0, C.c_uint64), _ptr(c1, C.c_uint64), len(c0),
                                     _ptr(out, C.c_float)))
    return out


def board_fliplr(color0, color1, device=0):
    c0, c1 = _u64(color0), _u64(color1)
    o0, o1 = np.zeros_like(c0), np.zeros_like(c1)
    L.check(L.load().c4_board_fliplr(device, _ptr(c0, C.c_uint64), _ptr(c1, C.c_uint64), len(c0),
                                     _ptr(o0, C.c_uint64), _ptr(o1, C.c_uint64)))
    return o0, o1


def training_tensors(boards, targets, policy, add_fliplr=True, stream=0):
    """native_to_pytorch (data.py:78-105) on device: torch device tensors in (int64 [n,2], f32 [n], f32 [n,7]),
    torch device tensors out (boards F32[m,3,6,7], values F32[m], priors F32[m,7]), m = 2n with the mirrored copies."""
    import torch
    n = int(boards.shape[0])
    dev = boards.device
    m = n * (2 if add_fliplr else 1)
    ob = torch.empty((m, 3, 6, 7), dtype=torch.float32, device=dev)
    ov = torch.empty(m, dtype=torch.float32, device=dev)
    op = torch.empty((m, 7), dtype=torch.float32, device=dev)
    if n:
        boards, targets, policy = boards.contiguous(), targets.contiguous(), policy.contiguous()
        if not stream:
            stream = torch.cuda.current_stream(dev).cuda_stream
        L.check(L.load().c4_training_tensors_dev(dev.index or 0, C.c_void_p(stream), C.c_void_p(boards.data_ptr()),
                                                 C.c_void_p(targets.data_ptr()), C.c_void_p(policy.data_ptr()), n,
                                                 1 if add_fliplr else 0, C.c_void_p(ob.data_ptr()), C.c_void_p(ov.data_ptr()),
                                                 C.c_void_p(op.data_ptr())))
    return ob, ov, op


def debug_root_noise(seed, alpha, game_id, ply, legal_mask, device=0):
    gid = np.ascontiguousarray(game_id, dtype=np.int64)
    pl = np.ascontiguousarray(ply, dtype=np.int32)
    lm = np.ascontiguousarray(legal_mask, dtype=np.int32)
    n = len(gid)
    raw = np.zeros((n, 7))
    dr = np.zeros((n, 7))
    L.check(L.load().c4_debug_root_noise(device, int(seed) & 0xFFFFFFFFFFFFFFFF, float(alpha), _ptr(gid, C.c_int64), _ptr(pl, C.c_int32),
                                         _ptr(lm, C.c_int32), n, _ptr(raw, C.c_double), _ptr(dr, C.c_double)))
    return raw, dr


def debug_sample_move(seed, game_id, ply, child_values, n_children, uniforms=None, device=0):
    gid = np.ascontiguousarray(game_id, dtype=np.int64)
    pl = np.ascontiguousarray(ply, dtype=np.int32)
    cv = np.ascontiguousarray(child_values, dtype=np.float64)
    nc = np.ascontiguousarray(n_children, dtype=np.int32)
    n = len(gid)
    assert cv.shape == (n, 7)
    u_in = None if uniforms is None else np.ascontiguousarray(uniforms, dtype=np.float64)
    u = np.zeros(n)
    ch = np.zeros(n, dtype=np.int32)
    L.check(L.load().c4_debug_sample_move(device, int(seed) & 0xFFFFFFFFFFFFFFFF, _ptr(gid, C.c_int64), _ptr(pl, C.c_int32),
                                          _ptr(cv, C.c_double), _ptr(nc, C.c_int32),
                                          None if u_in is None else _ptr(u_in, C.c_double), n, _ptr(u, C.c_double), _ptr(ch, C.c_int32)))
    return u, ch


def debug_div_mismatches(max_parent_visits=4096, max_child_visits=4096, n_random=1 << 26, device=0):
    """Quotients of the engine's written-out float64 division that differ from `a / b` on the device (must be 0)."""
    out = np.zeros(1, dtype=np.int64)
    L.check(L.load().c4_debug_div_mismatches(device, int(max_parent_visits), int(max_child_visits), int(n_random), _ptr(out, C.c_int64)))
    return int(out[0])


def board_centre_value(color0, color1, device=0):
    c0, c1 = _u64(color0), _u64(color1)
    out = np.zeros(len(c0), dtype=np.float64)
    L.check(L.load().c4_board_centre_value(device, _ptr(c0, C.c_uint64), _ptr(c1, C.c_uint64), len(c0),
                                           _ptr(out, C.c_double)))
    return out
