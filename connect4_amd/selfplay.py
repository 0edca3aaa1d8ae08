"""Device-resident self-play; no host round trip per simulation.  With a FusedNet and fused_loop=True
(what generate_games uses) the whole loop is ONE persistent kernel, c4_selfplay_steps: every wave
alternates the tree walk of its own games with the policy/value network on their leaves.  Otherwise
the HIP rollout-step kernel and the leaf-batch network (any device callable) alternate on one HIP
stream, captured as a HIP graph.

Stands in for TrainingLoop._generate_games' process/thread/pipe scaffolding
(oinkoink/neural/training.py:99-135, game_pool.py:15-42, inference_server.py:37-63): every slot is
one sequential-MCTS game (no virtual loss, exactly the reference's per-tree semantics); the
"inference server" is simply the next kernels on the stream.
"""
import time
from typing import Callable, List, Optional

import torch

from . import _lib as L
from .config import MCTSConfig
from .engine import Engine
from .training_game import GameData, game_data_from_record

_PLANES = {torch.float32: L.PLANES_F32, torch.float16: L.PLANES_F16, torch.bfloat16: L.PLANES_BF16}


class SelfPlay:
    """n_slots concurrent games on one GPU.  `net` maps planes [n,3,6,7] -> (values [n] fp32 in
    [0,1], priors [n,7] fp32) on the device (connect4_amd.net.InferenceNet or any callable)."""

    def __init__(self, net: Callable, n_slots: int, config: MCTSConfig, seed: int = 0, device: int = 0,
                 games_target: int = -1, record_capacity_games: int = 0, planes_dtype=torch.float32,
                 use_graph: bool = True, steps_per_graph: int = 8, max_inner_iters: int = 8,
                 eval_cache_log2_entries: int = 0, level_budget: int = 0, time_budget_cycles: int = 80000, pipeline: int = 1,
                 fused_loop: bool = False, steps_per_launch: int = 128, rng_mode: int = L.RNG_PHILOX):
        self.net = net
        self.n_slots = n_slots
        self.config = config
        self.device = torch.device("cuda", device)
        self.engine = Engine(n_slots, eval_mode=L.EVAL_EXTERNAL_F32, rng_mode=rng_mode, seed=seed,
                             stop_after_move=False, games_target=games_target,
                             record_capacity_games=record_capacity_games, max_inner_iters=max_inner_iters,
                             planes_dtype=_PLANES[planes_dtype], eval_cache_log2_entries=eval_cache_log2_entries, level_budget=level_budget, time_budget_cycles=time_budget_cycles,
                             device=device, **config.engine_kwargs())
        with torch.cuda.device(self.device):
            self.values = torch.zeros(n_slots, dtype=torch.float32, device=self.device)
            self.priors = torch.full((n_slots, 7), 1.0 / 7.0, dtype=torch.float32, device=self.device)
            self.planes = torch.zeros(n_slots, 3, 6, 7, dtype=planes_dtype, device=self.device)
        self._bits = bool(getattr(net, "from_bitboards", False))   # fused kernel reads the leaf bitboards
        # pipeline=2: the batch is split in two halves on two HIP streams, phase-shifted, so the
        # (latency-bound) tree kernel of one half runs under the (MFMA-bound) network of the other
        self._pipeline = 2 if (pipeline == 2 and self._bits and n_slots % 16 == 0) else 1
        self._streams = None
        # fused_loop: tree step + network in ONE persistent kernel (c4_selfplay_steps), no graph needed
        self._fused_loop = bool(fused_loop and self._bits)
        self._steps_per_launch = max(1, steps_per_launch)
        self._leaf_c0, self._leaf_c1, _ = self.engine.leaf_buffers()
        self.steps_done = 0
        self.steps_per_graph = max(1, steps_per_graph)
        self._graph = None
        self._use_graph = use_graph

    # one rollout step = tree kernel (apply previous answers, select, emit leaves) + leaf evaluation
    def _net_range(self, lo, cnt, stream):
        self.net.forward_bitboards(self._leaf_c0 + 8 * lo, self._leaf_c1 + 8 * lo, cnt, self.values[lo:lo + cnt],
                                   self.priors[lo:lo + cnt], stream.cuda_stream)

    def _steps_pipelined(self, k):
        """k rounds for both halves: A = tree,net,tree,net...  B = net,tree,net,tree..."""
        cur = torch.cuda.current_stream(self.device)
        if self._streams is None:
            self._streams = (torch.cuda.Stream(self.device), torch.cuda.Stream(self.device))
        sa, sb = self._streams
        half = self.n_slots // 2
        sa.wait_stream(cur)
        sb.wait_stream(cur)
        with torch.cuda.stream(sa):
            for _ in range(k):
                self.engine.step_range(self.values, self.priors, None, 0, half, sa.cuda_stream)
                self._net_range(0, half, sa)
        with torch.cuda.stream(sb):
            for _ in range(k):
                self._net_range(half, self.n_slots - half, sb)
                self.engine.step_range(self.values, self.priors, None, half, self.n_slots - half, sb.cuda_stream)
        cur.wait_stream(sa)
        cur.wait_stream(sb)

    def _step_eager(self):
        if self._pipeline == 2:
            self._steps_pipelined(1)
            return
        stream = torch.cuda.current_stream(self.device).cuda_stream
        self.engine.set_stream(stream)
        if self._bits:
            self.engine.step(self.values, self.priors, None)
            self.net.forward_bitboards(self._leaf_c0, self._leaf_c1, self.n_slots, self.values, self.priors, stream)
            return
        self.engine.step(self.values, self.priors, self.planes)
        v, p = self.net(self.planes)
        self.values.copy_(v)
        self.priors.copy_(p)

    def _capture(self):
        with torch.cuda.device(self.device):
            side = torch.cuda.Stream()
            side.wait_stream(torch.cuda.current_stream())
            with torch.cuda.stream(side):       # warm up allocator / MIOpen find before capture
                for _ in range(3):
                    self._step_eager()
                    self.steps_done += 1
            torch.cuda.current_stream().wait_stream(side)
            torch.cuda.synchronize()
            g = torch.cuda.CUDAGraph()
            with torch.cuda.graph(g):
                if self._pipeline == 2:
                    self._steps_pipelined(self.steps_per_graph)
                else:
                    for _ in range(self.steps_per_graph):
                        self._step_eager()
            self._graph = g

    def run_steps(self, k: int):
        """Advance every slot by k rollout steps (asynchronous; call synchronize() to wait)."""
        if self._fused_loop:
            import ctypes as C
            with torch.cuda.device(self.device):
                stream = torch.cuda.current_stream(self.device).cuda_stream
                self.engine.set_stream(stream)     # read-outs and exports order behind these launches
                while k > 0:
                    n = min(k, self._steps_per_launch)
                    rc = self.engine._lib.c4_selfplay_steps(self.engine._h, self.net._h, C.c_void_p(self.values.data_ptr()),
                                                            C.c_void_p(self.priors.data_ptr()), n, C.c_void_p(stream))
                    L.check(rc, self.engine._h)
                    k -= n
                    self.steps_done += n
            return
        with torch.cuda.device(self.device):
            if self._use_graph and self._graph is None and k >= self.steps_per_graph + 3:
                self._capture()
                k -= 3
            if self._graph is not None:
                self.engine.set_stream(torch.cuda.current_stream().cuda_stream)
                while k >= self.steps_per_graph:
                    self._graph.replay()
                    k -= self.steps_per_graph
                    self.steps_done += self.steps_per_graph
            for _ in range(k):
                self._step_eager()
                self.steps_done += 1

    def synchronize(self):
        torch.cuda.synchronize(self.device)

    def stats(self):
        s = self.engine.stats()
        s["launches"] = self.steps_done
        return s

    def drain(self) -> List[GameData]:
        return [game_data_from_record(r) for r in self.engine.drain_games()]

    def close(self):
        self._graph = None
        self.engine.close()


def _play_to_completion(config, net, n_games, n_slots, seed, device, planes_dtype, poll_steps, use_graph, timeout_s):
    n_slots = min(n_games, n_slots or 4096)
    fused = bool(getattr(net, "from_bitboards", False))
    sp = SelfPlay(net, n_slots, config, seed=seed, device=device, games_target=n_games,
                  record_capacity_games=n_games, planes_dtype=planes_dtype, use_graph=use_graph,
                  fused_loop=fused, max_inner_iters=32 if fused else 8)
    t0 = time.time()
    try:
        while True:
            sp.run_steps(poll_steps)
            st = sp.stats()
            if st["active_slots"] == 0:
                break
            if time.time() - t0 > timeout_s:
                raise TimeoutError("self-play did not finish: %r" % (st,))
        if st["dropped_games"]:
            raise RuntimeError("%d finished games were dropped by a full record ring" % st["dropped_games"])
    except BaseException:
        sp.close()
        raise
    return sp


def generate_games_packed(config: MCTSConfig, net: Callable, n_games: int, n_slots: Optional[int] = None,
                          seed: int = 0, device: int = 0, planes_dtype=torch.float32, poll_steps: int = 256,
                          use_graph: bool = True, timeout_s: float = 3600.0):
    """Play n_games self-play games and return them as PackedGames on the device, sorted by game id: the
    engine packs the finished games itself (c4_export_games_dev), nothing is looped over on the host."""
    from .packed import PackedGames
    if n_games <= 0:
        return PackedGames.empty(torch.device("cuda", device))
    sp = _play_to_completion(config, net, n_games, n_slots, seed, device, planes_dtype, poll_steps, use_graph, timeout_s)
    try:
        packed = sp.engine.export_games(n_games)
    finally:
        sp.close()
    assert packed.n_games == n_games, (packed.n_games, n_games)
    return packed.sorted_by_id()


def generate_games(config: MCTSConfig, net: Callable, n_games: int, n_slots: Optional[int] = None,
                   seed: int = 0, device: int = 0, planes_dtype=torch.float32, poll_steps: int = 256,
                   use_graph: bool = True, timeout_s: float = 3600.0) -> List[GameData]:
    """Play n_games self-play games and return them as GameData (training_game.py:42-67), the list
    TrainingLoop._generate_games hands to data_storage.save (training.py:131-135); float64 values and
    policies straight from the engine's records (c4_drain_games)."""
    if n_games <= 0:
        return []
    sp = _play_to_completion(config, net, n_games, n_slots, seed, device, planes_dtype, poll_steps, use_graph, timeout_s)
    try:
        games = sp.drain()
    finally:
        sp.close()
    games.sort(key=lambda g: g.game_id)
    assert len(games) == n_games, (len(games), n_games)
    return games
