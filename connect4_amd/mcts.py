"""MCTS player with the reference's constructor and ``make_move`` contract
(oinkoink/mcts.py:69-88), executed by the HIP engine.

``MCTS(name, MCTSConfig, evaluator).make_move(board)`` runs one search on the GPU, applies the chosen
move to the caller's board in place and returns ``(move, value, tree)``.  Randomness is drawn from
``np.random`` with the very calls the reference makes (``np.random.gamma`` for the root noise,
mcts.py:175-177; one uniform for ``np.random.choice``, tree.py:80) and injected into the kernel, so a
caller that seeds NumPy gets the reference's games move for move.
"""
from typing import List, Sequence

import numpy as np

from . import _lib as L
from .board import Board
from .config import MCTSConfig
from .engine import Engine
from .evaluators import DeviceNetEvaluator, Evaluator, evaluate_centre_with_prior, unwrap
from .player import BasePlayer
from .tree import Tree

__all__ = ["MCTSConfig", "MCTS", "search"]


def _host_eval(evaluator, board):
    v, p = evaluator(board)
    p = np.asarray(p)
    return float(v), p


class _Searcher:
    """Owns ONE stop-after-move engine per evaluator mode, sized for the largest batch seen; smaller batches
    run on its first n slots (c4_reset parks the rest), so a match whose live-board count shrinks every ply
    keeps a single node pool instead of one per batch size."""

    def __init__(self, config: MCTSConfig, evaluator, device=0):
        self.config = config
        self.evaluator = evaluator
        self.device = device
        self._engines = {}
        fn = unwrap(evaluator)
        if fn is evaluate_centre_with_prior:
            self.kind = "centre"
        elif isinstance(fn, DeviceNetEvaluator) or isinstance(evaluator, DeviceNetEvaluator):
            self.kind = "device"
            self.dev_eval = fn if isinstance(fn, DeviceNetEvaluator) else evaluator
        else:
            self.kind = "host"

    def _engine(self, mode, n):
        eng = self._engines.get(mode)
        if eng is None or eng.n_slots < n:
            if eng is not None:
                eng.close()
            eng = self._engines[mode] = Engine(n, eval_mode=mode, rng_mode=L.RNG_TAPE, stop_after_move=True,
                                               device=self.device, **self.config.engine_kwargs())
        return eng

    def close(self):
        for e in self._engines.values():
            e.close()
        self._engines = {}

    def _tapes(self, boards, pick=True):
        cfg = self.config
        n = len(boards)
        nz = np.zeros((n, 42, 7))
        u = np.full((n, 42), -1.0)
        noisy = bool(cfg.root_dirichlet_alpha and cfg.root_exploration_fraction)
        for i, b in enumerate(boards):           # same draw order as the reference, board by board
            if noisy:
                nz[i, 0] = np.random.gamma(cfg.root_dirichlet_alpha, 1, 7)
            if pick and b.age < cfg.num_sampling_moves:
                u[i, 0] = np.random.random_sample()
        return nz, u

    def run(self, boards: Sequence[Board], pick=True):
        for b in boards:
            if b.result is not None:
                raise ValueError("cannot search a finished position")
        n = len(boards)
        c0 = [b.color[0] for b in boards]
        c1 = [b.color[1] for b in boards]
        if self.kind == "centre":
            eng = self._engine(L.EVAL_CENTRE, n)
            eng.set_tapes(*self._tapes(boards, pick))
            eng.reset(c0, c1)
            eng.run_centre()
        elif self.kind == "device":
            eng = self._engine(L.EVAL_EXTERNAL_F32, n)
            eng.set_tapes(*self._tapes(boards, pick))
            eng.reset(c0, c1)
            self._drive_device(eng)
        else:
            # the prior's dtype decides the score arithmetic (float32 net output vs float64)
            _, p0 = _host_eval(self.evaluator, boards[0])
            f32 = p0.dtype == np.float32
            eng = self._engine(L.EVAL_EXTERNAL_F32 if f32 else L.EVAL_EXTERNAL_F64, n)
            eng.set_tapes(*self._tapes(boards, pick))
            eng.reset(c0, c1)
            self._drive_host(eng, np.float32 if f32 else np.float64)
        return list(eng.read_roots()[:n])

    def _drive_device(self, eng):
        import torch
        dev = torch.device("cuda", self.device)
        G = eng.n_slots
        values = torch.zeros(G, dtype=torch.float32, device=dev)
        priors = torch.zeros(G, 7, dtype=torch.float32, device=dev)
        planes = torch.zeros(G, 3, 6, 7, dtype=torch.float32, device=dev)
        stream = torch.cuda.current_stream(dev).cuda_stream
        eng.set_stream(stream)
        net = self.dev_eval.net
        bits = bool(getattr(net, "from_bitboards", False))   # fused kernel: evaluates the leaves' bitboards
        c0p, c1p, _ = eng.leaf_buffers()
        eng.step(None, None, None if bits else planes)
        steps = 0
        while True:
            if bits:
                net.forward_bitboards(c0p, c1p, G, values, priors, stream)
            else:
                v, p = net(planes)
                values.copy_(v)
                priors.copy_(p)
            eng.step(values, priors, None if bits else planes)
            steps += 1
            if steps % 64 == 0 and eng.stats()["active_slots"] == 0:
                break

    def _drive_host(self, eng, dtype):
        import torch
        dev = torch.device("cuda", self.device)
        G = eng.n_slots
        tdt = torch.float32 if dtype == np.float32 else torch.float64
        values = torch.zeros(G, dtype=tdt, device=dev)
        priors = torch.zeros(G, 7, dtype=tdt, device=dev)
        hv = np.zeros(G, dtype=dtype)
        hp = np.zeros((G, 7), dtype=dtype)
        eng.set_stream(torch.cuda.current_stream(dev).cuda_stream)
        eng.step(None, None, None)
        while True:
            c0, c1, has = eng.read_leaves()
            if not has.any() and eng.stats()["active_slots"] == 0:
                break
            for g in np.nonzero(has)[0]:
                v, p = _host_eval(self.evaluator, Board.from_bits(int(c0[g]), int(c1[g])))
                hv[g] = v
                hp[g] = p
            values.copy_(torch.from_numpy(hv))
            priors.copy_(torch.from_numpy(hp))
            eng.step(values, priors, None)


class MCTS(BasePlayer):
    def __init__(self, name: str, config: MCTSConfig, evaluator, device: int = 0):
        super().__init__(name)
        self.config = config
        self.evaluator = evaluator
        self.device = device
        self._searcher = None

    def _s(self):
        if self._searcher is None:
            self._searcher = _Searcher(self.config, self.evaluator, self.device)
        return self._searcher

    def make_moves(self, boards: List[Board]):
        """Batch form: one search per board, all on the GPU at once."""
        roots = self._s().run(boards)
        out = []
        for b, r in zip(boards, roots):
            tree = Tree(r, b)
            value = None if np.isnan(r.value) else float(r.value)
            b.make_move(int(r.move))
            out.append((int(r.move), value, tree))
        return out

    def make_move(self, board: Board):
        return self.make_moves([board])[0]

    def __copy__(self):                 # match.py:26-40 copies players per game
        return MCTS(self.name, self.config, self.evaluator, self.device)

    def __getstate__(self):             # picklable for Pool.map (match.py:72-76)
        d = dict(self.__dict__)
        d["_searcher"] = None
        return d

    def __str__(self):
        return super().__str__() + ", type: Computer"


def search(config: MCTSConfig, board: Board, evaluator, device: int = 0) -> Tree:
    """mcts.py:94-121: run the simulations and return the tree (the board is not modified)."""
    s = _Searcher(config, evaluator, device)
    try:
        return Tree(s.run([board], pick=False)[0], board)
    finally:
        s.close()
