"""Finished self-play games as a handful of tensors instead of Python objects.

``PackedGames`` is the compact record of SURVEY.md section 8e (what ``GameData`` holds,
oinkoink/neural/training_game.py:42-67, at ~50 bytes per position): per position the board before the
move (two int64 bitboards), the move, the chosen child's value, the values policy and the game's result
value; per game its length, result code and id.  The engine writes it on the device
(``Engine.export_games`` -> c4_export_games_dev), ``all_gather_packed`` moves it between ranks as
device tensors (RCCL) and ``training_tensors`` turns it into the reference's data.pth tensors on the
device (c4_training_tensors_dev) -- no per-position Python anywhere on that path.  ``to_game_data`` /
``from_game_data`` convert to and from the reference's object form when a caller wants it.
"""
from typing import List

import numpy as np

from .utils import CODE_FROM_RESULT, RESULT_FROM_CODE

_PER_POS = ("boards", "moves", "values", "policy", "targets", "game_index")
_PER_GAME = ("lengths", "results", "ids")


class PackedGames:
    def __init__(self, boards, moves, values, policy, targets, game_index, lengths, results, ids):
        self.boards, self.moves, self.values, self.policy = boards, moves, values, policy
        self.targets, self.game_index = targets, game_index
        self.lengths, self.results, self.ids = lengths, results, ids

    # -- shape -------------------------------------------------------------------------------
    @property
    def n_games(self):
        return int(self.lengths.shape[0])

    @property
    def n_positions(self):
        return int(self.moves.shape[0])

    def __len__(self):
        return self.n_games

    @property
    def device(self):
        return self.boards.device

    def _map(self, fn):
        return PackedGames(**{k: fn(getattr(self, k)) for k in _PER_POS + _PER_GAME})

    def to(self, device):
        return self._map(lambda t: t.to(device))

    def cpu(self):
        return self.to("cpu")

    @staticmethod
    def empty(device="cpu"):
        import torch
        z = lambda dt, *s: torch.zeros(s, dtype=dt, device=device)  # noqa: E731
        return PackedGames(z(torch.int64, 0, 2), z(torch.uint8, 0), z(torch.float32, 0), z(torch.float32, 0, 7),
                           z(torch.float32, 0), z(torch.int32, 0), z(torch.int32, 0), z(torch.int8, 0), z(torch.int64, 0))

    @staticmethod
    def cat(parts: List["PackedGames"]):
        import torch
        parts = [p for p in parts if p is not None]
        if not parts:
            return PackedGames.empty()
        out = {k: torch.cat([getattr(p, k) for p in parts]) for k in _PER_POS + _PER_GAME if k != "game_index"}
        offs, acc = [], 0
        for p in parts:
            offs.append(p.game_index + acc)
            acc += p.n_games
        out["game_index"] = torch.cat(offs)
        return PackedGames(**out)

    def offset_ids(self, k: int):
        self.ids = self.ids + int(k)
        return self

    def sorted_by_id(self):
        """Games in ascending id order (positions stay grouped by game, plies in order)."""
        import torch
        order = torch.argsort(self.ids, stable=True)
        rank = torch.empty_like(order)
        rank[order] = torch.arange(order.numel(), device=order.device)
        new_gi = rank[self.game_index.long()]
        pos_order = torch.argsort(new_gi, stable=True)      # stable: plies keep their order inside a game
        out = {k: getattr(self, k)[pos_order] for k in _PER_POS if k != "game_index"}
        out["game_index"] = new_gi[pos_order].to(self.game_index.dtype)
        out.update({k: getattr(self, k)[order] for k in _PER_GAME})
        return PackedGames(**out)

    # -- the reference's object form ---------------------------------------------------------
    def to_game_data(self):
        """List[GameData] (training_game.py:42-67).  Values/policies carry float32 precision here (the
        training tensors are float32 anyway, data.py:92-103); c4_drain_games keeps float64."""
        from .board import Board
        from .training_game import GameData
        p = self.cpu()
        boards = p.boards.numpy().view(np.uint64)
        moves, values, policy = p.moves.numpy(), p.values.numpy().astype(np.float64), p.policy.numpy().astype(np.float64)
        games, i = [], 0
        for length, res, gid in zip(p.lengths.tolist(), p.results.tolist(), p.ids.tolist()):
            g = GameData()
            g.game_id = int(gid)
            for j in range(i, i + length):
                v = values[j]
                g.add_move(Board.from_bits(int(boards[j, 0]), int(boards[j, 1])), int(moves[j]),
                           None if np.isnan(v) else float(v), policy[j].copy())
            i += length
            g.result = RESULT_FROM_CODE[int(res)]
            games.append(g)
        return games

    @staticmethod
    def from_game_data(games, id_offset: int = 0):
        import torch
        n = sum(len(g.moves) for g in games)
        boards = np.zeros((n, 2), dtype=np.uint64)
        moves = np.zeros(n, dtype=np.uint8)
        values = np.zeros(n, dtype=np.float32)
        policy = np.zeros((n, 7), dtype=np.float32)
        targets = np.zeros(n, dtype=np.float32)
        gidx = np.zeros(n, dtype=np.int32)
        i = 0
        for k, g in enumerate(games):
            for b, m, v, pr in zip(g.boards, g.moves, g.values, g.priors):
                boards[i] = (b.color[0], b.color[1])
                moves[i] = m
                values[i] = np.nan if v is None else v
                policy[i] = pr
                targets[i] = g.result.value
                gidx[i] = k
                i += 1
        t = torch.from_numpy
        return PackedGames(t(boards.view(np.int64)), t(moves), t(values), t(policy), t(targets), t(gidx),
                           t(np.array([len(g.moves) for g in games], dtype=np.int32)),
                           t(np.array([CODE_FROM_RESULT[g.result] for g in games], dtype=np.int8)),
                           t(np.array([g.game_id + id_offset for g in games], dtype=np.int64)))

    # -- training tensors --------------------------------------------------------------------
    def training_tensors(self, add_fliplr: bool = True):
        """(boards F32[m,3,6,7], values F32[m], priors F32[m,7]) = native_to_pytorch(...) of data.py:78-105, built
        on the device by the engine's kernel.  The tensors must live on a GPU: there is no CPU fallback."""
        from . import engine as _engine
        if self.boards.device.type != "cuda":
            raise RuntimeError("PackedGames.training_tensors runs on the GPU (c4_training_tensors_dev); move the games "
                               "to a cuda device first -- there is no CPU fallback")
        return _engine.training_tensors(self.boards, self.targets, self.policy, add_fliplr)


def all_gather_packed(p: PackedGames) -> PackedGames:
    """All-gather every rank's games as tensors on the device they already live on (cuda tensors over
    RCCL / xGMI with backend nccl, CPU tensors with gloo): sizes first, then one padded all_gather per
    field.  The only collective of a generation (training.py:131 games.extend over the pool results)."""
    import torch
    import torch.distributed as dist
    world = dist.get_world_size()
    dev = p.boards.device
    sizes = torch.tensor([p.n_positions, p.n_games], dtype=torch.int64, device=dev)
    all_sizes = [torch.zeros_like(sizes) for _ in range(world)]
    dist.all_gather(all_sizes, sizes)
    all_sizes = [[int(x) for x in s.tolist()] for s in all_sizes]
    max_pos = max(s[0] for s in all_sizes)
    max_games = max(s[1] for s in all_sizes)
    parts = [dict() for _ in range(world)]
    for key in _PER_POS + _PER_GAME:
        src = getattr(p, key)
        cap = max_pos if key in _PER_POS else max_games
        pad = torch.zeros((cap,) + tuple(src.shape[1:]), dtype=src.dtype, device=dev)
        pad[:src.shape[0]] = src
        outs = [torch.zeros_like(pad) for _ in range(world)]
        dist.all_gather(outs, pad)
        for r in range(world):
            parts[r][key] = outs[r][:all_sizes[r][0] if key in _PER_POS else all_sizes[r][1]]
    return PackedGames.cat([PackedGames(**d) for d in parts])
