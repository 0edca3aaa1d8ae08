"""Training-data storage compatible with the existing loop.

* ``games_to_tensors`` / ``PackedGames.training_tensors`` reproduce ``native_to_pytorch(...,
  add_fliplr=True)`` (oinkoink/neural/pytorch/data.py:78-105): boards F32[N,3,6,7], values F32[N], priors
  F32[N,7] with the left-right mirrored copies appended after the originals (boards mirrored by column,
  board.py:115-145; priors reversed; values duplicated).  Plane encoding and mirroring run in the engine's
  device code (c4_training_tensors_dev, or c4_board_planes / c4_board_fliplr for host GameData).
* ``TrainingDataStorage`` mirrors the reference's class of that name (data.py:47-75) and its base
  ``GameStorage`` (storage.py:11-22): ``save(games, folder)`` writes ``games.pkl`` + ``data.pth`` (the dict
  Connect4Dataset.load reads, data.py:22-33), ``get_dataset(base, gen)`` returns the sliding window of
  the last ``n = min(20, int((gen + 1) / 2))`` generations (data.py:66-75).
"""
import os
import pickle
from typing import List, Union

import numpy as np

from . import engine as _engine
from .packed import PackedGames
from .training_game import GameData


def games_to_arrays(games: List[GameData], add_fliplr: bool = True, device: int = 0):
    c0 = np.array([b.color[0] for g in games for b in g.boards], dtype=np.uint64)
    c1 = np.array([b.color[1] for g in games for b in g.boards], dtype=np.uint64)
    values = np.array([v for g in games for v in g.create_training_values()], dtype=np.float64)
    priors = np.array([p for g in games for p in g.priors], dtype=np.float64).reshape(-1, 7)
    if add_fliplr:
        f0, f1 = _engine.board_fliplr(c0, c1, device=device)
        c0, c1 = np.concatenate([c0, f0]), np.concatenate([c1, f1])
        values = np.concatenate([values, values])
        priors = np.concatenate([priors, priors[:, ::-1]])
    boards = _engine.board_planes(c0, c1, device=device)
    return boards, values.astype(np.float32), priors.astype(np.float32)


def games_to_tensors(games: Union[List[GameData], PackedGames], add_fliplr: bool = True, device: int = 0):
    """CPU tensors (what data.pth holds).  PackedGames on a GPU are converted there in one kernel."""
    import torch
    if isinstance(games, PackedGames):
        p = games if games.device.type == "cuda" else games.to(torch.device("cuda", device))
        b, v, pr = p.training_tensors(add_fliplr)
        return b.cpu(), v.cpu(), pr.cpu()
    b, v, p = games_to_arrays(games, add_fliplr, device)
    return torch.from_numpy(b), torch.from_numpy(v), torch.from_numpy(np.ascontiguousarray(p))


def save_generation(games: Union[List[GameData], PackedGames], folder_path: str, device: int = 0):
    """<folder>/data.pth = {'boards','values','priors'} (data.py:22-28)."""
    import torch
    b, v, p = games_to_tensors(games, True, device)
    os.makedirs(folder_path, exist_ok=True)
    torch.save({"boards": b, "values": v, "priors": p}, os.path.join(folder_path, "data.pth"))
    return len(b)


def window_generations(gen: int) -> List[int]:
    """data.py:66-72: the generations get_dataset concatenates, newest first."""
    n = min(20, int((gen + 1) / 2))
    return list(range(gen, gen - n, -1))


class GameStorage:
    """storage.py:11-22."""

    last_game = None

    def save(self, games: Union[List[GameData], PackedGames], folder_path: str):
        objs = games.to_game_data() if isinstance(games, PackedGames) else games
        os.makedirs(folder_path, exist_ok=True)
        with open(os.path.join(folder_path, "games.pkl"), "wb") as f:
            pickle.dump(objs, f)
        self.last_game = objs[-1] if objs else None

    def last_game_str(self):
        return game_str(self.last_game.moves, self.last_game.values, self.last_game.priors)


def game_str(moves, values, policies):      # storage.py:25-38
    from .board import Board
    board = Board()
    out = str(board)
    for move, value, policy in zip(moves, values, policies):
        board.make_move(move)
        out += "\nMove: {}  Value: {} Policy: {}\n{}".format(move, value, policy, board)
    return out


def load_games(folder_path: str) -> List[GameData]:
    """Read back a games.pkl this package wrote (a pickle of its own GameData objects; trusted input only)."""
    with open(os.path.join(folder_path, "games.pkl"), "rb") as f:
        return pickle.load(f)


class TrainingDataStorage(GameStorage):
    """data.py:47-75.  `write_games_pkl=False` skips the object form (65,536 games are ~2 M Board objects)."""

    def __init__(self, device: int = 0, write_games_pkl: bool = True):
        self.device = device
        self.write_games_pkl = write_games_pkl

    def td_file_name(self, folder_path, gen):
        return "{}/{}/data.pth".format(folder_path, gen)

    def save(self, games, folder_path: str):
        if self.write_games_pkl:
            super().save(games, folder_path)
        return save_generation(games, folder_path, self.device)

    def get_dataset(self, base_path: str, gen: int):
        """(boards, values, priors) of the window, concatenated in the reference's order (gen, gen-1, ...): the
        tensors its ConcatDataset([Connect4Dataset.load(f) ...]) indexes (data.py:66-75)."""
        import torch
        parts = [torch.load(self.td_file_name(base_path, g), weights_only=True) for g in window_generations(gen)]
        return tuple(torch.cat([p[k] for p in parts]) for k in ("boards", "values", "priors"))
