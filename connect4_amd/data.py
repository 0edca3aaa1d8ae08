"""Training-data writer compatible with the existing loop.

``games_to_tensors`` reproduces ``native_to_pytorch(..., add_fliplr=True)``
(oinkoink/neural/pytorch/data.py:78-105): boards F32[N,3,6,7], values F32[N], priors F32[N,7] with the
left-right mirrored copies appended after the originals (boards mirrored by column,
board.py:115-145; priors reversed; values duplicated).  Plane encoding and mirroring run through
the engine's device functions (c4_board_planes / c4_board_fliplr).  ``save_generation`` writes the
``data.pth`` dict the reference's Connect4Dataset.load reads (data.py:22-33, 52-64).
"""
import os
from typing import List

import numpy as np

from . import engine as _engine
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


def games_to_tensors(games: List[GameData], add_fliplr: bool = True, device: int = 0):
    import torch
    b, v, p = games_to_arrays(games, add_fliplr, device)
    return torch.from_numpy(b), torch.from_numpy(v), torch.from_numpy(np.ascontiguousarray(p))


def save_generation(games: List[GameData], folder_path: str, device: int = 0):
    """<folder>/data.pth = {'boards','values','priors'} (data.py:22-28)."""
    import torch
    b, v, p = games_to_tensors(games, True, device)
    os.makedirs(folder_path, exist_ok=True)
    torch.save({"boards": b, "values": v, "priors": p}, os.path.join(folder_path, "data.pth"))
    return len(b)
