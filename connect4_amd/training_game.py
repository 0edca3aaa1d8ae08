"""GameData / TrainingData with the reference's attribute names
(oinkoink/neural/training_game.py:22-67), filled from the engine's finished-game records."""
import math
from typing import List, Sequence

import numpy as np

from .board import Board
from .utils import RESULT_FROM_CODE


class TrainingData:
    def __init__(self, boards: List[Board], values: List[float], priors: List[Sequence[float]]):
        self.boards = boards
        self.values = values
        self.priors = priors

    def __add__(self, other):
        return TrainingData(self.boards + other.boards, self.values + other.values, self.priors + other.priors)

    def __radd__(self, other):   # lets sum()/np.sum start from 0 (data.py:57 np.sum(list(...)))
        return self if other == 0 else NotImplemented


class GameData:
    def __init__(self):
        self.result = None
        self.moves = []
        self.boards = []
        self.values = []
        self.priors = []
        self.game_id = -1

    def add_move(self, board, move, value, prior):
        self.moves.append(move)
        self.boards.append(board)
        self.values.append(value)
        self.priors.append(prior)

    def create_training_values(self):
        return [self.result.value] * len(self.values)   # training_game.py:57-60

    @property
    def data(self):
        assert self.result is not None
        return TrainingData(self.boards, self.create_training_values(), self.priors)


def game_data_from_record(rec) -> GameData:
    """c4_game_record (include/c4_engine.h) -> GameData."""
    gd = GameData()
    gd.game_id = int(rec.game_id)
    for i in range(rec.length):
        v = rec.value[i]
        gd.add_move(Board.from_bits(int(rec.color0[i]), int(rec.color1[i])), int(rec.move[i]),
                    None if math.isnan(v) else float(v), np.array(rec.policy[i], dtype=np.float64))
    gd.result = RESULT_FROM_CODE[int(rec.result)]
    return gd


def training_game(player) -> GameData:
    """One self-play game through the player protocol (training_game.py:8-19)."""
    board = Board()
    gd = GameData()
    while board.result is None:
        before = board.__copy__()
        move, value, tree = player.make_move(board)
        gd.add_move(before, move, value, tree.get_values_policy())
    gd.result = board.result
    return gd
