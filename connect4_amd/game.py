"""Two-player game loop (oinkoink/game.py:20-40)."""
import numpy as np

from .board import Board
from .utils import Side


class Game:
    def __init__(self, display, player_o, player_x, board: Board):
        self.display = display
        self._player_o = player_o
        self._player_x = player_x
        self._board = board
        self.move_history = np.empty((0,), dtype="uint8")

    def play(self):
        while self._board.result is None:
            player = self._player_o if self._board.player_to_move == Side.o else self._player_x
            move, value, tree = player.make_move(self._board)
            if self.display:
                print("{} selected move: {}, value: {}".format(player.name, move, value))
                print(self._board)
            self.move_history = np.append(self.move_history, move)
        return self._board.result
