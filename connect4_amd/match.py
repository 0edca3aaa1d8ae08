"""Strength-evaluation harness: the reference's Match
(oinkoink/match.py:15-70) over all `plies`-deep openings, both colours, scored from player_1's side.

Where the reference plays the games one after another (or in a process Pool, match.py:72-76), all
games here advance in lock-step: at every ply the boards waiting for player_1 form one GPU batch
(``MCTS.make_moves``), likewise for player_2.  Players without ``make_moves`` are called per board.
"""
from copy import copy

import numpy as np

from .board import make_random_ips


def _moves(player, boards):
    if not boards:
        return
    if hasattr(player, "make_moves"):
        player.make_moves(boards)
    else:
        for b in boards:
            player.make_move(b)


class Match:
    def __init__(self, display, player_1, player_2, plies: int = 0, switch: bool = False):
        self._player_1 = player_1
        self._player_2 = player_2
        ips = sorted(make_random_ips(plies), key=lambda b: b.to_int_tuple())
        # (board, player moving o, player moving x); switched games swap the colours (match.py:33-40)
        self.games = [(copy(b), player_1, player_2) for b in ips]
        self.n = len(self.games)
        if switch:
            self.games += [(copy(b), player_2, player_1) for b in ips]
        self.switch = switch
        self.display = display

    def play(self, agents=1):
        boards = [g[0] for g in self.games]
        while True:
            live = [i for i, b in enumerate(boards) if b.result is None]
            if not live:
                break
            for player in (self._player_1, self._player_2):
                todo = [boards[i] for i in live if boards[i].result is None and
                        (self.games[i][1] if boards[i].age % 2 == 0 else self.games[i][2]) is player]
                _moves(player, todo)
        results = np.array([b.result.value for b in boards], dtype="f")
        if self.switch:                       # flip the games where player_2 moved first (match.py:53-56)
            results[self.n:] *= -1.0
            results[self.n:] += 1.0
        wins = int(np.sum(results == 1))
        draws = int(np.sum(results == 0.5))
        losses = int(np.sum(results == 0))
        return_ = (1.0 * wins + 0.5 * draws) / (wins + draws + losses)
        if self.display:
            print("The results for {} vs {} are: {} wins, {} draws, {} losses, {:.3f} return".format(
                self._player_1.name, self._player_2.name, wins, draws, losses, return_))
        return {"wins": wins, "draws": draws, "losses": losses, "return": return_}
