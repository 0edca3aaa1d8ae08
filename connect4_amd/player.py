"""Player protocol of the reference (oinkoink/player.py:7-15): ``make_move(board)`` mutates the
caller's board and returns ``(move, value, tree)``.  HumanPlayer (interactive stdin) is out of scope."""


class BasePlayer:
    def __init__(self, name):
        self.name = name

    def __str__(self):
        return "Player: " + self.name

    def make_move(self, board):
        raise NotImplementedError
