"""Host-side Board: the mutable position object callers hand to ``player.make_move(board)``.

Same public surface as the reference's ``oinkoink.board.Board`` (board.py:35-222) so existing
callers (game.py:20-40, training_game.py:8-19, match.py) work unchanged, but kept as two Python
ints -- it is only the *boundary object*.  All search-time board work (make-move / legal moves /
win check for millions of nodes) happens in the HIP kernels (connect4_amd/csrc/c4_board.h); bulk
host-side conversions go through the device functions in connect4_amd.engine.
"""
from typing import Iterable, Set

import numpy as np

from .utils import Connect4Stats as info
from .utils import Result, Side

WIDTH, HEIGHT = info.width, info.height
H1 = HEIGHT + 1
SIZE = WIDTH * HEIGHT
COL1 = (1 << H1) - 1
BOTTOM = sum(1 << (H1 * c) for c in range(WIDTH))
TOP = BOTTOM << HEIGHT
ALL1 = (1 << (H1 * WIDTH)) - 1


def _wins(b: int) -> bool:
    for s in (HEIGHT, H1, H1 + 1, 1):      # diag \, horizontal, diag /, vertical (board.py:173-184)
        y = b & (b >> s)
        if y & (y >> (2 * s)):
            return True
    return False


def _bit(row_from_top: int, col: int) -> int:
    return 1 << (col * H1 + (HEIGHT - 1 - row_from_top))


class _Heights:
    """height[c] = 7*c + stones in column c (board.py:39), derived from the bitboards."""

    def __init__(self, board):
        self._b = board

    def __getitem__(self, col):
        occ = self._b.color[0] | self._b.color[1]
        return H1 * col + bin((occ >> (H1 * col)) & COL1).count("1")

    def __iter__(self):
        return (self[c] for c in range(WIDTH))

    def __len__(self):
        return WIDTH

    def __array__(self, dtype=None, copy=None):
        return np.array(list(self), dtype=dtype or np.int64)


class Board:
    __slots__ = ("color", "age", "result")

    def __init__(self):
        self.color = [0, 0]
        self.age = 0
        self.result = None

    # -- constructors ------------------------------------------------------------------------
    @classmethod
    def from_bits(cls, color0: int, color1: int):
        b = cls()
        b.color = [int(color0), int(color1)]
        b.age = bin(b.color[0]).count("1") + bin(b.color[1]).count("1")
        b._derive_result()
        return b

    @classmethod
    def from_pieces(cls, o_pieces, x_pieces):
        """[6][7] boolean arrays, row 0 = top (board.py:44-62)."""
        o = np.asarray(o_pieces, dtype=bool)
        x = np.asarray(x_pieces, dtype=bool)
        c0 = sum(_bit(r, c) for r in range(HEIGHT) for c in range(WIDTH) if o[r, c])
        c1 = sum(_bit(r, c) for r in range(HEIGHT) for c in range(WIDTH) if x[r, c])
        return cls.from_bits(c0, c1)

    def _derive_result(self):
        if _wins(self.color[0]):
            self.result = Result.o_win
        elif _wins(self.color[1]):
            self.result = Result.x_win
        elif self.age == SIZE:
            self.result = Result.draw
        else:
            self.result = None

    # -- views -------------------------------------------------------------------------------
    @property
    def height(self):
        return _Heights(self)

    def _pieces_of(self, stones):
        return np.array([[bool(stones & _bit(r, c)) for c in range(WIDTH)] for r in range(HEIGHT)], dtype=np.bool_)

    @property
    def o_pieces(self):
        return self._pieces_of(self.color[0])

    @property
    def x_pieces(self):
        return self._pieces_of(self.color[1])

    @property
    def pieces(self):
        return self.o_pieces, self.x_pieces

    @property
    def player_to_move(self):
        return Side(self.age % 2)

    @property
    def valid_moves(self) -> Set[int]:
        if self.result is not None:
            return set()
        return {c for c in range(WIDTH) if self._isplayable(c)}

    def _isplayable(self, col):
        return ((1 << self.height[col]) & TOP) == 0     # board.py:187-188

    @property
    def symmetrical(self):
        return self.flip_color(self.color[0]) == self.color[0] and self.flip_color(self.color[1]) == self.color[1]

    def is_symmetrical(self, pieces):
        return self.flip_color(pieces) == pieces

    # -- transforms --------------------------------------------------------------------------
    @staticmethod
    def flip_color(pieces: int) -> int:
        out = 0
        for c in range(WIDTH):
            out |= ((pieces >> (H1 * c)) & COL1) << (H1 * (WIDTH - 1 - c))
        return out

    def create_fliplr(self):
        b = self.__class__()
        b.color = [self.flip_color(self.color[0]), self.flip_color(self.color[1])]
        b.age = self.age
        b.result = self.result
        return b

    def to_array(self):
        """uint8 [3][6][7]: to-move plane, o stones, x stones (board.py:147-154)."""
        to_move = np.full((HEIGHT, WIDTH), 1 if self.age % 2 == 0 else 0, dtype=np.uint8)
        return np.stack([to_move, self.o_pieces.astype(np.uint8), self.x_pieces.astype(np.uint8)])

    def to_int_tuple(self):
        return self.color[0], self.color[1]

    # -- play --------------------------------------------------------------------------------
    def make_move(self, move: int):
        side = self.age & 1
        self.color[side] ^= 1 << self.height[move]
        won = _wins(self.color[side])
        self.age += 1
        if won:
            self.result = Result(self.age % 2)       # o moved => age odd => 1.0 (board.py:166-167)
        elif self.age == SIZE:
            self.result = Result.draw
        return self.result

    def __copy__(self):
        b = self.__class__()
        b.color = list(self.color)
        b.age = self.age
        b.result = self.result
        return b

    copy = __copy__

    def __eq__(self, other):
        return isinstance(other, Board) and other.color == self.color

    def __hash__(self):
        return hash((self.color[0], self.color[1]))

    def __str__(self):
        rows = []
        for r in range(HEIGHT):
            rows.append(" ".join("o" if self.color[0] & _bit(r, c) else "x" if self.color[1] & _bit(r, c) else "-"
                                 for c in range(WIDTH)))
        hdr = " ".join(str(c) for c in range(WIDTH))
        return "\n".join([hdr] + rows + [hdr])

    def __repr__(self):
        return "color: {}, age: {}, result: {}\n{}".format(self.color, self.age, self.result, self)


def expand(ips: Set, board: Board, plies: int) -> None:
    if plies == 0:
        if board.result is None:
            ips.add(board)
        return
    for move in sorted(board.valid_moves):
        nb = board.__copy__()
        nb.make_move(move)
        expand(ips, nb, plies - 1)


def make_random_ips(plies: int) -> Set[Board]:
    """All undecided positions `plies` deep (board.py:225-243)."""
    ips: Set[Board] = set()
    expand(ips, Board(), plies)
    return ips


def boards_to_bits(boards: Iterable[Board]):
    c0 = np.array([b.color[0] for b in boards], dtype=np.uint64)
    c1 = np.array([b.color[1] for b in boards], dtype=np.uint64)
    return c0, c1
