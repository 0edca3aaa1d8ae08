"""Read-only view of a finished search with the reference's Tree surface
(oinkoink/tree.py:61-117): callers use ``tree.get_values_policy()`` (training_game.py:14),
``tree.get_visit_count_policy()`` (game.py:35) and, in tests/tools, the root children's
``name`` / ``data.search_value.visit_count`` / ``data.absolute_value``.  Backed by one
c4_root_result copied from the device; the tree itself stays in HBM."""
import math

import numpy as np

from .board import Board
from .utils import RESULT_FROM_CODE, Side, value_to_side


class _Search:
    def __init__(self, n, w):
        self.visit_count = int(n)
        self.value_sum = float(w)

    def __float__(self):
        return self.value_sum / self.visit_count


class _ChildData:
    def __init__(self, board, status, n, w):
        self.board = board
        self.search_value = _Search(n, w) if n > 0 else None
        self._status = status

    @property
    def absolute_value(self):          # tree.py:27-38
        if self._status >= 0:
            return 0.5 * self._status
        if self.search_value is not None:
            return float(self.search_value)
        return None

    def value(self, side):             # tree.py:40-44
        v = self.absolute_value
        return 0.0 if v is None else value_to_side(v, side)


class _Node:
    def __init__(self, name, data, parent=None):
        self.name = name
        self.data = data
        self.parent = parent
        self.children = ()

    @property
    def is_root(self):
        return self.parent is None

    def __gt__(self, other):           # tree.py:11-15
        return self.name > other.name


class Tree:
    def __init__(self, root_result, board: Board):
        r = root_result
        self.side = Side(board.age % 2)
        rb = Board.from_bits(int(r.color0), int(r.color1))
        self.root = _Node("root", _ChildData(rb, -1, r.root_visits, r.root_value_sum))
        kids = []
        for m in range(7):
            st = r.child_status[m]
            if st == -2:
                continue
            cb = rb.__copy__()
            cb.make_move(m)
            kids.append(_Node(m, _ChildData(cb, st, r.child_visits[m], r.child_value_sum[m]), self.root))
        self.root.children = tuple(kids)
        self.root_prior = np.array(list(r.root_prior), dtype=np.float64)
        self._values_policy = np.array(list(r.values_policy), dtype=np.float64)
        self.expansions = int(r.expansions)
        self.simulations = int(r.simulations)

    def get_node_value(self, node):
        return node.data.value(self.side)

    def best_move(self):
        return max(((self.get_node_value(c), c) for c in self.root.children))[1]

    def most_visited(self):
        return max(((c.data.search_value.visit_count if c.data.search_value else 0, c)
                    for c in self.root.children))[1]

    def get_values_policy(self):
        """Computed on the device by the move-choice code (tree.py:104-109,139-147)."""
        return self._values_policy.copy()

    def get_visit_count_policy(self):
        p = np.zeros(7)
        for c in self.root.children:
            if c.data.search_value is not None:
                p[c.name] = c.data.search_value.visit_count
        s = p.sum()
        if s == 0.0:
            for c in self.root.children:
                p[c.name] = 1.0
            p /= len(self.root.children)
        else:
            p /= s
        return p

    def child(self, move):
        for c in self.root.children:
            if c.name == move:
                return c
        raise KeyError(move)
