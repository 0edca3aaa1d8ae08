"""Evaluator protocol of the reference (oinkoink/evaluators.py:9-63): a callable
``board -> (value in [0,1] from o's side, prior[7])``.

The engine recognises three kinds (connect4_amd/mcts.py):
  * ``evaluate_centre_with_prior`` -> computed inside the HIP kernel (C4_EVAL_CENTRE), no round trip;
  * ``DeviceNetEvaluator``         -> a device-resident net evaluates the whole leaf batch on the
                                      GPU (C4_EVAL_EXTERNAL_F32), nothing leaves the device;
  * any other Python callable      -> leaves are shipped to the host one batch per step and the
                                      callable is applied per board (slow path, full generality).
"""
from copy import deepcopy
from typing import Callable, Dict, Optional, Tuple

import numpy as np

from .board import Board
from .utils import Connect4Stats as info

_COL_W = np.array([0, 1, 2, 3, 2, 1, 0], dtype=float)
_ROW_W = np.array([0, 1, 2, 2, 1, 0], dtype=float)
value_grid = _ROW_W[:, None] + _COL_W[None, :]          # evaluators.py:48-58
value_grid_sum = float(value_grid.sum())                 # 96
prior = np.ones((info.width,), dtype=float) / info.width


class Evaluator:
    """Memoising wrapper keyed by (color0, color1) (evaluators.py:9-25)."""

    def __init__(self, evaluate_fn: Callable, position_table: Optional[Dict[Tuple, Tuple]] = None,
                 store_position: Optional[bool] = True):
        self.evaluate_fn = evaluate_fn
        self.position_table = {} if position_table is None else position_table
        self.store_position = store_position

    def __call__(self, board: Board):
        key = board.to_int_tuple()
        hit = self.position_table.get(key)
        if hit is None:
            hit = self.evaluate_fn(board)
            if self.store_position:
                self.position_table[key] = hit
        return deepcopy(hit)


def evaluate_centre(board: Board):
    o, x = board.pieces
    return 0.5 + (float((o * value_grid).sum()) - float((x * value_grid).sum())) / value_grid_sum


def evaluate_centre_with_prior(board: Board):
    return evaluate_centre(board), prior


def evaluate_nn(board: Board, model):
    value, p = model(board)
    return float(value), p


class DeviceNetEvaluator:
    """Marks a device-resident net (planes [n,3,6,7] -> values [n], priors [n,7], e.g.
    connect4_amd.net.InferenceNet) so searches evaluate leaf batches without leaving the GPU.
    Also callable on a single host Board like any evaluator (model.py:252-267 shape)."""

    def __init__(self, net, device=0):
        self.net = net
        self.device = device

    def __call__(self, board: Board):
        import torch
        if getattr(self.net, "from_bitboards", False):
            v, p = self.net.evaluate_bits([board.color[0]], [board.color[1]])
            return float(v[0]), p[0]
        x = torch.from_numpy(board.to_array().astype(np.float32)).unsqueeze(0).to("cuda:%d" % self.device)
        v, p = self.net(x)
        return float(v[0]), p[0].float().cpu().numpy()


def unwrap(evaluator):
    """The function an Evaluator ultimately calls (through functools.partial)."""
    fn = evaluator.evaluate_fn if isinstance(evaluator, Evaluator) else evaluator
    return getattr(fn, "func", fn)
