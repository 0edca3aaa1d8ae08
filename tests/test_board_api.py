"""Host Board mirror vs the reference's own board tests (tests/board_test.py:152-247, exported as
data in tests/golden/ref_tests.json) and vs reference playouts.  CPU only."""
import copy

import numpy as np

from conftest import load_json
from connect4_amd.board import Board, make_random_ips
from connect4_amd.evaluators import evaluate_centre
from connect4_amd.utils import Result


def test_check_valid():
    for case in load_json("ref_tests.json")["check_valid"]:
        b = Board.from_pieces(np.array(case["o"], dtype=np.bool_), np.array(case["x"], dtype=np.bool_))
        assert b.result == (Result(case["ans"]) if case["ans"] is not None else None)


def test_valid_moves():
    for case in load_json("ref_tests.json")["valid_moves"]:
        b = Board.from_pieces(np.array(case["o"], dtype=np.bool_), np.array(case["x"], dtype=np.bool_))
        assert b.valid_moves == set(case["valid"])


def test_playouts_and_views():
    data = load_json("board.json")
    for po in data["playouts"][:60]:
        b = Board()
        for mv, st in zip(po["moves"], po["states"]):
            before = copy.copy(b)
            b.make_move(mv)
            assert before != b and before.age == b.age - 1      # copy is independent
            assert (b.color[0], b.color[1], b.age) == (st["c0"], st["c1"], st["age"])
            assert (None if b.result is None else b.result.value) == st["result"]
            assert sorted(b.valid_moves) == st["valid"]
            fl = b.create_fliplr()
            assert [fl.color[0], fl.color[1]] == st["flip"]
            assert evaluate_centre(b) == st["centre"]
            if st["planes"] is not None:
                assert b.to_array().reshape(-1).tolist() == st["planes"]
            assert hash(b) == hash(Board.from_bits(*b.to_int_tuple())) and b == Board.from_bits(*b.to_int_tuple())
            assert list(b.height) == [7 * c + bin(((b.color[0] | b.color[1]) >> (7 * c)) & 0x7f).count("1") for c in range(7)]
    for plies, expect in data["ips"].items():
        got = sorted([bb.color[0], bb.color[1]] for bb in make_random_ips(int(plies)))
        assert got == expect
