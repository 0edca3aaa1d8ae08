"""GPU parity: device bitboard code (through the C ABI) vs the oracle and the golden fixtures.
Bar: bit-exact (integer work)."""
import numpy as np
import pytest

from conftest import load_json

pytestmark = pytest.mark.gpu

RES = {None: -1, 0.0: 0, 0.5: 1, 1.0: 2}


def mask_of(valid):
    m = 0
    for c in valid:
        m |= 1 << c
    return m


@pytest.fixture(scope="module")
def eng():
    from connect4_amd import engine
    return engine


def test_golden_playouts(eng):
    data = load_json("board.json")
    b0, b1, cols, e0, e1, eres, emask, eflip0, eflip1, ecentre = ([] for _ in range(10))
    planes_in, planes_exp = [], []
    for po in data["playouts"]:
        p0 = p1 = 0
        for mv, st in zip(po["moves"], po["states"]):
            b0.append(p0); b1.append(p1); cols.append(mv)
            e0.append(st["c0"]); e1.append(st["c1"]); eres.append(RES[st["result"]])
            emask.append(mask_of(st["valid"]))
            eflip0.append(st["flip"][0]); eflip1.append(st["flip"][1]); ecentre.append(st["centre"])
            if st["planes"] is not None:
                planes_in.append((st["c0"], st["c1"])); planes_exp.append(st["planes"])
            p0, p1 = st["c0"], st["c1"]
    o0, o1, res = eng.board_make_move(b0, b1, cols)
    assert o0.tolist() == e0 and o1.tolist() == e1 and res.tolist() == eres
    assert eng.board_valid_mask(e0, e1).tolist() == emask
    f0, f1 = eng.board_fliplr(e0, e1)
    assert f0.tolist() == eflip0 and f1.tolist() == eflip1
    assert eng.board_centre_value(e0, e1).tolist() == ecentre
    pl = eng.board_planes([p[0] for p in planes_in], [p[1] for p in planes_in])
    assert pl.reshape(len(planes_in), -1).astype(int).tolist() == planes_exp


def test_reference_test_positions(eng, oracle):
    ref = load_json("ref_tests.json")
    boards = [oracle.from_pieces(c["o"], c["x"]) for c in ref["check_valid"]]
    c0 = [int(b.color[0]) for b in boards]
    c1 = [int(b.color[1]) for b in boards]
    # result of a position = o wins / x wins / draw / undecided (board.py:56-62)
    wo, wx = eng.board_wins(c0), eng.board_wins(c1)
    for i, case in enumerate(ref["check_valid"]):
        full = bin(c0[i] | c1[i]).count("1") == 42
        got = 1.0 if wo[i] else (0.0 if wx[i] else (0.5 if full else None))
        assert got == case["ans"]
    vb = [oracle.from_pieces(c["o"], c["x"]) for c in ref["valid_moves"]]
    masks = eng.board_valid_mask([int(b.color[0]) for b in vb], [int(b.color[1]) for b in vb])
    assert masks.tolist() == [mask_of(c["valid"]) for c in ref["valid_moves"]]


def test_random_playouts_vs_oracle(eng, oracle):
    """Full-size property run: 20k random games, every intermediate position checked."""
    rng = np.random.RandomState(5)
    n = 20000
    cur = [oracle.Board.empty() for _ in range(n)]
    alive = list(range(n))
    while alive:
        c0 = [int(cur[i].color[0]) for i in alive]
        c1 = [int(cur[i].color[1]) for i in alive]
        masks = eng.board_valid_mask(c0, c1)
        cols = []
        for j, i in enumerate(alive):
            assert masks[j] == cur[i].valid_mask()
            legal = [c for c in range(7) if (masks[j] >> c) & 1]
            cols.append(int(rng.choice(legal)))
        o0, o1, res = eng.board_make_move(c0, c1, cols)
        nxt = []
        for j, i in enumerate(alive):
            r = cur[i].make_move(cols[j])
            assert (int(o0[j]), int(o1[j]), int(res[j])) == (int(cur[i].color[0]), int(cur[i].color[1]), r)
            if r == -1:
                nxt.append(i)
        alive = nxt


def test_bad_arguments(eng):
    from connect4_amd._lib import EngineError
    with pytest.raises(EngineError):
        eng.board_make_move([0], [0], [9])
