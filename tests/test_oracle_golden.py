"""Pins the CPU oracle (oracle/c4_oracle.c) against
  (a) the known-answer data held by the reference's own tests (tests/golden/ref_tests.json, exported
      from /root/reference/tests/board_test.py:10-161,164-247 and tests/player_test.py:13-118), and
  (b) outputs of the unmodified reference captured by tests/golden/gen_golden.py.
Bar: bit-exact for boards / visit counts / float64 value sums; net-driven searches are replayed
through the captured float32 position table so they are bit-exact too.
"""
import json

import numpy as np
import pytest

from conftest import load_json, load_npz, table_from_npz

RES = {None: -1, 0.0: 0, 0.5: 1, 1.0: 2}


def mask_of(valid):
    m = 0
    for c in valid:
        m |= 1 << c
    return m


def test_ref_check_valid(oracle):
    for case in load_json("ref_tests.json")["check_valid"]:
        b = oracle.from_pieces(case["o"], case["x"])
        assert b.result == RES[case["ans"]]


def test_ref_valid_moves(oracle):
    cases = load_json("ref_tests.json")["valid_moves"]
    assert len(cases) == 4
    for case in cases:
        b = oracle.from_pieces(case["o"], case["x"])
        assert b.valid_mask() == mask_of(case["valid"])


def test_board_playouts(oracle):
    data = load_json("board.json")
    for po in data["playouts"]:
        b = oracle.Board.empty()
        for mv, st in zip(po["moves"], po["states"]):
            b.make_move(mv)
            assert (int(b.color[0]), int(b.color[1]), b.age) == (st["c0"], st["c1"], st["age"])
            assert b.result == RES[st["result"]]
            assert b.valid_mask() == mask_of(st["valid"])
            assert [oracle.flip_color(st["c0"]), oracle.flip_color(st["c1"])] == st["flip"]
            assert oracle.evaluate_centre(b) == st["centre"]
            if st["planes"] is not None:
                assert b.planes().reshape(-1).tolist() == st["planes"]
            # from_bits must agree with incremental play
            b2 = oracle.Board.from_bits(st["c0"], st["c1"])
            assert (b2.age, b2.result) == (b.age, b.result)
    for plies, expect in data["ips"].items():
        got = oracle.make_random_ips(int(plies))
        assert [list(p) for p in got] == expect


def check_search(oracle, case, evaluator):
    c = case["config"]
    cfg = oracle.make_config(**c)
    b = oracle.Board.from_bits(case["board"]["c0"], case["board"]["c1"])
    assert b.age == case["board"]["age"]
    info, mv, av = oracle.search_and_pick(cfg, b, evaluator, case["noise"], -1.0)
    assert info.root_visits == case["root_N"]
    assert info.root_value_sum == case["root_W"]
    assert list(info.child_visits) == case["N"], case["name"]
    assert list(info.child_value_sum) == case["W"], case["name"]
    assert list(info.child_status) == case["status"]
    assert list(info.child_value) == case["child_value"]
    assert list(info.values_policy) == case["values_policy"]
    assert list(info.visit_policy) == case["visit_policy"]
    assert list(info.root_prior) == case["root_prior"]
    assert info.best_move == case["best_move"]
    assert info.n_expansions == case["n_expansions"]
    assert info.n_nodes == case["n_nodes"]
    if "accepted_moves" in case:
        assert mv in case["accepted_moves"]


def test_search_centre(oracle):
    ev = oracle.CentreEvaluator()
    cases = load_json("search_centre.json")
    assert len(cases) >= 40
    for case in cases:
        check_search(oracle, case, ev)


def test_search_net_table(oracle):
    npz = load_npz("search_net_tables.npz")
    for case in load_json("search_net.json"):
        ev = oracle.TableEvaluator(*table_from_npz(npz, case["name"]), prior_f32=True)
        check_search(oracle, case, ev)
        assert ev.table.misses == 0


def check_game(oracle, g, evaluator):
    cfg = oracle.make_config(**g["config"])
    n = len(g["moves"])
    noise = np.zeros((42, 7))
    noise[:n] = np.array(g["noise_tape"])
    u = np.full(42, -1.0)
    u[:len(g["uniforms"])] = g["uniforms"]      # one uniform per sampled ply (age < 6)
    out = oracle.selfplay_game(cfg, evaluator, noise, u)
    assert out["moves"] == g["moves"]
    assert out["boards"] == [tuple(b) for b in g["boards"]]
    assert out["result"] == RES[g["result"]]
    for a, b in zip(out["values"], g["values"]):
        assert (b is None and np.isnan(a)) or a == b
    assert out["policies"] == g["policies"]


def test_selfplay_centre(oracle):
    for g in load_json("selfplay.json"):
        check_game(oracle, g, oracle.CentreEvaluator())


def test_selfplay_net_table(oracle):
    npz = load_npz("selfplay_net_tables.npz")
    for g in load_json("selfplay_net.json"):
        ev = oracle.TableEvaluator(*table_from_npz(npz, g["name"]), prior_f32=True)
        check_game(oracle, g, ev)


def test_replay_pool_plays_the_golden_selfplay_games(oracle):
    """The lock-step replay pool (many tape-driven games behind one memoising evaluator table: what the full-size GPU
    parity tests replay device games with) plays the reference's recorded self-play games: the golden net-driven games,
    all at once, answered from their float32 position tables."""
    npz = load_npz("selfplay_net_tables.npz")
    games = load_json("selfplay_net.json")
    cfgs = {json.dumps(g["config"], sort_keys=True) for g in games}
    for cj in cfgs:
        group = [g for g in games if json.dumps(g["config"], sort_keys=True) == cj]
        cfg = oracle.make_config(**group[0]["config"])
        noise = np.zeros((len(group), 42, 7))
        u = np.full((len(group), 42), -1.0)
        table = {}
        for j, g in enumerate(group):
            noise[j, :len(g["moves"])] = np.array(g["noise_tape"])
            u[j, :len(g["uniforms"])] = g["uniforms"]
            c0s, c1s, vs, ps = table_from_npz(npz, g["name"])
            for a, b, v, p in zip(c0s, c1s, vs, ps):
                table[(int(a), int(b))] = (np.float32(v), np.asarray(p, dtype=np.float32))
        pool = oracle.ReplayPool(cfg, len(group), noise, u)
        while True:
            m = pool.collect()
            if m == 0:
                break
            ans = [table[(int(pool.c0[k]), int(pool.c1[k]))] for k in range(m)]
            pool.apply(m, np.array([a[0] for a in ans], dtype=np.float32), np.stack([a[1] for a in ans]))
        for j, g in enumerate(group):
            out = pool.game(j)
            assert out["moves"] == g["moves"] and out["boards"] == [tuple(b) for b in g["boards"]]
            assert out["result"] == RES[g["result"]] and out["policies"] == g["policies"]
            for a, b in zip(out["values"], g["values"]):
                assert (b is None and np.isnan(a)) or a == b
        st = pool.stats()
        assert st["hits"] > 0 and st["table_entries"] <= len(table)
        pool.close()
