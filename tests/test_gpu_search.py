"""GPU parity proper: the HIP MCTS (through the C ABI) vs golden vectors from the reference and vs
the CPU oracle on seeded inputs.  Bar: visit counts exact, float64 value sums / policies bit-exact
(the engine reproduces the reference's arithmetic operation for operation)."""
import numpy as np
import pytest

from conftest import load_json, load_npz, table_from_npz
from gpu_helpers import drive_external, table_lookup_fn

pytestmark = pytest.mark.gpu


def cfg_key(c):
    return (c["simulations"], c["pb_c_base"], c["pb_c_init"], c["root_dirichlet_alpha"],
            c["root_exploration_fraction"], c["num_sampling_moves"])


def make_engine(c, n_slots, eval_mode, rng_tape=False, **kw):
    from connect4_amd import _lib as L
    from connect4_amd.engine import Engine
    return Engine(n_slots, c["simulations"], c["pb_c_base"], c["pb_c_init"], c["root_dirichlet_alpha"],
                  c["root_exploration_fraction"], c["num_sampling_moves"], eval_mode=eval_mode,
                  rng_mode=L.RNG_TAPE if rng_tape else L.RNG_PHILOX, **kw)


def assert_root_matches(r, case):
    assert r.state == 2
    assert r.root_visits == case["root_N"]
    assert r.root_value_sum == case["root_W"]
    assert list(r.child_visits) == case["N"], case["name"]
    assert list(r.child_value_sum) == case["W"], case["name"]
    assert list(r.child_status) == case["status"]
    assert list(r.values_policy) == case["values_policy"]
    assert list(r.root_prior) == case["root_prior"]
    assert r.move == case["best_move"]
    assert r.expansions == case["n_expansions"]
    assert r.simulations == case["config"]["simulations"]


def tapes_for(cases):
    nz = np.zeros((len(cases), 42, 7))
    u = np.full((len(cases), 42), -1.0)
    for i, c in enumerate(cases):
        if c["noise"] is not None:
            nz[i, 0] = c["noise"]
    return nz, u


def test_search_centre_golden():
    """In-kernel evaluate_centre_with_prior searches vs the reference (float64 throughout)."""
    from connect4_amd import _lib as L
    cases = load_json("search_centre.json")
    groups = {}
    for c in cases:
        groups.setdefault(cfg_key(c["config"]), []).append(c)
    for group in groups.values():
        noisy = group[0]["config"]["root_dirichlet_alpha"] != 0
        with make_engine(group[0]["config"], len(group), L.EVAL_CENTRE, rng_tape=noisy,
                         stop_after_move=True) as eng:
            if noisy:
                eng.set_tapes(*tapes_for(group))
            eng.reset([c["board"]["c0"] for c in group], [c["board"]["c1"] for c in group])
            eng.run_centre()
            roots = eng.read_roots()
            for r, case in zip(roots, group):
                assert_root_matches(r, case)
                if "accepted_moves" in case:
                    assert r.move in case["accepted_moves"]


def test_search_net_table_golden():
    """Net-driven searches replayed through the captured float32 position table
    (EXTERNAL_F32: NumPy>=2 float32 score path; root prior float64 after Dirichlet noise)."""
    from connect4_amd import _lib as L
    npz = load_npz("search_net_tables.npz")
    for case in load_json("search_net.json"):
        fn = table_lookup_fn(*table_from_npz(npz, case["name"]))
        noisy = case["noise"] is not None
        with make_engine(case["config"], 1, L.EVAL_EXTERNAL_F32, rng_tape=noisy, stop_after_move=True) as eng:
            if noisy:
                eng.set_tapes(*tapes_for([case]))
            eng.reset([case["board"]["c0"]], [case["board"]["c1"]])
            drive_external(eng, fn, np.float32)
            assert_root_matches(eng.read_roots()[0], case)


def test_search_external_f64_matches_centre(oracle):
    """EXTERNAL_F64 with a host evaluator (float64 heuristic) == oracle == in-kernel centre mode."""
    from connect4_amd import _lib as L
    rng = np.random.RandomState(3)
    boards = []
    while len(boards) < 16:
        b = oracle.Board.empty()
        for _ in range(int(rng.randint(0, 25))):
            m = b.valid_mask()
            if not m:
                break
            b.make_move(int(rng.choice([c for c in range(7) if (m >> c) & 1])))
        if b.result == -1:
            boards.append(b)
    c = dict(simulations=150, pb_c_base=19652, pb_c_init=1.25, root_dirichlet_alpha=0.0,
             root_exploration_fraction=0.0, num_sampling_moves=0)

    def fn(c0, c1):
        return oracle.evaluate_centre(oracle.Board.from_bits(c0, c1)), np.full(7, 1.0 / 7.0)
    with make_engine(c, len(boards), L.EVAL_EXTERNAL_F64, stop_after_move=True) as eng:
        eng.reset([b.key()[0] for b in boards], [b.key()[1] for b in boards])
        drive_external(eng, fn, np.float64)
        roots = eng.read_roots()
        cfg = oracle.make_config(**c)
        for r, b in zip(roots, boards):
            info, mv, av = oracle.search_and_pick(cfg, b, oracle.CentreEvaluator())
            assert list(r.child_visits) == list(info.child_visits)
            assert list(r.child_value_sum) == list(info.child_value_sum)
            assert list(r.values_policy) == list(info.values_policy)
            assert r.move == mv and r.value == av


def test_search_centre_vs_oracle_many(oracle):
    """2048 seeded random positions x 400 sims: engine == oracle exactly (N, W, policy, move, value,
    expansions).  Covers near-full boards, forced wins, single-legal-move positions."""
    from connect4_amd import _lib as L
    rng = np.random.RandomState(17)
    boards = []
    while len(boards) < 2048:
        b = oracle.Board.empty()
        for _ in range(int(rng.randint(0, 41))):
            m = b.valid_mask()
            if not m:
                break
            b.make_move(int(rng.choice([c for c in range(7) if (m >> c) & 1])))
        if b.result == -1:
            boards.append(b)
    c = dict(simulations=400, pb_c_base=19652, pb_c_init=1.25, root_dirichlet_alpha=0.0,
             root_exploration_fraction=0.0, num_sampling_moves=0)
    cfg = oracle.make_config(**c)
    with make_engine(c, len(boards), L.EVAL_CENTRE, stop_after_move=True) as eng:
        eng.reset([b.key()[0] for b in boards], [b.key()[1] for b in boards])
        eng.run_centre()
        roots = eng.read_roots()
        st = eng.stats()
    exp_total = 0
    for r, b in zip(roots, boards):
        info, mv, av = oracle.search_and_pick(cfg, b, oracle.CentreEvaluator())
        assert r.root_visits == info.root_visits == 401
        assert list(r.child_visits) == list(info.child_visits)
        assert list(r.child_value_sum) == list(info.child_value_sum)
        assert list(r.child_status) == list(info.child_status)
        assert list(r.values_policy) == list(info.values_policy)
        assert r.move == mv
        assert (np.isnan(r.value) and np.isnan(av)) or r.value == av
        assert r.expansions == info.n_expansions
        exp_total += info.n_expansions
    assert st["simulations"] == 400 * len(boards)
    assert st["expansions"] == exp_total
    assert st["moves"] == len(boards)


def check_selfplay(case, eval_mode, fn=None, **engine_kw):
    from connect4_amd import _lib as L
    n = len(case["moves"])
    nz = np.zeros((1, 42, 7))
    nz[0, :n] = np.array(case["noise_tape"])
    u = np.full((1, 42), -1.0)
    u[0, :len(case["uniforms"])] = case["uniforms"]
    engine_kw.setdefault("eval_cache_log2_entries", -1)
    with make_engine(case["config"], 1, eval_mode, rng_tape=True, games_target=1,
                     record_capacity_games=4, **engine_kw) as eng:
        eng.set_tapes(nz, u)
        eng.reset()
        if eval_mode == L.EVAL_CENTRE:
            eng.run_centre(max_launches=200000)
        else:
            drive_external(eng, fn, np.float32)
        games = eng.drain_games()
        st = eng.stats()
    assert len(games) == 1
    g = games[0]
    assert g.length == n
    assert list(g.move[:n]) == case["moves"]
    assert [[int(g.color0[i]), int(g.color1[i])] for i in range(n)] == case["boards"]
    assert g.result == {0.0: 0, 0.5: 1, 1.0: 2}[case["result"]]
    for i in range(n):
        assert (case["values"][i] is None and np.isnan(g.value[i])) or g.value[i] == case["values"][i]
        assert list(g.policy[i]) == case["policies"][i]
    assert st["games_finished"] == 1 and st["moves"] == n
    assert st["simulations"] == n * case["config"]["simulations"]
    return st


def test_selfplay_centre_golden():
    """training_game() with Dirichlet noise + sampled opening moves, RNG injected from the
    reference's recorded tape: moves, values, policies and result identical."""
    from connect4_amd import _lib as L
    for case in load_json("selfplay.json"):
        check_selfplay(case, L.EVAL_CENTRE)


def test_selfplay_net_table_golden():
    from connect4_amd import _lib as L
    npz = load_npz("selfplay_net_tables.npz")
    for case in load_json("selfplay_net.json"):
        fn = table_lookup_fn(*table_from_npz(npz, case["name"]))
        check_selfplay(case, L.EVAL_EXTERNAL_F32, fn)


def test_eval_cache_is_transparent():
    """The device evaluation cache (the reference's Evaluator memo table, evaluators.py:18-25) must not
    change a single visit count: whole self-play games with the cache on (and several cached leaves
    applied per launch) equal the golden games; the new root of every move after the first is a hit."""
    from connect4_amd import _lib as L
    npz = load_npz("selfplay_net_tables.npz")
    for case in load_json("selfplay_net.json"):
        seen = []
        table = table_lookup_fn(*table_from_npz(npz, case["name"]))

        def fn(c0, c1):
            seen.append((int(c0), int(c1)))
            return table(c0, c1)
        st = check_selfplay(case, L.EVAL_EXTERNAL_F32, fn, eval_cache_log2_entries=16, max_inner_iters=4,
                            level_budget=5)
        # ... and with simulations rationed by elapsed cycles instead of by count
        seen.clear()
        check_selfplay(case, L.EVAL_EXTERNAL_F32, fn, eval_cache_log2_entries=16, max_inner_iters=32,
                       time_budget_cycles=20000)
        assert st["eval_cache_hits"] > 0.2 * st["leaf_evals"]
        assert st["eval_cache_hits"] + len(seen) == st["leaf_evals"]
        assert len(seen) - len(set(seen)) <= 0.02 * len(seen)   # re-evaluations only after direct-mapped evictions


def test_level_budget_suspension_is_transparent():
    """A tiny level budget forces most descents to be suspended and resumed across launches; games
    must still equal the golden ones move for move (centre evaluator, RNG tape)."""
    from connect4_amd import _lib as L
    for case in load_json("selfplay.json")[:2]:
        st = check_selfplay(case, L.EVAL_CENTRE, level_budget=3, max_inner_iters=2)
        assert st["capped_slots"] > 0


def test_non_finite_evaluator_answers_are_contained():
    """The reference asserts on NaN net outputs (model.py:258-263).  The engine must stay memory-safe:
    NaN / out-of-range answers are replaced, counted in stats.bad_evals, and the search completes."""
    from connect4_amd import _lib as L
    c = dict(simulations=60, pb_c_base=19652, pb_c_init=1.25, root_dirichlet_alpha=0.0,
             root_exploration_fraction=0.0, num_sampling_moves=0)
    calls = [0]

    def fn(c0, c1):
        calls[0] += 1
        p = np.full(7, 1.0 / 7.0, dtype=np.float32)
        if calls[0] % 3 == 0:
            p[:] = np.nan
        if calls[0] % 5 == 0:
            return np.float32(np.nan), p
        return np.float32(0.5), p
    with make_engine(c, 4, L.EVAL_EXTERNAL_F32, stop_after_move=True) as eng:
        eng.reset()
        drive_external(eng, fn, np.float32)
        st = eng.stats()
        roots = eng.read_roots()
    assert st["bad_evals"] > 0 and st["simulations"] == 4 * 60
    for r in roots:
        assert r.state == 2 and 0 <= r.move < 7 and r.root_visits == 61


def test_written_out_division_is_the_ieee_division():
    """The PUCT score's sqrt(N) / (n + 1) and the backups' W / N run a float64 division written out for normal-range
    operands (no v_div_scale / v_div_fmas / v_div_fixup).  On the device, for every parent / child visit count up to
    32768 x 4096 and 2^27 random value sums, its quotient must have the bits of `a / b`."""
    from connect4_amd.engine import debug_div_mismatches
    assert debug_div_mismatches(32768, 4096, 1 << 27) == 0
