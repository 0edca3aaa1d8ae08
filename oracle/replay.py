"""Replay of device self-play games on the CPU oracle (TEST INFRASTRUCTURE ONLY: tests/ and
__graft_entry__.smoke() import this; connect4_amd/ never does).

The device plays with injected RNG tapes (C4_RNG_TAPE); the oracle replays the same game with the same
tapes while its evaluator answers with what the DEVICE's evaluation cache holds for each position
(c4_eval_cache_lookup; the net itself for a position the direct-mapped table has since evicted).  Moves,
float64 values, float64 policies and the result must be identical (mcts.py:94-121, training_game.py:8-19).
"""
import numpy as np

from . import c4oracle as oc


def random_tapes(n_games, alpha, seed):
    """[games][42][7] raw Gamma(alpha,1) draws (mcts.py:175-177) and [games][42] uniforms (tree.py:80)."""
    rng = np.random.RandomState(seed)
    return rng.gamma(alpha, 1.0, size=(n_games, 42, 7)), rng.random_sample((n_games, 42))


def oracle_config(cfg):
    return oc.make_config(cfg.simulations, cfg.pb_c_base, cfg.pb_c_init, cfg.root_dirichlet_alpha,
                          cfg.root_exploration_fraction, cfg.num_sampling_moves)


def replay_game(ocfg, engine, net, rec, noise, u):
    """Asserts that the oracle plays exactly `rec` (a c4_game_record).  Returns dict(lookups, evicted)."""
    stats = dict(lookups=0, evicted=0)
    memo = {}

    def fn(c0, c1):
        k = (c0, c1)
        if k not in memo:
            v, p, found = engine.cache_lookup([c0], [c1])
            stats["lookups"] += 1
            if not found[0]:
                stats["evicted"] += 1
                v, p = net.evaluate_bits([c0], [c1], wave=True)
            memo[k] = (float(np.float32(v[0])), [float(x) for x in np.asarray(p[0], dtype=np.float32)])
        v, p = memo[k]
        return v, p, True
    g = oc.selfplay_game(ocfg, oc.CallbackEvaluator(fn), noise, u)
    n = rec.length
    assert g["moves"] == list(rec.move[:n]), "moves differ from the oracle's"
    assert g["boards"] == [(int(rec.color0[i]), int(rec.color1[i])) for i in range(n)]
    assert g["result"] == rec.result
    for i in range(n):
        assert (np.isnan(g["values"][i]) and np.isnan(rec.value[i])) or g["values"][i] == rec.value[i], "values differ"
        assert g["policies"][i] == list(rec.policy[i]), "policies differ"
    return stats


def replay_games_bulk(ocfg, engine, net, recs, noise, u, threads=None):
    """replay_game for MANY games at once: the oracle's lock-step replay pool (OpenMP over games, one memoising evaluator
    table as in evaluators.py:18-25) asks for every position once; each batch of new positions is answered with what the
    device's evaluation cache holds (one c4_eval_cache_lookup per batch; the net for entries the table has since lost).
    Asserts that the oracle plays exactly the given records.  Returns dict(games, positions_asked, lost_by_the_table, ...)."""
    if threads:
        oc.set_threads(threads)
    ids = [int(r.game_id) for r in recs]
    pool = oc.ReplayPool(ocfg, len(recs), np.ascontiguousarray(noise[ids]), np.ascontiguousarray(u[ids]))
    asked = lost = rounds = 0
    try:
        while True:
            m = pool.collect()
            if m == 0:
                break
            c0, c1 = pool.c0[:m].copy(), pool.c1[:m].copy()
            v, p, found = engine.cache_lookup(c0, c1)
            if not found.all():
                miss = np.nonzero(~found)[0]
                nv, npr = net.evaluate_bits(c0[miss], c1[miss], wave=True)
                v[miss], p[miss] = nv, npr
                lost += len(miss)
            asked += m
            rounds += 1
            pool.apply(m, v, p)
        for j, rec in enumerate(recs):
            g = pool.game(j)
            n = rec.length
            assert g["moves"] == list(rec.move[:n]), "game %d: moves differ from the oracle's" % rec.game_id
            assert g["boards"] == [(int(rec.color0[i]), int(rec.color1[i])) for i in range(n)], "game %d: boards differ" % rec.game_id
            assert g["result"] == rec.result, "game %d: result differs" % rec.game_id
            for i in range(n):
                assert (np.isnan(g["values"][i]) and np.isnan(rec.value[i])) or g["values"][i] == rec.value[i], "game %d: values differ" % rec.game_id
                assert g["policies"][i] == list(rec.policy[i]), "game %d: policies differ" % rec.game_id
        st = pool.stats()
    finally:
        pool.close()
    return dict(games=len(recs), positions_asked=asked, lost_by_the_table=lost, rounds=rounds, evaluator_calls=st["lookups"])
