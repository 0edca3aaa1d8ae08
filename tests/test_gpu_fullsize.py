"""BASELINE shapes at FULL size under -m gpu, on the PRODUCTION path: c4_selfplay_split_kernel (tree waves + network waves)
+ the evaluation cache + the fused MFMA net, with injected RNG tapes; then a sample of >= 256 finished games (64 at 3200
simulations) is replayed move for move on the CPU oracle (its lock-step replay pool, OpenMP over games), whose memoising
evaluator is answered with what the DEVICE's evaluation cache holds for each position (c4_eval_cache_lookup).  Moves,
float64 values, float64 policies and results must be IDENTICAL.  Reference: mcts.py:94-121, training_game.py:8-19,
evaluators.py:9-25.

(i)   configs[1]: 4096 games x 800 simulations, the shipped default net (reference precision, f32x3) and the opt-in fp16 net;
(ii)  configs[2]'s per-GPU share: 8192 games x 800 simulations (32 slots per workgroup: eight slots per tree wave; the fp16
      net's network waves take two requests per pass there);
(iii) configs[3]: 8192 games x 3200 simulations (deep trees), reference-precision net;
(iv)  configs[3] search-only: 8192 searches x 3200 simulations with the in-kernel centre evaluator against the oracle.

Positions the direct-mapped table has lost to a later collision (2^28..2^29 entries at a few per cent load: a few per cent
of the positions, bounded below) are RE-EVALUATED BY THE NET for the oracle -- for those the replay checks the engine
against the net's answer, not against the stored one.
"""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def play_and_replay(oracle, G, S, precision, n_replay, seed, lost_share_max):
    from connect4_amd import _lib as L
    from connect4_amd.config import MCTSConfig
    from connect4_amd.fused_net import FusedNet
    from connect4_amd.net import random_init_state_dict
    from connect4_amd.selfplay import SelfPlay
    from oracle.replay import oracle_config, random_tapes, replay_games_bulk
    cfg = MCTSConfig.self_play(S)
    net = FusedNet(random_init_state_dict(seed=0), precision=precision)
    assert net.precision == ("f32x3" if precision is None else precision)
    sp = SelfPlay(net, G, cfg, seed=0, games_target=G, record_capacity_games=G, use_graph=False, fused_loop=True,
                  steps_per_launch=128, max_inner_iters=32, rng_mode=L.RNG_TAPE)
    try:
        noise, u = random_tapes(G, cfg.root_dirichlet_alpha, seed=seed)
        sp.engine.set_tapes(noise, u)
        sp.engine.reset()
        for _ in range(4000):
            sp.run_steps(256)
            st = sp.stats()
            if st["active_slots"] == 0:
                break
        assert st["active_slots"] == 0 and st["games_finished"] == G and st["dropped_games"] == 0 and st["bad_evals"] == 0
        assert st["simulations"] == S * st["moves"]
        recs = sp.engine.drain_games()
        assert len(recs) == G and [r.game_id for r in recs] == list(range(G))
        lengths = np.array([r.length for r in recs])
        rng = np.random.RandomState(5)
        sample = set(rng.choice(G, size=n_replay - 2, replace=False).tolist())
        sample |= {int(np.argmax(lengths)), int(np.argmin(lengths))}          # the longest and the shortest game too
        res = replay_games_bulk(oracle_config(cfg), sp.engine, net, [recs[g] for g in sorted(sample)], noise, u, threads=16)
        print("%d x %d %s: %d games replayed on the oracle, %d positions asked in %d rounds, %d lost by the table (re-evaluated by the net)"
              % (G, S, net.precision, res["games"], res["positions_asked"], res["rounds"], res["lost_by_the_table"]))
        # the table really is what answered
        assert res["games"] >= n_replay - 2 and res["positions_asked"] > 1000
        assert res["lost_by_the_table"] <= lost_share_max * res["positions_asked"]
        return st
    finally:
        sp.close()
        net.close()


@pytest.mark.parametrize("precision", [None, "f16"])
def test_config1_4096_games_800_sims_split_kernel_vs_oracle(oracle, precision):
    play_and_replay(oracle, 4096, 800, precision, n_replay=256, seed=123, lost_share_max=0.10)


@pytest.mark.parametrize("precision", [None, "f16"])
def test_config2_share_8192_games_800_sims_split_kernel_vs_oracle(oracle, precision):
    st = play_and_replay(oracle, 8192, 800, precision, n_replay=256, seed=321, lost_share_max=0.15)
    assert st["eval_cache_hits"] > 0


def test_config3_8192_games_3200_sims_split_kernel_vs_oracle(oracle):
    play_and_replay(oracle, 8192, 3200, None, n_replay=64, seed=77, lost_share_max=0.25)


def test_config4_8192_searches_3200_sims_vs_oracle(oracle):
    from connect4_amd import _lib as L
    from connect4_amd.engine import Engine
    G, S = 8192, 3200
    rng = np.random.RandomState(0)
    boards = []
    while len(boards) < G:
        b = oracle.Board.empty()
        for _ in range(int(rng.randint(0, 12))):
            m = b.valid_mask()
            if not m:
                break
            b.make_move(int(rng.choice([c for c in range(7) if (m >> c) & 1])))
        if b.result == oracle.NONE:
            boards.append(b)
    with Engine(G, S, eval_mode=L.EVAL_CENTRE, stop_after_move=True) as eng:
        eng.reset([b.key()[0] for b in boards], [b.key()[1] for b in boards])
        eng.run_centre(max_launches=8)
        st = eng.stats()
        roots = eng.read_roots()
    assert st["active_slots"] == 0 and st["simulations"] == G * S and st["moves"] == G
    cfg = oracle.make_config(S)
    exp = 0
    for i in rng.choice(G, size=32, replace=False):
        info, mv, _ = oracle.search_and_pick(cfg, boards[i], oracle.CentreEvaluator())
        assert list(roots[i].child_visits) == list(info.child_visits), i
        assert list(roots[i].child_value_sum) == list(info.child_value_sum), i
        assert roots[i].move == mv and roots[i].expansions == info.n_expansions
        exp += info.n_expansions
    assert exp > 0


def _popcount_cols(occ):
    """[n] int64 occupancy -> [n, 7] stones per column (torch, any device)."""
    import torch
    cols = []
    for c in range(7):
        col = (occ >> (7 * c)) & 0x3f
        cnt = torch.zeros_like(col)
        for r in range(6):
            cnt = cnt + ((col >> r) & 1)
        cols.append(cnt)
    return torch.stack(cols, 1)


def _wins(b):
    """board.py:173-184 on a tensor of int64 bitboards."""
    out = None
    for s in (6, 7, 8, 1):
        y = b & (b >> s)
        hit = (y & (y >> (2 * s))) != 0
        out = hit if out is None else (out | hit)
    return out


def test_soak_production_selfplay_every_record_is_a_legal_complete_game():
    """Ten seconds of the production configuration -- 4096 games x 800 simulations, Philox RNG, evaluation cache,
    device-side export behind every launch -- and EVERY exported game (~150 k games, ~4.5 M positions) is checked on
    the device with vectorised bit arithmetic: plies chain (board + recorded move = next board, o and x alternate, the
    move is legal), the first board is empty, the last move ends the game with the recorded result (four in a row for
    the mover, or a full board without one) and no earlier position was already decided, policies are distributions
    over legal columns, values lie in [0, 1], ids are unique and nothing was dropped.  (The exact, move-for-move
    parity of this path is test_config1_...; this one is about the record machinery at scale: ring, staging, export.)"""
    import time
    import torch
    from connect4_amd.config import MCTSConfig
    from connect4_amd.fused_net import FusedNet
    from connect4_amd.net import random_init_state_dict
    from connect4_amd.packed import PackedGames
    from connect4_amd.selfplay import SelfPlay
    G = 4096
    net = FusedNet(random_init_state_dict(seed=0))
    sp = SelfPlay(net, G, MCTSConfig.self_play(800), seed=7, games_target=-1, record_capacity_games=2 * G, use_graph=False,
                  fused_loop=True, steps_per_launch=256, max_inner_iters=32)
    parts = []
    t0 = time.time()
    while time.time() - t0 < 10.0:
        for _ in range(8):
            sp.run_steps(256)
        parts.append(sp.engine.export_games())          # consumes what finished; synchronises
    st = sp.stats()
    parts.append(sp.engine.export_games())
    sp.close()
    net.close()
    p = PackedGames.cat(parts)
    assert st["dropped_games"] == 0 and st["bad_evals"] == 0
    assert p.n_games == st["games_finished"] and p.n_games > 50_000
    assert torch.unique(p.ids).numel() == p.n_games
    L = p.lengths.long()
    assert int(L.min()) >= 7 and int(L.max()) <= 42 and int(L.sum()) == p.n_positions
    first = torch.cumsum(L, 0) - L                          # index of every game's first position
    last = first + L - 1
    c0, c1 = p.boards[:, 0], p.boards[:, 1]
    occ = c0 | c1
    mv = p.moves.long()
    ply = torch.arange(p.n_positions, device=occ.device) - first[p.game_index.long()]
    heights = _popcount_cols(occ)
    assert torch.equal(heights.sum(1), ply)                 # ply = stones on the board; o moves on even plies
    assert bool(((c0 & c1) == 0).all()) and bool((c0[first] == 0).all()) and bool((c1[first] == 0).all())
    h = heights.gather(1, mv[:, None])[:, 0]
    assert bool((h < 6).all())                              # the recorded move is legal
    stone = torch.ones_like(occ) << (7 * mv + h)
    n0 = torch.where(ply % 2 == 0, c0 | stone, c0)
    n1 = torch.where(ply % 2 == 1, c1 | stone, c1)
    inner = torch.ones(p.n_positions, dtype=torch.bool, device=occ.device)
    inner[last] = False
    nxt = torch.nonzero(inner)[:, 0]
    assert torch.equal(n0[nxt], c0[nxt + 1]) and torch.equal(n1[nxt], c1[nxt + 1])     # plies chain inside a game
    assert not bool((_wins(c0) | _wins(c1)).any())          # no recorded position was already decided
    owin, xwin = _wins(n0[last]), _wins(n1[last])
    full = (ply[last] + 1) == 42
    res = p.results.long()
    assert torch.equal(res, torch.where(owin, 2, torch.where(xwin, 0, 1)))
    assert bool((owin | xwin | full).all()) and bool((~(owin & xwin)).all())
    assert torch.equal(p.targets, 0.5 * res[p.game_index.long()].float())
    pol = p.policy
    assert bool((pol >= 0).all()) and bool(((pol.sum(1) - 1).abs() < 1e-5).all())
    assert bool((pol[heights >= 6] == 0).all())             # no mass on full columns
    v = p.values[~torch.isnan(p.values)]
    assert bool(((v >= 0) & (v <= 1)).all())
    print("soak: %d games, %d positions checked on the device in %.1f s of self-play" % (p.n_games, p.n_positions, time.time() - t0))
