"""Drop-in API on the GPU: the reference's own player tests re-run against connect4_amd
(tests/player_test.py:151-179 test_mcts_next_move: same positions, same configuration, same
acceptance), np.random-seeded self-play reproducing the reference's games move for move, and the
batched device-net path."""
import numpy as np
import pytest

from conftest import load_json, load_npz, table_from_npz

pytestmark = pytest.mark.gpu


def test_mcts_next_move_reference_cases():
    from connect4_amd import evaluators
    from connect4_amd.board import Board
    from connect4_amd.mcts import MCTS, MCTSConfig
    cases = load_json("ref_tests.json")["player"]
    golden = {c["name"]: c for c in load_json("search_centre.json")}
    for i, p in enumerate(cases):
        plies = p["plies"]
        board = Board.from_pieces(np.array(p["o"], dtype=np.bool_), np.array(p["x"], dtype=np.bool_))
        computer = MCTS("mcts_test",
                        MCTSConfig(simulations=7 ** plies + 1 if plies <= 6 else 2 ** plies, pb_c_init=9999),
                        evaluators.Evaluator(evaluators.evaluate_centre_with_prior))
        board_copy = board.__copy__()
        move, value, tree = computer.make_move(board_copy)
        assert move in p["ans"]
        assert board_copy.age == board.age + 1                      # caller's board mutated in place
        g = golden["player%d_testcfg" % i]
        assert [c.data.search_value.visit_count if c.data.search_value else 0 for c in tree.root.children] == \
            [n for n, s in zip(g["N"], g["status"]) if s != -2]
        assert list(tree.get_values_policy()) == g["values_policy"]
        assert list(tree.get_visit_count_policy()) == g["visit_policy"]
        assert tree.best_move().name == g["best_move"] == move


def test_seeded_training_game_reproduces_reference():
    """np.random.seed(k) + training_game(MCTS(...)) gives the reference's game (same RNG calls)."""
    from connect4_amd import evaluators
    from connect4_amd.mcts import MCTS, MCTSConfig
    from connect4_amd.training_game import training_game
    for g in load_json("selfplay.json")[:3]:
        cfg = MCTSConfig(**g["config"])
        np.random.seed(g["seed"])
        gd = training_game(MCTS("az", cfg, evaluators.Evaluator(evaluators.evaluate_centre_with_prior)))
        assert gd.moves == g["moves"]
        assert gd.result.value == g["result"]
        assert [[b.color[0], b.color[1]] for b in gd.boards] == g["boards"]
        assert gd.values == g["values"]
        assert [list(p) for p in gd.priors] == g["policies"]
        td = gd.data
        assert td.values == [g["result"]] * len(g["moves"])


def test_host_evaluator_float32_table():
    """A generic Python evaluator returning float32 priors (what evaluate_nn gives, evaluators.py:41-44)
    takes the float32 score path and reproduces the reference's net-driven search."""
    from connect4_amd.board import Board
    from connect4_amd.evaluators import Evaluator
    from connect4_amd.mcts import MCTSConfig, search
    npz = load_npz("search_net_tables.npz")
    case = [c for c in load_json("search_net.json") if c["name"] == "net_random0_s200"][0]
    c0, c1, v, p = table_from_npz(npz, case["name"])
    table = {(int(a), int(b)): (float(vv), np.asarray(pp, dtype=np.float32)) for a, b, vv, pp in zip(c0, c1, v, p)}
    calls = []

    def evaluate_fn(board):
        calls.append(board.to_int_tuple())
        return table[board.to_int_tuple()]
    ev = Evaluator(evaluate_fn)
    tree = search(MCTSConfig(**case["config"]), Board.from_bits(case["board"]["c0"], case["board"]["c1"]), ev)
    assert [c.data.search_value.visit_count if c.data.search_value else 0 for c in tree.root.children] == \
        [n for n, s in zip(case["N"], case["status"]) if s != -2]
    assert list(tree.get_values_policy()) == case["values_policy"]
    assert len(set(calls)) == len(calls)      # memo table: each position evaluated once


def test_device_net_search_and_generate_games():
    import torch
    from connect4_amd.board import Board
    from connect4_amd.config import MCTSConfig
    from connect4_amd.evaluators import DeviceNetEvaluator
    from connect4_amd.mcts import MCTS
    from connect4_amd.net import InferenceNet, random_init_state_dict
    from connect4_amd.selfplay import generate_games
    net = InferenceNet(random_init_state_dict(seed=0), device="cuda", dtype=torch.float32)
    player = MCTS("dev", MCTSConfig(64), DeviceNetEvaluator(net))
    boards = [Board() for _ in range(5)]
    for b, mv in zip(boards, (0, 3, 3, 6, 2)):
        b.make_move(mv)
    outs = player.make_moves(boards)
    for (move, value, tree), b in zip(outs, boards):
        assert 0 <= move < 7 and b.age == 2
        assert tree.root.data.search_value.visit_count == 65
        assert abs(tree.get_values_policy().sum() - 1.0) < 1e-12
    # the fused MFMA net behind the same player API
    from connect4_amd.fused_net import FusedNet
    fnet = FusedNet(random_init_state_dict(seed=0))
    fboards = [Board() for _ in range(5)]
    for b, mv in zip(fboards, (0, 3, 3, 6, 2)):
        b.make_move(mv)
    fouts = MCTS("fused", MCTSConfig(64), DeviceNetEvaluator(fnet)).make_moves(fboards)
    for (move, value, tree) in fouts:
        assert tree.root.data.search_value.visit_count == 65
    v1, p1 = DeviceNetEvaluator(fnet)(Board())
    assert 0.0 <= v1 <= 1.0 and abs(float(p1.sum()) - 1.0) < 1e-5
    games = generate_games(MCTSConfig.self_play(32), net, n_games=24, n_slots=16, seed=3)
    assert len(games) == 24 and sorted(g.game_id for g in games) == list(range(24))
    for g in games:
        b = Board()
        for i, mv in enumerate(g.moves):
            assert g.boards[i] == b and mv in b.valid_moves
            assert abs(sum(g.priors[i]) - 1.0) < 1e-9
            b.make_move(mv)
        assert b.result == g.result and g.result is not None
    # determinism: same seed, same games (per game id)
    again = generate_games(MCTSConfig.self_play(32), net, n_games=24, n_slots=16, seed=3)
    assert [g.moves for g in again] == [g.moves for g in games]
    fgames = generate_games(MCTSConfig.self_play(32), fnet, n_games=40, n_slots=32, seed=5)
    assert len(fgames) == 40 and all(g.result is not None for g in fgames)
    fagain = generate_games(MCTSConfig.self_play(32), fnet, n_games=40, n_slots=32, seed=5)
    assert [g.moves for g in fagain] == [g.moves for g in fgames]


def test_data_writer_matches_reference_native_to_pytorch():
    """games -> data.pth tensors incl. left-right flip augmentation (data.py:78-105)."""
    from connect4_amd.board import Board
    from connect4_amd.data import games_to_arrays
    from connect4_amd.training_game import GameData
    from connect4_amd.utils import Result
    npz = load_npz("selfplay_net_tables.npz")
    for g in load_json("selfplay_net.json"):
        gd = GameData()
        for bb, mv, v, p in zip(g["boards"], g["moves"], g["values"], g["policies"]):
            gd.add_move(Board.from_bits(*bb), mv, v, np.array(p))
        gd.result = Result(g["result"])
        boards, values, priors = games_to_arrays([gd], add_fliplr=True)
        assert np.array_equal(boards.astype(np.uint8), npz[g["name"] + "__data_boards"])
        assert np.array_equal(values, npz[g["name"] + "__data_values"])
        assert np.array_equal(priors, npz[g["name"] + "__data_priors"])


def test_match_lockstep_equals_sequential_oracle(oracle):
    """Match over the 7 one-ply openings, both colours, two deterministic players of different
    strength: every game's result equals a sequential replay with the oracle."""
    from connect4_amd import evaluators
    from connect4_amd.match import Match
    from connect4_amd.mcts import MCTS, MCTSConfig
    ev = evaluators.Evaluator(evaluators.evaluate_centre_with_prior)
    p1, p2 = MCTS("strong", MCTSConfig(120), ev), MCTS("weak", MCTSConfig(12), ev)
    m = Match(False, p1, p2, plies=1, switch=True)
    assert m.n == 7 and len(m.games) == 14
    out = m.play()
    cfgs = {"strong": oracle.make_config(120), "weak": oracle.make_config(12)}
    wins = draws = losses = 0
    for i, (b0, po, px) in enumerate(Match(False, p1, p2, plies=1, switch=True).games):
        b = oracle.Board.from_bits(*b0.to_int_tuple())
        while b.result == -1:
            who = po if b.age % 2 == 0 else px
            _, mv, _ = oracle.search_and_pick(cfgs[who.name], b, oracle.CentreEvaluator())
            b.make_move(mv)
        r = 0.5 * b.result
        if i >= 7:
            r = 1.0 - r
        wins += r == 1.0
        draws += r == 0.5
        losses += r == 0.0
    assert (out["wins"], out["draws"], out["losses"]) == (wins, draws, losses)


def test_fused_selfplay_kernel_equals_separate_kernels(monkeypatch):
    """c4_selfplay_steps (tree step + network in one persistent kernel) must play exactly the games
    that alternating c4_step / c4_net_forward plays (same seed => same games, id by id)."""
    from connect4_amd.config import MCTSConfig
    from connect4_amd.fused_net import FusedNet
    from connect4_amd.net import random_init_state_dict
    from connect4_amd.selfplay import SelfPlay
    net = FusedNet(random_init_state_dict(seed=0), precision="f16")   # (the workgroup-synchronous kernel serves the fp16 net only)
    cfg = MCTSConfig.self_play(48)
    out = []
    # separate kernels; then the three fused kernels (tree waves + network waves, wave-autonomous, workgroup-synchronous),
    # each with 16 and 32 slots per workgroup (72 slots: ragged last workgroup)
    monkeypatch.setenv("C4_FUSED_PACK", "dense")    # 72 slots in 16- / 32-slot workgroups (the default would spread them one per CU)
    for mode, fused in (("wave", 0), ("split", 16), ("split", 32), ("wave", 16), ("wave", 32), ("block", 16), ("block", 32)):
        monkeypatch.setenv("C4_FUSED_MODE", mode)
        monkeypatch.setenv("C4_FUSED_SLOTS", str(fused or 16))
        sp = SelfPlay(net, 72, cfg, seed=11, games_target=96, record_capacity_games=96, use_graph=False,
                      fused_loop=bool(fused), steps_per_launch=16, max_inner_iters=3)
        for _ in range(400):
            sp.run_steps(64)
            if sp.stats()["active_slots"] == 0:
                break
        games = sorted(sp.drain(), key=lambda g: g.game_id)
        st = sp.stats()
        sp.close()
        assert len(games) == 96
        out.append(([(g.game_id, g.moves, g.result.value, g.values, [list(p) for p in g.priors]) for g in games], st))
    for other in out[1:]:
        assert out[0][0] == other[0]
        for k in ("simulations", "expansions", "moves", "games_finished", "terminal_sims"):
            assert out[0][1][k] == other[1][k]
        assert other[1]["bad_evals"] == 0   # the net never answers NaN
    assert out[0][1]["bad_evals"] == 0


@pytest.mark.parametrize("precision", ["f32x3", "f16"])
@pytest.mark.parametrize("cache_bits", [-1, 8, 0])
def test_split_kernel_edge_cases_play_the_same_games(monkeypatch, cache_bits, precision):
    """Tree waves + network waves against the wave-autonomous kernel where their hand-off is stressed: no evaluation
    cache at all (every leaf goes to a network wave; no speculation), a 256-entry cache (constant eviction, speculative
    inserts overwrite real ones and vice versa), the default cache; 19 slots (one full and one ragged workgroup, tree
    waves with 0..1 slots), more slots than games (slots park while others still play), launches of a single quantum
    (every launch ends with leaves in flight that the next one must pick up).  Same seed => same games, id by id.  Both nets:
    the fp16 net's network waves speculate, the reference-precision net's do not."""
    from connect4_amd.config import MCTSConfig
    from connect4_amd.fused_net import FusedNet
    from connect4_amd.net import random_init_state_dict
    from connect4_amd.selfplay import SelfPlay
    net = FusedNet(random_init_state_dict(seed=0), precision=precision)
    cfg = MCTSConfig.self_play(32)
    out = []
    monkeypatch.setenv("C4_FUSED_PACK", "dense")    # one full and one ragged workgroup
    for mode in ("wave", "split"):
        monkeypatch.setenv("C4_FUSED_MODE", mode)
        sp = SelfPlay(net, 19, cfg, seed=5, games_target=12, record_capacity_games=32, use_graph=False,
                      fused_loop=True, steps_per_launch=1, eval_cache_log2_entries=cache_bits, time_budget_cycles=20000)
        for _ in range(4000):
            sp.run_steps(1)
            if sp.stats()["active_slots"] == 0:
                break
        games = sorted(sp.drain(), key=lambda g: g.game_id)
        st = sp.stats()
        sp.close()
        assert len(games) == 12 and st["bad_evals"] == 0
        if mode == "split" and cache_bits >= 0 and precision == "f16":
            assert st["speculative_evals"] > 0
        if precision == "f32x3":
            assert st["speculative_evals"] == 0
        if cache_bits < 0:
            assert st["speculative_evals"] == 0 and st["eval_cache_hits"] == 0
        out.append(([(g.game_id, g.moves, g.result.value, g.values, [list(p) for p in g.priors]) for g in games],
                    {k: st[k] for k in ("simulations", "expansions", "moves", "games_finished", "terminal_sims", "leaf_evals")}))
    assert out[0] == out[1]
    net.close()


@pytest.mark.parametrize("kind", ["f32x3", "64f"])
def test_split_kernel_other_nets_play_the_same_games(monkeypatch, kind):
    """The reference-precision net (net_forward_wave16q in the network waves) and the 64-filter net (net_forward_wave16w;
    two tree waves of eight slots and six network waves, c4_selfplay_split_kernel<16, MODE, 2>): same seed => the games
    of the wave-autonomous kernel and of the separate kernels (c4_step + c4_net_forward), id by id."""
    from connect4_amd.config import MCTSConfig
    from connect4_amd.fused_net import FusedNet
    from connect4_amd.net import NetConfig, random_init_state_dict
    from connect4_amd.selfplay import SelfPlay
    if kind == "64f":
        net = FusedNet(random_init_state_dict(NetConfig(filters=64, n_fc_layers=6, n_residuals=2), seed=2))
    else:
        net = FusedNet(random_init_state_dict(seed=0), precision="f32x3")
    cfg = MCTSConfig.self_play(32)
    out = []
    monkeypatch.setenv("C4_FUSED_PACK", "dense")
    for mode, fused in (("wave", False), ("wave", True), ("split", True)):
        monkeypatch.setenv("C4_FUSED_MODE", mode)
        sp = SelfPlay(net, 40, cfg, seed=21, games_target=48, record_capacity_games=64, use_graph=False,
                      fused_loop=fused, steps_per_launch=8, max_inner_iters=3)
        for _ in range(2000):
            sp.run_steps(32)
            if sp.stats()["active_slots"] == 0:
                break
        games = sorted(sp.drain(), key=lambda g: g.game_id)
        st = sp.stats()
        sp.close()
        assert len(games) == 48 and st["bad_evals"] == 0
        out.append(([(g.game_id, g.moves, g.result.value, g.values, [list(p) for p in g.priors]) for g in games],
                    {k: st[k] for k in ("simulations", "expansions", "moves", "games_finished", "terminal_sims", "leaf_evals")}))
    assert out[0] == out[1] == out[2]
    net.close()


@pytest.mark.parametrize("precision", ["f32x3", "f16"])
def test_spread_workgroups_play_the_same_games(monkeypatch, precision):
    """A batch that does not fill 16 slots on every CU is spread over all CUs by the split kernel (slot = workgroup + p x
    workgroups: 1,200 games -- the reference's generation, config.py:64 -- are 4-5 slots on each of 256 CUs instead of 16
    slots on 75 of them).  Same seed => the games of the dense packing, id by id; also with three tree waves."""
    from connect4_amd.config import MCTSConfig
    from connect4_amd.fused_net import FusedNet
    from connect4_amd.net import random_init_state_dict
    from connect4_amd.selfplay import SelfPlay
    net = FusedNet(random_init_state_dict(seed=0), precision=precision)
    cfg = MCTSConfig.self_play(32)
    out = []
    for pack, tw in (("dense", "4"), ("spread", "4"), ("spread", "3"), ("dense", "3")):
        monkeypatch.setenv("C4_FUSED_PACK", pack)
        monkeypatch.setenv("C4_SPLIT_TW", tw)
        sp = SelfPlay(net, 600, cfg, seed=9, games_target=700, record_capacity_games=700, use_graph=False,
                      fused_loop=True, steps_per_launch=8)
        for _ in range(2000):
            sp.run_steps(32)
            if sp.stats()["active_slots"] == 0:
                break
        games = sorted(sp.drain(), key=lambda g: g.game_id)
        st = sp.stats()
        sp.close()
        assert len(games) == 700 and st["bad_evals"] == 0 and st["dropped_games"] == 0
        out.append(([(g.game_id, g.moves, g.result.value, g.values, [list(p) for p in g.priors]) for g in games],
                    {k: st[k] for k in ("simulations", "expansions", "moves", "games_finished", "terminal_sims", "leaf_evals")}))
    assert out[0] == out[1] == out[2] == out[3]
    net.close()


@pytest.mark.parametrize("mode", ["split", "wave"])
def test_fused_kernels_contain_a_net_that_answers_nan(monkeypatch, mode):
    """A checkpoint with a NaN in it must not cost memory safety (the reference asserts, model.py:258-263): the fused
    kernels replace non-finite values / priors (network waves before they publish an answer in the split kernel, the
    apply in the others), count them, and the games complete."""
    import torch
    from connect4_amd.config import MCTSConfig
    from connect4_amd.fused_net import FusedNet
    from connect4_amd.net import random_init_state_dict
    from connect4_amd.selfplay import SelfPlay
    sd = random_init_state_dict(seed=0)
    sd["value_head.fc1.bias"] = torch.full_like(sd["value_head.fc1.bias"], float("nan"))     # every value is NaN
    sd["policy_head.fc1.bias"][3] = float("nan")                                              # ... and one logit, so every prior
    net = FusedNet(sd)
    monkeypatch.setenv("C4_FUSED_MODE", mode)
    sp = SelfPlay(net, 24, MCTSConfig.self_play(24), seed=2, games_target=24, record_capacity_games=32, use_graph=False,
                  fused_loop=True, steps_per_launch=8)
    for _ in range(2000):
        sp.run_steps(32)
        if sp.stats()["active_slots"] == 0:
            break
    st = sp.stats()
    games = sp.drain()
    sp.close()
    assert st["bad_evals"] > 0 and len(games) == 24
    for g in games:
        assert 7 <= len(g.moves) <= 42
    # ... and launch by launch mixed with the separate kernels (include/c4_engine.h: the launches are interchangeable): c4_step
    # emits leaves, c4_net_forward writes NaN answers into the GLOBAL buffers, and the fused kernel that follows picks those
    # leaves up as "answered" -- answers no network wave of that kernel has looked at (the split kernel's tree
    # waves carry no check of their own, so its prologue has to make carried-over answers finite)
    sp = SelfPlay(net, 24, MCTSConfig.self_play(24), seed=3, games_target=24, record_capacity_games=32, use_graph=False,
                  fused_loop=False, steps_per_launch=8, max_inner_iters=3)
    for rnd in range(4000):
        sp._fused_loop = False
        sp.run_steps(1)                # c4_step + c4_net_forward: every slot with a leaf now holds a NaN answer in global memory
        sp._fused_loop = True
        sp.run_steps(2)                # the fused kernel applies them
        if rnd % 16 == 15 and sp.stats()["active_slots"] == 0:
            break
    st = sp.stats()
    games = sp.drain()
    sp.close()
    net.close()
    assert st["active_slots"] == 0 and st["bad_evals"] > 0 and len(games) == 24
    for g in games:
        assert 7 <= len(g.moves) <= 42 and all(0 <= m < 7 for m in g.moves)
        for pol in g.priors:
            assert all(x == x for x in pol) and abs(sum(pol) - 1.0) < 1e-9      # no NaN reached a policy


def test_mini_generation_selfplay_train_reload(tmp_path):
    """BASELINE configs[4] flow at toy size on one GPU: fused self-play -> data.pth -> train step ->
    checkpoint -> the next generation's self-play runs on the updated weights."""
    import os
    import torch
    from connect4_amd.config import MCTSConfig
    from connect4_amd.generation import run_generation
    from connect4_amd.training import ModelConfig, Trainer
    torch.manual_seed(0)
    tr = Trainer(ModelConfig(batch_size=256, n_training_epochs=2, use_gpu=True))
    w0 = tr.net.state_dict()["body.0.0.weight"].detach().cpu().clone()
    games0, loss0 = run_generation(tr, MCTSConfig.self_play(24), n_games=48, save_dir=str(tmp_path), gen=0, n_slots=48)
    assert len(games0) == 48 and loss0 is not None and loss0 == loss0
    d = torch.load(os.path.join(str(tmp_path), "0", "data.pth"), weights_only=True)
    n_pos = games0.n_positions
    assert n_pos == int(games0.lengths.sum()) and sorted(games0.ids.tolist()) == list(range(48))
    assert d["boards"].shape == (2 * n_pos, 3, 6, 7) and d["values"].shape == (2 * n_pos,) and d["priors"].shape == (2 * n_pos, 7)
    assert os.path.exists(os.path.join(str(tmp_path), "0", "net.pth"))
    assert not torch.equal(w0, tr.net.state_dict()["body.0.0.weight"].cpu())
    t1 = {}
    games1, _ = run_generation(tr, MCTSConfig.self_play(24), n_games=48, save_dir=str(tmp_path), gen=1, n_slots=48, timings=t1)
    assert len(games1) == 48 and t1["training_rows"] == 2 * games1.n_positions      # window of generation 1: itself
    # the sliding window (data.py:66-75): generation 3 trains on generations 3 and 2, newest first
    games2, _ = run_generation(tr, MCTSConfig.self_play(24), n_games=48, save_dir=str(tmp_path), gen=2, n_slots=48)
    t3 = {}
    games3, _ = run_generation(tr, MCTSConfig.self_play(24), n_games=48, save_dir=str(tmp_path), gen=3, n_slots=48,
                               write_games_pkl=True, timings=t3)
    assert t3["training_rows"] == 2 * (games3.n_positions + games2.n_positions)
    from connect4_amd.data import load_games
    back = load_games(os.path.join(str(tmp_path), "3"))
    assert [g.moves for g in back] == [g.moves for g in games3.to_game_data()]


def test_end_to_end_net_driven_search_matches_reference():
    """The whole hot path on the GPU -- HIP tree walk + the policy/value net -- against the reference's
    NN-driven searches with data/example_net.pth (tests/golden/search_net.json, captured from the
    unmodified reference on CPU).
      * fp32 PyTorch-ROCm net and the fused MFMA net AS SHIPPED (FusedNet / make_selfplay_net with no precision
        argument = the reference-precision mode: fp16 hi+lo split, fp32 accumulation): outputs differ from the CPU's
        by ~1e-6, far below the score gaps that decide an argmax in these positions -> visit counts must be
        IDENTICAL, root value sums within 1e-4;
      * fused net in the opt-in fp16-storage mode (precision="f16"): outputs differ by up to 5e-3, so individual
        visit counts may move; stated tolerance: the visit distribution stays within 0.08 total variation
        and the chosen move is the same."""
    import torch
    from connect4_amd.board import Board
    from connect4_amd.evaluators import DeviceNetEvaluator
    from connect4_amd.fused_net import FusedNet, make_selfplay_net
    from connect4_amd.mcts import MCTSConfig, search
    from connect4_amd.net import InferenceNet
    z = load_npz("net_golden.npz")
    sd = {k[3:]: torch.from_numpy(z[k]) for k in z.files if k.startswith("w__")}
    default_net = make_selfplay_net(sd)
    assert isinstance(default_net, FusedNet) and default_net.precision == "f32x3" and FusedNet(sd).precision == "f32x3"
    nets = {"fp32": InferenceNet(sd, device="cuda", dtype=torch.float32), "fused_f32x3": default_net,
            "fused_f16": FusedNet(sd, precision="f16")}
    cases = [c for c in load_json("search_net.json") if c["noise"] is None]
    assert len(cases) >= 7
    exact = {"fp32": 0, "fused_f32x3": 0}
    for case in cases:
        board = Board.from_bits(case["board"]["c0"], case["board"]["c1"])
        ref_n = np.array(case["N"], dtype=np.float64)
        for name, net in nets.items():
            tree = search(MCTSConfig(**case["config"]), board, DeviceNetEvaluator(net))
            got = np.zeros(7)
            for c in tree.root.children:
                got[c.name] = c.data.search_value.visit_count if c.data.search_value else 0
            if name in exact:
                assert got.tolist() == ref_n.tolist(), (name, case["name"], got, ref_n)
                assert abs(tree.root.data.search_value.value_sum - case["root_W"]) < 1e-4 * case["root_N"]
                exact[name] += 1
            else:
                tv = 0.5 * np.abs(got / got.sum() - ref_n / ref_n.sum()).sum()
                assert tv < 0.08, (case["name"], got, ref_n, tv)
                assert tree.best_move().name == case["best_move"], case["name"]
    assert exact == {"fp32": len(cases), "fused_f32x3": len(cases)}


def test_precise_fused_selfplay_equals_separate_kernels(oracle):
    """The reference-precision net inside the wave-autonomous self-play kernel plays exactly the games that
    alternating c4_step / c4_net_forward launches play, and they replay on the oracle from the device's cache."""
    from connect4_amd import _lib as L
    from connect4_amd.config import MCTSConfig
    from connect4_amd.fused_net import FusedNet
    from connect4_amd.net import random_init_state_dict
    from connect4_amd.selfplay import SelfPlay
    from oracle.replay import oracle_config, random_tapes, replay_game
    net = FusedNet(random_init_state_dict(seed=0), precision="f32x3")
    cfg = MCTSConfig.self_play(40)
    noise, u = random_tapes(80, cfg.root_dirichlet_alpha, seed=8)
    out = []
    for fused in (False, True):
        sp = SelfPlay(net, 40, cfg, seed=2, games_target=80, record_capacity_games=80, use_graph=False, fused_loop=fused,
                      steps_per_launch=16, max_inner_iters=3, rng_mode=L.RNG_TAPE)
        sp.engine.set_tapes(noise, u)
        sp.engine.reset()
        for _ in range(400):
            sp.run_steps(64)
            if sp.stats()["active_slots"] == 0:
                break
        recs = sp.engine.drain_games()
        assert len(recs) == 80 and sp.stats()["bad_evals"] == 0
        if fused:
            for r in recs[:6]:
                replay_game(oracle_config(cfg), sp.engine, net, r, noise[r.game_id], u[r.game_id])
        out.append([(r.game_id, list(r.move[:r.length]), r.result, list(r.value[:r.length])) for r in recs])
        sp.close()
    assert out[0] == out[1]


def test_wide_net_runs_on_the_fused_kernel(oracle):
    """example_config's 64-filter / 6-block / 6-fc net (data/example_config.py:8-16) runs on the fused one-position
    forward (two cout blocks per MFMA step); an unsupported width still gets the PyTorch-ROCm plan; the games of
    the wide net replay on the oracle from the device's evaluation cache."""
    from connect4_amd import _lib as L
    from connect4_amd.config import MCTSConfig
    from connect4_amd.fused_net import FusedNet, make_selfplay_net
    from connect4_amd.net import InferenceNet, NetConfig, random_init_state_dict
    from connect4_amd.selfplay import SelfPlay, generate_games
    from oracle.replay import oracle_config, random_tapes, replay_game
    wide = make_selfplay_net(random_init_state_dict(NetConfig(filters=64, n_fc_layers=6, n_residuals=6), seed=0))
    assert isinstance(wide, FusedNet) and wide.config.filters == 64
    assert isinstance(make_selfplay_net(random_init_state_dict(seed=0)), FusedNet)
    other = make_selfplay_net(random_init_state_dict(NetConfig(filters=48, n_residuals=1), seed=0))
    assert isinstance(other, InferenceNet)
    games = generate_games(MCTSConfig.self_play(16), other, n_games=8, n_slots=8, seed=0)
    assert len(games) == 8 and all(g.result is not None for g in games)
    with pytest.raises(L.EngineError):       # the hi/lo planes of 64 filters do not fit a wave's LDS: refused, not approximated
        FusedNet(random_init_state_dict(NetConfig(filters=64), seed=0), precision="f32x3")
    cfg = MCTSConfig.self_play(24)
    noise, u = random_tapes(40, cfg.root_dirichlet_alpha, seed=4)
    sp = SelfPlay(wide, 40, cfg, seed=0, games_target=40, record_capacity_games=40, use_graph=False, fused_loop=True,
                  steps_per_launch=16, rng_mode=L.RNG_TAPE)
    sp.engine.set_tapes(noise, u)
    sp.engine.reset()
    for _ in range(400):
        sp.run_steps(64)
        if sp.stats()["active_slots"] == 0:
            break
    recs = sp.engine.drain_games()
    assert len(recs) == 40 and sp.stats()["bad_evals"] == 0
    for r in recs[:5]:
        replay_game(oracle_config(cfg), sp.engine, wide, r, noise[r.game_id], u[r.game_id])
    sp.close()
