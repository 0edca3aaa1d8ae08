#!/usr/bin/env python3
"""Generate the golden fixtures under tests/golden/ by importing the UNMODIFIED reference.

Runs in the build container only (the reference never travels to the GPU box):

    cd /tmp && PYTHONDONTWRITEBYTECODE=1 \
      PYTHONPATH=/root/repo/oracle/refshim:/root/reference python /root/repo/tests/golden/gen_golden.py

oracle/refshim/ holds stand-ins for the two third-party modules the reference imports but this
image lacks (anytree, visdom); they contain no reference logic (see their docstrings).

What is written (all data: inputs + expected outputs, no reference source text):
  ref_tests.json      positions/answers held by the reference's own tests
                      (tests/board_test.py:10-161,164-247; tests/player_test.py:13-118)
  board.json          seeded random playouts through oinkoink.board.Board
  search_centre.json  mcts.search with evaluate_centre_with_prior (deterministic float64)
  search_net_*.npz    mcts.search driven by data/example_net.pth, captured together with the
                      evaluator's position_table so that tree logic replays without conv numerics
  selfplay.json / selfplay_net.npz   training_game() with recorded RNG tapes
  net_golden.npz      example_net.pth weights (data file) + ModelWrapper outputs on fixed positions
  train_step.npz      ModelWrapper.train (model.py:200-240) on a seeded net and a small seeded dataset:
                      initial weights, the dataset, torch seed, final weights / BN statistics / momentum
                      (`python gen_golden.py train` regenerates only this file)
"""
import json
import os
import sys
from copy import copy
from functools import partial

import numpy as np

OUT = os.path.dirname(os.path.abspath(__file__))

import torch  # noqa: E402

from oinkoink.board import Board, make_random_ips  # noqa: E402
from oinkoink.evaluators import Evaluator, evaluate_centre, evaluate_centre_with_prior, evaluate_nn  # noqa: E402
from oinkoink.mcts import MCTS, MCTSConfig, search  # noqa: E402
from oinkoink.neural.config import ModelConfig  # noqa: E402
from oinkoink.neural.pytorch.model import ModelWrapper  # noqa: E402
from oinkoink.neural.training_game import training_game  # noqa: E402
from oinkoink.neural.pytorch.data import native_to_pytorch  # noqa: E402
from oinkoink.utils import Result  # noqa: E402

REF = os.path.dirname(os.path.dirname(os.path.abspath(sys.modules["oinkoink"].__file__)))


def res_code(r):
    return None if r is None else float(r.value)


def board_dict(b):
    return dict(c0=int(b.color[0]), c1=int(b.color[1]), age=int(b.age), result=res_code(b.result),
                valid=sorted(int(m) for m in b.valid_moves))


# ---------------------------------------------------------------- reference's own tests -> data
def dump_ref_tests():
    sys.path.insert(0, REF)
    import tests.board_test as bt
    import tests.player_test as pt

    out = {}
    out["check_valid"] = [
        dict(o=np.asarray(o, dtype=int).tolist(), x=np.asarray(x, dtype=int).tolist(), ans=a)
        for o, x, a in zip(bt.pieces_1, bt.pieces_2, bt.ans)]

    # test_valid_moves keeps its positions inline: record them while the test itself runs.
    calls = []
    orig = Board.from_pieces.__func__

    def recording(cls, o_pieces, x_pieces):
        b = orig(cls, o_pieces, x_pieces)
        calls.append((np.asarray(o_pieces, dtype=int).tolist(), np.asarray(x_pieces, dtype=int).tolist(), b))
        return b
    Board.from_pieces = classmethod(recording)
    try:
        bt.test_valid_moves()
    finally:
        Board.from_pieces = classmethod(orig)
    out["valid_moves"] = [dict(o=o, x=x, valid=sorted(int(m) for m in b.valid_moves)) for o, x, b in calls]

    out["player"] = [
        dict(o=np.asarray(o, dtype=int).tolist(), x=np.asarray(x, dtype=int).tolist(), plies=int(p), ans=list(a))
        for o, x, p, a in zip(pt.o_pieces, pt.x_pieces, pt.plies, pt.ans)]
    with open(os.path.join(OUT, "ref_tests.json"), "w") as f:
        json.dump(out, f)
    return out


# ---------------------------------------------------------------- board playouts
def random_position(rng, plies):
    """Random legal playout of `plies` moves that is still undecided (retry otherwise)."""
    while True:
        b = Board()
        ok = True
        for _ in range(plies):
            moves = sorted(b.valid_moves)
            if not moves:
                ok = False
                break
            b.make_move(int(rng.choice(moves)))
        if ok and b.result is None:
            return b


def dump_board():
    rng = np.random.RandomState(1234)
    playouts = []
    for g in range(200):
        b = Board()
        moves, states = [], []
        while b.result is None:
            m = int(rng.choice(sorted(b.valid_moves)))
            b.make_move(m)
            moves.append(m)
            st = board_dict(b)
            st["planes"] = np.asarray(b.to_array(), dtype=int).reshape(-1).tolist() if g < 20 else None
            fl = b.create_fliplr()
            st["flip"] = [int(fl.color[0]), int(fl.color[1])]
            st["centre"] = float(evaluate_centre(b))
            states.append(st)
        playouts.append(dict(moves=moves, states=states))
    ips = {str(p): sorted([int(b.color[0]), int(b.color[1])] for b in make_random_ips(p)) for p in (0, 1, 2, 3)}
    with open(os.path.join(OUT, "board.json"), "w") as f:
        json.dump(dict(playouts=playouts, ips=ips), f)


# ---------------------------------------------------------------- searches
def tree_summary(tree, board):
    root = tree.root
    N, W, status = [0] * 7, [0.0] * 7, [-2] * 7
    val = [0.0] * 7
    for c in root.children:
        sv = c.data.search_value
        N[c.name] = 0 if sv is None else int(sv.visit_count)
        W[c.name] = 0.0 if sv is None else float(sv.value_sum)
        status[c.name] = -1 if c.data.board.result is None else int(c.data.board.result.value * 2)
        val[c.name] = float(tree.get_node_value(c))
    n_nodes = n_exp = 0
    stack = [root]
    while stack:
        n = stack.pop()
        n_nodes += 1
        if n.children:
            n_exp += 1
            stack.extend(n.children)
    return dict(root_N=int(root.data.search_value.visit_count), root_W=float(root.data.search_value.value_sum),
                N=N, W=W, status=status, child_value=val,
                values_policy=[float(x) for x in tree.get_values_policy()],
                visit_policy=[float(x) for x in tree.get_visit_count_policy()],
                root_prior=[float(x) for x in root.data.position_value.prior],
                best_move=int(tree.best_move().name), n_nodes=n_nodes, n_expansions=n_exp)


def cfg_dict(c):
    return dict(simulations=c.simulations, pb_c_base=c.pb_c_base, pb_c_init=c.pb_c_init,
                root_dirichlet_alpha=c.root_dirichlet_alpha,
                root_exploration_fraction=c.root_exploration_fraction,
                num_sampling_moves=c.num_sampling_moves)


class GammaRecorder:
    """Records what np.random.gamma returned / which uniform np.random.choice consumed."""

    def __enter__(self):
        self.noise, self.uniforms = [], []
        self._g, self._c = np.random.gamma, np.random.choice

        def gamma(*a, **k):
            r = self._g(*a, **k)
            self.noise.append([float(x) for x in np.asarray(r).reshape(-1)])
            return r

        def choice(*a, **k):
            st = np.random.get_state()
            replica = np.random.RandomState()
            replica.set_state(st)
            self.uniforms.append(float(replica.random_sample()))
            return self._c(*a, **k)
        np.random.gamma, np.random.choice = gamma, choice
        return self

    def __exit__(self, *exc):
        np.random.gamma, np.random.choice = self._g, self._c


def run_search_case(name, board, cfg, evaluator, seed=None):
    if seed is not None:
        np.random.seed(seed)
    with GammaRecorder() as rec:
        tree = search(cfg, board, evaluator)
    d = dict(name=name, board=board_dict(board), config=cfg_dict(cfg), noise=rec.noise[0] if rec.noise else None)
    d.update(tree_summary(tree, board))
    return d


def dump_search_centre(ref_tests):
    cases = []
    ev = lambda: Evaluator(evaluate_centre_with_prior)  # noqa: E731
    for s in (1, 2, 3, 8, 50, 200, 800, 3200):
        cases.append(run_search_case("empty_s%d" % s, Board(), MCTSConfig(s), ev()))
    for i, p in enumerate(ref_tests["player"]):
        b = Board.from_pieces(np.array(p["o"], dtype=np.bool_), np.array(p["x"], dtype=np.bool_))
        plies = p["plies"]
        sims = 7 ** plies + 1 if plies <= 6 else 2 ** plies
        c = run_search_case("player%d_testcfg" % i, b, MCTSConfig(simulations=sims, pb_c_init=9999), ev())
        c["accepted_moves"] = p["ans"]
        cases.append(c)
        cases.append(run_search_case("player%d_s800" % i, b, MCTSConfig(800), ev()))
    rng = np.random.RandomState(99)
    for i in range(24):
        b = random_position(rng, int(rng.randint(1, 36)))
        cases.append(run_search_case("random%d_s200" % i, b, MCTSConfig(200), ev()))
    for i in range(4):
        b = random_position(rng, int(rng.randint(0, 12)))
        cases.append(run_search_case("noise%d_s300" % i, b,
                                     MCTSConfig(300, root_dirichlet_alpha=0.3, root_exploration_fraction=0.25),
                                     ev(), seed=100 + i))
    with open(os.path.join(OUT, "search_centre.json"), "w") as f:
        json.dump(cases, f)
    return cases


def table_arrays(table):
    keys = sorted(table.keys(), key=lambda k: (int(k[0]), int(k[1])))
    c0 = np.array([int(k[0]) for k in keys], dtype=np.uint64)
    c1 = np.array([int(k[1]) for k in keys], dtype=np.uint64)
    v = np.array([np.float32(table[k][0]) for k in keys], dtype=np.float32)
    p = np.stack([np.asarray(table[k][1], dtype=np.float32) for k in keys]).astype(np.float32)
    for k in keys:  # the table must be exactly float32-typed for the replay to be faithful
        assert float(np.float32(table[k][0])) == float(table[k][0])
        assert np.asarray(table[k][1]).dtype == np.float32
    return c0, c1, v, p


def dump_search_net(model):
    cases, blobs = [], {}
    rng = np.random.RandomState(7)

    def one(name, board, cfg, seed=None):
        evaluator = Evaluator(partial(evaluate_nn, model=model))
        c = run_search_case(name, board, cfg, evaluator, seed)
        c0, c1, v, p = table_arrays(evaluator.position_table)
        for k, a in (("c0", c0), ("c1", c1), ("v", v), ("p", p)):
            blobs["%s__%s" % (name, k)] = a
        cases.append(c)

    one("net_empty_s800", Board(), MCTSConfig(800))
    one("net_empty_s3200", Board(), MCTSConfig(3200))
    one("net_empty_s800_noise", Board(),
        MCTSConfig(800, root_dirichlet_alpha=0.3, root_exploration_fraction=0.25), seed=5)
    for i in range(6):
        b = random_position(rng, int(rng.randint(2, 30)))
        one("net_random%d_s200" % i, b, MCTSConfig(200))
        one("net_random%d_s200_noise" % i, b,
            MCTSConfig(200, root_dirichlet_alpha=0.3, root_exploration_fraction=0.25), seed=50 + i)
    np.savez_compressed(os.path.join(OUT, "search_net_tables.npz"), **blobs)
    with open(os.path.join(OUT, "search_net.json"), "w") as f:
        json.dump(cases, f)


# ---------------------------------------------------------------- self-play games
def game_dict(gd, rec, cfg):
    return dict(config=cfg_dict(cfg), moves=[int(m) for m in gd.moves],
                boards=[[int(b.color[0]), int(b.color[1])] for b in gd.boards],
                values=[None if v is None else float(v) for v in gd.values],
                policies=[[float(x) for x in p] for p in gd.priors],
                result=float(gd.result.value), noise_tape=rec.noise, uniforms=rec.uniforms)


def dump_selfplay(model):
    games = []
    for seed, sims in ((0, 100), (1, 100), (2, 60), (3, 300)):
        cfg = MCTSConfig(sims, root_dirichlet_alpha=0.3, root_exploration_fraction=0.25, num_sampling_moves=6)
        np.random.seed(seed)
        with GammaRecorder() as rec:
            gd = training_game(MCTS("ref", cfg, Evaluator(evaluate_centre_with_prior)))
        g = game_dict(gd, rec, cfg)
        g["seed"] = seed
        games.append(g)
    with open(os.path.join(OUT, "selfplay.json"), "w") as f:
        json.dump(games, f)

    blobs, net_games = {}, []
    for seed, sims in ((0, 100), (1, 50)):
        cfg = MCTSConfig(sims, root_dirichlet_alpha=0.3, root_exploration_fraction=0.25, num_sampling_moves=6)
        evaluator = Evaluator(partial(evaluate_nn, model=model))
        np.random.seed(seed)
        with GammaRecorder() as rec:
            gd = training_game(MCTS("ref", cfg, evaluator))
        g = game_dict(gd, rec, cfg)
        g["seed"] = seed
        g["name"] = "netgame%d" % seed
        c0, c1, v, p = table_arrays(evaluator.position_table)
        for k, a in (("c0", c0), ("c1", c1), ("v", v), ("p", p)):
            blobs["%s__%s" % (g["name"], k)] = a
        net_games.append(g)
        # data.pth conversion of this game (data.py:78-105) incl. left-right flip augmentation
        td = gd.data
        bt, vt, ptt = native_to_pytorch(list(td.boards), list(td.values), list(td.priors), add_fliplr=True)
        blobs["%s__data_boards" % g["name"]] = bt.numpy().astype(np.uint8)
        blobs["%s__data_values" % g["name"]] = vt.numpy()
        blobs["%s__data_priors" % g["name"]] = ptt.numpy()
    np.savez_compressed(os.path.join(OUT, "selfplay_net_tables.npz"), **blobs)
    with open(os.path.join(OUT, "selfplay_net.json"), "w") as f:
        json.dump(net_games, f)


# ---------------------------------------------------------------- net
def dump_net(model):
    sd = model.net.state_dict()
    blobs = {"w__" + k: v.detach().cpu().numpy() for k, v in sd.items()}
    rng = np.random.RandomState(11)
    boards = [Board()] + [random_position(rng, int(rng.randint(1, 40))) for _ in range(95)]
    values, priors = model(boards)
    v1, p1 = model(boards[0])
    blobs["in_c0"] = np.array([int(b.color[0]) for b in boards], dtype=np.uint64)
    blobs["in_c1"] = np.array([int(b.color[1]) for b in boards], dtype=np.uint64)
    blobs["in_planes"] = np.stack([b.to_array() for b in boards]).astype(np.uint8)
    blobs["out_values"] = np.asarray(values, dtype=np.float32)
    blobs["out_priors"] = np.asarray(priors, dtype=np.float32)
    blobs["out_value_single"] = np.asarray(v1, dtype=np.float32)
    blobs["out_prior_single"] = np.asarray(p1, dtype=np.float32)
    np.savez_compressed(os.path.join(OUT, "net_golden.npz"), **blobs)


# ---------------------------------------------------------------- train step
def dump_train_step():
    """One generation of the reference's training recipe on CPU, everything seeded."""
    from oinkoink.neural.pytorch.data import Connect4Dataset
    torch.set_num_threads(1)
    torch.manual_seed(7)
    cfg = ModelConfig(use_gpu=False, batch_size=64, n_training_epochs=2)
    model = ModelWrapper(cfg)                                   # seeded random init
    blobs = {"init__" + k: v.detach().clone().numpy() for k, v in model.net.state_dict().items()}
    rng = np.random.RandomState(3)
    n = 200                                                     # 3 full batches + a ragged one of 8
    boards = [random_position(rng, int(rng.randint(0, 38))) for _ in range(n)]
    bt = torch.FloatTensor(np.stack([b.to_array() for b in boards]))
    vt = torch.FloatTensor(rng.choice([0.0, 0.5, 1.0], size=n))
    pt = torch.FloatTensor(rng.dirichlet(np.ones(7), size=n))
    blobs["data_boards"] = bt.numpy().astype(np.uint8)
    blobs["data_values"] = vt.numpy()
    blobs["data_priors"] = pt.numpy()
    blobs["torch_seed"] = np.array([99], dtype=np.int64)
    blobs["config"] = np.array([cfg.batch_size, cfg.n_training_epochs], dtype=np.int64)
    torch.manual_seed(99)
    model.train(Connect4Dataset(bt, vt, pt))
    for k, v in model.net.state_dict().items():
        blobs["final__" + k] = v.detach().clone().numpy()
    names = [k for k, _ in model.net.named_parameters()]
    for k, p in zip(names, model.net.parameters()):
        st = model.optimiser.state.get(p, {})
        if "momentum_buffer" in st and st["momentum_buffer"] is not None:
            blobs["momentum__" + k] = st["momentum_buffer"].detach().clone().numpy()
    blobs["lr_after"] = np.array([g["lr"] for g in model.optimiser.param_groups], dtype=np.float64)
    with torch.no_grad():
        xv, xp = model.net(bt[:32])
    blobs["eval_values"] = xv.numpy()
    blobs["eval_priors"] = xp.numpy()
    np.savez_compressed(os.path.join(OUT, "train_step.npz"), **blobs)


def main():
    if len(sys.argv) > 1 and sys.argv[1] == "train":
        dump_train_step()
        print("train_step.npz written to", OUT)
        return
    torch.manual_seed(0)
    torch.set_num_threads(1)  # deterministic accumulation order for the recorded net outputs
    ref_tests = dump_ref_tests()
    dump_board()
    dump_search_centre(ref_tests)
    model = ModelWrapper(ModelConfig(use_gpu=False), os.path.join(REF, "oinkoink", "data", "example_net.pth"))
    dump_net(model)
    dump_search_net(model)
    dump_selfplay(model)
    dump_train_step()
    print("golden fixtures written to", OUT)


if __name__ == "__main__":
    main()
