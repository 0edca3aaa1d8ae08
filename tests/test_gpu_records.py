"""Finished-game records on the GPU: per-slot staging + ring of finished games (no two live games can
share a row, overflow is counted, never silent), the one-copy host drain, the device-side export into
packed tensors, the on-device training tensors, and the evaluation-cache read-out."""
import numpy as np
import pytest

from conftest import load_json, load_npz

pytestmark = pytest.mark.gpu


def _selfplay(n_slots, sims, games_target, rec_cap, seed=3, precision=None, **kw):
    from connect4_amd.config import MCTSConfig
    from connect4_amd.fused_net import FusedNet
    from connect4_amd.net import random_init_state_dict
    from connect4_amd.selfplay import SelfPlay
    net = FusedNet(random_init_state_dict(seed=0), precision=precision)
    sp = SelfPlay(net, n_slots, MCTSConfig.self_play(sims), seed=seed, games_target=games_target,
                  record_capacity_games=rec_cap, use_graph=False, fused_loop=True, steps_per_launch=16, **kw)
    return sp, net


def _check_game_on_oracle(oracle, rec):
    """A drained record is ONE legal, complete game: boards chain through the recorded moves to the result."""
    b = oracle.Board.empty()
    for i in range(rec.length):
        assert b.key() == (int(rec.color0[i]), int(rec.color1[i])), "plies of two games mixed in one record"
        assert (b.valid_mask() >> rec.move[i]) & 1
        assert abs(sum(rec.policy[i]) - 1.0) < 1e-9
        b.make_move(int(rec.move[i]))
    assert b.result == rec.result and rec.result in (0, 1, 2)


def test_continuous_mode_periodic_drain_never_mixes_games(oracle):
    """games_target=-1 (continuous), a ring far smaller than the number of games played, short and long games
    in flight together, periodic drains: every drained game replays on the oracle, ids are unique, and
    finished == drained + waiting + dropped at every point."""
    sp, net = _selfplay(64, 20, -1, 48)
    seen, total_dropped = {}, 0
    for it in range(60):
        sp.run_steps(48)
        if it % 3 == 2:
            for r in sp.engine.drain_games():
                assert r.game_id not in seen
                seen[r.game_id] = r.length
                _check_game_on_oracle(oracle, r)
        st = sp.stats()
        ready, dropped = sp.engine.finished_games()
        assert st["dropped_games"] == dropped
        assert st["games_finished"] == len(seen) + ready + dropped
        total_dropped = dropped
    assert len(seen) > 200 and len(set(seen.values())) > 5      # many games, many different lengths
    # without draining the ring fills up: further games are counted as dropped, the rows stay intact
    for _ in range(40):
        sp.run_steps(48)
    ready, dropped = sp.engine.finished_games()
    assert ready == 48 and dropped > total_dropped
    for r in sp.engine.drain_games():
        _check_game_on_oracle(oracle, r)
    sp.close()
    net.close()


def test_device_export_equals_host_drain():
    """Same seed twice: games read through c4_drain_games (float64) and through c4_export_games_dev (packed
    float32 tensors written by the device) are the same games; the on-device training tensors equal the host
    writer's (which is pinned to the reference's native_to_pytorch by test_data_writer_matches_reference...)."""
    import torch
    from connect4_amd.data import games_to_arrays
    from connect4_amd.packed import PackedGames
    outs = []
    for mode in ("drain", "export"):
        sp, net = _selfplay(48, 24, 120, 120, seed=9)
        for _ in range(400):
            sp.run_steps(64)
            if sp.stats()["active_slots"] == 0:
                break
        assert sp.stats()["dropped_games"] == 0
        if mode == "drain":
            outs.append(sorted(sp.drain(), key=lambda g: g.game_id))
        else:
            first = sp.engine.export_games(max_games=50)          # two partial exports: whole games only
            rest = sp.engine.export_games()
            assert first.n_games == 50 and rest.n_games == 70 and sp.engine.finished_games()[0] == 0
            outs.append(PackedGames.cat([first, rest]).sorted_by_id())
        sp.close()
        net.close()
    games, packed = outs
    assert packed.n_games == 120 and packed.ids.tolist() == list(range(120))
    assert packed.lengths.tolist() == [len(g.moves) for g in games]
    assert packed.results.tolist() == [int(g.result.value * 2) for g in games]
    ref = PackedGames.from_game_data(games)
    assert torch.equal(packed.boards.cpu(), ref.boards)
    assert torch.equal(packed.moves.cpu(), ref.moves)
    assert torch.equal(packed.values.cpu().nan_to_num(nan=-1.0), ref.values.nan_to_num(nan=-1.0))
    assert torch.equal(packed.policy.cpu(), ref.policy)
    assert torch.equal(packed.targets.cpu(), ref.targets)
    assert torch.equal(packed.game_index.cpu(), ref.game_index)
    # object form round trip
    again = packed.to_game_data()
    assert [g.moves for g in again] == [g.moves for g in games]
    assert [[b.to_int_tuple() for b in g.boards] for g in again] == [[b.to_int_tuple() for b in g.boards] for g in games]
    # training tensors built on the device == the host writer
    b, v, p = packed.training_tensors(add_fliplr=True)
    hb, hv, hp = games_to_arrays(games, add_fliplr=True)
    assert np.array_equal(b.cpu().numpy(), hb) and np.array_equal(v.cpu().numpy(), hv) and np.array_equal(p.cpu().numpy(), hp)


def test_training_tensors_dev_against_reference_fixture():
    """c4_training_tensors_dev vs native_to_pytorch(add_fliplr=True) of the unmodified reference (data.py:78-105)."""
    import torch
    from connect4_amd.board import Board
    from connect4_amd.packed import PackedGames
    from connect4_amd.training_game import GameData
    from connect4_amd.utils import Result
    npz = load_npz("selfplay_net_tables.npz")
    for g in load_json("selfplay_net.json"):
        gd = GameData()
        for bb, mv, v, p in zip(g["boards"], g["moves"], g["values"], g["policies"]):
            gd.add_move(Board.from_bits(*bb), mv, v, np.array(p))
        gd.result = Result(g["result"])
        gd.game_id = 0
        b, v, p = PackedGames.from_game_data([gd]).to(torch.device("cuda", 0)).training_tensors(True)
        assert np.array_equal(b.cpu().numpy().astype(np.uint8), npz[g["name"] + "__data_boards"])
        assert np.array_equal(v.cpu().numpy(), npz[g["name"] + "__data_values"])
        assert np.array_equal(p.cpu().numpy(), npz[g["name"] + "__data_priors"])


def test_eval_cache_lookup_returns_the_nets_answers():
    """c4_eval_cache_lookup: every root of a finished game was evaluated, so (unless evicted) the cache
    answers for it -- with exactly the bits the wave-private forward produces; unknown positions are absent."""
    sp, net = _selfplay(32, 24, 64, 64, seed=4, eval_cache_log2_entries=22)   # roomy table: hardly any collision
    for _ in range(400):
        sp.run_steps(64)
        if sp.stats()["active_slots"] == 0:
            break
    recs = sp.engine.drain_games()
    c0 = np.array([r.color0[i] for r in recs for i in range(r.length)], dtype=np.uint64)
    c1 = np.array([r.color1[i] for r in recs for i in range(r.length)], dtype=np.uint64)
    v, p, found = sp.engine.cache_lookup(c0, c1)
    assert found.mean() > 0.95
    nv, npri = net.evaluate_bits(c0, c1, wave=True)
    assert np.array_equal(v[found], nv[found]) and np.array_equal(p[found], npri[found])
    # a position no game reaches under gravity rules is never in the table
    _, _, f2 = sp.engine.cache_lookup(np.array([1 << 5], dtype=np.uint64), np.array([1 << 40], dtype=np.uint64))
    assert not f2[0]
    sp.close()
    net.close()


def test_speculative_evaluations_are_the_nets_answers(oracle):
    """With the fp16 net the split kernel's network waves evaluate positions ahead of the search (the best-prior child of a
    position they just answered) and insert them into the evaluation cache.  That is invisible to the search only if such an
    entry holds exactly what the evaluator answers for that position: every cached child of every recorded root -- entries
    written by real requests and by speculative passes alike -- must equal the wave-private forward bit for bit.  (With the
    reference-precision net the network waves do not speculate: a pass is too long to put in front of a real request.)"""
    sp, net = _selfplay(32, 24, 64, 64, seed=9, precision="f16", eval_cache_log2_entries=22)
    for _ in range(400):
        sp.run_steps(64)
        if sp.stats()["active_slots"] == 0:
            break
    st = sp.stats()
    assert st["speculative_evals"] > 0
    recs = sp.engine.drain_games()
    kids0, kids1 = [], []
    for r in recs:
        for i in range(r.length):
            b = oracle.Board.from_bits(int(r.color0[i]), int(r.color1[i]))
            m = b.valid_mask()
            for c in range(7):
                if (m >> c) & 1:
                    k = b.copy()
                    k.make_move(c)
                    kids0.append(k.key()[0])
                    kids1.append(k.key()[1])
    c0, c1 = np.array(kids0, dtype=np.uint64), np.array(kids1, dtype=np.uint64)
    v, p, found = sp.engine.cache_lookup(c0, c1)
    assert found.sum() > 1000
    nv, npri = net.evaluate_bits(c0, c1, wave=True)
    assert np.array_equal(v[found], nv[found]) and np.array_equal(p[found], npri[found])
    sp.close()
    net.close()


def _sharded_worker(rank, world, port, q):
    import os
    import torch.distributed as dist
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from connect4_amd.config import MCTSConfig
    from connect4_amd.distributed import generate_games_sharded_packed
    from connect4_amd.fused_net import FusedNet
    from connect4_amd.net import random_init_state_dict
    net = FusedNet(random_init_state_dict(seed=0))
    p = generate_games_sharded_packed(MCTSConfig.self_play(16), net, 21, seed=5, device=0, n_slots=16)
    q.put((rank, p.ids.tolist(), p.lengths.tolist(), p.moves.cpu().tolist(), str(p.device)))
    dist.barrier()
    dist.destroy_process_group()


def test_sharded_generation_two_ranks_one_card():
    """BASELINE configs[2] shape in miniature: two ranks (both on this card, gloo) play disjoint shards with RNG
    streams seed+rank, export on the device and all-gather the packed tensors: every rank ends with all 21 games,
    sorted by global id, identical on both ranks; the shards are the ones a single rank plays with those seeds."""
    import socket
    import torch.multiprocessing as mp
    from connect4_amd.config import MCTSConfig
    from connect4_amd.distributed import rank_seed, shard_range
    from connect4_amd.fused_net import FusedNet
    from connect4_amd.net import random_init_state_dict
    from connect4_amd.selfplay import generate_games_packed
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_sharded_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    outs = sorted(q.get(timeout=300) for _ in range(2))
    for p in procs:
        p.join(120)
        assert p.exitcode == 0
    assert outs[0][1:4] == outs[1][1:4] and outs[0][1] == list(range(21)) and outs[0][4].startswith("cuda")
    net = FusedNet(random_init_state_dict(seed=0))
    want_moves = []
    for r in range(2):
        start, count = shard_range(21, r, 2)
        part = generate_games_packed(MCTSConfig.self_play(16), net, count, seed=rank_seed(5, r), n_slots=16)
        want_moves += part.moves.cpu().tolist()
    assert outs[0][3] == want_moves


def _generation_worker(rank, world, port, save_dir, q):
    import hashlib
    import os
    import torch
    import torch.distributed as dist
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from connect4_amd.config import MCTSConfig
    from connect4_amd.generation import run_generation
    from connect4_amd.net import NetConfig
    from connect4_amd.training import ModelConfig, Trainer
    torch.manual_seed(50 + rank)      # the ranks' trainers start from DIFFERENT weights: only the broadcast can make them equal
    tr = Trainer(ModelConfig(net_config=NetConfig(n_residuals=1), batch_size=64, n_training_epochs=1), device="cuda:0")
    start = tr.state()
    h0 = hashlib.sha256(b"".join(v.numpy().tobytes() for v in start["net_state_dict"].values())).hexdigest()
    tr.broadcast_state(src=0)         # generation 0 is played by ONE net on every rank (as after a previous generation's broadcast)
    out = []
    for gen in range(2):
        games, loss = run_generation(tr, MCTSConfig.self_play(16), n_games=26, save_dir=save_dir, gen=gen, seed=3, device=0, n_slots=16)
        st = tr.state()
        blob = b"".join(v.numpy().tobytes() for v in st["net_state_dict"].values())
        blob += b"".join(ps["momentum_buffer"].numpy().tobytes() for ps in st["optimiser_state_dict"]["state"].values())
        out.append((int(games.n_games), games.ids.tolist(), loss, hashlib.sha256(blob).hexdigest(), st["scheduler_state_dict"]["last_epoch"]))
    q.put((rank, h0, out))
    dist.barrier()
    dist.destroy_process_group()


def test_two_rank_generations_end_with_one_model(tmp_path):
    """run_generation on two ranks (both on this card, gloo).  Every rank plays its shard and receives all
    games; rank 0 trains and writes data.pth / net.pth; afterwards EVERY rank holds rank 0's net, batch-norm statistics and
    momentum buffers bit for bit -- over two generations, so the second generation's shards are played by one net."""
    import os
    import socket
    import torch
    import torch.multiprocessing as mp
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_generation_worker, args=(r, 2, port, str(tmp_path), q)) for r in range(2)]
    for p in procs:
        p.start()
    outs = sorted(q.get(timeout=600) for _ in range(2))
    for p in procs:
        p.join(120)
        assert p.exitcode == 0
    (r0, h0a, a), (r1, h0b, b) = outs
    assert h0a != h0b                                   # they really started apart
    for ga, gb in zip(a, b):
        assert ga[0] == gb[0] == 26 and ga[1] == gb[1] == list(range(26))
        assert ga[2] is not None and ga[2] == gb[2]     # the loss travels with the state
        assert ga[3] == gb[3] and ga[4] == gb[4]        # net + momentum: identical bytes; same scheduler step
    assert a[0][3] != a[1][3]                           # ... and training changed them between the generations
    for gen in (0, 1):
        assert os.path.exists(os.path.join(str(tmp_path), str(gen), "data.pth"))
        ck = torch.load(os.path.join(str(tmp_path), str(gen), "net.pth"), weights_only=True)
        assert set(ck) == {"net_state_dict", "optimiser_state_dict", "scheduler_state_dict"}


def _rccl_worker(port, q):
    import os
    import torch
    import torch.distributed as dist
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    dist.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device("cuda", 0))
    from connect4_amd.config import MCTSConfig
    from connect4_amd.distributed import generate_games_sharded_packed
    from connect4_amd.fused_net import FusedNet
    from connect4_amd.net import random_init_state_dict
    from connect4_amd.packed import all_gather_packed
    net = FusedNet(random_init_state_dict(seed=0))
    p = generate_games_sharded_packed(MCTSConfig.self_play(16), net, 12, seed=5, device=0, n_slots=12, gather=False)
    g = all_gather_packed(p)                      # RCCL all_gather of every field's dtype (int64, uint8, int8, int32, float32)
    ok = all(torch.equal(getattr(p, k), getattr(g, k)) for k in ("boards", "moves", "values", "policy", "targets", "lengths", "results", "ids"))
    t = torch.ones(3, device="cuda")
    dist.all_reduce(t)
    dist.barrier()
    q.put((ok, g.n_games, str(g.device), dist.get_backend(), float(t.sum())))
    dist.destroy_process_group()


def test_packed_all_gather_runs_on_rccl():
    """The generation-end collective on the real backend (RCCL; one rank is all this box has): every dtype of the packed
    record goes through dist.all_gather on device tensors and comes back unchanged."""
    import socket
    import torch.multiprocessing as mp
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    p = ctx.Process(target=_rccl_worker, args=(port, q))
    p.start()
    ok, n, dev, backend, red = q.get(timeout=300)
    p.join(120)
    assert p.exitcode == 0
    assert ok and n == 12 and dev.startswith("cuda") and backend == "nccl" and red == 3.0
