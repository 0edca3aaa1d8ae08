"""N>1 path on CPU: world_size-2 gloo run of the shard arithmetic and the end-of-generation
all-gather of finished-game records (the only exchange in the design; none inside the rollouts)."""
import os
import socket

import numpy as np
import torch.multiprocessing as mp


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _fake_games(rank, count):
    from connect4_amd.board import Board
    from connect4_amd.training_game import GameData
    rng = np.random.RandomState(100 + rank)
    games = []
    for i in range(count):
        g, b = GameData(), Board()
        g.game_id = i
        while b.result is None:
            mv = int(rng.choice(sorted(b.valid_moves)))
            pol = rng.rand(7).astype(np.float32).astype(np.float64)
            g.add_move(b.__copy__(), mv, float(np.float32(rng.rand())) if rng.rand() > 0.1 else None, pol)
            b.make_move(mv)
        g.result = b.result
        games.append(g)
    return games


def _worker(rank, world, port, n_games, q):
    import torch.distributed as dist
    from connect4_amd.distributed import all_gather_games, rank_seed, shard_range
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    start, count = shard_range(n_games, rank, world)
    mine = _fake_games(rank, count)
    allg = all_gather_games(mine, id_offset=start)
    q.put((rank, start, count, rank_seed(7, rank), [(g.game_id, g.moves, g.result.value, g.values,
                                                       [b.to_int_tuple() for b in g.boards],
                                                       [p.tolist() for p in g.priors]) for g in allg]))
    dist.barrier()
    dist.destroy_process_group()


def test_shard_range_covers_everything():
    from connect4_amd.distributed import shard_range
    for n in (0, 1, 7, 8, 65536, 1201):
        for world in (1, 2, 3, 8):
            spans = [shard_range(n, r, world) for r in range(world)]
            assert spans[0][0] == 0 and sum(c for _, c in spans) == n
            for (s0, c0), (s1, _) in zip(spans, spans[1:]):
                assert s0 + c0 == s1
            assert max(c for _, c in spans) - min(c for _, c in spans) <= 1


def test_all_gather_games_world2():
    world, n_games = 2, 9
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, world, port, n_games, q)) for r in range(world)]
    for p in procs:
        p.start()
    outs = [q.get(timeout=120) for _ in range(world)]
    for p in procs:
        p.join(60)
        assert p.exitcode == 0
    outs.sort()
    assert [(o[1], o[2]) for o in outs] == [(0, 5), (5, 4)]
    assert [o[3] for o in outs] == [7, 8]
    assert outs[0][4] == outs[1][4]                      # every rank holds the same gathered list
    games = outs[0][4]
    assert [g[0] for g in games] == list(range(n_games))
    # content equals what each rank produced locally
    expect = []
    for r, (start, count) in enumerate([(0, 5), (5, 4)]):
        for g in _fake_games(r, count):
            expect.append((g.game_id + start, g.moves, g.result.value, g.values,
                           [b.to_int_tuple() for b in g.boards], [p.tolist() for p in g.priors]))
    assert games == expect
