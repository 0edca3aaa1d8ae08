"""Host logic of the training-data storage (data.py:47-75, storage.py:11-22) -- no GPU needed: the sliding
window, data.pth round trip, games.pkl, PackedGames <-> GameData, Result equality with a foreign enum."""
import os
from enum import Enum

import numpy as np
import torch


def test_window_matches_reference_formula():
    from connect4_amd.data import window_generations
    for gen in range(0, 60):
        n = min(20, int((gen + 1) / 2))                     # data.py:69
        assert window_generations(gen) == list(range(gen, gen - n, -1))
    assert window_generations(0) == [] and window_generations(1) == [1] and window_generations(2) == [2]
    assert window_generations(3) == [3, 2] and len(window_generations(100)) == 20


def test_get_dataset_concatenates_newest_first(tmp_path):
    from connect4_amd.data import TrainingDataStorage
    st = TrainingDataStorage()
    for gen in range(6):
        os.makedirs(os.path.join(str(tmp_path), str(gen)))
        n = gen + 1
        torch.save({"boards": torch.full((n, 3, 6, 7), float(gen)), "values": torch.full((n,), float(gen)),
                    "priors": torch.full((n, 7), float(gen))}, st.td_file_name(str(tmp_path), gen))
    b, v, p = st.get_dataset(str(tmp_path), 5)          # n = min(20, int(6/2)) = 3 -> generations 5, 4, 3
    assert v.tolist() == [5.0] * 6 + [4.0] * 5 + [3.0] * 4
    assert b.shape == (15, 3, 6, 7) and p.shape == (15, 7)


def _games(n=5, seed=0):
    from connect4_amd.board import Board
    from connect4_amd.training_game import GameData
    rng = np.random.RandomState(seed)
    out = []
    for i in range(n):
        g, b = GameData(), Board()
        g.game_id = 10 - i          # unsorted ids
        while b.result is None:
            mv = int(rng.choice(sorted(b.valid_moves)))
            g.add_move(b.__copy__(), mv, float(np.float32(rng.rand())) if rng.rand() > 0.1 else None,
                       rng.rand(7).astype(np.float32).astype(np.float64))
            b.make_move(mv)
        g.result = b.result
        out.append(g)
    return out


def test_packed_games_round_trip_sort_and_cat():
    from connect4_amd.packed import PackedGames
    games = _games()
    p = PackedGames.from_game_data(games)
    assert p.n_games == 5 and p.n_positions == sum(len(g.moves) for g in games)
    back = p.to_game_data()
    for a, b in zip(back, games):
        assert (a.game_id, a.moves, a.result, a.values) == (b.game_id, b.moves, b.result, b.values)
        assert [x.to_int_tuple() for x in a.boards] == [x.to_int_tuple() for x in b.boards]
        assert [x.tolist() for x in a.priors] == [x.tolist() for x in b.priors]
    s = p.sorted_by_id()
    assert s.ids.tolist() == sorted(g.game_id for g in games)
    by_id = {g.game_id: g for g in games}
    for g in s.to_game_data():
        assert g.moves == by_id[g.game_id].moves
    both = PackedGames.cat([p, PackedGames.from_game_data(_games(3, seed=1), id_offset=100)])
    assert both.n_games == 8 and both.game_index.max().item() == 7
    assert [g.moves for g in both.to_game_data()[:5]] == [g.moves for g in games]
    assert PackedGames.cat([]).n_games == 0
    try:
        p.training_tensors()
        raise AssertionError("CPU tensors must be refused: the training tensors are built by the HIP kernel")
    except RuntimeError as e:
        assert "no CPU fallback" in str(e)


def test_games_pkl_round_trip(tmp_path):
    from connect4_amd.data import GameStorage, load_games
    games = _games()
    st = GameStorage()
    st.save(games, str(tmp_path))
    assert os.path.exists(os.path.join(str(tmp_path), "games.pkl"))
    back = load_games(str(tmp_path))
    assert [g.moves for g in back] == [g.moves for g in games] and [g.result for g in back] == [g.result for g in games]
    assert "Move: %d" % games[-1].moves[0] in st.last_game_str()


def test_result_equals_a_foreign_result_enum():
    """training.py:137-141 counts oinkoink.utils.Result members in a list of this package's results."""
    from connect4_amd.utils import Result

    class Foreign(Enum):
        o_win = 1.0
        x_win = 0.0
        draw = 0.5
    Foreign.__name__ = "Result"
    mine = [Result.o_win, Result.draw, Result.o_win, Result.x_win]
    assert mine.count(Foreign.o_win) == 2 and mine.count(Foreign.draw) == 1 and mine.count(Foreign.x_win) == 1
    assert Result.o_win != Foreign.draw and Result.o_win == Result.o_win and Result.o_win != 1.0
    assert {Result.o_win: 1}[Result.o_win] == 1 and Result(0.5) is Result.draw
