"""Container-only: the INTEGRATION.md drop-in executed against the UNMODIFIED reference's own call sites
(skipped where /root/reference is absent, e.g. on the GPU box -- nothing GPU-side reads the reference).

The reference imports with the two stand-in modules under oracle/refshim (anytree, visdom; no reference logic).
Checked: (1) the reference's TrainingDataStorage.save (data.py:52-64, storage.py:12-17) accepts this package's
GameData objects and writes the same data.pth tensors as the reference did for its own objects (golden
fixture) plus a games.pkl; (2) the W/D/L count of training.py:137-141 works on this package's results."""
import os
import sys

import numpy as np
import pytest

from conftest import ROOT, load_json, load_npz

REF = "/root/reference"
pytestmark = pytest.mark.skipif(not os.path.isdir(os.path.join(REF, "oinkoink")), reason="reference not present")


@pytest.fixture(scope="module")
def ref():
    added = [os.path.join(ROOT, "oracle", "refshim"), REF]
    sys.path[:0] = added
    sys.dont_write_bytecode = True
    try:
        import oinkoink.neural.pytorch.data as rdata
        import oinkoink.utils as rutils
        yield rdata, rutils
    finally:
        for p in added:
            sys.path.remove(p)


def _mirror_games():
    from connect4_amd.board import Board
    from connect4_amd.training_game import GameData
    from connect4_amd.utils import Result
    games = []
    for g in load_json("selfplay_net.json"):
        gd = GameData()
        for bb, mv, v, p in zip(g["boards"], g["moves"], g["values"], g["policies"]):
            gd.add_move(Board.from_bits(*bb), mv, v, np.array(p))
        gd.result = Result(g["result"])
        games.append((g["name"], gd))
    return games


def test_reference_storage_accepts_mirror_game_data(ref, tmp_path):
    import torch
    rdata, _ = ref
    npz = load_npz("selfplay_net_tables.npz")
    for name, gd in _mirror_games():
        folder = os.path.join(str(tmp_path), name)
        os.makedirs(folder)
        rdata.TrainingDataStorage().save([gd], folder)            # the reference's own writer, our objects
        d = torch.load(os.path.join(folder, "data.pth"), weights_only=True)
        assert np.array_equal(d["boards"].numpy().astype(np.uint8), npz[name + "__data_boards"])
        assert np.array_equal(d["values"].numpy(), npz[name + "__data_values"])
        assert np.array_equal(d["priors"].numpy(), npz[name + "__data_priors"])
        assert os.path.getsize(os.path.join(folder, "games.pkl")) > 0
        # and its dataset reader over a directory this package wrote the same way
        ds = rdata.Connect4Dataset.load(os.path.join(folder, "data.pth"))
        assert len(ds) == len(d["boards"])


def test_reference_result_count_works_on_mirror_results(ref):
    _, rutils = ref
    results = [gd.result for _, gd in _mirror_games()]
    counts = (results.count(rutils.Result.o_win), results.count(rutils.Result.draw), results.count(rutils.Result.x_win))
    assert sum(counts) == len(results) and sum(counts) > 0       # training.py:137-141 would print 0, 0, 0 otherwise
