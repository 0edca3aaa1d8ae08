#!/usr/bin/env python3
"""BASELINE config 5 at one GPU's share: 8192 self-play games (800 sims/move) -> device-side export of the
finished games -> training tensors with flip augmentation built on the device -> data.pth written ->
5 epochs x batch 4096 of the reference's train recipe on those positions -> checkpoint.  One JSON line.

    python tools/bench_generation.py [--games 8192] [--slots 4096] [--sims 800] [--out DIR]

(With 8 ranks every GPU plays this share and then trains on the all-gathered 65,536 games; the all-gather
of ~50 B/position packed tensors is the only collective, connect4_amd/packed.py:all_gather_packed.)"""
import argparse
import json
import os
import sys
import tempfile
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--games", type=int, default=8192)
    ap.add_argument("--slots", type=int, default=4096)
    ap.add_argument("--sims", type=int, default=800)
    ap.add_argument("--out", default=None)
    ap.add_argument("--games-pkl", type=int, default=0, help="1: also write the object form (games.pkl)")
    a = ap.parse_args()
    import torch
    import __graft_entry__ as entry
    entry.build()
    from connect4_amd.config import MCTSConfig
    from connect4_amd.generation import run_generation
    from connect4_amd.training import ModelConfig, Trainer
    torch.manual_seed(0)
    tr = Trainer(ModelConfig(), device="cuda:0")
    out = a.out or tempfile.mkdtemp(prefix="c4gen_")
    # warm the runtime (first launches, allocator) on a toy generation that is not timed
    run_generation(tr, MCTSConfig.self_play(32), n_games=64, save_dir=None, gen=0, n_slots=64)
    # ... and MIOpen's one-time kernel build for the training shape (batch 4096 forward/backward: ~15 s in a fresh
    # process, cached afterwards) on a throw-away trainer.  The ragged LAST batch of an epoch is no new shape: the
    # trainer pads it to the full batch and takes the batch-norm statistics from the real rows (net._BatchNorm2d).
    warm = Trainer(ModelConfig(n_training_epochs=1), device="cuda:0")
    wn = 2 * 4096
    warm.train((torch.rand(wn, 3, 6, 7, device="cuda") > 0.7).float(), torch.rand(wn, device="cuda"),
               torch.softmax(torch.rand(wn, 7, device="cuda"), 1))
    del warm
    torch.cuda.synchronize()
    timings = {}
    t0 = time.perf_counter()
    games, loss = run_generation(tr, MCTSConfig.self_play(a.sims), n_games=a.games, save_dir=out, gen=0, n_slots=a.slots,
                                 write_games_pkl=bool(a.games_pkl), timings=timings)
    torch.cuda.synchronize()
    total = time.perf_counter() - t0
    res = {"workload": "one GPU's share of BASELINE config 5: %d self-play games at %d sims/move on %d slots, export + data.pth "
                       "(flip-augmented) + %d epochs x batch %d + checkpoint" % (a.games, a.sims, a.slots, tr.config.n_training_epochs, tr.config.batch_size),
           "games": int(games.n_games), "positions": int(games.n_positions), "training_rows": timings["training_rows"],
           "total_s": total, "selfplay_export_s": timings["selfplay_and_gather_s"], "tensors_and_data_pth_s": timings["tensors_and_write_s"],
           "train_and_checkpoint_s": timings["train_s"], "games_per_s_end_to_end": a.games / total,
           "games_per_s_selfplay": a.games / timings["selfplay_and_gather_s"], "last_loss": loss,
           "data_pth_bytes": os.path.getsize(os.path.join(out, "0", "data.pth"))}
    print(json.dumps(res))


if __name__ == "__main__":
    main()
