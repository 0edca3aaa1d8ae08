"""One training generation on the engine (BASELINE.json configs[4] shape): sharded self-play on every
rank's GPU -> optional all-gather of the finished games (the only collective) -> data.pth with flip
augmentation -> the reference's train recipe -> checkpoint.  Mirrors TrainingLoop._loop
(oinkoink/neural/training.py:78-153) with the hot path replaced; the surrounding bookkeeping
(evaluation sets, match history, visdom) stays with the reference's loop.
"""
import os
from typing import Optional

from .config import MCTSConfig
from .data import games_to_tensors, save_generation
from .distributed import generate_games_sharded
from .fused_net import FusedNet
from .training import ModelConfig, Trainer


def run_generation(trainer: Trainer, config: MCTSConfig, n_games: int, save_dir: Optional[str] = None, gen: int = 0,
                   seed: int = 0, device: int = 0, n_slots: Optional[int] = None):
    """Returns (games, last_loss).  With torch.distributed initialised every rank plays its shard and
    all ranks receive all games; training here is per-rank on the full set (the reference trains on one
    device, model.py:143-147)."""
    net = FusedNet(trainer.net.state_dict(), device=device)      # weights are fixed within a generation
    try:
        games = generate_games_sharded(config, net, n_games, seed=seed + 1000 * gen, device=device, n_slots=n_slots)
    finally:
        net.close()
    boards, values, priors = games_to_tensors(games, add_fliplr=True, device=device)
    if save_dir is not None:
        folder = os.path.join(save_dir, str(gen))                 # save_dir/<gen>/{data.pth, net.pth} (storage.py:15-16)
        save_generation(games, folder, device=device)
    loss = trainer.train(boards, values, priors)
    if save_dir is not None:
        trainer.save(os.path.join(save_dir, str(gen)))
    return games, loss
