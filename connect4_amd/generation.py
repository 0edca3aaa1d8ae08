"""One training generation on the engine (BASELINE.json configs[4] shape): sharded self-play on every
rank's GPU -> device-side export of the finished games -> all-gather of the packed tensors (the only
collective, RCCL over xGMI) -> data.pth with flip augmentation built on the device -> the reference's
train recipe on the sliding window -> checkpoint.  Mirrors TrainingLoop._loop
(oinkoink/neural/training.py:78-153) with the hot path replaced; the surrounding bookkeeping
(evaluation sets, match history, visdom) stays with the reference's loop.
"""
import os
import time
from typing import Optional

from .config import MCTSConfig
from .data import TrainingDataStorage
from .distributed import generate_games_sharded_packed
from .fused_net import make_selfplay_net
from .training import Trainer


def run_generation(trainer: Trainer, config: MCTSConfig, n_games: int, save_dir: Optional[str] = None, gen: int = 0,
                   seed: int = 0, device: int = 0, n_slots: Optional[int] = None, write_games_pkl: bool = False,
                   timings: Optional[dict] = None):
    """Returns (PackedGames of all ranks, last_loss).  With torch.distributed initialised every rank plays
    its shard and all ranks receive all games; rank 0 writes save_dir/<gen>/{data.pth, games.pkl, net.pth}
    (storage.py:15-16, data.py:47-64); training runs on `trainer.device` over the window
    min(20, int((gen+1)/2)) generations (data.py:66-75) when save_dir holds them, else on this generation."""
    import torch
    import torch.distributed as dist
    rank = dist.get_rank() if dist.is_initialized() else 0
    t0 = time.perf_counter()
    net = make_selfplay_net(trainer.net.state_dict(), device=device)      # weights are fixed within a generation
    try:
        games = generate_games_sharded_packed(config, net, n_games, seed=seed + 1000 * gen, device=device, n_slots=n_slots)
    finally:
        if hasattr(net, "close"):
            net.close()
    torch.cuda.synchronize(device)
    t1 = time.perf_counter()
    boards, values, priors = games.training_tensors(add_fliplr=True)       # on the device
    storage = TrainingDataStorage(device=device, write_games_pkl=write_games_pkl)
    if save_dir is not None:
        folder = os.path.join(save_dir, str(gen))                 # save_dir/<gen>/{data.pth, net.pth} (storage.py:15-16)
        if rank == 0:
            os.makedirs(folder, exist_ok=True)
            if write_games_pkl:
                super(TrainingDataStorage, storage).save(games, folder)
            torch.save({"boards": boards.cpu(), "values": values.cpu(), "priors": priors.cpu()}, os.path.join(folder, "data.pth"))
        if dist.is_initialized():
            dist.barrier()
        from .data import window_generations
        if len(window_generations(gen)) > 1:
            boards, values, priors = storage.get_dataset(save_dir, gen)
    torch.cuda.synchronize(device)
    t2 = time.perf_counter()
    loss = trainer.train(boards, values, priors)
    if save_dir is not None and rank == 0:
        trainer.save(os.path.join(save_dir, str(gen)))
    if trainer.device.type == "cuda":
        torch.cuda.synchronize(trainer.device)
    t3 = time.perf_counter()
    if timings is not None:
        timings.update(selfplay_and_gather_s=t1 - t0, tensors_and_write_s=t2 - t1, train_s=t3 - t2,
                       positions=int(games.n_positions), training_rows=int(boards.shape[0]))
    return games, loss
