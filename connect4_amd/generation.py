"""One training generation on the engine (BASELINE.json configs[4] shape): sharded self-play on every
rank's GPU -> device-side export of the finished games -> all-gather of the packed tensors (RCCL over xGMI)
-> on rank 0: data.pth with flip augmentation built on the device, the reference's train recipe on the sliding
window, checkpoint -> broadcast of the trained state to every rank (one model, as in the reference).  Mirrors TrainingLoop._loop
(oinkoink/neural/training.py:78-153) with the hot path replaced; the surrounding bookkeeping
(evaluation sets, match history, visdom) stays with the reference's loop.
"""
import os
import time
from typing import Optional

from .config import MCTSConfig
from .data import GameStorage
from .distributed import generate_games_sharded_packed
from .fused_net import make_selfplay_net
from .training import Trainer


def existing_window(save_dir: str, gen: int):
    """Earlier generations of the sliding window (data.py:66-75) whose data.pth is there, newest first.  The reference
    concatenates all of window_generations(gen) and raises when one is missing; a run that starts or resumes at gen >= 3 in
    a directory without them trains on what exists (decided BEFORE self-play is paid for)."""
    from .data import window_generations
    return [g for g in window_generations(gen)[1:] if os.path.exists(os.path.join(save_dir, str(g), "data.pth"))]


def run_generation(trainer: Trainer, config: MCTSConfig, n_games: int, save_dir: Optional[str] = None, gen: int = 0,
                   seed: int = 0, device: int = 0, n_slots: Optional[int] = None, write_games_pkl: bool = False,
                   timings: Optional[dict] = None, precision: Optional[str] = None):
    """Returns (PackedGames of all ranks, last_loss).  With torch.distributed initialised every rank plays its shard and
    all ranks receive all games.  There is ONE model, as in the reference (training.py:147-153, model.py:143-147): rank 0
    writes save_dir/<gen>/{data.pth, games.pkl} (storage.py:15-16, data.py:47-64), trains on `trainer.device` over the
    window min(20, int((gen+1)/2)) generations (data.py:66-75; the earlier generations that save_dir holds), writes
    net.pth, and then every rank receives rank 0's net / optimiser / scheduler state (broadcast), so the next
    generation's shards are all played by the same net.  `precision`: see make_selfplay_net (None = the reference's)."""
    import torch
    import torch.distributed as dist
    multi = dist.is_initialized() and dist.get_world_size() > 1
    rank = dist.get_rank() if dist.is_initialized() else 0
    earlier = existing_window(save_dir, gen) if (save_dir is not None and rank == 0) else []
    t0 = time.perf_counter()
    net = make_selfplay_net(trainer.net.state_dict(), device=device, precision=precision)   # weights are fixed within a generation
    try:
        games = generate_games_sharded_packed(config, net, n_games, seed=seed + 1000 * gen, device=device, n_slots=n_slots)
    finally:
        if hasattr(net, "close"):
            net.close()
    torch.cuda.synchronize(device)
    t1 = time.perf_counter()
    loss, rows = None, 0
    t2 = t1
    if rank == 0:
        boards, values, priors = games.training_tensors(add_fliplr=True)       # on the device
        if save_dir is not None:
            folder = os.path.join(save_dir, str(gen))                 # save_dir/<gen>/{data.pth, net.pth} (storage.py:15-16)
            os.makedirs(folder, exist_ok=True)
            if write_games_pkl:
                GameStorage().save(games, folder)
            torch.save({"boards": boards.cpu(), "values": values.cpu(), "priors": priors.cpu()}, os.path.join(folder, "data.pth"))
            if earlier:   # data.py:66-75: this generation first, then the earlier ones, newest first
                parts = [torch.load(os.path.join(save_dir, str(g), "data.pth"), weights_only=True) for g in earlier]
                boards = torch.cat([boards] + [p["boards"].to(boards.device) for p in parts])
                values = torch.cat([values] + [p["values"].to(values.device) for p in parts])
                priors = torch.cat([priors] + [p["priors"].to(priors.device) for p in parts])
        torch.cuda.synchronize(device)
        t2 = time.perf_counter()
        rows = int(boards.shape[0])
        loss = trainer.train(boards, values, priors)
        if save_dir is not None:
            trainer.save(os.path.join(save_dir, str(gen)))
        if trainer.device.type == "cuda":
            torch.cuda.synchronize(trainer.device)
    if multi:
        loss = trainer.broadcast_state(src=0, extra=loss)
    t3 = time.perf_counter()
    if timings is not None:
        timings.update(selfplay_and_gather_s=t1 - t0, tensors_and_write_s=t2 - t1, train_s=t3 - t2,
                       positions=int(games.n_positions), training_rows=rows)
    return games, loss
