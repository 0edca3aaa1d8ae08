"""Multi-GPU self-play: one process per GPU, games shard embarrassingly, NO collective inside the
rollout path.  The only exchange is an optional all-gather of finished-game
tensors at generation end (the reference's ``games.extend(game_batch)`` over pool results,
training.py:131); with backend "nccl" that is RCCL over xGMI, with "gloo" it runs on CPU (tests).
"""
from typing import List, Tuple

from .training_game import GameData


def shard_range(n_games: int, rank: int, world: int) -> Tuple[int, int]:
    """GPU r of R plays games [start, start+count): contiguous, sizes differ by at most one."""
    base, rem = divmod(n_games, world)
    start = rank * base + min(rank, rem)
    return start, base + (1 if rank < rem else 0)


def rank_seed(seed: int, rank: int) -> int:
    return seed + rank           # disjoint Philox key per rank


def pack_games(games: List[GameData], id_offset: int = 0):
    """List[GameData] -> PackedGames (CPU tensors)."""
    from .packed import PackedGames
    return PackedGames.from_game_data(games, id_offset)


def unpack_games(packed) -> List[GameData]:
    return packed.to_game_data()


def all_gather_games(games: List[GameData], id_offset: int = 0, device=None) -> List[GameData]:
    """All-gather every rank's finished games given in the reference's object form (the device path is
    generate_games_sharded_packed: nothing is unpacked there).  Tensors live on `device` (cuda for RCCL)."""
    import torch.distributed as dist
    from .packed import all_gather_packed
    dev = device if device is not None else ("cuda" if dist.get_backend() == "nccl" else "cpu")
    out = all_gather_packed(pack_games(games, id_offset).to(dev)).sorted_by_id()
    return out.to_game_data()


def generate_games_sharded_packed(config, net, n_games: int, seed: int = 0, device: int = 0, gather: bool = True, **kw):
    """Each rank plays its shard on its own GPU with RNG stream seed+rank and exports the finished games on
    the device; the optional all-gather moves device tensors (RCCL over xGMI).  Returns PackedGames sorted
    by global game id, on the GPU."""
    import torch.distributed as dist
    from .packed import all_gather_packed
    from .selfplay import generate_games_packed
    rank = dist.get_rank() if dist.is_initialized() else 0
    world = dist.get_world_size() if dist.is_initialized() else 1
    start, count = shard_range(n_games, rank, world)
    packed = generate_games_packed(config, net, count, seed=rank_seed(seed, rank), device=device, **kw).offset_ids(start)
    if world > 1 and gather:
        if dist.get_backend() == "nccl":
            packed = all_gather_packed(packed)                      # device tensors over RCCL / xGMI
        else:                                                       # gloo (rehearsals, tests): through host memory
            packed = all_gather_packed(packed.cpu()).to(packed.device)
    return packed.sorted_by_id()


def generate_games_sharded(config, net, n_games: int, seed: int = 0, device: int = 0, gather: bool = True,
                           **kw) -> List[GameData]:
    """generate_games_sharded_packed in the reference's object form (List[GameData], training.py:131)."""
    return generate_games_sharded_packed(config, net, n_games, seed, device, gather, **kw).to_game_data()
