"""Multi-GPU self-play: one process per GPU, games shard embarrassingly, NO collective inside the
rollout path.  The only exchange is an optional all-gather of finished-game
tensors at generation end (the reference's ``games.extend(game_batch)`` over pool results,
training.py:131); with backend "nccl" that is RCCL over xGMI, with "gloo" it runs on CPU (tests).
"""
from typing import Dict, List, Tuple

import numpy as np

from .board import Board
from .training_game import GameData
from .utils import CODE_FROM_RESULT, RESULT_FROM_CODE


def shard_range(n_games: int, rank: int, world: int) -> Tuple[int, int]:
    """GPU r of R plays games [start, start+count): contiguous, sizes differ by at most one."""
    base, rem = divmod(n_games, world)
    start = rank * base + min(rank, rem)
    return start, base + (1 if rank < rem else 0)


def rank_seed(seed: int, rank: int) -> int:
    return seed + rank           # disjoint Philox key per rank


def pack_games(games: List[GameData], id_offset: int = 0) -> Dict[str, np.ndarray]:
    """Compact record ~50 B/position: 2xu64 board, u8 move, f32 value, 7xf32 policy (+ per-game
    length/result/id)."""
    n = sum(len(g.moves) for g in games)
    out = dict(boards=np.zeros((n, 2), dtype=np.int64), moves=np.zeros(n, dtype=np.uint8),
               values=np.zeros(n, dtype=np.float32), policy=np.zeros((n, 7), dtype=np.float32),
               lengths=np.array([len(g.moves) for g in games], dtype=np.int32),
               results=np.array([CODE_FROM_RESULT[g.result] for g in games], dtype=np.int8),
               ids=np.array([g.game_id + id_offset for g in games], dtype=np.int64))
    i = 0
    for g in games:
        for b, m, v, p in zip(g.boards, g.moves, g.values, g.priors):
            out["boards"][i] = np.array([b.color[0], b.color[1]], dtype=np.uint64).view(np.int64)
            out["moves"][i] = m
            out["values"][i] = np.nan if v is None else v
            out["policy"][i] = p
            i += 1
    return out


def unpack_games(packed: Dict[str, np.ndarray]) -> List[GameData]:
    games, i = [], 0
    for length, res, gid in zip(packed["lengths"], packed["results"], packed["ids"]):
        g = GameData()
        g.game_id = int(gid)
        for _ in range(int(length)):
            c = packed["boards"][i].view(np.uint64)
            v = float(packed["values"][i])
            g.add_move(Board.from_bits(int(c[0]), int(c[1])), int(packed["moves"][i]),
                       None if np.isnan(v) else v, packed["policy"][i].astype(np.float64))
            i += 1
        g.result = RESULT_FROM_CODE[int(res)]
        games.append(g)
    return games


def all_gather_games(games: List[GameData], id_offset: int = 0, device=None) -> List[GameData]:
    """All-gather every rank's finished games (variable length: sizes first, then padded tensors).
    Works on any initialised torch.distributed backend; tensors live on `device` (cuda for RCCL)."""
    import torch
    import torch.distributed as dist
    world = dist.get_world_size()
    packed = pack_games(games, id_offset)
    dev = device if device is not None else ("cuda" if dist.get_backend() == "nccl" else "cpu")
    sizes = torch.tensor([len(packed["moves"]), len(packed["lengths"])], dtype=torch.int64, device=dev)
    all_sizes = [torch.zeros_like(sizes) for _ in range(world)]
    dist.all_gather(all_sizes, sizes)
    max_pos = int(max(s[0] for s in all_sizes))
    max_games = int(max(s[1] for s in all_sizes))
    gathered = {}
    for key, arr in packed.items():
        cap = max_games if key in ("lengths", "results", "ids") else max_pos
        pad = np.zeros((cap,) + arr.shape[1:], dtype=arr.dtype)
        pad[:len(arr)] = arr
        t = torch.from_numpy(pad).to(dev)
        outs = [torch.zeros_like(t) for _ in range(world)]
        dist.all_gather(outs, t)
        gathered[key] = [o.cpu().numpy() for o in outs]
    result: List[GameData] = []
    for r in range(world):
        npos, ngames = int(all_sizes[r][0]), int(all_sizes[r][1])
        part = {k: (v[r][:ngames] if k in ("lengths", "results", "ids") else v[r][:npos]) for k, v in gathered.items()}
        result.extend(unpack_games(part))
    result.sort(key=lambda g: g.game_id)
    return result


def generate_games_sharded(config, net, n_games: int, seed: int = 0, device: int = 0, gather: bool = True,
                           **kw) -> List[GameData]:
    """Each rank plays its shard on its own GPU with RNG stream seed+rank; optionally all-gather."""
    import torch.distributed as dist
    from .selfplay import generate_games
    rank = dist.get_rank() if dist.is_initialized() else 0
    world = dist.get_world_size() if dist.is_initialized() else 1
    start, count = shard_range(n_games, rank, world)
    games = generate_games(config, net, count, seed=rank_seed(seed, rank), device=device, **kw) if count else []
    if world == 1 or not gather:
        for g in games:
            g.game_id += start
        return games
    return all_gather_games(games, id_offset=start)
