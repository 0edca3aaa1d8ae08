#!/usr/bin/env python3
"""Time line of a FINITE self-play batch (one generation's share) on one GPU: engine creation, then every `--poll` quanta
the games finished, simulations, cache hit rate and active slots; at the end the batch's games/s against the steady-state
rate of the same slot count (bench.py).  python tools/gen_profile.py --games 1200 --slots 1200 [--precision f32x3]"""
import argparse
import json
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--games", type=int, default=1200)
    ap.add_argument("--slots", type=int, default=0, help="0 = min(games, 4096)")
    ap.add_argument("--sims", type=int, default=800)
    ap.add_argument("--poll", type=int, default=64)
    ap.add_argument("--precision", default=None)
    ap.add_argument("--series", type=int, default=0, help="1: print the time line")
    a = ap.parse_args()
    import torch
    from connect4_amd.config import MCTSConfig
    from connect4_amd.fused_net import FusedNet
    from connect4_amd.net import random_init_state_dict
    from connect4_amd.selfplay import SelfPlay
    slots = a.slots or min(a.games, 4096)
    net = FusedNet(random_init_state_dict(seed=0), precision=a.precision)
    # warm the runtime
    sp = SelfPlay(net, 64, MCTSConfig.self_play(32), seed=0, games_target=64, record_capacity_games=64, use_graph=False, fused_loop=True)
    sp.run_steps(64)
    sp.synchronize()
    sp.close()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    sp = SelfPlay(net, slots, MCTSConfig.self_play(a.sims), seed=1, games_target=a.games, record_capacity_games=a.games,
                  use_graph=False, fused_loop=True, max_inner_iters=32)
    torch.cuda.synchronize()
    t_create = time.perf_counter() - t0
    series = []
    prev = None
    while True:
        sp.run_steps(a.poll)
        st = sp.stats()
        t = time.perf_counter() - t0
        series.append((t, st["games_finished"], st["simulations"], st["eval_cache_hits"], st["eval_cache_probes"], st["active_slots"]))
        if st["active_slots"] == 0:
            break
    t_play = time.perf_counter() - t0 - t_create
    t1 = time.perf_counter()
    packed = sp.engine.export_games(a.games)
    torch.cuda.synchronize()
    t_export = time.perf_counter() - t1
    sp.close()
    half = next(t for t, g, *_ in series if g >= a.games // 2)
    out = {"games": a.games, "slots": slots, "sims": a.sims, "precision": net.precision, "engine_create_s": t_create, "play_s": t_play,
           "export_s": t_export, "total_s": t_create + t_play + t_export, "games_per_s": a.games / (t_create + t_play + t_export),
           "games_per_s_play_only": a.games / t_play, "time_to_half_the_games_s": half - t_create,
           "sims_per_s_play_only": series[-1][2] / t_play, "final_hit_rate": series[-1][3] / max(1, series[-1][4]),
           "exported": int(packed.n_games)}
    print(json.dumps(out))
    if a.series:
        for row in series:
            print("t=%.3f games=%d sims=%d hit=%.3f active=%d" % (row[0], row[1], row[2], row[3] / max(1, row[4]), row[5]))


if __name__ == "__main__":
    main()
