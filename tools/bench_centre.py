#!/usr/bin/env python3
"""Search-only benchmark (BASELINE config 4 shape): N searches x S simulations with the in-kernel
evaluate_centre_with_prior evaluator -- no network, every simulation of every search inside one launch.
Checks a sample of the searches against the CPU oracle."""
import argparse
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--games", type=int, default=8192)
    ap.add_argument("--sims", type=int, default=3200)
    ap.add_argument("--check", type=int, default=32)
    args = ap.parse_args()
    from connect4_amd import _lib as L
    from connect4_amd.engine import Engine
    from oracle import c4oracle as oc
    rng = np.random.RandomState(0)
    boards = []
    while len(boards) < args.games:
        b = oc.Board.empty()
        for _ in range(int(rng.randint(0, 12))):
            m = b.valid_mask()
            if not m:
                break
            b.make_move(int(rng.choice([c for c in range(7) if (m >> c) & 1])))
        if b.result == oc.NONE:
            boards.append(b)
    with Engine(args.games, args.sims, eval_mode=L.EVAL_CENTRE, stop_after_move=True) as eng:
        eng.reset([b.key()[0] for b in boards], [b.key()[1] for b in boards])
        t0 = time.perf_counter()
        eng.run_centre(max_launches=8)
        dt = time.perf_counter() - t0
        st = eng.stats()
        roots = eng.read_roots()
    print("centre search: %d searches x %d sims in %.3f s -> %.3e sims/s, %.3e expansions/s, mean depth %.2f"
          % (args.games, args.sims, dt, st["simulations"] / dt, st["expansions"] / dt, st["depth_sum"] / st["simulations"]))
    cfg = oc.make_config(args.sims)
    for i in rng.choice(args.games, size=min(args.check, args.games), replace=False):
        info, mv, _ = oc.search_and_pick(cfg, boards[i], oc.CentreEvaluator())
        assert list(roots[i].child_visits) == list(info.child_visits), i
        assert list(roots[i].child_value_sum) == list(info.child_value_sum), i
        assert roots[i].move == mv
    print("checked %d searches against the oracle: identical" % min(args.check, args.games))


if __name__ == "__main__":
    main()
