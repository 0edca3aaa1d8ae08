#!/usr/bin/env python3
"""Diagnostic: where a slot's time goes inside the split kernel's tree walk.  Needs a build with -DC4_SPLIT_PHASES=1
(hipcc ... -DC4_SPLIT_PHASES=1 -o x.so; C4_ENGINE_LIB=x.so python tools/split_phases.py [slots])."""
import ctypes as C
import os
import sys

import numpy as np

os.environ["C4_TREE_STAMPS"] = "1"
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from connect4_amd.config import MCTSConfig  # noqa: E402
from connect4_amd.fused_net import FusedNet  # noqa: E402
from connect4_amd.net import random_init_state_dict  # noqa: E402
from connect4_amd.selfplay import SelfPlay  # noqa: E402

slots = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
net = FusedNet(random_init_state_dict(seed=0), precision=os.environ.get("C4_NET_PRECISION", "f32x3"))
sp = SelfPlay(net, slots, MCTSConfig.self_play(800), seed=0, use_graph=False, fused_loop=True, steps_per_launch=256, max_inner_iters=32)
sp.run_steps(256 * 150)
sp.synchronize()
s0 = sp.stats()
sp.run_steps(256)
sp.synchronize()
s1 = sp.stats()
out = (C.c_uint64 * 2048)()
assert sp.engine._lib.c4_debug_stamps(sp.engine._h, out) == 0
a = np.array(list(out), dtype=np.float64).reshape(256, 8)
a = a[a[:, 4] > 0]
it, wait, app, lev, n = a[:, 0], a[:, 1], a[:, 2], a[:, 3], a[:, 4]
sims = (s1["simulations"] - s0["simulations"]) / slots
print("slots sampled %d; per slot and launch: %.0f walking iterations, %.0f simulations (all slots)" % (len(a), n.mean(), sims))
probe, term = a[:, 5], a[:, 6]
print("cycles per walking iteration: %.0f  (apply %.0f, level loop %.0f, cache probe %.0f, terminal backup %.0f, rest %.0f)" %
      ((it / n).mean(), (app / n).mean(), (lev / n).mean(), (probe / n).mean(), (term / n).mean(), ((it - app - lev - probe - term) / n).mean()))
print("share of the launch a slot spends waiting for the network: %.3f   (walking %.3f)" %
      ((wait / (it + wait)).mean(), (it / (it + wait)).mean()))
print("hit rate %.3f, mean leaf depth %.2f" % ((s1["eval_cache_hits"] - s0["eval_cache_hits"]) / max(1, s1["eval_cache_probes"] - s0["eval_cache_probes"]),
      (s1["depth_sum"] - s0["depth_sum"]) / max(1, s1["simulations"] - s0["simulations"])))

lat = (C.c_uint64 * 512)()
if sp.engine._lib.c4_debug_latency_stamps(sp.engine._h, lat) == 0:
    q = np.array(list(lat), dtype=np.float64).reshape(128, 4)
    q = q[q[:, 3] > 0]
    if len(q):
        n = q[:, 3].sum()
        print("a request's life (mean over %d requests of the last launch, cycles): posted -> claimed %.0f, claimed -> answered %.0f, answered -> picked up %.0f"
              % (n, 64 * q[:, 0].sum() / n, 64 * q[:, 1].sum() / n, 64 * q[:, 2].sum() / n))
