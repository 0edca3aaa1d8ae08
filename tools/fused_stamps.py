#!/usr/bin/env python3
"""Diagnostic: tree-phase vs net-phase cycles inside the fused self-play kernel (C4_TREE_STAMPS=1)."""
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

mi = int(sys.argv[1]) if len(sys.argv) > 1 else 4
net = FusedNet(random_init_state_dict(seed=0))
sp = SelfPlay(net, 4096, MCTSConfig.self_play(800), seed=0, use_graph=False, fused_loop=True, steps_per_launch=32,
              max_inner_iters=mi)
sp.run_steps(3200)
sp.synchronize()
sp.run_steps(32)
sp.synchronize()
out = (C.c_uint64 * 2048)()
assert sp.engine._lib.c4_debug_stamps(sp.engine._h, out) == 0
a = np.array(list(out), dtype=np.float64).reshape(256, 8)
n = a[:, 2].mean()
print("per step, mean over 256 workgroups (cycles): tree phase %.0f (p95 %.0f)  net phase %.0f   own-wave tree work: %s"
      % (a[:, 0].mean() / n, np.percentile(a[:, 0], 95) / n, a[:, 1].mean() / n,
         " ".join("%.0f" % (a[:, 3 + i].mean() / n) for i in range(5))))
