#!/usr/bin/env python3
"""Diagnostic: tree-phase vs net-phase cycles inside the workgroup-synchronous fused self-play kernel
(C4_FUSED_MODE=block, C4_TREE_STAMPS=1).  For the default wave-autonomous kernel see wave_stamps.py."""
import ctypes as C
import os
import sys

import numpy as np

os.environ["C4_TREE_STAMPS"] = "1"
os.environ["C4_FUSED_MODE"] = "block"
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from connect4_amd.config import MCTSConfig  # noqa: E402
from connect4_amd.fused_net import FusedNet  # noqa: E402
from connect4_amd.net import random_init_state_dict  # noqa: E402
from connect4_amd.selfplay import SelfPlay  # noqa: E402

mi = int(sys.argv[1]) if len(sys.argv) > 1 else 32
budget = int(sys.argv[2]) if len(sys.argv) > 2 else 80000
net = FusedNet(random_init_state_dict(seed=0), precision=os.environ.get("C4_NET_PRECISION", "f32x3"))
sp = SelfPlay(net, 4096, MCTSConfig.self_play(800), seed=0, use_graph=False, fused_loop=True, steps_per_launch=32,
              max_inner_iters=mi, time_budget_cycles=budget)
sp.run_steps(3200)
sp.synchronize()
sp.run_steps(32)
sp.synchronize()
out = (C.c_uint64 * 2048)()
assert sp.engine._lib.c4_debug_stamps(sp.engine._h, out) == 0
a = np.array(list(out), dtype=np.float64).reshape(128, 16)   # rows: 8 per-wave own sums, tree, net, n_steps
n = a[:, 10].mean()
print("32 steps per launch, mean over 128 workgroups (cycles/step): tree phase %.0f (p95 %.0f)  net phase %.0f" %
      (a[:, 8].mean() / n, np.percentile(a[:, 8], 95) / n, a[:, 9].mean() / n))
print("  own tree work per wave:", " ".join("%.0f" % (a[:, w].mean() / n) for w in range(8)))
sp._steps_per_launch = 1
rows = []
for _ in range(60):
    sp.run_steps(1)
    sp.synchronize()
    assert sp.engine._lib.c4_debug_stamps(sp.engine._h, out) == 0
    rows.append(np.array(list(out), dtype=np.float64).reshape(128, 16))
a = np.stack(rows)
own, tree, netc = a[:, :, 0:8], a[:, :, 8], a[:, :, 9]
print("one step per launch, 60 steps x 128 workgroups (cycles):")
print("  own tree work per wave: mean %.0f  p95 %.0f  max %.0f" % (own.mean(), np.percentile(own, 95), own.max()))
print("  slowest wave of the workgroup: mean %.0f  p95 %.0f" % (own.max(axis=2).mean(), np.percentile(own.max(axis=2), 95)))
print("  tree phase incl. barrier: mean %.0f   net phase: mean %.0f" % (tree.mean(), netc.mean()))
