#!/usr/bin/env python3
"""Diagnostic: per-wave tree / network cycles inside the wave-autonomous fused kernel (C4_TREE_STAMPS=1)."""
import ctypes as C
import os
import sys

import numpy as np

os.environ["C4_TREE_STAMPS"] = "1"
os.environ["C4_FUSED_MODE"] = "wave"
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from connect4_amd.config import MCTSConfig  # noqa: E402
from connect4_amd.fused_net import FusedNet  # noqa: E402
from connect4_amd.net import random_init_state_dict  # noqa: E402
from connect4_amd.selfplay import SelfPlay  # noqa: E402

slots = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
mi = int(sys.argv[2]) if len(sys.argv) > 2 else 32
net = FusedNet(random_init_state_dict(seed=0), precision=os.environ.get("C4_NET_PRECISION", "f32x3"))
sp = SelfPlay(net, slots, MCTSConfig.self_play(800), seed=0, use_graph=False, fused_loop=True, steps_per_launch=64,
              max_inner_iters=mi)
sp.run_steps(int(sys.argv[3]) if len(sys.argv) > 3 else 6400)
sp.synchronize()
s0 = sp.stats()
sp.run_steps(64)
sp.synchronize()
s1 = sp.stats()
out = (C.c_uint64 * 2048)()
assert sp.engine._lib.c4_debug_stamps(sp.engine._h, out) == 0
a = np.array(list(out), dtype=np.uint64).reshape(128, 16)
tree = a[:, 0:8].astype(np.float64) / 64
netc = (a[:, 8:16] & np.uint64((1 << 48) - 1)).astype(np.float64) / 64
npass = (a[:, 8:16] >> np.uint64(48)).astype(np.float64) / 64
print("per wave and step (cycles): tree %.0f  net %.0f  (%.2f passes/step -> %.0f cycles per pass)" %
      (tree.mean(), netc.mean(), npass.mean(), netc.sum() / max(1.0, npass.sum())))
tot = tree + netc
print("per-wave total cycles/step: mean %.0f  p50 %.0f  p95 %.0f  max %.0f;  per-wave passes/step: min %.2f max %.2f" % (tot.mean(), np.median(tot), np.percentile(tot, 95), tot.max(), npass.min(), npass.max()))
print("per-wave tree cycles/step: p50 %.0f p95 %.0f max %.0f ; net: p50 %.0f p95 %.0f max %.0f" % (np.median(tree), np.percentile(tree, 95), tree.max(), np.median(netc), np.percentile(netc, 95), netc.max()))
import time
sp.synchronize(); t0 = time.perf_counter(); sp.run_steps(6400); sp.synchronize(); dt = time.perf_counter() - t0
s2 = sp.stats()
print("wall: %.1f us/step, %.1f M sims/s" % (dt / 6400 * 1e6, (s2["simulations"] - s1["simulations"]) / dt / 1e6))
print("simulations per slot and step: %.2f" % ((s1["simulations"] - s0["simulations"]) / 64.0 / slots))
# phases of the last network pass of each wave of workgroups 0..15, measured inside the fused kernel: needs a
# diagnostic build of the library (hipcc ... -DC4_FUSED_NET_STAMPS=1 -o x.so; C4_ENGINE_LIB=x.so): the production
# kernel carries no stamp pointer (it cost 3 %)
out2 = (C.c_uint64 * 2048)()
assert sp.engine._lib.c4_debug_fused_net_stamps(sp.engine._h, out2) == 0
st = np.array(list(out2), dtype=np.int64).reshape(128, 16)
st = st[st[:, 0] > 0]
if len(st) == 0:
    print("(no in-fused network stamps: production build)")
    sys.exit(0)
names = ["stem"] + ["L%d" % i for i in range(6)] + ["tower_end", "heads", "mlp"]
d = st[:, 1:11] - st[:, 0:10]
print("in-fused network pass (%d waves sampled), cycles per phase (median): " % len(st) +
      " ".join("%s=%d" % (n, x) for n, x in zip(names, np.median(d, axis=0))) + "  total=%d" % np.median(st[:, 10] - st[:, 0]))
print("   layer 2 split (median): k-loop=%d skip=%d epilogue=%d" %
      (np.median(st[:, 12] - st[:, 3]), np.median(st[:, 13] - st[:, 12]), np.median(st[:, 14] - st[:, 13])))
