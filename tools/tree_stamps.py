#!/usr/bin/env python3
"""Diagnostic: where does a c4_step launch spend its cycles?  (C4_TREE_STAMPS=1)"""
import ctypes as C
import os
import sys

import numpy as np
import torch

os.environ["C4_TREE_STAMPS"] = "1"
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from connect4_amd.config import MCTSConfig  # noqa: E402
from connect4_amd.fused_net import FusedNet  # noqa: E402
from connect4_amd.net import random_init_state_dict  # noqa: E402
from connect4_amd.selfplay import SelfPlay  # noqa: E402

slots = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
net = FusedNet(random_init_state_dict(seed=0), precision=os.environ.get("C4_NET_PRECISION", "f32x3"))
sp = SelfPlay(net, slots, MCTSConfig.self_play(800), seed=0, use_graph=False)
sp.run_steps(int(sys.argv[2]) if len(sys.argv) > 2 else 3000)
sp.synchronize()
acc = []
for _ in range(20):
    sp.run_steps(1)
    sp.synchronize()
    out = (C.c_uint64 * 2048)()
    rc = sp.engine._lib.c4_debug_stamps(sp.engine._h, out)
    assert rc == 0
    acc.append(np.array(list(out), dtype=np.int64).reshape(256, 8))
a = np.stack(acc)  # [20][256][8]
names = ["state_load", "apply", "descent", "tail(before emit)", "emit+persist"]
d = a[:, :, 1:6] - a[:, :, 0:5]
print("mean cycles per phase over %d block-launches:" % (a.shape[0] * a.shape[1]))
for i, n in enumerate(names):
    print("  %-18s mean %7.0f  p50 %7.0f  p95 %7.0f  max %7.0f" % (n, d[:, :, i].mean(), np.median(d[:, :, i]), np.percentile(d[:, :, i], 95), d[:, :, i].max()))
tot = a[:, :, 5] - a[:, :, 0]
print("  total per block    mean %7.0f  p95 %7.0f  max %7.0f   depth(lane0 slot) mean %.2f max %d" % (tot.mean(), np.percentile(tot, 95), tot.max(), (a[:, :, 6] & 0xffffffff).mean(), (a[:, :, 6] & 0xffffffff).max()))
lv = (a[:, :, 6] >> 32).astype(np.float64)
wait = (a[:, :, 7] >> 32).astype(np.float64)
alu = (a[:, :, 7] & 0xffffffff).astype(np.float64)
print("  wave-level loop iterations mean %.2f; per level: load-wait %.0f cycles, rest (ALU/DPP/LDS) %.0f cycles" % (lv.mean(), wait.sum() / lv.sum(), alu.sum() / lv.sum()))
span = a[:, :, 5].max(axis=1) - a[:, :, 0].min(axis=1)
print("  launch span over the first 256 blocks: mean %.0f cycles" % span.mean())
