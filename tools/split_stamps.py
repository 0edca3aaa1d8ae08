#!/usr/bin/env python3
"""Diagnostic: busy cycles of the tree waves and network waves of c4_selfplay_split_kernel (C4_TREE_STAMPS=1)."""
import ctypes as C
import os
import sys

import numpy as np

os.environ["C4_TREE_STAMPS"] = "1"
os.environ["C4_FUSED_MODE"] = "split"
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from connect4_amd.config import MCTSConfig  # noqa: E402
from connect4_amd.fused_net import FusedNet  # noqa: E402
from connect4_amd.net import random_init_state_dict  # noqa: E402
from connect4_amd.selfplay import SelfPlay  # noqa: E402

slots = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
prec = os.environ.get("C4_NET_PRECISION", "f32x3")   # f32x3 (the default net) | f16
tw = int(os.environ.get("C4_SPLIT_TW", "4"))
net = FusedNet(random_init_state_dict(seed=0), precision=prec)
sp = SelfPlay(net, slots, MCTSConfig.self_play(800), seed=0, use_graph=False, fused_loop=True, steps_per_launch=64, max_inner_iters=32)
sp.run_steps(6400)
sp.synchronize()
s0 = sp.stats()
sp.run_steps(64)
sp.synchronize()
s1 = sp.stats()
out = (C.c_uint64 * 2048)()
assert sp.engine._lib.c4_debug_stamps(sp.engine._h, out) == 0
a = np.array(list(out), dtype=np.uint64).reshape(128, 16)
nw = 8 - tw
mask = np.uint64((1 << 48) - 1)
tree = (a[:, 0:tw] & mask).astype(np.float64) / 64
netb = (a[:, 8:8 + nw] & mask).astype(np.float64) / 64
npass = (a[:, 8:8 + nw] >> np.uint64(48)).astype(np.float64) / 64
print("slots %d  tree waves %d: busy cycles per 80k-cycle step: mean %.0f p5 %.0f p95 %.0f" % (slots, tw, tree.mean(), np.percentile(tree, 5), np.percentile(tree, 95)))
print("network waves %d: busy %.0f cycles per step (p95 %.0f), %.2f passes per step -> %.0f cycles per pass" %
      (nw, netb.mean(), np.percentile(netb, 95), npass.mean(), netb.sum() / max(1.0, npass.sum())))
if tw == 4:
    fl = a[:, 4:8]
    print("placement: two waves per SIMD in %d of %d workgroups; SIMD ids of tree waves 0..3 of workgroup 0: %s" %
          (int(((fl >> np.uint64(8)) & np.uint64(1)).all(axis=1).sum()), len(fl), (fl[0] & np.uint64(3)).tolist()))
print("simulations per slot and step: %.2f" % ((s1["simulations"] - s0["simulations"]) / 64.0 / slots))
# phases of the last network pass of each network wave of workgroups 0..15, measured inside the split kernel: needs a
# diagnostic build of the library (hipcc ... -DC4_FUSED_NET_STAMPS=1 -o x.so; C4_ENGINE_LIB=x.so)
out2 = (C.c_uint64 * 2048)()
assert sp.engine._lib.c4_debug_fused_net_stamps(sp.engine._h, out2) == 0
st = np.array(list(out2), dtype=np.int64).reshape(128, 16)
st = st[st[:, 0] > 0]
if len(st):
    names = ["stem"] + ["L%d" % i for i in range(6)] + ["tower_end", "heads", "mlp"]
    dd = st[:, 1:11] - st[:, 0:10]
    print("network pass inside the split kernel (%d waves sampled), cycles per phase (median): " % len(st) +
          " ".join("%s=%d" % (n, x) for n, x in zip(names, np.median(dd, axis=0))) + "  total=%d" % np.median(st[:, 10] - st[:, 0]))
    sl = int(os.environ.get("C4_NET_STAMP_LAYER", "2"))   # the layer the diagnostic build splits (-DC4_NET_STAMP_LAYER=n)
    print("   layer %d split (median): k-loop=%d skip=%d epilogue=%d" %
          (sl, np.median(st[:, 12] - st[:, 1 + sl]), np.median(st[:, 13] - st[:, 12]), np.median(st[:, 14] - st[:, 13])))
