#!/usr/bin/env python3
"""Bit-for-bit comparison of the fused net's answers between builds of the engine library:
    python tools/compare_net_builds.py [--precision f32x3] a.so b.so [c.so ...]
Each build evaluates the same seeded positions (standalone forward, both entry points) in a process of its own."""
import os
import subprocess
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def child(out, precision):
    from connect4_amd.fused_net import FusedNet
    from connect4_amd.net import NetConfig, random_init_state_dict
    from oracle import c4oracle as oc
    rng = np.random.RandomState(1)
    c0, c1 = [], []
    while len(c0) < 3000:
        b = oc.Board.empty()
        for _ in range(int(rng.randint(0, 42))):
            m = b.valid_mask()
            if not m:
                break
            b.make_move(int(rng.choice([c for c in range(7) if (m >> c) & 1])))
        c0.append(b.key()[0])
        c1.append(b.key()[1])
    c0, c1 = np.array(c0, dtype=np.uint64), np.array(c1, dtype=np.uint64)
    res = []
    for n_res in (3, 1):
        net = FusedNet(random_init_state_dict(NetConfig(n_residuals=n_res), seed=n_res), precision=precision)
        v, p = net.evaluate_bits(c0, c1)
        wv, wp = net.evaluate_bits(c0, c1, wave=True)
        assert np.array_equal(v, wv) and np.array_equal(p, wp)
        res += [v, p]
        net.close()
    np.savez(out, *res)


if __name__ == "__main__":
    if sys.argv[1] == "--child":
        child(sys.argv[2], sys.argv[3])
        sys.exit(0)
    args = sys.argv[1:]
    precision = "f32x3"
    if args[0] == "--precision":
        precision, args = args[1], args[2:]
    outs = []
    for i, lib in enumerate(args):
        out = "/tmp/cmp_net_%d.npz" % i
        env = dict(os.environ, C4_ENGINE_LIB=os.path.abspath(lib))
        subprocess.check_call([sys.executable, os.path.abspath(__file__), "--child", out, precision], env=env)
        outs.append(np.load(out))
    ok = True
    for i in range(1, len(outs)):
        same = all(np.array_equal(outs[0][k], outs[i][k]) for k in outs[0].files)
        worst = max(float(np.abs(outs[0][k] - outs[i][k]).max()) for k in outs[0].files)
        print("%s vs %s (%s): %s (max |diff| %.3g)" % (args[0], args[i], precision, "bit-identical" if same else "DIFFERENT", worst))
        ok = ok and same
    sys.exit(0 if ok else 1)
