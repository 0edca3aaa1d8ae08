#!/usr/bin/env python3
"""Micro-benchmark of the fused network kernel: n positions, HIP-event timed."""
import argparse
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--n", type=int, default=4096)
    ap.add_argument("--iters", type=int, default=200)
    ap.add_argument("--wave", type=int, default=0, help="1: the wave-private forward (one position per wave)")
    ap.add_argument("--precision", default="f16", choices=["f16", "f32x3"])
    args = ap.parse_args()
    from connect4_amd.fused_net import FusedNet
    from connect4_amd.net import InferenceNet, random_init_state_dict
    from connect4_amd.engine import board_planes
    sd = random_init_state_dict(seed=0)
    net = FusedNet(sd, precision=args.precision)
    rng = np.random.RandomState(0)
    c0 = rng.randint(0, 2 ** 40, size=args.n).astype(np.uint64) & np.uint64(0x7EFDFBF7EFDF)
    c1 = (rng.randint(0, 2 ** 40, size=args.n).astype(np.uint64) & np.uint64(0x7EFDFBF7EFDF)) & ~c0
    d0 = torch.from_numpy(c0.view(np.int64)).cuda()
    d1 = torch.from_numpy(c1.view(np.int64)).cuda()
    v = torch.zeros(args.n, device="cuda")
    p = torch.zeros(args.n, 7, device="cuda")
    wave = bool(args.wave)
    for _ in range(10):
        net.forward_bitboards(d0.data_ptr(), d1.data_ptr(), args.n, v, p, wave=wave)
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(args.iters):
        net.forward_bitboards(d0.data_ptr(), d1.data_ptr(), args.n, v, p, wave=wave)
    b.record()
    torch.cuda.synchronize()
    us = a.elapsed_time(b) * 1000 / args.iters
    tf = 4.74e6 * args.n / (us * 1e-6) / 1e12
    print("fused net (%s, wave=%d, active waves per CU %s): n=%d  %.1f us/forward  %.1f TFLOP/s (%.1f%% of 2.5 PF fp16 dense)" % (args.precision, int(wave), os.environ.get("C4_NET_WAVE_ACTIVE", "8"), args.n, us, tf, tf / 25.0))
    ref = InferenceNet(sd, device="cuda", dtype=torch.float32)
    planes = torch.from_numpy(board_planes(c0, c1)).cuda()
    rv, rp = ref(planes)
    print("max |dv| %.3g max |dp| %.3g" % ((v - rv).abs().max().item(), (p - rp).abs().max().item()))
    if os.environ.get("C4_NET_STAMPS"):
        import ctypes as C
        out = (C.c_uint64 * 128)()
        rc = net._lib.c4_net_debug_stamps(net._h, out)
        st = np.array(list(out), dtype=np.int64).reshape(8, 16)
        if wave or args.precision != "f16":
            names = ["start", "stem"] + ["L%d" % i for i in range(6)] + ["tower_end", "heads", "mlp"]
            for w in range(8):
                d = st[w, 1:11] - st[w, 0:10]
                print("wave %d: " % w + " ".join("%s=%d" % (n, x) for n, x in zip(names[1:], d)) + "  total=%d" % (st[w, 10] - st[w, 0]) +
                      "  | L2: k-loop=%d skip=%d epilogue=%d" % (st[w, 12] - st[w, 3], st[w, 13] - st[w, 12], st[w, 14] - st[w, 13]))
            return
        names = ["start", "stem", "bar0"] + ["L%d" % i for i in range(6)] + ["tower_end", "heads", "fc_end"]
        for w in range(8):
            d = st[w, 1:12] - st[w, 0:11]
            print("wave %d: " % w + " ".join("%s=%d" % (n, x) for n, x in zip(names[1:], d)) + "  total=%d" % (st[w, 11] - st[w, 0]) +
                  "  | tile(L2): addr=%d chain=%d epi=%d" % (st[w, 13] - st[w, 12], st[w, 14] - st[w, 13], st[w, 15] - st[w, 14]))


if __name__ == "__main__":
    main()
