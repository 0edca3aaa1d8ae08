#!/usr/bin/env python3
"""The train step through Trainer.train on one GPU: stock / library batch-norm kernels x eager / captured HIP graph, ms per step and
the final states against the stock eager run (profiles/r03_train_step.txt).  python tools/train_step_bench.py"""
import sys, time, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from connect4_amd.training import ModelConfig, Trainer
g = torch.Generator().manual_seed(1)
n = 4096 * 64 + 1234
b = (torch.rand(n, 3, 6, 7, generator=g) > 0.7).float().cuda()
v = torch.rand(n, generator=g).cuda()
p = torch.softmax(torch.rand(n, 7, generator=g), 1).cuda()
res = {}
for tag, kw in (("stock eager", dict(fused_bn=False, use_graph=False)),
                ("fused eager", dict(fused_bn=True, use_graph=False)), ("fused graph", dict(fused_bn=True, use_graph=True))):
    torch.manual_seed(0)
    tr = Trainer(ModelConfig(n_training_epochs=2), device="cuda:0", **kw)
    tr.train(b[:4096 * 4], v[:4096 * 4], p[:4096 * 4])     # warm (MIOpen finds, allocator)
    torch.manual_seed(5)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    loss = tr.train(b, v, p)
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    steps = 2 * 65
    res[tag] = {k: x.detach().clone() for k, x in tr.net.state_dict().items()}
    print("%s: %.3f s for %d steps = %.3f ms/step, loss %.6f" % (tag, dt, steps, dt / steps * 1e3, loss), flush=True)
base = res["stock eager"]
for tag in res:
    print("   %s vs stock eager: max |diff| %.3g" % (tag, max(float((base[k].double() - res[tag][k].double()).abs().max()) for k in base)))
