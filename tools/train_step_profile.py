#!/usr/bin/env python3
"""Kernel table (torch profiler) of Trainer.train's default GPU step: python tools/train_step_profile.py (profiles/r03_train_step.txt)."""
import sys, time, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from torch.profiler import profile, ProfilerActivity
from connect4_amd.training import ModelConfig, Trainer
g = torch.Generator().manual_seed(1)
n = 4096 * 24
b = (torch.rand(n, 3, 6, 7, generator=g) > 0.7).float().cuda()
v = torch.rand(n, generator=g).cuda()
p = torch.softmax(torch.rand(n, 7, generator=g), 1).cuda()
torch.manual_seed(0)
tr = Trainer(ModelConfig(n_training_epochs=1), device="cuda:0")
tr.train(b, v, p)
torch.cuda.synchronize()
with profile(activities=[ProfilerActivity.CUDA]) as prof:
    tr.train(b, v, p)
    torch.cuda.synchronize()
print(prof.key_averages().table(sort_by="cuda_time_total", row_limit=40, max_name_column_width=90))
