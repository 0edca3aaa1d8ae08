"""Train step with the reference's recipe (oinkoink/neural/pytorch/model.py:138-169,
200-250): SGD(momentum 0.9, weight decay 1e-4) + MultiStepLR, loss = MSE(value) + BCE(policy), 5 epochs x
batch 4096 over a shuffled dataset, checkpoint dict with the three state dicts.  Stock PyTorch-ROCm; not
on the hot path.

Parity with ``ModelWrapper.train`` is pinned (tests/test_training_cpu.py against a fixture generated from
the reference by tests/golden/gen_golden.py): batches are drawn exactly as the reference's
``DataLoader(data, batch_size, shuffle=True)`` draws them -- same consumption of torch's global RNG,
same permutation, same batch boundaries -- but gathered with one tensor index per batch instead of
4096 Python ``__getitem__`` calls + collate.
"""
import os
from typing import Optional

import torch
import torch.nn as nn
from torch.optim.lr_scheduler import MultiStepLR

from .net import NetConfig, PolicyValueNet


class ModelConfig:
    """oinkoink/neural/config.py:19-39"""

    def __init__(self, net_config=None, weight_decay=1e-4, momentum=0.9, initial_lr=0.01,
                 milestones=(100, 300, 600), gamma=0.1, batch_size=4096, n_training_epochs=5, use_gpu=True):
        self.net_config = net_config or NetConfig()
        self.weight_decay = weight_decay
        self.momentum = momentum
        self.initial_lr = initial_lr
        self.milestones = list(milestones)
        self.gamma = gamma
        self.batch_size = batch_size
        self.n_training_epochs = n_training_epochs
        self.use_gpu = use_gpu


def dataloader_permutation(n: int, generator: Optional[torch.Generator] = None) -> torch.Tensor:
    """The index order one pass of ``DataLoader(dataset_of_n, shuffle=True)`` visits, consuming torch's global
    RNG exactly as that pass does (torch.utils.data: _BaseDataLoaderIter.__init__ draws a base seed, then
    RandomSampler.__iter__ draws the seed of a private generator for ``randperm``)."""
    if generator is not None:          # an explicit generator (tests): one permutation from it
        return torch.randperm(n, generator=generator)
    torch.empty((), dtype=torch.int64).random_()                       # DataLoader iterator: _base_seed
    seed = int(torch.empty((), dtype=torch.int64).random_().item())    # RandomSampler.__iter__
    g = torch.Generator()
    g.manual_seed(seed)
    return torch.randperm(n, generator=g)


class Trainer:
    """ModelWrapper's training half (model.py:138-169, 200-250).  `device`: where the net trains; default =
    the reference's rule (cuda:0 when use_gpu and a GPU is there, model.py:143-147) -- a multi-rank caller
    passes its own GPU."""

    def __init__(self, config: ModelConfig = None, file_name: str = None, device=None, pad_ragged_batches=None,
                 fused_bn=None, use_graph=None):
        self.config = config or ModelConfig()
        self.net = PolicyValueNet(self.config.net_config)
        if device is not None:
            self.device = torch.device(device)
        else:
            self.device = torch.device("cuda:0" if self.config.use_gpu and torch.cuda.is_available() else "cpu")
        self.net.to(self.device)
        # the ragged last batch of an epoch is padded to the full batch size (see net._BatchNorm2d); on by default on a GPU
        self.pad_ragged_batches = (self.device.type == "cuda") if pad_ragged_batches is None else bool(pad_ragged_batches)
        # On a GPU: batch normalisation (+ residual add + LeakyReLU) of the train step runs on the library's HIP kernels
        # (bn_train.py), and the full batches of a train() call are ONE captured HIP graph replayed per batch -- a stock
        # step is ~150 launches of ~25 us kernels, i.e. as much host dispatch as GPU work.  Both default on for cuda.
        self.fused_bn = (self.device.type == "cuda") if fused_bn is None else bool(fused_bn)
        self.use_graph = ((self.device.type == "cuda") if use_graph is None else bool(use_graph)) and self.fused_bn
        # (no graph with stock batch normalisation: under capture MIOpen's batch-norm backward leaves a small non-zero mean in
        #  dx, which is the whole gradient of the head convolutions' biases -- mathematically zero -- and they drift: -6.6
        #  after 138 steps where the eager step leaves 0.02; measured, build/train_probe3.py of round 3)
        from .net import _BatchNorm2d
        for m in self.net.modules():
            if isinstance(m, _BatchNorm2d):
                m.fused = self.fused_bn
        self.optimiser = torch.optim.SGD(self.net.parameters(), lr=self.config.initial_lr,
                                         momentum=self.config.momentum, weight_decay=self.config.weight_decay)
        self.scheduler = MultiStepLR(self.optimiser, milestones=self.config.milestones, gamma=self.config.gamma)
        if file_name is not None:   # model.py:157-161
            ckpt = torch.load(file_name, map_location=self.device, weights_only=True)
            self.net.load_state_dict(ckpt["net_state_dict"])
            self.optimiser.load_state_dict(ckpt["optimiser_state_dict"])
            self.scheduler.load_state_dict(ckpt["scheduler_state_dict"])
        self.value_loss = nn.MSELoss()
        self.prior_loss = nn.BCELoss()
        self.net.eval()

    def train(self, boards, values, priors, generator=None):
        """One generation (model.py:200-240): n_training_epochs shuffled passes over (boards F32[N,3,6,7],
        values F32[N], priors F32[N,7]) -- tensors on any device; they are moved to the trainer's device once.
        Returns the last batch's loss."""
        if self.device.type == "cuda":
            with torch.cuda.device(self.device):    # graph capture / replay and the library's launches go to the current device
                return self._train(boards, values, priors, generator)
        return self._train(boards, values, priors, generator)

    def _train(self, boards, values, priors, generator):
        n = int(boards.shape[0])
        bs = self.config.batch_size
        boards, values, priors = boards.to(self.device), values.to(self.device), priors.to(self.device)
        self.net.train()
        last = None
        graph = None            # the captured step of THIS call (the data tensors and the learning rate are baked into it)
        eager_full = 0
        want_graph = self.use_graph and self.device.type == "cuda" and (n // bs) * self.config.n_training_epochs >= 4
        for _ in range(self.config.n_training_epochs):
            perm = dataloader_permutation(n, generator).to(self.device)
            for i in range(0, n, bs):
                idx = perm[i:i + bs]
                k = int(idx.shape[0])
                if k == bs and want_graph:
                    if graph is None and eager_full >= 2:      # (the first steps run eagerly: momentum buffers, MIOpen's choices)
                        graph, sidx, sloss = self._capture_step(boards, values, priors)
                    if graph is not None:
                        sidx.copy_(idx)
                        graph.replay()
                        last = sloss
                        continue
                    eager_full += 1
                pad = self.pad_ragged_batches and k < bs and n > bs and k > 1
                if pad:   # DataLoader's drop_last=False batch (model.py:208-212), at the full batch's shape
                    idx = torch.cat([idx, idx[:1].expand(bs - k)])
                    self._set_valid_rows(k)
                last = self._eager_step(boards[idx], values[idx], priors[idx], k if pad else None)
                if pad:
                    self._set_valid_rows(None)
        last = None if last is None else float(last)
        del graph
        self.optimiser.zero_grad()            # (gradients that live in a graph's pool are not kept)
        self.scheduler.step()       # once per generation (model.py:239)
        self.net.eval()
        return last

    def _eager_step(self, b, v, p, k=None):
        """model.py:214-230 for one batch (its first k rows when the batch is a padded ragged one).  Returns the detached
        loss (read back once, after the last batch: no host synchronisation per step); nothing of the step's autograd graph
        outlives the call -- a graph captured later must not find AccumulateGrad nodes bound to this stream."""
        self.optimiser.zero_grad()
        xv, xp = self.net(b)
        if k is not None:
            xv, xp, v, p = xv[:k], xp[:k], v[:k], p[:k]
        loss = self.value_loss(xv, v) + self.prior_loss(xp, p)   # model.py:221-225
        loss.backward()
        self.optimiser.step()
        return loss.detach()

    def _capture_step(self, boards, values, priors):
        """One full-batch train step -- gather by a static index tensor, forward, loss, backward, SGD -- captured as a HIP
        graph (torch.cuda.graph): replayed per batch with the batch's indices copied into `sidx`.  Same kernels, same order,
        same arithmetic as the eager step."""
        sidx = torch.zeros(self.config.batch_size, dtype=torch.int64, device=self.device)
        self.optimiser.zero_grad(set_to_none=True)
        graph = torch.cuda.CUDAGraph()
        with torch.cuda.graph(graph):
            b, v, p = boards[sidx], values[sidx], priors[sidx]
            xv, xp = self.net(b)
            loss = self.value_loss(xv, v) + self.prior_loss(xp, p)
            loss.backward()
            self.optimiser.step()
            sloss = loss.detach()
        return graph, sidx, sloss

    def state(self):
        """The checkpoint dict of save() with CPU tensors (model.py:242-250)."""
        def cpu(o):
            if isinstance(o, torch.Tensor):
                return o.detach().cpu()
            if isinstance(o, dict):
                return {k: cpu(v) for k, v in o.items()}
            if isinstance(o, (list, tuple)):
                return type(o)(cpu(v) for v in o)
            return o
        return {"net_state_dict": cpu(self.net.state_dict()), "optimiser_state_dict": cpu(self.optimiser.state_dict()),
                "scheduler_state_dict": cpu(self.scheduler.state_dict())}

    def load_state(self, ckpt):
        self.net.load_state_dict(ckpt["net_state_dict"])
        self.optimiser.load_state_dict(ckpt["optimiser_state_dict"])     # (moves the momentum buffers to the parameters' device)
        self.scheduler.load_state_dict(ckpt["scheduler_state_dict"])

    def broadcast_state(self, src: int = 0, extra=None):
        """Every rank's trainer becomes rank `src`'s: net (weights, batch-norm statistics), optimiser (momentum buffers,
        learning rate) and scheduler state travel as one object (~0.5 MB for the default net; torch.distributed, RCCL or
        gloo).  The reference has ONE model per generation (training.py:147-153); ranks that trained by themselves would
        drift apart (non-deterministic GPU reductions, unshared shuffles).  `extra` (e.g. the loss) rides along; returns
        rank src's."""
        import torch.distributed as dist
        box = [(self.state(), extra) if dist.get_rank() == src else None]
        on_gpu = dist.get_backend() == "nccl" and self.device.type == "cuda"     # RCCL moves the pickled bytes through this rank's GPU
        dist.broadcast_object_list(box, src=src, device=self.device if on_gpu else None)
        if dist.get_rank() != src:
            self.load_state(box[0][0])
        return box[0][1]

    def _set_valid_rows(self, k):
        from .net import _BatchNorm2d
        for m in self.net.modules():
            if isinstance(m, _BatchNorm2d):
                m.valid_rows = k

    def save(self, folder_path):    # model.py:242-250
        os.makedirs(folder_path, exist_ok=True)
        torch.save({"net_state_dict": self.net.state_dict(),
                    "optimiser_state_dict": self.optimiser.state_dict(),
                    "scheduler_state_dict": self.scheduler.state_dict()}, os.path.join(folder_path, "net.pth"))
