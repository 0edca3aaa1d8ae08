"""Policy/value network for the leaf batch (the only dense contraction on the hot path).

Two classes:

* ``PolicyValueNet`` -- the trainable residual CNN with exactly the architecture and state_dict
  key names of the reference's ``Net`` (oinkoink/neural/pytorch/model.py:20-134), so checkpoints
  written by the existing training loop (model.py:242-250, key ``net_state_dict``) load here and
  vice versa.
* ``InferenceNet`` -- eval-only execution plan built from a state_dict for the self-play engine:
  BatchNorm folded into the convolutions, the value head's stack of activation-free
  ``Linear(42,42)`` layers (model.py:69-70,83) collapsed into one affine map, channels-last,
  optional fp16/bf16 storage.  Consumes the engine's ``[n,3,6,7]`` plane batch
  (board.py:147-154) and produces ``values[n]`` in [0,1] and ``priors[n,7]`` exactly like
  ``ModelWrapper._call_list`` (model.py:269-282), but without leaving the device.
"""
from typing import Dict, Optional

import torch
import torch.nn as nn
import torch.nn.functional as F

AREA, WIDTH = 42, 7
LEAK = 0.01  # nn.LeakyReLU() default negative slope (model.py:31,43,67,102)


class NetConfig:
    """oinkoink/neural/config.py:7-16"""

    def __init__(self, channels=3, filters=32, n_fc_layers=4, n_residuals=3):
        self.channels = channels
        self.filters = filters
        self.n_fc_layers = n_fc_layers
        self.n_residuals = n_residuals


class _BatchNorm2d(nn.BatchNorm2d):
    """nn.BatchNorm2d (same parameters and state-dict keys) with two additions for the train step on a GPU:

    * `forward(x, residual=None, slope=None)` = act(bn(x) + residual): in training mode on a CUDA float32 tensor this is ONE
      pair of HIP kernels of the library (bn_train.fused_bn_act; stock MIOpen spends half of a train step in batch
      normalisation at this net's shape), otherwise stock PyTorch operators in the reference's order (model.py:20-55);
    * it can take its batch statistics from the first `valid_rows` samples only.  The trainer pads the ragged last batch of
      an epoch to the full batch size -- every convolution then runs the shape MIOpen has already compiled kernels for
      instead of a new one per generation (6-9 s each) -- and sets valid_rows: the padding rows are normalised with the
      real rows' statistics, carry no loss, and so change neither the activations of the real rows nor any gradient."""
    valid_rows = None
    fused = True          # (False: stock operators on the GPU as well -- A/B runs and tests)

    def forward(self, x, residual=None, slope=None):
        if self.training and self.fused and x.is_cuda and x.dtype == torch.float32:
            from .bn_train import fused_bn_act
            return fused_bn_act(x, self, residual, 1.0 if slope is None else slope, self.valid_rows)
        y = self._stock(x)
        if residual is not None:
            y = y + residual
        return y if slope is None else F.leaky_relu(y, slope)

    def _stock(self, x):
        k = self.valid_rows
        if k is None or not self.training:
            return super().forward(x)
        xs = x[:k]
        mean = xs.mean(dim=(0, 2, 3))
        var = xs.var(dim=(0, 2, 3), unbiased=False)
        with torch.no_grad():      # running statistics as nn.BatchNorm2d keeps them (momentum, unbiased variance)
            cnt = xs.numel() // xs.shape[1]
            self.running_mean.mul_(1.0 - self.momentum).add_(mean, alpha=self.momentum)
            self.running_var.mul_(1.0 - self.momentum).add_(var * (cnt / (cnt - 1.0)), alpha=self.momentum)
            self.num_batches_tracked += 1
        y = (x - mean.view(1, -1, 1, 1)) * torch.rsqrt(var.view(1, -1, 1, 1) + self.eps)
        return y * self.weight.view(1, -1, 1, 1) + self.bias.view(1, -1, 1, 1)


class _ConvBNAct(nn.Sequential):
    """model.py:20-31: conv3x3 (no bias) + BN + LeakyReLU; indices 0/1/2 give the reference's key names."""

    def forward(self, x):
        return self[1](self[0](x), slope=self[2].negative_slope)


def _conv_bn_act(cin, cout):
    return _ConvBNAct(nn.Conv2d(cin, cout, 3, padding=1, bias=False), _BatchNorm2d(cout), nn.LeakyReLU(LEAK))


class _Residual(nn.Module):
    """model.py:36-55"""

    def __init__(self, f):
        super().__init__()
        self.conv1 = nn.Conv2d(f, f, 3, padding=1, bias=False)
        self.conv2 = nn.Conv2d(f, f, 3, padding=1, bias=False)
        self.batch_norm1 = _BatchNorm2d(f)
        self.batch_norm2 = _BatchNorm2d(f)

    def forward(self, x):
        if self.training and self.batch_norm1.fused and x.is_cuda:     # the GPU train step: the library's weight-gradient kernel behind the convolutions
            from .bn_train import conv3x3
            y = self.batch_norm1(conv3x3(self.conv1, x), slope=LEAK)
            return self.batch_norm2(conv3x3(self.conv2, y), residual=x, slope=LEAK)
        y = self.batch_norm1(self.conv1(x), slope=LEAK)
        return self.batch_norm2(self.conv2(y), residual=x, slope=LEAK)


class _ValueHead(nn.Module):
    """model.py:60-91"""

    def __init__(self, f, n_fc):
        super().__init__()
        self.conv1 = nn.Conv2d(f, 1, 1)
        self.batch_norm = _BatchNorm2d(1)
        self.fcN = nn.Sequential(*[nn.Linear(AREA, AREA) for _ in range(n_fc)])
        self.fc1 = nn.Linear(AREA, 1)
        self.w1 = nn.Parameter(torch.tensor(1.0), requires_grad=False)
        self.w2 = nn.Parameter(torch.tensor(0.5), requires_grad=False)

    def forward(self, x):
        x = self.batch_norm(self.conv1(x), slope=LEAK).flatten(1)
        x = F.leaky_relu(self.fcN(x), LEAK)
        x = torch.tanh(self.fc1(x))
        return ((x + self.w1) * self.w2).view(-1)


class _PolicyHead(nn.Module):
    """model.py:96-117"""

    def __init__(self, f):
        super().__init__()
        self.conv1 = nn.Conv2d(f, 2, 1)
        self.batch_norm = _BatchNorm2d(2)
        self.fc1 = nn.Linear(2 * AREA, WIDTH)

    def forward(self, x):
        x = self.batch_norm(self.conv1(x), slope=LEAK).flatten(1)
        return torch.softmax(self.fc1(x), dim=1)


class PolicyValueNet(nn.Module):
    """State-dict compatible with the reference's Net (model.py:120-134)."""

    def __init__(self, config: Optional[NetConfig] = None):
        super().__init__()
        config = config or NetConfig()
        self.config = config
        self.body = nn.Sequential(
            _conv_bn_act(config.channels, config.filters),
            nn.Sequential(*[_Residual(config.filters) for _ in range(config.n_residuals)]))
        self.value_head = _ValueHead(config.filters, config.n_fc_layers)
        self.policy_head = _PolicyHead(config.filters)

    def forward(self, x):
        x = self.body(x)
        return self.value_head(x), self.policy_head(x)

    @staticmethod
    def config_from_state_dict(sd: Dict[str, torch.Tensor]) -> NetConfig:
        filters, channels = sd["body.0.0.weight"].shape[:2]
        n_res = len({k.split(".")[2] for k in sd if k.startswith("body.1.")})
        n_fc = len({k.split(".")[2] for k in sd if k.startswith("value_head.fcN.")})
        return NetConfig(channels, filters, n_fc, n_res)


def _fold_bn(w, b, bn_w, bn_b, mean, var, eps=1e-5):
    # eval-mode BatchNorm2d: y = (x - mean) / sqrt(var + eps) * gamma + beta
    scale = bn_w.double() / torch.sqrt(var.double() + eps)
    w2 = w.double() * scale.view(-1, 1, 1, 1)
    b0 = torch.zeros_like(mean, dtype=torch.float64) if b is None else b.double()
    b2 = (b0 - mean.double()) * scale + bn_b.double()
    return w2, b2


class InferenceNet:
    """Eval-only plan of PolicyValueNet for the engine's leaf batch.  dtype float32 by default
    (the reference computes in fp32); float16/bfloat16 store activations and weights in half
    precision with fp32 accumulation in MIOpen/hipBLASLt (MFMA) and fp32 heads."""

    def __init__(self, state_dict: Dict[str, torch.Tensor], device="cuda", dtype=torch.float32,
                 channels_last=True):
        sd = {k: v.detach().to("cpu") for k, v in state_dict.items()}
        self.config = PolicyValueNet.config_from_state_dict(sd)
        self.device = torch.device(device)
        self.dtype = dtype
        self.mf = torch.channels_last if channels_last else torch.contiguous_format
        cfg = self.config

        def put_conv(w, b):
            return (w.to(self.device, dtype).contiguous(memory_format=self.mf), b.to(self.device, dtype))

        def bn(prefix):
            return (sd[prefix + ".weight"], sd[prefix + ".bias"], sd[prefix + ".running_mean"],
                    sd[prefix + ".running_var"])

        self.stem = put_conv(*_fold_bn(sd["body.0.0.weight"], None, *bn("body.0.1")))
        self.res = []
        for i in range(cfg.n_residuals):
            p = "body.1.%d." % i
            a = put_conv(*_fold_bn(sd[p + "conv1.weight"], None, *bn(p + "batch_norm1")))
            b = put_conv(*_fold_bn(sd[p + "conv2.weight"], None, *bn(p + "batch_norm2")))
            self.res.append((a, b))
        # both 1x1 head convolutions in one conv: channel 0 = value, 1..2 = policy
        vw, vb = _fold_bn(sd["value_head.conv1.weight"], sd["value_head.conv1.bias"], *bn("value_head.batch_norm"))
        pw, pb = _fold_bn(sd["policy_head.conv1.weight"], sd["policy_head.conv1.bias"], *bn("policy_head.batch_norm"))
        self.head_conv = put_conv(torch.cat([vw, pw], 0), torch.cat([vb, pb], 0))
        # collapse the activation-free Linear stack: y = W_k(...(W_1 x + b_1)...) + b_k
        W = torch.eye(AREA, dtype=torch.float64)
        bias = torch.zeros(AREA, dtype=torch.float64)
        for i in range(cfg.n_fc_layers):
            Wi = sd["value_head.fcN.%d.weight" % i].double()
            bi = sd["value_head.fcN.%d.bias" % i].double()
            W = Wi @ W
            bias = Wi @ bias + bi
        f32 = dict(device=self.device, dtype=torch.float32)
        self.v_fc_w, self.v_fc_b = W.to(**f32), bias.to(**f32)
        self.v_out_w = sd["value_head.fc1.weight"].to(**f32)
        self.v_out_b = sd["value_head.fc1.bias"].to(**f32)
        self.v_w1 = float(sd["value_head.w1"])
        self.v_w2 = float(sd["value_head.w2"])
        self.p_fc_w = sd["policy_head.fc1.weight"].to(**f32)
        self.p_fc_b = sd["policy_head.fc1.bias"].to(**f32)

    @classmethod
    def from_module(cls, net: nn.Module, **kw):
        return cls(net.state_dict(), **kw)

    @torch.no_grad()
    def __call__(self, planes: torch.Tensor):
        """planes [n,3,6,7] (any float dtype) -> (values fp32 [n], priors fp32 [n,7])."""
        x = planes.to(self.dtype).contiguous(memory_format=self.mf)
        x = F.leaky_relu(F.conv2d(x, self.stem[0], self.stem[1], padding=1), LEAK)
        for (w1, b1), (w2, b2) in self.res:
            y = F.leaky_relu(F.conv2d(x, w1, b1, padding=1), LEAK)
            y = F.conv2d(y, w2, b2, padding=1)
            x = F.leaky_relu(y + x, LEAK)
        h = F.leaky_relu(F.conv2d(x, self.head_conv[0], self.head_conv[1]), LEAK).float()
        n = h.shape[0]
        hv = h[:, 0].reshape(n, AREA)
        hp = h[:, 1:3].reshape(n, 2 * AREA)
        v = F.leaky_relu(F.linear(hv, self.v_fc_w, self.v_fc_b), LEAK)
        v = torch.tanh(F.linear(v, self.v_out_w, self.v_out_b)).view(-1)
        values = (v + self.v_w1) * self.v_w2
        priors = torch.softmax(F.linear(hp, self.p_fc_w, self.p_fc_b), dim=1)
        return values, priors


def random_init_state_dict(config: Optional[NetConfig] = None, seed: int = 0):
    """Seeded random-init weights of the reference architecture (BASELINE configs 2-5: 'random-init
    resnet'); eval-mode BN statistics are the initial mean 0 / var 1."""
    gen_state = torch.random.get_rng_state()
    torch.manual_seed(seed)
    try:
        net = PolicyValueNet(config)
    finally:
        torch.random.set_rng_state(gen_state)
    net.eval()
    return net.state_dict()
