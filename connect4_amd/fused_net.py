"""Host wrapper of the fused gfx950 network kernel (csrc/c4_net.hip) behind c4_net_* of the C ABI.

``FusedNet(state_dict)`` folds BatchNorm (eval mode) and collapses the value head's Linear stack on
the host (same algebra as connect4_amd.net.InferenceNet), hands plain float32 arrays across the ABI,
and evaluates leaf batches straight from the engine's bitboard buffers:
``net.forward_bitboards(c0_ptr, c1_ptr, n, values, priors)``.
"""
import ctypes as C
from typing import Optional

import numpy as np

from . import _lib as L
from .net import PolicyValueNet, _fold_bn


class FusedNet:
    from_bitboards = True

    PRECISIONS = {"f16": 0, "f32x3": 1}

    @staticmethod
    def default_precision(filters: int) -> str:
        """The reference evaluates leaves in float32 (model.py:252-282), so the default is the mode that reproduces its
        visit counts: "f32x3" wherever the engine offers it (32 filters); the 64-filter forward exists in fp16 storage only."""
        return "f32x3" if filters == 32 else "f16"

    def __init__(self, state_dict, device: int = 0, precision: Optional[str] = None):
        """precision: "f32x3" = reference precision (the default at 32 filters): every fp32 operand split into fp16 hi +
        scaled lo parts, three MFMAs per k-step, fp32 accumulation; "f16" = fp16 storage / fp32 accumulation (one MFMA
        per k-step), opt-in: faster, answers within 2e-2 of the reference's instead of 5e-5."""
        import torch
        sd = {k: v.detach().cpu() for k, v in state_dict.items()}
        cfg = PolicyValueNet.config_from_state_dict(sd)
        if precision is None:
            precision = self.default_precision(cfg.filters)
        if precision not in self.PRECISIONS:
            raise ValueError("precision must be one of %s" % sorted(self.PRECISIONS))
        self.precision = precision
        self.config = cfg
        self.device = device

        def bn(prefix):
            return (sd[prefix + ".weight"], sd[prefix + ".bias"], sd[prefix + ".running_mean"], sd[prefix + ".running_var"])

        f32 = lambda t: np.ascontiguousarray(t.to(torch.float32).numpy())  # noqa: E731
        stem_w, stem_b = _fold_bn(sd["body.0.0.weight"], None, *bn("body.0.1"))
        cw, cb = [], []
        for i in range(cfg.n_residuals):
            p = "body.1.%d." % i
            for j in (1, 2):
                w, b = _fold_bn(sd[p + "conv%d.weight" % j], None, *bn(p + "batch_norm%d" % j))
                cw.append(w)
                cb.append(b)
        vw, vb = _fold_bn(sd["value_head.conv1.weight"], sd["value_head.conv1.bias"], *bn("value_head.batch_norm"))
        pw, pb = _fold_bn(sd["policy_head.conv1.weight"], sd["policy_head.conv1.bias"], *bn("policy_head.batch_norm"))
        W = torch.eye(42, dtype=torch.float64)
        bias = torch.zeros(42, dtype=torch.float64)
        for i in range(cfg.n_fc_layers):
            Wi = sd["value_head.fcN.%d.weight" % i].double()
            W, bias = Wi @ W, Wi @ bias + sd["value_head.fcN.%d.bias" % i].double()
        self._arrays = dict(
            stem_w=f32(stem_w), stem_b=f32(stem_b),
            conv_w=f32(torch.stack(cw)) if cw else np.zeros(1, np.float32),
            conv_b=f32(torch.stack(cb)) if cb else np.zeros(1, np.float32),
            head_w=f32(torch.cat([vw, pw], 0).reshape(3, cfg.filters)), head_b=f32(torch.cat([vb, pb], 0)),
            vfc_w=f32(W), vfc_b=f32(bias), vout_w=f32(sd["value_head.fc1.weight"].reshape(42)),
            pfc_w=f32(sd["policy_head.fc1.weight"]), pfc_b=f32(sd["policy_head.fc1.bias"]))
        d = L.NetDesc()
        d.channels, d.filters, d.n_residuals = cfg.channels, cfg.filters, cfg.n_residuals
        d.precision = self.PRECISIONS[precision]
        for k, a in self._arrays.items():
            setattr(d, k, a.ctypes.data_as(C.POINTER(C.c_float)))
        d.vout_b = float(sd["value_head.fc1.bias"].reshape(-1)[0])
        d.w1 = float(sd["value_head.w1"])
        d.w2 = float(sd["value_head.w2"])
        self._lib = L.load()
        self._h = C.c_void_p()
        rc = self._lib.c4_net_create(device, C.byref(d), C.byref(self._h))
        if rc != L.OK:
            raise L.EngineError(rc, (self._lib.c4_net_last_error() or b"").decode())

    def forward_bitboards(self, c0_ptr, c1_ptr, n, values, priors, stream=None, wave=False):
        """values/priors: torch float32 device tensors; c0_ptr/c1_ptr: device addresses of uint64[n].
        wave=True: the wave-private forward the fused self-play kernel uses (bit-identical answers)."""
        import torch
        if stream is None:
            stream = torch.cuda.current_stream(values.device).cuda_stream
        fn = self._lib.c4_net_forward_wave if wave else self._lib.c4_net_forward
        rc = fn(self._h, C.c_void_p(stream), C.c_void_p(c0_ptr), C.c_void_p(c1_ptr), int(n),
                C.c_void_p(values.data_ptr()), C.c_void_p(priors.data_ptr()))
        if rc != L.OK:
            raise L.EngineError(rc, (self._lib.c4_net_last_error() or b"").decode())

    def evaluate_bits(self, color0, color1, wave=False):
        """Convenience for tests: numpy uint64 arrays -> (values, priors) numpy."""
        import torch
        dev = torch.device("cuda", self.device)
        c0 = torch.from_numpy(np.ascontiguousarray(color0, dtype=np.uint64).view(np.int64)).to(dev)
        c1 = torch.from_numpy(np.ascontiguousarray(color1, dtype=np.uint64).view(np.int64)).to(dev)
        n = c0.numel()
        v = torch.zeros(n, dtype=torch.float32, device=dev)
        p = torch.zeros(n, 7, dtype=torch.float32, device=dev)
        self.forward_bitboards(c0.data_ptr(), c1.data_ptr(), n, v, p, wave=wave)
        torch.cuda.synchronize(dev)
        return v.cpu().numpy(), p.cpu().numpy()

    def close(self):
        if getattr(self, "_h", None) is not None and self._h:
            self._lib.c4_net_destroy(self._h)
            self._h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


def make_selfplay_net(state_dict, device: int = 0, precision: Optional[str] = None):
    """The fastest evaluator this build has for a checkpoint at the requested precision: the fused MFMA forwards for 32
    filters (the reference's default, config.py:8-12; reference precision "f32x3" unless "f16" is asked for) and for
    64 filters (its example_config, data/example_config.py:8-16; fp16 storage only, so precision=None or "f16"), up to
    16 / 7 residual blocks (their biases live in LDS) and any number of value-head Linear layers; anything else -- and
    64 filters with precision="f32x3" -- runs through the PyTorch-ROCm plan (connect4_amd.net.InferenceNet, fp32).
    All plug into SelfPlay / generate_games / DeviceNetEvaluator unchanged."""
    import torch
    cfg = PolicyValueNet.config_from_state_dict(state_dict)
    if precision is None:
        precision = FusedNet.default_precision(cfg.filters)
    fits = (cfg.filters == 32 and cfg.n_residuals <= 16) or (cfg.filters == 64 and cfg.n_residuals <= 7 and precision == "f16")
    if cfg.channels == 3 and fits:
        return FusedNet(state_dict, device=device, precision=precision)
    from .net import InferenceNet
    return InferenceNet(state_dict, device="cuda:%d" % device, dtype=torch.float32)
