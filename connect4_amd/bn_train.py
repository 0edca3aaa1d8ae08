"""Training-mode batch normalisation on the GPU, fused with the residual add and LeakyReLU that follow it in the
reference's net (oinkoink/neural/pytorch/model.py:20-31, 36-55, 60-117): a torch.autograd.Function over the library's
c4_bn_train_forward / c4_bn_train_backward (connect4_amd/csrc/c4_train.hip).  Stock MIOpen spends half of a train
step (model.py:200-240) in batch normalisation at this net's shape ([4096, 32, 6, 7]: 77 us forward, 134 us backward per
layer); these kernels are HBM bound (~23 / ~42 us).  float32 NCHW, contiguous; no CPU path -- net._BatchNorm2d uses
this only for CUDA tensors in training mode and stock PyTorch otherwise.
"""
import torch

from . import _lib as L


def _ptr(t):
    return None if t is None else t.data_ptr()


class _FusedBNAct(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, weight, bias, residual, running_mean, running_var, num_batches_tracked, momentum, eps, slope, valid_rows):
        lib = L.load()
        x = x.contiguous()
        if residual is not None:
            residual = residual.contiguous()
        rows, ch = int(x.shape[0]), int(x.shape[1])
        hw = x.numel() // (rows * ch)
        k = rows if valid_rows is None else int(valid_rows)
        assert x.dtype == torch.float32 and x.is_cuda and weight.dtype == torch.float32
        y = torch.empty_like(x)
        save = torch.empty(2, ch, dtype=torch.float32, device=x.device)
        ws = torch.empty(int(lib.c4_bn_workspace_floats(rows, ch)), dtype=torch.float32, device=x.device)
        stream = torch.cuda.current_stream(x.device).cuda_stream
        with torch.cuda.device(x.device):       # the launch goes to the calling thread's current device
            L.check(lib.c4_bn_train_forward(_ptr(x), _ptr(residual), _ptr(weight), _ptr(bias), _ptr(running_mean), _ptr(running_var),
                                            _ptr(num_batches_tracked), _ptr(y), save[0].data_ptr(), save[1].data_ptr(), _ptr(ws),
                                            rows, k, ch, hw, float(momentum), float(eps), float(slope), stream))
        ctx.save_for_backward(x, y, weight, save)
        ctx.geo = (rows, k, ch, hw, float(slope), residual is not None)
        return y

    @staticmethod
    def backward(ctx, dy):
        lib = L.load()
        x, y, weight, save = ctx.saved_tensors
        rows, k, ch, hw, slope, has_res = ctx.geo
        dy = dy.contiguous()
        dx = torch.empty_like(x)
        dres = torch.empty_like(x) if has_res else None
        dwb = torch.empty(2, ch, dtype=torch.float32, device=x.device)
        ws = torch.empty(int(lib.c4_bn_workspace_floats(rows, ch)), dtype=torch.float32, device=x.device)
        stream = torch.cuda.current_stream(x.device).cuda_stream
        with torch.cuda.device(x.device):
            L.check(lib.c4_bn_train_backward(_ptr(x), _ptr(y), _ptr(dy), _ptr(weight), save[0].data_ptr(), save[1].data_ptr(), _ptr(dx), _ptr(dres),
                                             dwb[0].data_ptr(), dwb[1].data_ptr(), _ptr(ws), rows, k, ch, hw, slope, stream))
        return dx, dwb[0], dwb[1], dres, None, None, None, None, None, None, None


def fused_bn_act(x, bn, residual=None, slope=1.0, valid_rows=None):
    """act(bn(x) + residual) in training mode for a CUDA float32 tensor; `bn` is an nn.BatchNorm2d (its running
    statistics and num_batches_tracked are updated in place, as its own forward would)."""
    momentum = 0.1 if bn.momentum is None else bn.momentum
    track = bn.track_running_stats and bn.running_mean is not None
    return _FusedBNAct.apply(x, bn.weight, bn.bias, residual, bn.running_mean if track else None, bn.running_var if track else None,
                             bn.num_batches_tracked if track else None, momentum, bn.eps, slope, valid_rows)


class _Conv3x3(torch.autograd.Function):
    """F.conv2d(x, weight, None, padding=1) for the tower's 3x3 convolutions (model.py:36-55) whose weight gradient is the
    library's c4_conv3x3_wrw (one pass on the f32-input MFMA; MIOpen's pick is an NHWC implicit GEMM behind two layout
    transposes).  Forward and the input gradient stay MIOpen's (Winograd)."""

    @staticmethod
    def forward(ctx, x, weight):
        ctx.save_for_backward(x, weight)
        return torch.nn.functional.conv2d(x, weight, None, 1, 1)

    @staticmethod
    def backward(ctx, dy):
        x, weight = ctx.saved_tensors
        dx = dw = None
        if ctx.needs_input_grad[0]:
            dx = torch.ops.aten.convolution_backward(dy, x, weight, None, [1, 1], [1, 1], [1, 1], False, [0, 0], 1, [True, False, False])[0]
        if ctx.needs_input_grad[1]:
            lib = L.load()
            xc, dyc = x.contiguous(), dy.contiguous()
            dw = torch.empty_like(weight, memory_format=torch.contiguous_format)
            ws = torch.empty(int(lib.c4_conv3x3_wrw_workspace_floats()), dtype=torch.float32, device=x.device)
            stream = torch.cuda.current_stream(x.device).cuda_stream
            with torch.cuda.device(x.device):
                L.check(lib.c4_conv3x3_wrw(_ptr(xc), _ptr(dyc), _ptr(dw), _ptr(ws), int(x.shape[0]), int(x.shape[1]), int(x.shape[2]), int(x.shape[3]), stream))
        return dx, dw


def conv3x3(conv, x, fused=True):
    """conv(x) for an nn.Conv2d of the tower; the library's weight-gradient kernel behind it where it applies (CUDA float32,
    32 -> 32 filters on the 6 x 7 board, gradients wanted), the module's own forward otherwise."""
    w = conv.weight
    if (fused and x.is_cuda and x.dtype == torch.float32 and torch.is_grad_enabled() and w.requires_grad and conv.bias is None
            and tuple(w.shape) == (32, 32, 3, 3) and tuple(x.shape[1:]) == (32, 6, 7) and conv.padding == (1, 1) and conv.stride == (1, 1)):
        return _Conv3x3.apply(x, w)
    return conv(x)
