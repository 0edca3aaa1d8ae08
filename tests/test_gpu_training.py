"""The train step as a GPU actually runs it -- Trainer(device="cuda"): ragged last batch padded to the full batch size,
batch-norm statistics from the real rows only (connect4_amd/training.py, net._BatchNorm2d) -- against the reference's
ModelWrapper.train (oinkoink/neural/pytorch/model.py:200-240) through the fixture tests/golden/train_step.npz that
gen_golden.py wrote from the unmodified reference on CPU: same seeded initial weights, same 200-position dataset (three
full batches of 64 and a ragged one of 8), same torch seed -> same shuffles, SGD / momentum / weight decay, scheduler.

Stated tolerances (float32 on both sides; MIOpen / rocBLAS reduce in another order than the CPU kernels, and the difference
passes through 20 optimiser steps): 5e-5 absolute on every weight and batch-norm statistic, 5e-4 on the momentum buffers
(sums of gradients: the largest numbers in the fixture), 1e-5 on the eval-mode outputs.  Measured on MI355X: 4.9e-6,
1.2e-4, 1.5e-7 (printed by the test)."""
import numpy as np
import pytest
import torch

from conftest import load_npz

pytestmark = pytest.mark.gpu


def boards_full_batches(z, bs):
    return int(z["data_boards"].shape[0]) // bs


@pytest.mark.parametrize("pad,fused_bn,use_graph", [(True, True, True), (False, True, True), (True, True, False), (True, False, False),
                                                    (False, False, False)])
def test_gpu_train_step_matches_reference_fixture(pad, fused_bn, use_graph):
    """(True, True, True) is what Trainer(device="cuda") does by default: the library's HIP batch-norm kernels, the full
    batches replayed from one captured HIP graph, the ragged batch padded; the other rows switch each of the three off."""
    from connect4_amd.training import ModelConfig, Trainer
    z = load_npz("train_step.npz")
    bs, epochs = [int(x) for x in z["config"]]
    default = Trainer(ModelConfig(batch_size=bs, n_training_epochs=epochs), device="cuda")
    assert default.pad_ragged_batches and default.fused_bn and default.use_graph
    tr = Trainer(ModelConfig(batch_size=bs, n_training_epochs=epochs), device="cuda", pad_ragged_batches=pad, fused_bn=fused_bn, use_graph=use_graph)
    assert tr.device.type == "cuda" and tr.pad_ragged_batches == pad and tr.use_graph == use_graph
    assert not Trainer(ModelConfig(), device="cuda", fused_bn=False, use_graph=True).use_graph     # (graph replay needs the library's batch norm)
    assert (boards_full_batches(z, bs) * epochs >= 4) and all(m.fused == fused_bn for m in tr.net.modules() if hasattr(m, "fused"))
    tr.net.load_state_dict({k[6:]: torch.from_numpy(z[k]) for k in z.files if k.startswith("init__")})
    boards = torch.from_numpy(z["data_boards"].astype(np.float32))
    values, priors = torch.from_numpy(z["data_values"]), torch.from_numpy(z["data_priors"])
    assert boards.shape[0] % bs not in (0, 1) and boards.shape[0] > bs        # the fixture HAS a ragged last batch
    torch.manual_seed(int(z["torch_seed"][0]))      # the shuffles come from the CPU generator, as in the reference's DataLoader
    loss = tr.train(boards, values, priors)
    assert loss == loss
    sd = {k: v.detach().cpu() for k, v in tr.net.state_dict().items()}
    finals = [k for k in z.files if k.startswith("final__")]
    assert len(finals) == len(sd)
    worst_w = 0.0
    for k in finals:
        a, b = sd[k[7:]].numpy().astype(np.float64), z[k].astype(np.float64)
        assert a.shape == b.shape
        worst_w = max(worst_w, float(np.abs(a - b).max()))
    assert int(sd["body.0.1.num_batches_tracked"]) == int(z["final__body.0.1.num_batches_tracked"])
    worst_m = 0.0
    for k, p in tr.net.named_parameters():
        if "momentum__" + k in z.files:
            worst_m = max(worst_m, float(np.abs(tr.optimiser.state[p]["momentum_buffer"].cpu().numpy() - z["momentum__" + k]).max()))
    assert [g["lr"] for g in tr.optimiser.param_groups] == z["lr_after"].tolist()
    with torch.no_grad():
        xv, xp = tr.net(boards[:32].cuda())
    worst_o = max(float(np.abs(xv.cpu().numpy() - z["eval_values"]).max()), float(np.abs(xp.cpu().numpy() - z["eval_priors"]).max()))
    print("GPU train step (pad=%s fused_bn=%s graph=%s) vs reference fixture: weights %.3g  momentum %.3g  eval outputs %.3g"
          % (pad, fused_bn, use_graph, worst_w, worst_m, worst_o))
    assert worst_w <= 5e-5 and worst_m <= 5e-4 and worst_o <= 1e-5, (worst_w, worst_m, worst_o)


@pytest.mark.parametrize("rows,ch,valid,residual,slope", [(4096, 32, None, True, 0.01), (4096, 32, None, False, 0.01), (4096, 32, 1000, True, 0.01),
                                                           (300, 1, None, False, 0.01), (37, 2, 20, False, 0.01), (64, 32, None, False, None),
                                                           (5000, 64, None, True, 0.2), (70000, 2, None, False, 0.01), (1, 32, None, True, 0.01)])
def test_fused_batch_norm_kernels_against_float64_autograd(rows, ch, valid, residual, slope):
    """c4_bn_train_forward / c4_bn_train_backward (the library's HIP kernels behind net._BatchNorm2d on a GPU) against the
    same function written with stock operators in float64 (model.py:20-55: act(bn(x) + residual), batch statistics from
    the valid rows): outputs, running statistics, and all four gradients (70,000 rows: chunks of 128 rows, the two-read statistics
    kernels instead of the register-resident one; one row: the ragged batch DataLoader(drop_last=False) may end an epoch with).  Stated tolerance: 2e-5 relative to each
    tensor's largest entry (float32 kernels, fixed-order reductions, float64 across chunks)."""
    import torch.nn.functional as F
    from connect4_amd.bn_train import fused_bn_act
    from connect4_amd.net import _BatchNorm2d
    g = torch.Generator().manual_seed(rows + ch)
    x = (torch.randn(rows, ch, 6, 7, generator=g) * 1.7 + 0.3).cuda().requires_grad_(True)
    res = torch.randn(rows, ch, 6, 7, generator=g).cuda().requires_grad_(True) if residual else None
    dy = torch.randn(rows, ch, 6, 7, generator=g).cuda()
    k = rows if valid is None else valid
    if valid is not None:
        dy[k:] = 0          # padded rows carry no loss (training.py)
    bn = _BatchNorm2d(ch).cuda().train()
    with torch.no_grad():
        bn.weight.copy_(torch.rand(ch, generator=g) + 0.5)
        bn.bias.copy_(torch.randn(ch, generator=g) * 0.1)
        bn.running_mean.copy_(torch.randn(ch, generator=g))
        bn.running_var.copy_(torch.rand(ch, generator=g) + 0.5)
    rm0, rv0 = bn.running_mean.double().clone(), bn.running_var.double().clone()
    y = fused_bn_act(x, bn, res, 1.0 if slope is None else slope, valid)
    y.backward(dy)
    # float64 reference
    xd = x.detach().double().requires_grad_(True)
    rd = res.detach().double().requires_grad_(True) if residual else None
    wd, bd = bn.weight.detach().double().requires_grad_(True), bn.bias.detach().double().requires_grad_(True)
    xs = xd[:k]
    mean, var = xs.mean(dim=(0, 2, 3)), xs.var(dim=(0, 2, 3), unbiased=False)
    z = (xd - mean.view(1, -1, 1, 1)) / torch.sqrt(var.view(1, -1, 1, 1) + bn.eps) * wd.view(1, -1, 1, 1) + bd.view(1, -1, 1, 1)
    if residual:
        z = z + rd
    yd = z if slope is None else F.leaky_relu(z, slope)
    yd.backward(dy.double())
    cnt = k * 42

    def close(a, b, what):
        err = float((a.double() - b).abs().max()) / max(float(b.abs().max()), 1e-30)
        assert err <= 2e-5, (what, err)
        return err
    errs = [close(y, yd.detach(), "y"), close(x.grad, xd.grad, "dx"), close(bn.weight.grad, wd.grad, "dweight"), close(bn.bias.grad, bd.grad, "dbias"),
            close(bn.running_mean, 0.9 * rm0 + 0.1 * mean.detach(), "running_mean"),
            close(bn.running_var, 0.9 * rv0 + 0.1 * var.detach() * cnt / (cnt - 1.0), "running_var")]
    if residual:
        errs.append(close(res.grad, rd.grad, "dresidual"))
    assert int(bn.num_batches_tracked) == 1
    print("fused bn %s: worst relative error %.3g" % ((rows, ch, valid, residual, slope), max(errs)))


def test_fused_batch_norm_is_reproducible_and_rejects_bad_shapes():
    from connect4_amd import _lib as L
    from connect4_amd.bn_train import fused_bn_act
    from connect4_amd.net import _BatchNorm2d
    x = torch.randn(777, 32, 6, 7, generator=torch.Generator().manual_seed(3)).cuda()
    outs = []
    for _ in range(2):
        bn = _BatchNorm2d(32).cuda().train()
        xx = x.clone().requires_grad_(True)
        y = fused_bn_act(xx, bn, None, 0.01, None)
        y.sum().backward()
        outs.append((y.detach().clone(), xx.grad.clone(), bn.weight.grad.clone(), bn.running_var.clone()))
    assert all(torch.equal(a, b) for a, b in zip(*outs))          # fixed-order reductions: bit-identical from run to run
    lib = L.load()
    assert lib.c4_bn_workspace_floats(0, 32) == L.EINVAL
    assert lib.c4_bn_train_forward(None, None, None, None, None, None, None, None, None, None, None, 4, 4, 32, 42, 0.1, 1e-5, 1.0, None) == L.EINVAL
    ws = torch.empty(int(lib.c4_bn_workspace_floats(8, 4)), device="cuda")
    t = torch.zeros(8, 4, 42, device="cuda")
    s = torch.zeros(2, 4, device="cuda")
    args = [t.data_ptr(), None, s[0].data_ptr(), s[1].data_ptr(), None, None, None, t.data_ptr(), s[0].data_ptr(), s[1].data_ptr(), ws.data_ptr()]
    assert lib.c4_bn_train_forward(*args, 8, 9, 4, 42, 0.1, 1e-5, 1.0, None) == L.EINVAL       # valid_rows > rows
    assert lib.c4_bn_train_forward(*args, 8, 8, 4, 300, 0.1, 1e-5, 1.0, None) == L.EINVAL      # a row wider than a workgroup


def test_gpu_trainer_one_row_ragged_batch_and_short_datasets():
    """DataLoader(drop_last=False) may end an epoch with ONE position (model.py:208-212): it is trained on un-padded, through the
    library's kernels; a dataset shorter than a batch, or with too few batches for a graph, runs eagerly.  Default GPU trainer
    against the stock-operator trainer on the same shuffles: same weights to float32 accuracy."""
    from connect4_amd.training import ModelConfig, Trainer
    g = torch.Generator().manual_seed(11)
    for n, bs, epochs in ((129, 64, 3), (40, 64, 2), (64 * 7 + 1, 64, 2)):
        b = (torch.rand(n, 3, 6, 7, generator=g) > 0.6).float()
        v = torch.rand(n, generator=g)
        p = torch.softmax(torch.rand(n, 7, generator=g), 1)
        sds = []
        for kw in (dict(), dict(fused_bn=False, use_graph=False)):
            torch.manual_seed(3)
            tr = Trainer(ModelConfig(batch_size=bs, n_training_epochs=epochs), device="cuda", **kw)
            torch.manual_seed(4)
            loss = tr.train(b, v, p)
            assert loss == loss
            sds.append({k: x.detach().double().cpu() for k, x in tr.net.state_dict().items()})
        assert int(sds[0]["body.0.1.num_batches_tracked"]) == int(sds[1]["body.0.1.num_batches_tracked"]) == epochs * -(-n // bs)
        worst = max(float((sds[0][k] - sds[1][k]).abs().max()) for k in sds[0] if not k.endswith("conv1.bias"))
        assert worst <= 2e-4, (n, worst)     # (the head convolutions' biases have a zero gradient in exact arithmetic: rounding noise only)


@pytest.mark.parametrize("rows", [4096, 1, 37, 1030, 5000])
def test_conv3x3_weight_gradient_kernel_against_float64(rows):
    """c4_conv3x3_wrw (behind bn_train.conv3x3, the tower's convolutions in the GPU train step) against the weight gradient of
    F.conv2d in float64; the input gradient and the forward stay MIOpen's and are compared with float64 too.  Stated
    tolerance: 2e-5 of the largest entry (a float32 fma chain over rows x 42 products per weight)."""
    import torch.nn.functional as F
    from connect4_amd.bn_train import conv3x3
    g = torch.Generator().manual_seed(rows)
    conv = torch.nn.Conv2d(32, 32, 3, padding=1, bias=False).cuda()
    x = torch.randn(rows, 32, 6, 7, generator=g).cuda().requires_grad_(True)
    dy = torch.randn(rows, 32, 6, 7, generator=g).cuda()
    y = conv3x3(conv, x)
    assert type(y.grad_fn).__name__ == "_Conv3x3Backward"
    y.backward(dy)
    xd = x.detach().double().requires_grad_(True)
    wd = conv.weight.detach().double().requires_grad_(True)
    yd = F.conv2d(xd, wd, None, 1, 1)
    yd.backward(dy.double())
    for a, b, what in ((y.detach(), yd.detach(), "y"), (x.grad, xd.grad, "dx"), (conv.weight.grad, wd.grad, "dweight")):
        err = float((a.double() - b).abs().max()) / float(b.abs().max())
        assert err <= 2e-5, (what, err)
    # bit-reproducible
    g1 = conv.weight.grad.clone()
    conv.weight.grad = None
    conv3x3(conv, x).backward(dy)
    assert torch.equal(g1, conv.weight.grad)
    # other shapes keep PyTorch's own backward
    other = torch.nn.Conv2d(3, 32, 3, padding=1, bias=False).cuda()
    assert type(conv3x3(other, torch.randn(4, 3, 6, 7, device="cuda")).grad_fn).__name__ != "_Conv3x3Backward"
