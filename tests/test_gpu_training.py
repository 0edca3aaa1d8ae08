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


@pytest.mark.parametrize("pad", [True, False])
def test_gpu_train_step_matches_reference_fixture(pad):
    from connect4_amd.training import ModelConfig, Trainer
    z = load_npz("train_step.npz")
    bs, epochs = [int(x) for x in z["config"]]
    tr = Trainer(ModelConfig(batch_size=bs, n_training_epochs=epochs), device="cuda", pad_ragged_batches=pad)
    assert tr.device.type == "cuda" and tr.pad_ragged_batches == pad
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
    print("GPU train step (pad=%s) vs reference fixture: weights %.3g  momentum %.3g  eval outputs %.3g" % (pad, worst_w, worst_m, worst_o))
    assert worst_w <= 5e-5 and worst_m <= 5e-4 and worst_o <= 1e-5, (worst_w, worst_m, worst_o)
