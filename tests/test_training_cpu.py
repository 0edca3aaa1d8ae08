"""Train step (stock PyTorch) -- CPU smoke: loss falls, checkpoint round-trips with the
reference's key names."""
import os

import torch


def test_train_step_and_checkpoint(tmp_path):
    from connect4_amd.training import ModelConfig, Trainer
    torch.manual_seed(0)
    cfg = ModelConfig(batch_size=64, n_training_epochs=25, initial_lr=0.05, use_gpu=False)
    tr = Trainer(cfg)
    g = torch.Generator().manual_seed(0)
    boards = (torch.rand(256, 3, 6, 7, generator=g) > 0.7).float()
    values = boards[:, 1].mean((1, 2))                      # learnable targets
    priors = torch.softmax(torch.arange(7.0), 0).repeat(256, 1)
    with torch.no_grad():
        v0, p0 = tr.net(boards)
        before = float(tr.value_loss(v0, values) + tr.prior_loss(p0, priors))
    tr.train(boards, values, priors, generator=g)
    with torch.no_grad():
        v1, p1 = tr.net(boards)
        after = float(tr.value_loss(v1, values) + tr.prior_loss(p1, priors))
    assert after < before
    tr.save(str(tmp_path))
    ck = torch.load(os.path.join(str(tmp_path), "net.pth"), weights_only=True)
    assert set(ck) == {"net_state_dict", "optimiser_state_dict", "scheduler_state_dict"}
    tr2 = Trainer(cfg, os.path.join(str(tmp_path), "net.pth"))
    assert all(torch.equal(a, b) for a, b in zip(tr.net.state_dict().values(), tr2.net.state_dict().values()))


def test_train_step_reproduces_reference_model_wrapper_train():
    """Trainer.train vs the unmodified reference's ModelWrapper.train (model.py:200-240) -- fixture
    tests/golden/train_step.npz written by gen_golden.py: same seeded initial weights, same 200-position
    dataset (3 full batches of 64 + a ragged one), same torch seed -> same shuffles, same SGD/momentum/weight
    decay steps, same BatchNorm statistics, same scheduler step.  Tolerance 1e-6 (fp32, CPU)."""
    import numpy as np
    from conftest import load_npz
    from connect4_amd.training import ModelConfig, Trainer
    z = load_npz("train_step.npz")
    torch.set_num_threads(1)
    bs, epochs = [int(x) for x in z["config"]]
    tr = Trainer(ModelConfig(batch_size=bs, n_training_epochs=epochs, use_gpu=False))
    tr.net.load_state_dict({k[6:]: torch.from_numpy(z[k]) for k in z.files if k.startswith("init__")})
    boards = torch.from_numpy(z["data_boards"].astype(np.float32))
    values, priors = torch.from_numpy(z["data_values"]), torch.from_numpy(z["data_priors"])
    torch.manual_seed(int(z["torch_seed"][0]))
    loss = tr.train(boards, values, priors)
    assert loss == loss
    sd = tr.net.state_dict()
    finals = [k for k in z.files if k.startswith("final__")]
    assert len(finals) == len(sd)
    worst = 0.0
    for k in finals:
        a, b = sd[k[7:]].detach().numpy().astype(np.float64), z[k].astype(np.float64)
        assert a.shape == b.shape
        worst = max(worst, float(np.abs(a - b).max()))
    assert worst <= 1e-6, worst
    names = [k for k, _ in tr.net.named_parameters()]
    for k, p in zip(names, tr.net.parameters()):
        if "momentum__" + k in z.files:
            assert np.abs(tr.optimiser.state[p]["momentum_buffer"].numpy() - z["momentum__" + k]).max() <= 1e-6
    assert [g["lr"] for g in tr.optimiser.param_groups] == z["lr_after"].tolist()
    with torch.no_grad():
        xv, xp = tr.net(boards[:32])
    assert np.abs(xv.numpy() - z["eval_values"]).max() <= 1e-6 and np.abs(xp.numpy() - z["eval_priors"]).max() <= 1e-6


def test_dataloader_permutation_consumes_rng_like_dataloader():
    """The batches Trainer.train sees are the ones DataLoader(shuffle=True) would collate."""
    from torch.utils.data import DataLoader, TensorDataset
    from connect4_amd.training import dataloader_permutation
    x = torch.arange(37)
    torch.manual_seed(5)
    want = [b[0].tolist() for _ in range(3) for b in DataLoader(TensorDataset(x), batch_size=8, shuffle=True)]
    torch.manual_seed(5)
    got = []
    for _ in range(3):
        perm = dataloader_permutation(37)
        got += [x[perm[i:i + 8]].tolist() for i in range(0, 37, 8)]
    assert got == want


def test_padded_ragged_batch_trains_exactly_like_the_ragged_batch():
    """The trainer's GPU default pads the ragged last batch of an epoch to the full batch size (so every convolution
    keeps the one shape MIOpen has kernels for) and takes the batch-norm statistics from the real rows only.  That must
    be the same training step: identical weights, running statistics and loss as the unpadded ragged batch."""
    import copy
    import torch
    from connect4_amd.training import ModelConfig, Trainer
    torch.manual_seed(3)
    n = 2 * 64 + 23                       # two full batches and a ragged one of 23 rows
    boards = (torch.rand(n, 3, 6, 7) > 0.7).float()
    values = torch.rand(n)
    priors = torch.softmax(torch.rand(n, 7), 1)
    cfg = ModelConfig(batch_size=64, n_training_epochs=2, use_gpu=False)
    a = Trainer(cfg, device="cpu", pad_ragged_batches=False)
    b = Trainer(cfg, device="cpu", pad_ragged_batches=True)
    b.net.load_state_dict(copy.deepcopy(a.net.state_dict()))
    ga, gb = torch.Generator().manual_seed(11), torch.Generator().manual_seed(11)
    la = a.train(boards, values, priors, generator=ga)
    lb = b.train(boards, values, priors, generator=gb)
    assert abs(la - lb) < 1e-6
    sa, sb = a.net.state_dict(), b.net.state_dict()
    for k in sa:
        assert torch.allclose(sa[k].double(), sb[k].double(), atol=2e-6, rtol=1e-5), k
    assert int(sa["body.0.1.num_batches_tracked"]) == int(sb["body.0.1.num_batches_tracked"]) == 6
