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
