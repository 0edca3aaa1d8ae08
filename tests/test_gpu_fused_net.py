"""Fused gfx950 network kernel (csrc/c4_net.hip, through c4_net_* of the C ABI) vs
  (a) the reference's ModelWrapper outputs on example_net.pth (tests/golden/net_golden.npz) and
  (b) the fp32 PyTorch plan of the same weights on seeded random positions.
Tolerances (stated; the kernels' measured errors are printed): reference-precision mode "f32x3", the default at
32 filters: 5e-5 against the reference's own outputs, 2e-5 against the fp32 PyTorch-ROCm plan; the opt-in fp16-storage
mode (precision="f16") and the 64-filter forward: 2e-2 absolute on values and priors."""
import numpy as np
import pytest
import torch

from conftest import load_npz

pytestmark = pytest.mark.gpu


def random_positions(oracle, n, seed):
    rng = np.random.RandomState(seed)
    c0, c1 = [], []
    while len(c0) < n:
        b = oracle.Board.empty()
        for _ in range(int(rng.randint(0, 42))):
            m = b.valid_mask()
            if not m:
                break
            b.make_move(int(rng.choice([c for c in range(7) if (m >> c) & 1])))
        c0.append(b.key()[0])
        c1.append(b.key()[1])
    return np.array(c0, dtype=np.uint64), np.array(c1, dtype=np.uint64)


def test_fused_net_vs_reference_golden():
    from connect4_amd.fused_net import FusedNet
    z = load_npz("net_golden.npz")
    sd = {k[3:]: torch.from_numpy(z[k]) for k in z.files if k.startswith("w__")}
    net = FusedNet(sd, precision="f16")
    v, p = net.evaluate_bits(z["in_c0"], z["in_c1"])
    print("fused vs reference golden: max |dv| %.3g  max |dp| %.3g" %
          (np.abs(v - z["out_values"]).max(), np.abs(p - z["out_priors"]).max()))
    np.testing.assert_allclose(v, z["out_values"], atol=2e-2, rtol=0)
    np.testing.assert_allclose(p, z["out_priors"], atol=2e-2, rtol=0)
    np.testing.assert_allclose(p.sum(1), 1.0, atol=1e-5)


@pytest.mark.parametrize("n", [1, 15, 16, 17, 4099])
def test_fused_net_vs_pytorch_fp32(oracle, n):
    from connect4_amd.engine import board_planes
    from connect4_amd.fused_net import FusedNet
    from connect4_amd.net import InferenceNet, random_init_state_dict
    sd = random_init_state_dict(seed=0)
    c0, c1 = random_positions(oracle, n, seed=n)
    ref = InferenceNet(sd, device="cuda", dtype=torch.float32)
    rv, rp = ref(torch.from_numpy(board_planes(c0, c1)).cuda())
    v, p = FusedNet(sd, precision="f16").evaluate_bits(c0, c1)
    dv, dp = np.abs(v - rv.cpu().numpy()).max(), np.abs(p - rp.cpu().numpy()).max()
    print("n=%d fused vs torch fp32: max |dv| %.3g  max |dp| %.3g" % (n, dv, dp))
    assert dv < 2e-2 and dp < 2e-2


@pytest.mark.parametrize("n", [1, 2, 3, 16, 33, 4099])
def test_wave_private_forward_is_bit_identical(oracle, n):
    """c4_net_forward_wave (one wave = one position on 16-row MFMA tiles, no workgroup barrier: what the
    wave-autonomous self-play kernel evaluates its leaves with) must answer exactly what c4_net_forward
    answers (32-row tiles, 16 positions per workgroup) -- same per-element arithmetic, and one 16x16x32 MFMA
    step accumulates exactly like two 32x32x16 steps -- for any batch size, and with a different tower depth
    (the runtime-depth path)."""
    from connect4_amd.fused_net import FusedNet
    from connect4_amd.net import NetConfig, random_init_state_dict
    for n_res in (3, 1):
        sd = random_init_state_dict(NetConfig(n_residuals=n_res), seed=n_res)
        net = FusedNet(sd, precision="f16")
        c0, c1 = random_positions(oracle, n, seed=100 + n)
        v, p = net.evaluate_bits(c0, c1)
        wv, wp = net.evaluate_bits(c0, c1, wave=True)
        assert np.array_equal(v, wv) and np.array_equal(p, wp)


# ---------------------------------------------------------------- reference-precision mode (C4_NET_F32X3)
def test_precise_net_vs_reference_golden():
    """fp16 hi+lo split, three MFMAs per k-step, fp32 accumulation: the reference's own fp32 outputs on
    example_net.pth (tests/golden/net_golden.npz, written by the unmodified reference on CPU) within 5e-5."""
    from connect4_amd.fused_net import FusedNet
    z = load_npz("net_golden.npz")
    sd = {k[3:]: torch.from_numpy(z[k]) for k in z.files if k.startswith("w__")}
    net = FusedNet(sd, precision="f32x3")
    v, p = net.evaluate_bits(z["in_c0"], z["in_c1"])
    dv, dp = np.abs(v - z["out_values"]).max(), np.abs(p - z["out_priors"]).max()
    print("precise fused vs reference golden: max |dv| %.3g  max |dp| %.3g" % (dv, dp))
    assert dv <= 5e-5 and dp <= 5e-5
    np.testing.assert_allclose(p.sum(1), 1.0, atol=1e-5)
    wv, wp = net.evaluate_bits(z["in_c0"], z["in_c1"], wave=True)       # both entry points: one implementation
    assert np.array_equal(v, wv) and np.array_equal(p, wp)


@pytest.mark.parametrize("n", [1, 7, 8, 9, 4099])
def test_precise_net_vs_pytorch_fp32(oracle, n):
    """... and the fp32 PyTorch-ROCm plan of a random-init net on random positions (incl. ragged last
    workgroups and a second tower depth) within 2e-5."""
    from connect4_amd.engine import board_planes
    from connect4_amd.fused_net import FusedNet
    from connect4_amd.net import InferenceNet, NetConfig, random_init_state_dict
    for n_res in (3, 1):
        sd = random_init_state_dict(NetConfig(n_residuals=n_res), seed=n_res)
        c0, c1 = random_positions(oracle, n, seed=n)
        ref = InferenceNet(sd, device="cuda", dtype=torch.float32)
        rv, rp = ref(torch.from_numpy(board_planes(c0, c1)).cuda())
        v, p = FusedNet(sd, precision="f32x3").evaluate_bits(c0, c1)
        dv, dp = np.abs(v - rv.cpu().numpy()).max(), np.abs(p - rp.cpu().numpy()).max()
        print("n=%d n_res=%d precise vs torch fp32: max |dv| %.3g  max |dp| %.3g" % (n, n_res, dv, dp))
        assert dv < 2e-5 and dp < 2e-5


# ---------------------------------------------------------------- 64 filters (data/example_config.py:8-16)
@pytest.mark.parametrize("n", [1, 8, 9, 1031])
def test_fused_net_64_filters_vs_pytorch_fp32(oracle, n):
    """The reference's other shipped width: 64 filters, 6 residual blocks, 6 value-head Linear layers -- fused
    one-position forward (fp16 storage) vs the fp32 PyTorch-ROCm plan, same stated tolerance as the 32-filter
    fp16 kernel (2e-2); both entry points run the one implementation."""
    from connect4_amd.engine import board_planes
    from connect4_amd.fused_net import FusedNet
    from connect4_amd.net import InferenceNet, NetConfig, random_init_state_dict
    for cfg in (NetConfig(filters=64, n_fc_layers=6, n_residuals=6), NetConfig(filters=64, n_residuals=2)):
        sd = random_init_state_dict(cfg, seed=3)
        c0, c1 = random_positions(oracle, n, seed=n)
        ref = InferenceNet(sd, device="cuda", dtype=torch.float32)
        rv, rp = ref(torch.from_numpy(board_planes(c0, c1)).cuda())
        net = FusedNet(sd)
        v, p = net.evaluate_bits(c0, c1)
        wv, wp = net.evaluate_bits(c0, c1, wave=True)
        dv, dp = np.abs(v - rv.cpu().numpy()).max(), np.abs(p - rp.cpu().numpy()).max()
        print("n=%d 64f/%dres fused vs torch fp32: max |dv| %.3g  max |dp| %.3g" % (n, cfg.n_residuals, dv, dp))
        assert dv < 2e-2 and dp < 2e-2
        assert np.array_equal(v, wv) and np.array_equal(p, wp)
