"""The PRODUCTION random streams (C4_RNG_PHILOX): Marsaglia-Tsang Gamma(alpha) draws on Philox4x32-10 for the
root's Dirichlet noise (mcts.py:171-181) and the inverse-CDF sample of the opening moves proportional to value^2
(mcts.py:81-82, tree.py:75-82).  Parity tests inject tapes, so nothing else exercises these; here their
distributions are tested through read-outs that run the very device functions the kernels call
(c4_debug_root_noise / c4_debug_sample_move).  All bounds are >= 5 sigma for the sample sizes used."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu

N = 120_000
ALPHA = 0.3


def _keys(n, seed=0):
    rng = np.random.RandomState(seed)
    return rng.randint(0, 1 << 40, size=n).astype(np.int64), rng.randint(0, 42, size=n).astype(np.int32)


def test_gamma_draws_follow_gamma_alpha():
    from scipy import stats
    from connect4_amd.engine import debug_root_noise
    gid, ply = _keys(N)
    raw, _ = debug_root_noise(7, ALPHA, gid, ply, np.full(N, 0x7f, dtype=np.int32))
    x = raw.reshape(-1)
    n = x.size
    assert np.all(x >= 0) and np.all(np.isfinite(x))
    assert abs(x.mean() - ALPHA) < 5 * np.sqrt(ALPHA / n)                       # mean alpha, variance alpha
    assert abs(x.var() - ALPHA) < 5 * np.sqrt((6 * ALPHA + 2 * ALPHA ** 2) / n)  # var of the sample variance
    # distribution: Kolmogorov-Smirnov against Gamma(alpha, 1) on a 50k subsample (p-value floor 1e-4)
    d, pval = stats.kstest(x[:50_000], stats.gamma(ALPHA).cdf)
    assert pval > 1e-4, (d, pval)
    # the 7 columns are independent streams: no correlation between them
    c = np.corrcoef(raw.T)
    assert np.abs(c - np.eye(7)).max() < 5 / np.sqrt(N)


@pytest.mark.parametrize("mask", [0x7f, 0x5d, 0x41, 0x08])
def test_dirichlet_noise_moments_per_legal_move_count(mask):
    from connect4_amd.engine import debug_root_noise
    gid, ply = _keys(N, seed=mask)
    _, d = debug_root_noise(11, ALPHA, gid, ply, np.full(N, mask, dtype=np.int32))
    legal = [c for c in range(7) if (mask >> c) & 1]
    k = len(legal)
    assert np.all(d[:, [c for c in range(7) if c not in legal]] == 0.0)
    assert np.allclose(d.sum(1), 1.0, atol=1e-12)
    if k == 1:
        assert np.all(d[:, legal[0]] == 1.0)
        return
    # Dirichlet(alpha,...,alpha) over k legal moves: mean 1/k, var (1/k)(1-1/k)/(k alpha + 1)
    mean, var = 1.0 / k, (1.0 / k) * (1 - 1.0 / k) / (k * ALPHA + 1)
    for c in legal:
        assert abs(d[:, c].mean() - mean) < 5 * np.sqrt(var / N)
        assert abs(d[:, c].var() - var) < 0.03 * var
    # negative correlation between two components: -1/(k-1)
    r = np.corrcoef(d[:, legal[0]], d[:, legal[1]])[0, 1]
    assert abs(r + 1.0 / (k - 1)) < 0.02


def test_streams_are_disjoint_per_seed_game_and_ply():
    """rank r plays with seed + r (connect4_amd.distributed.rank_seed): same (game, ply) under neighbouring seeds,
    and neighbouring games / plies under one seed, give unrelated draws."""
    from connect4_amd.engine import debug_root_noise, debug_sample_move
    gid, ply = _keys(N, seed=3)
    full = np.full(N, 0x7f, dtype=np.int32)
    a, _ = debug_root_noise(100, ALPHA, gid, ply, full)
    b, _ = debug_root_noise(101, ALPHA, gid, ply, full)          # seed + 1  (the next rank)
    c, _ = debug_root_noise(100, ALPHA, gid + 1, ply, full)      # next game
    d, _ = debug_root_noise(100, ALPHA, gid, (ply + 1) % 42, full)
    again, _ = debug_root_noise(100, ALPHA, gid, ply, full)
    assert np.array_equal(a, again)                               # counter based: reproducible
    for other in (b, c, d):
        assert not np.any(a == other)
        assert abs(np.corrcoef(a.reshape(-1), other.reshape(-1))[0, 1]) < 5 / np.sqrt(a.size)
    V = np.tile(np.linspace(0.2, 0.8, 7), (N, 1))
    nc = np.full(N, 7, dtype=np.int32)
    u0, _ = debug_sample_move(100, gid, ply, V, nc)
    u1, _ = debug_sample_move(101, gid, ply, V, nc)
    assert np.all((u0 >= 0) & (u0 < 1)) and not np.any(u0 == u1)
    assert abs(u0.mean() - 0.5) < 5 / np.sqrt(12 * N) and abs(np.corrcoef(u0, u1)[0, 1]) < 5 / np.sqrt(N)
    # the move-choice uniform is not one of the noise streams
    assert abs(np.corrcoef(u0, a[:, 0])[0, 1]) < 5 / np.sqrt(N)


def test_sampled_move_frequencies_are_proportional_to_value_squared():
    from connect4_amd.engine import debug_sample_move
    gid, ply = _keys(N, seed=9)
    for vals in ([0.5, 0.1, 0.9, 0.3, 0.7, 0.0, 0.2], [0.4, 0.6], [0.0, 0.0, 1.0], [0.3]):
        nc = len(vals)
        V = np.zeros((N, 7))
        V[:, :nc] = vals
        _, ch = debug_sample_move(21, gid, ply, V, np.full(N, nc, dtype=np.int32))
        w = np.array(vals) ** 2
        prob = w / w.sum()
        freq = np.bincount(ch, minlength=nc)[:nc] / N
        assert ch.min() >= 0 and ch.max() < nc
        for k in range(nc):
            assert abs(freq[k] - prob[k]) <= 5 * np.sqrt(max(prob[k] * (1 - prob[k]), 1e-12) / N) + (0 if prob[k] > 0 else 0)
            if prob[k] == 0:
                assert freq[k] == 0
    # all-zero weights: the reference raises (probabilities contain NaN); the engine falls back to best_move (-1 here)
    _, ch = debug_sample_move(21, gid[:8], ply[:8], np.zeros((8, 7)), np.full(8, 4, dtype=np.int32))
    assert np.all(ch == -1)


def test_sampling_equals_numpy_choice_on_the_same_uniform(oracle):
    """Given the uniform, the device's choice is np.random.choice's (cdf = cumsum(p); cdf /= cdf[-1];
    searchsorted(cdf, u, 'right')) on weights x**2 -- checked on 100k random value vectors, and at constructed
    CDF boundaries where pow(x, 2.0) (CPython) and x*x (device) differ by one ulp: there the choice may only
    differ when u lies within 4 ulp of the boundary (the known deviation)."""
    import math
    from connect4_amd.engine import debug_sample_move
    rng = np.random.RandomState(1)
    n = 100_000
    nc = rng.randint(1, 8, size=n).astype(np.int32)
    V = rng.random_sample((n, 7))
    V[rng.random_sample((n, 7)) < 0.1] = 0.0
    for i in range(n):
        V[i, nc[i]:] = 0.0
    u = rng.random_sample(n)
    _, ch = debug_sample_move(0, np.zeros(n, np.int64), np.zeros(n, np.int32), V, nc, uniforms=u)

    def numpy_choice(vals, uu, square):
        w = np.array([square(x) for x in vals])
        if w.sum() == 0:
            return -1
        p = w / np.sum(w)
        cdf = np.cumsum(p)
        cdf /= cdf[-1]
        return int(np.searchsorted(cdf, uu, side="right"))
    mism = 0
    for i in range(n):
        want = numpy_choice(V[i, :nc[i]], u[i], lambda x: math.pow(x, 2.0))
        mism += int(want != ch[i])
    assert mism == 0
    # boundary construction: values whose libm square differs from the rounded product
    xs = [x for x in rng.random_sample(200_000) if math.pow(x, 2.0) != x * x][:64]
    assert len(xs) >= 16
    far = 0
    for x in xs:
        vals = [x, 0.5, 0.25]
        for square in (lambda t: math.pow(t, 2.0), lambda t: t * t):
            w = np.array([square(t) for t in vals])
            cdf = np.cumsum(w / np.sum(w))
            cdf /= cdf[-1]
            for delta in (-8, 0, 8):        # exactly on the boundary and 8 ulp either side
                uu = float(cdf[0]) + delta * np.spacing(cdf[0])
                Vb = np.zeros((1, 7))
                Vb[0, :3] = vals
                _, c = debug_sample_move(0, np.zeros(1, np.int64), np.zeros(1, np.int32), Vb, np.array([3], np.int32),
                                         uniforms=np.array([uu]))
                want = numpy_choice(vals, uu, lambda t: math.pow(t, 2.0))
                if delta != 0:
                    far += int(c[0] != want)
    assert far == 0      # 8 ulp away from a boundary the one-ulp difference in a weight cannot matter
