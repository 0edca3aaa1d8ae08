"""Helpers shared by the GPU parity tests: drive the HIP engine with a host-side evaluator."""
import numpy as np


def table_lookup_fn(c0s, c1s, vs, ps):
    table = {(int(a), int(b)): (np.float32(v), np.asarray(p, dtype=np.float32))
             for a, b, v, p in zip(c0s, c1s, vs, ps)}

    def fn(c0, c1):
        return table[(int(c0), int(c1))]
    return fn


def drive_external(eng, eval_fn, dtype, max_steps=10_000_000):
    """Run an EXTERNAL_* engine to completion with a Python evaluator (c0,c1)->(value, prior[7]).
    Mirrors how evaluators.py:18-25 serves mcts.py:130, one leaf per slot per step."""
    import torch
    G = eng.n_slots
    tdt = torch.float32 if dtype == np.float32 else torch.float64
    values = torch.zeros(G, dtype=tdt, device="cuda")
    priors = torch.zeros(G, 7, dtype=tdt, device="cuda")
    hv = np.zeros(G, dtype=dtype)
    hp = np.zeros((G, 7), dtype=dtype)
    eng.step(None, None, None)
    n_evals = 0
    for _ in range(max_steps):
        c0, c1, has = eng.read_leaves()
        if not has.any():
            if eng.stats()["active_slots"] == 0:
                break
        for g in np.nonzero(has)[0]:
            v, p = eval_fn(c0[g], c1[g])
            hv[g] = v
            hp[g] = p
            n_evals += 1
        values.copy_(torch.from_numpy(hv))
        priors.copy_(torch.from_numpy(hp))
        eng.step(values, priors, None)
    else:
        raise RuntimeError("engine did not finish")
    return n_evals
