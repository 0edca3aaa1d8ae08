"""Stand-in for `visdom` (absent here): the reference imports it at module load
(oinkoink/neural/training.py:20) but the hot path never touches it."""


class Visdom:
    def __init__(self, *a, **k):
        raise RuntimeError("visdom stand-in: plotting is not available")
