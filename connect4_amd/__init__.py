"""connect4_amd -- MI355X-native self-play / MCTS engine for Connect4.

Drop-in for the hot path of willis-richard/connect4 (`oinkoink`): the 6x7 bitboard, the MCTS
select/expand/evaluate/backup loop and batched self-play run as hand-written HIP kernels for gfx950
behind a C ABI (include/c4_engine.h); this package is the thin host-side mirror of the reference's
player / evaluator API.  Nothing here computes a search on the CPU.
"""
__version__ = "0.1.0"
