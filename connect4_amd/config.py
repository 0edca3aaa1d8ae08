"""Search configuration with the reference's field names (oinkoink/mcts.py:13-26) and the
self-play defaults of AlphaZeroConfig (oinkoink/neural/config.py:50-81)."""


class MCTSConfig:
    def __init__(self, simulations, pb_c_base=19652, pb_c_init=1.25, root_dirichlet_alpha=0.0,
                 root_exploration_fraction=0.0, num_sampling_moves=0):
        self.simulations = simulations
        self.pb_c_base = pb_c_base
        self.pb_c_init = pb_c_init
        self.root_dirichlet_alpha = root_dirichlet_alpha
        self.root_exploration_fraction = root_exploration_fraction
        self.num_sampling_moves = num_sampling_moves

    @classmethod
    def self_play(cls, simulations=800):
        """training.py:209-223 with training=True and config.py:56-62 defaults."""
        return cls(simulations, 19652, 1.25, 0.3, 0.25, 6)

    def engine_kwargs(self):
        return dict(simulations=self.simulations, pb_c_base=self.pb_c_base, pb_c_init=self.pb_c_init,
                    root_dirichlet_alpha=self.root_dirichlet_alpha,
                    root_exploration_fraction=self.root_exploration_fraction,
                    num_sampling_moves=self.num_sampling_moves)
