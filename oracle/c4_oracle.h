/*
 * c4_oracle.h -- CPU restatement of the oinkoink (willis-richard/connect4) self-play/MCTS hot path.
 *
 * TEST INFRASTRUCTURE ONLY.  Nothing under connect4_amd/ (the product) may include, link, import or
 * call this.  Allowed users: tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg.
 *
 * Parity status: PINNED.  The restatement is checked (tests/test_oracle_golden.py) against
 *  - the reference's own known-answer tests (tests/board_test.py:10-161, :164-247;
 *    tests/player_test.py:13-179), exported as data into tests/golden/ref_tests.json, and
 *  - outputs of the unmodified reference imported in the build container
 *    (tests/golden/gen_golden.py; fixtures under tests/golden/).
 *
 * Every function cites the reference file:line (relative to /root/reference) it follows.
 * Deliberately written as a *sequential, pointer-linked, lazily expanded* tree (like the reference)
 * so that it is structurally independent from the batched SoA engine it checks.
 */
#ifndef C4_ORACLE_H
#define C4_ORACLE_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* ---- utils.py:19-22 : Result enum, encoded so that value = code * 0.5 ---- */
#define C4O_NONE  (-1)
#define C4O_XWIN  0 /* Result.x_win = 0.0 */
#define C4O_DRAW  1 /* Result.draw  = 0.5 */
#define C4O_OWIN  2 /* Result.o_win = 1.0 */

/* board.py:36-41 : color[2], age, result.  height[] is derivable (7*col + stones in col). */
typedef struct {
    uint64_t color[2];
    int32_t age;
    int32_t result;
} c4o_board;

void     c4o_board_init(c4o_board *b);                                   /* board.py:36-41   */
int      c4o_board_from_bits(c4o_board *b, uint64_t c0, uint64_t c1);    /* board.py:44-62 (result rules) */
int      c4o_board_from_pieces(c4o_board *b, const uint8_t o[42], const uint8_t x[42]); /* board.py:44-62 */
int      c4o_wins(uint64_t stones);                                      /* board.py:173-184 */
int      c4o_make_move(c4o_board *b, int col);                           /* board.py:160-170 */
int      c4o_valid_mask(const c4o_board *b);                             /* board.py:88-92,187-188 */
void     c4o_planes(const c4o_board *b, uint8_t out[126]);               /* board.py:64-78,147-154 */
uint64_t c4o_flip_color(uint64_t stones);                                /* board.py:128-145 */
void     c4o_fliplr(const c4o_board *b, c4o_board *out);                 /* board.py:115-126 */
int      c4o_make_random_ips(int plies, uint64_t *c0, uint64_t *c1, int cap); /* board.py:225-243 (as sorted set) */

/* evaluators.py:28-38,48-63 : centre heuristic, uniform prior (float64) */
double   c4o_evaluate_centre(const c4o_board *b);

/* ---- MCTS (mcts.py, tree.py) ---- */
typedef struct {
    int32_t simulations;                 /* mcts.py:15 */
    int32_t pb_c_base;                   /* mcts.py:16 */
    double  pb_c_init;                   /* mcts.py:17 */
    double  root_dirichlet_alpha;        /* mcts.py:18 */
    double  root_exploration_fraction;   /* mcts.py:19 */
    int32_t num_sampling_moves;          /* mcts.py:20 */
} c4o_config;

/* Evaluator protocol (evaluators.py:18-25): board -> (value, prior[7]).
 * Return value: 1 when the prior is a float32 ndarray (net output; NumPy>=2 keeps the PUCT score in
 * float32 then), 0 when it is float64 (heuristic).  Negative = error. */
typedef int (*c4o_eval_fn)(void *ctx, const c4o_board *b, double *value, double prior[7]);

typedef struct c4o_tree c4o_tree;

c4o_tree *c4o_tree_new(const c4o_config *cfg, const c4o_board *root);    /* tree.py:62-64 */
void      c4o_tree_free(c4o_tree *t);

/* Resumable form of mcts.py:94-121 so many games can be advanced in lock-step on the CPU
 * (cpu_baseline) with batched leaf evaluation.  Protocol:
 *   c4o_tree_root_request(t, &leaf)                 -> board to evaluate (the root)
 *   c4o_tree_root_apply(t, v, prior, f32, noise)    mcts.py:101-105
 *   repeat `simulations` times:
 *     r = c4o_tree_select(t, &leaf)                 mcts.py:108-116; r==0: terminal leaf, sim complete
 *     if r==1: c4o_tree_apply(t, v, prior, f32)     mcts.py:118-120
 */
void c4o_tree_root_request(c4o_tree *t, c4o_board *leaf);
void c4o_tree_root_apply(c4o_tree *t, double value, const double prior[7], int prior_f32,
                         const double *gamma_noise /* 7 raw Gamma(alpha,1) draws or NULL */);
int  c4o_tree_select(c4o_tree *t, c4o_board *leaf);
void c4o_tree_apply(c4o_tree *t, double value, const double prior[7], int prior_f32);

/* mcts.py:94-121 in one call. */
int  c4o_search(c4o_tree *t, c4o_eval_fn eval, void *ctx, const double *gamma_noise);

/* Root read-out (tree.py:66-117). Arrays are indexed by column; absent children give N=0,W=0,
 * status C4O_NONE-1 (=-2). */
typedef struct {
    uint32_t root_visits;
    double   root_value_sum;
    uint32_t child_visits[7];
    double   child_value_sum[7];
    int32_t  child_status[7];   /* -2 absent, -1 non-terminal, else C4O_* result */
    double   child_value[7];    /* tree.py:66-67 get_node_value: from the root mover's side */
    double   values_policy[7];  /* tree.py:104-109 */
    double   visit_policy[7];   /* tree.py:111-117 */
    double   root_prior[7];     /* after noise */
    int32_t  best_move;         /* tree.py:69-73 */
    int64_t  n_nodes;
    int64_t  n_expansions;      /* expand_node calls that created children (mcts.py:115) */
    int64_t  n_children_created;
    int64_t  n_terminal_sims;
    int64_t  n_evals;           /* evaluator calls (no memo table) */
    int64_t  depth_sum;         /* sum over sims of leaf depth */
    int32_t  depth_max;
} c4o_root_info;
void c4o_tree_root_info(const c4o_tree *t, c4o_root_info *out);

/* mcts.py:78-88 move choice.  u in [0,1) is the uniform np.random.choice would draw (tree.py:75-82);
 * pass a negative u to force best_move.  Returns the move or -1 (all-zero sampling weights: the
 * reference raises there).  *abs_value receives child.data.absolute_value (NaN if None). */
int  c4o_tree_pick_move(const c4o_tree *t, int board_age, double u, double *abs_value);

/* training_game.py:8-19 : one self-play game.  noise_tape: [ply][7] raw gamma draws (or NULL),
 * u_tape: [ply] uniforms (or NULL => best_move only).  Records are per ply. */
typedef struct {
    uint64_t color0, color1;   /* board before the move (training_game.py:12) */
    int32_t  move;
    double   value;            /* child.data.absolute_value */
    double   policy[7];        /* tree.get_values_policy() */
} c4o_move_record;
int c4o_selfplay_game(const c4o_config *cfg, c4o_eval_fn eval, void *ctx,
                      const double *noise_tape, const double *u_tape,
                      c4o_move_record *records /* cap 42 */, int *result,
                      int64_t *n_sims, int64_t *n_expansions, int64_t *n_evals);

/* Built-in evaluators usable as c4o_eval_fn. */
int c4o_eval_centre_with_prior(void *ctx, const c4o_board *b, double *value, double prior[7]);

/* Table evaluator: sorted array of entries, binary search on (c0,c1).  Missing key => -1. */
typedef struct { uint64_t c0, c1; float value; float prior[7]; } c4o_table_entry;
typedef struct { const c4o_table_entry *entries; int64_t n; int prior_f32; int64_t misses; } c4o_table;
int c4o_eval_table(void *ctx, const c4o_board *b, double *value, double prior[7]);

/* ---- lock-step many-game driver for the CPU baseline (game_pool.py + inference_server.py shape) ---- */
typedef struct c4o_pool c4o_pool;
void      c4o_set_threads(int n);   /* OpenMP threads used by the pool */
c4o_pool *c4o_pool_new(const c4o_config *cfg, int n_games, uint64_t seed);
void      c4o_pool_free(c4o_pool *p);
/* Advance every game until it needs a leaf evaluation; writes one board per game (planes uint8
 * [n][126]).  Then the caller evaluates the batch and calls c4o_pool_apply. Uses OpenMP if built
 * with it. */
void      c4o_pool_collect(c4o_pool *p, uint8_t *planes);
void      c4o_pool_apply(c4o_pool *p, const float *values, const float *priors);
void      c4o_pool_stats(const c4o_pool *p, int64_t *sims, int64_t *expansions, int64_t *games,
                         int64_t *moves, int64_t *evals);

/* ---- lock-step replay pool (parity tests at full size): n tape-driven games of training_game.py:8-19 behind ONE
 * memoising evaluator table (evaluators.py:18-25).  Protocol:
 *   m = c4o_replay_collect(r, c0, c1, game_of)   every unfinished game runs until it needs a position the table lacks;
 *                                                m such positions come back (arrays of capacity n); 0 = all games ended
 *   c4o_replay_apply(r, m, game_of, values, priors)   the caller's answers (float32, as a net gives them)
 *   c4o_replay_game(r, i, rec, &result)               the game's plies (returns its length; < 0: not finished / failed) */
typedef struct c4o_replay c4o_replay;
c4o_replay *c4o_replay_new(const c4o_config *cfg, int n_games, const double *noise_tapes /* [n][42][7] or NULL */,
                           const double *u_tapes /* [n][42] or NULL */);
void        c4o_replay_free(c4o_replay *r);
int         c4o_replay_collect(c4o_replay *r, uint64_t *c0, uint64_t *c1, int32_t *game_of);
void        c4o_replay_apply(c4o_replay *r, int m, const int32_t *game_of, const float *values, const float *priors);
int         c4o_replay_game(const c4o_replay *r, int i, c4o_move_record *rec /* cap 42 */, int *result);
void        c4o_replay_stats(const c4o_replay *r, int64_t *lookups, int64_t *hits, int64_t *table_entries);

#ifdef __cplusplus
}
#endif
#endif
