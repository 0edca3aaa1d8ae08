/*
 * c4_engine.h -- C ABI of the MI355X-native Connect4 self-play / MCTS engine (libc4engine.so).
 *
 * This is the drop-in boundary for the hot path of willis-richard/connect4 (`oinkoink`).  The
 * reference has no FFI: its seam is two duck-typed Python protocols.  Each
 * entry point below names the reference interface it stands in for (file:line relative to the
 * reference root).  Host code (Python ctypes in connect4_amd/_lib.py, or any cgo/JNI/ctypes binding,
 * see INTEGRATION.md) calls ONLY these functions; no HIP, torch or C++ types cross the boundary.
 *
 * Conventions
 *   - every function returns 0 on success or a negative C4_E* code; nothing throws across the ABI;
 *     c4_last_error() returns a static/engine-owned message for the last failure;
 *   - the engine owns all tree memory in HBM; the caller owns I/O buffers and keeps them alive
 *     until the stream has been synchronised;
 *   - pointers named *_dev are DEVICE pointers (e.g. torch.Tensor.data_ptr()); all others are host;
 *   - one engine per GPU per process, one host thread drives a handle (mirrors "one player serves
 *     one game at a time", game_pool.py:29-32,45-49);
 *   - there is NO CPU backend: creating an engine without a usable gfx950 device fails loudly.
 */
#ifndef C4_ENGINE_H
#define C4_ENGINE_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define C4_ABI_VERSION 4

/* error codes */
#define C4_OK 0
#define C4_EINVAL (-1)   /* bad argument */
#define C4_EDEVICE (-2)  /* no usable HIP device / HIP call failed */
#define C4_ENOMEM (-3)
#define C4_ESTATE (-4)   /* call not valid in the current engine state */
#define C4_ECAPACITY (-5)

/* Result codes (utils.py:19-22): value = code * 0.5 */
#define C4_RESULT_NONE (-1)
#define C4_RESULT_XWIN 0
#define C4_RESULT_DRAW 1
#define C4_RESULT_OWIN 2

/* How leaf positions are evaluated (the `evaluator` argument of mcts.py:70-76 / :94-97). */
#define C4_EVAL_EXTERNAL_F32 0 /* values float32[n], priors float32[n][7] written by the caller (the
                                  net: model.py:269-282).  PUCT score arithmetic follows what the
                                  reference does with a float32 prior under NumPy>=2 (float32). */
#define C4_EVAL_EXTERNAL_F64 1 /* values float64[n], priors float64[n][7] (a Python evaluator
                                  returning float64, e.g. a table or heuristic): float64 scoring. */
#define C4_EVAL_CENTRE 2       /* evaluators.py:28-38 evaluate_centre_with_prior computed in-kernel;
                                  whole searches run inside one launch (no leaf round trip). */

#define C4_RNG_PHILOX 0 /* counter-based in-kernel RNG keyed by (seed, game id, ply) */
#define C4_RNG_TAPE 1   /* caller-injected Gamma draws / uniforms (parity tests: the reference uses
                           NumPy's global Mersenne state, mcts.py:175-177, tree.py:80) */

#define C4_PLANES_F32 0
#define C4_PLANES_F16 1
#define C4_PLANES_BF16 2

/* mcts.py:13-26 MCTSConfig + engine shape. */
typedef struct {
    int32_t abi_version;               /* = C4_ABI_VERSION */
    int32_t n_slots;                   /* games advanced in lock-step on this GPU */
    int32_t simulations;               /* mcts.py:15 */
    int32_t pb_c_base;                 /* mcts.py:16 (19652) */
    double  pb_c_init;                 /* mcts.py:17 (1.25) */
    double  root_dirichlet_alpha;      /* mcts.py:18 */
    double  root_exploration_fraction; /* mcts.py:19 */
    int32_t num_sampling_moves;        /* mcts.py:20 */
    int32_t eval_mode;                 /* C4_EVAL_* */
    int32_t rng_mode;                  /* C4_RNG_* */
    uint64_t seed;
    int32_t stop_after_move;           /* 1: a slot parks after choosing ONE move (MCTS.make_move,
                                          mcts.py:78-88); 0: continuous self-play (training_game.py:8-19) */
    int64_t games_target;              /* self-play: slots park once this many games were started;
                                          <0 = unbounded (bench) */
    int32_t record_capacity_games;     /* rows of the device ring of finished games (completion order).  Moves
                                          are staged per slot while a game is in flight and copied to a ring row
                                          when it ends; a game that finds every row unread is dropped and counted
                                          (c4_stats.dropped_games), never mixed with another game.  0 = 2 x n_slots */
    int32_t max_inner_iters;           /* cap on simulations a slot may complete inside one step call
                                          without needing the evaluator (terminal or cached leaves);
                                          bounds the duration of c4_step and of one tree call inside
                                          the fused kernels.  0 = default */
    int32_t planes_dtype;              /* C4_PLANES_* layout of the leaf batch handed to the net */
    int32_t eval_cache_log2_entries;   /* device-wide evaluation cache = the reference's memo table
                                          (evaluators.py:9-25), shared by every slot and game: a leaf
                                          whose position was already answered is applied in place, no
                                          evaluator round trip.  Results are unchanged for a
                                          deterministic evaluator.  <0 off, 0 auto (on for self-play
                                          with C4_EVAL_EXTERNAL_F32: 256 x n_slots x simulations entries
                                          of 48 bytes, at most 2^30 and a quarter of the free device
                                          memory), else log2 of the table size */
    int32_t level_budget;              /* descent levels a slot may walk per launch; a descent that runs
                                          out is suspended and resumed by the next launch, so no launch
                                          waits for the deepest tree of the batch.  0 = unlimited */
    int32_t time_budget_cycles;        /* c4_step / block-mode fused kernel: a slot starts no further evaluator-free
                                          simulation once its step call has run this many shader cycles (balances
                                          the waves of a launch by cost instead of by count).  Wave-autonomous
                                          fused kernel: length of one step in shader cycles.  Which launch runs
                                          a simulation never changes results.  0 = off / 80,000 */
    int32_t reserved[4];               /* must be 0 */
} c4_config;

typedef struct c4_engine c4_engine;

/* Aggregate counters since the last c4_reset (units defined in SURVEY.md section 8d). */
typedef struct {
    int64_t simulations;        /* iterations of mcts.py:107-120 completed */
    int64_t expansions;         /* expand_node calls that materialise children (mcts.py:115) */
    int64_t children_created;
    int64_t terminal_sims;      /* simulations that ended on a terminal leaf (no evaluator call) */
    int64_t leaf_evals;         /* leaves handed to the evaluator (incl. one root per move) */
    int64_t depth_sum;          /* sum over simulations of leaf depth */
    int64_t moves;              /* moves chosen (mcts.py:78-88) */
    int64_t games_started;
    int64_t games_finished;
    int64_t launches;           /* rollout-step kernel launches */
    int64_t active_slots;       /* slots not parked right now */
    int64_t capped_slots;       /* slot-launches that hit max_inner_iters */
    int64_t eval_cache_hits;    /* leaves answered by the evaluation cache (subset of leaf_evals) */
    int64_t eval_cache_probes;
    int64_t bad_evals;          /* evaluator answers that were not finite / out of range and were replaced
                                   (value 0.5, prior 0; an all-zero prior becomes uniform) -- the reference
                                   asserts instead (model.py:258-263); must be 0 in a healthy run */
    int64_t dropped_games;      /* finished games that found the record ring full (nobody drained or exported
                                   for record_capacity_games games) and were not recorded; 0 in a healthy run */
    int64_t speculative_evals;  /* c4_selfplay_steps: network passes on positions evaluated ahead of the search (the
                                   best-prior child of a fresh expansion) and put into the evaluation cache; not part
                                   of leaf_evals -- network passes in total = leaf_evals - eval_cache_hits + this */
} c4_stats;

/* Root read-out of one slot (tree.py:66-117; what MCTS.make_move returns, mcts.py:88). */
typedef struct {
    int32_t  state;             /* 0 searching, 1 parked (no game), 2 move chosen */
    int32_t  move;              /* chosen column or -1 */
    double   value;             /* child.data.absolute_value of the chosen child (NaN if None) */
    uint32_t root_visits;
    double   root_value_sum;
    uint32_t child_visits[7];
    double   child_value_sum[7];
    int32_t  child_status[7];   /* -2 absent, -1 non-terminal, else C4_RESULT_* */
    double   root_prior[7];     /* normalised (+ noise) prior used at the root */
    double   values_policy[7];  /* tree.py:104-109 */
    uint64_t color0, color1;    /* root position searched */
    int64_t  expansions;
    int64_t  simulations;
} c4_root_result;

/* One finished self-play game (training_game.py:42-67 GameData). */
typedef struct {
    int64_t  game_id;
    int32_t  length;
    int32_t  result;            /* C4_RESULT_* */
    uint64_t color0[42], color1[42];   /* board BEFORE each move (training_game.py:12) */
    int32_t  move[42];
    double   value[42];         /* child.data.absolute_value (training_game.py:13) */
    double   policy[42][7];     /* tree.get_values_policy() (training_game.py:14) */
} c4_game_record;

/* -- lifecycle ------------------------------------------------------------------------------- */
/* Replaces constructing MCTS(name, MCTSConfig, evaluator) (mcts.py:70-76) + the game_pool /
 * InferenceServer scaffolding (game_pool.py:15-42, inference_server.py:15-63). */
int c4_engine_create(const c4_config *cfg, int device, c4_engine **out);
int c4_engine_destroy(c4_engine *e);
const char *c4_last_error(const c4_engine *e /* may be NULL */);
/* Order the engine's kernels with the caller's stream (torch.cuda.current_stream().cuda_stream). */
int c4_set_stream(c4_engine *e, void *hip_stream);
/* Forget all cached evaluations (call when the evaluator's weights change; evaluators.py: a new
 * Evaluator / position_table per generation, game_pool.py:21-27). */
int c4_clear_eval_cache(c4_engine *e);

/* Start positions for the first n_active slots (Tree(board), tree.py:62-64); NULL => empty boards
 * (Board(), board.py:36-41).  Remaining slots park.  Clears counters, records and game ids. */
int c4_reset(c4_engine *e, const uint64_t *color0, const uint64_t *color1, int32_t n_active);

/* C4_RNG_TAPE: gamma_noise[game][ply][7] raw Gamma(alpha,1) draws (mcts.py:175-177) and
 * uniforms[game][ply] (the one uniform np.random.choice consumes, tree.py:80). `n_games` rows. */
int c4_set_tapes(c4_engine *e, const double *gamma_noise, const double *uniforms, int32_t n_games);

/* -- the hot path --------------------------------------------------------------------------- */
/* One rollout step for every live slot, asynchronous on the engine's stream:
 *   1. apply the evaluator's answer for the slot's pending leaf (mcts.py:124-135 evaluate_node:
 *      mask + normalise prior, store; root: add_exploration_noise mcts.py:171-181), create the
 *      children (tree.py:119-132) and back the value up the path (mcts.py:164-168);
 *   2. run simulations (mcts.py:107-116: PUCT descent ucb_score/select_child mcts.py:138-161)
 *      until the slot needs the evaluator again; terminal leaves are scored in place
 *      (mcts.py:125-128); after `simulations` sims choose the move (mcts.py:81-86), record it
 *      (training_game.py:12-15) and re-root;
 *   3. write the slot's leaf as NN input planes (board.py:147-154 to_array) into planes_dev
 *      [n_slots][3][6][7] and its bitboards into the engine's leaf buffers.
 * values_dev / priors_dev hold the answers for the leaves emitted by the PREVIOUS step, in slot
 * order (dtype per eval_mode; ignored by C4_EVAL_CENTRE; may be NULL on the first step).
 * planes_dev may be NULL when the caller evaluates from the bitboards instead. */
int c4_step(c4_engine *e, const void *values_dev, const void *priors_dev, void *planes_dev);

/* The same step for slots [slot_lo, slot_lo+slot_count) only, on `hip_stream` (NULL = the engine's
 * stream).  Lets the host pipeline two halves of the batch on two streams: the tree kernel of one half
 * runs while the network evaluates the other half's leaves.  Buffers stay indexed by absolute slot. */
int c4_step_range(c4_engine *e, const void *values_dev, const void *priors_dev, void *planes_dev,
                  int32_t slot_lo, int32_t slot_count, void *hip_stream);

/* C4_EVAL_CENTRE convenience: launch until every slot has parked (stop_after_move) or
 * max_launches is reached.  Synchronous. */
int c4_run_centre(c4_engine *e, int32_t max_launches);

/* Leaves emitted by the last step: device views (engine-owned, valid until destroy) ... */
int c4_leaf_buffers(c4_engine *e, const uint64_t **color0_dev, const uint64_t **color1_dev,
                    const int32_t **has_leaf_dev);
/* ... and a synchronous host copy (for Python evaluators, evaluators.py:18-25). */
int c4_read_leaves(c4_engine *e, uint64_t *color0, uint64_t *color1, int32_t *has_leaf);

/* -- read-out ------------------------------------------------------------------------------- */
int c4_get_stats(c4_engine *e, c4_stats *out);                 /* synchronous */
int c4_read_roots(c4_engine *e, c4_root_result *out /* [n_slots] */);   /* synchronous */
/* Up to `cap` finished games not yet consumed (oldest first; the batch is returned sorted by game id;
 * training.py:131 games.extend).  One or two device-to-host copies for the whole batch. */
int c4_drain_games(c4_engine *e, c4_game_record *out, int32_t cap, int32_t *n_out);
/* Finished games waiting in the ring / finished games lost to a full ring.  Synchronous. */
int c4_finished_games(c4_engine *e, int64_t *n_ready, int64_t *n_dropped);

/* Device-side export: pack up to max_games finished games (oldest first, whole games only, at most
 * cap_positions positions) into caller-owned DEVICE buffers and consume them -- no host round trip, so
 * it can be queued behind c4_selfplay_steps on the same stream.  Layout = the compact record of
 * SURVEY.md 8e (what training_game.py:42-67 GameData holds, ~50 B/position); any pointer may be NULL.
 * counts_dev (device int64[2], may be NULL) receives {games, positions} exported.
 * Stands in for pool.imap_unordered(...) / games.extend (training.py:122-131) + the per-position
 * Python of TrainingDataStorage.save (data.py:52-64). */
typedef struct {
    int64_t *boards_dev;      /* [cap_positions][2]  color0, color1 of the board BEFORE the move */
    uint8_t *moves_dev;       /* [cap_positions] */
    float   *values_dev;      /* [cap_positions]     child.data.absolute_value (NaN if None) */
    float   *policy_dev;      /* [cap_positions][7]  tree.get_values_policy() */
    float   *targets_dev;     /* [cap_positions]     game result value (create_training_values) */
    int32_t *game_index_dev;  /* [cap_positions]     index of the position's game inside this export */
    int32_t *lengths_dev;     /* [max_games] */
    int8_t  *results_dev;     /* [max_games]         C4_RESULT_* */
    int64_t *ids_dev;         /* [max_games] */
} c4_export_buffers;
int c4_export_games_dev(c4_engine *e, const c4_export_buffers *bufs, int32_t max_games, int64_t cap_positions,
                        int64_t *counts_dev, void *hip_stream);
/* native_to_pytorch(boards, values, priors, add_fliplr) (data.py:78-105) on device: boards
 * float32 [m][3][6][7] (board.py:147-154), values float32 [m], priors float32 [m][7], m = n or 2n with
 * the left-right mirrored copies after the originals (board.py:115-145, priors reversed). */
int c4_training_tensors_dev(int device, void *hip_stream, const int64_t *boards_dev, const float *targets_dev,
                            const float *policy_dev, int64_t n, int32_t add_fliplr, float *out_boards_dev,
                            float *out_values_dev, float *out_priors_dev);

/* Read-out of the evaluation cache (the memo table of evaluators.py:9-25): what it answers for the given
 * positions.  found[i] = 0 when the position is absent (never evaluated, or evicted).  Synchronous.
 * Lets a test replay device games on the oracle with exactly the evaluations the device used. */
int c4_eval_cache_lookup(c4_engine *e, const uint64_t *color0, const uint64_t *color1, int32_t n, float *value,
                         float *prior /* [n][7] */, int32_t *found);

/* Read-outs of the PRODUCTION random streams (C4_RNG_PHILOX), computed by the very device functions the
 * engine's kernels call, for statistical tests: the root noise of (seed, game id, ply) -- gamma_raw[n][7] =
 * Gamma(alpha,1) draws (mcts.py:175-177), dirichlet[n][7] = zeroed on illegal columns and normalised
 * (mcts.py:178) -- and the sampled opening move (tree.py:75-82): child index chosen among n_children
 * values (from the root mover's side, child k in lane k) with the engine's uniform of (seed, game id, ply),
 * or with uniforms[i] when given. */
int c4_debug_root_noise(int device, uint64_t seed, double alpha, const int64_t *game_id, const int32_t *ply,
                        const int32_t *legal_mask, int32_t n, double *gamma_raw, double *dirichlet);
int c4_debug_sample_move(int device, uint64_t seed, const int64_t *game_id, const int32_t *ply,
                         const double *child_values /* [n][7] */, const int32_t *n_children,
                         const double *uniforms /* may be NULL */, int32_t n, double *uniform_out, int32_t *choice_out);

/* The level loop and the backups divide with an instruction sequence written out for normal-range operands (the compiler's
 * IEEE float64 division minus its scaling / fix-up instructions; c4_engine.hip: div_normal).  This runs that sequence and the
 * plain division on the device for every (parent visits N < max_parent_visits, child visits n < max_child_visits) pair of the
 * PUCT score's sqrt(N) / (n + 1) (mcts.py:153-154) and for n_random value-sum / visit-count quotients of the backups
 * (mcts.py:164-168), and returns the number of quotients whose bits differ (must be 0). */
int c4_debug_div_mismatches(int device, int32_t max_parent_visits, int32_t max_child_visits, int64_t n_random, int64_t *mismatches);

/* -- pure board functions, executed by the device code (bit-exact parity tests) ------------- */
/* board.py:160-170 make_move + result */
int c4_board_make_move(int device, const uint64_t *color0, const uint64_t *color1, const int32_t *col,
                       int32_t n, uint64_t *out0, uint64_t *out1, int32_t *result);
/* board.py:173-184 _check_terminal_position */
int c4_board_wins(int device, const uint64_t *stones, int32_t n, int32_t *out);
/* board.py:88-92 valid_moves as a 7-bit mask (0 when the position is decided, board.py:56-62) */
int c4_board_valid_mask(int device, const uint64_t *color0, const uint64_t *color1, int32_t n, int32_t *out);
/* board.py:147-154 to_array, float32 [n][3][6][7] */
int c4_board_planes(int device, const uint64_t *color0, const uint64_t *color1, int32_t n, float *out);
/* board.py:115-145 create_fliplr / flip_color */
int c4_board_fliplr(int device, const uint64_t *color0, const uint64_t *color1, int32_t n,
                    uint64_t *out0, uint64_t *out1);
/* evaluators.py:28-33 evaluate_centre (float64) */
int c4_board_centre_value(int device, const uint64_t *color0, const uint64_t *color1, int32_t n, double *out);

/* -- fused leaf-batch network (c4_net.hip) -------------------------------------------------- */
/* Eval-mode forward of the reference's Net (oinkoink/neural/pytorch/model.py:120-134) as ONE
 * gfx950 MFMA kernel reading the leaves' bitboards; stands in for ModelWrapper._call_list
 * (model.py:269-282) + the InferenceServer round trip (inference_server.py:50-63).
 * All weights are host float32 with BatchNorm already folded (eval mode):
 *   stem_w [F][3][3][3], stem_b [F]            body.0 (model.py:20-31)
 *   conv_w [2R][F][F][3][3], conv_b [2R][F]    residual blocks, conv1 then conv2 (model.py:36-55)
 *   head_w [3][F], head_b [3]                  value 1x1 conv, then the 2 policy 1x1 channels
 *   vfc_w [42][42], vfc_b [42]                 the activation-free Linear stack collapsed (model.py:69-70,83)
 *   vout_w [42], vout_b                        value_head.fc1 (model.py:72,85)
 *   pfc_w [7][84], pfc_b [7]                   policy_head.fc1 (model.py:104,113)
 *   w1, w2                                     value_head.w1/w2 (model.py:74-75,88) */
/* arithmetic of the fused forwards */
#define C4_NET_F16 0    /* fp16 storage, fp32 accumulation: one MFMA per k-step */
#define C4_NET_F32X3 1  /* reference precision: fp32 operands split into fp16 hi + scaled lo, three MFMAs per k-step */
typedef struct {
    int32_t channels, filters, n_residuals;
    int32_t precision;   /* C4_NET_F16 or C4_NET_F32X3 */
    const float *stem_w, *stem_b, *conv_w, *conv_b, *head_w, *head_b;
    const float *vfc_w, *vfc_b, *vout_w, *pfc_w, *pfc_b;
    float vout_b, w1, w2, reserved2;
} c4_net_desc;
typedef struct c4_net c4_net;
int c4_net_create(int device, const c4_net_desc *desc, c4_net **out);
int c4_net_destroy(c4_net *net);
/* values_dev float32[n] in [0,1] (o's side), priors_dev float32[n][7]; bitboards as emitted by
 * c4_step (c4_leaf_buffers).  Asynchronous on hip_stream. */
int c4_net_forward(c4_net *net, void *hip_stream, const uint64_t *color0_dev, const uint64_t *color1_dev,
                   int32_t n, float *values_dev, float *priors_dev);
/* c4_net_forward evaluated by the wave-private forward the fused self-play kernel uses (one wave = one
 * position, no workgroup barrier).  Same arguments, bit-identical answers. */
int c4_net_forward_wave(c4_net *net, void *hip_stream, const uint64_t *color0_dev, const uint64_t *color1_dev,
                        int32_t n, float *values_dev, float *priors_dev);
const char *c4_net_last_error(void);

/* Fused persistent self-play: the rollout step and the leaf evaluation of n_steps steps in ONE launch,
 * no kernel boundary and no global barrier; slot state, leaves and answers stay in LDS for the whole
 * launch.  Default (split kernel): on every CU four tree waves own the workgroup's 16 (32) slots and walk their
 * trees; a slot that needs the evaluator posts its leaf in LDS and idles while its wave's other slots walk on;
 * four network waves claim posted leaves, run the forward and write the answer back (and, when no leaf waits,
 * evaluate the best-prior child of the position just answered into the evaluation cache: c4_stats.speculative_evals)
 * -- no workgroup barrier;
 * a "step" is a time quantum of c4_config.time_budget_cycles shader cycles (80,000 if 0) and the launch runs
 * for n_steps quanta, however many simulations the trees needed per network answer.  With the environment
 * variable C4_FUSED_MODE=wave: every wave owns 2 (4) slots and alternates their tree walk with the network on
 * exactly their leaves; C4_FUSED_MODE=block: n_steps rounds of {tree step of the workgroup's 16/32 slots;
 * barrier; network on
 * the emitted leaves, 16 per pass; barrier}.  Either way the games are exactly those that alternating
 * c4_step / c4_net_forward_wave launches play (which launch runs a simulation never changes a result).
 * values_dev float32 [n_slots], priors_dev float32 [n_slots][7] are the hand-off buffers (must persist
 * between calls).  Needs C4_EVAL_EXTERNAL_F32.  Per-slot rows of the statistics are only exact per
 * workgroup after this call (c4_get_stats sums them; the sums are exact). */
int c4_selfplay_steps(c4_engine *e, c4_net *net, float *values_dev, float *priors_dev, int32_t n_steps,
                      void *hip_stream);
/* diagnostic build aid: per-phase s_memtime stamps of workgroup 0, [8 waves][16]; needs the
 * environment variable C4_NET_STAMPS=1 when the net is created, else C4_ESTATE. */
int c4_net_debug_stamps(c4_net *net, unsigned long long *out);

/* diagnostic build aid: per-phase s_memtime stamps of the last c4_step launch, [256 workgroups][8]
 * (0 start, 1 slot state loaded, 2 apply done, 3 descent done, 4 before emit, 5 end, 6 depth), or of
 * the last c4_selfplay_steps launch, [128 workgroups][16] (0-7 tree-phase cycles of each wave, 8 tree
 * phase incl. barrier, 9 network phase, 10 steps);
 * needs C4_TREE_STAMPS=1 in the environment when the engine is created, else C4_ESTATE. */
int c4_debug_stamps(c4_engine *e, unsigned long long *out);

/* diagnostic build aid: s_memtime stamps of the last network pass of every wave of workgroups 0..15 inside the last
 * c4_selfplay_steps launch, [16][8][16] (phase order as c4_net_debug_stamps); needs C4_TREE_STAMPS=1, else C4_ESTATE. */
int c4_debug_fused_net_stamps(c4_engine *e, unsigned long long *out);

/* diagnostic build aid (library built with -DC4_SPLIT_PHASES=1; zeros otherwise): the life of the evaluator requests of the last
 * c4_selfplay_steps launch, workgroups 0..127, [128][4] = {posted -> claimed by a network wave, claimed -> answered,
 * answered -> picked up by its tree wave} summed in units of 64 shader cycles, and the number of requests; needs
 * C4_TREE_STAMPS=1, else C4_ESTATE. */
int c4_debug_latency_stamps(c4_engine *e, unsigned long long *out);

/* ---- train step (the reference: ModelWrapper.train, neural/pytorch/model.py:200-240) ----------
 * The convolutions, linear layers, losses and SGD of the train step are stock PyTorch-ROCm; batch normalisation in
 * training mode -- half of a stock step's GPU time at this net's shape -- is the library's own, fused with the residual add
 * and the LeakyReLU that follow it in the reference's net (model.py:20-31, 36-55, 60-117):
 *     y = act(bn(x) + residual),   bn(x) = (x - mean_c) * invstd_c * weight_c + bias_c,   act = LeakyReLU(slope) (1 = none)
 * float32, contiguous [rows][channels][hw] (NCHW, hw = 42).  Batch statistics (mean, biased variance; two passes) come from
 * the first valid_rows rows (the trainer pads the ragged last batch of an epoch, connect4_amd/net.py:_BatchNorm2d);
 * running_mean / running_var (may both be NULL) and *num_batches_tracked (may be NULL) are updated as torch.nn.BatchNorm2d
 * updates them; save_mean / save_invstd [channels] are kept for the backward.  workspace: c4_bn_workspace_floats(rows,
 * channels) floats.  Launches on hip_stream (capturable in a HIP graph); every reduction has a fixed order. */
long long c4_bn_workspace_floats(int rows, int channels);
int c4_bn_train_forward(const float *x_dev, const float *residual_dev, const float *weight_dev, const float *bias_dev,
                        float *running_mean_dev, float *running_var_dev, long long *num_batches_tracked_dev, float *y_dev,
                        float *save_mean_dev, float *save_invstd_dev, float *workspace_dev, int rows, int valid_rows, int channels,
                        int hw, float momentum, float eps, float slope, void *hip_stream);
/* dz = dy * (y > 0 ? 1 : slope); dresidual (may be NULL) = dz; dbias = sum dz; dweight = sum dz * xhat;
 * dx = weight * invstd * (dz - dbias / M - xhat * dweight / M) over the valid rows (M = valid_rows * hw). */
int c4_bn_train_backward(const float *x_dev, const float *y_dev, const float *dy_dev, const float *weight_dev, const float *save_mean_dev,
                         const float *save_invstd_dev, float *dx_dev, float *dresidual_dev, float *dweight_dev, float *dbias_dev,
                         float *workspace_dev, int rows, int valid_rows, int channels, int hw, float slope, void *hip_stream);

/* Weight gradient of the tower's convolutions (model.py:36-55: 3x3, 32 -> 32 filters, padding 1, no bias; the autograd backward of
 * F.conv2d inside ModelWrapper.train, model.py:226): dweight[co][ci][ky][kx] = sum_n sum_yx dy[n][co][y][x] * x[n][ci][y+ky-1][x+kx-1],
 * float32, contiguous NCHW [rows][32][6][7] (other shapes: C4_EINVAL -- the caller keeps PyTorch's own backward for them).  One pass
 * over x and dy on the f32-input MFMA (exact float32 products), per-workgroup partials summed in a fixed order: reproducible.
 * workspace: c4_conv3x3_wrw_workspace_floats() floats.  Launches on hip_stream (capturable). */
long long c4_conv3x3_wrw_workspace_floats(void);
int c4_conv3x3_wrw(const float *x_dev, const float *dy_dev, float *dweight_dev, float *workspace_dev, int rows, int channels, int height, int width,
                   void *hip_stream);

int c4_abi_version(void);

#ifdef __cplusplus
}
#endif
#endif
