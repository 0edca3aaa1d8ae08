// c4_engine.hip -- MI355X (gfx950) self-play / MCTS engine behind the C ABI of include/c4_engine.h.
//
// Design (DESIGN.md has the long form):
//   * thousands of games ("slots") are searched side by side; each slot owns a private node pool in HBM
//     of 32-byte records {W f64, Q f64, P f64, N u32, info u32};
//   * the children of a node are created together as one 8-aligned 256-byte sibling block in ascending
//     column order, so the 7 lanes that score them read it with two 16-byte loads each (two cache lines);
//   * one slot = one 8-lane group of a wave64: lane k scores child k, the PUCT argmax is a 3-step DPP
//     butterfly over the group (quad_perm x2 + row_half_mirror) that carries the winner's record along;
//   * boards are never stored per node: the descent replays moves on a register bitboard;
//   * children are materialised when their parent is EVALUATED (so the prior can live with the
//     child); the reference's "expand on second visit" (mcts.py:114-115) is the moment a node with
//     N==1 is first descended through -- that is what the expansion counter counts;
//   * score arithmetic reproduces the reference bit-for-bit: float64 everywhere, or NumPy>=2's
//     float32 path when the prior is a float32 net output (see ucb_score below); log() comes from a
//     host-built table so device libm never enters the comparison.  Build with -ffp-contract=off;
//   * three drivers share tree_step(): c4_step_kernel (one rollout step per launch, evaluator outside),
//     c4_selfplay_kernel (workgroup-synchronous tree / network phases in one persistent kernel) and
//     c4_selfplay_wave_kernel (the default: every wave alternates the tree walk of its own slots with
//     the network on their leaves, no barrier; slot state lives in LDS for the whole launch).
//
// No CUDA-compat headers, no dual code paths, no CPU fallback: every entry point needs a HIP device.
#include <hip/hip_runtime.h>
#include <hip/hip_fp16.h>
#include <hip/hip_bfloat16.h>

#include <algorithm>
#include <cmath>
#include <cstdarg>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <limits>
#include <vector>

#include "../../include/c4_engine.h"
#include "c4_board.h"
#include "c4_net_dev.h"

#ifndef C4_DEADLINE_EVERY
#define C4_DEADLINE_EVERY 1   // a tree call looks at the clock every this many simulations (power of two; 4 measured no faster)
#endif
#ifndef C4_TERMQ
#define C4_TERMQ 1            // q of a terminal node holds its exact result from creation on
#endif
#ifndef C4_FUSED_NET_STAMPS
#define C4_FUSED_NET_STAMPS 0 // diagnostic build only (-DC4_FUSED_NET_STAMPS=1, tools/wave_stamps.py): stamps of the network pass
                              // inside the fused kernel; the extra pointer costs the production kernel 3 %
#endif
#ifndef C4_ASM_PICK
#define C4_ASM_PICK 1         // butterfly step written out in assembly (compare, select and DPP moves in one block)
#endif
#ifndef C4_SPECULATE
#define C4_SPECULATE 1         // split kernel: a network wave with nothing else to do evaluates the best-prior child of the position it just answered (tuning aid: 0 = off)
#endif
#ifndef C4_SPLIT_PHASES
#define C4_SPLIT_PHASES 0       // diagnostic build only: per-slot cycles per phase of the split kernel's tree walk (tools/split_stamps.py)
#endif
#ifndef C4_DIV_NORMAL
#define C4_DIV_NORMAL 1
#endif
#ifndef C4_SCORE_ALL
#define C4_SCORE_ALL 1          // level loop: all eight lanes score, a select instead of a branch around the division (+1.6 %)
#endif
#ifndef C4_PATH_ALL_LANES
#define C4_PATH_ALL_LANES 1      // the descent's path entries are stored by all eight lanes of the group (same words, same address): no
                                // exec-mask region per level (+0.5 %; the same for the parent's record in the apply: nothing)
#endif
#ifndef C4_PEEL_BACKUP
#define C4_PEEL_BACKUP 1
#endif
#ifndef C4_NO_SUSPEND
#define C4_NO_SUSPEND 1
#endif
#ifndef C4_NET_SANITISES
#define C4_NET_SANITISES 1
#endif
#ifndef C4_WAIT_CLOCK_LAZY
#define C4_WAIT_CLOCK_LAZY 1
#endif
#ifndef C4_SPLIT_PAIRS
#define C4_SPLIT_PAIRS 1      // split kernel: a network wave takes two waiting requests into one pass (tuning aid: 0 = one position per pass)
#endif
#ifndef C4_SORT_SLOTS
#define C4_SORT_SLOTS 1       // split kernel: at launch start the workgroup's slots are sorted by ply and neighbours in that order share a tree
                              // wave (games at a similar stage spend their iterations in similar phases: +1.6 % with the fp16 net, +0.3 % with the
                              // reference-precision net; sorting by simulations done or by the measured leaf depth: nothing); 0 = slot p -> wave p % TW
#endif
#ifndef C4_FILLERS
#define C4_FILLERS (C4_ORIENTED_Q && C4_SCORE_ALL)   // a sibling block's unused records are written too, with q = -inf: they score -inf by arithmetic, the level loop needs no "is this lane a child" logic
#endif
#ifndef C4_VGPR_BASES
#define C4_VGPR_BASES 1
#endif
#ifndef C4_ORIENTED_Q
#define C4_ORIENTED_Q 1      // a record's q is stored from the point of view of the player who chooses AT ITS PARENT: the level loop reads the value it scores
#endif
#ifndef C4_COOP_LEGAL
#define C4_COOP_LEGAL 1      // the legal-move mask of a group's position by one lane per column + a ballot; the prior sum's seven fetches issued together
#endif
#ifndef C4_PICK_BPERM
#define C4_PICK_BPERM 1      // the level loop's argmax: butterfly on the score alone + one fetch of the winner's record through the LDS crossbar
#endif
#ifndef C4_EARLY_REQUEST
#define C4_EARLY_REQUEST 1   // software-pipelined level loop (tuning aid: -DC4_EARLY_REQUEST=0 restores the plain loop)
#endif

namespace {

using namespace c4;

constexpr int GROUP = 8;            // lanes per slot
constexpr int SLOTS_PER_BLOCK = 8;  // one wave64 per block (2 or 4 slots per wave measured the same)
constexpr int BLOCK = GROUP * SLOTS_PER_BLOCK;
constexpr int MAX_DEPTH = 44;       // root + 42 plies + 1

// slot states
constexpr int SLOT_ACTIVE = 0;
constexpr int SLOT_PARKED = 1;
constexpr int SLOT_MOVE_DONE = 2;

// info word: [0:18] child block (child_base/8) | [19:21] nchild | [22:24] status | [25:30] bit index of
// the stone this move drops (col*7 + row, so the descent replays it with one shift/xor) | [31] prior is f64
__host__ __device__ __forceinline__ uint32_t pack_info(uint32_t base, uint32_t nchild, uint32_t status,
                                                       uint32_t bit, uint32_t pf64)
{
    return (base >> 3) | (nchild << 19) | (status << 22) | (bit << 25) | (pf64 << 31);
}
__host__ __device__ __forceinline__ uint32_t info_base(uint32_t i) { return (i & 0x7ffffu) << 3; }
__host__ __device__ __forceinline__ uint32_t info_nchild(uint32_t i) { return (i >> 19) & 7u; }
__host__ __device__ __forceinline__ uint32_t info_status(uint32_t i) { return (i >> 22) & 7u; }
__host__ __device__ __forceinline__ uint32_t info_bit(uint32_t i) { return (i >> 25) & 63u; }
__host__ __device__ __forceinline__ uint32_t info_move(uint32_t i) { return info_bit(i) / 7u; }
__host__ __device__ __forceinline__ uint32_t info_pf64(uint32_t i) { return i >> 31; }

struct __attribute__((aligned(16))) PathEntry {
    uint32_t node;
    uint32_t n;   // visit count seen during the descent
    double w;     // value sum seen during the descent
};

constexpr int BLOCK_BYTES = 256;
// One node = one 32-byte record; the (up to 7) children of a node are 8-aligned consecutive records,
// i.e. one 256-byte sibling block = two 128-byte lines.  Lane k of a slot's 8-lane group reads child
// k's whole record with two 16-byte loads, so the group's loads coalesce into exactly those two
// lines.  (Measured on MI355X: per-field arrays cost 5 load instructions per level, each touching the
// same lines again)
struct __attribute__((aligned(32))) Rec {
    double w;        // value sum          (SearchEvaluation.value_sum, mcts.py:49)
    double q;        // w / n, refreshed by every backup (what tree.py:35 divides out on each read)
    double p;        // prior of this node as seen from its parent
    uint32_t n;      // visit count        (mcts.py:50)
    uint32_t info;   // child_base | nchild | status | move | prior-is-f64
};
static_assert(sizeof(Rec) == 32, "node record must be 32 bytes");
struct Pool {   // view of one slot's node pool; node id = block*8 + k
    uint8_t *base;
    __device__ __forceinline__ Rec *rec(uint32_t i) const { return reinterpret_cast<Rec *>(base) + i; }
    __device__ __forceinline__ uint32_t &n(uint32_t i) const { return rec(i)->n; }
    __device__ __forceinline__ uint32_t &info(uint32_t i) const { return rec(i)->info; }
    __device__ __forceinline__ double &w(uint32_t i) const { return rec(i)->w; }
    __device__ __forceinline__ double &q(uint32_t i) const { return rec(i)->q; }
    __device__ __forceinline__ double &p(uint32_t i) const { return rec(i)->p; }
};

// Evaluation cache entry: what the evaluator answered for one position (evaluators.py:18-25 keeps the
// same thing in a dict keyed by (color0, color1)).  `check` ties key and payload together so that an
// entry torn by two concurrent writers is rejected instead of believed.
struct CacheEntry {
    uint64_t key;      // color0 + (color0 | color1): unique for gravity-consistent positions (0xFF..F = empty)
    uint64_t check;
    float value;
    float prior[7];
};
static_assert(sizeof(CacheEntry) == 48, "cache entry must be 48 bytes");

struct SlotStats {  // per-slot counters (summed on the host; no atomics => deterministic)
    uint64_t sims, expansions, children, terminal_sims, leaf_evals, depth_sum, moves, games_started,
        games_finished, capped, cache_hits, cache_probes, bad_evals, spec_evals;
};
constexpr int N_STATS = sizeof(SlotStats) / sizeof(uint64_t);

// One slot's state as the fused self-play kernel keeps it in LDS between its steps (loaded from the
// Dev arrays when the launch starts, written back when it ends): a step then begins and ends without
// a global-memory round trip, and the leaf hand-over to the network phase never leaves the CU.
struct SlotMem {
    uint64_t root0, root1, leaf0, leaf1;
    long long gid;
    double root_w;             // W of the root node (its N is sims + 1, its info follows from the root board)
    uint32_t sims, nalloc;
    int32_t pend;
    uint32_t pdepth, pinfo;
    uint32_t ply;
    uint32_t flags;            // state | has_leaf << 8 | need_root << 16
    __device__ int state() const { return (int)(flags & 0xffu); }
    __device__ bool has_leaf() const { return ((flags >> 8) & 0xffu) != 0; }
    __device__ int need_root() const { return (int)(flags >> 16); }
    __device__ static uint32_t pack(int state, int has_leaf, int need_root) { return (uint32_t)state | ((uint32_t)has_leaf << 8) | ((uint32_t)need_root << 16); }
};
static_assert(sizeof(SlotMem) == 80, "32 of these must fit the fused kernel's LDS budget");

// Rarely touched pointers (move choice, game end, suspended descents, diagnostics) live in device
// memory behind one pointer: a kernel argument block with ~60 pointers does not fit the scalar
// register file and the spills land in the descent loop.
struct Cold {
    uint32_t *cont_cur, *cont_n;   // a descent suspended by the level budget: current node, its N ...
    double *cont_w;                // ... and W (its info lives in pending_info, the board in leaf_c0/1)
    // per-slot result of the last chosen move
    int32_t *res_move;
    double *res_value;
    double *res_policy;      // [G][7]
    // RNG tapes (C4_RNG_TAPE)
    const double *noise_tape;  // [tape_games][42][7]
    const double *u_tape;      // [tape_games][42]
    int tape_games;
    // moves of the game in flight, per slot (training_game.py:12-15): [G][42] rows; copied to the ring
    // of finished games when the game ends, so a slow game can never share a row with a newer one
    uint64_t *stg_c0, *stg_c1;
    int32_t *stg_move;
    double *stg_value;
    double *stg_policy;          // [G][42][7]
    // finished games: ring of c4_game_record rows in completion order.  head = games recorded so far
    // (device, CAS), tail = games consumed (drain / export); a game that finds the ring full is
    // counted in `dropped` instead of overwriting an unread row.
    c4_game_record *ring;        // [rec_cap]
    unsigned long long *ring_head, *ring_tail, *dropped;
    unsigned long long *next_game;
    unsigned long long *stamps;   // diagnostic (C4_TREE_STAMPS=1): [block][8] s_memtime values, first 256 blocks
};

struct Dev {
    // node pools: per slot `cap` records of 32 B (struct Rec), children in 8-aligned sibling blocks
    uint8_t *pool;
    // slot state
    uint64_t *root_c0, *root_c1, *leaf_c0, *leaf_c1;
    int32_t *has_leaf;
    int32_t *pending;        // node awaiting its evaluation, -1 none
    uint32_t *pending_depth;
    uint32_t *pending_info;
    uint32_t *sims_done;
    uint32_t *n_alloc;       // next free 8-slot block
    int32_t *state;
    int32_t *need_root;
    uint32_t *ply;
    long long *game_id;
    PathEntry *path;         // [G][MAX_DEPTH]
    uint64_t *stats;         // [G][N_STATS]
    // score tables, index = parent visit count
    const double2 *tabAB;    // .x = log((n + base + 1)/base) + init (mcts.py:150-152), .y = sqrt(n) (mcts.py:156)
    CacheEntry *cache;       // evaluation cache (evaluators.py:18-25 memo table), direct mapped; null = off
    const Cold *cold;
    // config
    int G;
    int slot_lo, slot_hi;    // slots advanced by this launch (c4_step_range; whole engine by default)
    uint32_t cap;
    int S;
    int nsm;
    int use_noise;
    int rng_tape;
    int stop_after_move;
    int max_inner;
    int level_budget;        // descent levels a slot may walk per launch (0 = unlimited)
    int time_budget;         // shader cycles after which a slot starts no new simulation in this call (0 = off)
    int planes_dtype;
    int rec_cap;
    int cache_bits;
    int has_stamps;          // diagnostic build aid enabled (C4_TREE_STAMPS=1)
    long long games_target;
    double alpha, frac;
    uint64_t seed;
};

// ------------------------------------------------------------------------------------------
// device helpers
// ------------------------------------------------------------------------------------------
__device__ __forceinline__ double gshfl(double v, int src) { return __shfl(v, src, GROUP); }
__device__ __forceinline__ float gshfl(float v, int src) { return __shfl(v, src, GROUP); }
__device__ __forceinline__ int gshfl(int v, int src) { return __shfl(v, src, GROUP); }
__device__ __forceinline__ uint32_t gshfl(uint32_t v, int src) { return (uint32_t)__shfl((int)v, src, GROUP); }

// Orders this wave's global + LDS stores before its later loads (lanes of a group communicate
// through memory inside one wave).  Workgroup scope = the same CU's L1, which is what we need.
// LDS-only variant: lane 0 wrote the path stack, other lanes of the same wave read it next.  LDS
// operations of a wave execute in order; this only has to stop the compiler and drain lgkmcnt.
__device__ __forceinline__ void lds_fence() { asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); }
// Words in LDS that waves of a workgroup exchange (request states, answers).  The pointers reach the device functions
// as generic pointers, and a volatile access through a generic pointer is a FLAT instruction with system-scope cache
// bits -- the wave then waits for vmcnt(0) AND lgkmcnt(0), i.e. for every store it has in flight to global memory.
// An explicit LDS address-space access is a ds_read / ds_write that waits for the LDS queue only.
typedef __attribute__((address_space(3))) volatile uint32_t lds_vu32_t;
typedef __attribute__((address_space(3))) volatile float lds_vf32_t;
typedef __attribute__((address_space(3))) volatile uint64_t lds_vu64_t;
__device__ __forceinline__ uint32_t lds_ld(const uint32_t *p) { return *(lds_vu32_t *)p; }
__device__ __forceinline__ void lds_st(uint32_t *p, uint32_t v) { *(lds_vu32_t *)p = v; }
__device__ __forceinline__ float lds_ldf(const float *p) { return *(lds_vf32_t *)p; }
__device__ __forceinline__ void lds_st64(uint64_t *p, uint64_t v) { *(lds_vu64_t *)p = v; }

__device__ __forceinline__ void group_fence()
{
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
}

// Philox4x32-10 (counter-based; Salmon et al. 2011)
__device__ __forceinline__ void philox_round(uint32_t c[4], const uint32_t k[2])
{
    const uint64_t p0 = (uint64_t)0xD2511F53u * c[0];
    const uint64_t p1 = (uint64_t)0xCD9E8D57u * c[2];
    const uint32_t n0 = (uint32_t)(p1 >> 32) ^ c[1] ^ k[0];
    const uint32_t n1 = (uint32_t)p1;
    const uint32_t n2 = (uint32_t)(p0 >> 32) ^ c[3] ^ k[1];
    const uint32_t n3 = (uint32_t)p0;
    c[0] = n0; c[1] = n1; c[2] = n2; c[3] = n3;
}
__device__ inline void philox4x32(uint32_t c[4], uint64_t seed)
{
    uint32_t k[2] = {(uint32_t)seed, (uint32_t)(seed >> 32)};
#pragma unroll
    for (int r = 0; r < 10; ++r) {
        philox_round(c, k);
        k[0] += 0x9E3779B9u;
        k[1] += 0xBB67AE85u;
    }
}
// two uniforms in [0,1) from one Philox block
__device__ inline void rng_uniform2(uint64_t seed, long long gid, uint32_t ply, uint32_t stream, uint32_t idx,
                                    double &u0, double &u1)
{
    uint32_t c[4] = {(uint32_t)gid, (uint32_t)((uint64_t)gid >> 32), ply * 64u + stream, idx};
    philox4x32(c, seed);
    u0 = (double)((((uint64_t)c[0] << 32) | c[1]) >> 11) * (1.0 / 9007199254740992.0);
    u1 = (double)((((uint64_t)c[2] << 32) | c[3]) >> 11) * (1.0 / 9007199254740992.0);
}
// Gamma(alpha,1), Marsaglia-Tsang with the alpha<1 boost; bounded rejection loop.
__device__ inline double rng_gamma(uint64_t seed, long long gid, uint32_t ply, uint32_t stream, double alpha)
{
    double u0, u1;
    rng_uniform2(seed, gid, ply, stream, 0, u0, u1);
    double boost = 1.0;
    if (alpha < 1.0) {
        boost = pow(u0 > 0.0 ? u0 : 1e-300, 1.0 / alpha);
        alpha += 1.0;
    }
    const double dd = alpha - 1.0 / 3.0, cc = 1.0 / sqrt(9.0 * dd);
    for (uint32_t it = 1; it <= 24; ++it) {
        double a, b, uu, unused;
        rng_uniform2(seed, gid, ply, stream, 2 * it, a, b);
        rng_uniform2(seed, gid, ply, stream, 2 * it + 1, uu, unused);
        const double x = sqrt(-2.0 * log(a > 0.0 ? a : 1e-300)) * cos(6.283185307179586 * b);
        double v = 1.0 + cc * x;
        if (v <= 0.0) continue;
        v = v * v * v;
        if (uu < 1.0 - 0.0331 * x * x * x * x) return boost * dd * v;
        if (log(uu > 0.0 ? uu : 1e-300) < 0.5 * x * x + dd * (1.0 - v + log(v))) return boost * dd * v;
    }
    return boost * dd;
}

// tree.py:27-44 + utils.py:33-34: value of a child from `side`'s point of view (branch-free).
// terminal: the exact result, not a running mean -- it is what q holds for a terminal node from its creation on
// (every visit adds that same value, so w / n reproduces it exactly); visited: q = value_sum / visit_count
// (refreshed by every backup); unknown: 0.0 for either side ("assume lost")
__device__ __forceinline__ double child_value_for(uint32_t status, uint32_t n, double q, int side)
{
    const bool term = status >= ST_XWIN;
#if C4_TERMQ
    const double v = q;
#else
    const double v = term ? 0.5 * (double)(status - ST_XWIN) : q;
#endif
    const double sv = side == 0 ? v : 1.0 - v;
    return (term || n > 0) ? sv : 0.0;
}

// C4_ORIENTED_Q: tree.py:27-44's value of a node "for the side to move at its parent" is what a record's q field holds -- the flip
// 1 - v for an x-to-move parent is done once by whoever writes q (expansion, backup: same operation on the same float64, so the
// same bits), an unvisited child's q is 0 and a terminal child's its exact result, flipped likewise: the level loop's
// child_value_for (select by side, select by visited / terminal: eight instructions per level) becomes a plain read.
__device__ __forceinline__ double orient_q(double v, int parent_side)
{
#if C4_ORIENTED_Q
    return (parent_side & 1) ? 1.0 - v : v;
#else
    (void)parent_side;
    return v;
#endif
}

// mcts.py:147-161 ucb_score.  A = log(..)+pb_c_init and B = sqrt(Np) come from the host table.
// pf64: the parent's prior is float64 (heuristic / table evaluator, or the root after Dirichlet
// noise, mcts.py:180).  Otherwise it is a float32 net output and NumPy>=2 keeps
// `pb_c * prior[c]` and `prior_score + value_score` in float32 (weak Python scalars, NEP 50).
// Both forms are computed and one is selected: no divergent branch in the descent loop.
// a / b for operands in the normal range, written out: the instruction sequence the compiler emits for an IEEE float64
// division (v_rcp_f64, two Newton steps, one residual correction) without its v_div_scale / v_div_fmas / v_div_fixup
// instructions, which change something only for denormal, infinite, NaN or huge-ratio operands (a zero numerator
// comes out as +0 either way).  For 0 <= a < 2^32 and an integer 1 <= b < 2^32 the quotient is bit-identical to a / b
// (#if C4_DIV_NORMAL == 0 restores the plain division for A/B runs: same games); three dependent instructions less on
// the level loop's critical chain.
__device__ __forceinline__ double div_normal(double a, double b)
{
#if C4_DIV_NORMAL
    double r = __builtin_amdgcn_rcp(b);
    double e = __builtin_fma(-b, r, 1.0);
    r = __builtin_fma(r, e, r);
    e = __builtin_fma(-b, r, 1.0);
    r = __builtin_fma(r, e, r);
    const double q = a * r;
    const double rem = __builtin_fma(-b, q, a);
    return __builtin_fma(rem, r, q);
#else
    return a / b;
#endif
}

__device__ __forceinline__ double ucb_score(double A, double B, uint32_t nc, double p, double V, uint32_t pf64)
{
    const double pbc = A * div_normal(B, (double)(nc + 1));
    const double prior_score64 = pbc * p;
    const double s64 = prior_score64 + V;
    const float prior_score32 = (float)pbc * (float)p;
    const float s32 = prior_score32 + (float)V;
    return pf64 ? s64 : (double)s32;
}

// board.py:88-92 for a position all eight lanes of a group hold: lane c looks at column c (count based, as c4_board.h's legal_mask),
// one ballot makes the mask -- 7 instructions where every lane computing all seven columns takes 40.  Call with the group's
// lanes all active.
__device__ __forceinline__ int legal_mask_group(uint64_t occ)
{
#if C4_COOP_LEGAL
    const int lane = (int)(threadIdx.x & (GROUP - 1));
    const bool legal = lane < WIDTH && col_count(occ, lane < WIDTH ? lane : 0) < HEIGHT;
    const unsigned long long b = __builtin_amdgcn_ballot_w64(legal);
    return (int)((b >> (threadIdx.x & 63u & ~(uint32_t)(GROUP - 1))) & 0x7fu);
#else
    return legal_mask(occ);
#endif
}

// argmax over the group on (score, k); ties -> larger k == higher column (tree.py:11-15).
__device__ __forceinline__ int group_argmax(double s, int k)
{
#pragma unroll
    for (int m = 1; m < GROUP; m <<= 1) {
        const double os = __shfl_xor(s, m, GROUP);
        const int ok = __shfl_xor(k, m, GROUP);
        if (os > s || (os == s && ok > k)) { s = os; k = ok; }
    }
    return k;
}

// ---- DPP butterfly over an 8-lane group (no LDS crossbar: v_mov_dpp runs at VALU speed) --------
// step 0: quad_perm [1,0,3,2] (lane^1), step 1: quad_perm [2,3,0,1] (lane^2), step 2: row_half_mirror
// (lane i <-> 7-i inside each aligned 8-lane half row).  After the three steps every lane of the
// group holds the same winner, because ties are broken by a total order on (score, k).
template <int CTRL>
__device__ __forceinline__ uint32_t dpp_u32(uint32_t v)
{
    // bound_ctrl with full row/bank masks: every lane has a valid source in these permutations, and the
    // compiler then needs no "old value" register initialised in front of each v_mov_b32_dpp
    return (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, CTRL, 0xF, 0xF, true);
}
template <int CTRL>
__device__ __forceinline__ double dpp_f64(double v)
{
    const uint64_t b = (uint64_t)__double_as_longlong(v);
    const uint64_t r = ((uint64_t)dpp_u32<CTRL>((uint32_t)(b >> 32)) << 32) | dpp_u32<CTRL>((uint32_t)b);
    return __longlong_as_double((long long)r);
}
struct Pick {      // a candidate child with its record riding along
    double s;      // PUCT score
    int k;         // child index in the block (ascending column => tie-break on k == on column)
    uint32_t n, info;
    double w;
};
// One butterfly step.  Score and index are exchanged with v_mov_b32_dpp and compared; the record
// fields that only ride along are selected with v_cndmask_b32_dpp (own value where VCC, else the
// partner's, permuted inside the select), one instruction per field instead of mov_dpp + cndmask.
// keep-own = (s > os) | (s == os & k >= ok): ties go to the higher column (tree.py:11-15).
// DPP hazard: 2 wait states between the VALU write of a source and its DPP read -- s_mov + s_nop.
#define C4_DPP_SELECT(CTRL)                                                                              \
    asm volatile("s_mov_b64 vcc, %[m]\n"                                                                 \
                 "s_nop 0\n"                                                                             \
                 "v_cndmask_b32_dpp %[n], %[n], %[n], vcc " CTRL " row_mask:0xf bank_mask:0xf\n"         \
                 "v_cndmask_b32_dpp %[i], %[i], %[i], vcc " CTRL " row_mask:0xf bank_mask:0xf\n"         \
                 "v_cndmask_b32_dpp %[wl], %[wl], %[wl], vcc " CTRL " row_mask:0xf bank_mask:0xf\n"      \
                 "v_cndmask_b32_dpp %[wh], %[wh], %[wh], vcc " CTRL " row_mask:0xf bank_mask:0xf\n"      \
                 : [n] "+v"(n), [i] "+v"(info), [wl] "+v"(wl), [wh] "+v"(wh)                              \
                 : [m] "s"(m)                                                                             \
                 : "vcc")
#if C4_ASM_PICK
// keep-own = (s > os) | (s == os & own index > partner's index).  In a butterfly over lanes that start with k == lane,
// the candidates a lane holds at a step all lie on its side of the exchanged bit (step 1: the lane itself; step 2: its
// pair; step 3: its quad), so "own index > partner's index" is a constant of the lane: the exchanged bit of its lane id.
// The tie-break therefore needs no index compare and the index itself need not travel through the compares: keep-own =
// gt | (ge & LANES_WITH_THE_BIT) as wave masks (two v_cmp into SGPR pairs + two scalar ops), to VCC once; the selects of
// the score and the v_cndmask_b32_dpp selects of the fields that ride along (the index rides in the top bits of n) then
// all read VCC in one block.
#define C4_PICK_SELECT(CTRL)                                                                             \
    asm volatile("s_mov_b64 vcc, %[m]\n"                                                                 \
                 "s_nop 0\n"                                                                             \
                 "v_cndmask_b32 %[sl], %[ol], %[sl], vcc\n"                                               \
                 "v_cndmask_b32 %[sh], %[oh], %[sh], vcc\n"                                               \
                 "v_cndmask_b32_dpp %[n], %[n], %[n], vcc " CTRL " row_mask:0xf bank_mask:0xf\n"          \
                 "v_cndmask_b32_dpp %[i], %[i], %[i], vcc " CTRL " row_mask:0xf bank_mask:0xf\n"          \
                 "v_cndmask_b32_dpp %[wl], %[wl], %[wl], vcc " CTRL " row_mask:0xf bank_mask:0xf\n"       \
                 "v_cndmask_b32_dpp %[wh], %[wh], %[wh], vcc " CTRL " row_mask:0xf bank_mask:0xf\n"       \
                 : [sl] "+v"(sl), [sh] "+v"(sh), [n] "+v"(n), [i] "+v"(info), [wl] "+v"(wl), [wh] "+v"(wh) \
                 : [m] "s"(m), [ol] "v"(ol), [oh] "v"(oh)                                                   \
                 : "vcc")
template <int CTRL>
__device__ __forceinline__ void pick_step(double &s, uint32_t &n, uint32_t &info, uint32_t &wl, uint32_t &wh)
{
    const uint64_t sb = (uint64_t)__double_as_longlong(s);
    uint32_t sl = (uint32_t)sb, sh = (uint32_t)(sb >> 32);
    const uint32_t ol = dpp_u32<CTRL>(sl), oh = dpp_u32<CTRL>(sh);
    const double os = __longlong_as_double((long long)(((uint64_t)oh << 32) | ol));
    // lanes whose id has the bit this step exchanges: they hold the higher indices
    constexpr unsigned long long HIGH = CTRL == 0xB1 ? 0xAAAAAAAAAAAAAAAAull : (CTRL == 0x4E ? 0xCCCCCCCCCCCCCCCCull : 0xF0F0F0F0F0F0F0F0ull);
    const unsigned long long m = __builtin_amdgcn_fcmp(s, os, 2 /* ogt */) | (__builtin_amdgcn_fcmp(s, os, 3 /* oge */) & HIGH);
    if (CTRL == 0xB1) C4_PICK_SELECT("quad_perm:[1,0,3,2]");
    else if (CTRL == 0x4E) C4_PICK_SELECT("quad_perm:[2,3,0,1]");
    else C4_PICK_SELECT("row_half_mirror");
    s = __longlong_as_double((long long)(((uint64_t)sh << 32) | sl));
}
#undef C4_PICK_SELECT
#else
template <int CTRL>
__device__ __forceinline__ void pick_step(double &s, int &k, uint32_t &n, uint32_t &info, uint32_t &wl, uint32_t &wh)
{
    const double os = dpp_f64<CTRL>(s);
    const int ok = (int)dpp_u32<CTRL>((uint32_t)k);
    const bool keep = (s > os) | ((s == os) & (k >= ok));
    s = keep ? s : os;
    k = keep ? k : ok;
    const unsigned long long m = __builtin_amdgcn_ballot_w64(keep);
    if (CTRL == 0xB1) C4_DPP_SELECT("quad_perm:[1,0,3,2]");
    else if (CTRL == 0x4E) C4_DPP_SELECT("quad_perm:[2,3,0,1]");
    else C4_DPP_SELECT("row_half_mirror");
}
#endif
#undef C4_DPP_SELECT
#if C4_PICK_BPERM
// The argmax with the record moved ONCE: three butterfly steps on the score alone (v_max_f64 returns one of its operands: the
// maximum is exact), the lanes that hold it found by one compare, the highest of them (ties -> higher column, tree.py:11-15)
// by a count of leading zeros on the group's byte of the wave mask, and the winner's record fetched from its lane through the
// LDS crossbar (four ds_bpermute_b32) -- 21 instructions where the butterfly with the record riding along takes 51.
__device__ __forceinline__ void group_pick(Pick &a)
{
    double m = a.s, o;
    o = dpp_f64<0xB1>(m);
    asm("v_max_f64 %0, %1, %2" : "=v"(m) : "v"(m), "v"(o));
    o = dpp_f64<0x4E>(m);
    asm("v_max_f64 %0, %1, %2" : "=v"(m) : "v"(m), "v"(o));
    o = dpp_f64<0x141>(m);
    asm("v_max_f64 %0, %1, %2" : "=v"(m) : "v"(m), "v"(o));
    const unsigned long long eq = __builtin_amdgcn_fcmp(a.s, m, 1 /* oeq */);
    const int gb = (int)(threadIdx.x & 63u & ~(uint32_t)(GROUP - 1));
    const uint32_t bits = (uint32_t)(eq >> gb) & 0xffu;
    const int k = 31 - __builtin_clz(bits);
    const int src = (gb + k) << 2;
    const uint64_t wb = (uint64_t)__double_as_longlong(a.w);
    const uint32_t wl = (uint32_t)__builtin_amdgcn_ds_bpermute(src, (int)(uint32_t)wb), wh = (uint32_t)__builtin_amdgcn_ds_bpermute(src, (int)(uint32_t)(wb >> 32));
    a.n = (uint32_t)__builtin_amdgcn_ds_bpermute(src, (int)a.n);
    a.info = (uint32_t)__builtin_amdgcn_ds_bpermute(src, (int)a.info);
    a.w = __longlong_as_double((long long)(((uint64_t)wh << 32) | wl));
    a.k = k;
    a.s = m;
}
#else
__device__ __forceinline__ void group_pick(Pick &a)
{
    const uint64_t wb = (uint64_t)__double_as_longlong(a.w);
    uint32_t wl = (uint32_t)wb, wh = (uint32_t)(wb >> 32);
#if C4_ASM_PICK
    // the index rides in the top three bits of the visit count (a search has fewer than 2^19 simulations, c4_engine_create)
    uint32_t nk = a.n | ((uint32_t)(threadIdx.x & (GROUP - 1)) << 29);
    pick_step<0xB1>(a.s, nk, a.info, wl, wh);    // quad_perm [1,0,3,2]
    pick_step<0x4E>(a.s, nk, a.info, wl, wh);    // quad_perm [2,3,0,1]
    pick_step<0x141>(a.s, nk, a.info, wl, wh);   // row_half_mirror
    a.k = (int)(nk >> 29);
    a.n = nk & 0x1fffffffu;
#else
    pick_step<0xB1>(a.s, a.k, a.n, a.info, wl, wh);    // quad_perm [1,0,3,2]
    pick_step<0x4E>(a.s, a.k, a.n, a.info, wl, wh);    // quad_perm [2,3,0,1]
    pick_step<0x141>(a.s, a.k, a.n, a.info, wl, wh);   // row_half_mirror
#endif
    a.w = __longlong_as_double((long long)(((uint64_t)wh << 32) | wl));
}
#endif

// mcts.py:175-178: noise = Gamma(alpha, 1, size=7), zeroed on illegal columns and normalised (mcts.py:197-202)
// with NumPy's sequential sum: a Dirichlet draw over the legal moves.  Lane k holds column k's raw draw.
__device__ __forceinline__ double dirichlet_from_gamma(double nz, bool legal)
{
    if (!legal) nz = 0.0;
    double s = 0.0;
#pragma unroll
    for (int i = 0; i < 7; ++i) s = s + gshfl(nz, i);
    return nz / s;
}

// tree.py:75-82 sample_value_fn(lambda x: x ** 2): weights V^2 over the root's children (lane k = child k,
// `act` for k < nc), probabilities = weights / sum, then np.random.choice's inverse CDF for the uniform u:
// cdf = cumsum(p); cdf /= cdf[-1]; index = searchsorted(cdf, u, side='right').  Returns -1 when every
// weight is zero (the reference raises there).  V * V stands in for CPython's pow(V, 2.0): glibc's pow
// differs from the correctly rounded square by one ulp for ~0.09 % of arguments, which can only move a
// choice whose uniform lies within a few ulp of a CDF boundary (tests/test_gpu_rng.py bounds it).
__device__ __forceinline__ int sample_child_sq(double V, bool act, uint32_t nc, double u, int lane)
{
    const double w2 = act ? V * V : 0.0;
    double s2 = 0.0;
#pragma unroll
    for (int i = 0; i < 7; ++i) s2 = s2 + gshfl(w2, i);
    if (!(s2 > 0.0)) return -1;
    const double pk = w2 / s2;
    double acc = 0.0, cdf = 0.0;
#pragma unroll
    for (int i = 0; i < 7; ++i) {
        acc = acc + gshfl(pk, i);
        if (i == lane) cdf = acc;
    }
    const double last = gshfl(cdf, (int)nc - 1);
    cdf = cdf / last;
    const unsigned long long bal = __ballot(act && cdf <= u);
    const int cnt = __popcll((bal >> (((threadIdx.x & 63) / GROUP) * GROUP)) & 0xffull);
    return cnt < (int)nc ? cnt : (int)nc - 1;
}

// A finished game moves from its slot's staging rows into the ring of finished games (the reference's
// games.extend(game_batch), training.py:131): one c4_game_record per game, in completion order.  A game
// that finds the ring full (the host has not drained / exported for rec_cap games) is dropped and counted.
template <class DevT>
__device__ __forceinline__ void publish_game(DevT &d, int g, long long gid, int len, int32_t result, int lane)
{
    group_fence();   // this move's staging stores (lane 0 / lanes 0..6) are read back by other lanes below
    int ok = 0;
    unsigned long long t = 0;
    if (lane == 0) {
        const unsigned long long tail = *(volatile unsigned long long *)d.cold->ring_tail;
        unsigned long long old = *(volatile unsigned long long *)d.cold->ring_head;
        for (;;) {
            if (old - tail >= (unsigned long long)d.rec_cap) { atomicAdd(d.cold->dropped, 1ULL); break; }
            const unsigned long long seen = atomicCAS(d.cold->ring_head, old, old + 1);
            if (seen == old) { ok = 1; t = old; break; }
            old = seen;
        }
    }
    ok = gshfl(ok, 0);
    if (!ok) return;
    t = ((unsigned long long)gshfl((uint32_t)(t >> 32), 0) << 32) | gshfl((uint32_t)t, 0);
    c4_game_record *row = d.cold->ring + (size_t)(t % (unsigned long long)d.rec_cap);
    const size_t s0 = (size_t)g * 42;
    for (int i = lane; i < len; i += GROUP) {
        row->color0[i] = d.cold->stg_c0[s0 + i];
        row->color1[i] = d.cold->stg_c1[s0 + i];
        row->move[i] = d.cold->stg_move[s0 + i];
        row->value[i] = d.cold->stg_value[s0 + i];
    }
    double *rp = &row->policy[0][0];
    for (int i = lane; i < len * 7; i += GROUP) rp[i] = d.cold->stg_policy[s0 * 7 + i];
    if (lane == 0) { row->game_id = gid; row->length = len; row->result = result; }
}

// ---- evaluation cache ------------------------------------------------------------------------
__device__ __forceinline__ uint64_t group_xor64(uint64_t v)
{
    uint32_t lo = (uint32_t)v, hi = (uint32_t)(v >> 32);
    lo ^= dpp_u32<0xB1>(lo); hi ^= dpp_u32<0xB1>(hi);
    lo ^= dpp_u32<0x4E>(lo); hi ^= dpp_u32<0x4E>(hi);
    lo ^= dpp_u32<0x141>(lo); hi ^= dpp_u32<0x141>(hi);
    return ((uint64_t)hi << 32) | lo;
}
__device__ __forceinline__ uint64_t cache_key(uint64_t c0, uint64_t c1) { return c0 + (c0 | c1); }
template <class DevT>
__device__ __forceinline__ CacheEntry *cache_slot(DevT &d, uint64_t key)
{
    return d.cache + ((key * 0x9E3779B97F4A7C15ULL) >> (64 - d.cache_bits));
}
// check word over (key, value, prior[7]); lane k < 7 contributes prior[k], lane 7 the value.  Its only job is to reject an entry
// TORN by two concurrent writers (a key from one, payload words from the other): each lane's 32 payload bits, rotated by a
// lane-specific amount inside a 64-bit word, xor-reduced over the group and xor-ed with the key.  A mixed entry passes only if the
// xor of the differing lanes' (rotated) differences vanishes -- 2^-32 or less per tear, and tears need two writers on one
// entry within nanoseconds.  (Until round 3 every lane hashed its word with two 64-bit multiplies: ~25 instructions per probe
// and per insert on the tree waves' critical chain.)
__device__ __forceinline__ uint64_t cache_check(uint64_t key, float value, float prior_lane, int lane)
{
    const uint32_t bits = lane < 7 ? __float_as_uint(prior_lane) : __float_as_uint(value);
    const uint64_t w = ((uint64_t)bits << 32) | (uint32_t)(bits * 0x9E3779B1u + (uint32_t)lane);
    const int r = (lane * 7 + 3) & 63;
    return key ^ group_xor64((w << r) | (w >> ((64 - r) & 63)));
}
template <class DevT>
__device__ __forceinline__ bool cache_probe(DevT &d, uint64_t c0, uint64_t c1, int lane, float &value, float &prior_lane)
{
    const uint64_t key = cache_key(c0, c1);
    const CacheEntry *e = cache_slot(d, key);
    const uint64_t k = e->key, chk = e->check;
    value = e->value;
    prior_lane = lane < 7 ? e->prior[lane] : 0.0f;
    return k == key && chk == cache_check(key, value, prior_lane, lane);
}
template <class DevT>
__device__ __forceinline__ void cache_insert(DevT &d, uint64_t c0, uint64_t c1, int lane, float value, float prior_lane)
{
    const uint64_t key = cache_key(c0, c1);
    CacheEntry *e = cache_slot(d, key);
    const uint64_t chk = cache_check(key, value, prior_lane, lane);
    if (lane < 7) e->prior[lane] = prior_lane;
    if (lane == 0) { e->value = value; e->check = chk; e->key = key; }
}

template <typename T>
__device__ __forceinline__ void store_plane(void *planes, size_t idx, float v);
template <> __device__ __forceinline__ void store_plane<float>(void *p, size_t i, float v) { ((float *)p)[i] = v; }
template <> __device__ __forceinline__ void store_plane<__half>(void *p, size_t i, float v) { ((__half *)p)[i] = __float2half(v); }
template <> __device__ __forceinline__ void store_plane<hip_bfloat16>(void *p, size_t i, float v) { ((hip_bfloat16 *)p)[i] = hip_bfloat16(v); }

// ------------------------------------------------------------------------------------------
// the rollout-step kernel
// ------------------------------------------------------------------------------------------
// One rollout step of slot `g` by its 8-lane group (`gl` = the group's row in s_path).  Called by the
// standalone step kernel and by the fused self-play kernel.  `leaf_out` (optional, LDS or global) gets
// the emitted leaf's bitboards {color0, color1} or {0,0} when the slot emits nothing.
// `sm` (fused kernel): the slot's state lives in LDS for the whole launch; the evaluator's answers are
// then read at values_in[ai] / priors_in[ai*7..] (LDS as well).  Without `sm` the state is read from
// and written to the Dev arrays and ai == g.
// WAVE_SYNC (wave-autonomous fused kernel): the call ends as soon as any slot of this wave has left the
// loop (it needs the evaluator, parked, ...), so the wave can evaluate that leaf right away.
// DevT: `const Dev` (kernel argument, held in SGPRs) or a constant-address-space view of the engine's device
// copy: the wave kernel reads each field with a scalar load where it is used instead of holding ~70
// argument SGPRs (most of them spilled to VGPR lanes) for its whole lifetime.
// SPLIT (c4_selfplay_split_kernel: tree waves and network waves): a slot that needs the evaluator posts its leaf
// (`*req = REQ_POSTED`) and idles INSIDE the loop while the other slots of the wave keep walking: at the top of
// every iteration it looks at `*req` and applies the answer once a network wave has written REQ_ANSWERED.  The call
// ends at the deadline; a slot still waiting then carries its leaf into the next launch.
constexpr uint32_t REQ_IDLE = 0, REQ_POSTED = 1, REQ_TAKEN = 2, REQ_ANSWERED = 3;
template <int EVAL, bool STAMPS = true, bool LDS_STATE = false, bool WAVE_SYNC = false, bool PATH_KEPT = false, class DevT = const Dev,
          bool SPLIT = false>
__device__ __forceinline__ void tree_step(DevT &d, const int g, const int lane, const int gl,
                                          PathEntry (*s_path)[MAX_DEPTH], Rec (*s_l1)[GROUP],
                                          const void *__restrict__ values_in,
                                          const void *__restrict__ priors_in, void *__restrict__ planes_out,
                                          uint64_t *leaf_out, SlotMem *sm = nullptr, const int ai_lds = 0,
                                          uint32_t *wg_stats = nullptr, const unsigned long long deadline = 0,
                                          uint32_t *req = nullptr, const int diag_stride = 0)
{
    (void)diag_stride;   // (diagnostic builds only: -DC4_SPLIT_PHASES=1)
    static_assert(!SPLIT || (WAVE_SYNC && LDS_STATE && PATH_KEPT), "the split kernel keeps slot states and paths in LDS");
    if (leaf_out && lane < 2) leaf_out[lane] = 0;
    if (g >= d.slot_hi) return;
    if (LDS_STATE) {
        if (sm->state() != SLOT_ACTIVE) {
            if (lane == 0) sm->flags = SlotMem::pack(sm->state(), 0, sm->need_root());
            return;
        }
    } else if (d.state[g] != SLOT_ACTIVE) {
        if (lane == 0) d.has_leaf[g] = 0;
        return;
    }
    constexpr bool SCORE_F32 = (EVAL == C4_EVAL_EXTERNAL_F32);
    const int ai = LDS_STATE ? ai_lds : g;            // row of the evaluator's answer for this slot

    const Pool pool{d.pool + (size_t)g * d.cap * (BLOCK_BYTES / 8)};
    PathEntry *gpath = d.path + (size_t)g * MAX_DEPTH;

    // slot state (group-uniform registers)
    uint64_t root0, root1;
    uint32_t sims, nalloc, pdepth, pinfo, ply;
    int32_t pend, need_root;
    long long gid;
    if (LDS_STATE) {
        root0 = sm->root0; root1 = sm->root1;
        sims = sm->sims; nalloc = sm->nalloc; pend = sm->pend; pdepth = sm->pdepth; pinfo = sm->pinfo;
        need_root = sm->need_root(); ply = sm->ply; gid = sm->gid;
    } else {
        root0 = d.root_c0[g]; root1 = d.root_c1[g];
        sims = d.sims_done[g]; nalloc = d.n_alloc[g]; pend = d.pending[g]; pdepth = d.pending_depth[g];
        pinfo = d.pending_info[g]; need_root = d.need_root[g]; ply = d.ply[g]; gid = d.game_id[g];
    }
    double root_w = LDS_STATE ? sm->root_w : 0.0;   // LDS mode: the root record is never re-read (see the descent)
    int state = SLOT_ACTIVE;
    int has_leaf = 0;
    struct { uint32_t sims, expansions, children, terminal_sims, leaf_evals, depth_sum, moves, games_started,
                      games_finished, capped, cache_hits, cache_probes, bad_evals, spec_evals; } st = {};
    static_assert(sizeof(st) == N_STATS * 4, "launch-local stats mirror SlotStats");

    // evaluator answer for the pending leaf
    uint64_t leaf0 = 0, leaf1 = 0;
    double ev_value = 0.0, ev_prior = 0.0;   // ev_prior: lane k holds prior[k]
    bool apply_now = false;
    // where the pending leaf's descent path lives.  PATH_KEPT: the caller keeps every slot's path stack in LDS of its
    // own for the whole launch (no global round trip between emitting a leaf and applying its answer)
    bool path_lds = (EVAL == C4_EVAL_CENTRE) || PATH_KEPT;
    bool fresh_eval = false;                    // the answer came from the evaluator: remember it
    bool cached_answer = false;                 // the answer came from the evaluation cache (finite by construction)
    if (SPLIT && pend >= 0) {
        if (lds_ld(req) != REQ_ANSWERED) return;   // its leaf is still with the network waves
        if (lane == 0) lds_st(req, REQ_IDLE);
    }
    if (pend >= 0) {
        leaf0 = LDS_STATE ? sm->leaf0 : d.leaf_c0[g];
        leaf1 = LDS_STATE ? sm->leaf1 : d.leaf_c1[g];
        if (EVAL == C4_EVAL_EXTERNAL_F32) {
            ev_value = (double)((const float *)values_in)[ai];
            ev_prior = lane < 7 ? (double)((const float *)priors_in)[(size_t)ai * 7 + lane] : 0.0;
        } else if (EVAL == C4_EVAL_EXTERNAL_F64) {
            ev_value = ((const double *)values_in)[ai];
            ev_prior = lane < 7 ? ((const double *)priors_in)[(size_t)ai * 7 + lane] : 0.0;
        }
        apply_now = true;   // (C4_EVAL_CENTRE never leaves a leaf pending across launches)
        fresh_eval = (EVAL == C4_EVAL_EXTERNAL_F32) && d.cache != nullptr;
    }

    auto stamp = [&](int i) {
        if (STAMPS && d.has_stamps && blockIdx.x < 256 && threadIdx.x == 0) d.cold->stamps[blockIdx.x * 8 + i] = __builtin_amdgcn_s_memtime();
    };
    stamp(0);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    stamp(1);
    // A slot may walk at most `level_budget` descent levels per launch; a descent that runs out is
    // suspended (node, board and path are saved) and resumed by the next launch.  Every wave then
    // does about the same amount of work per launch instead of waiting for the deepest tree.
    int levels_left = d.level_budget > 0 ? d.level_budget : 0x7fffffff;
    // time budget: a slot starts no further evaluator-free simulation once the call has run this many
    // shader cycles, so cheap (shallow / terminal / cached) simulations are not rationed by count and
    // every wave of a launch ends at about the same time
    const unsigned long long t_begin = (!WAVE_SYNC && d.time_budget > 0) ? __builtin_amdgcn_s_memtime() : 0;
    bool resume = !(WAVE_SYNC && C4_NO_SUSPEND) && (pend == -2);   // the wave-autonomous kernels have no level budget: no descent is ever suspended there
    // Hot subtree in LDS: once a launch has loaded the root's sibling block it stays in s_l1 for all
    // further simulations of the launch (write-through on every backup), so level 1 of every later
    // descent is an LDS read instead of an HBM round trip.
    // (keeping the block valid from one tree call to the next, possible where the slot has LDS of its own, measured
    // -0.2 %: the reload is an L1/L2 hit, the bookkeeping is not free)
    bool l1_valid = false;
    int inner = 0;
    const unsigned long long wave_mask0 = WAVE_SYNC ? __builtin_amdgcn_ballot_w64(true) : 0ull;
    bool waiting = false;   // SPLIT: the leaf is with the network waves
#if C4_SPLIT_PHASES
    unsigned long long ph_t0 = 0, ph_iter = 0, ph_wait = 0, ph_apply = 0, ph_levels = 0, ph_n = 0, ph_ta = 0, ph_probe = 0, ph_term = 0;
    bool ph_was_waiting = false;
#endif
    for (;;) {
#if C4_SPLIT_PHASES
        if (SPLIT) { const unsigned long long tn = __builtin_amdgcn_s_memtime(); if (ph_t0) { (ph_was_waiting ? ph_wait : ph_iter) += tn - ph_t0; ph_n += ph_was_waiting ? 0 : 1; } ph_t0 = tn; ph_was_waiting = waiting; }
#endif
        if (SPLIT) {
#if C4_WAIT_CLOCK_LAZY
            // Nobody in this wave can walk (every lane still in the loop waits): sleep, and look at the clock HERE only -- while
            // some slot walks, the walking lanes' deadline check ends their part of the call and the waiting lanes arrive here.
            if (__builtin_amdgcn_ballot_w64(!waiting) == 0) {
                __builtin_amdgcn_s_sleep(4);
                if ((long long)(__builtin_amdgcn_s_memtime() - deadline) > 0) { has_leaf = 1; break; }
            }
            if (waiting) {
#else
            if (__builtin_amdgcn_ballot_w64(!waiting) == 0) __builtin_amdgcn_s_sleep(4);   // nobody in this wave can walk
            if (waiting) {
                if ((long long)(__builtin_amdgcn_s_memtime() - deadline) > 0) { has_leaf = 1; break; }
#endif
                if (lds_ld(req) != REQ_ANSWERED) continue;
#if C4_SPLIT_PHASES
                if (lane == 0) {   // diagnostic: answered -> picked up, in units of 64 cycles
                    uint32_t *sums = req - ai + 3 * diag_stride;
                    atomicAdd(&sums[2], ((uint32_t)__builtin_amdgcn_s_memtime() - lds_ld(req + 2 * diag_stride)) >> 6);
                }
#endif
                ev_value = (double)lds_ldf((const float *)values_in + ai);
                ev_prior = lane < 7 ? (double)lds_ldf((const float *)priors_in + (size_t)ai * 7 + lane) : 0.0;
                if (lane == 0) lds_st(req, REQ_IDLE);
                waiting = false;
                apply_now = true;
                fresh_eval = d.cache != nullptr;
            }
        }
        // ---------------------------------------------------------------- evaluate_node + expand + backup
        if (apply_now) {
#if C4_SPLIT_PHASES
            ph_ta = __builtin_amdgcn_s_memtime();
#endif
            apply_now = false;
            // The reference asserts that the net never answers NaN (model.py:258-263).  A NaN here would
            // poison every comparison of the argmax, so answers are made finite first: memory safety of
            // the walk must not depend on the evaluator (counted in stats as bad_evals).  Only finite answers reach the
            // evaluation cache, so a cache hit needs no check; in the split kernel the network wave has looked at a fresh
            // answer before it published it (sanitise_answer), and the tree waves carry no check at all.
            if (!(SPLIT && C4_NET_SANITISES) && !cached_answer) {   // (an answer of any evaluator: the device net, a host callable, the centre heuristic)
                const bool bad = !(ev_value >= 0.0 && ev_value <= 1.0) || !(ev_prior >= 0.0 && ev_prior <= 3.0e38);
                if (__ballot(bad) >> (((threadIdx.x & 63) / GROUP) * GROUP) & 0xffull) {
                    if (!(ev_value >= 0.0 && ev_value <= 1.0)) ev_value = 0.5;
                    if (!(ev_prior >= 0.0 && ev_prior <= 3.0e38)) ev_prior = 0.0;
                    st.bad_evals += 1;
                }
            }
            if (fresh_eval) {   // evaluators.py:21-24: position_table[key] = evaluate_fn(board)
                fresh_eval = false;
                cache_insert(d, leaf0, leaf1, lane, (float)ev_value, (float)ev_prior);
            }
            cached_answer = false;
            const uint64_t occ = leaf0 | leaf1;
            const int age = popc64(occ);
            const int mask = legal_mask_group(occ);              // tree.py:23 valid_moves (leaf is undecided)
            const bool legal = lane < 7 && ((mask >> lane) & 1);
            // mcts.py:197-202 normalise, in the prior's own dtype
            double prn;
            uint32_t pf64;
            if (SCORE_F32) {
                const float pf = legal ? (float)ev_prior : 0.0f;
                float s = 0.0f;
#if C4_COOP_LEGAL
                float pall[7];   // (all seven fetches in flight before the first add: the sum keeps NumPy's order)
#pragma unroll
                for (int i = 0; i < 7; ++i) pall[i] = gshfl(pf, i);
#pragma unroll
                for (int i = 0; i < 7; ++i) s = s + pall[i];
#else
#pragma unroll
                for (int i = 0; i < 7; ++i) s = s + gshfl(pf, i);
#endif
                prn = s > 0.0f ? (double)(pf / s) : (legal ? 1.0 / (double)__popc(mask) : 0.0);
                pf64 = 0;
            } else {
                const double pd = legal ? ev_prior : 0.0;
                double s = 0.0;
#pragma unroll
                for (int i = 0; i < 7; ++i) s = s + gshfl(pd, i);
                prn = s > 0.0 ? pd / s : (legal ? 1.0 / (double)__popc(mask) : 0.0);
                pf64 = 1;
            }
            // mcts.py:171-181 add_exploration_noise (root only)
            if (pdepth == 0 && d.use_noise) {
                double nz = 0.0;
                if (lane < 7) {
                    if (d.rng_tape) nz = (gid < d.cold->tape_games) ? d.cold->noise_tape[((size_t)gid * 42 + ply) * 7 + lane] : 0.0;
                    else nz = rng_gamma(d.seed, gid, ply, (uint32_t)lane, d.alpha);
                }
                nz = dirichlet_from_gamma(nz, legal);
                const double keep = 1.0 - d.frac;
                const double a = SCORE_F32 ? (double)((float)prn * (float)keep) : prn * keep;
                const double b = nz * d.frac;
                prn = a + b;
                pf64 = 1;
            }
            // tree.py:119-132: children in ascending column order, one 8-aligned block
            const uint32_t nchild = (uint32_t)__popc(mask);
            const uint32_t base = nalloc * GROUP;
            nalloc += 1;
#if C4_FILLERS
            {   // all eight records of the block are written: the children in column order, then fillers nobody can choose -- their q
                // is -inf, so their score (prior term + value, either form) is -inf by arithmetic
                const int below = (1 << lane) - 1;
                Rec nr{0.0, -std::numeric_limits<double>::infinity(), 0.0, 0u, pack_info(0, 0, ST_FRESH, 0, 0)};
                uint32_t k = nchild + (uint32_t)__popc(~mask & 0x7f & below);
                if (legal) {
                    k = (uint32_t)__popc(mask & below);
                    uint64_t c0 = leaf0, c1 = leaf1;
                    const uint32_t bit = (uint32_t)(H1 * lane + col_count(occ, lane));
                    const uint32_t cst = make_move(c0, c1, lane);
                    // q of a terminal child = its exact result (utils.py:19-22), see child_value_for
                    nr = Rec{0.0, cst >= ST_XWIN ? orient_q(0.5 * (double)(cst - ST_XWIN), age) : 0.0, prn, 0u, pack_info(0, 0, cst, bit, 0)};
                }
                *pool.rec(base + k) = nr;
            }
#else
            if (legal) {
                const uint32_t k = (uint32_t)__popc(mask & ((1 << lane) - 1));
                uint64_t c0 = leaf0, c1 = leaf1;
                const uint32_t bit = (uint32_t)(H1 * lane + col_count(occ, lane));
                const uint32_t cst = make_move(c0, c1, lane);
                const uint32_t idx = base + k;
                // q of a terminal child = its exact result (utils.py:19-22), see child_value_for
                *pool.rec(idx) = Rec{0.0, cst >= ST_XWIN ? orient_q(0.5 * (double)(cst - ST_XWIN), age) : 0.0, prn, 0u, pack_info(0, 0, cst, bit, 0)};
            }
#endif
            if (lane == 0) {   // mcts.py:132-134: position_value / search_value.add(value)
                if (l1_valid && pdepth == 1) {   // the leaf is a child of the root: keep the LDS copy current
                    Rec &c = s_l1[gl][pend & 7];
                    c.n = 1;
                    c.w = ev_value;
                    c.q = orient_q(ev_value, age - 1);
                    c.info = pack_info(base, nchild, ST_EVALUATED, info_bit(pinfo), pf64);
                }
                pool.n(pend) = 1;
                pool.w(pend) = ev_value;
                pool.q(pend) = orient_q(ev_value, age - 1);
                pool.info(pend) = pack_info(base, nchild, ST_EVALUATED, info_bit(pinfo), pf64);
            }
            // mcts.py:164-168 backpropagate over the ancestors (values captured during the descent)
            {   // (a path is rarely longer than the group is wide: the first eight entries without a loop around them)
                auto back_up = [&](uint32_t i) {
                    const PathEntry e = path_lds ? s_path[gl][i] : gpath[i];
                    const double nw = e.w + ev_value, nq = orient_q(div_normal(nw, (double)(e.n + 1)), age + (int)pdepth + (int)i + 1);   // parent's side: (root age + i - 1) & 1
                    pool.n(e.node) = e.n + 1;
                    pool.w(e.node) = nw;
                    pool.q(e.node) = nq;
                    if (l1_valid && i == 1) { Rec &c = s_l1[gl][e.node & 7]; c.n = e.n + 1; c.w = nw; c.q = nq; }
                };
#if C4_PEEL_BACKUP
                if ((uint32_t)lane < pdepth) back_up((uint32_t)lane);
                for (uint32_t i = lane + GROUP; i < pdepth; i += GROUP) back_up(i);
#else
                for (uint32_t i = lane; i < pdepth; i += GROUP) back_up(i);
#endif
            }
            root_w = pdepth == 0 ? ev_value : root_w + ev_value;
            if (pdepth == 0) l1_valid = false;   // a new sibling block under the root
            st.leaf_evals += 1;
            st.children += nchild;
            if (pdepth > 0) sims += 1;
            (void)age;
            pend = -1;
            group_fence();
            stamp(2);
#if C4_SPLIT_PHASES
            ph_apply += __builtin_amdgcn_s_memtime() - ph_ta;
#endif
        }

        // ---------------------------------------------------------------- new root (Tree(board), tree.py:62-64)
        if (need_root) {
            need_root = 0;
            l1_valid = false;
            nalloc = 1;
            sims = 0;
            pend = 0;
            pdepth = 0;
            pinfo = pack_info(0, 0, ST_FRESH, 0, 0);
            leaf0 = root0;
            leaf1 = root1;
            if (EVAL == C4_EVAL_CENTRE) {
                ev_value = centre_value(leaf0, leaf1);
                ev_prior = 1.0 / 7.0;
                apply_now = true;
                continue;
            }
            if (EVAL == C4_EVAL_EXTERNAL_F32 && d.cache) {   // evaluators.py:19-20 position_table.get
                float cv, cp;
                st.cache_probes += 1;
                if (cache_probe(d, leaf0, leaf1, lane, cv, cp)) {
                    st.cache_hits += 1;
                    ev_value = (double)cv;
                    ev_prior = (double)cp;
                    apply_now = true;
                    cached_answer = true;
                    continue;
                }
            }
            if (SPLIT) {   // hand the leaf to the network waves now: the other slots of this wave walk on
                if (lane == 0) {
                    lds_st64(&sm->leaf0, leaf0);
                    lds_st64(&sm->leaf1, leaf1);
#if C4_SPLIT_PHASES
                    lds_st(req + diag_stride, (uint32_t)__builtin_amdgcn_s_memtime());   // diagnostic: when the request was posted
#endif
                    lds_st(req, REQ_POSTED);   // LDS executes a wave's accesses in order: the board is there before the word says so
                }
                waiting = true;
                continue;
            }
            has_leaf = 1;
            break;
        }

        // ---------------------------------------------------------------- move choice (mcts.py:78-88)
        if (sims >= (uint32_t)d.S) {
            const uint32_t rinfo = pool.info(0);
            const uint32_t cb = info_base(rinfo), nc = info_nchild(rinfo);
            const bool act = lane < (int)nc;
            const uint32_t cn = act ? pool.n(cb + lane) : 0;
            const double cq = act ? pool.q(cb + lane) : 0.0;
            const uint32_t ci = act ? pool.info(cb + lane) : 0;
            const int root_age = popc64(root0 | root1);
            const int side = root_age & 1;
            const double V = act ? (C4_ORIENTED_Q ? cq : child_value_for(info_status(ci), cn, cq, side)) : 0.0;
            // tree.py:104-109 + :139-147 values policy
            double vs = 0.0;
#pragma unroll
            for (int i = 0; i < 7; ++i) vs = vs + gshfl(V, i);
            const double pol = act ? (vs == 0.0 ? 1.0 / (double)nc : V / vs) : 0.0;
            // choose
            int kb = -1;
            double u = -1.0;
            if (root_age < d.nsm) {
                if (d.rng_tape) u = (gid < d.cold->tape_games) ? d.cold->u_tape[(size_t)gid * 42 + ply] : -1.0;
                else { double u1; rng_uniform2(d.seed, gid, ply, 32u, 0, u, u1); }
            }
            if (u >= 0.0) kb = sample_child_sq(V, act, nc, u, lane);   // tree.py:75-82 sample_value_fn(x**2)
            if (kb < 0) kb = group_argmax(act ? V : -1.0, act ? lane : -1);   // tree.py:69-73 best_move
            const uint32_t bi = gshfl(ci, kb);
            const uint32_t bn = gshfl(cn, kb);
#if C4_ORIENTED_Q
            // (the record's q is the value for the root's mover; the absolute value is value_sum / visit_count: the very division the
            //  backups do, on the record's own w and n)
            const double cw = act ? pool.w(cb + lane) : 0.0;
            const double bq = div_normal(gshfl(cw, kb), (double)(bn > 0 ? bn : 1u));
#else
            const double bq = gshfl(cq, kb);
#endif
            const int mv = (int)info_move(bi);
            const uint32_t bst = info_status(bi);
            double absv;   // child.data.absolute_value (mcts.py:88)
            if (bst >= ST_XWIN) absv = 0.5 * (double)(bst - ST_XWIN);
            else if (bn > 0) absv = bq;
            else absv = __longlong_as_double(0x7ff8000000000000LL);
            // policy by column
            const int rmask = legal_mask(root0 | root1);
            double pol_col = 0.0;
            {
                const int kk = __popc(rmask & ((1 << lane) - 1));
                const double pv = gshfl(pol, kk & 7);
                if (lane < 7 && ((rmask >> lane) & 1)) pol_col = pv;
            }
            // training_game.py:12-15 record (board before the move), staged per slot until the game ends
            if (d.rec_cap > 0 && !d.stop_after_move) {
                const size_t r = (size_t)g * 42 + ply;
                if (lane == 0) {
                    d.cold->stg_c0[r] = root0;
                    d.cold->stg_c1[r] = root1;
                    d.cold->stg_move[r] = mv;
                    d.cold->stg_value[r] = absv;
                }
                if (lane < 7) d.cold->stg_policy[r * 7 + lane] = pol_col;
            }
            if (lane == 0) { d.cold->res_move[g] = mv; d.cold->res_value[g] = absv; }
            if (lane < 7) d.cold->res_policy[(size_t)g * 7 + lane] = pol_col;
            st.moves += 1;
            if (d.stop_after_move) {   // MCTS.make_move returns here; the tree stays readable
                state = SLOT_MOVE_DONE;
                break;
            }
            make_move(root0, root1, mv);   // board.make_move(child.name)
            ply += 1;
            if (bst >= ST_XWIN) {          // game over: training_game.py:17 game_data.result
                if (d.rec_cap > 0) publish_game(d, g, gid, (int)ply, (int32_t)(bst - ST_XWIN), lane);
                st.games_finished += 1;
                unsigned long long ng = 0;
                if (lane == 0) ng = atomicAdd(d.cold->next_game, 1ULL);
                ng = ((unsigned long long)gshfl((uint32_t)(ng >> 32), 0) << 32) | gshfl((uint32_t)ng, 0);
                if (d.games_target >= 0 && (long long)ng >= d.games_target) {
                    state = SLOT_PARKED;
                    break;
                }
                gid = (long long)ng;
                ply = 0;
                root0 = 0;
                root1 = 0;
                st.games_started += 1;
            }
            need_root = 1;
            continue;
        }

        // bound the launch: at most max_inner evaluator-free simulations per launch
        // a neighbour needs the network, or the launch's time quantum is over
        if (SPLIT) {
            if ((long long)(__builtin_amdgcn_s_memtime() - deadline) > 0) break;
        } else if (WAVE_SYNC && !resume && (__builtin_amdgcn_ballot_w64(true) != wave_mask0 ||
                                     ((inner & (C4_DEADLINE_EVERY - 1)) == 0 && (long long)(__builtin_amdgcn_s_memtime() - deadline) > 0))) break;
        if (!SPLIT && !resume && (inner >= d.max_inner || levels_left <= 0 ||
                        (!WAVE_SYNC && d.time_budget > 0 && inner > 0 && (long long)(__builtin_amdgcn_s_memtime() - t_begin) > d.time_budget))) {
            st.capped += 1;
            break;
        }
        inner += 1;

        // ---------------------------------------------------------------- descent (mcts.py:108-116)
        uint32_t cur = 0, cinfo, cN, depth = 0;
        double cW;
        uint64_t b0, b1;
        if (resume) {            // pick a suspended descent up where the previous launch left it
            resume = false;
            cur = d.cold->cont_cur[g];
            cinfo = pinfo;
            cN = d.cold->cont_n[g];
            cW = d.cold->cont_w[g];
            b0 = LDS_STATE ? sm->leaf0 : d.leaf_c0[g];
            b1 = LDS_STATE ? sm->leaf1 : d.leaf_c1[g];
            depth = pdepth;
            for (uint32_t i = lane; i <= depth; i += GROUP) s_path[gl][i] = gpath[i];
            pend = -1;
        } else {
            if (LDS_STATE) {
                // the root record is implied by the slot state: N = completed simulations + 1 (mcts.py:132-134,
                // :164-168), W is carried along, and its children are the first block of the fresh tree
                cinfo = pack_info(GROUP, (uint32_t)__popc(legal_mask_group(root0 | root1)), ST_EVALUATED, 0,
                                  (!SCORE_F32 || d.use_noise) ? 1u : 0u);
                cN = sims + 1;
                cW = root_w;
            } else {
                const Rec rr = *pool.rec(0);
                cinfo = rr.info;
                cN = rr.n;
                cW = rr.w;
            }
            b0 = root0;
            b1 = root1;
#if C4_PATH_ALL_LANES
            s_path[gl][0] = PathEntry{0u, cN, cW};
#else
            if (lane == 0) s_path[gl][0] = PathEntry{0u, cN, cW};
#endif
        }
        int age = popc64(b0 | b1);
#if C4_SPLIT_PHASES
        const unsigned long long ph_tl = __builtin_amdgcn_s_memtime();
#endif
        unsigned long long lvl_t0 = (STAMPS && d.has_stamps) ? __builtin_amdgcn_s_memtime() : 0, lvl_wait = 0, lvl_alu = 0, lvl_cnt = 0;
#if C4_EARLY_REQUEST
        // Software-pipelined level loop: the sibling block (and score-table entry) of the NEXT level is requested
        // the moment the argmax is known, in front of this level's bookkeeping (board replay, path entry,
        // counters), so that bookkeeping runs under the load instead of in front of it.
        // All 8 lanes of the group load "their" record of the 8-record sibling block, also the lanes beyond the node's
        // children (their records exist in the pool and are never scored): no predicate, no zero fill on the loads.
        // The level budget is a feature of c4_step; the wave-autonomous kernel (WAVE_SYNC) bounds its calls by time.
        constexpr bool BUDGET = !WAVE_SYNC;
        Rec r;
        double2 ab;
        bool go = info_status(cinfo) == ST_EVALUATED && (!BUDGET || levels_left > 0);
        if (go) {
            ab = d.tabAB[cN];
            asm volatile("" ::: "memory");   // the score-table entry is requested first (its latency hides under the block's)
            if (depth == 0 && l1_valid) {
                r = s_l1[gl][lane];                                    // hot subtree: LDS
            } else {
                r = *pool.rec(info_base(cinfo) + lane);                // two 16-byte loads per lane
                if (depth == 0) {                                      // first descent of the launch: stage the block
                    s_l1[gl][lane] = r;
                    l1_valid = true;
                }
            }
        } else {
            r = Rec{};
            ab = double2{0.0, 0.0};
        }
#if C4_VGPR_BASES
        // the two bases of the level loop's requests in vector registers (the kernel has ~130 scalars spilled to lanes: four
        // v_readlane per level otherwise)
        const double2 *tab_v = d.tabAB;
        const Rec *rec_v = pool.rec((uint32_t)lane);
        asm volatile("" : "+v"(tab_v), "+v"(rec_v));
#endif
#if C4_FILLERS
        b1 ^= b0;   // inside the loop b1 is the occupancy: every stone goes in, only o's stones need a select
#endif
        while (go) {
            if (BUDGET) levels_left -= 1;
            const uint32_t cb = info_base(cinfo), nc = info_nchild(cinfo), pf64 = info_pf64(cinfo);
            if (cN == 1) st.expansions += 1;   // first descent through an evaluated node == expand_node
            const bool act = lane < (int)nc;
            const uint32_t n = r.n, inf = r.info;
            const double w = r.w, q = r.q, p = r.p;
            const double A = ab.x, B = ab.y;
            if (STAMPS && d.has_stamps) {   // diagnostic: cycles this level still waits for its loads
                asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                const unsigned long long t1 = __builtin_amdgcn_s_memtime();
                lvl_wait += t1 - lvl_t0;
                lvl_t0 = t1;
            }
            const double V = C4_ORIENTED_Q ? q : child_value_for(info_status(inf), n, q, age & 1);
#if C4_SCORE_ALL
            // all eight lanes score (the lanes beyond the node's children hold records nobody reads otherwise): a select
            // instead of a branch around the division
#if C4_FILLERS
            (void)act;
            const double s = ucb_score(A, B, n, p, V, pf64);   // (a filler's q is -inf: so is its score)
#else
            const double sc = ucb_score(A, B, act ? n : 0u, p, V, pf64);
            const double s = act ? sc : -std::numeric_limits<double>::infinity();
#endif
#else
            const double s = act ? ucb_score(A, B, n, p, V, pf64) : -std::numeric_limits<double>::infinity();
#endif
            Pick best{s, act ? lane : -1, n, inf, w};
            group_pick(best);                                   // mcts.py:141-142 max((score, child))
            // ---- next level's request
            go = info_status(best.info) == ST_EVALUATED && (!BUDGET || levels_left > 0);
            Rec rn = r;
            double2 abn = ab;
            if (go) {
#if C4_VGPR_BASES
                abn = tab_v[best.n];
                asm volatile("" ::: "memory");
                rn = rec_v[info_base(best.info)];
#else
                abn = d.tabAB[best.n];
                asm volatile("" ::: "memory");
                rn = *pool.rec(info_base(best.info) + lane);
#endif
            }
            // ---- this level's bookkeeping
            cN = best.n;
            cW = best.w;
            cinfo = best.info;
            cur = cb + (uint32_t)best.k;
            {   // board.py:160-163 replayed: xor the recorded stone into the mover's colour (b1 is kept as occupancy ^ b0 inside the loop)
                const uint64_t stone = 1ULL << info_bit(cinfo);
                b0 ^= (age & 1) ? 0ULL : stone;
#if C4_FILLERS
                b1 ^= stone;
#else
                b1 ^= (age & 1) ? stone : 0ULL;
#endif
            }
            age += 1;
            depth += 1;
#if C4_PATH_ALL_LANES
            s_path[gl][depth] = PathEntry{cur, cN, cW};   // all eight lanes store the same entry: no exec-mask region in the level loop
#else
            if (lane == 0) s_path[gl][depth] = PathEntry{cur, cN, cW};
#endif
            r = rn;
            ab = abn;
            if (STAMPS && d.has_stamps) {
                const unsigned long long t2 = __builtin_amdgcn_s_memtime();
                lvl_alu += t2 - lvl_t0;
                lvl_t0 = t2;
                lvl_cnt += 1;
            }
        }
#if C4_FILLERS
        b1 ^= b0;
#endif
#else
        while (info_status(cinfo) == ST_EVALUATED && levels_left > 0) {
            levels_left -= 1;
            const uint32_t cb = info_base(cinfo), nc = info_nchild(cinfo), pf64 = info_pf64(cinfo);
            if (cN == 1) st.expansions += 1;   // first descent through an evaluated node == expand_node
            const bool act = lane < (int)nc;
            const uint32_t idx = cb + lane;
            // the score-table entry is requested FIRST so that its (L1/L2) latency hides under the
            // sibling-block loads instead of being paid after them; the barrier keeps the order
            const double2 ab = d.tabAB[cN];
            asm volatile("" ::: "memory");
            Rec r = {};
            if (depth == 0 && l1_valid) {
                if (act) r = s_l1[gl][lane];                  // hot subtree: LDS
            } else {
                if (act) r = *pool.rec(idx);                  // two 16-byte loads per lane
                if (depth == 0) {                             // first descent of the launch: stage the block
                    if (act) s_l1[gl][lane] = r;
                    l1_valid = true;
                }
            }
            const uint32_t n = r.n, inf = r.info;
            const double w = r.w, q = r.q, p = r.p;
            const double A = ab.x, B = ab.y;
            if (STAMPS && d.has_stamps) {   // diagnostic: cycles spent waiting for this level's loads
                asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                const unsigned long long t1 = __builtin_amdgcn_s_memtime();
                lvl_wait += t1 - lvl_t0;
                lvl_t0 = t1;
            }
            const double V = C4_ORIENTED_Q ? q : child_value_for(info_status(inf), n, q, age & 1);
            const double s = act ? ucb_score(A, B, n, p, V, pf64) : -std::numeric_limits<double>::infinity();
            Pick best{s, act ? lane : -1, n, inf, w};
            group_pick(best);                                   // mcts.py:141-142 max((score, child))
            cN = best.n;
            cW = best.w;
            cinfo = best.info;
            cur = cb + (uint32_t)best.k;
            {   // board.py:160-163 replayed: xor the recorded stone into the mover's colour
                const uint64_t stone = 1ULL << info_bit(cinfo);
                b0 ^= (age & 1) ? 0ULL : stone;
                b1 ^= (age & 1) ? stone : 0ULL;
            }
            age += 1;
            depth += 1;
            if (lane == 0) s_path[gl][depth] = PathEntry{cur, cN, cW};
            if (STAMPS && d.has_stamps) {
                const unsigned long long t2 = __builtin_amdgcn_s_memtime();
                lvl_alu += t2 - lvl_t0;
                lvl_t0 = t2;
                lvl_cnt += 1;
            }
        }
#endif
        if (STAMPS && d.has_stamps && blockIdx.x < 256 && threadIdx.x == 0) {
            d.cold->stamps[blockIdx.x * 8 + 7] = (lvl_wait << 32) | (lvl_alu & 0xffffffffu);
            d.cold->stamps[blockIdx.x * 8 + 6] = ((unsigned long long)lvl_cnt << 32) | depth;
        }
        if (!(WAVE_SYNC && C4_NO_SUSPEND) && info_status(cinfo) == ST_EVALUATED) {   // level budget exhausted mid-descent: suspend
            lds_fence();
            for (uint32_t i = lane; i <= depth; i += GROUP) gpath[i] = s_path[gl][i];
            if (lane == 0) {
                d.cold->cont_cur[g] = cur;
                d.cold->cont_n[g] = cN;
                d.cold->cont_w[g] = cW;
                if (LDS_STATE) { sm->leaf0 = b0; sm->leaf1 = b1; }
                else { d.leaf_c0[g] = b0; d.leaf_c1[g] = b1; }
            }
            pend = -2;
            pdepth = depth;
            pinfo = cinfo;
            break;
        }
#if C4_SPLIT_PHASES
        ph_levels += __builtin_amdgcn_s_memtime() - ph_tl;
#endif
        st.depth_sum += depth;
        stamp(3);
        const uint32_t lst = info_status(cinfo);
        if (lst >= ST_XWIN) {
#if C4_SPLIT_PHASES
            const unsigned long long ph_tt = __builtin_amdgcn_s_memtime();
#endif
            // mcts.py:125-128,134 + :164-168: terminal leaf, exact result, no evaluator
            const double value = 0.5 * (double)(lst - ST_XWIN);
            lds_fence();   // s_path written by lane 0
            {
                auto back_up = [&](uint32_t i) {
                    const PathEntry e = s_path[gl][i];
                    const double nw = e.w + value, nq = orient_q(div_normal(nw, (double)(e.n + 1)), age + (int)depth + (int)i + 1);
                    pool.n(e.node) = e.n + 1;
                    pool.w(e.node) = nw;
                    pool.q(e.node) = nq;
                    if (l1_valid && i == 1) { Rec &c = s_l1[gl][e.node & 7]; c.n = e.n + 1; c.w = nw; c.q = nq; }
                };
#if C4_PEEL_BACKUP
                if ((uint32_t)lane <= depth) back_up((uint32_t)lane);
                for (uint32_t i = lane + GROUP; i <= depth; i += GROUP) back_up(i);
#else
                for (uint32_t i = lane; i <= depth; i += GROUP) back_up(i);
#endif
            }
            root_w = root_w + value;
            sims += 1;
            st.sims += 1;
            st.terminal_sims += 1;
            group_fence();
#if C4_SPLIT_PHASES
            ph_term += __builtin_amdgcn_s_memtime() - ph_tt;
#endif
            continue;
        }
        // fresh non-terminal leaf: needs the evaluator (mcts.py:130)
        pend = (int32_t)cur;
        pdepth = depth;
        pinfo = cinfo;
        leaf0 = b0;
        leaf1 = b1;
        st.sims += 1;   // counted when issued; completes in the next launch's apply
        if (EVAL == C4_EVAL_CENTRE) {
            ev_value = centre_value(b0, b1);
            ev_prior = 1.0 / 7.0;
            lds_fence();   // s_path -> read by the apply above
            apply_now = true;
            continue;
        }
        lds_fence();
        if (EVAL == C4_EVAL_EXTERNAL_F32 && d.cache) {   // evaluators.py:19-20 position_table.get
            float cv, cp;
            st.cache_probes += 1;
#if C4_SPLIT_PHASES
            const unsigned long long ph_tp = __builtin_amdgcn_s_memtime();
            const bool ph_hit = cache_probe(d, b0, b1, lane, cv, cp);
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            ph_probe += __builtin_amdgcn_s_memtime() - ph_tp;
            if (ph_hit) {
#else
            if (cache_probe(d, b0, b1, lane, cv, cp)) {   // a hit is applied in place, like a terminal leaf
#endif
                st.cache_hits += 1;
                ev_value = (double)cv;
                ev_prior = (double)cp;
                path_lds = true;
                apply_now = true;
                cached_answer = true;
                continue;
            }
        }
        if (!PATH_KEPT)
            for (uint32_t i = lane; i < depth; i += GROUP) gpath[i] = s_path[gl][i];
        if (SPLIT) {   // hand the leaf to the network waves now: the other slots of this wave walk on
            if (lane == 0) {
                lds_st64(&sm->leaf0, leaf0);
                lds_st64(&sm->leaf1, leaf1);
#if C4_SPLIT_PHASES
                lds_st(req + diag_stride, (uint32_t)__builtin_amdgcn_s_memtime());
#endif
                lds_st(req, REQ_POSTED);   // LDS executes a wave's accesses in order: the board is there before the word says so
            }
            waiting = true;
            continue;
        }
        has_leaf = 1;
        break;
    }

    // ---------------------------------------------------------------- emit leaf + persist slot state
    stamp(4);
    if (has_leaf) {
        if (lane == 0) {
            if (LDS_STATE) { sm->leaf0 = leaf0; sm->leaf1 = leaf1; }
            else { d.leaf_c0[g] = leaf0; d.leaf_c1[g] = leaf1; }
        }
        if (leaf_out && lane == 0) { leaf_out[0] = leaf0; leaf_out[1] = leaf1; }
        if (planes_out) {
            const int o_to_move = (popc64(leaf0 | leaf1) & 1) ? 0 : 1;
            const size_t pb = (size_t)g * 126;
            for (int e = lane; e < 126; e += GROUP) {
                const float v = plane_element(leaf0, leaf1, o_to_move, e);
                if (d.planes_dtype == C4_PLANES_F32) store_plane<float>(planes_out, pb + e, v);
                else if (d.planes_dtype == C4_PLANES_F16) store_plane<__half>(planes_out, pb + e, v);
                else store_plane<hip_bfloat16>(planes_out, pb + e, v);
            }
        }
    }
    if (LDS_STATE) {
        // node records, cache lines and game records drain in the background: only this group reads
        // them back, and a wave's own memory operations stay in order
        if (lane == 0) {
            sm->root_w = root_w;
            sm->root0 = root0;
            sm->root1 = root1;
            sm->sims = sims;
            sm->nalloc = nalloc;
            sm->pend = (has_leaf || pend == -2) ? pend : -1;
            sm->pdepth = pdepth;
            sm->pinfo = pinfo;
            sm->ply = ply;
            sm->gid = gid;
            sm->flags = SlotMem::pack(state, has_leaf, need_root);
#if C4_SPLIT_PHASES
            if (SPLIT && d.has_stamps && g < 256) { unsigned long long *o = d.cold->stamps + (size_t)g * 8; o[0] = ph_iter; o[1] = ph_wait; o[2] = ph_apply; o[3] = ph_levels; o[4] = ph_n; o[5] = ph_probe; o[6] = ph_term; }
#endif
            const uint32_t *sv = (const uint32_t *)&st;   // counters: per workgroup, flushed once per launch
#pragma unroll
            for (int i = 0; i < N_STATS; ++i)
                if (sv[i]) atomicAdd(&wg_stats[i], sv[i]);
        }
        return;
    }
    if (lane == 0) {
        d.has_leaf[g] = has_leaf;
        d.root_c0[g] = root0;
        d.root_c1[g] = root1;
        d.sims_done[g] = sims;
        d.n_alloc[g] = nalloc;
        d.pending[g] = (has_leaf || pend == -2) ? pend : -1;
        d.pending_depth[g] = pdepth;
        d.pending_info[g] = pinfo;
        d.need_root[g] = need_root;
        d.ply[g] = ply;
        d.game_id[g] = gid;
        d.state[g] = state;
        uint64_t *sp = d.stats + (size_t)g * N_STATS;
        const uint32_t *sv = (const uint32_t *)&st;
#pragma unroll
        for (int i = 0; i < N_STATS; ++i) sp[i] += sv[i];
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    stamp(5);
}

template <int EVAL>
__global__ __launch_bounds__(BLOCK) void c4_step_kernel(Dev d, const void *__restrict__ values_in,
                                                        const void *__restrict__ priors_in,
                                                        void *__restrict__ planes_out)
{
    __shared__ PathEntry s_path[SLOTS_PER_BLOCK][MAX_DEPTH];
    __shared__ Rec s_l1[SLOTS_PER_BLOCK][GROUP];   // hot subtree: the root's sibling block, LDS-resident for the launch
    const int lane = threadIdx.x & (GROUP - 1);
    const int gl = threadIdx.x / GROUP;
    const int g = d.slot_lo + blockIdx.x * SLOTS_PER_BLOCK + gl;
    tree_step<EVAL>(d, g, lane, gl, s_path, s_l1, values_in, priors_in, planes_out, nullptr);
}

// ------------------------------------------------------------------------------------------
// fused persistent self-play kernel: one workgroup = TS slots = one CU-sized unit that alternates
//   tree phase  (all 8 waves, TS/8 slots per wave in 8-lane groups, run tree_step)
//   net phase   (all 8 waves: net_forward_block on the emitted leaves, 16 per pass, straight from LDS)
// for n_steps rounds with NO kernel boundary, no inter-workgroup traffic and no global barrier:
// a workgroup only ever waits for its own trees, not for the deepest tree of the whole batch.
// The tree phase is bound by the latency of dependent sibling-block loads, not by issue slots, so
// more slots per wave cost it little while every network pass stays a full 16-position batch:
// TS = 32 when the batch is large enough to give every CU such a workgroup, else 16.
// ------------------------------------------------------------------------------------------
template <int TS>
__global__ __launch_bounds__(c4net::NTHREADS) void c4_selfplay_kernel(Dev d, c4net::NetDev nd, float *__restrict__ values,
                                                                      float *__restrict__ priors, int n_steps)
{
    using namespace c4net;
    static_assert(TS % NWAVES == 0 && TS % P == 0 && TS <= 64, "slots per workgroup");
    __shared__ __attribute__((aligned(16))) _Float16 act[2][(ROWS + 1) * CS];
    __shared__ __attribute__((aligned(16))) half8 wbuf[2][WCHUNKS];
    __shared__ __attribute__((aligned(16))) float4 mlp[MLP_F4];
    __shared__ uint64_t sleaf[2][TS];  // leaf bitboards handed from the tree phase to the net phase (compacted)
    __shared__ int smap[TS];           // compacted row -> slot of the workgroup
    __shared__ int sn;                 // number of leaves this round
    __shared__ SlotMem smem[TS];       // the workgroup's slot states, LDS-resident for the whole launch
    __shared__ float s_val[TS];        // network answers per slot
    __shared__ float s_pri[TS * 7];
    __shared__ uint32_t s_stats[N_STATS];   // counters of this launch (integer sums: order does not matter)
    static_assert((sizeof(PathEntry) * MAX_DEPTH + sizeof(Rec) * GROUP) * TS <= sizeof(_Float16) * ROWS * CS, "tree-phase LDS must fit the activation buffer");
    // the tree phase's path stacks live in activation buffer 0, which the net overwrites afterwards
    PathEntry (*s_path)[MAX_DEPTH] = reinterpret_cast<PathEntry (*)[MAX_DEPTH]>(&act[0][0]);
    Rec (*s_l1)[GROUP] = reinterpret_cast<Rec (*)[GROUP]>(&act[0][0] + sizeof(PathEntry) * MAX_DEPTH * TS / sizeof(_Float16));
    const int slot0 = blockIdx.x * TS;
    const int wv = threadIdx.x >> 6;
    // ---- launch prologue: slot states and the pending network answers, global -> LDS
    if (threadIdx.x < TS) {
        const int p = threadIdx.x, g = slot0 + p;
        SlotMem m = {};
        m.flags = SlotMem::pack(SLOT_PARKED, 0, 0);
        if (g < d.G) {
            m.root0 = d.root_c0[g]; m.root1 = d.root_c1[g]; m.leaf0 = d.leaf_c0[g]; m.leaf1 = d.leaf_c1[g];
            m.gid = d.game_id[g]; m.sims = d.sims_done[g]; m.nalloc = d.n_alloc[g]; m.pend = d.pending[g];
            m.pdepth = d.pending_depth[g]; m.pinfo = d.pending_info[g];
            m.ply = d.ply[g]; m.flags = SlotMem::pack(d.state[g], d.has_leaf[g], d.need_root[g]);
            m.root_w = *(const double *)(d.pool + (size_t)g * d.cap * (BLOCK_BYTES / 8));   // Rec::w of node 0
        }
        smem[p] = m;
    }
    if (threadIdx.x < N_STATS) s_stats[threadIdx.x] = 0;
    for (int i = threadIdx.x; i < TS * 8; i += NTHREADS) {   // answers of the previous launch
        const int p = i >> 3, k = i & 7;
        const bool ok = slot0 + p < d.G;
        if (k == 7) s_val[p] = ok ? values[slot0 + p] : 0.0f;
        else s_pri[p * 7 + k] = ok ? priors[(size_t)(slot0 + p) * 7 + k] : 0.0f;
    }
    __syncthreads();
    unsigned long long t_tree = 0, t_net = 0, t_own = 0;   // diagnostic (C4_TREE_STAMPS=1)
    for (int step = 0; step < n_steps; ++step) {
        const unsigned long long ta = d.has_stamps ? __builtin_amdgcn_s_memtime() : 0;
        // The thread id is laundered once per step: everything the tree phase derives from it (LDS and
        // pool addresses) is then recomputed here instead of being hoisted out of the step loop, where it
        // would stay live across the network phase (which needs every VGPR) and be spilled to scratch.
        int tid = threadIdx.x;
        asm volatile("" : "+v"(tid));
        const int lane = tid & (GROUP - 1);
        // slots that sit in one wave execute in SIMT lock-step, so consecutive slots go to different waves
        const int grp = (tid & 63) / GROUP;
        const int sl = (tid >> 6) + NWAVES * grp;     // slot of this 8-lane group inside the workgroup
        if (grp < TS / NWAVES)
            tree_step<C4_EVAL_EXTERNAL_F32, false, true>(d, slot0 + sl, lane, sl, s_path, s_l1, s_val, s_pri, nullptr, nullptr,
                                                         &smem[sl], sl, s_stats);
        const unsigned long long tb = d.has_stamps ? __builtin_amdgcn_s_memtime() : 0;
        __syncthreads();   // slot states (LDS) are visible to the whole workgroup
        const unsigned long long tc = d.has_stamps ? __builtin_amdgcn_s_memtime() : 0;
        t_own += tb - ta;
        t_tree += tc - ta;
        // compact the leaves: only slots that emitted one (cache miss) take a row of a network pass;
        // a pass skips the tiles beyond its last real row, and passes without any row are skipped
        if (threadIdx.x < 64) {
            const int p = threadIdx.x;
            const bool has = p < TS && smem[p < TS ? p : 0].has_leaf();
            const unsigned long long m = __ballot(has);
            if (has) {
                const int j = __popcll(m & ((1ULL << p) - 1));
                sleaf[0][j] = smem[p].leaf0;
                sleaf[1][j] = smem[p].leaf1;
                smap[j] = p;
            }
            if (p == 0) sn = __popcll(m);
        }
        __syncthreads();
        const int nleaf = sn;
        for (int r0 = 0; r0 < nleaf; r0 += P) {
            if (r0) __syncthreads();   // the previous pass has read its inputs and written its answers
            net_forward_block(nd, NetLds{act, wbuf, mlp}, sleaf[0] + r0, sleaf[1] + r0, min(P, nleaf - r0), 0, s_val, s_pri,
                              smap + r0);
        }
        __syncthreads();   // answers written; LDS free for the next tree phase
        if (d.has_stamps) t_net += __builtin_amdgcn_s_memtime() - tc;
    }
    // ---- launch epilogue: LDS -> global (the next launch, c4_step, read-outs and the host see the Dev arrays)
    if (threadIdx.x < TS && slot0 + threadIdx.x < d.G) {
        const int p = threadIdx.x, g = slot0 + p;
        const SlotMem m = smem[p];
        d.root_c0[g] = m.root0; d.root_c1[g] = m.root1; d.leaf_c0[g] = m.leaf0; d.leaf_c1[g] = m.leaf1;
        d.game_id[g] = m.gid; d.sims_done[g] = m.sims; d.n_alloc[g] = m.nalloc; d.pending[g] = m.pend;
        d.pending_depth[g] = m.pdepth; d.pending_info[g] = m.pinfo; d.need_root[g] = m.need_root();
        d.ply[g] = m.ply; d.state[g] = m.state(); d.has_leaf[g] = m.has_leaf() ? 1 : 0;
    }
    if (threadIdx.x < N_STATS) d.stats[(size_t)slot0 * N_STATS + threadIdx.x] += s_stats[threadIdx.x];   // the workgroup's row
    for (int i = threadIdx.x; i < TS * 8; i += NTHREADS) {
        const int p = i >> 3, k = i & 7;
        if (slot0 + p < d.G) {
            if (k == 7) values[slot0 + p] = s_val[p];
            else priors[(size_t)(slot0 + p) * 7 + k] = s_pri[p * 7 + k];
        }
    }
    if (d.has_stamps && blockIdx.x < 128 && (threadIdx.x & 63) == 0) {   // rows 2b: per-wave own tree work, 2b+1: phases
        unsigned long long *o = d.cold->stamps + blockIdx.x * 16;
        o[wv] = t_own;
        if (wv == 0) { o[8] = t_tree; o[9] = t_net; o[10] = (unsigned long long)n_steps; }
    }
}

// ------------------------------------------------------------------------------------------
// wave-autonomous fused self-play kernel: like c4_selfplay_kernel, but there is no workgroup barrier
// inside the step loop at all.  Every wave owns TS/8 slots and alternates
//   tree_step for its slots until one of them needs the network (or max_inner simulations each),
//   the wave-private network forward on that wave's own leaves (one position per pass, private LDS planes),
// so a tree never waits for the deepest tree of the workgroup or for a full 16-row batch: the answer
// is there ~20 k cycles after the miss.  Same games as every other path (the network arithmetic is
// bit-identical to net_forward_block's).
// ------------------------------------------------------------------------------------------
typedef __attribute__((address_space(4))) const Dev const_dev;
template <int TS, int MODE>
__global__ __launch_bounds__(c4net::NTHREADS) void c4_selfplay_wave_kernel(const Dev *d_dev, c4net::NetDev nd, float *__restrict__ values,
                                                                           float *__restrict__ priors, int n_steps)
{
    using namespace c4net;
    const_dev &d = *(const_dev *)d_dev;   // engine description: scalar loads from constant memory at the point of use
    constexpr int SPW = TS / NWAVES;   // slots per wave
    static_assert(TS % NWAVES == 0 && SPW >= 1 && SPW <= 8, "slots per workgroup");
    // private activation planes of the waves: the one-position forward on 16-row tiles needs 2 x 4,128 B (4 x with the
    // low-part planes of the reference-precision mode), the 64-filter forward 2 x 7,760 B
    constexpr int WBUF = WaveBuf<MODE>::HALVES;
    __shared__ __attribute__((aligned(16))) _Float16 act[NWAVES][WBUF];
    __shared__ __attribute__((aligned(16))) float4 mlp[MLP_F4];
    __shared__ SlotMem smem[TS];
    __shared__ float s_val[TS];
    __shared__ float s_pri[TS * 7];
    __shared__ uint32_t s_stats[N_STATS];
    __shared__ __attribute__((aligned(16))) float s_bias[BIAS_LDS_FLOATS];   // stem + conv biases (when the tower fits)
    __shared__ __attribute__((aligned(16))) uint16_t s_tab16[64 * TAB16];   // tap offsets of the 16-row forwards
    // 32-filter fp16 net (small planes): per slot and for the whole launch, LDS of its own for the descent path of the
    // simulation in flight (a leaf that waits for the network needs no copy of its path in global memory) and for the
    // root's sibling block (hot subtree).  The wider planes of the other modes leave no room: there the path stacks
    // alias the wave's planes (dead while the tree runs) and a waiting leaf's path goes through global memory.
    constexpr bool OWN_PATH = MODE == NETMODE_F32_F16;
    __shared__ __attribute__((aligned(16))) PathEntry s_path_own[OWN_PATH ? TS : 1][MAX_DEPTH];
    __shared__ __attribute__((aligned(16))) Rec s_l1_own[OWN_PATH ? TS : 1][GROUP];
    static_assert(OWN_PATH || (sizeof(PathEntry) * MAX_DEPTH + sizeof(Rec) * GROUP) * SPW <= sizeof(_Float16) * WBUF, "a wave's path stacks must fit its activation planes");
    const int slot0 = blockIdx.x * TS;
    const int wv = threadIdx.x >> 6;
    // ---- launch prologue: slot states, pending answers and the MLP tables, global -> LDS
    if (threadIdx.x < TS) {
        const int p = threadIdx.x, g = slot0 + p;
        SlotMem m = {};
        m.flags = SlotMem::pack(SLOT_PARKED, 0, 0);
        if (g < d.G) {
            m.root0 = d.root_c0[g]; m.root1 = d.root_c1[g]; m.leaf0 = d.leaf_c0[g]; m.leaf1 = d.leaf_c1[g];
            m.gid = d.game_id[g]; m.sims = d.sims_done[g]; m.nalloc = d.n_alloc[g]; m.pend = d.pending[g];
            m.pdepth = d.pending_depth[g]; m.pinfo = d.pending_info[g];
            m.ply = d.ply[g]; m.flags = SlotMem::pack(d.state[g], d.has_leaf[g], d.need_root[g]);
            m.root_w = *(const double *)(d.pool + (size_t)g * d.cap * (BLOCK_BYTES / 8));   // Rec::w of node 0
        }
        smem[p] = m;
    }
    if (threadIdx.x < N_STATS) s_stats[threadIdx.x] = 0;
    for (int i = threadIdx.x; i < TS * 8; i += NTHREADS) {   // answers of the previous launch
        const int p = i >> 3, k = i & 7;
        const bool ok = slot0 + p < d.G;
        if (k == 7) s_val[p] = ok ? values[slot0 + p] : 0.0f;
        else s_pri[p * 7 + k] = ok ? priors[(size_t)(slot0 + p) * 7 + k] : 0.0f;
    }
    for (int i = threadIdx.x; i < MLP_F4; i += NTHREADS) mlp[i] = nd.mlp[i];
    stage_bias_lds(nd, s_bias);
    if (threadIdx.x < 64) build_tab16<MODE == NETMODE_F64 ? CS64 : CS16>(s_tab16, threadIdx.x);
    if (OWN_PATH)
        for (int i = threadIdx.x; i < TS * MAX_DEPTH; i += NTHREADS) {   // paths of leaves pending from the previous launch
            const int p = i / MAX_DEPTH, k = i - p * MAX_DEPTH;
            if (slot0 + p < d.G) s_path_own[p][k] = d.path[(size_t)(slot0 + p) * MAX_DEPTH + k];
        }
    __syncthreads();
    PathEntry (*s_path)[MAX_DEPTH] = OWN_PATH ? s_path_own : reinterpret_cast<PathEntry (*)[MAX_DEPTH]>(&act[wv][0]);
    Rec (*s_l1)[GROUP] = OWN_PATH ? s_l1_own : reinterpret_cast<Rec (*)[GROUP]>(&act[wv][0] + sizeof(PathEntry) * MAX_DEPTH * SPW / sizeof(_Float16));
    // (a tree call ends when a slot of the wave blocks; max_inner bounds it)
    unsigned long long t_tree = 0, t_net = 0, n_pass = 0;   // diagnostic (C4_TREE_STAMPS=1)
    // A launch is a time quantum, not a number of rounds: every wave keeps alternating tree work and
    // network passes until n_steps * time_budget cycles have passed, so all waves of the launch end
    // together however many simulations their trees needed per network answer.
    const unsigned long long t_launch = __builtin_amdgcn_s_memtime();
    const unsigned long long quantum = (unsigned long long)n_steps * (unsigned long long)(d.time_budget > 0 ? d.time_budget : 80000);
    while (__builtin_amdgcn_s_memtime() - t_launch < quantum) {
        const unsigned long long ta = d.has_stamps ? __builtin_amdgcn_s_memtime() : 0;
        int tid = threadIdx.x;
        asm volatile("" : "+v"(tid));   // see c4_selfplay_kernel: keep the tree phase's addresses out of the net phase
        const int lane = tid & (GROUP - 1);
        const int grp = (tid & 63) / GROUP;
        const int sl = (tid >> 6) + NWAVES * grp;     // slot of this 8-lane group inside the workgroup
        if (grp < SPW)
            tree_step<C4_EVAL_EXTERNAL_F32, false, true, true, OWN_PATH>(d, slot0 + sl, lane, OWN_PATH ? sl : grp, s_path, s_l1, s_val, s_pri, nullptr,
                                                                         nullptr, &smem[sl], sl, s_stats, t_launch + quantum);
        lds_fence();   // the slot states written by the groups' first lanes are read by the whole wave
        const unsigned long long tb = d.has_stamps ? __builtin_amdgcn_s_memtime() : 0;
        // the wave's own leaves, two per network pass
        int pend_slot[SPW];
        int cnt = 0;
#pragma unroll
        for (int q = 0; q < SPW; ++q) {
            const int sq = wv + NWAVES * q;
            const int has = __builtin_amdgcn_readfirstlane(smem[sq].has_leaf() ? 1 : 0);
            if (has) pend_slot[cnt++] = sq;
        }
        if (MODE == NETMODE_F32_F16) {   // 32 filters, fp16: one position per pass on 16-row MFMA tiles
            for (int i = 0; i < cnt; ++i) {
                const int sa = pend_slot[i];
                net_forward_wave16(nd, &act[wv][0], mlp, s_bias, s_tab16, smem[sa].leaf0, smem[sa].leaf1, s_val, s_pri, sa,
                                   (C4_FUSED_NET_STAMPS && d.has_stamps && blockIdx.x < 16) ? d.cold->stamps + 2048 + (blockIdx.x * NWAVES + wv) * 16 : nullptr);
            }
        } else {   // reference-precision net or 64 filters: one position per pass on 32-row tiles
            for (int i = 0; i < cnt; ++i) {
                const int sa = pend_slot[i];
                net_forward_wave1_mode<MODE>(nd, &act[wv][0], mlp, s_bias, s_tab16, smem[sa].leaf0, smem[sa].leaf1, s_val, s_pri, sa, nd.w0);   // (eight waves' planes fill this kernel's LDS: the pass-start fragments come from L2 here)
            }
        }
        lds_fence();   // answers (LDS) before the next tree_step reads them
        if (d.has_stamps) { t_tree += tb - ta; t_net += __builtin_amdgcn_s_memtime() - tb; n_pass += cnt; }
    }
    if (d.has_stamps && blockIdx.x < 128 && (threadIdx.x & 63) == 0) {   // per wave: tree cycles, net cycles | passes << 48
        d.cold->stamps[blockIdx.x * 16 + wv] = t_tree;
        d.cold->stamps[blockIdx.x * 16 + 8 + wv] = t_net | (n_pass << 48);
    }
    __syncthreads();
    // ---- launch epilogue: LDS -> global
    if (OWN_PATH)
        for (int i = threadIdx.x; i < TS * MAX_DEPTH; i += NTHREADS) {   // paths of the leaves still waiting for their answers
            const int p = i / MAX_DEPTH, k = i - p * MAX_DEPTH;
            if (slot0 + p < d.G && smem[p].has_leaf()) d.path[(size_t)(slot0 + p) * MAX_DEPTH + k] = s_path_own[p][k];
        }
    if (threadIdx.x < TS && slot0 + threadIdx.x < d.G) {
        const int p = threadIdx.x, g = slot0 + p;
        const SlotMem m = smem[p];
        d.root_c0[g] = m.root0; d.root_c1[g] = m.root1; d.leaf_c0[g] = m.leaf0; d.leaf_c1[g] = m.leaf1;
        d.game_id[g] = m.gid; d.sims_done[g] = m.sims; d.n_alloc[g] = m.nalloc; d.pending[g] = m.pend;
        d.pending_depth[g] = m.pdepth; d.pending_info[g] = m.pinfo; d.need_root[g] = m.need_root();
        d.ply[g] = m.ply; d.state[g] = m.state(); d.has_leaf[g] = m.has_leaf() ? 1 : 0;
    }
    if (threadIdx.x < N_STATS) d.stats[(size_t)slot0 * N_STATS + threadIdx.x] += s_stats[threadIdx.x];
    for (int i = threadIdx.x; i < TS * 8; i += NTHREADS) {
        const int p = i >> 3, k = i & 7;
        if (slot0 + p < d.G) {
            if (k == 7) values[slot0 + p] = s_val[p];
            else priors[(size_t)(slot0 + p) * 7 + k] = s_pri[p * 7 + k];
        }
    }
}

// A network wave looks at an answer before it publishes it (the eight lanes 0..7 of the wave: value, prior[lane]): a value
// outside [0, 1] or a prior that is not a finite non-negative number is replaced (0.5 / 0) and counted (bad_evals), as
// tree_step's apply does it in the other kernels.
__device__ __forceinline__ void sanitise_answer(float *s_val, float *s_pri, int row, int lw, uint32_t *wg_stats)
{
    if (lw < GROUP) {
        const float v = s_val[row], p = lw < 7 ? s_pri[row * 7 + lw] : 0.0f;
        const bool bad_v = !(v >= 0.0f && v <= 1.0f), bad_p = !(p >= 0.0f && p <= 3.0e38f);
        if (__builtin_amdgcn_ballot_w64(bad_v || bad_p) & 0xffull) {
            if (bad_v && lw == 0) s_val[row] = 0.5f;
            if (bad_p && lw < 7) s_pri[row * 7 + lw] = 0.0f;
            if (lw == 0) atomicAdd(&wg_stats[offsetof(SlotStats, bad_evals) / sizeof(uint64_t)], 1u);
        }
    }
}

// ------------------------------------------------------------------------------------------
// c4_selfplay_split_kernel<TS, MODE>: tree waves and network waves.
// The wave-autonomous kernel above gives every wave 2 slots (16 of its 64 lanes walk trees) and lets it stop
// walking whenever one of them needs the network.  Here the first wave on every SIMD is a TREE wave that owns
// TS/4 slots (4 at 16 slots per CU: every tree instruction serves twice the lanes) and never runs the network;
// the second wave on every SIMD is a NETWORK wave.  A slot that misses the evaluation cache posts its leaf in LDS
// and steps out of the wave's loop, the other slots of the wave walk on; any idle network wave claims the request
// (compare-and-swap on the slot's request word), runs the one-position forward and marks the request answered; the
// waiting slot sees that at the top of its next loop iteration and applies the answer.
// Launch end: tree waves stop at the deadline; the network waves drain every posted request before they leave,
// so between launches "has a leaf" means "answered", exactly as with the wave kernel (the two are interchangeable
// launch by launch and play the same games).
// ------------------------------------------------------------------------------------------
template <int TS, int MODE, int TW = 4>
__global__ __launch_bounds__(c4net::NTHREADS) void c4_selfplay_split_kernel(const Dev *d_dev, c4net::NetDev nd, float *__restrict__ values,
                                                                            float *__restrict__ priors, int n_steps, int spread)
{
    using namespace c4net;
    const_dev &d = *(const_dev *)d_dev;
    constexpr int NW = NWAVES - TW;           // tree waves, network waves (TW = 4: one of each per SIMD)
    constexpr int SPW = (TS + TW - 1) / TW;   // slots per tree wave (at most)
    static_assert(TW >= 1 && NW >= 1 && SPW >= 1 && SPW <= 8 && TS <= 64, "slots per workgroup");
    // two positions per pass when two requests wait: pays once the network waves are the busier half (32 slots per CU: +1.5 %;
    // 16 slots: -1.5 %)
    constexpr bool PAIRS = MODE == NETMODE_F32_F16 && TS == 32 && C4_SPLIT_PAIRS;
    constexpr int WBUF = WaveBuf<MODE>::HALVES * (PAIRS ? 2 : 1);
    // speculative evaluation by network waves that have nothing else to do (4096 games: +4.3 %, f32x3 +1.2 %, 8192 games +0.2 %;
    // 64 filters -1.5 %: its network waves are never idle and the check is not free)
    // A speculative pass occupies its wave for a whole forward, and a real request that arrives meanwhile waits.  With the fp16 net
    // (17 k cycles per pass, network waves busy half the time) that is rare and speculation is worth +4 %; with the reference-precision
    // net (34 k cycles per pass, network waves busy 80 % with real requests alone) it COSTS 3.5 % (r03 A/B, profiles/r03_ab_speculation.json:
    // always 235 M, only while one / two / three other network waves are idle 236 / 239 / 241 M, never 244 M): off there, as for
    // 64 filters.
    constexpr bool SPECULATE = C4_SPECULATE && MODE == NETMODE_F32_F16;
    // (two network waves per position -- the forward split by cout tile, shared planes, a flag handshake per layer -- was built,
    // bit-identical, and measured: latency -28 %, but 43 k instead of 31 k wave-cycles per evaluation; -17 %.  profiles/r03_ab_pair_split_and_roles.json)
    __shared__ __attribute__((aligned(16))) _Float16 act[NW][WBUF];   // planes of the network waves
    __shared__ __attribute__((aligned(16))) float4 mlp[MLP_F4];
    __shared__ SlotMem smem[TS];
    __shared__ float s_val[TS + NW];        // rows TS..: scratch of the network waves' speculative passes
    __shared__ float s_pri[(TS + NW) * 7];
    __shared__ uint32_t s_stats[N_STATS];
    __shared__ __attribute__((aligned(16))) float s_bias[BIAS_LDS_FLOATS];
    __shared__ __attribute__((aligned(16))) uint16_t s_tab16[64 * TAB16];
    __shared__ __attribute__((aligned(16))) half8 s_w0[W0Lds<MODE>::FRAGS];  // reference-precision forward: the fragments a pass needs first
    __shared__ __attribute__((aligned(16))) PathEntry s_path[TS][MAX_DEPTH];   // every slot's descent path, for the whole launch
    __shared__ __attribute__((aligned(16))) Rec s_l1[TS][GROUP];               // every slot's root block
    __shared__ uint32_t s_req[C4_SPLIT_PHASES ? 3 * TS + 4 : TS];     // REQ_* of the slot's leaf (diagnostic build: + post / answer times, latency sums)
    __shared__ uint32_t s_simd[4];     // waves seen per SIMD (role assignment)
    __shared__ uint8_t s_perm[TW * SPW];   // the slot a (tree wave, group) pair walks (0xff: none), see the launch prologue
    __shared__ uint32_t s_tree_done;   // tree waves past the deadline
    // slot p of this workgroup.  Dense: TS consecutive slots per workgroup.  Spread (fewer slots than TS per CU): slot
    // blockIdx.x + p * gridDim.x, so that a batch smaller than TS x CUs still puts work on EVERY CU (1,200 games -- the
    // reference's generation, config.py:64 -- are 4-5 slots on each of 256 CUs instead of 16 slots on 75 of them).
    const int wg_first = spread ? (int)blockIdx.x : (int)blockIdx.x * TS, wg_stride = spread ? (int)gridDim.x : 1;
    auto gslot = [&](int p) -> int { return wg_first + p * wg_stride; };
    // ---- launch prologue: slot states, pending answers and the MLP tables, global -> LDS
    if (threadIdx.x < TS) {
        const int p = threadIdx.x, g = gslot(p);
        SlotMem m = {};
        m.flags = SlotMem::pack(SLOT_PARKED, 0, 0);
        if (g < d.G) {
            m.root0 = d.root_c0[g]; m.root1 = d.root_c1[g]; m.leaf0 = d.leaf_c0[g]; m.leaf1 = d.leaf_c1[g];
            m.gid = d.game_id[g]; m.sims = d.sims_done[g]; m.nalloc = d.n_alloc[g]; m.pend = d.pending[g];
            m.pdepth = d.pending_depth[g]; m.pinfo = d.pending_info[g];
            m.ply = d.ply[g]; m.flags = SlotMem::pack(d.state[g], d.has_leaf[g], d.need_root[g]);
            m.root_w = *(const double *)(d.pool + (size_t)g * d.cap * (BLOCK_BYTES / 8));   // Rec::w of node 0
        }
        smem[p] = m;
        s_req[p] = m.has_leaf() ? REQ_ANSWERED : REQ_IDLE;   // a leaf carried over from the previous launch has its answer
    }
    if (threadIdx.x < N_STATS) s_stats[threadIdx.x] = 0;
    if (threadIdx.x < 4) s_simd[threadIdx.x] = 0;
#if C4_SPLIT_PHASES
    if (threadIdx.x < 4) s_req[3 * TS + threadIdx.x] = 0;
#endif
    if (threadIdx.x == 0) s_tree_done = 0;
    for (int i = threadIdx.x; i < TS * 8; i += NTHREADS) {   // answers of the previous launch
        const int p = i >> 3, k = i & 7;
        const int g = gslot(p);
        const bool ok = g < d.G;
        if (k == 7) s_val[p] = ok ? values[g] : 0.0f;
        else s_pri[p * 7 + k] = ok ? priors[(size_t)g * 7 + k] : 0.0f;
    }
    for (int i = threadIdx.x; i < MLP_F4; i += NTHREADS) mlp[i] = nd.mlp[i];
    stage_bias_lds(nd, s_bias);
    if (W0Lds<MODE>::FRAGS > 1) stage_w0_lds(nd, s_w0);
    if (threadIdx.x < 64) build_tab16<MODE == NETMODE_F64 ? CS64 : CS16>(s_tab16, threadIdx.x);
    for (int i = threadIdx.x; i < TS * MAX_DEPTH; i += NTHREADS) {   // paths of leaves pending from the previous launch
        const int p = i / MAX_DEPTH, k = i - p * MAX_DEPTH;
        if (gslot(p) < d.G) s_path[p][k] = d.path[(size_t)gslot(p) * MAX_DEPTH + k];
    }
    __syncthreads();
    // Which slots a tree wave walks: the workgroup's ACTIVE slots, sorted by ply, dealt out in TW contiguous chunks of equal size
    // (+-1).  Games at a similar stage end their simulations in similar ways, so a wave's lock-step iterations spend less time in
    // phases only one of its slots needs (+1.6 % fp16 net, +0.3 % reference precision); and a workgroup that is not full -- a
    // small batch spread over the CUs, the tail of a generation when slots have parked -- keeps ALL its tree waves busy with
    // one or two slots each instead of filling wave 0 first.  (All writers are threads of wave 0: LDS executes them in order.)
    if (C4_SORT_SLOTS && threadIdx.x < 64) {
        const int p = threadIdx.x;
        if (p < TW * SPW) s_perm[p] = 0xff;
        const bool act = p < TS && smem[p < TS ? p : 0].state() == SLOT_ACTIVE;
        const int n_act = __popcll(__builtin_amdgcn_ballot_w64(act));
        if (act) {
            const uint32_t kp = smem[p].ply;
            int rank = 0;
            for (int q = 0; q < TS; ++q)
                if (smem[q].state() == SLOT_ACTIVE) { const uint32_t kq = smem[q].ply; rank += (kq < kp) || (kq == kp && q < p); }
            const int t = (rank * TW) / n_act;                       // chunk t holds the ranks r with floor(r TW / n_act) == t
            const int first = (t * n_act + TW - 1) / TW;             // ... the first of them
            s_perm[t * SPW + (rank - first)] = (uint8_t)p;
        }
    }
    // Answers carried over from the previous launch were written by whatever ran last -- a network wave of this kernel (already
    // finite) or c4_net_forward / a host evaluator between c4_step launches (include/c4_engine.h: the launches are
    // interchangeable) -- and the tree waves below apply answers without a check of their own: make them finite here.
    if (C4_NET_SANITISES && threadIdx.x < TS && smem[threadIdx.x].has_leaf()) {
        const int p = threadIdx.x;
        bool bad = false;
        const float v = s_val[p];
        if (!(v >= 0.0f && v <= 1.0f)) { s_val[p] = 0.5f; bad = true; }
        for (int k = 0; k < 7; ++k) {
            const float q = s_pri[p * 7 + k];
            if (!(q >= 0.0f && q <= 3.0e38f)) { s_pri[p * 7 + k] = 0.0f; bad = true; }
        }
        if (bad) atomicAdd(&s_stats[offsetof(SlotStats, bad_evals) / sizeof(uint64_t)], 1u);
    }
    // ---- roles: HW_ID.SIMD_ID says where the wave runs; the first wave to register on a SIMD walks trees
    const int simd = (int)__builtin_amdgcn_s_getreg((1 << 11) | (4 << 6) | 4) & 3;   // HW_REG_HW_ID bits [5:4]
    int rank = 0;
    if ((threadIdx.x & 63) == 0) rank = (int)atomicAdd(&s_simd[simd], 1u);
    rank = __builtin_amdgcn_readfirstlane(rank);
    __syncthreads();
    // (a placement other than two waves per SIMD -- not seen, the kernel's VGPR count allows no other -- falls back to
    // roles by wave index: a tree wave index nobody holds would leave its slots unserved)
    // TW == 3 (five network waves: the reference-precision forward costs twice the fp16 one, so the network waves are the
    // busier half there): SIMDs 0..2 carry a tree wave and a network wave, SIMD 3 two network waves.
    const bool even = (TW == 4 || TW == 3) && s_simd[0] == 2 && s_simd[1] == 2 && s_simd[2] == 2 && s_simd[3] == 2;
    const int wv = threadIdx.x >> 6;
    // (segregated roles -- both waves of SIMDs 0-1 walk trees, both waves of SIMDs 2-3 run the network, so that no tree wave loses
    // vector-issue slots to a neighbour's MFMAs -- measured -9.5 % with the f32x3 net, -2 % with the fp16 net: two network waves
    // then share one MFMA pipe.  profiles/r03_ab_pair_split_and_roles.json)
    const bool is_tree = even ? (rank == 0 && simd < TW) : wv < TW;
    const int role_idx = even ? (is_tree ? simd : (TW == 4 ? simd : (rank == 0 ? 4 : simd))) : (wv < TW ? wv : wv - TW);
    const unsigned long long t_launch = __builtin_amdgcn_s_memtime();
    const unsigned long long quantum = (unsigned long long)n_steps * (unsigned long long)(d.time_budget > 0 ? d.time_budget : 80000);
    unsigned long long t_busy = 0, n_pass = 0;   // diagnostic (C4_TREE_STAMPS=1)
    if (is_tree) {
        const int tw = role_idx;   // tree wave index
        while (__builtin_amdgcn_s_memtime() - t_launch < quantum) {
            const unsigned long long ta = d.has_stamps ? __builtin_amdgcn_s_memtime() : 0;
            int tid = threadIdx.x;
            asm volatile("" : "+v"(tid));   // keep lane-derived addresses of the call out of the loop's live ranges
            const int lane = tid & (GROUP - 1);
            const int grp = (tid & 63) / GROUP;
            const int sl0 = tw + TW * grp;        // slot of this 8-lane group inside the workgroup
            const int sl = (C4_SORT_SLOTS && grp < SPW) ? (int)s_perm[tw * SPW + grp] : sl0;   // (0xff = no slot: >= TS)
            bool runnable = false;
            if (grp < SPW && sl < TS) {
                const uint32_t fl = smem[sl].flags;
                runnable = (fl & 0xffu) == SLOT_ACTIVE && (((fl >> 8) & 0xffu) == 0 || lds_ld(&s_req[sl]) == REQ_ANSWERED);
            }
            if (__builtin_amdgcn_ballot_w64(runnable) == 0) {   // every slot waits for the network (or is parked)
                __builtin_amdgcn_s_sleep(8);
                continue;
            }
            if (grp < SPW && sl < TS)
                tree_step<C4_EVAL_EXTERNAL_F32, false, true, true, true, const_dev, true>(d, wg_first + sl * wg_stride, lane, sl, s_path, s_l1, s_val, s_pri, nullptr,
                                                                                          nullptr, &smem[sl], sl, s_stats, t_launch + quantum,
                                                                                          &s_req[sl], TS);
            lds_fence();
            if (d.has_stamps) t_busy += __builtin_amdgcn_s_memtime() - ta;
        }
        if ((threadIdx.x & 63) == 0) atomicAdd(&s_tree_done, 1u);
    } else {
        const int lw = threadIdx.x & 63;
        const int nw = role_idx;   // network wave index: its planes, its scratch row
        unsigned long long *const nstamps = (C4_FUSED_NET_STAMPS && d.has_stamps && blockIdx.x < 16) ? d.cold->stamps + 2048 + (blockIdx.x * NWAVES + nw) * 16 : nullptr;
        {
            // ---- speculative evaluation.  The evaluator's answer depends on the position alone and the evaluation cache is
            // transparent (evaluators.py:9-25 memo table), so evaluating a position EARLY changes no result.  The next
            // simulation that reaches the node just answered ends on its child with the highest prior (all children
            // unvisited: the PUCT score is pb_c x prior, mcts.py:147-161), and first-play-urgency 0 keeps the search on
            // that child for a long time: most evaluator calls of a search are such positions.  After a real answer, with no
            // request waiting, the wave notes that child (if the cache does not hold it) and evaluates it the next time
            // it finds NO real request: the answer goes into the cache, and when the search gets there its probe hits
            // instead of costing the slot a network round trip.  Tree waves do nothing for it.  (One call site of the forward
            // serves both kinds of pass: the kernel carries one copy of it, +1.5 %.)
            bool have_spec = false;
            uint64_t sc0 = 0, sc1 = 0;
            for (;;) {
                const uint32_t done = lds_ld(&s_tree_done);   // read BEFORE the requests: no post can follow a full count
                const uint32_t r = lw < TS ? lds_ld(&s_req[lw]) : REQ_IDLE;
                unsigned long long m = __builtin_amdgcn_ballot_w64(r == REQ_POSTED);
                // start looking at a different slot on every network wave, so that they do not race for the same request
                const int rot = (nw * TS) / NW;
                constexpr unsigned long long ALL = (TS == 64) ? ~0ull : ((1ull << TS) - 1);
                auto first_from = [&](unsigned long long mm) -> int {
                    const unsigned long long mr = ((mm >> rot) | (mm << (TS - rot))) & ALL;
                    return __builtin_amdgcn_readfirstlane((__builtin_ctzll(mr) + rot) % TS);
                };
                auto claim = [&](int slot) -> int {
                    int ok = 0;
                    if (lw == 0) ok = atomicCAS(&s_req[slot], REQ_POSTED, REQ_TAKEN) == REQ_POSTED;
                    return __builtin_amdgcn_readfirstlane(ok);
                };
                int c = -1, c2 = -1, row;
                uint64_t b0, b1;
                if (m != 0) {                       // real work first
                    c = first_from(m);
                    if (!claim(c)) continue;
                    m &= ~(1ull << c);
                    if (PAIRS && m != 0) {   // a second request is waiting: both in one pass (12 independent accumulator tiles keep the MFMA pipe busier)
                        c2 = first_from(m);
                        if (!claim(c2)) c2 = -1;
                    }
                    b0 = smem[c].leaf0;
                    b1 = smem[c].leaf1;
                    row = c;
                } else if (SPECULATE && have_spec) {
                    have_spec = false;
                    b0 = sc0;
                    b1 = sc1;
                    row = TS + nw;
                } else {
                    if (done == (uint32_t)TW) break;
                    __builtin_amdgcn_s_sleep(4);
                    continue;
                }
                const bool is_spec = c < 0;
                const unsigned long long ta = d.has_stamps ? __builtin_amdgcn_s_memtime() : 0;
#if C4_SPLIT_PHASES
                const uint32_t t_claim = (uint32_t)__builtin_amdgcn_s_memtime();
                if (!is_spec && lw == 0) { atomicAdd(&s_req[3 * TS + 0], (t_claim - lds_ld(&s_req[TS + c])) >> 6); atomicAdd(&s_req[3 * TS + 3], 1u); }   // posted -> claimed
#endif
                if (PAIRS && c2 >= 0) {
                    const uint64_t a0[2] = {b0, smem[c2].leaf0}, a1[2] = {b1, smem[c2].leaf1};
                    const int o[2] = {c, c2};
                    net_forward_wave16n<2>(nd, &act[nw][0], mlp, s_bias, s_tab16, a0, a1, s_val, s_pri, o);
                } else {
                    net_forward_wave1_mode<MODE>(nd, &act[nw][0], mlp, s_bias, s_tab16, b0, b1, s_val, s_pri, row, s_w0, nstamps);
                }
                lds_fence();   // the answer is in LDS before the request word says so
                if (C4_NET_SANITISES) {   // see tree_step's apply: answers leave this server finite
                    sanitise_answer(s_val, s_pri, row, lw, s_stats);
                    if (PAIRS && c2 >= 0) sanitise_answer(s_val, s_pri, c2, lw, s_stats);
                    lds_fence();
                }
                if (is_spec) {
                    if (lw < GROUP) cache_insert(d, b0, b1, lw, s_val[row], lw < 7 ? s_pri[row * 7 + lw] : 0.0f);
                    if (lw == 0) atomicAdd(&s_stats[offsetof(SlotStats, spec_evals) / sizeof(uint64_t)], 1u);
                    // (walking further down the line of highest priors -- stepping over cached positions, a second pass -- measured
                    // 3.5 % slower than stopping here: real requests wait while the wave probes)
                    continue;
                }
                // the answered position's priors, read before the slot may reuse its rows
                const float ppr = (SPECULATE && lw < 7) ? s_pri[c * 7 + lw] : -1.0f;
                if (lw == 0) {
#if C4_SPLIT_PHASES
                    const uint32_t t_ans = (uint32_t)__builtin_amdgcn_s_memtime();
                    lds_st(&s_req[2 * TS + c], t_ans);
                    atomicAdd(&s_req[3 * TS + 1], (t_ans - t_claim) >> 6);   // claimed -> answered
#endif
                    lds_st(&s_req[c], REQ_ANSWERED);
                    if (PAIRS && c2 >= 0) lds_st(&s_req[c2], REQ_ANSWERED);
                }
                if (PAIRS && c2 >= 0) n_pass += 1;
                if (d.has_stamps) { t_busy += __builtin_amdgcn_s_memtime() - ta; n_pass += 1; }
                if (SPECULATE && !(PAIRS && c2 >= 0) && d.cache != nullptr) {
                    const uint32_t r2 = lw < TS ? lds_ld(&s_req[lw]) : REQ_IDLE;
                    if (__builtin_amdgcn_ballot_w64(r2 == REQ_POSTED) != 0) continue;   // real work first
                    uint64_t c0 = b0, c1 = b1;
                    int go_spec = 0;
                    if (lw < GROUP) {
                        const int mask = legal_mask(b0 | b1);
                        const bool legal = lw < 7 && ((mask >> lw) & 1);
                        const int kb = group_argmax(legal ? (double)ppr : -1.0, legal ? lw : -1);   // ties: the higher column (tree.py:11-15)
                        const uint32_t cst = make_move(c0, c1, kb);
                        float cv, cp;
                        go_spec = (cst < ST_XWIN && !cache_probe(d, c0, c1, lw, cv, cp)) ? 1 : 0;
                    }
                    go_spec = __builtin_amdgcn_readfirstlane(go_spec);
                    if (!go_spec) continue;
                    sc0 = ((uint64_t)(uint32_t)__builtin_amdgcn_readfirstlane((int)(c0 >> 32)) << 32) | (uint32_t)__builtin_amdgcn_readfirstlane((int)c0);
                    sc1 = ((uint64_t)(uint32_t)__builtin_amdgcn_readfirstlane((int)(c1 >> 32)) << 32) | (uint32_t)__builtin_amdgcn_readfirstlane((int)c1);
                    have_spec = true;
                }
            }
        }
    }
    if (!C4_SPLIT_PHASES && d.has_stamps && blockIdx.x < 128 && (threadIdx.x & 63) == 0) {   // per wave: busy cycles (tree waves 0..3, network waves 8..11 | passes << 48)
        d.cold->stamps[blockIdx.x * 16 + (is_tree ? 0 : 8) + role_idx] = t_busy | (n_pass << 48);
        if (TW == 4) d.cold->stamps[blockIdx.x * 16 + (is_tree ? 4 : 12) + role_idx] = (unsigned long long)simd | ((unsigned long long)even << 8);
    }
    __syncthreads();
#if C4_SPLIT_PHASES
    if (d.has_stamps && blockIdx.x < 128 && threadIdx.x < 4) d.cold->stamps[4096 + blockIdx.x * 4 + threadIdx.x] = s_req[3 * TS + threadIdx.x];
#endif
    // ---- launch epilogue: LDS -> global
    for (int i = threadIdx.x; i < TS * MAX_DEPTH; i += NTHREADS) {   // paths of the leaves whose answers wait for the next launch
        const int p = i / MAX_DEPTH, k = i - p * MAX_DEPTH;
        if (gslot(p) < d.G && smem[p].has_leaf()) d.path[(size_t)gslot(p) * MAX_DEPTH + k] = s_path[p][k];
    }
    if (threadIdx.x < TS && gslot(threadIdx.x) < d.G) {
        const int p = threadIdx.x, g = gslot(p);
        const SlotMem m = smem[p];
        d.root_c0[g] = m.root0; d.root_c1[g] = m.root1; d.leaf_c0[g] = m.leaf0; d.leaf_c1[g] = m.leaf1;
        d.game_id[g] = m.gid; d.sims_done[g] = m.sims; d.n_alloc[g] = m.nalloc; d.pending[g] = m.pend;
        d.pending_depth[g] = m.pdepth; d.pending_info[g] = m.pinfo; d.need_root[g] = m.need_root();
        d.ply[g] = m.ply; d.state[g] = m.state(); d.has_leaf[g] = m.has_leaf() ? 1 : 0;
    }
    if (threadIdx.x < N_STATS && wg_first < d.G) d.stats[(size_t)wg_first * N_STATS + threadIdx.x] += s_stats[threadIdx.x];   // the row of the workgroup's first slot
    for (int i = threadIdx.x; i < TS * 8; i += NTHREADS) {
        const int p = i >> 3, k = i & 7;
        const int g = gslot(p);
        if (g < d.G) {
            if (k == 7) values[g] = s_val[p];
            else priors[(size_t)g * 7 + k] = s_pri[p * 7 + k];
        }
    }
}

// ------------------------------------------------------------------------------------------
// small kernels
// ------------------------------------------------------------------------------------------
__global__ void c4_reset_kernel(Dev d, const uint64_t *c0, const uint64_t *c1, int n_active)
{
    const int g = blockIdx.x * blockDim.x + threadIdx.x;
    if (g >= d.G) return;
    const bool active = g < n_active;
    d.root_c0[g] = (active && c0) ? c0[g] : 0;
    d.root_c1[g] = (active && c1) ? c1[g] : 0;
    d.leaf_c0[g] = 0;
    d.leaf_c1[g] = 0;
    d.has_leaf[g] = 0;
    d.pending[g] = -1;
    d.pending_depth[g] = 0;
    d.pending_info[g] = 0;
    d.sims_done[g] = 0;
    d.n_alloc[g] = 1;
    d.state[g] = active ? SLOT_ACTIVE : SLOT_PARKED;
    d.need_root[g] = active ? 1 : 0;
    d.ply[g] = 0;
    d.game_id[g] = g;
    d.cold->res_move[g] = -1;
    d.cold->res_value[g] = 0.0;
    for (int i = 0; i < 7; ++i) d.cold->res_policy[(size_t)g * 7 + i] = 0.0;
    uint64_t *sp = d.stats + (size_t)g * N_STATS;
    for (int i = 0; i < N_STATS; ++i) sp[i] = 0;
    if (active) sp[offsetof(SlotStats, games_started) / 8] = 1;
    if (g == 0) *d.cold->next_game = (unsigned long long)n_active;
    if (g == 0) { *d.cold->ring_head = 0; *d.cold->ring_tail = 0; *d.cold->dropped = 0; }
}

__global__ void c4_gather_roots_kernel(Dev d, c4_root_result *out)
{
    const int g = blockIdx.x * blockDim.x + threadIdx.x;
    if (g >= d.G) return;
    c4_root_result r;
    memset(&r, 0, sizeof(r));
    const Pool pool{d.pool + (size_t)g * d.cap * (BLOCK_BYTES / 8)};
    r.state = d.state[g];
    r.move = d.cold->res_move[g];
    r.value = d.cold->res_value[g];
    r.color0 = d.root_c0[g];
    r.color1 = d.root_c1[g];
    const uint32_t rinfo = pool.info(0);
    for (int i = 0; i < 7; ++i) { r.child_status[i] = -2; r.values_policy[i] = d.cold->res_policy[(size_t)g * 7 + i]; }
    if (info_status(rinfo) == ST_EVALUATED) {
        r.root_visits = pool.n(0);
        r.root_value_sum = pool.w(0);
        const uint32_t cb = info_base(rinfo), nc = info_nchild(rinfo);
        for (uint32_t k = 0; k < nc; ++k) {
            const uint32_t ci = pool.info(cb + k);
            const int m = (int)info_move(ci);
            r.child_visits[m] = pool.n(cb + k);
            r.child_value_sum[m] = pool.w(cb + k);
            r.child_status[m] = info_status(ci) >= ST_XWIN ? (int32_t)(info_status(ci) - ST_XWIN) : -1;
            r.root_prior[m] = pool.p(cb + k);
        }
    }
    const uint64_t *sp = d.stats + (size_t)g * N_STATS;
    r.expansions = (int64_t)sp[offsetof(SlotStats, expansions) / 8];
    r.simulations = (int64_t)sp[offsetof(SlotStats, sims) / 8];
    out[g] = r;
}

__global__ void k_make_move(const uint64_t *c0, const uint64_t *c1, const int32_t *col, int n, uint64_t *o0,
                            uint64_t *o1, int32_t *res)
{
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    uint64_t a = c0[i], b = c1[i];
    const uint32_t st = make_move(a, b, col[i]);
    o0[i] = a;
    o1[i] = b;
    res[i] = st >= ST_XWIN ? (int32_t)(st - ST_XWIN) : C4_RESULT_NONE;
}
__global__ void k_wins(const uint64_t *s, int n, int32_t *out)
{
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) out[i] = wins(s[i]) ? 1 : 0;
}
__global__ void k_valid_mask(const uint64_t *c0, const uint64_t *c1, int n, int32_t *out)
{
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    out[i] = position_status(c0[i], c1[i]) == ST_FRESH ? legal_mask(c0[i] | c1[i]) : 0;
}
__global__ void k_planes(const uint64_t *c0, const uint64_t *c1, int n, float *out)
{
    const size_t t = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= (size_t)n * 126) return;
    const size_t i = t / 126;
    const int e = (int)(t - i * 126);
    const int o_to_move = (popc64(c0[i] | c1[i]) & 1) ? 0 : 1;
    out[t] = plane_element(c0[i], c1[i], o_to_move, e);
}
__global__ void k_fliplr(const uint64_t *c0, const uint64_t *c1, int n, uint64_t *o0, uint64_t *o1)
{
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    o0[i] = flip_color(c0[i]);
    o1[i] = flip_color(c1[i]);
}
__global__ void k_centre(const uint64_t *c0, const uint64_t *c1, int n, double *out)
{
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) out[i] = centre_value(c0[i], c1[i]);
}

__global__ void k_count_active(const int32_t *state, int G, int32_t *out)   // one block
{
    __shared__ int total;
    if (threadIdx.x == 0) total = 0;
    __syncthreads();
    int c = 0;
    for (int g = threadIdx.x; g < G; g += blockDim.x) c += state[g] == SLOT_ACTIVE ? 1 : 0;
    atomicAdd(&total, c);
    __syncthreads();
    if (threadIdx.x == 0) *out = total;
}

// ---- device-side export of finished games (no host round trip: head, tail and the counts stay on the device)
// scratch: [0] games exported, [1] positions exported, [2] ring tail they start at, [4 + i] first position of game i
constexpr int EXPORT_OFFS = 4;
__global__ __launch_bounds__(1024) void k_export_scan(const Cold *cold, int rec_cap, int max_games, long long cap_pos, int64_t *scratch)
{
    __shared__ long long part[1024];
    __shared__ long long base;
    __shared__ int n_ok;
    const int tid = threadIdx.x;
    const unsigned long long head = *cold->ring_head, tail = *cold->ring_tail;
    long long avail = (long long)(head - tail);
    if (avail > max_games) avail = max_games;
    if (tid == 0) { base = 0; n_ok = 0; }
    __syncthreads();
    for (long long chunk = 0; chunk < avail; chunk += 1024) {
        const long long i = chunk + tid;
        const long long len = i < avail ? cold->ring[(tail + (unsigned long long)i) % (unsigned long long)rec_cap].length : 0;
        part[tid] = len;
        __syncthreads();
        for (int o = 1; o < 1024; o <<= 1) {   // inclusive scan of the chunk
            const long long v = tid >= o ? part[tid - o] : 0;
            __syncthreads();
            part[tid] += v;
            __syncthreads();
        }
        const long long incl = base + part[tid];
        if (i < avail) {
            scratch[EXPORT_OFFS + i] = incl - len;
            if (incl <= cap_pos) atomicMax(&n_ok, (int)(i + 1));   // lengths >= 1: the games that fit are a prefix
        }
        __syncthreads();
        if (tid == 0) base += part[1023];
        __syncthreads();
    }
    if (tid == 0) {
        const int n = n_ok;
        scratch[0] = n;
        scratch[1] = n ? scratch[EXPORT_OFFS + n - 1] + cold->ring[(tail + (unsigned long long)(n - 1)) % (unsigned long long)rec_cap].length : 0;
        scratch[2] = (int64_t)tail;
    }
}
__global__ __launch_bounds__(64) void k_export_write(const Cold *cold, int rec_cap, const int64_t *scratch, c4_export_buffers b)
{
    const long long n = scratch[0];
    const unsigned long long tail = (unsigned long long)scratch[2];
    const int t = threadIdx.x;
    for (long long i = blockIdx.x; i < n; i += gridDim.x) {
        const c4_game_record *row = cold->ring + (size_t)((tail + (unsigned long long)i) % (unsigned long long)rec_cap);
        const int len = row->length;
        const long long p = scratch[EXPORT_OFFS + i] + t;
        if (t < len) {
            if (b.boards_dev) { b.boards_dev[2 * p] = (int64_t)row->color0[t]; b.boards_dev[2 * p + 1] = (int64_t)row->color1[t]; }
            if (b.moves_dev) b.moves_dev[p] = (uint8_t)row->move[t];
            if (b.values_dev) b.values_dev[p] = (float)row->value[t];
            if (b.targets_dev) b.targets_dev[p] = 0.5f * (float)row->result;        // training_game.py:57-60
            if (b.policy_dev)
                for (int k = 0; k < 7; ++k) b.policy_dev[p * 7 + k] = (float)row->policy[t][k];
            if (b.game_index_dev) b.game_index_dev[p] = (int32_t)i;
        }
        if (t == 0) {
            if (b.lengths_dev) b.lengths_dev[i] = len;
            if (b.results_dev) b.results_dev[i] = (int8_t)row->result;
            if (b.ids_dev) b.ids_dev[i] = row->game_id;
        }
    }
}
__global__ void k_export_commit(const Cold *cold, const int64_t *scratch, int64_t *counts_out)
{
    *cold->ring_tail = (unsigned long long)(scratch[2] + scratch[0]);
    if (counts_out) { counts_out[0] = scratch[0]; counts_out[1] = scratch[1]; }
}

// data.py:78-105 native_to_pytorch(add_fliplr=True) on device: originals, then the mirrored copies
// (boards mirrored by column board.py:115-145, priors reversed, values duplicated)
__global__ void k_training_tensors(const int64_t *boards, const float *targets, const float *policy, long long n, int flip,
                                   float *ob, float *ov, float *op)
{
    const long long t = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    const long long total = n * (flip ? 2 : 1);
    if (t >= total * 126) return;
    const long long j = t / 126;
    const int e = (int)(t - j * 126);
    const bool mirrored = j >= n;
    const long long src = mirrored ? j - n : j;
    uint64_t c0 = (uint64_t)boards[2 * src], c1 = (uint64_t)boards[2 * src + 1];
    if (mirrored) { c0 = flip_color(c0); c1 = flip_color(c1); }
    const int o_to_move = (popc64(c0 | c1) & 1) ? 0 : 1;
    ob[t] = plane_element(c0, c1, o_to_move, e);
    if (e < 7) op[j * 7 + e] = policy[src * 7 + (mirrored ? 6 - e : e)];
    if (e == 7) ov[j] = targets[src];
}

// evaluation-cache read-out: what the cache answers for given positions (8 lanes per position, like the walk)
__global__ __launch_bounds__(64) void k_cache_lookup(Dev d, const uint64_t *c0, const uint64_t *c1, int n, float *value, float *prior,
                                                     int32_t *found)
{
    const int lane = threadIdx.x & (GROUP - 1);
    const int idx = (blockIdx.x * 64 + threadIdx.x) / GROUP;
    const int i = idx < n ? idx : n - 1;   // every lane takes part in the group reductions
    float v, pl;
    const bool hit = cache_probe(d, c0[i], c1[i], lane, v, pl);
    if (idx < n) {
        if (lane < 7) prior[(size_t)i * 7 + lane] = hit ? pl : 0.0f;
        if (lane == 0) { value[i] = hit ? v : 0.0f; found[i] = hit ? 1 : 0; }
    }
}

// production RNG read-outs: exactly the device functions the move choice / root noise use
__global__ __launch_bounds__(64) void k_debug_noise(uint64_t seed, double alpha, const long long *gid, const int32_t *ply,
                                                    const int32_t *legal_mask, int n, double *raw, double *dirichlet)
{
    const int lane = threadIdx.x & (GROUP - 1);
    const int idx = (blockIdx.x * 64 + threadIdx.x) / GROUP;
    const int i = idx < n ? idx : n - 1;
    const bool legal = lane < 7 && ((legal_mask[i] >> lane) & 1);
    const double g = lane < 7 ? rng_gamma(seed, gid[i], (uint32_t)ply[i], (uint32_t)lane, alpha) : 0.0;
    const double nz = dirichlet_from_gamma(g, legal);
    if (idx < n && lane < 7) { raw[(size_t)i * 7 + lane] = g; dirichlet[(size_t)i * 7 + lane] = nz; }
}
// div_normal against the compiler's IEEE division, on the device, over the operands the engine divides:
//   part 0: a = sqrt(N) (the score table's B of parent visit count N), b = n + 1   for 0 <= N < max_parent, 0 <= n < max_child
//   part 1: a = a value sum (float32 evaluator answers accumulated in float64, as a backup does), b = n + 1
__global__ void k_debug_div(int max_parent, int max_child, unsigned long long n_random, unsigned long long *mismatches)
{
    const unsigned long long tid = (unsigned long long)blockIdx.x * blockDim.x + threadIdx.x, nthreads = (unsigned long long)gridDim.x * blockDim.x;
    unsigned long long bad = 0;
    const unsigned long long pairs = (unsigned long long)max_parent * (unsigned long long)max_child;
    for (unsigned long long i = tid; i < pairs; i += nthreads) {
        const double a = sqrt((double)(i / (unsigned long long)max_child)), b = (double)(i % (unsigned long long)max_child + 1);
        const double q0 = a / b, q1 = div_normal(a, b);
        bad += __double_as_longlong(q0) != __double_as_longlong(q1);
    }
    for (unsigned long long i = tid; i < n_random; i += nthreads) {
        uint32_t c[4] = {(uint32_t)i, (uint32_t)(i >> 32), 0x5eedu, 0xd1f1u};
        philox4x32(c, 0x123456789abcdefULL);
        const uint32_t n = c[0] % (uint32_t)max_child;                       // visits so far
        // a sum of float32 values in [0, 1], the way visits add them (sometimes with draws' halves on top)
        const float v0 = (float)(c[1] >> 8) * (1.0f / 16777216.0f), v1 = (float)(c[2] >> 8) * (1.0f / 16777216.0f);
        const double w = (double)v0 * (double)(n / 2 + 1) + (double)v1 * (double)(n - n / 2) + ((c[3] & 1u) ? 0.5 * (double)(c[3] >> 20) : 0.0);
        const double b = (double)(n + 1);
        const double q0 = w / b, q1 = div_normal(w, b);
        bad += __double_as_longlong(q0) != __double_as_longlong(q1);
    }
    if (bad) atomicAdd(mismatches, bad);
}

__global__ __launch_bounds__(64) void k_debug_sample(uint64_t seed, const long long *gid, const int32_t *ply, const double *child_values,
                                                     const int32_t *n_children, const double *u_in, int n, double *u_out, int32_t *choice)
{
    const int lane = threadIdx.x & (GROUP - 1);
    const int idx = (blockIdx.x * 64 + threadIdx.x) / GROUP;
    const int i = idx < n ? idx : n - 1;
    const uint32_t nc = (uint32_t)n_children[i];
    const bool act = lane < (int)nc;
    const double V = act ? child_values[(size_t)i * 7 + lane] : 0.0;
    double u, u1;
    if (u_in) u = u_in[i];
    else rng_uniform2(seed, gid[i], (uint32_t)ply[i], 32u, 0, u, u1);
    const int kb = sample_child_sq(V, act, nc, u, lane);
    if (idx < n && lane == 0) { u_out[i] = u; choice[i] = kb; }
}

thread_local char g_err[512] = "";

void set_err(char *dst, const char *fmt, ...)
{
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(dst, 512, fmt, ap);
    va_end(ap);
    if (dst != g_err) { strncpy(g_err, dst, 511); g_err[511] = 0; }
}

}  // namespace

struct c4_engine {
    c4_config cfg;
    int device;
    hipStream_t stream;
    Dev d;
    Cold cold;        // host copy of *d.cold
    Cold *cold_dev;
    Dev *d_dev;       // device copy of `d` (wave-autonomous kernel), refreshed when `d` changed
    Dev d_uploaded;
    std::vector<void *> allocs;
    int32_t *active_dev;                  // scratch: live-slot count (c4_run_centre)
    int64_t *export_scratch;              // device: [0] games, [1] positions of the last export, [2..] per-game offsets
    int64_t launches;
    int fused_slots;      // slots per workgroup of the fused self-play kernel (16 or 32)
    int cus;              // compute units of the device
    int pack_dense;       // C4_FUSED_PACK=dense: consecutive slots per workgroup even when that leaves CUs idle (tests of ragged workgroups)
    int fused_wave;       // 1: wave-autonomous fused kernel (c4_selfplay_wave_kernel); 2: tree waves + network waves (c4_selfplay_split_kernel)
    int split_tw;         // tuning aid (C4_SPLIT_TW=2|4): tree waves of the split kernel at 16 slots per workgroup; 0 = by net mode
    int tape_games;
    double *tape_noise, *tape_u;
    char err[512];
};

#define HIPCHK(e, call)                                                                        \
    do {                                                                                       \
        hipError_t _r = (call);                                                                \
        if (_r != hipSuccess) {                                                                \
            set_err((e) ? (e)->err : g_err, "%s failed: %s (%s:%d)", #call, hipGetErrorString(_r), \
                    __FILE__, __LINE__);                                                       \
            return C4_EDEVICE;                                                                 \
        }                                                                                      \
    } while (0)

namespace {

int sync_cold(c4_engine *e)
{
    hipError_t r = hipMemcpy(e->cold_dev, &e->cold, sizeof(Cold), hipMemcpyHostToDevice);
    if (r != hipSuccess) { set_err(e->err, "cold-state upload failed: %s", hipGetErrorString(r)); return C4_EDEVICE; }
    return C4_OK;
}

template <typename T>
int dev_alloc(c4_engine *e, T **p, size_t count)
{
    void *q = nullptr;
    hipError_t r = hipMalloc(&q, count * sizeof(T) ? count * sizeof(T) : 16);
    if (r != hipSuccess) {
        set_err(e->err, "hipMalloc(%zu bytes) failed: %s", count * sizeof(T), hipGetErrorString(r));
        return C4_ENOMEM;
    }
    e->allocs.push_back(q);
    *p = (T *)q;
    return C4_OK;
}

int check_device(int device, char *err)
{
    int n = 0;
    hipError_t r = hipGetDeviceCount(&n);
    if (r != hipSuccess || n <= 0) {
        set_err(err, "no HIP device available (%s): the engine has no CPU fallback",
                r == hipSuccess ? "device count 0" : hipGetErrorString(r));
        return C4_EDEVICE;
    }
    if (device < 0 || device >= n) {
        set_err(err, "device %d out of range (have %d); there is no CPU backend", device, n);
        return C4_EDEVICE;
    }
    r = hipSetDevice(device);
    if (r != hipSuccess) {
        set_err(err, "hipSetDevice(%d) failed: %s", device, hipGetErrorString(r));
        return C4_EDEVICE;
    }
    return C4_OK;
}

// scratch helper for the pure board entry points
struct Scratch {
    std::vector<void *> ptrs;
    ~Scratch() { for (void *p : ptrs) (void)hipFree(p); }
    template <typename T>
    T *up(const T *host, size_t n, hipError_t &r)
    {
        void *q = nullptr;
        if (r != hipSuccess) return nullptr;
        r = hipMalloc(&q, n * sizeof(T) ? n * sizeof(T) : 16);
        if (r != hipSuccess) return nullptr;
        ptrs.push_back(q);
        if (host) r = hipMemcpy(q, host, n * sizeof(T), hipMemcpyHostToDevice);
        return (T *)q;
    }
};

}  // namespace

extern "C" {

int c4_abi_version(void) { return C4_ABI_VERSION; }

/* diagnostic: s_memtime stamps of the last launch, [256 blocks][8] (needs C4_TREE_STAMPS=1 at create) */
/* diagnostic: s_memtime stamps of the LAST network pass of each wave of workgroups 0..15 inside the last
 * c4_selfplay_steps launch, [16][8 waves][16] (phases as c4_net_debug_stamps); needs C4_TREE_STAMPS=1 */
int c4_debug_fused_net_stamps(c4_engine *e, unsigned long long *out)
{
    if (!e || !out || !e->cold.stamps) return C4_ESTATE;
    if (hipDeviceSynchronize() != hipSuccess) return C4_EDEVICE;
    return hipMemcpy(out, e->cold.stamps + 2048, 16 * 8 * 16 * 8, hipMemcpyDeviceToHost) == hipSuccess ? C4_OK : C4_EDEVICE;
}

/* diagnostic (build with -DC4_SPLIT_PHASES=1): request latency sums of workgroups 0..127 of the last split-kernel launch, [128][4] =
 * {posted->claimed, claimed->answered, answered->picked up} in units of 64 cycles, and the number of requests */
int c4_debug_latency_stamps(c4_engine *e, unsigned long long *out)
{
    if (!e || !out || !e->cold.stamps) return C4_ESTATE;
    if (hipDeviceSynchronize() != hipSuccess) return C4_EDEVICE;
    return hipMemcpy(out, e->cold.stamps + 4096, 128 * 4 * 8, hipMemcpyDeviceToHost) == hipSuccess ? C4_OK : C4_EDEVICE;
}

int c4_debug_stamps(c4_engine *e, unsigned long long *out)
{
    if (!e || !out || !e->cold.stamps) return C4_ESTATE;
    if (hipDeviceSynchronize() != hipSuccess) return C4_EDEVICE;
    return hipMemcpy(out, e->cold.stamps, 256 * 8 * 8, hipMemcpyDeviceToHost) == hipSuccess ? C4_OK : C4_EDEVICE;
}

const char *c4_last_error(const c4_engine *e) { return e ? e->err : g_err; }

int c4_engine_create(const c4_config *cfg, int device, c4_engine **out)
{
    if (!cfg || !out) { set_err(g_err, "c4_engine_create: null argument"); return C4_EINVAL; }
    *out = nullptr;
    if (cfg->abi_version != C4_ABI_VERSION) { set_err(g_err, "ABI version mismatch: caller %d, library %d", cfg->abi_version, C4_ABI_VERSION); return C4_EINVAL; }
    if (cfg->n_slots <= 0 || cfg->simulations <= 0 || cfg->pb_c_base <= 0) { set_err(g_err, "n_slots, simulations and pb_c_base must be positive"); return C4_EINVAL; }
    if (cfg->eval_mode < 0 || cfg->eval_mode > 2 || cfg->rng_mode < 0 || cfg->rng_mode > 1 || cfg->planes_dtype < 0 || cfg->planes_dtype > 2) { set_err(g_err, "bad eval_mode / rng_mode / planes_dtype"); return C4_EINVAL; }
    // every evaluated node takes one 8-slot block; evaluations per move <= simulations + 1
    const uint64_t cap = (uint64_t)GROUP * ((uint64_t)cfg->simulations + 3);
    if (cap > (1u << 22)) { set_err(g_err, "simulations=%d exceeds the 22-bit node index (max %d)", cfg->simulations, (1 << 19) - 3); return C4_ECAPACITY; }
    int rc = check_device(device, g_err);
    if (rc) return rc;

    c4_engine *e = new c4_engine();
    e->cfg = *cfg;
    e->device = device;
    e->stream = nullptr;
    e->launches = 0;
    {
        int cus = 0;
        if (hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, device) != hipSuccess || cus <= 0) cus = 256;
        e->cus = cus;
        // 16 slots per workgroup while that fills the chip once (one workgroup per CU: its LDS), 32 beyond; a batch that does
        // not fill TS x CUs slots is SPREAD over all CUs by the split kernel (see c4_selfplay_steps)
        e->fused_slots = (cfg->n_slots > 16 * cus) ? 32 : 16;
        e->pack_dense = 0;
        if (const char *pk = getenv("C4_FUSED_PACK")) e->pack_dense = strcmp(pk, "dense") == 0;
        e->fused_wave = 2;   // tree waves + network waves; C4_FUSED_MODE=wave / block select the wave-autonomous / workgroup-synchronous kernels
        if (const char *fm = getenv("C4_FUSED_MODE")) e->fused_wave = strcmp(fm, "block") == 0 ? 0 : (strcmp(fm, "wave") == 0 ? 1 : 2);
        e->split_tw = 0;     // 0: chosen by the net's mode
        if (const char *tw = getenv("C4_SPLIT_TW")) e->split_tw = atoi(tw);
        if (const char *fs = getenv("C4_FUSED_SLOTS")) {   // tuning aid: force 16 or 32
            const int v = atoi(fs);
            if (v == 16 || v == 32) e->fused_slots = v;
        }
    }
    e->tape_games = 0;
    e->tape_noise = nullptr;
    e->tape_u = nullptr;
    e->err[0] = 0;
    memset(&e->d, 0, sizeof(Dev));
    memset(&e->cold, 0, sizeof(Cold));
    e->cold_dev = nullptr;
    e->d_dev = nullptr;
    memset(&e->d_uploaded, 0xff, sizeof(Dev));   // differs from any real description: the first launch uploads
    Dev &d = e->d;
    const size_t G = (size_t)cfg->n_slots;
    d.G = cfg->n_slots;
    d.slot_lo = 0;
    d.slot_hi = cfg->n_slots;
    d.cap = (uint32_t)cap;
    d.S = cfg->simulations;
    d.nsm = cfg->num_sampling_moves;
    d.alpha = cfg->root_dirichlet_alpha;
    d.frac = cfg->root_exploration_fraction;
    d.use_noise = (cfg->root_dirichlet_alpha != 0.0 && cfg->root_exploration_fraction != 0.0) ? 1 : 0;  // mcts.py:174
    d.rng_tape = cfg->rng_mode == C4_RNG_TAPE;
    d.stop_after_move = cfg->stop_after_move ? 1 : 0;
    d.max_inner = cfg->max_inner_iters > 0 ? cfg->max_inner_iters
                                            : (cfg->eval_mode == C4_EVAL_CENTRE ? 1 << 20 : 1);
    d.level_budget = cfg->level_budget > 0 ? cfg->level_budget : 0;
    d.time_budget = cfg->time_budget_cycles > 0 ? cfg->time_budget_cycles : 0;
    d.planes_dtype = cfg->planes_dtype;
    d.games_target = cfg->games_target;
    d.seed = cfg->seed;
    d.rec_cap = cfg->stop_after_move ? 0 : (cfg->record_capacity_games > 0 ? cfg->record_capacity_games : 2 * cfg->n_slots);

#define ALLOC(ptr, count)                                   \
    if ((rc = dev_alloc(e, &(ptr), (count))) != C4_OK) {    \
        strncpy(g_err, e->err, 511);                        \
        c4_engine_destroy(e);                               \
        return rc;                                          \
    }
    ALLOC(d.pool, G * cap * (BLOCK_BYTES / 8));
    ALLOC(d.root_c0, G); ALLOC(d.root_c1, G); ALLOC(d.leaf_c0, G); ALLOC(d.leaf_c1, G);
    ALLOC(d.has_leaf, G); ALLOC(d.pending, G); ALLOC(d.pending_depth, G); ALLOC(d.pending_info, G);
    ALLOC(d.sims_done, G); ALLOC(d.n_alloc, G); ALLOC(d.state, G); ALLOC(d.need_root, G);
    ALLOC(d.ply, G); ALLOC(d.game_id, G);

    ALLOC(d.path, G * MAX_DEPTH);
    ALLOC(e->cold.cont_cur, G); ALLOC(e->cold.cont_n, G); ALLOC(e->cold.cont_w, G);
    ALLOC(d.stats, G * N_STATS);
    ALLOC(e->cold.res_move, G); ALLOC(e->cold.res_value, G); ALLOC(e->cold.res_policy, G * 7);
    ALLOC(e->cold.next_game, 1);
    const size_t R = (size_t)d.rec_cap;
    const size_t SG = R ? G : 0;   // staging rows only when games are recorded
    ALLOC(e->cold.stg_c0, SG * 42); ALLOC(e->cold.stg_c1, SG * 42); ALLOC(e->cold.stg_move, SG * 42);
    ALLOC(e->cold.stg_value, SG * 42); ALLOC(e->cold.stg_policy, SG * 42 * 7);
    ALLOC(e->cold.ring, R);
    ALLOC(e->cold.ring_head, 1); ALLOC(e->cold.ring_tail, 1); ALLOC(e->cold.dropped, 1);
    ALLOC(e->active_dev, 1);
    ALLOC(e->export_scratch, R + 4);
    // score tables (host libm so that log() is the very function Python's math.log calls)
    {
        const size_t nt = (size_t)cfg->simulations + 4;
        std::vector<double2> AB(nt);
        for (size_t n = 0; n < nt; ++n) {
            AB[n].x = std::log((double)((long long)n + cfg->pb_c_base + 1) / (double)cfg->pb_c_base) + cfg->pb_c_init;
            AB[n].y = std::sqrt((double)n);
        }
        double2 *tAB;
        ALLOC(tAB, nt);
        if (hipMemcpy(tAB, AB.data(), nt * sizeof(double2), hipMemcpyHostToDevice) != hipSuccess) { set_err(g_err, "table upload failed"); c4_engine_destroy(e); return C4_EDEVICE; }
        d.tabAB = tAB;
    }
    {   // evaluation cache: <0 off, 0 auto (self-play with a float32 evaluator only), else log2(entries)
        int bits = cfg->eval_cache_log2_entries;
        if (bits == 0 && cfg->eval_mode == C4_EVAL_EXTERNAL_F32 && !cfg->stop_after_move) {
            // The reference's table lives as long as its player and is shared by all its games
            // (evaluators.py:9-25), so positions of earlier games keep answering: size it for many
            // games' worth of evaluations -- 256 x slots x simulations entries (2^30 = 51.5 GB for the 4096-game
            // configuration; measured there with the fp16 net in round 2: 2^25 227 M, 2^26 238, 2^27 245, 2^28 249, 2^29 252,
            // 2^30 254 M expansions/s as the hit rate climbs from 74 % to 81 % and direct-mapped collisions thin out; with the
            // reference-precision net in round 3, where a miss costs twice as much: 2^28 239 M, 2^29 245 M, 2^30 248 M), at most
            // 2^30 entries and a quarter of the FREE device memory as it is after this engine's node pools are allocated (288 GB
            // of HBM is there to be used; ranks that share a card in rehearsals each take a quarter of what they find).
            const uint64_t want = 256ULL * (uint64_t)cfg->n_slots * ((uint64_t)cfg->simulations + 1);
            size_t free_b = 0, total_b = 0;
            if (hipMemGetInfo(&free_b, &total_b) != hipSuccess) free_b = (size_t)8 << 30;
            bits = 16;
            while (bits < 30 && (1ULL << bits) < want && (sizeof(CacheEntry) << (bits + 1)) <= free_b / 4) ++bits;
        }
        if (bits > 0 && cfg->eval_mode == C4_EVAL_EXTERNAL_F32) {
            if (bits < 8 || bits > 30) { set_err(g_err, "eval_cache_log2_entries=%d out of range [8,30]", bits); c4_engine_destroy(e); *out = nullptr; return C4_EINVAL; }
            CacheEntry *c = nullptr;
            if (dev_alloc(e, &c, (size_t)1 << bits) != C4_OK) { strncpy(g_err, e->err, 511); c4_engine_destroy(e); *out = nullptr; return C4_ENOMEM; }
            if (hipMemset(c, 0xFF, sizeof(CacheEntry) << bits) != hipSuccess) { set_err(g_err, "cache memset failed"); c4_engine_destroy(e); *out = nullptr; return C4_EDEVICE; }
            d.cache = c;
            d.cache_bits = bits;
        }
    }
    if (getenv("C4_TREE_STAMPS")) {
        unsigned long long *q = nullptr;
        if (dev_alloc(e, &q, 3 * 256 * 8) == C4_OK) { (void)hipMemset(q, 0, 3 * 256 * 8 * 8); e->cold.stamps = q; }
    }
    d.has_stamps = e->cold.stamps != nullptr;
    ALLOC(e->cold_dev, 1);
    ALLOC(e->d_dev, 1);
#undef ALLOC
    d.cold = e->cold_dev;
    if ((rc = sync_cold(e)) != C4_OK) { strncpy(g_err, e->err, 511); c4_engine_destroy(e); return rc; }
    *out = e;
    rc = c4_reset(e, nullptr, nullptr, cfg->n_slots);
    if (rc) { strncpy(g_err, e->err, 511); c4_engine_destroy(e); *out = nullptr; return rc; }
    return C4_OK;
}

int c4_engine_destroy(c4_engine *e)
{
    if (!e) return C4_OK;
    (void)hipSetDevice(e->device);
    (void)hipDeviceSynchronize();
    for (void *p : e->allocs) (void)hipFree(p);
    if (e->tape_noise) (void)hipFree(e->tape_noise);
    if (e->tape_u) (void)hipFree(e->tape_u);
    delete e;
    return C4_OK;
}

int c4_clear_eval_cache(c4_engine *e)
{
    if (!e) return C4_EINVAL;
    if (!e->d.cache) return C4_OK;
    HIPCHK(e, hipSetDevice(e->device));
    HIPCHK(e, hipDeviceSynchronize());
    HIPCHK(e, hipMemset(e->d.cache, 0xFF, sizeof(CacheEntry) << e->d.cache_bits));
    return C4_OK;
}

int c4_set_stream(c4_engine *e, void *hip_stream)
{
    if (!e) return C4_EINVAL;
    e->stream = (hipStream_t)hip_stream;
    return C4_OK;
}

int c4_reset(c4_engine *e, const uint64_t *color0, const uint64_t *color1, int32_t n_active)
{
    if (!e) return C4_EINVAL;
    if (n_active < 0 || n_active > e->d.G) { set_err(e->err, "n_active=%d out of range [0,%d]", n_active, e->d.G); return C4_EINVAL; }
    if ((color0 == nullptr) != (color1 == nullptr)) { set_err(e->err, "color0/color1 must both be given or both NULL"); return C4_EINVAL; }
    HIPCHK(e, hipSetDevice(e->device));
    HIPCHK(e, hipDeviceSynchronize());
    uint64_t *d0 = nullptr, *d1 = nullptr;
    if (color0) {
        for (int i = 0; i < n_active; ++i)
            if (position_status(color0[i], color1[i]) != ST_FRESH || (color0[i] & color1[i])) {
                set_err(e->err, "start position %d is decided or inconsistent (search on a finished board is an error in the reference too)", i);
                return C4_EINVAL;
            }
        HIPCHK(e, hipMalloc((void **)&d0, sizeof(uint64_t) * (size_t)e->d.G));
        HIPCHK(e, hipMalloc((void **)&d1, sizeof(uint64_t) * (size_t)e->d.G));
        HIPCHK(e, hipMemcpyAsync(d0, color0, sizeof(uint64_t) * (size_t)n_active, hipMemcpyHostToDevice, e->stream));
        HIPCHK(e, hipMemcpyAsync(d1, color1, sizeof(uint64_t) * (size_t)n_active, hipMemcpyHostToDevice, e->stream));
    }
    int na = n_active;
    if (e->d.games_target >= 0 && !e->d.stop_after_move && na > e->d.games_target) na = (int)e->d.games_target;
    hipLaunchKernelGGL(c4_reset_kernel, dim3((e->d.G + 255) / 256), dim3(256), 0, e->stream, e->d, d0, d1, na);
    HIPCHK(e, hipGetLastError());
    HIPCHK(e, hipStreamSynchronize(e->stream));
    if (d0) (void)hipFree(d0);
    if (d1) (void)hipFree(d1);
    e->launches = 0;
    return C4_OK;
}

int c4_set_tapes(c4_engine *e, const double *gamma_noise, const double *uniforms, int32_t n_games)
{
    if (!e || n_games <= 0 || !gamma_noise || !uniforms) { if (e) set_err(e->err, "c4_set_tapes: bad argument"); return C4_EINVAL; }
    if (!e->d.rng_tape) { set_err(e->err, "engine was not created with C4_RNG_TAPE"); return C4_ESTATE; }
    HIPCHK(e, hipSetDevice(e->device));
    HIPCHK(e, hipDeviceSynchronize());
    if (e->tape_noise) (void)hipFree(e->tape_noise);
    if (e->tape_u) (void)hipFree(e->tape_u);
    e->tape_noise = e->tape_u = nullptr;
    HIPCHK(e, hipMalloc((void **)&e->tape_noise, sizeof(double) * 42 * 7 * (size_t)n_games));
    HIPCHK(e, hipMalloc((void **)&e->tape_u, sizeof(double) * 42 * (size_t)n_games));
    HIPCHK(e, hipMemcpy(e->tape_noise, gamma_noise, sizeof(double) * 42 * 7 * (size_t)n_games, hipMemcpyHostToDevice));
    HIPCHK(e, hipMemcpy(e->tape_u, uniforms, sizeof(double) * 42 * (size_t)n_games, hipMemcpyHostToDevice));
    e->cold.noise_tape = e->tape_noise;
    e->cold.u_tape = e->tape_u;
    e->cold.tape_games = n_games;
    return sync_cold(e);
}

int c4_step_range(c4_engine *e, const void *values_dev, const void *priors_dev, void *planes_dev, int32_t slot_lo,
                  int32_t slot_count, void *hip_stream)
{
    if (!e) return C4_EINVAL;
    if (slot_lo < 0 || slot_count <= 0 || slot_lo + slot_count > e->d.G || (slot_lo % SLOTS_PER_BLOCK)) { set_err(e->err, "c4_step_range: bad slot range [%d,+%d) (start must be a multiple of %d)", slot_lo, slot_count, SLOTS_PER_BLOCK); return C4_EINVAL; }
    if (e->d.rng_tape && (e->d.use_noise || e->d.nsm > 0) && !e->cold.noise_tape) { set_err(e->err, "C4_RNG_TAPE engine needs c4_set_tapes before stepping"); return C4_ESTATE; }
    if (e->cfg.eval_mode != C4_EVAL_CENTRE && e->launches > 0 && (!values_dev || !priors_dev)) { set_err(e->err, "c4_step: values/priors are required after the first step"); return C4_EINVAL; }
    Dev d = e->d;
    d.slot_lo = slot_lo;
    d.slot_hi = slot_lo + slot_count;
    hipStream_t st = hip_stream ? (hipStream_t)hip_stream : e->stream;
    const dim3 grid((slot_count + SLOTS_PER_BLOCK - 1) / SLOTS_PER_BLOCK), block(BLOCK);
    switch (e->cfg.eval_mode) {
    case C4_EVAL_EXTERNAL_F32:
        hipLaunchKernelGGL(c4_step_kernel<C4_EVAL_EXTERNAL_F32>, grid, block, 0, st, d, values_dev, priors_dev, planes_dev);
        break;
    case C4_EVAL_EXTERNAL_F64:
        hipLaunchKernelGGL(c4_step_kernel<C4_EVAL_EXTERNAL_F64>, grid, block, 0, st, d, values_dev, priors_dev, planes_dev);
        break;
    default:
        hipLaunchKernelGGL(c4_step_kernel<C4_EVAL_CENTRE>, grid, block, 0, st, d, values_dev, priors_dev, planes_dev);
        break;
    }
    HIPCHK(e, hipGetLastError());
    e->launches += 1;
    return C4_OK;
}

int c4_selfplay_steps(c4_engine *e, c4_net *net, float *values_dev, float *priors_dev, int32_t n_steps, void *hip_stream)
{
    if (!e || !net || !values_dev || !priors_dev || n_steps <= 0) { if (e) set_err(e->err, "c4_selfplay_steps: bad argument"); return C4_EINVAL; }
    if (e->cfg.eval_mode != C4_EVAL_EXTERNAL_F32) { set_err(e->err, "c4_selfplay_steps needs C4_EVAL_EXTERNAL_F32"); return C4_ESTATE; }
    if (net->device != e->device) { set_err(e->err, "engine and net live on different devices"); return C4_EINVAL; }
    if (e->d.rng_tape && (e->d.use_noise || e->d.nsm > 0) && !e->cold.noise_tape) { set_err(e->err, "C4_RNG_TAPE engine needs c4_set_tapes before stepping"); return C4_ESTATE; }
    c4net::NetDev nd = net->d;
    nd.stamps = nullptr;
    hipStream_t st = hip_stream ? (hipStream_t)hip_stream : e->stream;
    // 32 slots per workgroup once that still gives every CU a workgroup (the tree phase is latency bound,
    // so its cost is shared by twice the slots); smaller batches keep 16 so that no CU stays idle
    if (e->fused_wave) {   // wave-autonomous variant; it reads the engine description from device memory
        if (memcmp(&e->d_uploaded, &e->d, sizeof(Dev)) != 0) {
            HIPCHK(e, hipMemcpyAsync(e->d_dev, &e->d, sizeof(Dev), hipMemcpyHostToDevice, st));
            HIPCHK(e, hipStreamSynchronize(st));   // the source is a member that may change right after this call
            e->d_uploaded = e->d;
        }
        const dim3 g32((e->d.G + 31) / 32), g16((e->d.G + 15) / 16), blk(c4net::NTHREADS);
        // tree waves of the split kernel: 4 (one per SIMD) for the 32-filter nets (f32x3: 230 M expansions/s against 222 M with
        // 2 tree waves); the 64-filter forward is slow enough to want more network waves: 2 tree waves of 8 slots and 6
        // network waves at 16 slots per CU (111 against 104 M)
        const int tw = e->split_tw ? e->split_tw : (nd.mode == c4net::NETMODE_F64 ? 2 : 4);
        // the split kernel spreads a batch that would leave CUs without a workgroup over all of them (slot = workgroup + p x workgroups)
        const int dense_wgs = (e->d.G + e->fused_slots - 1) / e->fused_slots;
        const int spread = (!e->pack_dense && dense_wgs < e->cus && e->d.G > dense_wgs) ? 1 : 0;
        const dim3 gsp(spread ? std::min(e->cus, e->d.G) : dense_wgs);
#define C4_LAUNCH_SPLIT(TSV, MODE, TWV) hipLaunchKernelGGL((c4_selfplay_split_kernel<TSV, MODE, TWV>), gsp, blk, 0, st, e->d_dev, nd, values_dev, priors_dev, (int)n_steps, spread)
#define C4_LAUNCH_WAVE(MODE)                                                                                                       \
    do {                                                                                                                           \
        if (e->fused_wave == 2) {                                                                                                  \
            if (e->fused_slots == 32) C4_LAUNCH_SPLIT(32, MODE, 4);   /* (a wave holds at most 8 slots: no fewer than 4 tree waves) */ \
            else if (tw == 2) C4_LAUNCH_SPLIT(16, MODE, 2);                                                                        \
            else if (tw == 3) C4_LAUNCH_SPLIT(16, MODE, 3);                                                                        \
            else C4_LAUNCH_SPLIT(16, MODE, 4);                                                                                     \
        } else if (e->fused_slots == 32) hipLaunchKernelGGL((c4_selfplay_wave_kernel<32, MODE>), g32, blk, 0, st, e->d_dev, nd, values_dev, priors_dev, (int)n_steps); \
        else hipLaunchKernelGGL((c4_selfplay_wave_kernel<16, MODE>), g16, blk, 0, st, e->d_dev, nd, values_dev, priors_dev, (int)n_steps); \
    } while (0)
        if (nd.mode == c4net::NETMODE_F64) C4_LAUNCH_WAVE(c4net::NETMODE_F64);
        else if (nd.mode == c4net::NETMODE_F32_PRECISE) C4_LAUNCH_WAVE(c4net::NETMODE_F32_PRECISE);
        else C4_LAUNCH_WAVE(c4net::NETMODE_F32_F16);
#undef C4_LAUNCH_WAVE
#undef C4_LAUNCH_SPLIT
    } else if (nd.mode != c4net::NETMODE_F32_F16) {
        set_err(e->err, "C4_FUSED_MODE=block serves only the 32-filter fp16 net; use the default wave-autonomous kernel");
        return C4_ESTATE;
    } else if (e->fused_slots == 32)
        hipLaunchKernelGGL(c4_selfplay_kernel<32>, dim3((e->d.G + 31) / 32), dim3(c4net::NTHREADS), 0, st, e->d, nd, values_dev, priors_dev, (int)n_steps);
    else
        hipLaunchKernelGGL(c4_selfplay_kernel<16>, dim3((e->d.G + 15) / 16), dim3(c4net::NTHREADS), 0, st, e->d, nd, values_dev, priors_dev, (int)n_steps);
    HIPCHK(e, hipGetLastError());
    e->launches += n_steps;
    return C4_OK;
}

int c4_step(c4_engine *e, const void *values_dev, const void *priors_dev, void *planes_dev)
{
    if (!e) return C4_EINVAL;
    return c4_step_range(e, values_dev, priors_dev, planes_dev, 0, e->d.G, nullptr);
}

int c4_get_stats(c4_engine *e, c4_stats *out)
{
    if (!e || !out) return C4_EINVAL;
    HIPCHK(e, hipSetDevice(e->device));
    HIPCHK(e, hipDeviceSynchronize());   // launches may sit on any stream the caller passed
    const size_t G = (size_t)e->d.G;
    std::vector<uint64_t> s(G * N_STATS);
    std::vector<int32_t> stt(G);
    HIPCHK(e, hipMemcpy(s.data(), e->d.stats, s.size() * sizeof(uint64_t), hipMemcpyDeviceToHost));
    HIPCHK(e, hipMemcpy(stt.data(), e->d.state, G * sizeof(int32_t), hipMemcpyDeviceToHost));
    SlotStats t = {};
    uint64_t *tp = (uint64_t *)&t;
    for (size_t g = 0; g < G; ++g)
        for (int i = 0; i < N_STATS; ++i) tp[i] += s[g * N_STATS + i];
    memset(out, 0, sizeof(*out));
    out->simulations = (int64_t)t.sims;
    out->expansions = (int64_t)t.expansions;
    out->children_created = (int64_t)t.children;
    out->terminal_sims = (int64_t)t.terminal_sims;
    out->leaf_evals = (int64_t)t.leaf_evals;
    out->depth_sum = (int64_t)t.depth_sum;
    out->moves = (int64_t)t.moves;
    out->games_started = (int64_t)t.games_started;
    out->games_finished = (int64_t)t.games_finished;
    out->capped_slots = (int64_t)t.capped;
    out->speculative_evals = (int64_t)t.spec_evals;
    out->eval_cache_hits = (int64_t)t.cache_hits;
    out->eval_cache_probes = (int64_t)t.cache_probes;
    out->bad_evals = (int64_t)t.bad_evals;
    out->launches = e->launches;
    if (e->cold.dropped) {
        unsigned long long dr = 0;
        HIPCHK(e, hipMemcpy(&dr, e->cold.dropped, sizeof(dr), hipMemcpyDeviceToHost));
        out->dropped_games = (int64_t)dr;
    }
    for (size_t g = 0; g < G; ++g) out->active_slots += stt[g] == SLOT_ACTIVE;
    return C4_OK;
}

int c4_run_centre(c4_engine *e, int32_t max_launches)
{
    if (!e) return C4_EINVAL;
    if (e->cfg.eval_mode != C4_EVAL_CENTRE) { set_err(e->err, "c4_run_centre needs C4_EVAL_CENTRE"); return C4_ESTATE; }
    for (int i = 0; i < max_launches; ++i) {
        int rc = c4_step(e, nullptr, nullptr, nullptr);
        if (rc) return rc;
        // four bytes back per launch: how many slots are still searching
        hipLaunchKernelGGL(k_count_active, dim3(1), dim3(1024), 0, e->stream, e->d.state, e->d.G, e->active_dev);
        HIPCHK(e, hipGetLastError());
        int32_t active = 0;
        HIPCHK(e, hipMemcpyAsync(&active, e->active_dev, sizeof(active), hipMemcpyDeviceToHost, e->stream));
        HIPCHK(e, hipStreamSynchronize(e->stream));
        if (active == 0) return C4_OK;
    }
    return C4_OK;
}

int c4_leaf_buffers(c4_engine *e, const uint64_t **c0, const uint64_t **c1, const int32_t **has_leaf)
{
    if (!e) return C4_EINVAL;
    if (c0) *c0 = e->d.leaf_c0;
    if (c1) *c1 = e->d.leaf_c1;
    if (has_leaf) *has_leaf = e->d.has_leaf;
    return C4_OK;
}

int c4_read_leaves(c4_engine *e, uint64_t *c0, uint64_t *c1, int32_t *has_leaf)
{
    if (!e || !c0 || !c1 || !has_leaf) return C4_EINVAL;
    HIPCHK(e, hipSetDevice(e->device));
    HIPCHK(e, hipDeviceSynchronize());
    const size_t G = (size_t)e->d.G;
    HIPCHK(e, hipMemcpy(c0, e->d.leaf_c0, G * sizeof(uint64_t), hipMemcpyDeviceToHost));
    HIPCHK(e, hipMemcpy(c1, e->d.leaf_c1, G * sizeof(uint64_t), hipMemcpyDeviceToHost));
    HIPCHK(e, hipMemcpy(has_leaf, e->d.has_leaf, G * sizeof(int32_t), hipMemcpyDeviceToHost));
    return C4_OK;
}

int c4_read_roots(c4_engine *e, c4_root_result *out)
{
    if (!e || !out) return C4_EINVAL;
    HIPCHK(e, hipSetDevice(e->device));
    c4_root_result *dout = nullptr;
    const size_t G = (size_t)e->d.G;
    HIPCHK(e, hipDeviceSynchronize());
    HIPCHK(e, hipMalloc((void **)&dout, G * sizeof(c4_root_result)));
    hipLaunchKernelGGL(c4_gather_roots_kernel, dim3((e->d.G + 63) / 64), dim3(64), 0, e->stream, e->d, dout);
    hipError_t r = hipGetLastError();
    if (r == hipSuccess) r = hipDeviceSynchronize();
    if (r == hipSuccess) r = hipMemcpy(out, dout, G * sizeof(c4_root_result), hipMemcpyDeviceToHost);
    (void)hipFree(dout);
    HIPCHK(e, r);
    return C4_OK;
}

// ---------------------------------------------------------------- pure board functions
#define BOARD_PROLOGUE()                                   \
    if (n < 0) { set_err(g_err, "n < 0"); return C4_EINVAL; } \
    int rc = check_device(device, g_err);                  \
    if (rc) return rc;                                     \
    if (n == 0) return C4_OK;                              \
    hipError_t r = hipSuccess;                             \
    Scratch sc;

#define BOARD_EPILOGUE()                                                                    \
    if (r != hipSuccess) { set_err(g_err, "HIP failure: %s", hipGetErrorString(r)); return C4_EDEVICE; } \
    return C4_OK;

namespace {
int ring_counters(c4_engine *e, unsigned long long &head, unsigned long long &tail)
{
    HIPCHK(e, hipMemcpy(&head, e->cold.ring_head, sizeof(head), hipMemcpyDeviceToHost));
    HIPCHK(e, hipMemcpy(&tail, e->cold.ring_tail, sizeof(tail), hipMemcpyDeviceToHost));
    return C4_OK;
}
}  // namespace

int c4_finished_games(c4_engine *e, int64_t *n_ready, int64_t *n_dropped)
{
    if (!e) return C4_EINVAL;
    if (n_ready) *n_ready = 0;
    if (n_dropped) *n_dropped = 0;
    if (e->d.rec_cap == 0) return C4_OK;
    HIPCHK(e, hipSetDevice(e->device));
    HIPCHK(e, hipDeviceSynchronize());
    unsigned long long head = 0, tail = 0, dr = 0;
    int rc = ring_counters(e, head, tail);
    if (rc) return rc;
    HIPCHK(e, hipMemcpy(&dr, e->cold.dropped, sizeof(dr), hipMemcpyDeviceToHost));
    if (n_ready) *n_ready = (int64_t)(head - tail);
    if (n_dropped) *n_dropped = (int64_t)dr;
    return C4_OK;
}

int c4_drain_games(c4_engine *e, c4_game_record *out, int32_t cap, int32_t *n_out)
{
    if (!e || !out || !n_out || cap < 0) return C4_EINVAL;
    *n_out = 0;
    const unsigned long long R = (unsigned long long)e->d.rec_cap;
    if (R == 0) return C4_OK;
    HIPCHK(e, hipSetDevice(e->device));
    HIPCHK(e, hipDeviceSynchronize());
    unsigned long long head = 0, tail = 0;
    int rc = ring_counters(e, head, tail);
    if (rc) return rc;
    unsigned long long n = head - tail;
    if (n > (unsigned long long)cap) n = (unsigned long long)cap;
    if (n == 0) return C4_OK;
    // the rows [tail, tail+n) of the ring, at most two contiguous pieces: one or two copies for ALL games
    const unsigned long long first = tail % R;
    const unsigned long long n1 = std::min(n, R - first);
    HIPCHK(e, hipMemcpy(out, e->cold.ring + first, n1 * sizeof(c4_game_record), hipMemcpyDeviceToHost));
    if (n > n1) HIPCHK(e, hipMemcpy(out + n1, e->cold.ring, (n - n1) * sizeof(c4_game_record), hipMemcpyDeviceToHost));
    tail += n;
    HIPCHK(e, hipMemcpy(e->cold.ring_tail, &tail, sizeof(tail), hipMemcpyHostToDevice));
    std::sort(out, out + n, [](const c4_game_record &a, const c4_game_record &b) { return a.game_id < b.game_id; });
    *n_out = (int32_t)n;
    return C4_OK;
}

int c4_export_games_dev(c4_engine *e, const c4_export_buffers *bufs, int32_t max_games, int64_t cap_positions, int64_t *counts_dev,
                        void *hip_stream)
{
    if (!e || !bufs || max_games < 0 || cap_positions < 0) { if (e) set_err(e->err, "c4_export_games_dev: bad argument"); return C4_EINVAL; }
    if (e->d.rec_cap == 0) { set_err(e->err, "engine keeps no game records (stop_after_move)"); return C4_ESTATE; }
    if (max_games > e->d.rec_cap) max_games = e->d.rec_cap;
    hipStream_t st = hip_stream ? (hipStream_t)hip_stream : e->stream;
    hipLaunchKernelGGL(k_export_scan, dim3(1), dim3(1024), 0, st, (const Cold *)e->cold_dev, e->d.rec_cap, (int)max_games,
                       (long long)cap_positions, e->export_scratch);
    HIPCHK(e, hipGetLastError());
    const int blocks = std::max(1, std::min((int)max_games, 65536));
    hipLaunchKernelGGL(k_export_write, dim3(blocks), dim3(64), 0, st, (const Cold *)e->cold_dev, e->d.rec_cap,
                       (const int64_t *)e->export_scratch, *bufs);
    HIPCHK(e, hipGetLastError());
    hipLaunchKernelGGL(k_export_commit, dim3(1), dim3(1), 0, st, (const Cold *)e->cold_dev, (const int64_t *)e->export_scratch, counts_dev);
    HIPCHK(e, hipGetLastError());
    return C4_OK;
}

int c4_training_tensors_dev(int device, void *hip_stream, const int64_t *boards_dev, const float *targets_dev, const float *policy_dev,
                            int64_t n, int32_t add_fliplr, float *out_boards_dev, float *out_values_dev, float *out_priors_dev)
{
    if (n < 0 || !boards_dev || !targets_dev || !policy_dev || !out_boards_dev || !out_values_dev || !out_priors_dev) { set_err(g_err, "c4_training_tensors_dev: bad argument"); return C4_EINVAL; }
    int rc = check_device(device, g_err);
    if (rc) return rc;
    if (n == 0) return C4_OK;
    const long long tot = (long long)n * (add_fliplr ? 2 : 1) * 126;
    hipLaunchKernelGGL(k_training_tensors, dim3((unsigned)((tot + 255) / 256)), dim3(256), 0, (hipStream_t)hip_stream, boards_dev, targets_dev,
                       policy_dev, (long long)n, (int)add_fliplr, out_boards_dev, out_values_dev, out_priors_dev);
    hipError_t r = hipGetLastError();
    if (r != hipSuccess) { set_err(g_err, "k_training_tensors launch failed: %s", hipGetErrorString(r)); return C4_EDEVICE; }
    return C4_OK;
}

int c4_eval_cache_lookup(c4_engine *e, const uint64_t *color0, const uint64_t *color1, int32_t n, float *value, float *prior, int32_t *found)
{
    if (!e || n < 0 || !color0 || !color1 || !value || !prior || !found) { if (e) set_err(e->err, "c4_eval_cache_lookup: bad argument"); return C4_EINVAL; }
    if (!e->d.cache) { set_err(e->err, "this engine has no evaluation cache"); return C4_ESTATE; }
    if (n == 0) return C4_OK;
    HIPCHK(e, hipSetDevice(e->device));
    HIPCHK(e, hipDeviceSynchronize());
    hipError_t r = hipSuccess;
    Scratch sc;
    uint64_t *d0 = sc.up(color0, n, r), *d1 = sc.up(color1, n, r);
    float *dv = sc.up<float>(nullptr, n, r), *dp = sc.up<float>(nullptr, (size_t)n * 7, r);
    int32_t *df = sc.up<int32_t>(nullptr, n, r);
    if (r == hipSuccess) {
        hipLaunchKernelGGL(k_cache_lookup, dim3((n * GROUP + 63) / 64), dim3(64), 0, 0, e->d, d0, d1, (int)n, dv, dp, df);
        r = hipGetLastError();
    }
    if (r == hipSuccess) r = hipMemcpy(value, dv, sizeof(float) * n, hipMemcpyDeviceToHost);
    if (r == hipSuccess) r = hipMemcpy(prior, dp, sizeof(float) * (size_t)n * 7, hipMemcpyDeviceToHost);
    if (r == hipSuccess) r = hipMemcpy(found, df, sizeof(int32_t) * n, hipMemcpyDeviceToHost);
    HIPCHK(e, r);
    return C4_OK;
}

int c4_debug_root_noise(int device, uint64_t seed, double alpha, const int64_t *game_id, const int32_t *ply, const int32_t *legal_mask,
                        int32_t n, double *gamma_raw, double *dirichlet)
{
    if (!game_id || !ply || !legal_mask || !gamma_raw || !dirichlet || !(alpha > 0.0)) { set_err(g_err, "c4_debug_root_noise: bad argument"); return C4_EINVAL; }
    BOARD_PROLOGUE();
    long long *dg = (long long *)sc.up((const int64_t *)game_id, n, r);
    int32_t *dp = sc.up(ply, n, r), *dm = sc.up(legal_mask, n, r);
    double *draw = sc.up<double>(nullptr, (size_t)n * 7, r), *ddir = sc.up<double>(nullptr, (size_t)n * 7, r);
    if (r == hipSuccess) {
        hipLaunchKernelGGL(k_debug_noise, dim3((n * GROUP + 63) / 64), dim3(64), 0, 0, seed, alpha, dg, dp, dm, (int)n, draw, ddir);
        r = hipGetLastError();
    }
    if (r == hipSuccess) r = hipMemcpy(gamma_raw, draw, sizeof(double) * (size_t)n * 7, hipMemcpyDeviceToHost);
    if (r == hipSuccess) r = hipMemcpy(dirichlet, ddir, sizeof(double) * (size_t)n * 7, hipMemcpyDeviceToHost);
    BOARD_EPILOGUE();
}

int c4_debug_sample_move(int device, uint64_t seed, const int64_t *game_id, const int32_t *ply, const double *child_values,
                         const int32_t *n_children, const double *uniforms, int32_t n, double *uniform_out, int32_t *choice_out)
{
    if (!game_id || !ply || !child_values || !n_children || !uniform_out || !choice_out) { set_err(g_err, "c4_debug_sample_move: bad argument"); return C4_EINVAL; }
    BOARD_PROLOGUE();
    for (int i = 0; i < n; ++i)
        if (n_children[i] < 1 || n_children[i] > 7) { set_err(g_err, "n_children[%d]=%d out of range", i, n_children[i]); return C4_EINVAL; }
    long long *dg = (long long *)sc.up((const int64_t *)game_id, n, r);
    int32_t *dp = sc.up(ply, n, r), *dn = sc.up(n_children, n, r);
    double *dv = sc.up(child_values, (size_t)n * 7, r);
    double *du_in = uniforms ? sc.up(uniforms, n, r) : nullptr;
    double *du = sc.up<double>(nullptr, n, r);
    int32_t *dc = sc.up<int32_t>(nullptr, n, r);
    if (r == hipSuccess) {
        hipLaunchKernelGGL(k_debug_sample, dim3((n * GROUP + 63) / 64), dim3(64), 0, 0, seed, dg, dp, dv, dn, du_in, (int)n, du, dc);
        r = hipGetLastError();
    }
    if (r == hipSuccess) r = hipMemcpy(uniform_out, du, sizeof(double) * n, hipMemcpyDeviceToHost);
    if (r == hipSuccess) r = hipMemcpy(choice_out, dc, sizeof(int32_t) * n, hipMemcpyDeviceToHost);
    BOARD_EPILOGUE();
}

int c4_debug_div_mismatches(int device, int32_t max_parent_visits, int32_t max_child_visits, int64_t n_random, int64_t *mismatches)
{
    if (!mismatches || max_parent_visits <= 0 || max_child_visits <= 0 || n_random < 0) { set_err(g_err, "c4_debug_div_mismatches: bad argument"); return C4_EINVAL; }
    const int n = 1;
    BOARD_PROLOGUE();
    unsigned long long *dm = sc.up<unsigned long long>(nullptr, 1, r);
    if (r == hipSuccess) r = hipMemset(dm, 0, sizeof(unsigned long long));
    if (r == hipSuccess) {
        hipLaunchKernelGGL(k_debug_div, dim3(4096), dim3(256), 0, 0, (int)max_parent_visits, (int)max_child_visits, (unsigned long long)n_random, dm);
        r = hipGetLastError();
    }
    unsigned long long out = 0;
    if (r == hipSuccess) r = hipMemcpy(&out, dm, sizeof(out), hipMemcpyDeviceToHost);
    *mismatches = (int64_t)out;
    BOARD_EPILOGUE();
}

int c4_board_make_move(int device, const uint64_t *c0, const uint64_t *c1, const int32_t *col, int32_t n,
                       uint64_t *o0, uint64_t *o1, int32_t *result)
{
    if (!c0 || !c1 || !col || !o0 || !o1 || !result) { set_err(g_err, "null argument"); return C4_EINVAL; }
    BOARD_PROLOGUE();
    for (int i = 0; i < n; ++i)
        if (col[i] < 0 || col[i] >= 7) { set_err(g_err, "column %d out of range at %d", col[i], i); return C4_EINVAL; }
    uint64_t *d0 = sc.up(c0, n, r), *d1 = sc.up(c1, n, r);
    int32_t *dc = sc.up(col, n, r);
    uint64_t *e0 = sc.up<uint64_t>(nullptr, n, r), *e1 = sc.up<uint64_t>(nullptr, n, r);
    int32_t *dr = sc.up<int32_t>(nullptr, n, r);
    if (r == hipSuccess) {
        hipLaunchKernelGGL(k_make_move, dim3((n + 255) / 256), dim3(256), 0, 0, d0, d1, dc, n, e0, e1, dr);
        r = hipGetLastError();
    }
    if (r == hipSuccess) r = hipMemcpy(o0, e0, sizeof(uint64_t) * n, hipMemcpyDeviceToHost);
    if (r == hipSuccess) r = hipMemcpy(o1, e1, sizeof(uint64_t) * n, hipMemcpyDeviceToHost);
    if (r == hipSuccess) r = hipMemcpy(result, dr, sizeof(int32_t) * n, hipMemcpyDeviceToHost);
    BOARD_EPILOGUE();
}

int c4_board_wins(int device, const uint64_t *stones, int32_t n, int32_t *out)
{
    if (!stones || !out) { set_err(g_err, "null argument"); return C4_EINVAL; }
    BOARD_PROLOGUE();
    uint64_t *ds = sc.up(stones, n, r);
    int32_t *dout = sc.up<int32_t>(nullptr, n, r);
    if (r == hipSuccess) {
        hipLaunchKernelGGL(k_wins, dim3((n + 255) / 256), dim3(256), 0, 0, ds, n, dout);
        r = hipGetLastError();
    }
    if (r == hipSuccess) r = hipMemcpy(out, dout, sizeof(int32_t) * n, hipMemcpyDeviceToHost);
    BOARD_EPILOGUE();
}

int c4_board_valid_mask(int device, const uint64_t *c0, const uint64_t *c1, int32_t n, int32_t *out)
{
    if (!c0 || !c1 || !out) { set_err(g_err, "null argument"); return C4_EINVAL; }
    BOARD_PROLOGUE();
    uint64_t *d0 = sc.up(c0, n, r), *d1 = sc.up(c1, n, r);
    int32_t *dout = sc.up<int32_t>(nullptr, n, r);
    if (r == hipSuccess) {
        hipLaunchKernelGGL(k_valid_mask, dim3((n + 255) / 256), dim3(256), 0, 0, d0, d1, n, dout);
        r = hipGetLastError();
    }
    if (r == hipSuccess) r = hipMemcpy(out, dout, sizeof(int32_t) * n, hipMemcpyDeviceToHost);
    BOARD_EPILOGUE();
}

int c4_board_planes(int device, const uint64_t *c0, const uint64_t *c1, int32_t n, float *out)
{
    if (!c0 || !c1 || !out) { set_err(g_err, "null argument"); return C4_EINVAL; }
    BOARD_PROLOGUE();
    uint64_t *d0 = sc.up(c0, n, r), *d1 = sc.up(c1, n, r);
    float *dout = sc.up<float>(nullptr, (size_t)n * 126, r);
    if (r == hipSuccess) {
        const size_t tot = (size_t)n * 126;
        hipLaunchKernelGGL(k_planes, dim3((unsigned)((tot + 255) / 256)), dim3(256), 0, 0, d0, d1, n, dout);
        r = hipGetLastError();
    }
    if (r == hipSuccess) r = hipMemcpy(out, dout, sizeof(float) * (size_t)n * 126, hipMemcpyDeviceToHost);
    BOARD_EPILOGUE();
}

int c4_board_fliplr(int device, const uint64_t *c0, const uint64_t *c1, int32_t n, uint64_t *o0, uint64_t *o1)
{
    if (!c0 || !c1 || !o0 || !o1) { set_err(g_err, "null argument"); return C4_EINVAL; }
    BOARD_PROLOGUE();
    uint64_t *d0 = sc.up(c0, n, r), *d1 = sc.up(c1, n, r);
    uint64_t *e0 = sc.up<uint64_t>(nullptr, n, r), *e1 = sc.up<uint64_t>(nullptr, n, r);
    if (r == hipSuccess) {
        hipLaunchKernelGGL(k_fliplr, dim3((n + 255) / 256), dim3(256), 0, 0, d0, d1, n, e0, e1);
        r = hipGetLastError();
    }
    if (r == hipSuccess) r = hipMemcpy(o0, e0, sizeof(uint64_t) * n, hipMemcpyDeviceToHost);
    if (r == hipSuccess) r = hipMemcpy(o1, e1, sizeof(uint64_t) * n, hipMemcpyDeviceToHost);
    BOARD_EPILOGUE();
}

int c4_board_centre_value(int device, const uint64_t *c0, const uint64_t *c1, int32_t n, double *out)
{
    if (!c0 || !c1 || !out) { set_err(g_err, "null argument"); return C4_EINVAL; }
    BOARD_PROLOGUE();
    uint64_t *d0 = sc.up(c0, n, r), *d1 = sc.up(c1, n, r);
    double *dout = sc.up<double>(nullptr, n, r);
    if (r == hipSuccess) {
        hipLaunchKernelGGL(k_centre, dim3((n + 255) / 256), dim3(256), 0, 0, d0, d1, n, dout);
        r = hipGetLastError();
    }
    if (r == hipSuccess) r = hipMemcpy(out, dout, sizeof(double) * n, hipMemcpyDeviceToHost);
    BOARD_EPILOGUE();
}

}  // extern "C"
