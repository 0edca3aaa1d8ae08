/*
 * c4_oracle.c -- CPU restatement of the oinkoink self-play/MCTS hot path (see c4_oracle.h).
 * TEST INFRASTRUCTURE ONLY: never linked or called by the product (connect4_amd/).
 *
 * Floating-point policy: the reference is Python floats (IEEE double) and NumPy arrays.  All
 * arithmetic below is written so that each Python/NumPy operation is one IEEE operation in the
 * same precision and order (compile with -ffp-contract=off, no -ffast-math).  NumPy facts used
 * (probed with numpy 2.2.6 in the build container, see DESIGN.md "FP semantics"):
 *   - np.sum over <=7 contiguous elements is a left-to-right sequential sum in the array dtype;
 *   - python_float (op) np.float32 scalar  ->  float32(python_float) (op) x  in float32 (NEP 50);
 *   - float32 ndarray (op) float64 ndarray ->  float64.
 */
#include "c4_oracle.h"

#include <math.h>
#include <stdlib.h>
#include <string.h>
#ifdef _OPENMP
#include <omp.h>
#endif

/* ------------------------------------------------------------------ board.py:9-32 constants */
#define WIDTH 7
#define HEIGHT 6
#define H1 (HEIGHT + 1)
#define H2 (HEIGHT + 2)
#define SIZE (HEIGHT * WIDTH)
#define COL1 ((uint64_t)0x7f)
#define BOTTOM ((uint64_t)0x40810204081ULL) /* bits i*H1 */
#define TOP (BOTTOM << HEIGHT)
#define SHIFT ((WIDTH - 1) * H1)

static int popcnt64(uint64_t x) { return __builtin_popcountll(x); }

/* board.py:39 height[i] = H1*i + stones already in column i */
static int col_height_bit(const c4o_board *b, int col)
{
    uint64_t occ = b->color[0] | b->color[1];
    return H1 * col + popcnt64((occ >> (H1 * col)) & COL1);
}

void c4o_board_init(c4o_board *b)
{
    b->color[0] = 0;
    b->color[1] = 0;
    b->age = 0;
    b->result = C4O_NONE;
}

/* board.py:173-184 _check_terminal_position */
int c4o_wins(uint64_t nb)
{
    uint64_t y = nb & (nb >> HEIGHT);
    if (y & (y >> (2 * HEIGHT))) return 1; /* diagonal \ */
    y = nb & (nb >> H1);
    if (y & (y >> (2 * H1))) return 1;     /* horizontal */
    y = nb & (nb >> H2);
    if (y & (y >> (2 * H2))) return 1;     /* diagonal / */
    y = nb & (nb >> 1);
    return (y & (y >> 2)) != 0;            /* vertical */
}

/* board.py:56-62 : result rules of from_pieces applied to raw bitboards */
int c4o_board_from_bits(c4o_board *b, uint64_t c0, uint64_t c1)
{
    b->color[0] = c0;
    b->color[1] = c1;
    b->age = popcnt64(c0) + popcnt64(c1);
    if (c4o_wins(c0)) b->result = C4O_OWIN;
    else if (c4o_wins(c1)) b->result = C4O_XWIN;
    else if (b->age == SIZE) b->result = C4O_DRAW;
    else b->result = C4O_NONE;
    return b->result;
}

/* board.py:44-62 from_pieces; pieces are [row][col] with row 0 = TOP of the board (BITMASK,
 * board.py:21-30: bit index col*7 + (5-row)). */
int c4o_board_from_pieces(c4o_board *b, const uint8_t o[42], const uint8_t x[42])
{
    uint64_t c0 = 0, c1 = 0;
    for (int r = 0; r < HEIGHT; ++r)
        for (int c = 0; c < WIDTH; ++c) {
            uint64_t bit = (uint64_t)1 << (c * H1 + (HEIGHT - 1 - r));
            if (o[r * WIDTH + c]) c0 |= bit;
            if (x[r * WIDTH + c]) c1 |= bit;
        }
    return c4o_board_from_bits(b, c0, c1);
}

/* board.py:160-170 make_move (no legality check, like the reference) */
int c4o_make_move(c4o_board *b, int col)
{
    int side = b->age & 1;
    b->color[side] ^= (uint64_t)1 << col_height_bit(b, col);
    int winner = c4o_wins(b->color[side]);
    b->age += 1;
    if (winner) b->result = (b->age % 2) ? C4O_OWIN : C4O_XWIN; /* Result(age % 2): 1.0 / 0.0 */
    else if (b->age == SIZE) b->result = C4O_DRAW;
    return b->result;
}

/* board.py:88-92 valid_moves + :187-188 _isplayable.  Python precedence: (x | bit) & TOP == 0
 * parses as ((x | bit) & TOP) == 0. */
int c4o_valid_mask(const c4o_board *b)
{
    if (b->result != C4O_NONE) return 0;
    int mask = 0;
    for (int c = 0; c < WIDTH; ++c) {
        uint64_t probe = b->color[b->age & 1] | ((uint64_t)1 << col_height_bit(b, c));
        if ((probe & TOP) == 0) mask |= 1 << c;
    }
    return mask;
}

/* board.py:64-78 o_pieces/x_pieces (flipud => row 0 is the top) and :147-154 to_array:
 * channel 0 = ones iff o to move (age even), channel 1 = o stones, channel 2 = x stones. */
void c4o_planes(const c4o_board *b, uint8_t out[126])
{
    uint8_t to_move = (b->age % 2 == 0) ? 1 : 0;
    for (int r = 0; r < HEIGHT; ++r)
        for (int c = 0; c < WIDTH; ++c) {
            int bit = c * H1 + (HEIGHT - 1 - r);
            out[0 * SIZE + r * WIDTH + c] = to_move;
            out[1 * SIZE + r * WIDTH + c] = (uint8_t)((b->color[0] >> bit) & 1);
            out[2 * SIZE + r * WIDTH + c] = (uint8_t)((b->color[1] >> bit) & 1);
        }
}

/* board.py:128-145 flip_color: mirror columns left<->right */
uint64_t c4o_flip_color(uint64_t p)
{
    uint64_t np_ = 0;
    for (int i = 0; i < WIDTH; ++i) {
        uint64_t col = (p >> (H1 * i)) & COL1;
        np_ |= col << (H1 * (WIDTH - 1 - i));
    }
    return np_;
}

/* board.py:115-126 create_fliplr */
void c4o_fliplr(const c4o_board *b, c4o_board *out)
{
    out->color[0] = c4o_flip_color(b->color[0]);
    out->color[1] = c4o_flip_color(b->color[1]);
    out->age = b->age;
    out->result = b->result;
}

/* board.py:225-243 make_random_ips/expand: the *set* of undecided positions `plies` deep. */
static int cmp_pair(const void *a, const void *b)
{
    const uint64_t *x = (const uint64_t *)a, *y = (const uint64_t *)b;
    if (x[0] != y[0]) return x[0] < y[0] ? -1 : 1;
    if (x[1] != y[1]) return x[1] < y[1] ? -1 : 1;
    return 0;
}
static void ips_expand(const c4o_board *b, int plies, uint64_t *buf, int *n, int cap)
{
    if (plies == 0) {
        if (b->result == C4O_NONE && *n < cap) {
            buf[2 * (*n)] = b->color[0];
            buf[2 * (*n) + 1] = b->color[1];
            (*n)++;
        }
        return;
    }
    int mask = c4o_valid_mask(b);
    for (int m = 0; m < WIDTH; ++m)
        if (mask & (1 << m)) {
            c4o_board nb = *b;
            c4o_make_move(&nb, m);
            ips_expand(&nb, plies - 1, buf, n, cap);
        }
}
int c4o_make_random_ips(int plies, uint64_t *c0, uint64_t *c1, int cap)
{
    int total = 1;
    for (int i = 0; i < plies; ++i) total *= 7;
    uint64_t *buf = (uint64_t *)malloc(sizeof(uint64_t) * 2 * (size_t)total);
    int n = 0;
    c4o_board b;
    c4o_board_init(&b);
    ips_expand(&b, plies, buf, &n, total);
    qsort(buf, (size_t)n, 2 * sizeof(uint64_t), cmp_pair);
    int u = 0;
    for (int i = 0; i < n; ++i)
        if (i == 0 || buf[2 * i] != buf[2 * (i - 1)] || buf[2 * i + 1] != buf[2 * (i - 1) + 1]) {
            if (u < cap) { c0[u] = buf[2 * i]; c1[u] = buf[2 * i + 1]; }
            u++;
        }
    free(buf);
    return u;
}

/* ------------------------------------------------------------------ evaluators.py:28-63 */
/* value_grid[r][c] = [0,1,2,3,2,1,0][c] + [0,1,2,2,1,0][r]; sum = 96 (evaluators.py:48-61).
 * The grid is left-right and up-down symmetric so the flipud in o_pieces does not matter. */
double c4o_evaluate_centre(const c4o_board *b)
{
    static const int gc[7] = {0, 1, 2, 3, 2, 1, 0};
    static const int gr[6] = {0, 1, 2, 2, 1, 0};
    double so = 0.0, sx = 0.0; /* einsum over bool*float: exact small integers */
    for (int c = 0; c < WIDTH; ++c)
        for (int r = 0; r < HEIGHT; ++r) {
            int bit = c * H1 + r;
            double g = (double)(gc[c] + gr[r]);
            if ((b->color[0] >> bit) & 1) so += g;
            if ((b->color[1] >> bit) & 1) sx += g;
        }
    return 0.5 + (so - sx) / 96.0; /* evaluators.py:29-33 */
}

int c4o_eval_centre_with_prior(void *ctx, const c4o_board *b, double *value, double prior[7])
{
    (void)ctx;
    *value = c4o_evaluate_centre(b);
    for (int i = 0; i < 7; ++i) prior[i] = 1.0 / 7.0; /* np.ones(7)/7, evaluators.py:63 */
    return 0;
}

int c4o_eval_table(void *ctx, const c4o_board *b, double *value, double prior[7])
{
    c4o_table *t = (c4o_table *)ctx;
    int64_t lo = 0, hi = t->n - 1;
    while (lo <= hi) {
        int64_t mid = (lo + hi) / 2;
        const c4o_table_entry *e = &t->entries[mid];
        int c = (e->c0 != b->color[0]) ? (e->c0 < b->color[0] ? -1 : 1)
                                       : (e->c1 != b->color[1] ? (e->c1 < b->color[1] ? -1 : 1) : 0);
        if (c == 0) {
            *value = (double)e->value; /* evaluators.py:44 float(value) */
            for (int i = 0; i < 7; ++i) prior[i] = (double)e->prior[i];
            return t->prior_f32;
        }
        if (c < 0) lo = mid + 1; else hi = mid - 1;
    }
    t->misses++;
    return -1;
}

/* ------------------------------------------------------------------ tree.py / mcts.py */
typedef struct {
    c4o_board board;       /* tree.py:21 NodeData.board */
    int32_t parent;
    int32_t first_child;   /* children are created contiguously, ascending column (tree.py:126-129) */
    int32_t n_children;
    int32_t name;          /* move that led here (anytree Node.name) */
    int32_t valid_mask;    /* tree.py:23 */
    int32_t has_position;  /* position_value is not None */
    int32_t has_search;    /* search_value is not None */
    int32_t prior_f32;     /* dtype of position_value.prior: 1 float32, 0 float64 */
    double  position_value;
    double  prior[7];
    double  value_sum;     /* mcts.py:49 */
    uint32_t visit_count;  /* mcts.py:50 */
} onode;

struct c4o_tree {
    c4o_config cfg;
    onode *nodes;
    int64_t n_nodes, cap;
    int side;              /* tree.py:63 root mover */
    int32_t pending;       /* node awaiting evaluation (-1 none) */
    int32_t pending_depth;
    int64_t n_expansions, n_children_created, n_terminal_sims, n_evals, depth_sum;
    int32_t depth_max;
};

static int32_t tree_add_node(c4o_tree *t, const c4o_board *b, int32_t parent, int32_t name)
{
    if (t->n_nodes == t->cap) {
        t->cap = t->cap ? t->cap * 2 : 1024;
        t->nodes = (onode *)realloc(t->nodes, sizeof(onode) * (size_t)t->cap);
    }
    onode *n = &t->nodes[t->n_nodes];
    memset(n, 0, sizeof(*n));
    n->board = *b;
    n->parent = parent;
    n->first_child = -1;
    n->name = name;
    n->valid_mask = c4o_valid_mask(b);
    return (int32_t)t->n_nodes++;
}

c4o_tree *c4o_tree_new(const c4o_config *cfg, const c4o_board *root)
{
    c4o_tree *t = (c4o_tree *)calloc(1, sizeof(c4o_tree));
    t->cfg = *cfg;
    t->side = root->age % 2;     /* tree.py:63 board.player_to_move */
    t->pending = -1;
    tree_add_node(t, root, -1, -1);
    return t;
}

void c4o_tree_free(c4o_tree *t)
{
    if (!t) return;
    free(t->nodes);
    free(t);
}

/* mcts.py:197-202 normalise: zero illegal entries, divide by np.sum in the array's dtype. */
static void normalise(int valid_mask, double p[7], int f32)
{
    for (int i = 0; i < 7; ++i)
        if (!(valid_mask & (1 << i))) p[i] = 0.0;
    if (f32) {
        float s = 0.0f;
        for (int i = 0; i < 7; ++i) s = s + (float)p[i];
        for (int i = 0; i < 7; ++i) p[i] = (double)((float)p[i] / s);
    } else {
        double s = 0.0;
        for (int i = 0; i < 7; ++i) s = s + p[i];
        for (int i = 0; i < 7; ++i) p[i] = p[i] / s;
    }
}

/* tree.py:27-44 NodeData.absolute_value / value(side); utils.py:33-34 value_to_side */
static int node_abs_value(const onode *n, double *v)
{
    if (n->board.result != C4O_NONE) { *v = 0.5 * (double)n->board.result; return 1; }
    if (n->has_search) { *v = n->value_sum / (double)n->visit_count; return 1; }
    if (n->has_position) { *v = n->position_value; return 1; }
    return 0;
}
static double node_value_for(const onode *n, int side)
{
    double v;
    if (!node_abs_value(n, &v)) return 0.0;  /* "position is unknown - assume lost" */
    return side == 0 ? v : (1.0 - v);
}

/* mcts.py:147-161 ucb_score.  Returns the score as a double; in the float32 case the value is
 * exactly the float32 the reference compares. */
static double ucb_score(const c4o_tree *t, const onode *parent, const onode *child)
{
    double np_ = (double)parent->visit_count;
    double pb_c = log((double)((int64_t)parent->visit_count + t->cfg.pb_c_base + 1) /
                      (double)t->cfg.pb_c_base) + t->cfg.pb_c_init;
    uint32_t cv = child->has_search ? child->visit_count : 0;
    pb_c = pb_c * (sqrt(np_) / (double)(cv + 1));
    double value_score = node_value_for(child, parent->board.age % 2);
    if (parent->prior_f32) {
        float prior_score = (float)pb_c * (float)parent->prior[child->name];
        float s = prior_score + (float)value_score;
        return (double)s;
    }
    double prior_score = pb_c * parent->prior[child->name];
    return prior_score + value_score;
}

/* mcts.py:138-144 select_child: max over (score, child); ties -> larger child.name (tree.py:11-15).
 * Children are stored in ascending name order, so ">=" keeps the later (higher) column. */
static int32_t select_child(const c4o_tree *t, int32_t ni)
{
    const onode *parent = &t->nodes[ni];
    int32_t best = -1;
    double best_s = 0.0;
    for (int k = 0; k < parent->n_children; ++k) {
        int32_t ci = parent->first_child + k;
        double s = ucb_score(t, parent, &t->nodes[ci]);
        if (best < 0 || s >= best_s) { best = ci; best_s = s; }
    }
    return best;
}

/* tree.py:119-132 expand_node(node, 1) */
static void expand_node(c4o_tree *t, int32_t ni)
{
    if (t->nodes[ni].board.result != C4O_NONE) return;
    if (t->nodes[ni].n_children) return;
    int mask = t->nodes[ni].valid_mask;
    int32_t first = -1;
    int cnt = 0;
    for (int m = 0; m < WIDTH; ++m)
        if (mask & (1 << m)) {
            c4o_board nb = t->nodes[ni].board;     /* copy(board) */
            c4o_make_move(&nb, m);
            int32_t ci = tree_add_node(t, &nb, ni, m);
            if (first < 0) first = ci;
            cnt++;
        }
    t->nodes[ni].first_child = first;
    t->nodes[ni].n_children = cnt;
    t->n_expansions++;
    t->n_children_created += cnt;
}

/* mcts.py:164-168 backpropagate: every ancestor (the node itself was added in evaluate_node) */
static void backpropagate(c4o_tree *t, int32_t ni, double value)
{
    while (t->nodes[ni].parent >= 0) {
        ni = t->nodes[ni].parent;
        t->nodes[ni].value_sum += value;
        t->nodes[ni].visit_count += 1;
    }
}

/* mcts.py:124-135 evaluate_node, non-terminal branch, given the evaluator's answer */
static void node_store_eval(c4o_tree *t, int32_t ni, double value, const double prior[7], int f32)
{
    onode *n = &t->nodes[ni];
    for (int i = 0; i < 7; ++i) n->prior[i] = prior[i];
    normalise(n->valid_mask, n->prior, f32);
    n->prior_f32 = f32;
    n->position_value = value;
    n->has_position = 1;
    n->has_search = 1;
    n->value_sum = 0.0;
    n->visit_count = 0;
    n->value_sum += value;
    n->visit_count += 1;
    t->n_evals++;
}

void c4o_tree_root_request(c4o_tree *t, c4o_board *leaf)
{
    *leaf = t->nodes[0].board;
    t->pending = 0;
    t->pending_depth = 0;
}

/* mcts.py:101-105 + :171-181 add_exploration_noise */
void c4o_tree_root_apply(c4o_tree *t, double value, const double prior[7], int f32,
                         const double *gamma_noise)
{
    node_store_eval(t, 0, value, prior, f32);
    t->pending = -1;
    onode *r = &t->nodes[0];
    if (t->cfg.root_dirichlet_alpha != 0.0 && t->cfg.root_exploration_fraction != 0.0 && gamma_noise) {
        double noise[7];
        for (int i = 0; i < 7; ++i) noise[i] = gamma_noise[i];
        normalise(r->valid_mask, noise, 0); /* float64 gamma draws */
        double frac = t->cfg.root_exploration_fraction;
        for (int i = 0; i < 7; ++i) {
            double a;
            if (f32) a = (double)((float)r->prior[i] * (float)(1 - frac)); /* f32 array * py float */
            else a = r->prior[i] * (1 - frac);
            double b = noise[i] * frac;
            r->prior[i] = a + b;                                           /* result float64 */
        }
        r->prior_f32 = 0;
    }
}

/* mcts.py:108-116 descent (+ :124-128,134 and :120 for a terminal leaf).
 * Returns 1 if `leaf` needs the evaluator, 0 if the simulation completed on a terminal leaf. */
int c4o_tree_select(c4o_tree *t, c4o_board *leaf)
{
    int32_t ni = 0;
    int depth = 0;
    while (t->nodes[ni].n_children) { ni = select_child(t, ni); depth++; }
    if (t->nodes[ni].has_position) {           /* previously evaluated, so expand */
        expand_node(t, ni);
        ni = select_child(t, ni);
        depth++;
    }
    t->depth_sum += depth;
    if (depth > t->depth_max) t->depth_max = depth;
    onode *n = &t->nodes[ni];
    if (n->board.result != C4O_NONE) {
        double value = 0.5 * (double)n->board.result;
        if (!n->has_search) { n->has_search = 1; n->value_sum = 0.0; n->visit_count = 0; }
        n->value_sum += value;
        n->visit_count += 1;
        backpropagate(t, ni, value);
        t->n_terminal_sims++;
        return 0;
    }
    *leaf = n->board;
    t->pending = ni;
    t->pending_depth = depth;
    return 1;
}

void c4o_tree_apply(c4o_tree *t, double value, const double prior[7], int f32)
{
    int32_t ni = t->pending;
    node_store_eval(t, ni, value, prior, f32);
    backpropagate(t, ni, value);
    t->pending = -1;
}

int c4o_search(c4o_tree *t, c4o_eval_fn eval, void *ctx, const double *gamma_noise)
{
    c4o_board leaf;
    double v, p[7];
    c4o_tree_root_request(t, &leaf);
    int f = eval(ctx, &leaf, &v, p);
    if (f < 0) return f;
    c4o_tree_root_apply(t, v, p, f, gamma_noise);
    for (int s = 0; s < t->cfg.simulations; ++s) {
        if (c4o_tree_select(t, &leaf)) {
            f = eval(ctx, &leaf, &v, p);
            if (f < 0) return f;
            c4o_tree_apply(t, v, p, f);
        }
    }
    return 0;
}

/* tree.py:139-147 _normalise_policy */
static void normalise_policy(const c4o_tree *t, double policy[7])
{
    double s = 0.0;
    for (int i = 0; i < 7; ++i) s = s + policy[i];
    const onode *r = &t->nodes[0];
    if (s == 0.0) {
        for (int k = 0; k < r->n_children; ++k) policy[t->nodes[r->first_child + k].name] = 1.0;
        for (int i = 0; i < 7; ++i) policy[i] = policy[i] / (double)r->n_children;
    } else {
        for (int i = 0; i < 7; ++i) policy[i] = policy[i] / s;
    }
}

void c4o_tree_root_info(const c4o_tree *t, c4o_root_info *o)
{
    memset(o, 0, sizeof(*o));
    const onode *r = &t->nodes[0];
    o->root_visits = r->visit_count;
    o->root_value_sum = r->value_sum;
    for (int i = 0; i < 7; ++i) { o->child_status[i] = -2; o->root_prior[i] = r->prior[i]; }
    double best_v = 0.0;
    o->best_move = -1;
    for (int k = 0; k < r->n_children; ++k) {
        const onode *c = &t->nodes[r->first_child + k];
        int m = c->name;
        o->child_visits[m] = c->has_search ? c->visit_count : 0;
        o->child_value_sum[m] = c->has_search ? c->value_sum : 0.0;
        o->child_status[m] = c->board.result;
        double v = node_value_for(c, t->side);           /* tree.py:66-67 */
        o->child_value[m] = v;
        o->values_policy[m] = v;                          /* tree.py:104-109 */
        o->visit_policy[m] = c->has_search ? (double)c->visit_count : 0.0; /* tree.py:111-117 */
        if (o->best_move < 0 || v >= best_v) { o->best_move = m; best_v = v; } /* tree.py:69-73 */
    }
    if (r->n_children) { normalise_policy(t, o->values_policy); normalise_policy(t, o->visit_policy); }
    o->n_nodes = t->n_nodes;
    o->n_expansions = t->n_expansions;
    o->n_children_created = t->n_children_created;
    o->n_terminal_sims = t->n_terminal_sims;
    o->n_evals = t->n_evals;
    o->depth_sum = t->depth_sum;
    o->depth_max = t->depth_max;
}

/* mcts.py:78-88 + tree.py:69-82.  np.random.choice(range(n), p=probabilities) draws ONE uniform u:
 * cdf = cumsum(p); cdf /= cdf[-1]; idx = searchsorted(cdf, u, side='right'). */
int c4o_tree_pick_move(const c4o_tree *t, int board_age, double u, double *abs_value)
{
    const onode *r = &t->nodes[0];
    int32_t pick = -1;
    if (board_age < t->cfg.num_sampling_moves && u >= 0.0) {
        double w[7], s = 0.0;
        int n = r->n_children;
        for (int k = 0; k < n; ++k) {
            double v = node_value_for(&t->nodes[r->first_child + k], t->side);
            w[k] = pow(v, 2.0);                            /* lambda x: x ** 2 (CPython float_pow -> libm pow) */
        }
        for (int k = 0; k < n; ++k) s = s + w[k];          /* np.sum(values) */
        if (!(s > 0.0)) return -1;
        double cdf[7], acc = 0.0;
        for (int k = 0; k < n; ++k) { acc = acc + w[k] / s; cdf[k] = acc; }
        double last = cdf[n - 1];
        int idx = 0;
        for (int k = 0; k < n; ++k) { cdf[k] = cdf[k] / last; if (cdf[k] <= u) idx = k + 1; }
        if (idx >= n) idx = n - 1;
        pick = r->first_child + idx;
    } else {
        double best_v = 0.0;
        for (int k = 0; k < r->n_children; ++k) {
            double v = node_value_for(&t->nodes[r->first_child + k], t->side);
            if (pick < 0 || v >= best_v) { pick = r->first_child + k; best_v = v; }
        }
    }
    if (pick < 0) return -1;
    double av;
    if (abs_value) *abs_value = node_abs_value(&t->nodes[pick], &av) ? av : NAN;
    return t->nodes[pick].name;
}

/* training_game.py:8-19 */
int c4o_selfplay_game(const c4o_config *cfg, c4o_eval_fn eval, void *ctx,
                      const double *noise_tape, const double *u_tape,
                      c4o_move_record *rec, int *result,
                      int64_t *n_sims, int64_t *n_expansions, int64_t *n_evals)
{
    c4o_board b;
    c4o_board_init(&b);
    int ply = 0;
    int64_t sims = 0, exps = 0, evals = 0;
    while (b.result == C4O_NONE) {
        c4o_tree *t = c4o_tree_new(cfg, &b);
        int rc = c4o_search(t, eval, ctx, noise_tape ? noise_tape + 7 * ply : NULL);
        if (rc < 0) { c4o_tree_free(t); return rc; }
        double av;
        int mv = c4o_tree_pick_move(t, b.age, u_tape ? u_tape[ply] : -1.0, &av);
        if (mv < 0) { c4o_tree_free(t); return -2; }
        c4o_root_info info;
        c4o_tree_root_info(t, &info);
        rec[ply].color0 = b.color[0];
        rec[ply].color1 = b.color[1];
        rec[ply].move = mv;
        rec[ply].value = av;
        for (int i = 0; i < 7; ++i) rec[ply].policy[i] = info.values_policy[i];
        sims += cfg->simulations;
        exps += info.n_expansions;
        evals += info.n_evals;
        c4o_tree_free(t);
        c4o_make_move(&b, mv);
        ply++;
    }
    *result = b.result;
    if (n_sims) *n_sims = sims;
    if (n_expansions) *n_expansions = exps;
    if (n_evals) *n_evals = evals;
    return ply;
}

/* ------------------------------------------------------------------ lock-step pool (CPU baseline)
 * Shape of game_pool.py:15-42 + inference_server.py:37-63: many sequential-MCTS games, leaves of all
 * games evaluated as one batch.  RNG here is the baseline's own (splitmix64); it only feeds the
 * Dirichlet noise and the sampled opening moves. */
typedef struct {
    c4o_board board;
    c4o_tree *tree;
    int sims_done;
    int need_root;
    uint64_t rng;
} ogame;

struct c4o_pool {
    c4o_config cfg;
    int n;
    ogame *g;
    int64_t sims, expansions, games, moves, evals;
};

static uint64_t splitmix64(uint64_t *s)
{
    uint64_t z = (*s += 0x9E3779B97F4A7C15ULL);
    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ULL;
    z = (z ^ (z >> 27)) * 0x94D049BB133111EBULL;
    return z ^ (z >> 31);
}
static double rng_uniform(uint64_t *s) { return (double)(splitmix64(s) >> 11) * (1.0 / 9007199254740992.0); }
static double rng_normal(uint64_t *s)
{
    double u1 = rng_uniform(s), u2 = rng_uniform(s);
    if (u1 < 1e-300) u1 = 1e-300;
    return sqrt(-2.0 * log(u1)) * cos(6.283185307179586 * u2);
}
/* Marsaglia-Tsang, with the alpha<1 boost */
static double rng_gamma(uint64_t *s, double alpha)
{
    double boost = 1.0;
    if (alpha < 1.0) { boost = pow(rng_uniform(s), 1.0 / alpha); alpha += 1.0; }
    double d = alpha - 1.0 / 3.0, c = 1.0 / sqrt(9.0 * d);
    for (int it = 0; it < 64; ++it) {
        double x = rng_normal(s), v = 1.0 + c * x;
        if (v <= 0.0) continue;
        v = v * v * v;
        double u = rng_uniform(s);
        if (u < 1.0 - 0.0331 * x * x * x * x) return boost * d * v;
        if (log(u) < 0.5 * x * x + d * (1.0 - v + log(v))) return boost * d * v;
    }
    return boost * d;
}

void c4o_set_threads(int n)
{
#ifdef _OPENMP
    if (n > 0) omp_set_num_threads(n);
#else
    (void)n;
#endif
}

c4o_pool *c4o_pool_new(const c4o_config *cfg, int n_games, uint64_t seed)
{
    c4o_pool *p = (c4o_pool *)calloc(1, sizeof(c4o_pool));
    p->cfg = *cfg;
    p->n = n_games;
    p->g = (ogame *)calloc((size_t)n_games, sizeof(ogame));
    for (int i = 0; i < n_games; ++i) {
        c4o_board_init(&p->g[i].board);
        p->g[i].rng = seed * 0x100000001B3ULL + (uint64_t)i;
        p->g[i].tree = NULL;
    }
    return p;
}

void c4o_pool_free(c4o_pool *p)
{
    if (!p) return;
    for (int i = 0; i < p->n; ++i) c4o_tree_free(p->g[i].tree);
    free(p->g);
    free(p);
}

static void pool_advance(c4o_pool *p, ogame *g, uint8_t *planes, int64_t *sims, int64_t *exps,
                         int64_t *games, int64_t *moves)
{
    c4o_board leaf;
    for (;;) {
        if (!g->tree) {
            if (g->board.result != C4O_NONE) { c4o_board_init(&g->board); (*games)++; }
            g->tree = c4o_tree_new(&p->cfg, &g->board);
            g->sims_done = 0;
            g->need_root = 1;
            c4o_tree_root_request(g->tree, &leaf);
            break;
        }
        if (g->sims_done >= p->cfg.simulations) {
            double av;
            double u = rng_uniform(&g->rng);
            int mv = c4o_tree_pick_move(g->tree, g->board.age, u, &av);
            if (mv < 0) mv = c4o_tree_pick_move(g->tree, g->board.age, -1.0, &av);
            *exps += g->tree->n_expansions;
            c4o_tree_free(g->tree);
            g->tree = NULL;
            c4o_make_move(&g->board, mv);
            (*moves)++;
            continue;
        }
        if (c4o_tree_select(g->tree, &leaf)) break;
        g->sims_done++;
        (*sims)++;
    }
    c4o_planes(&leaf, planes);
}

void c4o_pool_collect(c4o_pool *p, uint8_t *planes)
{
    int64_t sims = 0, exps = 0, games = 0, moves = 0;
#pragma omp parallel for schedule(dynamic, 8) reduction(+ : sims, exps, games, moves)
    for (int i = 0; i < p->n; ++i)
        pool_advance(p, &p->g[i], planes + (size_t)i * 126, &sims, &exps, &games, &moves);
    p->sims += sims;
    p->expansions += exps;
    p->games += games;
    p->moves += moves;
}

void c4o_pool_apply(c4o_pool *p, const float *values, const float *priors)
{
    int64_t sims = 0;
#pragma omp parallel for schedule(static) reduction(+ : sims)
    for (int i = 0; i < p->n; ++i) {
        ogame *g = &p->g[i];
        double pr[7];
        for (int k = 0; k < 7; ++k) pr[k] = (double)priors[(size_t)i * 7 + k];
        if (g->need_root) {
            double noise[7];
            int use = p->cfg.root_dirichlet_alpha != 0.0 && p->cfg.root_exploration_fraction != 0.0;
            if (use) for (int k = 0; k < 7; ++k) noise[k] = rng_gamma(&g->rng, p->cfg.root_dirichlet_alpha);
            c4o_tree_root_apply(g->tree, (double)values[i], pr, 1, use ? noise : NULL);
            g->need_root = 0;
        } else {
            c4o_tree_apply(g->tree, (double)values[i], pr, 1);
            g->sims_done++;
            sims++;
        }
    }
    p->sims += sims;
    p->evals += p->n;
}

void c4o_pool_stats(const c4o_pool *p, int64_t *sims, int64_t *expansions, int64_t *games,
                    int64_t *moves, int64_t *evals)
{
    int64_t live = 0;
    for (int i = 0; i < p->n; ++i)
        if (p->g[i].tree) live += p->g[i].tree->n_expansions;
    if (sims) *sims = p->sims;
    if (expansions) *expansions = p->expansions + live;
    if (games) *games = p->games;
    if (moves) *moves = p->moves;
    if (evals) *evals = p->evals;
}

/* ------------------------------------------------------------------ lock-step REPLAY pool (parity tests at full size)
 * n games of training_game.py:8-19, each with its own RNG tape (the tapes the device played with), advanced side by side.
 * The evaluator is the reference's memoising Evaluator (evaluators.py:18-25): a position is asked for ONCE -- the caller
 * answers it (with what the device's evaluation cache holds) -- and every later request of any game is served from the
 * table.  c4o_replay_collect advances every unfinished game until it needs a position the table does not hold
 * (OpenMP over games; the table is read-only then); c4o_replay_apply stores the answers and hands them to the games. */
typedef struct {
    c4o_board board;
    c4o_tree *tree;
    int sims_done, need_root, waiting, ply, result, failed;
    c4o_board leaf;
    c4o_move_record rec[42];
} rgame;

typedef struct { uint64_t c0, c1; float value; float prior[7]; } memo_entry;

struct c4o_replay {
    c4o_config cfg;
    int n;
    rgame *g;
    const double *noise;   /* [n][42][7] */
    const double *u;       /* [n][42] */
    memo_entry *tab;       /* open addressing, key (c0, c1); c0 == c1 == ~0 marks an empty cell */
    uint64_t cap, used;
    int64_t lookups, hits;
};

static uint64_t memo_hash(uint64_t c0, uint64_t c1)
{
    uint64_t x = c0 * 0x9E3779B97F4A7C15ULL ^ (c1 + 0xD1B54A32D192ED03ULL) * 0xBF58476D1CE4E5B9ULL;
    x ^= x >> 29;
    x *= 0x94D049BB133111EBULL;
    return x ^ (x >> 32);
}
static const memo_entry *memo_find(const c4o_replay *r, uint64_t c0, uint64_t c1)
{
    uint64_t i = memo_hash(c0, c1) & (r->cap - 1);
    for (;;) {
        const memo_entry *e = &r->tab[i];
        if (e->c0 == c0 && e->c1 == c1) return e;
        if (e->c0 == ~0ULL && e->c1 == ~0ULL) return NULL;
        i = (i + 1) & (r->cap - 1);
    }
}
static void memo_put_raw(memo_entry *tab, uint64_t cap, const memo_entry *v)
{
    uint64_t i = memo_hash(v->c0, v->c1) & (cap - 1);
    while (!(tab[i].c0 == ~0ULL && tab[i].c1 == ~0ULL)) {
        if (tab[i].c0 == v->c0 && tab[i].c1 == v->c1) return;   /* first answer wins (they are all the same net's) */
        i = (i + 1) & (cap - 1);
    }
    tab[i] = *v;
}
static void memo_put(c4o_replay *r, const memo_entry *v)
{
    if (memo_find(r, v->c0, v->c1)) return;
    if ((r->used + 1) * 2 > r->cap) {
        uint64_t ncap = r->cap * 2;
        memo_entry *nt = (memo_entry *)malloc(ncap * sizeof(memo_entry));
        memset(nt, 0xFF, ncap * sizeof(memo_entry));
        for (uint64_t i = 0; i < r->cap; ++i)
            if (!(r->tab[i].c0 == ~0ULL && r->tab[i].c1 == ~0ULL)) memo_put_raw(nt, ncap, &r->tab[i]);
        free(r->tab);
        r->tab = nt;
        r->cap = ncap;
    }
    memo_put_raw(r->tab, r->cap, v);
    r->used++;
}

c4o_replay *c4o_replay_new(const c4o_config *cfg, int n_games, const double *noise_tapes, const double *u_tapes)
{
    c4o_replay *r = (c4o_replay *)calloc(1, sizeof(c4o_replay));
    r->cfg = *cfg;
    r->n = n_games;
    r->g = (rgame *)calloc((size_t)n_games, sizeof(rgame));
    r->noise = noise_tapes;
    r->u = u_tapes;
    r->cap = 1u << 16;
    r->tab = (memo_entry *)malloc(r->cap * sizeof(memo_entry));
    memset(r->tab, 0xFF, r->cap * sizeof(memo_entry));
    for (int i = 0; i < n_games; ++i) { c4o_board_init(&r->g[i].board); r->g[i].result = C4O_NONE; }
    return r;
}

void c4o_replay_free(c4o_replay *r)
{
    if (!r) return;
    for (int i = 0; i < r->n; ++i) c4o_tree_free(r->g[i].tree);
    free(r->g);
    free(r->tab);
    free(r);
}

static void replay_answer(c4o_replay *r, int i, const memo_entry *e)
{
    rgame *g = &r->g[i];
    double pr[7];
    for (int k = 0; k < 7; ++k) pr[k] = (double)e->prior[k];
    if (g->need_root) {
        const int use = r->cfg.root_dirichlet_alpha != 0.0 && r->cfg.root_exploration_fraction != 0.0;
        c4o_tree_root_apply(g->tree, (double)e->value, pr, 1, (use && r->noise) ? r->noise + ((size_t)i * 42 + g->ply) * 7 : NULL);
        g->need_root = 0;
    } else {
        c4o_tree_apply(g->tree, (double)e->value, pr, 1);
        g->sims_done++;
    }
    g->waiting = 0;
}

/* training_game.py:8-19 for game i until it needs a position the table does not hold (returns 1) or has ended (0) */
static int replay_advance(c4o_replay *r, int i, int64_t *lookups, int64_t *hits)
{
    rgame *g = &r->g[i];
    if (g->failed || g->waiting) return g->waiting;
    for (;;) {
        if (!g->tree) {
            if (g->board.result != C4O_NONE) { g->result = g->board.result; return 0; }
            g->tree = c4o_tree_new(&r->cfg, &g->board);
            g->sims_done = 0;
            g->need_root = 1;
            c4o_tree_root_request(g->tree, &g->leaf);
        } else if (g->sims_done >= r->cfg.simulations) {
            double av;
            const int mv = c4o_tree_pick_move(g->tree, g->board.age, r->u ? r->u[(size_t)i * 42 + g->ply] : -1.0, &av);
            if (mv < 0) { g->failed = 1; return 0; }
            c4o_root_info info;
            c4o_tree_root_info(g->tree, &info);
            c4o_move_record *m = &g->rec[g->ply];
            m->color0 = g->board.color[0];
            m->color1 = g->board.color[1];
            m->move = mv;
            m->value = av;
            for (int k = 0; k < 7; ++k) m->policy[k] = info.values_policy[k];
            c4o_tree_free(g->tree);
            g->tree = NULL;
            c4o_make_move(&g->board, mv);
            g->ply++;
            continue;
        } else if (!c4o_tree_select(g->tree, &g->leaf)) {   /* terminal leaf: the simulation is complete */
            g->sims_done++;
            continue;
        }
        (*lookups)++;
        const memo_entry *e = memo_find(r, g->leaf.color[0], g->leaf.color[1]);
        if (!e) { g->waiting = 1; return 1; }
        (*hits)++;
        replay_answer(r, i, e);
    }
}

int c4o_replay_collect(c4o_replay *r, uint64_t *c0, uint64_t *c1, int32_t *game_of)
{
    int64_t lookups = 0, hits = 0;
#pragma omp parallel for schedule(dynamic, 1) reduction(+ : lookups, hits)
    for (int i = 0; i < r->n; ++i) replay_advance(r, i, &lookups, &hits);
    r->lookups += lookups;
    r->hits += hits;
    int m = 0;
    for (int i = 0; i < r->n; ++i)
        if (r->g[i].waiting) { c0[m] = r->g[i].leaf.color[0]; c1[m] = r->g[i].leaf.color[1]; game_of[m] = i; m++; }
    return m;
}

void c4o_replay_apply(c4o_replay *r, int m, const int32_t *game_of, const float *values, const float *priors)
{
    for (int j = 0; j < m; ++j) {
        memo_entry e;
        const rgame *g = &r->g[game_of[j]];
        e.c0 = g->leaf.color[0];
        e.c1 = g->leaf.color[1];
        e.value = values[j];
        for (int k = 0; k < 7; ++k) e.prior[k] = priors[(size_t)j * 7 + k];
        memo_put(r, &e);
    }
    for (int j = 0; j < m; ++j) {
        const int i = game_of[j];
        const memo_entry *e = memo_find(r, r->g[i].leaf.color[0], r->g[i].leaf.color[1]);
        replay_answer(r, i, e);
    }
}

int c4o_replay_game(const c4o_replay *r, int i, c4o_move_record *rec, int *result)
{
    const rgame *g = &r->g[i];
    if (g->failed) return -2;
    if (g->result == C4O_NONE) return -1;
    memcpy(rec, g->rec, sizeof(c4o_move_record) * (size_t)g->ply);
    *result = g->result;
    return g->ply;
}

void c4o_replay_stats(const c4o_replay *r, int64_t *lookups, int64_t *hits, int64_t *table_entries)
{
    if (lookups) *lookups = r->lookups;
    if (hits) *hits = r->hits;
    if (table_entries) *table_entries = (int64_t)r->used;
}
