// c4_board.h -- 6x7 Connect4 bitboard for gfx950 device code (and the host side of the engine).
//
// Layout follows the reference (oinkoink/board.py:9-32): 7 bits per column (6 cells + one
// sentinel bit that is never set), bit index = col*7 + row, row 0 = bottom.  color[0] = o stones,
// color[1] = x stones, age = number of stones; o moves when age is even.
//
// All functions are branch-light integer code: a make-move + win-check is ~25 VALU ops, so the
// tree walk never stores boards per node -- it replays moves in registers during the descent.
#pragma once
#include <stdint.h>

#if defined(__HIPCC__)
#define C4_HD __host__ __device__ __forceinline__
#else
#define C4_HD static inline
#endif

namespace c4 {

constexpr int WIDTH = 7;
constexpr int HEIGHT = 6;
constexpr int H1 = 7;
constexpr int CELLS = 42;
constexpr uint64_t BOTTOM = 0x40810204081ULL;  // bit 7*c for every column (board.py:19)
constexpr uint64_t COLMASK = 0x3f;             // the 6 playable cells of column 0

// node / result status codes shared by the kernels (value of a terminal = (status-2)*0.5,
// utils.py:19-22: x_win 0.0, draw 0.5, o_win 1.0)
constexpr uint32_t ST_FRESH = 0;      // non-terminal, not yet evaluated (position_value is None)
constexpr uint32_t ST_EVALUATED = 1;  // non-terminal, evaluated: children exist
constexpr uint32_t ST_XWIN = 2;
constexpr uint32_t ST_DRAW = 3;
constexpr uint32_t ST_OWIN = 4;

C4_HD int popc64(uint64_t x)
{
#if defined(__HIP_DEVICE_COMPILE__)
    return __popcll(x);
#else
    return __builtin_popcountll(x);
#endif
}

// board.py:173-184 _check_terminal_position: four-in-a-row on one colour's stones.
C4_HD bool wins(uint64_t b)
{
    uint64_t d1 = b & (b >> 6);   // diagonal '\'
    uint64_t hz = b & (b >> 7);   // horizontal
    uint64_t d2 = b & (b >> 8);   // diagonal '/'
    uint64_t vt = b & (b >> 1);   // vertical
    return ((d1 & (d1 >> 12)) | (hz & (hz >> 14)) | (d2 & (d2 >> 16)) | (vt & (vt >> 2))) != 0;
}

// board.py:39,163: height[col] = 7*col + stones already in the column (count based, exactly as
// the reference keeps it, so even "floating" from_pieces positions agree).
C4_HD int col_count(uint64_t occ, int col) { return popc64((occ >> (H1 * col)) & 0x7f); }
C4_HD uint64_t drop_bit(uint64_t occ, int col) { return 1ULL << (H1 * col + col_count(occ, col)); }

// board.py:88-92,187-188: a column is playable while its height bit is not the sentinel.
C4_HD int legal_mask(uint64_t occ)
{
    int m = 0;
#pragma unroll
    for (int c = 0; c < WIDTH; ++c) m |= (col_count(occ, c) < HEIGHT ? 1 : 0) << c;
    return m;
}

// board.py:56-62: result of an arbitrary position (used for start positions handed to the engine)
C4_HD uint32_t position_status(uint64_t c0, uint64_t c1)
{
    if (wins(c0)) return ST_OWIN;
    if (wins(c1)) return ST_XWIN;
    if (popc64(c0 | c1) == CELLS) return ST_DRAW;
    return ST_FRESH;
}

// board.py:160-170 make_move for the side to move; returns the status of the new position.
C4_HD uint32_t make_move(uint64_t &c0, uint64_t &c1, int col)
{
    const uint64_t occ = c0 | c1;
    const int age = popc64(occ);
    const uint64_t bit = drop_bit(occ, col);
    uint64_t mine = (age & 1) ? c1 : c0;
    mine ^= bit;
    if (age & 1) c1 = mine; else c0 = mine;
    const int nage = age + 1;
    if (wins(mine)) return (nage & 1) ? ST_OWIN : ST_XWIN;   // Result(age % 2)
    return nage == CELLS ? ST_DRAW : ST_FRESH;
}

// board.py:128-145 flip_color (mirror left<->right)
C4_HD uint64_t flip_color(uint64_t p)
{
    uint64_t r = 0;
#pragma unroll
    for (int c = 0; c < WIDTH; ++c) r |= ((p >> (H1 * c)) & 0x7f) << (H1 * (WIDTH - 1 - c));
    return r;
}

// board.py:147-154 to_array element e of the [3][6][7] NN input (row 0 = top of the board).
C4_HD float plane_element(uint64_t c0, uint64_t c1, int o_to_move, int e)
{
    const int ch = e / CELLS;
    const int rc = e - ch * CELLS;
    const int r = rc / WIDTH;
    const int c = rc - r * WIDTH;
    const int bit = c * H1 + (HEIGHT - 1 - r);
    const uint64_t src = ch == 1 ? c0 : c1;
    return ch == 0 ? (float)o_to_move : (float)((src >> bit) & 1);
}

// evaluators.py:28-33,48-61 evaluate_centre: 0.5 + (sum_o grid - sum_x grid)/96 in float64.
// grid[r][c] = [0,1,2,3,2,1,0][c] + [0,1,2,2,1,0][r].
C4_HD double centre_value(uint64_t c0, uint64_t c1)
{
    int so = 0, sx = 0;
#pragma unroll
    for (int c = 0; c < WIDTH; ++c) {
        const int gc = c < 4 ? c : 6 - c;
        const uint64_t a = (c0 >> (H1 * c)) & COLMASK, b = (c1 >> (H1 * c)) & COLMASK;
        // row weights 0,1,2,2,1,0 -> rows {1,4} weight 1, rows {2,3} weight 2
        so += gc * popc64(a) + popc64(a & 0x12) + 2 * popc64(a & 0x0c);
        sx += gc * popc64(b) + popc64(b & 0x12) + 2 * popc64(b & 0x0c);
    }
    return 0.5 + ((double)so - (double)sx) / 96.0;
}

}  // namespace c4
