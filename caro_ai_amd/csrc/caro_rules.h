// caro_rules.h -- bit-packed game rules, one source for the HIP kernels and
// for the host-side single-state helpers of the C-ABI.
//
// Connect four (reference lib/game/connect_four/connect_four.py):
//   the board IS the reference's 63-bit state int (connect_four.py:36-56):
//   cell (column c, row r from the bottom) at bit 62-(6c+r) holding the token,
//   3-bit free-slot count of column c at bits [3(6-c)+2 : 3(6-c)].
//   Nothing is decoded into lists; moves, legality, the win test and the NN
//   planes are computed on the packed word.  The win test (connect_four.py:206-239:
//   a run of four through the new stone, vertical / horizontal / both diagonals)
//   is a branch-free shift-AND over the mover's stones: along a direction with bit
//   stride s, x & x<<s & x<<2s & x<<3s flags every window of four, a row mask
//   keeps windows from wrapping into the next column, and the windows that
//   contain the new stone decide.
// m,n,k game (reference lib/game/tictactoe/tictactoe.py, tictactoe_helpers.py):
//   two bit-planes of n*n bits (cell i = row*n + col, row-major from the top
//   left, as tictactoe.py:14-24 numbers the squares): plane 0 = cells holding
//   token 0, plane 1 = cells holding token 1, W64 64-bit words each.  The
//   reference's base-10 digit-string int (225 digits at 15x15) is converted at
//   the Python edge only.  check_win (tictactoe_helpers.py:7-179: any run >= k on
//   the row, column, diagonal and anti-diagonal through the move) gathers each of
//   the four lines into an n-bit word and finds a run with k-1 shift-ANDs; in the
//   tree kernel the lanes of a descent gather the 4n cells with ONE ballot
//   (move_group).
#ifndef CARO_RULES_H
#define CARO_RULES_H

#include <stdint.h>

#if defined(__HIPCC__)
#define CR_HD __host__ __device__ __forceinline__
#define CR_D __device__ __forceinline__
#else
#define CR_HD inline
#endif

namespace caro {

struct GameParams {
  int kind;  // 0 connect four, 1 m,n,k
  int n, k;  // m,n,k only
  int A, rows, cols;
};

template <int KW_>
struct BoardT {
  static constexpr int KW = KW_;
  uint64_t w[KW_];
};

CR_HD int popc64(uint64_t x) {
#if defined(__HIP_DEVICE_COMPILE__)
  return __popcll(x);
#else
  return __builtin_popcountll(x);
#endif
}

#if defined(__HIPCC__)
// bits [first, first + LPD) of a wave ballot: the lanes of one descent group
template <int LPD>
CR_D uint64_t group_bits(uint64_t ballot, int first) {
  if constexpr (LPD == 64) return ballot;
  else return (ballot >> first) & ((1ull << LPD) - 1ull);
}
#endif

// ------------------------------------------------------------------ connect four
struct C4Rules {
  static constexpr int KW = 1;
  static constexpr int A = 7, ROWS = 6, COLS = 7, HW = 42;
  using Board = BoardT<1>;
  // the occupied cells, carried along a descent next to the state int (the int alone cannot tell an empty cell
  // from a token-0 cell without the column heights)
  struct Aux {
    uint64_t occ;
  };

  static constexpr uint64_t ROWS012 = 0x71C71C71C7000000ULL;  // cells with r in {0,1,2}: 0b111000 per column group
  static constexpr uint64_t ROWS345 = 0x0E38E38E38E00000ULL;  // cells with r in {3,4,5}: 0b000111 per column group

  static CR_HD Board initial(const GameParams&) {
    Board b;
    b.w[0] = 0x1b6db6ULL;  // seven free counts of 6 (connect_four.py:67-74)
    return b;
  }
  static CR_HD int free_of(uint64_t s, int c) { return (int)((s >> (3 * (6 - c))) & 7ULL); }
  static CR_HD int height(uint64_t s, int c) { return 6 - free_of(s, c); }
  static CR_HD int cell(uint64_t s, int c, int r) { return (int)((s >> (62 - (6 * c + r))) & 1ULL); }
  static CR_HD bool legal(const GameParams&, const Board& b, int a) {  // connect_four.py:157-165
    return a < 7 && free_of(b.w[0], a) > 0;
  }
  static CR_HD bool full(const GameParams&, const Board& b) { return (b.w[0] & 0x1fffffULL) == 0; }

  // column c's cells are the 6 bits [57-6c, 62-6c], row 0 on top: h stones = the top h bits of the group
  static CR_HD Aux aux_of(const GameParams&, const Board& b) {
    const uint64_t s = b.w[0];
    uint64_t occ = 0;
    for (int c = 0; c < 7; ++c) {
      const int h = height(s, c);
      occ |= (uint64_t)((0x3Fu << (6 - h)) & 0x3Fu) << (57 - 6 * c);
    }
    return Aux{occ};
  }
  // windows of four along bit stride s: bit b is set iff cells b, b-s, b-2s, b-3s are all set
  static CR_HD uint64_t four(uint64_t x, int s) {
    const uint64_t y = x & (x << s);
    return y & (y << (2 * s));
  }
  // the window starts whose four cells include the cell `nb`
  static CR_HD uint64_t span(uint64_t nb, int s) {
    const uint64_t w = nb | (nb << s);
    return w | (w << (2 * s));
  }
  // bit strides towards the next cell of a line: up 1, right 6, up-right 7, down-right 5.  Horizontal windows
  // cannot wrap (beyond column 6 lie the counter bits, which x never holds); the others are cut by the start row.
  static CR_HD uint64_t wins_dir(uint64_t x, uint64_t nb, int dir) {
    const int s = dir == 0 ? 1 : dir == 1 ? 6 : dir == 2 ? 7 : 5;
    const uint64_t m = dir == 1 ? ~0ULL : dir == 3 ? ROWS345 : ROWS012;
    return four(x, s) & m & span(nb, s);
  }
  // put `player`'s stone into column `col` (must not be full); returns the stone's bit
  static CR_HD uint64_t drop(Board& b, Aux& aux, int col, int player) {
    uint64_t s = b.w[0];
    const int h = height(s, col);
    const uint64_t nb = 1ULL << (62 - (6 * col + h));
    s |= player ? nb : 0ULL;
    s -= 1ULL << (3 * (6 - col));
    b.w[0] = s;
    aux.occ |= nb;
    return nb;
  }
  static CR_HD uint64_t stones(const Board& b, const Aux& aux, int player) {
    return (player ? b.w[0] : ~b.w[0]) & aux.occ;
  }
  // connect_four.py:241-265.  Returns won; the column must not be full.
  static CR_HD bool move(const GameParams&, Board& b, Aux& aux, int col, int player) {
    const uint64_t nb = drop(b, aux, col, player);
    const uint64_t x = stones(b, aux, player);
    return (wins_dir(x, nb, 0) | wins_dir(x, nb, 1) | wins_dir(x, nb, 2) | wins_dir(x, nb, 3)) != 0;
  }
  static CR_HD bool move(const GameParams& gp, Board& b, int col, int player) {
    Aux aux = aux_of(gp, b);
    return move(gp, b, aux, col, player);
  }
#if defined(__HIPCC__)
  // the same move made by the LPD lanes of one descent (all hold the same board): lane l tests direction l & 3,
  // one ballot joins the four.  `first` = the group's first lane in the wave.
  // What lane l contributes does not change along a descent (its direction's stride and start-row mask): formed once
  // per descent (lane_k) instead of by selects on the dependent chain of every level.
  struct LaneK { int s; uint64_t m; };
  static CR_D LaneK lane_k(const GameParams&, int l) {
    const int dir = l & 3;
    LaneK k{dir == 0 ? 1 : dir == 1 ? 6 : dir == 2 ? 7 : 5, dir == 1 ? ~0ULL : dir == 3 ? ROWS345 : ROWS012};
    asm volatile("" : "+v"(k.s), "+v"(k.m));  // keep them in registers: not re-formed inside the loop
    return k;
  }
  template <int LPD>
  static CR_D bool move_group(const GameParams&, Board& b, Aux& aux, int col, int player, const LaneK& k, int first) {
    const uint64_t nb = drop(b, aux, col, player);
    const uint64_t hit = four(stones(b, aux, player), k.s) & k.m & span(nb, k.s);
    return group_bits<LPD>(__ballot(hit != 0), first) != 0;
  }
#endif

  // the same in two steps for a loop over many boards: what depends on the element index alone (plane, the cell's
  // bit, its column's counter, its row) is formed once, the per-board part is a handful of shifts
  static constexpr int PLANE_ITEMS = 2;  // ceil(2 * HW / 64): elements a lane of a 64-lane wave writes per board
  struct PlaneAt { int p, bit, cshift, r; };
  static CR_HD PlaneAt plane_at(const GameParams&, int idx) {
    const int p = idx / 42, i = idx - p * 42, row_idx = i / 7, c = i - row_idx * 7, r = 5 - row_idx;
    return PlaneAt{p, 62 - (6 * c + r), 3 * (6 - c), r};
  }
  static CR_HD float plane_val(const Board& b, int who_move, const PlaneAt& at) {
    const uint64_t s = b.w[0];
    const int h = 6 - (int)((s >> at.cshift) & 7ULL);
    const int mine = (int)((s >> at.bit) & 1ULL) == who_move;
    return (at.r < h && (at.p == 0 ? mine : !mine)) ? 1.0f : 0.0f;
  }
  // value (0/1) of plane `p` at flat index i = row_idx*7 + c: connect_four.py:175-204
  static CR_HD float plane(const GameParams&, const Board& b, int who_move, int p, int i) {
    const int row_idx = i / 7, c = i % 7;
    const int r = 5 - row_idx;
    const uint64_t s = b.w[0];
    if (r >= height(s, c)) return 0.0f;
    const int mine = cell(s, c, r) == who_move;
    return (p == 0 ? mine : !mine) ? 1.0f : 0.0f;
  }
  // every stone of `root` is also on `node` (a position reachable from root contains it): per column the
  // node is at least as high and agrees on the root's occupied rows
  static CR_HD bool contains(const GameParams& gp, const Board& node, const Board& root) {
    const uint64_t ro = aux_of(gp, root).occ, no = aux_of(gp, node).occ;
    return (ro & ~no) == 0 && ((node.w[0] ^ root.w[0]) & ro) == 0;
  }
  // slot hash of the transposition table (home_slot takes the low bits): the HIGH half of ONE 64-bit product, folded
  // once -- three 32-bit multiplies on the GPU instead of the eight of a two-round mixer, on the dependent chain of
  // every descent level.  On the states of whole-game search trees it spreads as well as the two-round form did
  // (0.131 against 0.130 extra probes per insert into a half-full table of 2^15 slots).
  static CR_HD uint64_t hash(const Board& b) {
    const uint32_t hi = (uint32_t)((b.w[0] * 0x9E3779B97F4A7C15ULL) >> 32);
    return hi ^ (hi >> 15);
  }
};

// ------------------------------------------------------------------ m,n,k
template <int W64>
struct MnkRules {
  static constexpr int KW = 2 * W64;
  using Board = BoardT<2 * W64>;
  struct Aux {};  // the two planes are bitboards already

  static CR_HD Board initial(const GameParams&) {
    Board b;
    for (int i = 0; i < KW; ++i) b.w[i] = 0;
    return b;
  }
  static CR_HD Aux aux_of(const GameParams&, const Board&) { return Aux{}; }
  // word `wi` of plane `plane`, by selects over the (few) words: a runtime index into b.w would push the board
  // out of registers
  static CR_HD uint64_t word(const Board& b, int plane, int wi) {
    uint64_t x = 0;
#if defined(__HIPCC__)
#pragma unroll
#endif
    for (int j = 0; j < KW; ++j) x = (plane * W64 + wi == j) ? b.w[j] : x;
    return x;
  }
  static CR_HD uint64_t occupied_word(const Board& b, int wi) {
    uint64_t x = 0;
#if defined(__HIPCC__)
#pragma unroll
#endif
    for (int j = 0; j < W64; ++j) x = (wi == j) ? (b.w[j] | b.w[W64 + j]) : x;
    return x;
  }
  static CR_HD bool bit(const Board& b, int plane, int i) { return (word(b, plane, i >> 6) >> (i & 63)) & 1ULL; }
  static CR_HD bool legal(const GameParams& gp, const Board& b, int a) {  // tictactoe.py:137-150
    return a < gp.A && !((occupied_word(b, a >> 6) >> (a & 63)) & 1ULL);
  }
  static CR_HD bool full(const GameParams& gp, const Board& b) {
    int cnt = 0;
    for (int i = 0; i < W64; ++i) cnt += popc64(b.w[i] | b.w[W64 + i]);
    return cnt >= gp.A;
  }
  // tictactoe.py:226-233: the square is overwritten with the mover's token (the reference does not check it is empty)
  static CR_HD void put(Board& b, int mv, int player) {
    const uint64_t m = 1ULL << (mv & 63);
    const int wi = mv >> 6;
#if defined(__HIPCC__)
#pragma unroll
#endif
    for (int j = 0; j < W64; ++j) {
      const uint64_t clr = (wi == j) ? ~m : ~0ULL, set = (wi == j) ? m : 0ULL;
      b.w[j] = (b.w[j] & clr) | (player == 0 ? set : 0ULL);
      b.w[W64 + j] = (b.w[W64 + j] & clr) | (player == 1 ? set : 0ULL);
    }
  }
  // cell index of element t of line d through (row, col), or -1 when the line has no such element:
  // d = 0 the row (get_row), 1 the column (get_col), 2 the diagonal (get_diag, helpers:86-132), 3 the
  // anti-diagonal (get_antidiag, helpers:135-179; a run is a run in either direction of travel)
  static CR_HD int line_cell(int n, int row, int col, int d, int t) {
    int r, c;
    if (d == 0) { r = row; c = t; }
    else if (d == 1) { r = t; c = col; }
    else if (d == 2) { const int m = row < col ? row : col; r = row - m + t; c = col - m + t; }
    else { r = row + col - t; c = t; }
    return (t < n && r >= 0 && r < n && c < n) ? r * n + c : -1;
  }
  // a run of >= k ones in the low n bits of f (k_in_a_row, helpers:27-58)
  static CR_HD bool has_run(uint64_t f, int k) {
    for (int i = 1; i < k; ++i) f &= f >> 1;
    return f != 0;
  }
  // tictactoe.py:210-235: overwrite the square, then check_win over the four lines through it
  static CR_HD bool move(const GameParams& gp, Board& b, Aux&, int mv, int player) {
    put(b, mv, player);
    const int n = gp.n, row = mv / n, col = mv % n;
    for (int d = 0; d < 4; ++d) {
      uint64_t f = 0;
      for (int t = 0; t < n; ++t) {
        const int i = line_cell(n, row, col, d, t);
        f |= (uint64_t)(i >= 0 && bit(b, player, i)) << t;
      }
      if (has_run(f, gp.k)) return true;
    }
    return false;
  }
  static CR_HD bool move(const GameParams& gp, Board& b, int mv, int player) {
    Aux aux;
    return move(gp, b, aux, mv, player);
  }
#if defined(__HIPCC__)
  // The LPD lanes of one descent (all hold the same board) gather the four lines with one ballot: lane l looks at
  // element l % NL of line l / NL (NL = LPD / 4 >= n for every geometry: n <= 4 | 16 lanes, 5 | 32, <= 15 | 64),
  // the group's ballot bits are the four line words side by side, and a run of k is found in all four at once;
  // `starts` keeps a run from straddling two lines when n == NL.
  struct LaneK { int l; };
  static CR_D LaneK lane_k(const GameParams&, int l) { return LaneK{l}; }
  template <int LPD>
  static CR_D bool move_group(const GameParams& gp, Board& b, Aux&, int mv, int player, const LaneK& lk, int first) {
    constexpr int NL = LPD / 4;
    const int l = lk.l;
    put(b, mv, player);
    const int n = gp.n, row = mv / n, col = mv % n;
    const int i = line_cell(n, row, col, l / NL, l % NL);
    uint64_t f = group_bits<LPD>(__ballot(i >= 0 && bit(b, player, i)), first);
    for (int j = 1; j < gp.k; ++j) f &= f >> 1;
    uint64_t starts = (1ull << (NL - gp.k + 1)) - 1ull;  // a run may start at elements 0 .. NL-k of a line
    starts |= starts << NL;
    starts |= starts << (2 * NL);
    return (f & starts) != 0;
  }
#endif
  // tictactoe.py:164-176: plane 0 = who_move's tokens, plane 1 = the other token; no row flip
  static CR_HD float plane(const GameParams&, const Board& b, int who_move, int p, int i) {
    return bit(b, p == 0 ? who_move : 1 - who_move, i) ? 1.0f : 0.0f;
  }
  // (two-step form, as C4Rules)
  static constexpr int PLANE_ITEMS = 2 * W64;  // 2 * HW <= 2 * 64 * W64 elements over 64 lanes
  struct PlaneAt { int p, i; };
  static CR_HD PlaneAt plane_at(const GameParams& gp, int idx) {
    const int hw = gp.rows * gp.cols, p = idx / hw;
    return PlaneAt{p, idx - p * hw};
  }
  static CR_HD float plane_val(const Board& b, int who_move, const PlaneAt& at) {
    return bit(b, at.p == 0 ? who_move : 1 - who_move, at.i) ? 1.0f : 0.0f;
  }
  static CR_HD bool contains(const GameParams&, const Board& node, const Board& root) {
    bool ok = true;
    for (int i = 0; i < KW; ++i) ok = ok && ((node.w[i] & root.w[i]) == root.w[i]);
    return ok;
  }
  static CR_HD uint64_t hash(const Board& b) {
    uint64_t z = 0x243F6A8885A308D3ULL;
    for (int i = 0; i < KW; ++i) {
      z ^= b.w[i];
      z *= 0x9E3779B97F4A7C15ULL;
      z ^= z >> 31;
    }
    z *= 0xbf58476d1ce4e5b9ULL;
    z ^= z >> 32;
    return z;
  }
};

template <class R>
CR_HD bool board_eq(const typename R::Board& a, const typename R::Board& b) {
  bool eq = true;
  for (int i = 0; i < R::KW; ++i) eq = eq && (a.w[i] == b.w[i]);
  return eq;
}

}  // namespace caro
#endif
