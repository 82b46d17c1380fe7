// caro_rules.h -- bit-packed game rules, one source for the HIP kernels and
// for the host-side single-state helpers of the C-ABI.
//
// Connect four (reference lib/game/connect_four/connect_four.py):
//   the board IS the reference's 63-bit state int (connect_four.py:36-56):
//   cell (column c, row r from the bottom) at bit 62-(6c+r) holding the token,
//   3-bit free-slot count of column c at bits [3(6-c)+2 : 3(6-c)].
//   Nothing is decoded into lists; moves, legality, the win test and the NN
//   planes are computed on the packed word.
// m,n,k game (reference lib/game/tictactoe/tictactoe.py, tictactoe_helpers.py):
//   two bit-planes of n*n bits (cell i = row*n + col, row-major from the top
//   left, as tictactoe.py:14-24 numbers the squares): plane 0 = cells holding
//   token 0, plane 1 = cells holding token 1, W64 64-bit words each.  The
//   reference's base-10 digit-string int (225 digits at 15x15) is converted at
//   the Python edge only.
#ifndef CARO_RULES_H
#define CARO_RULES_H

#include <stdint.h>

#if defined(__HIPCC__)
#define CR_HD __host__ __device__ __forceinline__
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

// ------------------------------------------------------------------ connect four
struct C4Rules {
  static constexpr int KW = 1;
  static constexpr int A = 7, ROWS = 6, COLS = 7, HW = 42;
  using Board = BoardT<1>;

  static CR_HD Board initial(const GameParams&) {
    Board b;
    b.w[0] = 0x1b6db6ULL;  // seven free counts of 6 (connect_four.py:67-74)
    return b;
  }
  static CR_HD int free_of(uint64_t s, int c) { return (int)((s >> (3 * (6 - c))) & 7ULL); }
  static CR_HD int height(uint64_t s, int c) { return 6 - free_of(s, c); }
  static CR_HD int cell(uint64_t s, int c, int r) { return (int)((s >> (62 - (6 * c + r))) & 1ULL); }
  // cell (c,r) is occupied by `player`
  static CR_HD bool is(uint64_t s, int c, int r, int player) {
    return r >= 0 && r < height(s, c) && cell(s, c, r) == player;
  }
  static CR_HD bool legal(const GameParams&, const Board& b, int a) {  // connect_four.py:157-165
    return a < 7 && free_of(b.w[0], a) > 0;
  }
  static CR_HD bool full(const GameParams&, const Board& b) { return (b.w[0] & 0x1fffffULL) == 0; }

  // contiguous run through (col,row) along (dc=+-1, dr=delta): connect_four.py:206-239
  static CR_HD bool line(uint64_t s, int col, int row, int player, int delta) {
    int total = 1;
    int cur = row - delta;
    for (int c = col - 1; c >= 0; --c) {
      if (!is(s, c, cur, player)) break;
      if (++total == 4) return true;
      cur -= delta;
    }
    cur = row + delta;
    for (int c = col + 1; c < 7; ++c) {
      if (!is(s, c, cur, player)) break;
      if (++total == 4) return true;
      cur += delta;
    }
    return false;
  }

  // connect_four.py:241-265.  Returns won; the column must not be full.
  static CR_HD bool move(const GameParams&, Board& b, int col, int player) {
    uint64_t s = b.w[0];
    const int h = height(s, col);
    s |= (uint64_t)player << (62 - (6 * col + h));
    s -= 1ULL << (3 * (6 - col));
    b.w[0] = s;
    bool won = false;
    if (h >= 3)
      won = cell(s, col, h - 1) == player && cell(s, col, h - 2) == player && cell(s, col, h - 3) == player;
    if (!won) won = line(s, col, h, player, 0) || line(s, col, h, player, 1) || line(s, col, h, player, -1);
    return won;
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
  static CR_HD bool contains(const GameParams&, const Board& node, const Board& root) {
    const uint64_t a = node.w[0], r = root.w[0];
    for (int c = 0; c < 7; ++c) {
      const int hr = height(r, c);
      if (height(a, c) < hr) return false;
      const uint64_t col_mask = ((1ULL << hr) - 1ULL) << (62 - (6 * c + hr - 1));  // rows 0..hr-1 of column c
      if (hr > 0 && ((a ^ r) & col_mask)) return false;
    }
    return true;
  }
  static CR_HD uint64_t hash(const Board& b) {
    uint64_t z = b.w[0] * 0x9E3779B97F4A7C15ULL;
    z ^= z >> 29;
    z *= 0xbf58476d1ce4e5b9ULL;
    z ^= z >> 32;
    return z;
  }
};

// ------------------------------------------------------------------ m,n,k
template <int W64>
struct MnkRules {
  static constexpr int KW = 2 * W64;
  using Board = BoardT<2 * W64>;

  static CR_HD Board initial(const GameParams&) {
    Board b;
    for (int i = 0; i < KW; ++i) b.w[i] = 0;
    return b;
  }
  static CR_HD bool bit(const Board& b, int plane, int i) {
    return (b.w[plane * W64 + (i >> 6)] >> (i & 63)) & 1ULL;
  }
  static CR_HD bool legal(const GameParams& gp, const Board& b, int a) {  // tictactoe.py:137-150
    return a < gp.A && !bit(b, 0, a) && !bit(b, 1, a);
  }
  static CR_HD bool full(const GameParams& gp, const Board& b) {
    int cnt = 0;
    for (int i = 0; i < W64; ++i) cnt += popc64(b.w[i] | b.w[W64 + i]);
    return cnt >= gp.A;
  }
  // any run >= k of `player` along the whole line (r0 + t*dr, c0 + t*dc), t = 0..len-1:
  // tictactoe_helpers.py:27-58 (k_in_a_row over the full row / column / diagonal)
  static CR_HD bool run(const GameParams& gp, const Board& b, int player, int r0, int c0, int dr, int dc,
                        int len) {
    int cur = 0;
    for (int t = 0; t < len; ++t) {
      const int i = (r0 + t * dr) * gp.n + (c0 + t * dc);
      cur = bit(b, player, i) ? cur + 1 : 0;
      if (cur >= gp.k) return true;
    }
    return false;
  }
  // tictactoe.py:210-235: overwrite the square, then check_win over the four lines through it
  static CR_HD bool move(const GameParams& gp, Board& b, int mv, int player) {
    const int n = gp.n;
    const uint64_t m = 1ULL << (mv & 63);
    const int wi = mv >> 6;
    b.w[wi] &= ~m;
    b.w[W64 + wi] &= ~m;
    b.w[player * W64 + wi] |= m;
    const int row = mv / n, col = mv % n;
    if (run(gp, b, player, row, 0, 0, 1, n)) return true;  // get_row
    if (run(gp, b, player, 0, col, 1, 0, n)) return true;  // get_col
    {                                                      // get_diag, helpers:86-132
      const int d = row < col ? row : col;
      const int r0 = row - d, c0 = col - d;
      const int len = n - (r0 > c0 ? r0 : c0);
      if (run(gp, b, player, r0, c0, 1, 1, len)) return true;
    }
    {  // get_antidiag, helpers:135-179: from the bottom-left end going up-right
      int r0, c0;
      if (row + col < n) {
        r0 = row + col;
        c0 = 0;
      } else {
        r0 = n - 1;
        c0 = row + col - (n - 1);
      }
      const int len = (r0 + 1) < (n - c0) ? (r0 + 1) : (n - c0);
      if (run(gp, b, player, r0, c0, -1, 1, len)) return true;
    }
    return false;
  }
  // tictactoe.py:164-176: plane 0 = who_move's tokens, plane 1 = the other token; no row flip
  static CR_HD float plane(const GameParams&, const Board& b, int who_move, int p, int i) {
    return bit(b, p == 0 ? who_move : 1 - who_move, i) ? 1.0f : 0.0f;
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
