/*
 * caro_oracle.c -- TEST INFRASTRUCTURE ONLY.
 *
 * CPU restatement of the self-play hot path of nh273/caro-ai (reference tree
 * at /root/reference, cited as file:line below).  It exists to CHECK the HIP
 * engine: only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg
 * may load it.  Nothing under caro_ai_amd/ links, imports or calls it.
 *
 * Parity status: PINNED.  tests/test_oracle_golden.py checks this file against
 * (a) the reference's own known-answer tests restated as data inside that test
 * file (the numbers of lib/test_mcts.py:25-38, lib/game/connect_four/
 * test_connect_four.py:28-191, lib/game/tictactoe/test_tictactoe.py:12-144),
 * and (b) vectors recorded by running the reference itself
 * in the build container (tests/golden/make_golden.py, outputs committed under
 * tests/golden/).
 *
 * The restatement is deliberately list/array based like the reference (columns
 * as lists, boards as cell matrices, dict -> open hash map) and shares no code
 * with the bit-packed product kernels.
 *
 * Numeric semantics restated: the reference run under numpy >= 2 (NEP 50), the
 * interpreter the vectors were recorded with.  A tree value is either a Python
 * float (binary64) or an np.float32; mixed arithmetic follows NEP 50 (Python
 * scalars are weak).  `pyval` below models exactly that.
 */
#include <math.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include "../include/caro_noise.h" /* public spec of the generated random inputs */

#if defined(__GNUC__)
#pragma GCC optimize("fp-contract=off")
#endif

#define MAXA 225
#define MAXCELLS 225

/* ------------------------------------------------------------------ games */

typedef struct {
  int kind; /* 0 connect four, 1 m,n,k (n x n board, k in a row) */
  int n, k; /* m,n,k only */
  int rows, cols, A;
} ogame;

/* Oracle state: cell matrix, row-major from the top-left as the net sees it
 * for m,n,k (tictactoe.py:14-24); for connect four `cells` holds the seven
 * column lists bottom-up: cells[c*6 + r], 2 = empty (connect_four.py:16-34). */
typedef struct {
  uint8_t cells[MAXCELLS];
} ostate;

#define EMPTY 2

static void game_init(ogame* g, int kind, int n, int k) {
  g->kind = kind;
  if (kind == 0) {
    g->rows = 6; g->cols = 7; g->A = 7; g->n = 0; g->k = 4; /* connect_four.py:59-65 */
  } else {
    g->n = n; g->k = k; g->rows = n; g->cols = n; g->A = n * n; /* tictactoe.py:27-43,86 */
  }
}

static int ncells(const ogame* g) { return g->rows * g->cols; }

static void initial_state(const ogame* g, ostate* s) {
  memset(s->cells, EMPTY, sizeof s->cells); /* connect_four.py:67-74, tictactoe.py:56-63 */
  (void)g;
}

/* ---- connect four codec: connect_four.py:94-147 ---- */
static int c4_col_len(const ostate* s, int c) {
  int len = 0;
  while (len < 6 && s->cells[c * 6 + len] != EMPTY) ++len;
  return len;
}

uint64_t oracle_c4_encode(const ostate* s) { /* encode_lists, :108-128 */
  int bits[63];
  int nb = 0;
  int len_bits[21];
  int nl = 0;
  for (int c = 0; c < 7; ++c) {
    int len = c4_col_len(s, c);
    for (int r = 0; r < len; ++r) bits[nb++] = s->cells[c * 6 + r];
    int free_len = 6 - len;
    for (int r = 0; r < free_len; ++r) bits[nb++] = 0;
    /* int_to_bits(free_len, 3): MSB first (:101-106) */
    len_bits[nl++] = (free_len >> 2) & 1;
    len_bits[nl++] = (free_len >> 1) & 1;
    len_bits[nl++] = free_len & 1;
  }
  for (int i = 0; i < nl; ++i) bits[nb++] = len_bits[i];
  uint64_t res = 0; /* bits_to_int :94-99 */
  for (int i = 0; i < nb; ++i) res = res * 2 + (uint64_t)bits[i];
  return res;
}

void oracle_c4_decode(uint64_t state_int, ostate* s) { /* decode_binary, :130-147 */
  int bits[63];
  uint64_t num = state_int;
  for (int i = 62; i >= 0; --i) { bits[i] = (int)(num % 2); num /= 2; }
  memset(s->cells, EMPTY, sizeof s->cells);
  for (int c = 0; c < 7; ++c) {
    int lens = bits[42 + c * 3] * 4 + bits[42 + c * 3 + 1] * 2 + bits[42 + c * 3 + 2];
    int keep = 6 - lens; /* vals[:-lens] */
    if (lens <= 0) keep = 6;
    if (keep < 0) keep = 0;
    for (int r = 0; r < keep; ++r) s->cells[c * 6 + r] = (uint8_t)bits[c * 6 + r];
  }
}

/* ---- legal moves ---- */
static int possible_moves(const ogame* g, const ostate* s, int* out) {
  int cnt = 0;
  if (g->kind == 0) { /* connect_four.py:157-165 */
    for (int c = 0; c < 7; ++c)
      if (c4_col_len(s, c) < 6) out[cnt++] = c;
  } else { /* tictactoe.py:137-150 */
    for (int i = 0; i < g->A; ++i)
      if (s->cells[i] == EMPTY) out[cnt++] = i;
  }
  return cnt;
}

/* ---- connect four move + win check: connect_four.py:206-265 ---- */
static int c4_check_won(const ostate* f, int col, int delta_row) {
  int coord = c4_col_len(f, col) - 1;
  int player = f->cells[col * 6 + coord];
  int total = 1;
  int cur = coord - delta_row;
  for (int c = col - 1; c >= 0; --c) {
    if (c4_col_len(f, c) <= cur || cur < 0 || cur >= 6) break;
    if (f->cells[c * 6 + cur] != player) break;
    if (++total == 4) return 1;
    cur -= delta_row;
  }
  cur = coord + delta_row;
  for (int c = col + 1; c < 7; ++c) {
    if (c4_col_len(f, c) <= cur || cur < 0 || cur >= 6) break;
    if (f->cells[c * 6 + cur] != player) break;
    if (++total == 4) return 1;
    cur += delta_row;
  }
  return 0;
}

static int c4_move(ostate* s, int col, int player) {
  int len = c4_col_len(s, col);
  if (len >= 6) return -1; /* assert len(field[col]) < game_rows, :255 */
  s->cells[col * 6 + len] = (uint8_t)player;
  ++len;
  int won = 0;
  if (len >= 4) { /* suff == [player]*4, :258-259 */
    won = 1;
    for (int r = len - 4; r < len; ++r)
      if (s->cells[col * 6 + r] != player) won = 0;
  }
  if (!won) won = c4_check_won(s, col, 0) || c4_check_won(s, col, 1) || c4_check_won(s, col, -1);
  return won;
}

/* ---- m,n,k move + win check: tictactoe.py:210-235, tictactoe_helpers.py ---- */
static int k_in_a_row(const int* arr, int len, int k, int token) { /* helpers:27-58 */
  if (len < k) return 0;
  int start = -1;
  for (int i = 0; i < len; ++i) {
    if (arr[i] == token) {
      if (start < 0) start = i;
      else if (i - start + 1 >= k) return 1;
    } else {
      start = -1;
      if (i >= len - k) return 0;
    }
  }
  return 0;
}

static int mnk_check_win(const ogame* g, const ostate* s, int row, int col, int token) {
  int n = g->n, arr[16], len;
  /* get_row :61-72 */
  len = 0;
  for (int c = 0; c < n; ++c) arr[len++] = s->cells[row * n + c];
  if (k_in_a_row(arr, len, g->k, token)) return 1;
  /* get_col :75-83 */
  len = 0;
  for (int r = 0; r < n; ++r) arr[len++] = s->cells[r * n + col];
  if (k_in_a_row(arr, len, g->k, token)) return 1;
  /* get_diag :86-132 */
  {
    int rs, cs, re, ce;
    if (row >= col) { cs = 0; rs = row - col; re = n - 1; ce = re - cs; }
    else { rs = 0; cs = col - row; ce = n - 1; re = ce - cs; }
    len = 0;
    for (int x = rs, y = cs; x <= re && y <= ce; ++x, ++y) arr[len++] = s->cells[x * n + y];
    if (k_in_a_row(arr, len, g->k, token)) return 1;
  }
  /* get_antidiag :135-179 */
  {
    int rs, cs, re, ce;
    if (row + col < n) { cs = 0; rs = row + col; re = 0; ce = rs; }
    else { ce = n - 1; cs = col + row - ce; rs = n - 1; re = cs; }
    len = 0;
    for (int x = rs, y = cs; x >= re && y <= ce; --x, ++y) arr[len++] = s->cells[x * n + y];
    if (k_in_a_row(arr, len, g->k, token)) return 1;
  }
  return 0;
}

/* returns won (0/1) or -1 on a rejected move */
static int game_move(const ogame* g, ostate* s, int move, int player) {
  if (g->kind == 0) {
    if (move < 0 || move >= 7) return -1;
    return c4_move(s, move, player);
  }
  if (move < 0 || move >= g->A) return -1;
  int row = move / g->n, col = move % g->n;
  s->cells[move] = (uint8_t)player; /* overwrites without checking, :231 */
  return mnk_check_win(g, s, row, col, player);
}

/* ---- NN planes: connect_four.py:175-204, tictactoe.py:164-208 ---- */
static void encode_planes(const ogame* g, const ostate* s, int who_move, float* dst) {
  int hw = ncells(g);
  memset(dst, 0, sizeof(float) * 2 * hw);
  if (g->kind == 0) {
    for (int c = 0; c < 7; ++c) {
      int len = c4_col_len(s, c);
      for (int rev = 0; rev < len; ++rev) {
        int row_idx = 6 - rev - 1;
        if (s->cells[c * 6 + rev] == who_move) dst[0 * hw + row_idx * 7 + c] = 1.0f;
        else dst[1 * hw + row_idx * 7 + c] = 1.0f;
      }
    }
  } else {
    for (int i = 0; i < hw; ++i) {
      if (s->cells[i] == who_move) dst[i] = 1.0f;
      else if (s->cells[i] != EMPTY) dst[hw + i] = 1.0f;
    }
  }
}

/* ------------------------------------------------- NEP 50 scalar model */

typedef struct {
  double v;
  int f32; /* 0: Python float (binary64, weak)   1: np.float32 */
} pyval;

static pyval py(double v) { pyval r = {v, 0}; return r; }
static pyval f32v(float v) { pyval r = {(double)v, 1}; return r; }

static pyval pv_add(pyval a, pyval b) {
  if (a.f32 || b.f32) return f32v((float)a.v + (float)b.v);
  return py(a.v + b.v);
}
static pyval pv_neg(pyval a) { pyval r = {-a.v, a.f32}; return r; }
static pyval pv_div_int(pyval a, int n) { /* value / visit_count, mcts.py:244-245 */
  if (a.f32) return f32v((float)a.v / (float)n);
  return py(a.v / (double)n);
}

/* ------------------------------------------------------------- MCTS store */

typedef struct {
  ostate key;
  int N[MAXA];
  pyval W[MAXA];
  pyval Q[MAXA];
  float P[MAXA];
} onode;

typedef struct {
  const ogame* game;
  onode* nodes;
  int n_nodes, cap_nodes;
  int* slots; /* open addressing, -1 empty */
  int n_slots;
  double c_puct;
} omcts;

static uint64_t state_hash(const ogame* g, const ostate* s) {
  uint64_t h = 1469598103934665603ULL;
  int hw = ncells(g);
  for (int i = 0; i < hw; ++i) { h ^= s->cells[i]; h *= 1099511628211ULL; }
  return h;
}

static void mcts_init(omcts* m, const ogame* g, double c_puct) { /* mcts.py:27-37 */
  m->game = g;
  m->cap_nodes = 1024;
  m->nodes = (onode*)malloc(sizeof(onode) * m->cap_nodes);
  m->n_nodes = 0;
  m->n_slots = 4096;
  m->slots = (int*)malloc(sizeof(int) * m->n_slots);
  for (int i = 0; i < m->n_slots; ++i) m->slots[i] = -1;
  m->c_puct = c_puct;
}
static void mcts_free(omcts* m) { free(m->nodes); free(m->slots); }
static void mcts_clear(omcts* m) { /* mcts.py:39-43 */
  m->n_nodes = 0;
  for (int i = 0; i < m->n_slots; ++i) m->slots[i] = -1;
}

static int mcts_find(const omcts* m, const ostate* s) { /* `state in self.probs`, mcts.py:160 */
  int hw = ncells(m->game);
  uint64_t h = state_hash(m->game, s);
  int mask = m->n_slots - 1;
  for (int i = (int)(h & (uint64_t)mask);; i = (i + 1) & mask) {
    int idx = m->slots[i];
    if (idx < 0) return -1;
    if (memcmp(m->nodes[idx].key.cells, s->cells, hw) == 0) return idx;
  }
}

static void mcts_rehash(omcts* m) {
  int ns = m->n_slots * 2;
  int* slots = (int*)malloc(sizeof(int) * ns);
  for (int i = 0; i < ns; ++i) slots[i] = -1;
  for (int n = 0; n < m->n_nodes; ++n) {
    uint64_t h = state_hash(m->game, &m->nodes[n].key);
    int i = (int)(h & (uint64_t)(ns - 1));
    while (slots[i] >= 0) i = (i + 1) & (ns - 1);
    slots[i] = n;
  }
  free(m->slots);
  m->slots = slots;
  m->n_slots = ns;
}

static void mcts_create_node(omcts* m, const ostate* s, const float* prob) { /* mcts.py:178-190 */
  int A = m->game->A;
  int idx = mcts_find(m, s);
  if (idx < 0) {
    if (m->n_nodes == m->cap_nodes) {
      m->cap_nodes *= 2;
      m->nodes = (onode*)realloc(m->nodes, sizeof(onode) * m->cap_nodes);
    }
    if ((m->n_nodes + 1) * 2 > m->n_slots) mcts_rehash(m);
    idx = m->n_nodes++;
    m->nodes[idx].key = *s;
    uint64_t h = state_hash(m->game, s);
    int i = (int)(h & (uint64_t)(m->n_slots - 1));
    while (m->slots[i] >= 0) i = (i + 1) & (m->n_slots - 1);
    m->slots[i] = idx;
  }
  onode* nd = &m->nodes[idx];
  for (int a = 0; a < A; ++a) {
    nd->N[a] = 0;
    nd->W[a] = py(0.0);
    nd->Q[a] = py(0.0);
    nd->P[a] = prob[a];
  }
}

/* ------------------------------------------------------ inputs / outputs */

/* Net callback: L states -> P (softmax already applied, float32 [L,A]) and
 * value (float32 [L]).  Mirrors _expand_tree's net + F.softmax, mcts.py:212-218. */
typedef void (*onet_fn)(void* ctx, int L, const float* planes, const uint8_t* cells,
                        const int32_t* players, float* P, float* v);

typedef struct {
  onet_fn fn[2]; /* net1, net2 */
  void* ctx[2];
  /* random inputs: explicit tables or the caro_noise.h generator */
  const double* noise_table; /* [n_rows, A] consumed in call order, or NULL */
  long noise_rows, noise_pos;
  const double* uniform_table; /* [n_plies] or NULL */
  long uniform_rows, uniform_pos;
  uint64_t seed, game_uid;
  double alpha, explore;
  /* counters */
  long sims, levels, expansions, terminals, dropped, net_calls, net_rows;
} oenv;

typedef struct {
  int n;
  ostate states[MAXCELLS + 1];
  int actions[MAXCELLS + 1];
} opath;

/* mcts.py:97-148.  `ply`/`sim` only key the generated noise. */
static void find_leaf(omcts* m, oenv* env, const ostate* root, int player, uint32_t ply,
                      uint32_t sim, int* has_value, pyval* value, ostate* leaf, int* leaf_player,
                      opath* path) {
  const ogame* g = m->game;
  int A = g->A;
  int hw = ncells(g);
  ostate cur = *root;
  int cur_player = player;
  *has_value = 0;
  path->n = 0;
  env->sims++;
  int idx;
  while ((idx = mcts_find(m, &cur)) >= 0) {
    onode* nd = &m->nodes[idx];
    path->states[path->n] = cur;
    env->levels++;
    int is_root = memcmp(cur.cells, root->cells, hw) == 0; /* cur_state == state_int :131 */
    double scores[MAXA];
    /* _calculate_upper_bound :64-84: total_sqrt = m.sqrt(sum(counts)) */
    long total = 0;
    for (int a = 0; a < A; ++a) total += nd->N[a];
    double total_sqrt = sqrt((double)total);
    if (is_root) {
      /* _add_noise :48-62 -> float64 probs; scores in float64 */
      double noise[MAXA], tmp[256];
      if (env->noise_table) {
        if (env->noise_pos >= env->noise_rows) { fprintf(stderr, "oracle: noise table exhausted\n"); abort(); }
        memcpy(noise, env->noise_table + env->noise_pos * A, sizeof(double) * A);
        env->noise_pos++;
      } else {
        caro_noise_row(env->seed, env->game_uid, ply, sim, A, env->alpha, noise, tmp);
      }
      for (int a = 0; a < A; ++a) {
        float keep = (float)(1.0 - env->explore) * nd->P[a]; /* py float * np.float32 -> float32 */
        double prob = (double)keep + env->explore * noise[a];  /* float32 + float64 -> float64 */
        double u = ((m->c_puct * prob) * total_sqrt) / (double)(1 + nd->N[a]);
        scores[a] = nd->Q[a].v + u; /* float32|pyfloat + float64 -> float64 */
      }
    } else {
      for (int a = 0; a < A; ++a) {
        float t = (float)m->c_puct * nd->P[a]; /* all float32 under NEP 50 */
        t = t * (float)total_sqrt;
        t = t / (float)(1 + nd->N[a]);
        float sc = (float)nd->Q[a].v + t;
        scores[a] = (double)sc;
      }
    }
    /* _mask_invalid_actions :86-95 */
    int legal[MAXA], nl = possible_moves(g, &cur, legal);
    uint8_t ok[MAXA];
    memset(ok, 0, sizeof ok);
    for (int i = 0; i < nl; ++i) ok[legal[i]] = 1;
    for (int a = 0; a < A; ++a) if (!ok[a]) scores[a] = -INFINITY;
    /* np.argmax: first maximum :136 */
    int action = 0;
    for (int a = 1; a < A; ++a) if (scores[a] > scores[action]) action = a;
    path->actions[path->n] = action;
    path->n++;
    int won = game_move(g, &cur, action, cur_player); /* :138 */
    if (won < 0) { fprintf(stderr, "oracle: illegal move selected\n"); abort(); }
    if (won) { *has_value = 1; *value = py(-1.0); } /* :140-142 */
    cur_player = 1 - cur_player;
    if (!*has_value) { /* :145-146 */
      int tmpm[MAXA];
      if (possible_moves(g, &cur, tmpm) == 0) { *has_value = 1; *value = py(0.0); }
    }
  }
  *leaf = cur;
  *leaf_player = cur_player;
}

static void backup(omcts* m, pyval value, const opath* path) { /* mcts.py:225-246 */
  pyval cur = pv_neg(value);
  for (int i = path->n - 1; i >= 0; --i) {
    int idx = mcts_find(m, &path->states[i]);
    onode* nd = &m->nodes[idx];
    int a = path->actions[i];
    nd->N[a] += 1;
    nd->W[a] = pv_add(nd->W[a], cur);
    nd->Q[a] = pv_div_int(nd->W[a], nd->N[a]);
    cur = pv_neg(cur);
  }
}

#define MAXB 64

/* mcts.py:248-287 */
static void search_minibatch(omcts* m, oenv* env, int which_net, int batch_size,
                             const ostate* root, int player, uint32_t ply, uint32_t sim0) {
  const ogame* g = m->game;
  int A = g->A, hw = ncells(g);
  static opath paths[MAXB];     /* per find_leaf call */
  pyval bq_value[2 * MAXB];
  int bq_path[2 * MAXB], nbq = 0;
  ostate ex_state[MAXB];
  int ex_player[MAXB], ex_path[MAXB], nex = 0;
  for (int b = 0; b < batch_size; ++b) {
    int has_value, leaf_player;
    pyval value;
    ostate leaf;
    find_leaf(m, env, root, player, ply, sim0 + (uint32_t)b, &has_value, &value, &leaf,
              &leaf_player, &paths[b]);
    if (has_value) {
      env->terminals++;
      bq_value[nbq] = value; bq_path[nbq] = b; nbq++;
    } else {
      int planned = 0;
      for (int i = 0; i < nex; ++i)
        if (memcmp(ex_state[i].cells, leaf.cells, hw) == 0) planned = 1;
      if (!planned) {
        ex_state[nex] = leaf; ex_player[nex] = leaf_player; ex_path[nex] = b; nex++;
      } else {
        env->dropped++;
      }
    }
  }
  if (nex) { /* _expand_tree :192-223 */
    float* planes = (float*)malloc(sizeof(float) * nex * 2 * hw);
    uint8_t* cells = (uint8_t*)malloc((size_t)nex * hw);
    int32_t players[MAXB];
    float* P = (float*)malloc(sizeof(float) * nex * A);
    float v[MAXB];
    for (int i = 0; i < nex; ++i) {
      encode_planes(g, &ex_state[i], ex_player[i], planes + (size_t)i * 2 * hw);
      memcpy(cells + (size_t)i * hw, ex_state[i].cells, hw);
      players[i] = ex_player[i];
    }
    env->fn[which_net](env->ctx[which_net], nex, planes, cells, players, P, v);
    env->net_calls++;
    env->net_rows += nex;
    for (int i = 0; i < nex; ++i) {
      mcts_create_node(m, &ex_state[i], P + (size_t)i * A);
      env->expansions++;
      bq_value[nbq] = f32v(v[i]); bq_path[nbq] = ex_path[i]; nbq++;
    }
    free(planes); free(cells); free(P);
  }
  for (int i = 0; i < nbq; ++i) backup(m, bq_value[i], &paths[bq_path[i]]);
}

/* mcts.py:289-313 */
static void get_policy(const omcts* m, const ostate* s, int tau, double* probs) {
  int A = m->game->A;
  int idx = mcts_find(m, s);
  const onode* nd = &m->nodes[idx];
  if (tau == 0) {
    int best = 0;
    for (int a = 1; a < A; ++a) if (nd->N[a] > nd->N[best]) best = a;
    for (int a = 0; a < A; ++a) probs[a] = 0.0;
    probs[best] = 1.0;
  } else {
    double total = 0.0;
    for (int a = 0; a < A; ++a) total += (double)nd->N[a]; /* count ** 1.0 summed as floats */
    for (int a = 0; a < A; ++a) probs[a] = (double)nd->N[a] / total;
  }
}

/* ------------------------------------------------------------ C interface */

typedef struct {
  ogame game;
  omcts stores[2];
  int n_stores;
  oenv env;
} oracle;

oracle* oracle_create(int kind, int n, int k, int n_stores, double c_puct, double alpha,
                      double explore) {
  oracle* o = (oracle*)calloc(1, sizeof(oracle));
  game_init(&o->game, kind, n, k);
  o->n_stores = n_stores;
  for (int i = 0; i < 2; ++i) mcts_init(&o->stores[i], &o->game, c_puct);
  o->env.alpha = alpha;
  o->env.explore = explore;
  return o;
}
void oracle_destroy(oracle* o) {
  for (int i = 0; i < 2; ++i) mcts_free(&o->stores[i]);
  free(o);
}
void oracle_clear(oracle* o) { for (int i = 0; i < 2; ++i) mcts_clear(&o->stores[i]); }
void oracle_set_net(oracle* o, int which, onet_fn fn, void* ctx) { o->env.fn[which] = fn; o->env.ctx[which] = ctx; }
void oracle_set_noise_table(oracle* o, const double* t, long rows) { o->env.noise_table = t; o->env.noise_rows = rows; o->env.noise_pos = 0; }
void oracle_set_uniform_table(oracle* o, const double* t, long rows) { o->env.uniform_table = t; o->env.uniform_rows = rows; o->env.uniform_pos = 0; }
void oracle_set_stream(oracle* o, uint64_t seed, uint64_t game_uid) { o->env.seed = seed; o->env.game_uid = game_uid; }
int oracle_action_space(const oracle* o) { return o->game.A; }
int oracle_store_len(const oracle* o, int store) { return o->stores[store].n_nodes; } /* mcts.py:45-46 */
long oracle_noise_pos(const oracle* o) { return o->env.noise_pos; }
void oracle_counters(const oracle* o, long* out) {
  out[0] = o->env.sims; out[1] = o->env.levels; out[2] = o->env.expansions;
  out[3] = o->env.terminals; out[4] = o->env.dropped; out[5] = o->env.net_calls; out[6] = o->env.net_rows;
}

/* single-state rule helpers (cells in the oracle layout) */
void oracle_initial_state(const oracle* o, uint8_t* cells) { ostate s; initial_state(&o->game, &s); memcpy(cells, s.cells, ncells(&o->game)); }
int oracle_move(const oracle* o, uint8_t* cells, int move, int player) {
  ostate s; memset(&s, EMPTY, sizeof s); memcpy(s.cells, cells, ncells(&o->game));
  int won = game_move(&o->game, &s, move, player);
  memcpy(cells, s.cells, ncells(&o->game));
  return won;
}
int oracle_possible_moves(const oracle* o, const uint8_t* cells, int32_t* out) {
  ostate s; memset(&s, EMPTY, sizeof s); memcpy(s.cells, cells, ncells(&o->game));
  int tmp[MAXA]; int n = possible_moves(&o->game, &s, tmp);
  for (int i = 0; i < n; ++i) out[i] = tmp[i];
  return n;
}
void oracle_encode_planes(const oracle* o, const uint8_t* cells, int who_move, float* dst) {
  ostate s; memset(&s, EMPTY, sizeof s); memcpy(s.cells, cells, ncells(&o->game));
  encode_planes(&o->game, &s, who_move, dst);
}
uint64_t oracle_c4_cells_to_int(const uint8_t* cells) { ostate s; memset(&s, EMPTY, sizeof s); memcpy(s.cells, cells, 42); return oracle_c4_encode(&s); }
void oracle_c4_int_to_cells(uint64_t v, uint8_t* cells) { ostate s; oracle_c4_decode(v, &s); memcpy(cells, s.cells, 42); }

/* search_batch, mcts.py:162-176 */
void oracle_search_batch(oracle* o, int store, int which_net, int count, int batch_size,
                         const uint8_t* cells, int player, uint32_t ply) {
  ostate root; memset(&root, EMPTY, sizeof root); memcpy(root.cells, cells, ncells(&o->game));
  for (int i = 0; i < count; ++i)
    search_minibatch(&o->stores[store], &o->env, which_net, batch_size, &root, player, ply,
                     (uint32_t)(i * batch_size));
}

/* node lookup for tests: returns 1 if present */
int oracle_get_node(const oracle* o, int store, const uint8_t* cells, int32_t* N, double* W,
                    int32_t* W_is_f32, double* Q, float* P) {
  ostate s; memset(&s, EMPTY, sizeof s); memcpy(s.cells, cells, ncells(&o->game));
  int idx = mcts_find(&o->stores[store], &s);
  if (idx < 0) return 0;
  const onode* nd = &o->stores[store].nodes[idx];
  for (int a = 0; a < o->game.A; ++a) {
    N[a] = nd->N[a]; W[a] = nd->W[a].v; W_is_f32[a] = nd->W[a].f32; Q[a] = nd->Q[a].v; P[a] = nd->P[a];
  }
  return 1;
}
void oracle_get_policy(const oracle* o, int store, const uint8_t* cells, int tau, double* probs) {
  ostate s; memset(&s, EMPTY, sizeof s); memcpy(s.cells, cells, ncells(&o->game));
  get_policy(&o->stores[store], &s, tau, probs);
}

/* poke a node + run one backup (restates lib/test_mcts.py's use of _backup) */
void oracle_poke_node(oracle* o, int store, const uint8_t* cells, const int32_t* N, const double* W,
                      const double* Q, const float* P) {
  ostate s; memset(&s, EMPTY, sizeof s); memcpy(s.cells, cells, ncells(&o->game));
  mcts_create_node(&o->stores[store], &s, P);
  onode* nd = &o->stores[store].nodes[mcts_find(&o->stores[store], &s)];
  for (int a = 0; a < o->game.A; ++a) { nd->N[a] = N[a]; nd->W[a] = py(W[a]); nd->Q[a] = py(Q[a]); }
}
void oracle_backup(oracle* o, int store, double value, int value_is_f32, int n, const uint8_t* path_cells,
                   const int32_t* actions) {
  static opath p;
  p.n = n;
  int hw = ncells(&o->game);
  for (int i = 0; i < n; ++i) {
    memset(&p.states[i], EMPTY, sizeof(ostate));
    memcpy(p.states[i].cells, path_cells + (size_t)i * hw, hw);
    p.actions[i] = actions[i];
  }
  backup(&o->stores[store], value_is_f32 ? f32v((float)value) : py(value), &p);
}

/* play_game, lib/utils.py:25-108.  Stores are cleared first (fresh per game,
 * SURVEY Q3).  n_stores == 1: one tree shared by both players (utils.py:60-61);
 * 2: one per player (utils.py:58-59).
 * History outputs (one row per ply, forward order): cells, player, pi[A],
 * chosen action, root N after search, node count of the mover's store.
 * Returns net1_result (+1/0/-1); *out_steps = `step` as returned by the
 * reference; *out_plies = len(game_history). */
int oracle_play_game(oracle* o, int steps_before_tau_0, int searches, int batch_size,
                     int first_player, int max_plies, uint8_t* h_cells, int32_t* h_player,
                     double* h_pi, int32_t* h_action, int32_t* h_rootN, int32_t* h_nodes,
                     int32_t* h_z, int32_t* out_steps, int32_t* out_plies) {
  const ogame* g = &o->game;
  int A = g->A, hw = ncells(g);
  oracle_clear(o);
  ostate state;
  initial_state(g, &state);
  int cur_player = first_player; /* 0 if net1_plays_first else 1, utils.py:65-68 */
  int step = 0;
  int tau = steps_before_tau_0 > 0 ? 1 : 0; /* :70 */
  int plies = 0;
  int result = -2, net1_result = 0;
  while (result == -2) {
    int st = o->n_stores == 2 ? cur_player : 0;
    omcts* m = &o->stores[st];
    for (int i = 0; i < searches; ++i) /* search_batch :77-79 */
      search_minibatch(m, &o->env, cur_player, batch_size, &state, cur_player, (uint32_t)plies,
                       (uint32_t)(i * batch_size));
    double probs[MAXA];
    get_policy(m, &state, tau, probs); /* :80-81 */
    if (plies >= max_plies) { fprintf(stderr, "oracle: history overflow\n"); abort(); }
    memcpy(h_cells + (size_t)plies * hw, state.cells, hw);
    h_player[plies] = cur_player;
    memcpy(h_pi + (size_t)plies * A, probs, sizeof(double) * A);
    {
      const onode* nd = &m->nodes[mcts_find(m, &state)];
      for (int a = 0; a < A; ++a) h_rootN[(size_t)plies * A + a] = nd->N[a];
      h_nodes[plies] = m->n_nodes;
    }
    double u;
    if (o->env.uniform_table) {
      if (o->env.uniform_pos >= o->env.uniform_rows) { fprintf(stderr, "oracle: uniform table exhausted\n"); abort(); }
      u = o->env.uniform_table[o->env.uniform_pos++];
    } else {
      u = caro_move_uniform(o->env.seed, o->env.game_uid, (uint32_t)plies);
    }
    int action = caro_sample_index(probs, A, u); /* np.random.choice(A, p=probs) :83 */
    h_action[plies] = action;
    plies++;
    int won = game_move(g, &state, action, cur_player); /* :86 */
    if (won < 0) { fprintf(stderr, "oracle: impossible action\n"); abort(); }
    if (won) { result = 1; net1_result = cur_player == 0 ? 1 : -1; break; } /* :87-90 */
    cur_player = 1 - cur_player;
    int tmpm[MAXA];
    if (possible_moves(g, &state, tmpm) == 0) { result = 0; net1_result = 0; break; } /* :93-96 */
    step++;
    if (step >= steps_before_tau_0) tau = 0; /* :97-99 */
  }
  /* replay: reversed history, result alternates, :101-106 */
  int r = result;
  for (int i = plies - 1; i >= 0; --i) { h_z[i] = r; r = -r; }
  *out_steps = step;
  *out_plies = plies;
  return net1_result;
}

/* ------------------------------------------- synthetic hash net (tests) */
/* A deterministic "net" whose P and v are exact dyadic float32 values of an
 * integer hash of the input planes, so that CPU and GPU sides can evaluate it
 * with identical bits (tests/synth_net.py is the torch twin; include/caro_hip.h
 * states the definition for the engine's table evaluator).  `salt` (0 in every
 * vector recorded from the reference) tells the two nets of an arena apart. */
static uint64_t synth_coef(int i) { return caro_mix64(0x5851f42d4c957f2dULL + (uint64_t)i) | 1ULL; }

typedef struct { const oracle* o; uint64_t salt; } synth_ctx;
static synth_ctx g_synth[2][64]; /* per net; a small pool so that several oracles can coexist in one process */
static int g_synth_used = 0;

static void synth_eval(const oracle* o, uint64_t salt, int L, const float* planes, float* P, float* v) {
  int A = o->game.A, hw2 = 2 * ncells(&o->game);
  for (int i = 0; i < L; ++i) {
    uint64_t h = 0;
    for (int j = 0; j < hw2; ++j)
      if (planes[(size_t)i * hw2 + j] != 0.0f) h += synth_coef(j);
    h += salt;
    for (int a = 0; a < A; ++a) {
      uint64_t ha = caro_mix64(h + 0x9E3779B97F4A7C15ULL * (uint64_t)(a + 1));
      P[(size_t)i * A + a] = (float)(((ha >> 20) & 1023ULL) + 1ULL) / 8192.0f;
    }
    uint64_t hv = caro_mix64(h ^ 0xA5A5A5A5A5A5A5A5ULL);
    v[i] = (float)((long long)((hv >> 20) % 2001ULL) - 1000LL) / 1024.0f;
  }
}
void oracle_synth_net(void* ctx, int L, const float* planes, const uint8_t* cells,
                      const int32_t* players, float* P, float* v) {
  (void)cells; (void)players;
  synth_eval((const oracle*)ctx, 0, L, planes, P, v);
}
static void synth_net_salted(void* ctx, int L, const float* planes, const uint8_t* cells,
                             const int32_t* players, float* P, float* v) {
  const synth_ctx* c = (const synth_ctx*)ctx;
  (void)cells; (void)players;
  synth_eval(c->o, c->salt, L, planes, P, v);
}
/* direct evaluation of the salted table net (twin checks in tests) */
void oracle_synth_eval(const oracle* o, uint64_t salt, int L, const float* planes, float* P, float* v) {
  synth_eval(o, salt, L, planes, P, v);
}
void oracle_use_synth_net(oracle* o) { oracle_set_net(o, 0, oracle_synth_net, o); oracle_set_net(o, 1, oracle_synth_net, o); }
/* net 0 / net 1 = the table net with salt0 / salt1 */
void oracle_use_synth_nets(oracle* o, uint64_t salt0, uint64_t salt1) {
  int k = g_synth_used++ % 64;
  g_synth[0][k].o = o; g_synth[0][k].salt = salt0;
  g_synth[1][k].o = o; g_synth[1][k].salt = salt1;
  oracle_set_net(o, 0, synth_net_salted, &g_synth[0][k]);
  oracle_set_net(o, 1, synth_net_salted, &g_synth[1][k]);
}

/* noise helpers exported for tests (host form of the public spec) */
void oracle_noise_row(uint64_t seed, uint64_t uid, uint32_t ply, uint32_t sim, int A, double alpha, double* out) {
  double tmp[256];
  caro_noise_row(seed, uid, ply, sim, A, alpha, out, tmp);
}
double oracle_move_uniform(uint64_t seed, uint64_t uid, uint32_t ply) { return caro_move_uniform(seed, uid, ply); }
int oracle_sample_index(const double* pi, int A, double u) { return caro_sample_index(pi, A, u); }
