// caro_engine.hip -- MI355X (gfx950) self-play engine: kernels + C-ABI.
//
// What runs where (reference file:line in brackets):
//   k_select         B find_leaf descents per game on the frozen tree
//                    [lib/mcts.py:97-148,64-95,48-62] + leaf de-duplication
//                    [lib/mcts.py:265-278].  One workgroup per game, one group
//                    of LPD lanes per descent, lanes = actions; PUCT argmax is
//                    a xor-butterfly over (score, action) inside the group.
//   k_encode         deterministic compaction of the unique leaves into dense net rows (each block sums the
//                    counts of the games before it) + NN planes [game.states_to_training_batch].
//   k_expand_backup  _create_node + ordered _backup [lib/mcts.py:178-190,225-246,281-287].
//   k_step           get_policy_value + one ply of play_game
//                    [lib/mcts.py:289-313, lib/utils.py:80-99].
//   k_drain_*        replay emission [lib/utils.py:101-106] + slot recycling.
//   k_tree           the three per-game bodies fused (expand + backup of minibatch i-1, select of minibatch i, NN
//                    planes into the game's slot rows): one launch per minibatch beside the net kernel, lock-step.
//   k_tree_stag      the same with every game on its OWN minibatch clock (staggered mode): the ply (step_body) and
//                    the restart of a finished game's slot happen inside the kernel, so every launch carries the same
//                    mix of minibatch indices and the net launch the same leaf count.  Both fused kernels run a
//                    second wavefront per game that generates the Dirichlet rows while the tree wave waits for memory.
//
// Data layout in HBM (T = G * n_stores trees, AP = padded action count).  The transposition table IS
// the node store: open addressing with linear probing over hcap = 2^k >= 2*cap slots, node id = slot,
// so one descent level costs ONE memory latency (key and action rows of the home slot load together).
//   node_key u64 [T][hcap][KW]      board of the node in the slot (the key); word 0 == ~0 marks an empty slot
//   edges    u32x4 [T][hcap][AP]    per node and action one 16-byte record {N, W, Q, P} (array of structs, round 4: a lane
//                                   loads its action with ONE 16-byte load and an edge update dirties one cache line;
//                                   connect four: 128 bytes per node); N carries the "strong" flag in bit 30 (W has
//                                   absorbed a float32 value).
// Arithmetic: non-root PUCT in float32, root PUCT in float64, no FMA
// contraction (compiled with -ffp-contract=off), exactly the order of
// lib/mcts.py:79-84 under numpy>=2 scalar promotion.
#include <hip/hip_runtime.h>

#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <type_traits>
#include <vector>

#include "../../include/caro_hip.h"
#include "../../include/caro_noise.h"
#include "caro_rules.h"
#include "caro_variants.h"

namespace caro {

constexpr uint32_t NSTRONG = 1u << 30;
constexpr uint32_t NMASK = NSTRONG - 1u;
constexpr int ST_DROPPED = 0, ST_TERMINAL = 1, ST_LEAF = 2;
constexpr int MAXB = 64;

enum Counter { C_SIMS, C_LEVELS, C_EXPANSIONS, C_TERMINALS, C_DROPPED, C_OVERFLOW, C_PLIES, C_FINISHED, C_N };

struct View {
  GameParams gp;
  int G, n_stores, n_nets, cap, hcap, A, HW, maxply, maxd, maxB, sbt0, first_mode, ntab, etab;
  int tstride;        // slots from one key / row table to the next: hcap + a skew (tables are not 2^k apart in memory)
  uint32_t slot_rot;  // tree t's home slots are rotated by t * slot_rot (0: off): the same board sits in a different
                      // slot -- another L2 set / channel -- of every tree.  Slot ids carry no meaning: result-neutral
  float c_puct;
  double alpha, explore;
  uint64_t seed, uid_base, uid_stride;
  long long games_limit;  // > 0: slot g plays its k-th game only while k * G + g < games_limit (caro_config.games_limit)
  // trees (ntab == 2: two tables per tree, tbl[t] is the live one; see k_evict)
  int32_t* tbl;
  uint64_t* node_key;
  uint32_t* edges;
  int32_t* n_nodes;
  int32_t* n_created;
  // games
  uint64_t* root;
  int32_t* player;
  int32_t* ply;
  int32_t* step;
  uint64_t* uid;
  int32_t* done;    // 0 live, 1 finished (awaiting drain)
  int32_t* result;  // net1_result
  int32_t* final_r; // 1 win of the last mover, 0 draw
  int32_t* first;
  // history
  uint64_t* h_key;
  int32_t* h_player;
  double* h_pi;
  // minibatch scratch: what select leaves behind for expand + backup
  //   d_rec    [G][maxB]        per descent: x = status | path length << 8 | leaf rank << 16 | player to move << 24,
  //                             y = terminal value (float bits), z = home slot of the leaf board | bit 31 if that slot
  //                             was empty when the descent ended
  //   path_rec [G][maxB][maxd]  per level of a descent: x = node slot | action << 24, y / z = the N word (visit count +
  //                             strong flag) and the W word of the chosen edge AS SELECT SAW THEM.  Nothing touches a
  //                             tree between a minibatch's descents and its backup, so the backup starts from these
  //                             values instead of loading the edges again (one dependent memory round less per block)
  uint4* d_rec;
  uint4* path_rec;
  uint64_t* d_key;
  int32_t* g_nleaf;
  int32_t* g_off;
  int32_t* g_tree;
  int32_t* g_class;
  int32_t* g_pack;      // fused form: unique-leaf count | net class << 8 (what the net kernel's row map reads)
  int32_t* slot_list;   // [2][G * maxB] multi-wave fused form: the slot rows of the launch's leaves per net class, in the
                        // order the blocks got there (caro_net_forward_slot_list)
  int32_t* leaf_count;  // [4]: L0, L1, batch of the pending minibatch, -
  unsigned long long* counters;  // [G][C_N] per-game tallies (no atomics on the hot path), summed on read
  unsigned long long* counters_sum;  // [C_N]
  unsigned long long* dbg;           // diagnostic stamps [G][8] or null (never set in product runs)
  // staggered mode (caro_stagger_enable; k_tree_stag): every game runs its own minibatch clock
  int stag_S;           // minibatches per move (0: lock-step engine)
  int stag_recycle;     // finished games restart in-kernel (uid += uid_stride)
  int stag_pool;        // caro_config.stagger_recycle == 2 (with games_limit): a finished slot does not restart in-kernel
                        // with ITS next uid but waits (done = 2) for the next drain, where k_stag_assign hands the free
                        // slots the next games not started yet, in slot order (deterministic): the slots stay busy until
                        // the wanted games run out, whatever the lengths of the games a slot happened to get
  int32_t* handed;      // [1] staggered pool mode: local indices of the wanted set handed out so far
  int32_t* lm;          // [G] index of the game's next minibatch, 0..stag_S (== stag_S: the move is due)
  int32_t* pend;        // [G] 1: a selected minibatch awaits its expand + backup
  int32_t* wait;        // [G] launches the game still sits out before its first search (the initial stagger)
  int32_t* dirty;       // [T] 1: the tree's OTHER key table holds the keys of a finished game (cleared at the next drain)
  // a finished game is PARKED (its record and history copied aside) so that the slot restarts at once; the parked
  // rows wait for the next drain.  Same field meaning as the live arrays they are copied from.
  int32_t* pk_flag;     // [G] 1 parked and not drained yet, otherwise free
  int32_t* pk_ply; int32_t* pk_final_r; int32_t* pk_first; int32_t* pk_result; int32_t* pk_step;
  uint64_t* pk_uid;
  uint64_t* ph_key; int32_t* ph_player; double* ph_pi;
  // drain scratch
  int32_t* dr_off;
  int32_t* dr_gidx;
  int32_t* dr_sel;
  int64_t* dr_tot;  // [2] tuples, games
};

// ------------------------------------------------------------------ device helpers
template <class R>
__device__ __forceinline__ typename R::Board load_board(const uint64_t* p) {
  typename R::Board b;
#pragma unroll
  for (int i = 0; i < R::KW; ++i) b.w[i] = p[i];
  return b;
}
template <class R>
__device__ __forceinline__ void store_board(uint64_t* p, const typename R::Board& b) {
#pragma unroll
  for (int i = 0; i < R::KW; ++i) p[i] = b.w[i];
}

// first slot of tree t's live table
__device__ __forceinline__ size_t tbase(const View& v, int t) {
  return (size_t)(t * v.ntab + (v.ntab == 2 ? v.tbl[t] : 0)) * (size_t)v.tstride;
}
// first row of tree t's action rows: they follow the key table only where a second copy exists (etab == 2:
// eviction moves the survivors' rows; a staggered restart needs a clean KEY table, the rows are rewritten anyway)
__device__ __forceinline__ size_t ebase(const View& v, int t) {
  return (size_t)(t * v.etab + (v.etab == 2 ? v.tbl[t] : 0)) * (size_t)v.tstride;
}

constexpr uint64_t EMPTY_KEY = ~0ULL;  // no board has bit 63 set (C4) / overlapping planes (m,n,k)

template <class R>
__device__ __forceinline__ uint32_t home_slot(const View& v, int t, const typename R::Board& b) {
  return ((uint32_t)R::hash(b) + (uint32_t)t * v.slot_rot) & ((uint32_t)v.hcap - 1u);
}

// Everything the per-game kernels need to know about game g that depends on g alone, loaded in ONE round of
// independent loads at the top of a kernel (a block is a chain of dependent memory latencies: done -> root / player
// -> table selector -> first row were four of them).  The fused kernels carry it in registers from the backup
// through the ply to the descents; whoever changes the game (step_body, park_and_restart) updates both copies.
template <class GEO>
struct GameRegs {
  typename GEO::R::Board root;
  int done, player, ply, step;
  int tbl[2], nn[2];  // per store: live key table (0 unless a second one exists), nodes held
  uint64_t uid;
};
template <class GEO>
__device__ __forceinline__ GameRegs<GEO> load_game(const View& v, int g) {
  using R = typename GEO::R;
  GameRegs<GEO> r;
  r.done = v.done[g];
  r.player = v.player[g];
  r.ply = v.ply[g];
  r.step = v.step[g];
  r.uid = v.uid[g];
  r.root = load_board<R>(v.root + (size_t)g * GEO::KW);
#pragma unroll
  for (int st = 0; st < 2; ++st) {
    const int t = g * v.n_stores + (st < v.n_stores ? st : 0);
    r.tbl[st] = v.ntab == 2 ? v.tbl[t] : 0;
    r.nn[st] = v.n_nodes[t];
  }
  return r;
}
// tbase / ebase with the table selector in hand
__device__ __forceinline__ size_t tbase_sel(const View& v, int t, int sel) {
  return (size_t)(t * v.ntab + sel) * (size_t)v.tstride;
}
__device__ __forceinline__ size_t ebase_sel(const View& v, int t, int sel) {
  return (size_t)(t * v.etab + (v.etab == 2 ? sel : 0)) * (size_t)v.tstride;
}

// Synchronisation inside a per-game block.  ONE = the block's tree work is done by ONE wavefront (the fused kernels):
// its lanes run in lock-step, so all that is needed is that earlier memory operations have completed -- no
// s_barrier, which would also wait for the block's noise wave (k_tree / k_tree_stag) to finish.
// Nor a wait: the wave's memory instructions are issued in program order and the hardware keeps accesses of one
// wavefront to the same address in order (LDS: one in-order queue; global: the same path for every lane), which is
// all a block-local hand-over between lanes of the SAME wave needs -- so nothing stalls until data is really used.
template <bool ONE>
__device__ __forceinline__ void block_sync() {
  if constexpr (ONE) {
    asm volatile("" ::: "memory");  // the compiler keeps the order of the memory operations around this point
    __builtin_amdgcn_wave_barrier();
  } else {
    __syncthreads();
  }
}
template <bool ONE>
__device__ __forceinline__ int block_threads() { return ONE ? 64 : (int)blockDim.x; }

// `state in self.probs` (lib/mcts.py:160): probe of tree t, returns the node's slot or -1
template <class R>
__device__ __forceinline__ int probe_from(const View& v, int t, const typename R::Board& b, uint32_t i) {
  const uint32_t mask = (uint32_t)v.hcap - 1u;
  const uint64_t* keys = v.node_key + tbase(v, t) * R::KW;
  for (int it = 0; it < v.hcap; ++it) {
    const uint64_t* k = keys + (size_t)i * R::KW;
    const uint64_t k0 = k[0];
    if (k0 == EMPTY_KEY) return -1;
    bool eq = k0 == b.w[0];
#pragma unroll
    for (int w = 1; w < R::KW; ++w) eq = eq && (k[w] == b.w[w]);
    if (eq) return (int)i;
    i = (i + 1u) & mask;
  }
  return -1;
}
template <class R>
__device__ __forceinline__ int probe(const View& v, int t, const typename R::Board& b) {
  return probe_from<R>(v, t, b, home_slot<R>(v, t, b));
}

// first free slot from the board's home slot; writes the key; returns the slot (-1: table full)
template <class R>
__device__ __forceinline__ int insert_key(const View& v, int t, const typename R::Board& b) {
  const uint32_t mask = (uint32_t)v.hcap - 1u;
  uint32_t i = home_slot<R>(v, t, b);
  uint64_t* keys = v.node_key + tbase(v, t) * R::KW;
  for (int it = 0; it < v.hcap; ++it) {
    uint64_t* k = keys + (size_t)i * R::KW;
    if (k[0] == EMPTY_KEY) {
#pragma unroll
      for (int w = R::KW - 1; w >= 0; --w) k[w] = b.w[w];
      return (int)i;
    }
    i = (i + 1u) & mask;
  }
  return -1;
}

template <int LPD>
__device__ __forceinline__ double group_sum_f64(double x) {
#pragma unroll
  for (int m = 1; m < LPD; m <<= 1) x = x + __shfl_xor(x, m, LPD);
  return x;
}

// Cross-lane moves inside a descent group on the DPP path (no LDS round trip): xor 1 / xor 2 inside a quad, then
// the mirror of 8 and of 16 lanes.  After the quad steps every lane of a quad holds the quad's result, so the
// mirrors pair lanes of different quads / different halves exactly as an xor-butterfly would: the all-reduce of
// a commutative operation comes out the same in every lane.  Steps of 32 / 64 lanes go through __shfl_xor.
template <int CTRL>
__device__ __forceinline__ int dpp_i32(int x) {
  return __builtin_amdgcn_update_dpp(0, x, CTRL, 0xF, 0xF, true);
}
template <int LPD, class OP>
__device__ __forceinline__ int group_allreduce_i32(int x, OP op) {
  if constexpr (LPD >= 2) x = op(x, dpp_i32<0xB1>(x));   // quad_perm [1,0,3,2]
  if constexpr (LPD >= 4) x = op(x, dpp_i32<0x4E>(x));   // quad_perm [2,3,0,1]
  if constexpr (LPD >= 8) x = op(x, dpp_i32<0x141>(x));  // row_half_mirror
  if constexpr (LPD >= 16) x = op(x, dpp_i32<0x140>(x)); // row_mirror
  if constexpr (LPD >= 32) x = op(x, __shfl_xor(x, 16, 64));
  if constexpr (LPD >= 64) x = op(x, __shfl_xor(x, 32, 64));
  return x;
}
template <int LPD>
__device__ __forceinline__ int group_sum_i32(int x) {
  return group_allreduce_i32<LPD>(x, [](int a, int b) { return a + b; });
}
template <int LPD>
__device__ __forceinline__ uint32_t group_max_u32(uint32_t x) {
  return (uint32_t)group_allreduce_i32<LPD>((int)x, [](int a, int b) { return (int)max((uint32_t)a, (uint32_t)b); });
}
// float -> uint32 with the same order (-inf lowest); -0.0 must not reach it (callers add +0.0f first)
// sqrtf of a visit count: x is an integer in [0, 2^24].  v_sqrt_f32 is within one ulp; the two residuals pick the
// correctly rounded neighbour -- the fix-up sqrtf itself expands to, without its scaling of tiny / huge arguments and
// its zero / infinity class test (x = 0: both residual tests fail on NaN / zero and 0 stays).  Equal to sqrtf on every
// such x (caro_debug_sqrt_check compares them all; tests/test_gpu_engine.py).
__device__ __forceinline__ float sqrt_count(float x) {
  float s = __builtin_amdgcn_sqrtf(x);
  const float sd = __uint_as_float(__float_as_uint(s) - 1u), su = __uint_as_float(__float_as_uint(s) + 1u);
  const float rd = __builtin_fmaf(-sd, s, x), ru = __builtin_fmaf(-su, s, x);
  s = rd <= 0.0f ? sd : s;
  s = ru > 0.0f ? su : s;
  return s;
}
__global__ void k_sqrt_check(uint32_t n, unsigned long long* bad) {
  for (uint32_t i = blockIdx.x * blockDim.x + threadIdx.x; i <= n; i += gridDim.x * blockDim.x)
    if (__float_as_uint(sqrt_count((float)i)) != __float_as_uint(sqrtf((float)i))) atomicAdd(bad, 1ull);
}

__device__ __forceinline__ uint32_t orderable(float f) {
  const uint32_t u = __float_as_uint(f);
  return u ^ ((uint32_t)((int32_t)u >> 31) | 0x80000000u);
}

// The gammas of ONE Dirichlet row by the 64 lanes of a wavefront (the geometries with several actions per lane: one
// wave = one descent = one row), same values as caro_gamma_small action by action.  Three things cost the plain form
// (every lane calls caro_gamma_small for its APL actions in turn) most of its time, all of them SIMT effects:
//   * the rejection loop of an action runs until the last of the 64 lanes has accepted (3-4 rounds against 1.3 on
//     average), for each of the APL actions anew;
//   * the full acceptance test (two caro_log) runs for the whole wave whenever ONE lane's candidate misses the squeeze --
//     8 % of the candidates, i.e. nearly every round;
//   * the U^(1/alpha) boost (a log, a division, an exp) sits inside that loop.
// Here the row's actions are a POOL: a lane without work takes the next action not handed out yet, makes candidates for
// it (the action's own draws j = 0, 1, 2, ... in order: a candidate is a pure function of (key, action, j)), and puts the
// accepted (v, j) into the row's LDS mailbox; lanes whose candidate misses the squeeze WAIT with it until a dozen have
// gathered (or nothing else is left to do) and take the full test together.  The wave then runs ~ (A x 1.4 / 64 + a short
// tail) rounds of candidates, 2-3 full tests, and the boosts as straight-line code per lane afterwards.
// mail_v / mail_k: AP entries of LDS owned by this wave: the accepted v of every action and the key of the draw that
// follows its last candidate (the boost's uniform).
template <int APL>
__device__ __forceinline__ void gamma_row_pool(uint64_t key, int lane, int A, double alpha, double* mail_v,
                                               uint64_t* mail_k, double* g) {
  constexpr uint64_t KA = 0x9E3779B97F4A7C15ULL, KJ = 0xD1B54A32D192ED03ULL;
  constexpr int FULL_BATCH = 12;
  const double d = (alpha + 1.0) - 1.0 / 3.0;
  const double c = 1.0 / caro_sqrt(9.0 * d);
  int next = 0, remaining = A;  // uniform: first action not handed out yet, actions without a result
  int a = -1;                   // this lane's action (-1: none)
  uint32_t j = 0;
  uint64_t ka = 0;
  bool pend = false;            // a candidate waits for the full test
  double px = 0.0, pt3 = 0.0, pu = 0.0;
  while (remaining > 0) {
    if (next < A) {  // hand out actions to the lanes without one, in lane order
      const unsigned long long need = __ballot(a < 0);
      const int rank = __popcll(need & ((1ull << lane) - 1ull));
      if (a < 0 && next + rank < A) {
        a = next + rank;
        j = 0;
        ka = caro_mix64(key + KA * (uint64_t)(a + 1));  // from here on ka = the action's key + KJ * j (64-bit multiplies
      }                                                 // are four quarter-rate instructions each: added up instead)
      next += __popcll(need);
      next = next < A ? next : A;
    }
    bool acc = false;
    double vacc = 1.0;  // (the draw bound was hit: v stays 1.0, as in caro_gamma_small)
    if (a >= 0 && !pend) {
      if (j + 4 <= CARO_NOISE_MAX_DRAWS) {
        const double u1 = 2.0 * caro_u01(caro_mix64(ka)) - 1.0;
        const double u2 = 2.0 * caro_u01(caro_mix64(ka + KJ)) - 1.0;
        j += 2;
        ka += 2ull * KJ;
        const double s = u1 * u1 + u2 * u2;
        if (s > 0.0 && s < 1.0) {
          const double x = u1 * caro_sqrt((-2.0 * caro_log(s)) / s);
          const double t = 1.0 + c * x;
          if (t > 0.0) {
            const double t3 = (t * t) * t;
            const double u = caro_u01(caro_mix64(ka));
            j += 1;
            ka += KJ;
            if (caro_squeeze_accepts(u, x)) {
              acc = true;
              vacc = t3;
            } else {
              pend = true;
              px = x; pt3 = t3; pu = u;
            }
          }
        }
      } else {
        acc = true;
      }
    }
    const unsigned long long pm = __ballot(pend);
    if (pm) {
      const bool more = next < A || __ballot(a >= 0 && !pend && !acc) != 0ull;  // candidates still to be made elsewhere
      if (__popcll(pm) >= FULL_BATCH || !more) {
        if (pend) {
          if (caro_log(pu) < (0.5 * px) * px + d * ((1.0 - pt3) + caro_log(pt3))) {
            acc = true;
            vacc = pt3;
          }
          pend = false;
        }
      }
    }
    if (acc) {
      mail_v[a] = vacc;
      mail_k[a] = ka;  // = the action's key + KJ * j: the key of draw j, which the boost takes
      a = -1;
    }
    remaining -= __popcll(__ballot(acc));
  }
  asm volatile("" ::: "memory");  // one wave: its LDS accesses complete in program order
  __builtin_amdgcn_wave_barrier();
#pragma unroll
  for (int i = 0; i < APL; ++i) {
    const int ai = lane * APL + i;
    double gi = 0.0;
    if (ai < A) {
      const double ub = caro_u01(caro_mix64(mail_k[ai]));
      gi = (d * mail_v[ai]) * caro_exp(caro_log(ub) / alpha);
    }
    g[i] = gi;
  }
  asm volatile("" ::: "memory");
  __builtin_amdgcn_wave_barrier();  // the mailbox may be written again (the wave's next row)
}

// One Dirichlet row of caro_noise.h for the LPD lanes of a descent group:
// lane l holds actions l*APL .. l*APL+APL-1.  Per-lane adjacent tree, then the
// xor butterfly: the balanced adjacent-pair tree sum of the spec.
// `mail`: LDS of the calling wave for gamma_row_pool (APL > 1 only): AP doubles followed by AP uint64
template <int LPD, int APL>
__device__ __forceinline__ void noise_group(uint64_t key, int l, int A, double alpha, double* out, double* mail = nullptr) {
  double g[APL];
  if constexpr (APL == 1) {
    g[0] = l < A ? caro_gamma_small(key, (uint32_t)l, alpha) : 0.0;
  } else {
    static_assert(LPD == 64, "several actions per lane: one wavefront per row");
    gamma_row_pool<APL>(key, l, A, alpha, mail, reinterpret_cast<uint64_t*>(mail + LPD * APL), g);
  }
  double s;
  if constexpr (APL == 1) s = g[0];
  else if constexpr (APL == 2) s = g[0] + g[1];
  else s = (g[0] + g[1]) + (g[2] + g[3]);
  s = group_sum_f64<LPD>(s);
#pragma unroll
  for (int j = 0; j < APL; ++j) out[j] = g[j] / s;
}

// ------------------------------------------------------------------ select
// State of one descent; every lane of the descent's group holds the same copy.
template <class GEO>
struct Descent {
  typename GEO::R::Board cur;
  typename GEO::R::Aux aux;
  int player, depth, status;
  float value;
  uint32_t home;  // where the descent ended outside the tree: home slot of that board | bit 31 if the slot is empty
  typename GEO::R::LaneK lk;  // the lane's constants of the win test (move_group)
};

// One level of find_leaf (lib/mcts.py:123-147) for the descents of a wave.  ROOT: the level at the game's root
// (_add_noise, float64 scores, mcts.py:131-132); otherwise float32 scores.  Returns false when the descent has
// ended (state not in the tree = the leaf, a win, or a full board).  Everything a group needs from its other lanes
// travels by DPP / ballot; the only memory access is the node's row.
// What a level reads from memory: the key in the board's home slot and this lane's share of the action rows
// N | W | Q | P of that slot (W only at the root), all issued together: one latency per level.
template <class GEO>
struct NodeRow {
  uint32_t slot;
  uint32_t nraw[GEO::APL], wraw[GEO::APL];
  float q[GEO::APL], p[GEO::APL];
  uint64_t k[GEO::KW];
};
template <class GEO, bool WITH_W>
__device__ __forceinline__ void load_row(NodeRow<GEO>& r, const uint64_t* __restrict__ tkeys,
                                         const uint32_t* __restrict__ tedges, uint32_t slot, int l) {
  constexpr int APL = GEO::APL, AP = GEO::AP, KW = GEO::KW;
  const uint4* row = reinterpret_cast<const uint4*>(tedges + (size_t)slot * 4 * AP);
  r.slot = slot;
#pragma unroll
  for (int j = 0; j < APL; ++j) {
    const uint4 e = row[l * APL + j];  // {N, W, Q, P} of the lane's action: one 16-byte load
    r.nraw[j] = e.x;
    r.wraw[j] = e.y;
    r.q[j] = __uint_as_float(e.z);
    r.p[j] = __uint_as_float(e.w);
  }
  const uint64_t* k = tkeys + (size_t)slot * KW;
#pragma unroll
  for (int w = 0; w < KW; ++w) r.k[w] = k[w];
}

template <class GEO, bool ROOT>
__device__ __forceinline__ bool descend_level(const View& v, Descent<GEO>& d, int t, const uint64_t* __restrict__ tkeys,
                                              const uint32_t* __restrict__ tedges, uint4* __restrict__ prec,
                                              uint4* lprec, int l, int first, const double* nz, NodeRow<GEO>& r) {
  using R = typename GEO::R;
  constexpr int LPD = GEO::LPD, APL = GEO::APL, KW = GEO::KW;
  if (!ROOT) load_row<GEO, true>(r, tkeys, tedges, home_slot<R>(v, t, d.cur), l);  // the root's row is loaded by the caller
  uint32_t(&nraw)[APL] = r.nraw;
  uint32_t(&wraw)[APL] = r.wraw;
  float(&q)[APL] = r.q;
  float(&p)[APL] = r.p;
  int node;
  {
    bool eq = r.k[0] == d.cur.w[0];
#pragma unroll
    for (int w = 1; w < KW; ++w) eq = eq && (r.k[w] == d.cur.w[w]);
    if (eq) node = (int)r.slot;
    else if (r.k[0] == EMPTY_KEY) node = -1;
    else {  // collision with another board: walk the probe sequence, then reload the rows
      node = probe_from<R>(v, t, d.cur, (r.slot + 1u) & ((uint32_t)v.hcap - 1u));
      if (node >= 0) load_row<GEO, true>(r, tkeys, tedges, (uint32_t)node, l);
    }
  }
  if (node < 0) {  // not in the tree: this is the leaf (mcts.py:123)
    // r.slot is still the board's home slot: expand_body takes it (and whether it is empty) from here instead of
    // looking again -- nothing touches the tree between a minibatch's descents and its expansion
    d.home = r.slot | (r.k[0] == EMPTY_KEY ? 0x80000000u : 0u);
    return false;
  }
  int nsum = 0;
#pragma unroll
  for (int j = 0; j < APL; ++j) nsum += (int)(nraw[j] & NMASK);
  nsum = group_sum_i32<LPD>(nsum);
  int besta;
  if (ROOT) {
    // _add_noise (mcts.py:48-62) -> float64 probs, float64 scores, first maximum by butterfly
    const double sq = caro_sqrt((double)nsum);  // m.sqrt(sum(counts)), mcts.py:79
    const double c64 = (double)v.c_puct;
    const float keepf = (float)(1.0 - v.explore);
    double best = -__builtin_huge_val();
    besta = 0x7fffffff;
#pragma unroll
    for (int j = 0; j < APL; ++j) {
      const int a = l * APL + j;
      const int n = (int)(nraw[j] & NMASK);
      const float keep = keepf * p[j];                            // py float * float32 -> float32
      const double prob = (double)keep + v.explore * nz[j];       // float32 + float64 -> float64
      const double u = ((c64 * prob) * sq) / (double)(1 + n);
      double qd;
      if (nraw[j] & NSTRONG) qd = (double)q[j];                   // np.float32 Q
      else if (n > 0) qd = (double)__uint_as_float(wraw[j]) / (double)n;  // python-float W / int
      else qd = 0.0;
      double sc = qd + u;
      if (!R::legal(v.gp, d.cur, a)) sc = -__builtin_huge_val();
      if (sc > best || (sc == best && a < besta)) {
        best = sc;
        besta = a;
      }
    }
    // np.argmax: first maximum (mcts.py:136) -- an all-reduce of (score, action) under "greater score, then lower action",
    // which is commutative and associative: the quad / mirror steps of group_allreduce_i32 serve (DPP moves of the three
    // words instead of nine trips through the LDS crossbar), the wider steps stay shuffles
    {
      auto step = [&](double ob, int oa) {
        if (ob > best || (ob == best && oa < besta)) {
          best = ob;
          besta = oa;
        }
      };
      auto dpp64 = [&](auto ctrl_tag) {
        constexpr int CTRL = decltype(ctrl_tag)::value;
        const long long bits = __double_as_longlong(best);
        const int lo = dpp_i32<CTRL>((int)(uint32_t)bits), hi = dpp_i32<CTRL>((int)(uint32_t)((unsigned long long)bits >> 32));
        const int oa = dpp_i32<CTRL>(besta);
        step(__longlong_as_double((long long)(((unsigned long long)(uint32_t)hi << 32) | (uint32_t)lo)), oa);
      };
      if constexpr (LPD >= 2) dpp64(std::integral_constant<int, 0xB1>{});    // quad_perm [1,0,3,2]
      if constexpr (LPD >= 4) dpp64(std::integral_constant<int, 0x4E>{});    // quad_perm [2,3,0,1]
      if constexpr (LPD >= 8) dpp64(std::integral_constant<int, 0x141>{});   // row_half_mirror
      if constexpr (LPD >= 16) dpp64(std::integral_constant<int, 0x140>{});  // row_mirror
#pragma unroll
      for (int m = 16; m < LPD; m <<= 1) step(__shfl_xor(best, m, LPD), __shfl_xor(besta, m, LPD));
    }
  } else {
    // Q + ((c * P) * sqrt(sum N)) / (1 + N) in float32, no contraction (mcts.py:79-84 under numpy >= 2).
    // float32(sqrt(float64(n))) == sqrtf(float32(n)) for n < 2^24: rounding a correctly rounded 53-bit root
    // again to 24 bits is innocuous (53 >= 2 * 24 + 2), so the float64 root is not needed here.
    const float sqf = sqrt_count((float)nsum);  // = sqrtf: IEEE correctly rounded
    const float c32 = v.c_puct;
    float bs = -__builtin_huge_valf();
    int ba = l * APL;
#pragma unroll
    for (int j = 0; j < APL; ++j) {
      const int a = l * APL + j;
      const int n = (int)(nraw[j] & NMASK);
      float tt = c32 * p[j];
      tt = tt * sqf;
      tt = tt / (float)(1 + n);
      float sc = q[j] + tt;
      if (!R::legal(v.gp, d.cur, a)) sc = -__builtin_huge_valf();
      if (sc > bs) {  // strict: the lowest action of the lane keeps a tie
        bs = sc;
        ba = a;
      }
    }
    // first maximum of the group (np.argmax, mcts.py:136): the maximum by DPP, then the lowest lane holding it
    const uint32_t u = orderable(bs + 0.0f);
    const uint32_t um = group_max_u32<LPD>(u);
    const uint64_t holders = group_bits<LPD>(__ballot(u == um), first);
    // (one action per lane: the first holder's lane index IS its action -- no trip through the LDS crossbar)
    if constexpr (APL == 1) besta = __ffsll((unsigned long long)holders) - 1;
    else besta = __shfl(ba, __ffsll((unsigned long long)holders) - 1, LPD);
  }
  {  // the level's record, written by the lane that holds the chosen edge: node, action, and the edge's N and W words
    const int ol = besta / APL, oj = besta - ol * APL;
    if (l == ol) {
      uint32_t en = nraw[0], ew = wraw[0];
#pragma unroll
      for (int j = 1; j < APL; ++j) {
        en = oj == j ? nraw[j] : en;
        ew = oj == j ? wraw[j] : ew;
      }
      // (staged in LDS when the block's records fit: a store to memory here sits in front of the next level's loads in
      // the wave's in-order memory counter, and the level would wait for its acknowledgement)
      const uint4 rec = make_uint4((uint32_t)node | ((uint32_t)besta << 24), en, ew, 0u);
      if (lprec) lprec[d.depth] = rec;
      else prec[d.depth] = rec;
    }
  }
  const bool won = R::template move_group<LPD>(v.gp, d.cur, d.aux, besta, d.player, d.lk, first);  // game.move, mcts.py:138
  d.player ^= 1;
  ++d.depth;
  if (won) {  // mcts.py:140-142
    d.status = ST_TERMINAL;
    d.value = -1.0f;
    return false;
  }
  if (R::full(v.gp, d.cur)) {  // mcts.py:145-146
    d.status = ST_TERMINAL;
    d.value = 0.0f;
    return false;
  }
  return d.depth < v.maxd;
}

// `rows` (fused form, see k_tree): when non-null the block also places its unique leaves itself, in SLOT rows:
// the j-th unique leaf of game g goes to row g * B + j of planes / leaf_keys (and its priors / value come back in
// the same row), so no block needs to know what the others found.  rows[cls] only accumulates the launch's
// total per net (a sum: the order of the adds does not matter); the net kernel maps its dense tiles onto the
// slot rows in game order by itself (caro_net.hip tile_rows), which keeps the whole path free of any dependence
// on block arrival order.  (k_encode produces DENSE rows for the step-wise form instead.)
// `helper_nz` (fused kernels): the block's noise wave writes this minibatch's Dirichlet rows to LDS ([B][AP] doubles
// at helper_nz, then the flag behind them); the tree wave picks them up here instead of generating them.
// dynamic LDS of the kernels that generate Dirichlet rows in line on a geometry with several actions per lane: one
// mailbox of gamma_row_pool per wavefront (descent)
extern __shared__ double caro_dyn_lds[];
template <class GEO>
static inline size_t mail_bytes(int B) {
  return GEO::APL > 1 ? (size_t)((B * GEO::LPD + 63) / 64) * (2 * GEO::AP) * sizeof(double) : 0;
}

template <class GEO>
constexpr int geo_max_batch() { return GEO::LPD >= 64 ? 16 : GEO::LPD >= 32 ? 32 : 64; }

template <class GEO, bool ONE = false>
__device__ __forceinline__ void select_body(const View& v, const GameRegs<GEO>& gr, int B, int mb_index,
                                            const double* __restrict__ noise, int32_t* __restrict__ rows,
                                            float* __restrict__ planes, uint64_t* __restrict__ leaf_keys,
                                            const double* helper_nz = nullptr, volatile int* helper_flag = nullptr) {
  using R = typename GEO::R;
  using Board = typename R::Board;
  constexpr int LPD = GEO::LPD, APL = GEO::APL, AP = GEO::AP, KW = GEO::KW;
  const int g = blockIdx.x;
  const int tid = threadIdx.x;
  const int b = tid / LPD, l = tid % LPD;
  const int first = (tid & 63) - l;  // first lane of this descent's group inside its wave

  // (LDS sized by the geometry: a block has at most 1024 threads, so at most 1024 / LPD descents -- 16 on the boards with
  // one wavefront per descent, whose 64-byte keys would otherwise take 4 KB here and 4 KB in expand_body)
  constexpr int MB = geo_max_batch<GEO>();
  __shared__ uint64_t s_key[MB][KW];
  __shared__ int s_status[MB];
  __shared__ int s_first[MB];
  __shared__ int s_depth[MB];
  __shared__ int s_player[MB];
  __shared__ float s_value[MB];
  __shared__ uint32_t s_dhome[MB];

  if (gr.done) {
    if (tid == 0) {
      v.g_nleaf[g] = 0;
      v.g_class[g] = 0;
      if (rows) v.g_pack[g] = 0;
    }
    return;
  }
  unsigned long long st0 = 0, st_noise = 0, st_root = 0, st_loop = 0;
  if (v.dbg) st0 = __builtin_amdgcn_s_memtime();
  Descent<GEO> d;
  d.cur = gr.root;
  d.aux = R::aux_of(v.gp, d.cur);
  d.lk = R::lane_k(v.gp, l);
  const int player0 = gr.player;
  d.player = player0;
  d.depth = 0;
  d.status = ST_LEAF;
  d.value = 0.0f;
  d.home = 0u;
  const int st_sel = v.n_stores == 2 ? player0 : 0;
  const int t = g * v.n_stores + st_sel;
  const int A = v.A;
  uint4* prec = v.path_rec + ((size_t)g * v.maxB + b) * v.maxd;
  const int tsel = st_sel ? gr.tbl[1] : gr.tbl[0];
  const size_t tb = tbase_sel(v, t, tsel);
  const uint64_t* tkeys = v.node_key + tb * KW;
  const uint32_t* tedges = v.edges + ebase_sel(v, t, tsel) * 4 * AP;

  // the root's row is requested first; the descent's Dirichlet row (only used if the root is in the tree) is
  // generated while it is on its way
  NodeRow<GEO> r;
  load_row<GEO, true>(r, tkeys, tedges, home_slot<R>(v, t, d.cur), l);
  double nz[APL];
  if (noise) {
#pragma unroll
    for (int j = 0; j < APL; ++j) {
      const int a = l * APL + j;
      nz[j] = a < A ? noise[((size_t)g * B + b) * A + a] : 0.0;
    }
  } else if (helper_nz) {
    while (*helper_flag == 0) __builtin_amdgcn_s_sleep(2);  // the noise wave started when this one did: rarely waits
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");  // pairs with the noise wave's release: the rows behind the flag
#pragma unroll
    for (int j = 0; j < APL; ++j) nz[j] = helper_nz[b * AP + l * APL + j];
  } else {
    const uint64_t key = caro_noise_key(v.seed, gr.uid, (uint32_t)gr.ply, (uint32_t)(mb_index * B + b));
    // (several actions per lane: the row's gammas go through an LDS mailbox of the descent's wave, gamma_row_pool --
    // dynamic LDS, mail_bytes<GEO>(B) at the launch)
    noise_group<LPD, APL>(key, l, A, v.alpha, nz, APL > 1 ? caro_dyn_lds + (tid >> 6) * (2 * AP) : nullptr);
  }
  if (v.dbg) st_noise = __builtin_amdgcn_s_memtime();

  // the descents' path records are collected in LDS and written out after the last level
  // (boards of more than 64 cells never fit -- batch x cells <= 512 entries --: no LDS is set aside for them)
  constexpr int PREC_LDS = GEO::APL > 1 ? 1 : 512;
  __shared__ uint4 s_prec[PREC_LDS];
  const bool stage = B * v.maxd <= PREC_LDS;
  uint4* lprec = stage ? s_prec + b * v.maxd : nullptr;
  bool live = descend_level<GEO, true>(v, d, t, tkeys, tedges, prec, lprec, l, first, nz, r);
  if (v.dbg) st_root = __builtin_amdgcn_s_memtime();
  // (a group's lanes leave the loop together: everything a level exchanges stays inside the group)
  while (live) live = descend_level<GEO, false>(v, d, t, tkeys, tedges, prec, lprec, l, first, nullptr, r);
  if (stage) {  // a group's lanes share a wavefront: its LDS writes are in order with these reads
    __builtin_amdgcn_wave_barrier();
    for (int i = l; i < d.depth; i += LPD) prec[i] = s_prec[b * v.maxd + i];
  }
  if (v.dbg) st_loop = __builtin_amdgcn_s_memtime();

  if constexpr (ONE) {
    // One wavefront, descent b in lanes [b * LPD, (b + 1) * LPD), every lane of a group holding the same copy of its
    // descent: the planned-set de-duplication (mcts.py:272-278: the first occurrence of a new leaf is kept), the
    // ranks, the tallies and the records are done from registers -- the leaf of descent o is read from the first lane
    // of its group (a uniform lane index: v_readlane), nothing goes through LDS.
    const bool head = l == 0;  // one speaker per descent
    const int st = d.status;
    bool dup = false;
    for (int o = 0; o < B; ++o) {
      const int src = o * LPD;
      bool eq = o < b && __builtin_amdgcn_readlane(st, src) == ST_LEAF;
#pragma unroll
      for (int w = 0; w < KW; ++w) {
        const uint32_t lo = (uint32_t)__builtin_amdgcn_readlane((int)(uint32_t)d.cur.w[w], src);
        const uint32_t hi = (uint32_t)__builtin_amdgcn_readlane((int)(uint32_t)(d.cur.w[w] >> 32), src);
        eq = eq && ((((uint64_t)hi << 32) | lo) == d.cur.w[w]);
      }
      dup = dup || eq;
    }
    const bool first_seen = st == ST_LEAF && !dup;
    const unsigned long long m_first = __ballot(head && first_seen);
    const unsigned long long m_term = __ballot(head && st == ST_TERMINAL);
    const unsigned long long m_drop = __ballot(head && st == ST_LEAF && dup);
    const int my_rank = __popcll(m_first & ((1ull << tid) - 1ull));  // leaves of earlier descents (heads only carry bits)
    const int levels = group_sum_i32<64>(head ? d.depth : 0);
    int maxdep = 0;
    if (v.dbg) maxdep = group_allreduce_i32<64>(d.depth, [](int x, int y) { return x > y ? x : y; });
    if (head) {
      const size_t di = (size_t)g * v.maxB + b;
      const uint32_t stw = (uint32_t)((st == ST_LEAF && dup) ? ST_DROPPED : st);
      v.d_rec[di] = make_uint4(stw | ((uint32_t)d.depth << 8) | ((uint32_t)my_rank << 16) | ((uint32_t)d.player << 24),
                               __float_as_uint(d.value), d.home, 0u);
      store_board<R>(v.d_key + di * KW, d.cur);
    }
    const int nleaf = __popcll(m_first);
    if (tid == 0) {
      v.g_nleaf[g] = nleaf;
      v.g_tree[g] = t;
      v.g_class[g] = v.n_nets == 2 ? player0 : 0;
      if (v.dbg) {  // cycles since kernel start: noise generated | root level done | descents done | end; max depth
        unsigned long long* dd = v.dbg + (size_t)g * 8;
        dd[0] = st_noise - st0; dd[1] = st_root - st0; dd[2] = st_loop - st0;
        dd[3] = __builtin_amdgcn_s_memtime() - st0; dd[4] = (unsigned long long)maxdep;
      }
      // this block is the only writer of game g's tallies; adds without a return value do not wait for memory
      unsigned long long* ctr = v.counters + (size_t)g * C_N;
      atomicAdd(ctr + C_SIMS, (unsigned long long)B);
      atomicAdd(ctr + C_LEVELS, (unsigned long long)levels);
      atomicAdd(ctr + C_TERMINALS, (unsigned long long)__popcll(m_term));
      atomicAdd(ctr + C_DROPPED, (unsigned long long)__popcll(m_drop));
      if (rows) {
        const int cls = v.n_nets == 2 ? player0 : 0;
        if (nleaf) atomicAdd(rows + cls, nleaf);
        v.g_off[g] = g * B;
        v.g_pack[g] = nleaf | (cls << 8);
      }
    }
    if (rows) {
      // NN planes of the unique leaves into the game's slot rows, leaf by leaf in first-seen order: the board and the
      // player to move come from the first lane of the leaf's group
      const int HW = v.HW;
      int local = 0;
      // what a lane's elements of a plane row are (plane, cell) does not depend on the leaf: formed once.  (Per leaf
      // and element the index arithmetic -- two divisions by a run-time HW -- cost a block with 5..8 leaves 5-10 k cycles.)
      typename R::PlaneAt pat[R::PLANE_ITEMS];
#pragma unroll
      for (int k = 0; k < R::PLANE_ITEMS; ++k) pat[k] = R::plane_at(v.gp, tid + 64 * k < 2 * HW ? tid + 64 * k : 0);
      for (unsigned long long m = m_first; m; m &= m - 1ull) {
        const int src = __ffsll(m) - 1;
        Board brd;
#pragma unroll
        for (int w = 0; w < KW; ++w) {
          const uint32_t lo = (uint32_t)__builtin_amdgcn_readlane((int)(uint32_t)d.cur.w[w], src);
          const uint32_t hi = (uint32_t)__builtin_amdgcn_readlane((int)(uint32_t)(d.cur.w[w] >> 32), src);
          brd.w[w] = ((uint64_t)hi << 32) | lo;
        }
        const int who = __builtin_amdgcn_readlane(d.player, src);
        const int rowi = g * B + local++;
        float* dst = planes + (size_t)rowi * 2 * HW;
#pragma unroll
        for (int k = 0; k < R::PLANE_ITEMS; ++k)
          if (tid + 64 * k < 2 * HW) dst[tid + 64 * k] = R::plane_val(brd, who, pat[k]);
        if (leaf_keys && tid < KW) {
          uint64_t kw = brd.w[0];
#pragma unroll
          for (int w = 1; w < KW; ++w) kw = tid == w ? brd.w[w] : kw;  // no run-time index into the registers
          leaf_keys[(size_t)rowi * KW + tid] = kw;
        }
      }
    }
    return;
  }
  if (l == 0) {
#pragma unroll
    for (int w = 0; w < KW; ++w) s_key[b][w] = d.cur.w[w];
    s_status[b] = d.status;
    s_depth[b] = d.depth;
    s_player[b] = d.player;
    s_value[b] = d.value;
    s_dhome[b] = d.home;
  }
  block_sync<ONE>();
  // From here on thread bb < B (all in the first wavefront) speaks for descent bb: planned-set de-duplication
  // (mcts.py:272-278: the first occurrence of a new leaf is kept), ranks and tallies by ballot.
  if (tid < 64) {
    const int bb = tid;
    const bool mine = bb < B;
    const int st = mine ? s_status[bb] : ST_DROPPED;
    const int dep = mine ? s_depth[bb] : 0;
    Board key;
#pragma unroll
    for (int w = 0; w < KW; ++w) key.w[w] = mine ? s_key[bb][w] : 0ull;
    bool dup = false;
    for (int o = 0; o < B; ++o) {  // the same LDS words for every lane: broadcast reads
      bool eq = o < bb && s_status[o] == ST_LEAF;
#pragma unroll
      for (int w = 0; w < KW; ++w) eq = eq && (s_key[o][w] == key.w[w]);
      dup = dup || eq;
    }
    const bool first_seen = st == ST_LEAF && !dup;
    const unsigned long long m_first = __ballot(first_seen);
    const unsigned long long m_term = __ballot(st == ST_TERMINAL);
    const unsigned long long m_drop = __ballot(st == ST_LEAF && dup);
    int levels, maxdep = 0;
    if (ONE || blockDim.x >= 64) {  // a whole wavefront: cross-lane reduction
      levels = group_sum_i32<64>(dep);
      if (v.dbg) maxdep = group_allreduce_i32<64>(dep, [](int x, int y) { return x > y ? x : y; });
    } else {                 // a partial wavefront (batch x lanes < 64): lanes that do not exist cannot be read
      levels = 0;
      for (int o = 0; o < B; ++o) {
        levels += s_depth[o];
        maxdep = s_depth[o] > maxdep ? s_depth[o] : maxdep;
      }
    }
    if (mine) {
      const size_t di = (size_t)g * v.maxB + bb;
      const uint32_t stw = (uint32_t)((st == ST_LEAF && dup) ? ST_DROPPED : st);
      const uint32_t rank = (uint32_t)__popcll(m_first & ((1ull << bb) - 1ull));
      v.d_rec[di] = make_uint4(stw | ((uint32_t)dep << 8) | (rank << 16) | ((uint32_t)s_player[bb] << 24),
                               __float_as_uint(s_value[bb]), s_dhome[bb], 0u);
      store_board<R>(v.d_key + di * KW, key);
      s_first[bb] = first_seen;
    }
    if (tid == 0) {
      const int nleaf = __popcll(m_first);
      v.g_nleaf[g] = nleaf;
      v.g_tree[g] = t;
      v.g_class[g] = v.n_nets == 2 ? player0 : 0;
      if (v.dbg) {  // cycles since kernel start: noise generated | root level done | descents done | end; max depth
        unsigned long long* dd = v.dbg + (size_t)g * 8;
        dd[0] = st_noise - st0; dd[1] = st_root - st0; dd[2] = st_loop - st0;
        dd[3] = __builtin_amdgcn_s_memtime() - st0; dd[4] = (unsigned long long)maxdep;
      }
      // this block is the only writer of game g's tallies; adds without a return value do not wait for memory
      unsigned long long* ctr = v.counters + (size_t)g * C_N;
      atomicAdd(ctr + C_SIMS, (unsigned long long)B);
      atomicAdd(ctr + C_LEVELS, (unsigned long long)levels);
      atomicAdd(ctr + C_TERMINALS, (unsigned long long)__popcll(m_term));
      atomicAdd(ctr + C_DROPPED, (unsigned long long)__popcll(m_drop));
      if (rows) {
        const int cls = v.n_nets == 2 ? player0 : 0;
        v.g_off[g] = g * B;
        v.g_pack[g] = nleaf | (cls << 8);
        if (nleaf) {
          // (this form keeps the add's return value: where the game's rows stand in the launch's list of leaves -- any
          // order serves the one-board-per-workgroup net kernel, caro_net_forward_slot_list)
          const int at = atomicAdd(rows + cls, nleaf);
          int32_t* sl = v.slot_list + (size_t)cls * v.G * B;
          for (int j = 0; j < nleaf; ++j) sl[at + j] = g * B + j;
        }
      }
    }
  }
  if (rows) block_sync<ONE>();  // s_first for the plane writers
  if (rows) {
    const int HW = v.HW;
    int local = 0;
    for (int bb = 0; bb < B; ++bb) {
      if (!s_first[bb]) continue;
      const int rowi = g * B + local++;
      Board brd;
#pragma unroll
      for (int w = 0; w < KW; ++w) brd.w[w] = s_key[bb][w];
      const int who = s_player[bb];
      float* dst = planes + (size_t)rowi * 2 * HW;
      for (int i = tid; i < 2 * HW; i += block_threads<ONE>()) dst[i] = R::plane(v.gp, brd, who, i / HW, i % HW);
      if (leaf_keys && tid < KW) leaf_keys[(size_t)rowi * KW + tid] = s_key[bb][tid];
    }
  }
}

template <class GEO>
__global__ void k_select(View v, int B, int mb_index, const double* __restrict__ noise) {
  const GameRegs<GEO> gr = load_game<GEO>(v, blockIdx.x);
  select_body<GEO>(v, gr, B, mb_index, noise, nullptr, nullptr, nullptr);
}

// NN planes of the unique leaves, written as dense rows (rows of net 0 first, then net 1).  Every block
// derives its game's row offset itself by summing the unique-leaf counts of the games before it (a few KB of
// L2-resident ints), which replaces a separate scan launch; the last block also publishes the totals.
template <class GEO>
__global__ void k_encode(View v, int B, float* __restrict__ planes, uint64_t* __restrict__ leaf_keys) {
  using R = typename GEO::R;
  constexpr int KW = GEO::KW;
  __shared__ int s_sum[4][128];
  const int g = blockIdx.x;
  const int tid = threadIdx.x;
  // c0b / c1b: leaves of class 0 / 1 in games before g; c0a: class 0 in all games
  int c0b = 0, c1b = 0, c0a = 0, c1a = 0;
  for (int k = tid; k < v.G; k += blockDim.x) {
    const int n = v.g_nleaf[k];
    const int cls = v.g_class[k];
    if (cls) { c1a += n; if (k < g) c1b += n; } else { c0a += n; if (k < g) c0b += n; }
  }
  s_sum[0][tid] = c0b; s_sum[1][tid] = c1b; s_sum[2][tid] = c0a; s_sum[3][tid] = c1a;
  __syncthreads();
  for (int d = 64; d > 0; d >>= 1) {
    if (tid < d)
      for (int c = 0; c < 4; ++c) s_sum[c][tid] += s_sum[c][tid + d];
    __syncthreads();
  }
  const int tot0 = s_sum[2][0], tot1 = s_sum[3][0];
  const int off = v.g_class[g] ? tot0 + s_sum[1][0] : s_sum[0][0];
  if (tid == 0) {
    v.g_off[g] = off;
    if (g == v.G - 1) {
      v.leaf_count[0] = tot0;
      v.leaf_count[1] = tot1;
      v.leaf_count[2] = B;
    }
  }
  if (v.g_nleaf[g] == 0) return;
  const int HW = v.HW;
  for (int b = 0; b < B; ++b) {
    const size_t di = (size_t)g * v.maxB + b;
    const uint32_t rx = v.d_rec[di].x;
    if ((int)(rx & 0xFFu) != ST_LEAF) continue;
    const int rowi = off + (int)((rx >> 16) & 0xFFu);
    const typename R::Board brd = load_board<R>(v.d_key + di * KW);
    const int who = (int)(rx >> 24);
    float* dst = planes + (size_t)rowi * 2 * HW;
    for (int i = tid; i < 2 * HW; i += blockDim.x) dst[i] = R::plane(v.gp, brd, who, i / HW, i % HW);
    if (leaf_keys && tid < KW) leaf_keys[(size_t)rowi * KW + tid] = v.d_key[di * KW + tid];
  }
}

// _backup of one path by one lane (mcts.py:225-246), float32 running sums in queue order
template <int AP>
__device__ __forceinline__ void backup_path(const View& v, int t, float value, bool strong_val, const int32_t* pn,
                                            const int32_t* pa, int len) {
  float cur = -value;
  for (int i = len - 1; i >= 0; --i) {
    uint32_t* row = v.edges + (ebase(v, t) + pn[i]) * 4 * AP;
    const int a = pa[i];
    const uint32_t nraw = row[4 * a];
    const int n = (int)(nraw & NMASK) + 1;
    const uint32_t strong = (nraw & NSTRONG) | (strong_val ? NSTRONG : 0u);
    const float w = __uint_as_float(row[4 * a + 1]) + cur;
    const float q = w / (float)n;
    row[4 * a] = (uint32_t)n | strong;
    row[4 * a + 1] = __float_as_uint(w);
    row[4 * a + 2] = __float_as_uint(q);
    cur = -cur;
  }
}
// the same from the path records of a descent (reads the edges afresh: the sequential fallback of expand_body)
template <int AP>
__device__ __forceinline__ void backup_path_rec(const View& v, int t, float value, bool strong_val, const uint4* rec,
                                                int len) {
  float cur = -value;
  for (int i = len - 1; i >= 0; --i) {
    const uint32_t na = rec[i].x;
    uint32_t* row = v.edges + (ebase(v, t) + (na & 0xFFFFFFu)) * 4 * AP;
    const int a = (int)(na >> 24);
    const uint32_t nraw = row[4 * a];
    const int n = (int)(nraw & NMASK) + 1;
    const uint32_t strong = (nraw & NSTRONG) | (strong_val ? NSTRONG : 0u);
    const float w = __uint_as_float(row[4 * a + 1]) + cur;
    row[4 * a] = (uint32_t)n | strong;
    row[4 * a + 1] = __float_as_uint(w);
    row[4 * a + 2] = __float_as_uint(w / (float)n);
    cur = -cur;
  }
}

// Everything expand_body needs from memory, requested in ONE round of independent loads -- they depend on the game
// index and the lane only, so the fused kernels issue them at the very top, beside the game's scalars (load_game),
// and the whole expand + backup of a block costs that one memory round: the descent records (with the home slot of
// each leaf and whether it was free, as the descent saw it), the leaf boards, the net's value and prior rows of the
// game's slot rows, and the first 16 levels of every path WITH the N and W words of their edges as select saw them
// (nothing touches the tree in between), so the backup only WRITES edges.
template <class GEO>
struct ExpandPre {
  static constexpr int NPR = (MAXB * 0 + 64 * GEO::APL + 63) / 64;  // prior values per lane (B x AP <= 64 x APL in the one-wave forms)
  uint4 rec;                       // lane < B: the descent's record
  typename GEO::R::Board brd;      // lane < B: its leaf board
  float row_val;                   // lane < B: values[off + lane]
  uint4 p0, p1;                    // lane = descent * 8 + level: path records of levels (lane & 7) and 8 + (lane & 7)
  float pr[NPR];                   // priors of the slot rows: index lane + 64 k over (leaf rank, action)
  int nleaf;
};
template <class GEO, bool ONE>
__device__ __forceinline__ ExpandPre<GEO> expand_preload(const View& v, int g, int B, int off, const float* __restrict__ probs,
                                                         const float* __restrict__ values) {
  using R = typename GEO::R;
  constexpr int KW = GEO::KW, AP = GEO::AP;
  ExpandPre<GEO> e;
  const int lane = threadIdx.x;
  e.nleaf = v.g_nleaf[g];
  e.rec = make_uint4(ST_DROPPED, 0u, 0u, 0u);
  e.row_val = 0.f;
#pragma unroll
  for (int w = 0; w < KW; ++w) e.brd.w[w] = 0;
  if (lane < B) {
    const size_t di = (size_t)g * v.maxB + lane;
    e.rec = v.d_rec[di];
    e.brd = load_board<R>(v.d_key + di * KW);
    if ((long long)off + lane < (long long)v.G * v.maxB) e.row_val = values[off + lane];  // leaf rows off .. off + nleaf - 1
  }
  e.p0 = e.p1 = make_uint4(0u, 0u, 0u, 0u);
  if (B <= 8 && lane < 64 && (lane >> 3) < B) {
    const uint4* pr = v.path_rec + ((size_t)g * v.maxB + (lane >> 3)) * v.maxd;
    if ((lane & 7) < v.maxd) e.p0 = pr[lane & 7];
    if (8 + (lane & 7) < v.maxd) e.p1 = pr[8 + (lane & 7)];
  }
#pragma unroll
  for (int k = 0; k < ExpandPre<GEO>::NPR; ++k) {
    e.pr[k] = 0.f;
    if (ONE) {  // slot rows: leaf rank b of the game sits at row off + b whatever the other games found
      const int idx = lane + 64 * k, b = idx / AP, a = idx - b * AP;
      if (idx < B * AP && a < v.A && lane < 64) e.pr[k] = probs[(size_t)(off + b) * v.A + a];
    }
  }
  return e;
}

// _create_node (mcts.py:178-190) for every unique leaf, then the queued _backup calls (mcts.py:225-246,
// 286-287).  The reference applies the backups one after another; contributions to DIFFERENT edges commute,
// contributions to the SAME edge must keep queue order because W is a float32 running sum.  So the queue is
// flattened into (edge, +-value) entries in reference order (terminals by sim index, then new leaves first
// seen; inside a path from the leaf upwards), the first entry of every distinct edge becomes its owner and
// applies all entries of that edge in order, and the owners' writes proceed in parallel.
// Memory: one round of loads (ExpandPre, issued by the caller -- in the fused kernels at the top of the kernel), then
// only stores: keys and rows of the new nodes, the owners' edges (their old N / W words travel in the path records).
// Rare cases take extra rounds: a leaf whose home slot is taken (probe sequence), paths deeper than 16 levels.
template <class GEO, bool ONE = false>
__device__ __forceinline__ void expand_body(const View& v, GameRegs<GEO>& gr, int B, const ExpandPre<GEO>& pre, int off,
                                            const float* __restrict__ probs) {
  using R = typename GEO::R;
  using Board = typename R::Board;
  constexpr int AP = GEO::AP, KW = GEO::KW;
  constexpr int MAXE = GEO::APL > 1 ? 256 : 512;  // entries held in LDS; longer queues fall back to the sequential form
  constexpr int MAXB = geo_max_batch<GEO>();      // (shadows the global bound: LDS sized by the geometry, see select_body)
  __shared__ int e_node[MAXE];
  __shared__ short e_act[MAXE];
  __shared__ float e_val[MAXE];
  __shared__ unsigned char e_strong[MAXE];
  __shared__ uint32_t e_n[MAXE], e_w[MAXE];  // the edge's N and W words as select saw them
  __shared__ int s_total;
  __shared__ uint64_t s_brd[MAXB][KW];
  __shared__ int s_node[MAXB], s_row[MAXB];
  __shared__ uint32_t s_home[MAXB];
  __shared__ unsigned char s_free[MAXB];
  __shared__ int q_len[MAXB], q_off[MAXB], q_b[MAXB];   // queue position -> length, entry offset, descent
  __shared__ float q_val[MAXB];
  __shared__ unsigned char q_strong[MAXB];
  __shared__ int s_nq;
  __shared__ uint4 s_p0[64], s_p1[64];   // the preloaded path records: [descent * 8 + level], levels 0..7 and 8..15
  __shared__ float s_rowval[MAXB];
  const int g = blockIdx.x;
  if (gr.done) return;
  const int lane = threadIdx.x;
  // diagnostic stamps (tools/probe_stag.py --expand): cycles at the phase boundaries of this function, game g
  unsigned long long* xs = v.dbg ? v.dbg + ((size_t)v.G + g) * 8 : nullptr;
#define CARO_XS(n) if (xs && lane == 0) xs[n] = __builtin_amdgcn_s_memtime();
  CARO_XS(0)
  // the tree the pending minibatch was selected on: the mover's (the ply comes after the backup)
  const int st_sel = v.n_stores == 2 ? gr.player : 0;
  const int t = g * v.n_stores + st_sel;
  const int nleaf = pre.nleaf;
  const int base = st_sel ? gr.nn[1] : gr.nn[0];
  const bool overflow = base + nleaf > v.cap;
  const int A = v.A;
  const int tsel = st_sel ? gr.tbl[1] : gr.tbl[0];
  const size_t tb = tbase_sel(v, t, tsel);
  const size_t eb = ebase_sel(v, t, tsel);
  const int my_st = (int)(pre.rec.x & 0xFFu), my_len = (int)((pre.rec.x >> 8) & 0xFFu);
  const int my_local = (int)((pre.rec.x >> 16) & 0xFFu);
  const uint32_t my_home = pre.rec.z;
  float my_val = __uint_as_float(pre.rec.y);  // meaningful for terminals
  const Board& brd = pre.brd;
  const bool pre_paths = B <= 8 && block_threads<ONE>() >= 64;
  if (pre_paths && lane < 64) {
    s_p0[lane] = pre.p0;
    s_p1[lane] = pre.p1;
  }
  const bool is_leaf = lane < B && my_st == ST_LEAF && !overflow;
  // the value of this descent's leaf: row off + my_local, held by lane my_local
  if (lane < B) s_rowval[lane] = pre.row_val;
  block_sync<ONE>();
  if (lane < B) {
    s_node[lane] = -2;  // not a leaf
    if (is_leaf) {
      my_val = s_rowval[my_local];
      s_home[lane] = my_home & 0x7fffffffu;
      s_free[lane] = (my_home >> 31) != 0u;
      s_row[lane] = off + my_local;
      s_node[lane] = -1;
#pragma unroll
      for (int w = 0; w < KW; ++w) s_brd[lane][w] = brd.w[w];
    }
  }
  block_sync<ONE>();
  CARO_XS(1)  // the preloaded round has arrived, leaves published
  // the block's first wavefront as a whole (the sections below that talk across lanes by readlane / ballot); the other
  // wavefronts of a multi-wave block skip those sections (no barrier sits inside them)
  const bool whole_wave = ONE || blockDim.x >= 64;
  const bool wave0 = whole_wave && lane < 64;
  if (!overflow) {
    // _create_node.  The reference inserts the leaves one after another (first-seen order): a leaf whose home slot is
    // free and not taken by an earlier leaf of this minibatch lands in its home slot, anything else walks the probe
    // sequence.  Which slot a node gets carries no meaning (node id = slot; every lookup walks the probe sequence from
    // the home slot to the first empty one), so: the leaves that meet a free home slot of their own -- nearly all --
    // are placed by their own lanes at once, and lane 0 then inserts the few others one by one behind them (a
    // wavefront's accesses to one address stay in order: its probes see the keys just written).
    if (wave0) {
      bool clash = false;
      for (int bb = 0; bb < B; ++bb) {  // bb is uniform: readlane, not a trip through the LDS crossbar
        const int ob = __builtin_amdgcn_readlane((int)is_leaf, bb);
        const uint32_t oh = (uint32_t)__builtin_amdgcn_readlane((int)(my_home & 0x7fffffffu), bb);
        clash = clash || (bb < lane && ob && oh == (my_home & 0x7fffffffu));
      }
      const bool fast = is_leaf && (my_home >> 31) != 0u && !clash;
      if (fast) {
        uint64_t* k = v.node_key + (tb + (my_home & 0x7fffffffu)) * KW;
#pragma unroll
        for (int w = KW - 1; w >= 0; --w) k[w] = brd.w[w];
        s_node[lane] = (int)(my_home & 0x7fffffffu);
      }
      const unsigned long long slow = __ballot(is_leaf && !fast);
      if (slow) {
        asm volatile("" ::: "memory");
        __builtin_amdgcn_wave_barrier();
        if (lane == 0)
          for (unsigned long long m = slow; m; m &= m - 1ull) {
            const int b = __ffsll(m) - 1;
            Board lb;
#pragma unroll
            for (int w = 0; w < KW; ++w) lb.w[w] = s_brd[b][w];
            s_node[b] = insert_key<R>(v, t, lb);
          }
      }
    } else if (!whole_wave && lane == 0) {  // a block of less than one wavefront (tiny boards, small batches)
      for (int b = 0; b < B; ++b) {
        if (s_node[b] == -2) continue;
        Board lb;
#pragma unroll
        for (int w = 0; w < KW; ++w) lb.w[w] = s_brd[b][w];
        s_node[b] = insert_key<R>(v, t, lb);
      }
    }
    if (lane == 0) {
      unsigned long long* ctr = v.counters + (size_t)g * C_N;
      v.n_nodes[t] = base + nleaf;
      atomicAdd(v.n_created + t, nleaf);  // no return value: nothing waits for it
      atomicAdd(ctr + C_EXPANSIONS, (unsigned long long)nleaf);
    }
    if (st_sel) gr.nn[1] = base + nleaf;
    else gr.nn[0] = base + nleaf;
  } else if (lane == 0) {
    atomicAdd(v.counters + (size_t)g * C_N + C_OVERFLOW, 1ull);
  }
  CARO_XS(2)  // leaves inserted
  // Flatten the backup queue, lane-parallel.  Queue order (reference): terminals by sim index, then new
  // leaves by first-seen index; inside one backup from the leaf upwards.
  const unsigned long long m_term = __ballot(lane < B && my_st == ST_TERMINAL);
  const unsigned long long m_leaf = __ballot(is_leaf);
  const int n_term = __popcll(m_term);
  const unsigned long long below = (1ull << lane) - 1ull;
  int qpos = -1;
  if (lane < B && my_st == ST_TERMINAL) qpos = __popcll(m_term & below);
  else if (is_leaf) qpos = n_term + __popcll(m_leaf & below);
  if (qpos >= 0) {
    q_len[qpos] = my_len;
    q_b[qpos] = lane;
    q_val[qpos] = my_val;
    q_strong[qpos] = my_st == ST_LEAF;
  }
  if (lane == 0) s_nq = n_term + __popcll(m_leaf);
  block_sync<ONE>();  // also publishes s_node of the new leaves
  const int nq = s_nq;
  if (lane == 0) {  // exclusive scan over at most B queue items
    int acc = 0;
    for (int k = 0; k < nq; ++k) {
      q_off[k] = acc;
      acc += q_len[k];
    }
    s_total = acc;
  }
  block_sync<ONE>();
  CARO_XS(3)  // queue flattened
  const int total = s_total;
  // the path entries (levels beyond the preloaded 16 are fetched here, by the few entries that need them)
  if (total <= MAXE)
    for (int j = lane; j < total; j += block_threads<ONE>()) {
      int k = 0;
      while (k + 1 < nq && q_off[k + 1] <= j) ++k;  // queue item of entry j
      const int r = j - q_off[k];                    // r-th entry of that backup, counted from the leaf
      const int i = q_len[k] - 1 - r;
      uint4 rec;
      if (pre_paths && i < 8) rec = s_p0[q_b[k] * 8 + i];
      else if (pre_paths && i < 16) rec = s_p1[q_b[k] * 8 + i - 8];
      else rec = v.path_rec[((size_t)g * v.maxB + q_b[k]) * v.maxd + i];
      e_node[j] = (int)(rec.x & 0xFFFFFFu);
      e_act[j] = (short)(rec.x >> 24);
      e_n[j] = rec.y;
      e_w[j] = rec.z;
      e_val[j] = (r & 1) ? q_val[k] : -q_val[k];     // cur = -value at the leaf's parent, sign flips each ply (mcts.py:238,246)
      e_strong[j] = q_strong[k];
    }
  CARO_XS(4)  // path entries listed
  if (!overflow)
    for (int idx = lane; idx < B * AP; idx += block_threads<ONE>()) {  // lanes over (leaf, action)
      const int b = idx / AP, a = idx - b * AP;
      const int node = s_node[b];
      // the leaf's prior row.  Slot rows (ONE) were preloaded by (leaf rank, action): the value sits in lane src & 63,
      // slot src >> 6 -- fetched by EVERY lane (B * AP is a multiple of 64 there: the loop's trip count is uniform and a
      // shuffle needs its source lane active), used by the lanes of leaves
      float pv = 0.f;
      if (ONE) {
        const int src = node >= 0 ? (s_row[b] - off) * AP + a : lane;
#pragma unroll
        for (int k = 0; k < ExpandPre<GEO>::NPR; ++k) {
          const float x = __shfl(pre.pr[k], src & 63, 64);
          pv = (src >> 6) == k ? x : pv;
        }
      } else if (node >= 0 && a < A) {
        pv = probs[(size_t)s_row[b] * A + a];
      }
      if (node < 0) continue;  // not a leaf (a failed insert cannot happen while n_nodes <= cap < hcap)
      uint32_t* row = v.edges + (eb + node) * 4 * AP;
      if (a >= A) pv = 0.f;
      reinterpret_cast<uint4*>(row)[a] = make_uint4(0u, 0u, 0u, __float_as_uint(pv));  // N = W = Q = 0, P
    }
  if (total > MAXE) {  // queue does not fit the LDS list: apply sequentially, in order (never at B*depth <= 512)
    if (lane == 0)
      for (int k = 0; k < nq; ++k) {
        const size_t di = (size_t)g * v.maxB + q_b[k];
        backup_path_rec<AP>(v, t, q_val[k], q_strong[k] != 0, v.path_rec + di * v.maxd, q_len[k]);
      }
    return;
  }
  block_sync<ONE>();
  CARO_XS(5)  // rows of the new nodes written
  // ---- the owners' writes
  const int n = total;
  if (whole_wave && n <= 128) {
    if (wave0) {
    // One wavefront, at most two entries per lane (j0 = lane, j1 = lane + 64).  The entries of one edge are found by
    // matching: the first entry not grouped yet is broadcast, a ballot marks its equals (bit = entry index), the lane
    // holding that first entry becomes the owner and keeps the two masks.  One round per DISTINCT edge, a few scalar
    // instructions each -- the quadratic search of the general form below cost the slowest blocks 20 k cycles.
    const int j0 = lane, j1 = lane + 64;
    const bool v0 = j0 < n, v1 = j1 < n;
    const int key0 = v0 ? (e_node[j0] << 8) | (int)(unsigned short)e_act[j0] : -1;
    const int key1 = v1 ? (e_node[j1] << 8) | (int)(unsigned short)e_act[j1] : -1;
    unsigned long long todo0 = __ballot(v0), todo1 = __ballot(v1);
    bool own0 = false, own1 = false;
    unsigned long long g0m0 = 0ull, g0m1 = 0ull, g1m1 = 0ull;
    while (todo0 | todo1) {
      const bool lo = todo0 != 0ull;
      const int lead = lo ? __ffsll((unsigned long long)todo0) - 1 : __ffsll((unsigned long long)todo1) - 1;
      const int k = lo ? __builtin_amdgcn_readlane(key0, lead) : __builtin_amdgcn_readlane(key1, lead);
      const unsigned long long m0 = __ballot(v0 && key0 == k);
      const unsigned long long m1 = __ballot(v1 && key1 == k);
      if (lane == lead) {
        if (lo) { own0 = true; g0m0 = m0; g0m1 = m1; }
        else { own1 = true; g1m1 = m1; }
      }
      todo0 &= ~m0;
      todo1 &= ~m1;
    }
    // every owner starts from the edge's words in its own entry (all entries of an edge carry the same ones) and adds
    // its entries in queue order (ascending index)
    uint32_t* row0 = own0 ? v.edges + (eb + (key0 >> 8)) * 4 * AP : nullptr;
    uint32_t* row1 = own1 ? v.edges + (eb + (key1 >> 8)) * 4 * AP : nullptr;
    const int a0 = key0 & 0xff, a1 = key1 & 0xff;
    if (own0) {
      const uint32_t n0 = e_n[j0];
      int cnt = (int)(n0 & NMASK);
      uint32_t strong = n0 & NSTRONG;
      float w = __uint_as_float(e_w[j0]);
      for (unsigned long long m = g0m0; m; m &= m - 1ull) {
        const int e = __ffsll(m) - 1;
        cnt += 1;
        w = w + e_val[e];
        if (e_strong[e]) strong = NSTRONG;
      }
      for (unsigned long long m = g0m1; m; m &= m - 1ull) {
        const int e = 64 + __ffsll(m) - 1;
        cnt += 1;
        w = w + e_val[e];
        if (e_strong[e]) strong = NSTRONG;
      }
      row0[4 * a0] = (uint32_t)cnt | strong;  // N, W, Q: twelve adjacent bytes (P stays)
      row0[4 * a0 + 1] = __float_as_uint(w);
      row0[4 * a0 + 2] = __float_as_uint(w / (float)cnt);
    }
    if (own1) {
      const uint32_t n1 = e_n[j1];
      int cnt = (int)(n1 & NMASK);
      uint32_t strong = n1 & NSTRONG;
      float w = __uint_as_float(e_w[j1]);
      for (unsigned long long m = g1m1; m; m &= m - 1ull) {
        const int e = 64 + __ffsll(m) - 1;
        cnt += 1;
        w = w + e_val[e];
        if (e_strong[e]) strong = NSTRONG;
      }
      row1[4 * a1] = (uint32_t)cnt | strong;
      row1[4 * a1 + 1] = __float_as_uint(w);
      row1[4 * a1 + 2] = __float_as_uint(w / (float)cnt);
    }
    }
  } else
  for (int j = lane; j < n; j += block_threads<ONE>()) {
    const int node = e_node[j], a = e_act[j];
    bool owner = true;
    for (int k = 0; k < j; ++k) owner = owner && !(e_node[k] == node && e_act[k] == a);
    if (!owner) continue;
    uint32_t* row = v.edges + (eb + node) * 4 * AP;
    const uint32_t nraw = e_n[j];
    int cnt = (int)(nraw & NMASK);
    uint32_t strong = nraw & NSTRONG;
    float w = __uint_as_float(e_w[j]);
    for (int k = j; k < n; ++k) {
      if (e_node[k] == node && e_act[k] == a) {
        cnt += 1;                 // visit_count += 1
        w = w + e_val[k];         // value += cur_value (float32, queue order)
        if (e_strong[k]) strong = NSTRONG;
      }
    }
    row[4 * a] = (uint32_t)cnt | strong;
    row[4 * a + 1] = __float_as_uint(w);
    row[4 * a + 2] = __float_as_uint(w / (float)cnt);  // value_avg = value / visit_count
  }
  CARO_XS(6)  // backups applied
  if (xs && lane == 0) xs[7] = (unsigned long long)total;
#undef CARO_XS
}

template <class GEO>
__global__ void k_expand_backup(View v, const float* __restrict__ probs, const float* __restrict__ values) {
  GameRegs<GEO> gr = load_game<GEO>(v, blockIdx.x);
  const int B = v.leaf_count[2], off = v.g_off[blockIdx.x];  // dense rows: the game's first row comes from k_encode
  const ExpandPre<GEO> pre = expand_preload<GEO, false>(v, blockIdx.x, B, off, probs, values);
  expand_body<GEO>(v, gr, B, pre, off, probs);
}

// The block's NOISE WAVE (fused kernels: 128 threads, wave 1): the Dirichlet rows of this minibatch's descents --
// pure float64 arithmetic, ~9 k cycles, a quarter of a median block -- are generated here while the tree wave
// (wave 0) sits in the memory latencies of the backup, and handed over through LDS.  Same lanes, same functions,
// same bits as the in-line form (select_body without a helper).  `go` = 0: nothing to generate (the flag is set all
// the same: the tree wave may wait for it).
template <class GEO>
__device__ __forceinline__ void noise_wave(const View& v, int B, int go, uint64_t uid, uint32_t ply, int mb,
                                           double* s_nz, volatile int* s_flag) {
  constexpr int LPD = GEO::LPD, APL = GEO::APL, AP = GEO::AP;
  const int lane = threadIdx.x & 63;
  const int b = lane / LPD, l = lane % LPD;
  if (go) {
    double nz[APL];
    const uint64_t key = caro_noise_key(v.seed, uid, ply, (uint32_t)(mb * B + b));
    __shared__ double s_mail[APL > 1 ? 2 * AP : 1];  // (several actions per lane: LPD = 64, this wave is one row)
    noise_group<LPD, APL>(key, l, v.A, v.alpha, nz, s_mail);
#pragma unroll
    for (int j = 0; j < APL; ++j) s_nz[b * AP + l * APL + j] = nz[j];
  }
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
  __builtin_amdgcn_wave_barrier();
  if (lane == 0) *s_flag = 1;
}

// Fused form used by caro_search_batch (one 64-lane wavefront per game): expand + backup of the previous
// minibatch, then the descents of the next one on the updated tree, then row reservation + NN planes -- all
// per-game work, so one block does it back to back and a minibatch costs two launches (this + the net) instead
// of four.  rows_cur[0 / 1] count the leaves of this minibatch per net (the net kernel reads them), rows_next is
// cleared for the launch after this one.  Leaves, priors and values travel in slot rows (select_body).
template <class GEO>
__global__ void k_tree(View v, int B, int mb_index, const double* __restrict__ noise, const float* __restrict__ probs,
                       const float* __restrict__ values, float* __restrict__ planes, uint64_t* __restrict__ leaf_keys,
                       int32_t* __restrict__ rows_cur, int32_t* __restrict__ rows_next, int do_expand,
                       int do_select) {
  __shared__ double s_nz[MAXB * 0 + 64 * GEO::APL];  // [B][AP] with B x LPD = 64
  __shared__ int s_flag;
  if (threadIdx.x == 0) s_flag = 0;
  __syncthreads();  // the only s_barrier of the block: the flag is clear before the noise wave can set it
  if (threadIdx.x >= 64) {  // the noise wave
    const int g = blockIdx.x;
    const int go = do_select && !noise && !v.done[g];
    noise_wave<GEO>(v, B, go, go ? v.uid[g] : 0ull, go ? (uint32_t)v.ply[g] : 0u, mb_index, s_nz, &s_flag);
    return;
  }
  __builtin_amdgcn_s_setprio(3);  // (as in k_tree_stag: the latency-bound tree wave issues ahead of the noise waves)
  if (blockIdx.x == 0 && threadIdx.x == 0) {
    rows_next[0] = 0;
    rows_next[1] = 0;
    rows_next[2] = B;
    rows_cur[2] = B;
  }
  unsigned long long t0 = 0;
  if (v.dbg) t0 = __builtin_amdgcn_s_memtime();
  GameRegs<GEO> gr = load_game<GEO>(v, blockIdx.x);
  const ExpandPre<GEO> pre = expand_preload<GEO, true>(v, blockIdx.x, B, blockIdx.x * B, probs, values);  // one round with load_game
  if (do_expand) {
    expand_body<GEO, true>(v, gr, B, pre, blockIdx.x * B, probs);
    block_sync<true>();  // the block's own tree updates are visible to its descents
  }
  const unsigned long long t1 = v.dbg ? __builtin_amdgcn_s_memtime() : 0;
  if (v.dbg && threadIdx.x == 0 && !do_select) v.dbg[(size_t)blockIdx.x * 8 + 5] = t1 - t0;  // the closing launch: expand + backup alone
  if (do_select) select_body<GEO, true>(v, gr, B, mb_index, noise, rows_cur, planes, leaf_keys, s_nz, &s_flag);
  if (v.dbg && threadIdx.x == 0 && do_select) {  // a launch in the middle of a move: expand + backup | whole block
    v.dbg[(size_t)blockIdx.x * 8 + 6] = t1 - t0;
    v.dbg[(size_t)blockIdx.x * 8 + 7] = __builtin_amdgcn_s_memtime() - t0;
  }
}

// ------------------------------------------------------------------ policy / step
template <class GEO>
__device__ __forceinline__ void root_policy(const View& v, int g, int t, const typename GEO::R::Board& root,
                                            double* s_pi, int* s_n) {
  using R = typename GEO::R;
  constexpr int AP = GEO::AP;
  const int node = probe<R>(v, t, root);
  for (int a = threadIdx.x; a < AP; a += blockDim.x)
    s_n[a] = (node >= 0 && a < v.A) ? (int)(v.edges[((ebase(v, t) + node) * AP + a) * 4] & NMASK) : 0;
  __syncthreads();
  __shared__ int s_best;
  __shared__ double s_total;
  if (threadIdx.x == 0) {
    const int tau = (v.sbt0 > 0 && v.step[g] < v.sbt0) ? 1 : 0;  // utils.py:70,97-99
    int best = 0;
    long long tot = 0;
    for (int a = 0; a < v.A; ++a) {
      if (s_n[a] > s_n[best]) best = a;
      tot += s_n[a];
    }
    s_best = tau == 0 ? best : -1;
    s_total = (double)tot;
  }
  __syncthreads();
  for (int a = threadIdx.x; a < AP; a += blockDim.x) {
    double p = 0.0;
    if (a < v.A) p = s_best >= 0 ? (a == s_best ? 1.0 : 0.0) : (double)s_n[a] / s_total;  // mcts.py:305-311
    s_pi[a] = p;
  }
  __syncthreads();
}

template <class GEO>
__global__ void k_policy(View v, double* __restrict__ pi_out, int32_t* __restrict__ counts_out) {
  using R = typename GEO::R;
  constexpr int AP = GEO::AP, KW = GEO::KW;
  __shared__ double s_pi[AP];
  __shared__ int s_n[AP];
  const int g = blockIdx.x;
  const typename R::Board root = load_board<R>(v.root + (size_t)g * KW);
  const int t = g * v.n_stores + (v.n_stores == 2 ? v.player[g] : 0);
  root_policy<GEO>(v, g, t, root, s_pi, s_n);
  for (int a = threadIdx.x; a < v.A; a += blockDim.x) {
    if (pi_out) pi_out[(size_t)g * v.A + a] = s_pi[a];
    if (counts_out) counts_out[(size_t)g * v.A + a] = s_n[a];
  }
}

// One ply of play_game for game g (utils.py:80-99): pi from the root's visit counts, the history row, the sampled
// move, game.move, win / draw, the tau switch.  All threads of the block take part; `gr` (the game's scalars, in
// registers) is read instead of memory and comes back updated; returns (to every thread) 1 if the game has ended
// with this ply.  s_pi / s_n: AP entries of LDS each.
template <class GEO, bool ONE = false>
__device__ __forceinline__ int step_body(const View& v, int g, GameRegs<GEO>& gr, const double* __restrict__ uniforms,
                                         double* s_pi, int* s_n, int32_t* __restrict__ actions,
                                         int32_t* __restrict__ done_out, int32_t* __restrict__ result_out) {
  using R = typename GEO::R;
  using Board = typename R::Board;
  constexpr int AP = GEO::AP, KW = GEO::KW;
  __shared__ int s_action;
  __shared__ int s_best;
  __shared__ int s_refuse;
  __shared__ double s_total;
  Board root = gr.root;
  const int player = gr.player;
  // A root WITHOUT VISITS (one search on an unexpanded root, lib/mcts.py:123: nothing was backed up): the reference's
  // get_policy_value divides by zero at tau = 1 (mcts.py:311) and at tau = 0 plays argmax of an all-zero row = action 0.
  // Here: the ply is refused -- the game stays where it was, the tally every caller checks is bumped -- unless tau = 0
  // and action 0 is legal (then the reference's move is made).
  auto refuse = [&]() {
    if (threadIdx.x == 0) {
      atomicAdd(v.counters + (size_t)g * C_N + C_OVERFLOW, 1ull);
      if (actions) actions[g] = -1;
      if (done_out) done_out[g] = 0;
      if (result_out) result_out[g] = 0;
    }
  };
  const int st_sel = v.n_stores == 2 ? player : 0;
  const int t = g * v.n_stores + st_sel;
  const int tsel = st_sel ? gr.tbl[1] : gr.tbl[0];
  // get_policy_value (mcts.py:289-313): the root's visit counts -- its key and its N row are requested together from
  // the home slot (one latency); a collision falls back to the probe sequence
  {
    const uint32_t hs = home_slot<R>(v, t, root);
    const uint64_t* kp = v.node_key + (tbase_sel(v, t, tsel) + hs) * KW;
    const uint32_t* erow = v.edges + ebase_sel(v, t, tsel) * 4 * AP;
    uint32_t nraw[(AP + 63) / 64];
#pragma unroll
    for (int j = 0; j < (AP + 63) / 64; ++j) {
      const int a = threadIdx.x + j * 64;
      nraw[j] = a < AP && threadIdx.x < 64 ? erow[((size_t)hs * AP + a) * 4] : 0u;
    }
    bool eq = true;
#pragma unroll
    for (int w = 0; w < KW; ++w) eq = eq && (kp[w] == root.w[w]);
    int node = (int)hs;
    if (!eq) {
      node = kp[0] == EMPTY_KEY ? -1 : probe_from<R>(v, t, root, (hs + 1u) & ((uint32_t)v.hcap - 1u));
#pragma unroll
      for (int j = 0; j < (AP + 63) / 64; ++j) {
        const int a = threadIdx.x + j * 64;
        nraw[j] = node >= 0 && a < AP && threadIdx.x < 64 ? erow[((size_t)node * AP + a) * 4] : 0u;
      }
    }
#pragma unroll
    for (int j = 0; j < (AP + 63) / 64; ++j) {
      const int a = threadIdx.x + j * 64;
      if (a < AP && threadIdx.x < 64) s_n[a] = (node >= 0 && a < v.A) ? (int)(nraw[j] & NMASK) : 0;
    }
  }
  const int ply = gr.ply;
  const size_t hi = (size_t)g * v.maxply + ply;  // game_history.append((state, cur_player, probs)), utils.py:82
  int action;
  if constexpr (ONE && AP <= 64) {
    // One wavefront, one action per lane: the policy and the sampled move from registers.  Integer total and first
    // maximum by cross-lane reduction / ballot (exact); pi[a] = N[a] / total is the same float64 division in every
    // form; np.random.choice's cumulative sums are SEQUENTIAL float64 additions in action order: every lane runs the
    // same chain over the broadcast pi values (v_readlane: uniform index) and keeps the prefix of its own action, so
    // the comparisons acc / total <= u are the serial loop's, one per lane.
    block_sync<ONE>();
    const int lane = threadIdx.x;
    const int n = lane < AP ? s_n[lane] : 0;
    const int tau = (v.sbt0 > 0 && gr.step < v.sbt0) ? 1 : 0;  // utils.py:70,97-99
    const int tot = group_sum_i32<64>(n);
    const int nmax = group_allreduce_i32<64>(n, [](int x, int y) { return x > y ? x : y; });
    const int best = __ffsll((unsigned long long)__ballot(lane < v.A && n == nmax)) - 1;  // first maximum
    if (tot == 0 && (tau == 1 || !R::legal(v.gp, root, 0))) {  // uniform: every lane holds the same tot
      refuse();
      return 0;
    }
    double pa = 0.0;
    if (lane < v.A) pa = tau == 0 ? (lane == best ? 1.0 : 0.0) : (double)n / (double)tot;  // mcts.py:305-311
    if (lane < v.A) v.h_pi[hi * v.A + lane] = pa;
    if (lane == 0) {
      store_board<R>(v.h_key + hi * KW, root);
      v.h_player[hi] = player;
    }
    const double u = uniforms ? uniforms[g] : caro_move_uniform(v.seed, gr.uid, (uint32_t)ply);
    const uint64_t pbits = (uint64_t)__double_as_longlong(pa);
    double run = 0.0, mine = 0.0;
    for (int a2 = 0; a2 < v.A; ++a2) {  // caro_sample_index: tot = tot + pi[a]; acc = acc + pi[a] -- the same chain
      const uint32_t lo = (uint32_t)__builtin_amdgcn_readlane((int)(uint32_t)pbits, a2);
      const uint32_t hi32 = (uint32_t)__builtin_amdgcn_readlane((int)(uint32_t)(pbits >> 32), a2);
      run = run + __longlong_as_double((long long)(((uint64_t)hi32 << 32) | lo));
      if (a2 == lane) mine = run;
    }
    const unsigned long long below = __ballot(lane < v.A && mine / run <= u);  // acc / tot <= u  =>  idx = a + 1
    int idx = below ? 64 - __clzll((long long)below) : 0;
    action = idx < v.A ? idx : v.A - 1;
  } else {
  block_sync<ONE>();
  if (threadIdx.x == 0) {
    const int tau = (v.sbt0 > 0 && gr.step < v.sbt0) ? 1 : 0;  // utils.py:70,97-99
    int best = 0;
    long long tot = 0;
    for (int a = 0; a < v.A; ++a) {
      if (s_n[a] > s_n[best]) best = a;
      tot += s_n[a];
    }
    s_best = tau == 0 ? best : -1;
    s_total = (double)tot;
    s_refuse = tot == 0 && (tau == 1 || !R::legal(v.gp, root, 0));
  }
  block_sync<ONE>();
  if (s_refuse) {
    refuse();
    return 0;
  }
  for (int a = threadIdx.x; a < AP; a += block_threads<ONE>()) {
    double p = 0.0;
    if (a < v.A) p = s_best >= 0 ? (a == s_best ? 1.0 : 0.0) : (double)s_n[a] / s_total;  // mcts.py:305-311
    s_pi[a] = p;
  }
  block_sync<ONE>();
  for (int a = threadIdx.x; a < v.A; a += block_threads<ONE>()) v.h_pi[hi * v.A + a] = s_pi[a];
  if (threadIdx.x == 0) {
    store_board<R>(v.h_key + hi * KW, root);
    v.h_player[hi] = player;
    const double u = uniforms ? uniforms[g] : caro_move_uniform(v.seed, gr.uid, (uint32_t)ply);
    s_action = caro_sample_index(s_pi, v.A, u);  // np.random.choice(A, p=probs), utils.py:83
  }
  block_sync<ONE>();
    action = s_action;
  }
  // every thread replays the move on its own copy of the game (the same integers everywhere)
  const bool won = R::move(v.gp, root, action, player);  // utils.py:86
  gr.root = root;
  gr.ply = ply + 1;
  int done = 0, res = 0, final_r = 0;
  if (won) {  // utils.py:87-90
    done = 1;
    final_r = 1;
    res = player == 0 ? 1 : -1;
  } else {
    gr.player = 1 - player;
    if (R::full(v.gp, root)) {  // utils.py:93-96
      done = 1;
    } else {
      gr.step = gr.step + 1;  // utils.py:97
    }
  }
  gr.done = done;
  if (threadIdx.x == 0) {
    store_board<R>(v.root + (size_t)g * KW, root);
    v.ply[g] = gr.ply;
    v.player[g] = gr.player;
    v.step[g] = gr.step;
    unsigned long long* ctr = v.counters + (size_t)g * C_N;
    if (done) {
      v.final_r[g] = final_r;
      v.done[g] = 1;
      v.result[g] = res;
      atomicAdd(ctr + C_FINISHED, 1ull);  // no return value: nothing waits for the old count
    }
    atomicAdd(ctr + C_PLIES, 1ull);
    if (actions) actions[g] = action;
    if (done_out) done_out[g] = done;
    if (result_out) result_out[g] = res;
  }
  return done;
}

template <class GEO>
__global__ void k_step(View v, const double* __restrict__ uniforms, int32_t* __restrict__ actions,
                       int32_t* __restrict__ done_out, int32_t* __restrict__ result_out) {
  constexpr int AP = GEO::AP;
  __shared__ double s_pi[AP];
  __shared__ int s_n[AP];
  const int g = blockIdx.x;
  if (v.done[g]) {
    if (threadIdx.x == 0) {
      if (actions) actions[g] = -1;
      if (done_out) done_out[g] = 1;
      if (result_out) result_out[g] = v.result[g];
    }
    return;
  }
  GameRegs<GEO> gr = load_game<GEO>(v, g);
  step_body<GEO>(v, g, gr, uniforms, s_pi, s_n, actions, done_out, result_out);
}

// ------------------------------------------------------------------ eviction
// After a move only nodes whose board CONTAINS the new root can ever be looked up again (stones are never
// removed), so every other node is dropped: the survivors are re-inserted into the tree's second table, the
// old table is cleared on the way, and the tables swap roles.  Result-neutral: the reference keeps the dead
// nodes in its dict but can never reach them.  n_created (= len(MCTS)) keeps counting every node ever made.
template <class GEO>
__device__ __forceinline__ void evict_body(const View& v, int g, const typename GEO::R::Board& root, int done) {
  using R = typename GEO::R;
  using Board = typename R::Board;
  constexpr int AP = GEO::AP, KW = GEO::KW;
  // Round by round the block looks at blockDim slots of the old table: a thread whose slot holds a survivor claims a
  // slot of the new table (all keys are distinct: atomicCAS on word 0), writes the key there and QUEUES the pair
  // (old slot, new slot); then the whole block copies the queued rows together -- coalesced 16-byte accesses, AP of
  // them per row.  (One thread per row, 256 loads and stores in a chain on the 15 x 15 board, made the slowest block
  // of a launch ~180 us long once the eviction rode inside the staggered kernel's ply.)
  __shared__ int s_cnt, s_nq;
  __shared__ int2 s_q[1024];
  for (int st = 0; st < v.n_stores; ++st) {
    const int t = g * v.n_stores + st;
    const int live = v.tbl[t];
    const size_t ob = (size_t)(t * 2 + live) * v.tstride, nb = (size_t)(t * 2 + (1 - live)) * v.tstride;
    uint64_t* okeys = v.node_key + ob * KW;
    uint64_t* nkeys = v.node_key + nb * KW;
    const uint32_t mask = (uint32_t)v.hcap - 1u;
    if (threadIdx.x == 0) { s_cnt = 0; s_nq = 0; }
    __syncthreads();
    for (int base = 0; base < v.hcap; base += blockDim.x) {  // uniform trip count: barriers inside
      const int i = base + (int)threadIdx.x;
      if (i < v.hcap) {
        uint64_t* k = okeys + (size_t)i * KW;
        if (k[0] != EMPTY_KEY) {
          const Board b = load_board<R>(k);
          k[0] = EMPTY_KEY;  // the old table ends up empty
          if (!done && R::contains(v.gp, b, root)) {
            uint32_t j = home_slot<R>(v, t, b);
            for (int it = 0; it < v.hcap; ++it) {
              unsigned long long* w0 = (unsigned long long*)(nkeys + (size_t)j * KW);
              if (atomicCAS(w0, (unsigned long long)EMPTY_KEY, (unsigned long long)b.w[0]) == (unsigned long long)EMPTY_KEY) break;
              j = (j + 1u) & mask;
            }
            for (int w = 1; w < KW; ++w) nkeys[(size_t)j * KW + w] = b.w[w];
            s_q[atomicAdd(&s_nq, 1)] = make_int2(i, (int)j);
          }
        }
      }
      __syncthreads();
      const int nq = s_nq;
      for (int idx = threadIdx.x; idx < nq * AP; idx += blockDim.x) {
        const int e = idx / AP, q = idx - e * AP;
        const int2 pr = s_q[e];
        reinterpret_cast<uint4*>(v.edges + (nb + pr.y) * 4 * AP)[q] =
            reinterpret_cast<const uint4*>(v.edges + (ob + pr.x) * 4 * AP)[q];
      }
      __syncthreads();
      if (threadIdx.x == 0) {
        s_cnt += nq;
        s_nq = 0;
      }
      __syncthreads();
    }
    if (threadIdx.x == 0) {
      v.n_nodes[t] = s_cnt;
      v.tbl[t] = 1 - live;
    }
    __syncthreads();
  }
}

template <class GEO>
__global__ void k_evict(View v) {
  const int g = blockIdx.x;
  evict_body<GEO>(v, g, load_board<typename GEO::R>(v.root + (size_t)g * GEO::KW), v.done[g]);
}

// The fusion of k_tree for geometries with SEVERAL wavefronts per game (batch x lanes per descent a multiple of 64
// above 64: 15 x 15 with 8 descents = 8 waves): one block = one game does expand + backup of minibatch i - 1 (the first
// wavefront talks across its lanes, the whole block writes the new nodes' rows), a block barrier, the descents of
// minibatch i (one wavefront each) and the NN planes of the unique leaves into the game's slot rows: two launches per
// minibatch (this + the net) instead of four, same protocol as k_tree (+ the list of the leaves' slot rows, slot_list).
// do_step (the closing launch of a move, do_select = 0): the ply itself (step_body = k_step's code, lib/utils.py:80-99)
// and, with eviction on, k_evict's work for this game follow the last backup in the same block -- one launch per move
// instead of three.
template <class GEO>
__global__ void k_tree_mw(View v, int B, int mb_index, const double* __restrict__ noise, const float* __restrict__ probs,
                          const float* __restrict__ values, float* __restrict__ planes, uint64_t* __restrict__ leaf_keys,
                          int32_t* __restrict__ rows_cur, int32_t* __restrict__ rows_next, int do_expand, int do_select,
                          int do_step, const double* __restrict__ uniforms, int32_t* __restrict__ actions,
                          int32_t* __restrict__ done_out, int32_t* __restrict__ result_out) {
  constexpr int AP = GEO::AP;
  const int g = blockIdx.x;
  if (g == 0 && threadIdx.x == 0) {
    rows_next[0] = 0;
    rows_next[1] = 0;
    rows_next[2] = B;
    rows_cur[2] = B;
  }
  const unsigned long long t0 = v.dbg ? __builtin_amdgcn_s_memtime() : 0;
  GameRegs<GEO> gr = load_game<GEO>(v, g);
  const ExpandPre<GEO> pre = expand_preload<GEO, false>(v, g, B, g * B, probs, values);
  if (do_expand) {
    expand_body<GEO, false>(v, gr, B, pre, g * B, probs);
    __syncthreads();  // the block's own tree updates are visible to its descents (and to the ply)
  }
  const unsigned long long t1 = v.dbg ? __builtin_amdgcn_s_memtime() : 0;
  // (Generating the descents' Dirichlet rows BESIDE the first wavefront's expand + backup -- they do not depend on the
  // tree -- was built and measured: 113 us per launch, and 120 us with the first wavefront's row in front of the barrier as
  // well, against 99 us like this.  The kernel is bound by the arithmetic of those rows; with two blocks per compute unit
  // one block's latency phase (expand + backup, the descents) is what the other block's arithmetic runs under, and rows
  // generated at the top make both blocks compute at the same time.  NOTES, round 6.)
  if (do_select) select_body<GEO, false>(v, gr, B, mb_index, noise, rows_cur, planes, leaf_keys);
  if (do_step) {
    __shared__ double s_pi[AP];
    __shared__ int s_n[AP];
    if (gr.done) {
      if (threadIdx.x == 0) {
        if (actions) actions[g] = -1;
        if (done_out) done_out[g] = 1;
        if (result_out) result_out[g] = v.result[g];
      }
    } else {
      step_body<GEO>(v, g, gr, uniforms, s_pi, s_n, actions, done_out, result_out);
    }
    if (v.etab == 2) {
      __syncthreads();
      evict_body<GEO>(v, g, gr.root, gr.done);
    }
  }
  if (v.dbg && threadIdx.x == 0) {
    v.dbg[(size_t)g * 8 + (do_select ? 6 : 5)] = t1 - t0;
    if (do_select) v.dbg[(size_t)g * 8 + 7] = __builtin_amdgcn_s_memtime() - t0;
  }
}

// ------------------------------------------------------------------ reset / drain
// caro_config.games_limit: slot g's k-th game (uid = uid_base + g + k * uid_stride) belongs to the wanted set while
// k * G + g < games_limit.  (A 64-bit division on a path a game takes once, when it ends.)
__device__ __forceinline__ bool game_wanted(const View& v, int g, uint64_t uid) {
  if (v.games_limit <= 0) return true;
  const long long k = (long long)((uid - v.uid_base - (uint64_t)g) / v.uid_stride);
  return k * (long long)v.G + g < v.games_limit;
}

template <class GEO>
__device__ __forceinline__ void reset_game(const View& v, int g, uint64_t uid, int first) {
  using R = typename GEO::R;
  constexpr int KW = GEO::KW;
  for (int s = 0; s < v.n_stores; ++s) {
    const int t = g * v.n_stores + s;
    for (int tb = 0; tb < v.ntab; ++tb) {
      uint64_t* keys = v.node_key + (size_t)(t * v.ntab + tb) * v.tstride * KW;
      for (int i = threadIdx.x; i < v.hcap * KW; i += blockDim.x) keys[i] = EMPTY_KEY;
    }
    if (threadIdx.x == 0) {
      v.n_nodes[t] = 0;
      v.n_created[t] = 0;
      v.tbl[t] = 0;
    }
  }
  if (threadIdx.x == 0) {
    const typename R::Board b0 = R::initial(v.gp);
    store_board<R>(v.root + (size_t)g * KW, b0);
    int fp = first;
    if (fp < 0) fp = v.first_mode == 2 ? (int)(uid & 1ull) : v.first_mode;
    v.player[g] = fp;
    v.first[g] = fp;
    v.ply[g] = 0;
    v.step[g] = 0;
    v.uid[g] = uid;
    v.done[g] = game_wanted(v, g, uid) ? 0 : 2;  // beyond the wanted set: the slot never starts (2 = drained, finished)
    v.result[g] = 0;
    v.final_r[g] = 0;
  }
}

template <class GEO>
__global__ void k_reset(View v, const int32_t* __restrict__ first_player) {
  const int g = blockIdx.x;
  reset_game<GEO>(v, g, v.uid_base + (uint64_t)g, first_player ? first_player[g] : -1);
}

// ------------------------------------------------------------------ staggered mode
// Why: in lock-step all G games sit at the same minibatch index, and the first minibatches after a move carry far
// more new leaves than the later ones (connect four, 1024 games: 2312 / 1718 / 1610 / 1536 ... 1300, mean 1434) -- the
// net launch overflows one round of tiles (256 CUs x 6 boards) exactly there and pays a second, short round.  Games
// are independent, so nothing forces them to move at the same time: here every game keeps its OWN minibatch clock
// lm[g], the clocks are spread evenly over the S minibatches of a move (game g sits out g % S launches at the start),
// and every launch then sees 1/S of the games at each index: the same ~mean leaf count every time.  A game whose S
// minibatches are done makes its ply inside the tree kernel (step_body), and a finished game is parked and its slot
// restarted on the spot, so the clocks never re-align.  Per game nothing changes: the same minibatches in the same
// order on the same tree with the same noise keys (seed, uid, ply, sim) -- results equal the lock-step engine's and
// the oracle's game by game (tests/test_gpu_stagger.py).

// Moves the finished game of slot g aside (record + history rows) and restarts the slot; false if the previous
// parked game of this slot has not been drained yet (the game then stays finished and tries again next launch).
// ONE: the block's tree work is one wavefront (k_tree_stag); otherwise every thread of the block takes part (k_tree_stag_mw).
template <class GEO, bool ONE = true>
__device__ __forceinline__ bool park_and_restart(const View& v, int g, GameRegs<GEO>& gr) {
  using R = typename GEO::R;
  constexpr int KW = GEO::KW;
  const int nth = block_threads<ONE>();
  if constexpr (ONE) {
    if (v.pk_flag[g] == 1) return false;
  } else {  // (one reader: a wavefront that came late would see the flag this very call sets below)
    __shared__ int s_parked;
    if (threadIdx.x == 0) s_parked = v.pk_flag[g];
    __syncthreads();
    if (s_parked == 1) return false;
  }
  const int n = gr.ply;
  const size_t h0 = (size_t)g * v.maxply;
  for (int idx = threadIdx.x; idx < n * v.A; idx += nth) v.ph_pi[h0 * v.A + idx] = v.h_pi[h0 * v.A + idx];
  for (int idx = threadIdx.x; idx < n * KW; idx += nth) v.ph_key[h0 * KW + idx] = v.h_key[h0 * KW + idx];
  for (int j = threadIdx.x; j < n; j += nth) v.ph_player[h0 + j] = v.h_player[h0 + j];
  const uint64_t uid = gr.uid;
  if (threadIdx.x == 0) {
    v.pk_ply[g] = n;
    v.pk_final_r[g] = v.final_r[g];
    v.pk_first[g] = v.first[g];
    v.pk_result[g] = v.result[g];
    v.pk_step[g] = gr.step;
    v.pk_uid[g] = uid;
    v.pk_flag[g] = 1;
  }
  block_sync<ONE>();  // the live record has been read by every thread
  if (!v.stag_recycle || v.stag_pool || !game_wanted(v, g, uid + v.uid_stride)) {
    if (threadIdx.x == 0) v.done[g] = 2;  // no restart asked for: parked, the slot stays finished
    gr.done = 2;
    return false;
  }
  // restart without clearing anything on this wave's time: the slot's trees move to their OTHER key table, which is
  // clean (both are at creation; the one left behind is cleared by k_stag_clean at the next drain, and no slot
  // restarts twice between two drains: pk_flag above)
  const uint64_t nuid = uid + v.uid_stride;
  const int fp = v.first_mode == 2 ? (int)(nuid & 1ull) : v.first_mode;
  gr.root = R::initial(v.gp);
  gr.player = fp;
  gr.ply = 0;
  gr.step = 0;
  gr.uid = nuid;
  gr.done = 0;
  // (with eviction on, the ply that ended the game has dropped every node and left both tables clean -- evict_body on a
  // finished game --: nothing to flip, nothing to clean later)
  const bool flip = v.etab != 2;
#pragma unroll
  for (int st = 0; st < 2; ++st) {
    if (flip) gr.tbl[st] = 1 - gr.tbl[st];
    gr.nn[st] = 0;
  }
  if (threadIdx.x == 0) {
#pragma unroll
    for (int st = 0; st < 2; ++st) {  // unrolled: gr.tbl must stay in registers
      if (st < v.n_stores) {
        const int t = g * v.n_stores + st;
        if (flip) {
          v.tbl[t] = gr.tbl[st];
          v.dirty[t] = 1;
        }
        v.n_nodes[t] = 0;
        v.n_created[t] = 0;
      }
    }
    store_board<R>(v.root + (size_t)g * KW, gr.root);
    v.player[g] = fp;
    v.first[g] = fp;
    v.ply[g] = 0;
    v.step[g] = 0;
    v.uid[g] = nuid;
    v.done[g] = 0;
    v.result[g] = 0;
    v.final_r[g] = 0;
  }
  return true;
}

// The fused tree kernel of the staggered mode (one 64-lane wavefront per game; launch geometry and the slot-row
// interface to the net kernel are k_tree's): expand + backup of the game's pending minibatch, the ply if its S
// minibatches are done (+ park / restart if the game ended), then the descents of its next minibatch.
template <class GEO>
__global__ void k_tree_stag(View v, int B, const float* __restrict__ probs, const float* __restrict__ values,
                            float* __restrict__ planes, uint64_t* __restrict__ leaf_keys,
                            int32_t* __restrict__ rows_cur, int32_t* __restrict__ rows_next) {
  constexpr int AP = GEO::AP;
  __shared__ double s_pi[AP];
  __shared__ int s_n[AP];
  __shared__ double s_nz[64 * GEO::APL];  // [B][AP] with B x LPD = 64
  __shared__ int s_flag;
  const int g = blockIdx.x;
  if (threadIdx.x == 0) s_flag = 0;
  // One round of loads for everything that depends on g alone, by BOTH waves and in FRONT of the block's only
  // s_barrier: the barrier waits for them (the fence in __syncthreads drains vmcnt), so the noise wave holds the
  // game's uid / ply / clock as they were when the block started -- whatever the tree wave writes later (the ply in
  // step_body, the restart in park_and_restart, lm at the end) cannot reach it.  Both waves wait the same latency side
  // by side; the tree wave needed these values first thing anyway.
  const int w = v.wait[g];
  int lm = v.lm[g];
  const int pend = v.pend[g];
  GameRegs<GEO> gr = load_game<GEO>(v, g);
  // ... and, in the same round, everything the pending minibatch's expand + backup reads (tree wave; the addresses depend
  // on g and the lane only, so the loads are issued whether or not a minibatch is pending)
  ExpandPre<GEO> pre;
  if (threadIdx.x < 64) pre = expand_preload<GEO, true>(v, g, B, g * B, probs, values);
  __syncthreads();  // also: the flag is clear before the noise wave can set it
  if (threadIdx.x >= 64) {
    // the noise wave: rows of the minibatch the tree wave is about to select.  A game whose ply is due moves first:
    // its rows are those of the NEXT ply's minibatch 0 (if the ply ends the game the slot restarts on an empty tree,
    // or stays finished: no descent gets as far as using a row).
    const int go = w == 0 && gr.done == 0;
    int lm_h = go ? lm : 0;
    uint32_t ply_h = go ? (uint32_t)gr.ply : 0u;
    if (lm_h == v.stag_S) {
      lm_h = 0;
      ply_h += 1u;
    }
    noise_wave<GEO>(v, B, go, go ? gr.uid : 0ull, ply_h, lm_h, s_nz, &s_flag);
    return;
  }
  // The tree wave is a chain of memory latencies with short bursts of instructions between them; the noise waves that
  // share its SIMD (this block's or a neighbour's) are pure float64 arithmetic.  With the issue priority raised the tree
  // wave's bursts go first when its data arrive: +0.4 % on the headline (same-box A/B of two builds, four alternations:
  // 8.452-8.473 M against 8.418-8.433 M; NOTES, round 6).
  __builtin_amdgcn_s_setprio(3);
  if (blockIdx.x == 0 && threadIdx.x == 0) {
    rows_next[0] = 0;
    rows_next[1] = 0;
    rows_next[2] = B;
    rows_cur[2] = B;
  }
  const unsigned long long t0 = v.dbg ? __builtin_amdgcn_s_memtime() : 0;
  if (w > 0) {  // not started yet
    if (threadIdx.x == 0) {
      v.wait[g] = w - 1;
      v.g_nleaf[g] = 0;
      v.g_class[g] = 0;
      v.g_pack[g] = 0;
    }
    return;
  }
  if (gr.done == 2) {  // parked without restart: nothing left to do in this slot
    if (threadIdx.x == 0) {
      v.g_nleaf[g] = 0;
      v.g_class[g] = 0;
      v.g_pack[g] = 0;
    }
    return;
  }
  if (pend) {
    expand_body<GEO, true>(v, gr, B, pre, g * B, probs);
    block_sync<true>();  // the block's own tree updates are visible to what follows
  }
  const unsigned long long t1 = v.dbg ? __builtin_amdgcn_s_memtime() : 0;
  int over = gr.done == 1;  // finished earlier and could not be parked (its slot's previous game is not drained yet)
  if (!over && lm == v.stag_S) {
    over = step_body<GEO, true>(v, g, gr, nullptr, s_pi, s_n, nullptr, nullptr, nullptr);
    lm = 0;
  }
  if (over) {
    if (park_and_restart<GEO>(v, g, gr)) over = 0;  // a new game sits in the slot: its first minibatch follows
  }
  const unsigned long long t2 = v.dbg ? __builtin_amdgcn_s_memtime() : 0;
  // select_body returns at once (zero leaves) for a finished game
  select_body<GEO, true>(v, gr, B, lm, nullptr, rows_cur, planes, leaf_keys, s_nz, &s_flag);
  if (threadIdx.x == 0) {
    v.lm[g] = over ? 0 : lm + 1;
    v.pend[g] = over ? 0 : 1;
    if (v.dbg) {  // diagnostic stamps (tools/probe_stag.py): ply + park | expand + backup | whole block
      // (slot 5 carries the 100 MHz wall clock at the block's end above bit 24: tools/probe_engine_net.py's time line)
      v.dbg[(size_t)g * 8 + 5] = ((t2 - t1) & 0xFFFFFFull) | (__builtin_amdgcn_s_memrealtime() << 24);
      v.dbg[(size_t)g * 8 + 6] = t1 - t0;
      v.dbg[(size_t)g * 8 + 7] = __builtin_amdgcn_s_memtime() - t0;
      // where the tree wave ran (HW_ID: wave 3:0, SIMD 5:4, CU 11:8, SH 12, SE 15:13), above the depth in slot 4
      v.dbg[(size_t)g * 8 + 4] |= (unsigned long long)__builtin_amdgcn_s_getreg((31 << 11) | 4) << 8;
    }
  }
}

// The staggered tree kernel for geometries with SEVERAL wavefronts per game (round 6; k_tree_mw's block layout, k_tree_stag's
// protocol): expand + backup of the game's pending minibatch, the ply if its S minibatches are done -- with eviction on,
// followed by the eviction of what the move made unreachable (a finished game: of everything, which also leaves both
// key tables clean for the restart) --, park + restart if the game ended, then the descents of the next minibatch.
template <class GEO>
__global__ void k_tree_stag_mw(View v, int B, const float* __restrict__ probs, const float* __restrict__ values,
                               float* __restrict__ planes, uint64_t* __restrict__ leaf_keys,
                               int32_t* __restrict__ rows_cur, int32_t* __restrict__ rows_next) {
  constexpr int AP = GEO::AP;
  __shared__ double s_pi[AP];
  __shared__ int s_n[AP];
  const int g = blockIdx.x;
  if (g == 0 && threadIdx.x == 0) {
    rows_next[0] = 0;
    rows_next[1] = 0;
    rows_next[2] = B;
    rows_cur[2] = B;
  }
  // everything that depends on g alone, read by every thread BEFORE anything of it is written (lm / pend / wait at the
  // end, the ply, the restart): the barrier below keeps a late wavefront from reading a value this call has changed
  const int w = v.wait[g];
  int lm = v.lm[g];
  const int pend = v.pend[g];
  GameRegs<GEO> gr = load_game<GEO>(v, g);
  const ExpandPre<GEO> pre = expand_preload<GEO, false>(v, g, B, g * B, probs, values);
  __syncthreads();
  if (w > 0 || gr.done == 2) {  // not started yet / parked without restart: no leaves from this slot
    if (threadIdx.x == 0) {
      if (w > 0) v.wait[g] = w - 1;
      v.g_nleaf[g] = 0;
      v.g_class[g] = 0;
      v.g_pack[g] = 0;
    }
    return;
  }
  if (pend) {
    expand_body<GEO, false>(v, gr, B, pre, g * B, probs);
    __syncthreads();  // the block's own tree updates are visible to what follows
  }
  int over = gr.done == 1;  // finished earlier and could not be parked (its slot's previous game is not drained yet)
  if (!over && lm == v.stag_S) {
    over = step_body<GEO>(v, g, gr, nullptr, s_pi, s_n, nullptr, nullptr, nullptr);
    lm = 0;
    if (v.etab == 2) {
      __syncthreads();
      evict_body<GEO>(v, g, gr.root, gr.done);  // ends with a barrier: the flipped tables and the counts are visible
#pragma unroll
      for (int st = 0; st < 2; ++st)
        if (st < v.n_stores) {
          gr.tbl[st] = v.tbl[g * v.n_stores + st];
          gr.nn[st] = v.n_nodes[g * v.n_stores + st];
        }
    }
  }
  if (over) {
    __syncthreads();
    if (park_and_restart<GEO, false>(v, g, gr)) over = 0;  // a new game sits in the slot: its first minibatch follows
    __syncthreads();
  }
  // select_body returns at once (zero leaves) for a finished game
  select_body<GEO, false>(v, gr, B, lm, nullptr, rows_cur, planes, leaf_keys);
  if (threadIdx.x == 0) {
    v.lm[g] = over ? 0 : lm + 1;
    v.pend[g] = over ? 0 : 1;
  }
}

// Staggered pool mode (caro_config.stagger_recycle == 2), part of every drain: the slots whose game is over and parked
// (done == 2) get the next local indices of the wanted set that have not been started yet -- in slot order, by one
// block: which slot plays which game is a function of the games' progress alone.  Local index i = uid
// uid_base + i % G + (i / G) * uid_stride: the same SET of games the static layout plays.  The slot's trees move to their
// other, clean key table (the one left behind is cleared by k_stag_clean, launched right behind this kernel).
template <class GEO>
__global__ void k_stag_assign(View v) {
  using R = typename GEO::R;
  constexpr int KW = GEO::KW;
  __shared__ int s_c[1024];
  const int tid = threadIdx.x;
  const int chunk = (v.G + 1023) / 1024;
  const int lo = tid * chunk, hi = min(v.G, lo + chunk);
  int c = 0;
  for (int g = lo; g < hi; ++g) c += v.done[g] == 2;
  s_c[tid] = c;
  __syncthreads();
  for (int d = 1; d < 1024; d <<= 1) {
    const int a = tid >= d ? s_c[tid - d] : 0;
    __syncthreads();
    s_c[tid] += a;
    __syncthreads();
  }
  const int handed = v.handed[0];
  long long idx = (long long)handed + (s_c[tid] - c);
  const int total = s_c[1023];
  __syncthreads();
  for (int g = lo; g < hi; ++g) {
    if (v.done[g] != 2) continue;
    if (idx < v.games_limit) {
      const uint64_t uid = v.uid_base + (uint64_t)(idx % v.G) + (uint64_t)(idx / v.G) * v.uid_stride;
      const int fp = v.first_mode == 2 ? (int)(uid & 1ull) : v.first_mode;
      for (int st = 0; st < v.n_stores; ++st) {
        const int t = g * v.n_stores + st;
        if (v.etab != 2) {  // (with eviction both tables are clean already: the finished game's ply dropped every node)
          v.tbl[t] = 1 - v.tbl[t];
          v.dirty[t] = 1;
        }
        v.n_nodes[t] = 0;
        v.n_created[t] = 0;
      }
      store_board<R>(v.root + (size_t)g * KW, R::initial(v.gp));
      v.player[g] = fp;
      v.first[g] = fp;
      v.ply[g] = 0;
      v.step[g] = 0;
      v.uid[g] = uid;
      v.result[g] = 0;
      v.final_r[g] = 0;
      v.lm[g] = 0;
      v.pend[g] = 0;
      v.wait[g] = 0;
      v.done[g] = 0;
    }
    ++idx;
  }
  if (tid == 0) {
    const long long h = (long long)handed + total;
    v.handed[0] = (int)(h < v.games_limit ? h : v.games_limit);
  }
}

__global__ void k_stag_init(View v) {
  const int g = blockIdx.x * blockDim.x + threadIdx.x;
  if (g == 0 && v.handed) v.handed[0] = (int)(v.games_limit > 0 && v.games_limit < v.G ? v.games_limit : v.G);
  if (g >= v.G) return;
  v.lm[g] = 0;
  v.pend[g] = 0;
  v.wait[g] = g % v.stag_S;
  v.pk_flag[g] = 0;
}

// clears the key table a restarted slot left behind (256 threads per tree; a no-op for every other tree)
template <class GEO>
__global__ void k_stag_clean(View v) {
  constexpr int KW = GEO::KW;
  const int t = blockIdx.x;
  if (!v.dirty[t]) return;
  uint64_t* keys = v.node_key + (size_t)(t * 2 + (1 - v.tbl[t])) * v.tstride * KW;
  for (int i = threadIdx.x; i < v.hcap * KW; i += blockDim.x) keys[i] = EMPTY_KEY;
  if (threadIdx.x == 0) v.dirty[t] = 0;
}

template <class GEO>
__global__ void k_set_roots(View v, const uint64_t* __restrict__ keys, const int32_t* __restrict__ players) {
  constexpr int KW = GEO::KW;
  const int g = blockIdx.x * blockDim.x + threadIdx.x;
  if (g >= v.G) return;
  for (int w = 0; w < KW; ++w) v.root[(size_t)g * KW + w] = keys[(size_t)g * KW + w];
  v.player[g] = players[g];
  v.done[g] = 0;
}

// which finished games fit into `cap` tuples: exclusive scan of their ply counts (single block)
__global__ void k_drain_scan(View v, long long cap) {
  __shared__ long long s_t[1024];
  __shared__ int s_g[1024];
  const int tid = threadIdx.x;
  const int chunk = (v.G + 1023) / 1024;
  const int lo = tid * chunk, hi = min(v.G, lo + chunk);
  long long ct = 0;
  int cg = 0;
  for (int g = lo; g < hi; ++g)
    if (v.done[g] == 1) { ct += v.ply[g]; cg += 1; }
  s_t[tid] = ct;
  s_g[tid] = cg;
  __syncthreads();
  for (int d = 1; d < 1024; d <<= 1) {
    const long long a = tid >= d ? s_t[tid - d] : 0;
    const int b = tid >= d ? s_g[tid - d] : 0;
    __syncthreads();
    s_t[tid] += a;
    s_g[tid] += b;
    __syncthreads();
  }
  long long ot = s_t[tid] - ct;
  int og = s_g[tid] - cg;
  __shared__ long long s_tot;
  __shared__ int s_totg;
  if (tid == 0) { s_tot = 0; s_totg = 0; }
  __syncthreads();
  for (int g = lo; g < hi; ++g) {
    int sel = 0;
    if (v.done[g] == 1) {
      if (ot + v.ply[g] <= cap) {
        sel = 1;
        v.dr_off[g] = (int)ot;
        v.dr_gidx[g] = og;
        atomicMax((unsigned long long*)&s_tot, (unsigned long long)(ot + v.ply[g]));
        atomicMax(&s_totg, og + 1);
      }
      ot += v.ply[g];
      og += 1;
    }
    v.dr_sel[g] = sel;
  }
  __syncthreads();
  if (tid == 0) {
    v.dr_tot[0] = s_tot;
    v.dr_tot[1] = s_totg;
  }
}

template <class GEO>
__global__ void k_drain_copy(View v, uint64_t* __restrict__ states, int32_t* __restrict__ players,
                             double* __restrict__ pi, int32_t* __restrict__ z, int64_t* __restrict__ games,
                             int recycle) {
  constexpr int KW = GEO::KW;
  const int g = blockIdx.x;
  if (!v.dr_sel[g]) return;
  const int n = v.ply[g];
  const int off = v.dr_off[g];
  const int r = v.final_r[g];
  const uint64_t uid0 = v.uid[g];  // read by every thread BEFORE the barrier below: thread 0 rewrites it in reset_game
  // reversed(game_history), result alternating from the last mover (utils.py:101-106)
  for (int idx = threadIdx.x; idx < n * v.A; idx += blockDim.x) {
    const int j = idx / v.A, a = idx % v.A;
    const int plyi = n - 1 - j;
    pi[((size_t)off + j) * v.A + a] = v.h_pi[((size_t)g * v.maxply + plyi) * v.A + a];
  }
  for (int j = threadIdx.x; j < n; j += blockDim.x) {
    const int plyi = n - 1 - j;
    const size_t hi = (size_t)g * v.maxply + plyi;
    for (int w = 0; w < KW; ++w) states[((size_t)off + j) * KW + w] = v.h_key[hi * KW + w];
    players[off + j] = v.h_player[hi];
    z[off + j] = (j & 1) ? -r : r;
  }
  if (threadIdx.x == 0 && games) {
    int64_t* rec = games + (size_t)v.dr_gidx[g] * 4;
    rec[0] = (int64_t)uid0;
    rec[1] = v.first[g];
    rec[2] = v.result[g];
    rec[3] = v.step[g];
  }
  __syncthreads();
  if (recycle && game_wanted(v, g, uid0 + v.uid_stride)) {
    reset_game<GEO>(v, g, uid0 + v.uid_stride, -1);
  } else if (threadIdx.x == 0) {
    v.done[g] = 2;  // drained, stays finished
  }
}

__global__ void k_sum_counters(View v) {
  __shared__ unsigned long long s[256];
  for (int c = 0; c < C_N; ++c) {
    unsigned long long acc = 0;
    for (int g = threadIdx.x; g < v.G; g += blockDim.x) acc += v.counters[(size_t)g * C_N + c];
    s[threadIdx.x] = acc;
    __syncthreads();
    for (int d = 128; d > 0; d >>= 1) {
      if (threadIdx.x < d) s[threadIdx.x] += s[threadIdx.x + d];
      __syncthreads();
    }
    if (threadIdx.x == 0) v.counters_sum[c] = s[0];
    __syncthreads();
  }
}

__global__ void k_count_live(View v, int32_t* out) {
  int c = 0;
  for (int g = threadIdx.x; g < v.G; g += blockDim.x) c += v.done[g] == 0;
  atomicAdd(out, c);
}

// unique leaves selected but not booked as expansions yet: the counters' identity at any point of a run is
// sims == expansions + terminals + dropped + pending (+ the leaves of overflowed minibatches)
__global__ void k_count_pending(View v, int all_pending, int32_t* out) {
  int c = 0;
  for (int g = threadIdx.x; g < v.G; g += blockDim.x)
    if (v.stag_S ? v.pend[g] != 0 : all_pending) c += v.g_nleaf[g];
  atomicAdd(out, c);
}

// ------------------------------------------------------------------ inspection
template <class GEO>
__global__ void k_lookup(View v, long long M, const int32_t* __restrict__ game, const int32_t* __restrict__ store,
                         const uint64_t* __restrict__ keys, int32_t* found, int32_t* N, float* W, float* Q, float* P,
                         int32_t* strong) {
  using R = typename GEO::R;
  constexpr int AP = GEO::AP, KW = GEO::KW;
  const long long m = blockIdx.x;
  if (m >= M) return;
  const int t = game[m] * v.n_stores + store[m];
  const typename R::Board b = load_board<R>(keys + (size_t)m * KW);
  const int node = probe<R>(v, t, b);
  if (threadIdx.x == 0) found[m] = node >= 0;
  if (node < 0) return;
  const uint32_t* row = v.edges + (ebase(v, t) + node) * 4 * AP;
  for (int a = threadIdx.x; a < v.A; a += blockDim.x) {
    const size_t o = (size_t)m * v.A + a;
    N[o] = (int)(row[4 * a] & NMASK);
    strong[o] = (row[4 * a] & NSTRONG) ? 1 : 0;
    W[o] = __uint_as_float(row[4 * a + 1]);
    Q[o] = __uint_as_float(row[4 * a + 2]);
    P[o] = __uint_as_float(row[4 * a + 3]);
  }
}

template <class GEO>
__global__ void k_poke(View v, long long M, const int32_t* __restrict__ game, const int32_t* __restrict__ store,
                       const uint64_t* __restrict__ keys, const int32_t* N, const float* W, const float* Q,
                       const float* P, const int32_t* strong) {
  using R = typename GEO::R;
  constexpr int AP = GEO::AP, KW = GEO::KW;
  // single block, sequential over m: insertion order is deterministic
  for (long long m = 0; m < M; ++m) {
    const int t = game[m] * v.n_stores + store[m];
    const typename R::Board b = load_board<R>(keys + (size_t)m * KW);
    __shared__ int s_node;
    if (threadIdx.x == 0) {
      int node = probe<R>(v, t, b);
      if (node < 0 && v.n_nodes[t] < v.cap) {
        node = insert_key<R>(v, t, b);
        if (node >= 0) {
          v.n_nodes[t]++;
          v.n_created[t]++;
        }
      }
      s_node = node;
    }
    __syncthreads();
    const int node = s_node;
    if (node >= 0) {
      uint32_t* row = v.edges + (ebase(v, t) + node) * 4 * AP;
      for (int a = threadIdx.x; a < AP; a += blockDim.x) {
        const size_t o = (size_t)m * v.A + a;
        const bool in = a < v.A;
        row[4 * a] = in ? ((uint32_t)N[o] | (strong[o] ? NSTRONG : 0u)) : 0u;
        row[4 * a + 1] = in ? __float_as_uint(W[o]) : 0u;
        row[4 * a + 2] = in ? __float_as_uint(Q[o]) : 0u;
        row[4 * a + 3] = in ? __float_as_uint(P[o]) : 0u;
      }
    }
    __syncthreads();
  }
}

template <class GEO>
__global__ void k_backup_one(View v, int game, int store, float value, int strong, int len,
                             const uint64_t* __restrict__ keys, const int32_t* __restrict__ actions, int32_t* scratch) {
  using R = typename GEO::R;
  constexpr int AP = GEO::AP, KW = GEO::KW;
  if (threadIdx.x != 0 || blockIdx.x != 0) return;
  const int t = game * v.n_stores + store;
  for (int i = 0; i < len; ++i) scratch[i] = probe<R>(v, t, load_board<R>(keys + (size_t)i * KW));
  for (int i = 0; i < len; ++i)
    if (scratch[i] < 0) return;  // KeyError in the reference
  backup_path<AP>(v, t, value, strong != 0, scratch, actions, len);
}

template <class GEO>
__global__ void k_dump(View v, int game, int store, long long cap, uint64_t* keys, int32_t* N, float* W, float* Q,
                       float* P, int32_t* strong, int32_t* cursor) {
  constexpr int AP = GEO::AP, KW = GEO::KW;
  const int t = game * v.n_stores + store;
  const long long slot = blockIdx.x;
  if (slot >= v.hcap) return;
  const uint64_t* k = v.node_key + (tbase(v, t) + slot) * KW;
  if (k[0] == EMPTY_KEY) return;
  __shared__ int s_out;
  if (threadIdx.x == 0) s_out = atomicAdd(cursor, 1);
  __syncthreads();
  const long long node = s_out;
  if (node >= cap) return;
  const uint32_t* row = v.edges + (ebase(v, t) + slot) * 4 * AP;
  for (int a = threadIdx.x; a < v.A; a += blockDim.x) {
    const size_t o = (size_t)node * v.A + a;
    N[o] = (int)(row[4 * a] & NMASK);
    strong[o] = (row[4 * a] & NSTRONG) ? 1 : 0;
    W[o] = __uint_as_float(row[4 * a + 1]);
    Q[o] = __uint_as_float(row[4 * a + 2]);
    P[o] = __uint_as_float(row[4 * a + 3]);
  }
  if (threadIdx.x < KW) keys[(size_t)node * KW + threadIdx.x] = k[threadIdx.x];
}

// one descent of the pending select, for MCTS.find_leaf (lib/mcts.py:97-148)
template <class GEO>
__global__ void k_get_descent(View v, int game, int b, int32_t* info, float* value, uint64_t* leaf_key,
                              uint64_t* path_keys, int32_t* path_actions) {
  constexpr int KW = GEO::KW;
  const size_t di = (size_t)game * v.maxB + b;
  const uint4 rec = v.d_rec[di];
  const int len = (int)((rec.x >> 8) & 0xFFu);
  const int t = v.g_tree[game];
  if (threadIdx.x == 0) {
    info[0] = (int)(rec.x & 0xFFu);
    info[1] = len;
    info[2] = (int)(rec.x >> 24);
    info[3] = (int)((rec.x >> 16) & 0xFFu);
    *value = __uint_as_float(rec.y);
    for (int w = 0; w < KW; ++w) leaf_key[w] = v.d_key[di * KW + w];
  }
  for (int i = threadIdx.x; i < len; i += blockDim.x) {
    const uint32_t na = v.path_rec[di * v.maxd + i].x;
    path_actions[i] = (int)(na >> 24);
    for (int w = 0; w < KW; ++w) path_keys[(size_t)i * KW + w] = v.node_key[(tbase(v, t) + (na & 0xFFFFFFu)) * KW + w];
  }
}

__global__ void k_tree_sizes(View v, int32_t* out) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < v.G * v.n_stores) out[i] = v.n_created[i];
}

__global__ void k_tree_live(View v, int32_t* out) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < v.G * v.n_stores) out[i] = v.n_nodes[i];
}

template <class GEO>
__global__ void k_get_roots(View v, uint64_t* keys, int32_t* players, int32_t* ply, uint64_t* uid) {
  constexpr int KW = GEO::KW;
  const int g = blockIdx.x * blockDim.x + threadIdx.x;
  if (g >= v.G) return;
  if (keys)
    for (int w = 0; w < KW; ++w) keys[(size_t)g * KW + w] = v.root[(size_t)g * KW + w];
  if (players) players[g] = v.player[g];
  if (ply) ply[g] = v.ply[g];
  if (uid) uid[g] = v.uid[g];
}

// ------------------------------------------------------------------ batched rules
template <class GEO>
__global__ void k_rules_move(GameParams gp, long long M, uint64_t* keys, const int32_t* moves, const int32_t* players,
                             int32_t* won, int32_t* full) {
  using R = typename GEO::R;
  constexpr int KW = GEO::KW;
  const long long m = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (m >= M) return;
  typename R::Board b = load_board<R>(keys + (size_t)m * KW);
  const bool w = R::move(gp, b, moves[m], players[m]);
  store_board<R>(keys + (size_t)m * KW, b);
  won[m] = w;
  full[m] = R::full(gp, b);
}
template <class GEO>
__global__ void k_rules_legal(GameParams gp, long long M, const uint64_t* keys, uint8_t* legal) {
  using R = typename GEO::R;
  constexpr int KW = GEO::KW;
  const long long idx = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (idx >= M * gp.A) return;
  const long long m = idx / gp.A;
  const int a = (int)(idx % gp.A);
  const typename R::Board b = load_board<R>(keys + (size_t)m * KW);
  legal[idx] = R::legal(gp, b, a);
}
template <class GEO>
__global__ void k_rules_encode(GameParams gp, long long M, const uint64_t* keys, const int32_t* who, float* planes) {
  using R = typename GEO::R;
  constexpr int KW = GEO::KW;
  const int HW = gp.rows * gp.cols;
  const long long idx = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (idx >= M * 2 * HW) return;
  const long long m = idx / (2 * HW);
  const int i = (int)(idx % (2 * HW));
  const typename R::Board b = load_board<R>(keys + (size_t)m * KW);
  planes[idx] = R::plane(gp, b, who[m], i / HW, i % HW);
}
template <class GEO>
__global__ void k_noise(uint64_t seed, long long M, int A, double alpha, const uint64_t* uid, const uint32_t* ply,
                        const uint32_t* sim, double* out) {
  constexpr int LPD = GEO::LPD, APL = GEO::APL;
  const int grp = threadIdx.x / LPD, l = threadIdx.x % LPD;
  const long long m = (long long)blockIdx.x * (blockDim.x / LPD) + grp;
  const long long mm = m < M ? m : M - 1;  // keep whole groups active for the shuffles
  const uint64_t key = caro_noise_key(seed, uid[mm], ply[mm], sim[mm]);
  double nz[APL];
  __shared__ double s_mail[APL > 1 ? 2 * GEO::AP : 1];  // one wave per block
  noise_group<LPD, APL>(key, l, A, alpha, nz, s_mail);
  if (m < M)
    for (int j = 0; j < APL; ++j) {
      const int a = l * APL + j;
      if (a < A) out[(size_t)m * A + a] = nz[j];
    }
}

}  // namespace caro

// =================================================================== host side
using namespace caro;

static thread_local std::string g_err;
static int fail(int code, const std::string& msg) {
  g_err = msg;
  return code;
}
#define HIPCHK(x)                                                                                  \
  do {                                                                                             \
    hipError_t e_ = (x);                                                                           \
    if (e_ != hipSuccess)                                                                          \
      return fail(CARO_E_HIP, std::string(#x) + ": " + hipGetErrorString(e_));                     \
  } while (0)

struct caro_engine {
  caro_config cfg;
  Variant var;
  View v;
  std::vector<void*> allocs;
  int32_t* pinned;   // host pinned [8]
  int64_t* pinned64; // host pinned [8]
  int32_t* scratch;  // device i32 [maxply]
  int32_t* live;     // device i32
  int select_pending;
  int drain_pending;       // caro_drain_tuples_begin without its _end
  int stag_batch;          // staggered mode: the batch size of the first caro_search_staggered call
  hipEvent_t drain_ev;     // the totals of that drain have reached pinned memory
  // optional HIP-event timing of the hot kernels (bench.py's live roofline)
  int prof_on;
  int prof_gate;  // 0: skip event records for this launch (sampling inside caro_search_batch)
  uint64_t prof_ctr;  // minibatches enqueued by caro_search_batch since the engine was created
  uint64_t prof_cal;  // sampled minibatches seen by prof_calibrate
  int32_t* rows;      // [2][4] leaf counters of the fused tree kernel (ping-pong)
  int rows_par;
  int fused_ok;       // CARO_NO_FUSED_TREE=1 in the environment selects the four-launch form (A/B measurements)
  std::vector<hipEvent_t> ev;      // pairs: [2*i] start, [2*i+1] stop
  std::vector<int> ev_kind;        // kernel id of pair i
  size_t ev_used;
  double prof_ms[8];
  long long prof_n[8];
};

constexpr unsigned PROF_EVERY = 23;  // HIP-event pairs around every 23rd minibatch's launches (search_batch_impl)
enum ProfKind { PK_SELECT = 0, PK_COMPACT = 1, PK_EXPAND = 2, PK_STEP = 3, PK_NET = 4, PK_NULL1 = 5, PK_NULL2 = 6, PK_N = 8 };
// Calibration of the event pairs themselves.  A pair around a kernel reads  E = K + o  (o: what bracketing adds -- the
// dispatch behind an event's barrier packet; ~3 us, rocprofv3 sees K alone).  An EMPTY pair does not measure o (two
// barrier packets back to back: 5 us).  So now and then two more pairs are recorded behind a sampled net launch: one
// around ONE launch of an empty kernel, one around TWO:  E1 = K0 + o,  E2 = 2 K0 + g + o  (g: the gap between two
// dependent launches, < 1 us)  =>  o = 2 E1 - E2 + g.  The reader subtracts 2 E1 - E2 (o underestimated by g: the
// kernels' times stay on the conservative side).
__global__ void k_prof_null() {}
static void prof_calibrate(caro_engine* h, hipStream_t st);

static void prof_flush(caro_engine* h) {
  for (size_t i = 0; i < h->ev_used; ++i) {
    float ms = 0.f;
    if (hipEventSynchronize(h->ev[2 * i + 1]) == hipSuccess &&
        hipEventElapsedTime(&ms, h->ev[2 * i], h->ev[2 * i + 1]) == hipSuccess) {
      h->prof_ms[h->ev_kind[i]] += ms;
      h->prof_n[h->ev_kind[i]] += 1;
    }
  }
  h->ev_used = 0;
}
static int prof_begin(caro_engine* h, int kind, hipStream_t st) {
  if (!h->prof_on || !h->prof_gate) return -1;
  if (h->ev_used * 2 + 2 > h->ev.size()) {
    if (h->ev.size() >= 2 * 16384) prof_flush(h);
    else
      for (int i = 0; i < 512; ++i) {
        hipEvent_t e;
        if (hipEventCreate(&e) != hipSuccess) return -1;
        h->ev.push_back(e);
      }
  }
  const int i = (int)h->ev_used++;
  if ((size_t)i >= h->ev_kind.size()) h->ev_kind.resize(i + 1);
  h->ev_kind[i] = kind;
  (void)hipEventRecord(h->ev[2 * i], st);
  return i;
}
static void prof_end(caro_engine* h, int i, hipStream_t st) {
  if (i >= 0) (void)hipEventRecord(h->ev[2 * i + 1], st);
}
static void prof_calibrate(caro_engine* h, hipStream_t st) {
  if (!h->prof_on || !h->prof_gate || (h->prof_cal++ & 3)) return;  // every 4th sampled minibatch: 3 empty launches
  const int a = prof_begin(h, PK_NULL1, st);
  hipLaunchKernelGGL(k_prof_null, dim3(1), dim3(64), 0, st);
  prof_end(h, a, st);
  const int b = prof_begin(h, PK_NULL2, st);
  hipLaunchKernelGGL(k_prof_null, dim3(1), dim3(64), 0, st);
  hipLaunchKernelGGL(k_prof_null, dim3(1), dim3(64), 0, st);
  prof_end(h, b, st);
}

template <class T>
static int dalloc(caro_engine* h, T** p, size_t n) {
  void* q = nullptr;
  hipError_t e = hipMalloc(&q, n * sizeof(T) + 256);
  if (e != hipSuccess) return fail(CARO_E_NOMEM, std::string("hipMalloc failed: ") + hipGetErrorString(e));
  h->allocs.push_back(q);
  *p = (T*)q;
  return 0;
}

extern "C" {

const char* caro_last_error(void) { return g_err.c_str(); }
void caro__set_error(const char* msg) { g_err = msg ? msg : ""; }  // for the other translation units
int caro_version(void) { return 100; }

#include "caro_host.inc"

// ---- batched rules
int caro_rules_move_batch(int kind, int n, int k, int64_t M, uint64_t* keys, const int32_t* moves,
                          const int32_t* players, int32_t* won, int32_t* full, void* stream) {
  const Variant var = pick_variant(kind, n);
  const GameParams gp = make_gp(kind, n, k);
  if (M <= 0) return 0;
  const unsigned grid = (unsigned)((M + 255) / 256);
  DISPATCH(var, hipLaunchKernelGGL(k_rules_move<GEO>, dim3(grid), dim3(256), 0, (hipStream_t)stream, gp,
                                   (long long)M, keys, moves, players, won, full));
  HIPCHK(hipGetLastError());
  return 0;
}
int caro_rules_legal_batch(int kind, int n, int k, int64_t M, const uint64_t* keys, uint8_t* legal, void* stream) {
  const Variant var = pick_variant(kind, n);
  const GameParams gp = make_gp(kind, n, k);
  if (M <= 0) return 0;
  const unsigned grid = (unsigned)((M * gp.A + 255) / 256);
  DISPATCH(var, hipLaunchKernelGGL(k_rules_legal<GEO>, dim3(grid), dim3(256), 0, (hipStream_t)stream, gp,
                                   (long long)M, keys, legal));
  HIPCHK(hipGetLastError());
  return 0;
}
int caro_rules_encode_batch(int kind, int n, int k, int64_t M, const uint64_t* keys, const int32_t* who,
                            float* planes, void* stream) {
  const Variant var = pick_variant(kind, n);
  const GameParams gp = make_gp(kind, n, k);
  if (M <= 0) return 0;
  const unsigned grid = (unsigned)((M * 2 * gp.rows * gp.cols + 255) / 256);
  DISPATCH(var, hipLaunchKernelGGL(k_rules_encode<GEO>, dim3(grid), dim3(256), 0, (hipStream_t)stream, gp,
                                   (long long)M, keys, who, planes));
  HIPCHK(hipGetLastError());
  return 0;
}
int caro_noise_batch(uint64_t seed, int64_t M, int A, double alpha, const uint64_t* uid, const uint32_t* ply,
                     const uint32_t* sim, double* out, void* stream) {
  if (M <= 0) return 0;
  // same lane geometry the select kernel uses for this action count
  Variant var = A == 7 ? V_C4 : A <= 16 ? V_M16 : A <= 32 ? V_M32 : A <= 64 ? V_M64 : A <= 128 ? V_M128 : V_M256;
  if (A > 256) return fail(CARO_E_INVAL, "A out of range");
  const int lpd = variant_lpd(var);
  const int per_block = 64 / lpd;
  const unsigned grid = (unsigned)((M + per_block - 1) / per_block);
  DISPATCH(var, hipLaunchKernelGGL(k_noise<GEO>, dim3(grid), dim3(64), 0, (hipStream_t)stream, seed, (long long)M, A,
                                   alpha, uid, ply, sim, out));
  HIPCHK(hipGetLastError());
  return 0;
}

// ---- engine
static int reset_games_impl(caro_engine* h, const int32_t* first_player_dev, void* stream);
// the part of a configuration that does not shape memory: what a run on the engine is keyed by (create and restart)
static void apply_run_params(caro_engine* h, const caro_config* cfg) {
  View& v = h->v;
  v.sbt0 = cfg->steps_before_tau_0;
  v.first_mode = cfg->first_player_mode;
  v.c_puct = cfg->c_puct;
  v.alpha = cfg->alpha;
  v.explore = cfg->explore;
  v.seed = cfg->seed;
  v.uid_base = cfg->uid_base;
  v.uid_stride = cfg->uid_stride ? cfg->uid_stride : (uint64_t)cfg->n_games;
  v.games_limit = cfg->games_limit > 0 ? (long long)cfg->games_limit : 0;
  if (cfg->stagger > 0) {
    v.stag_S = cfg->stagger;
    v.stag_recycle = cfg->stagger_recycle ? 1 : 0;
    v.stag_pool = (cfg->stagger_recycle == 2 && cfg->games_limit > 0) ? 1 : 0;
  }
}
// everything a fresh engine starts from, enqueued on `stream`: games at the initial position, empty trees (every key
// table cleared, the first one live), zero counters, no pending minibatch, fresh clocks, nothing parked
static int fresh_state(caro_engine* h, hipStream_t st) {
  View& v = h->v;
  const size_t T = (size_t)v.G * v.n_stores, G = (size_t)v.G;
  HIPCHK(hipMemsetAsync(v.counters, 0, sizeof(unsigned long long) * C_N * G, st));
  HIPCHK(hipMemsetAsync(v.leaf_count, 0, sizeof(int32_t) * 4, st));
  HIPCHK(hipMemsetAsync(h->rows, 0, sizeof(int32_t) * 8, st));
  HIPCHK(hipMemsetAsync(v.g_nleaf, 0, sizeof(int32_t) * G, st));
  HIPCHK(hipMemsetAsync(v.g_pack, 0, sizeof(int32_t) * G, st));
  h->rows_par = 0;
  h->stag_batch = 0;
  const int rc = reset_games_impl(h, nullptr, st);
  if (rc) return rc;
  if (v.stag_S) {
    HIPCHK(hipMemsetAsync(v.dirty, 0, sizeof(int32_t) * T, st));
    hipLaunchKernelGGL(k_stag_init, dim3((v.G + 255) / 256), dim3(256), 0, st, v);
    HIPCHK(hipGetLastError());
  }
  return 0;
}
int caro_engine_create(const caro_config* cfg, caro_engine** out) {
  if (!cfg || !out) return fail(CARO_E_INVAL, "null argument");
  const Variant var = pick_variant(cfg->game_kind, cfg->n);
  if (var == V_BAD) return fail(CARO_E_INVAL, "unsupported game (connect four, or m,n,k with 2 <= n <= 15)");
  if (cfg->n_games < 1) return fail(CARO_E_INVAL, "n_games must be >= 1");
  if (cfg->n_stores != 1 && cfg->n_stores != 2) return fail(CARO_E_INVAL, "n_stores must be 1 or 2");
  if (cfg->n_nets != 1 && cfg->n_nets != 2) return fail(CARO_E_INVAL, "n_nets must be 1 or 2");
  const int lpd = variant_lpd(var);
  if (cfg->max_batch < 1 || cfg->max_batch > MAXB || cfg->max_batch * lpd > 1024)
    return fail(CARO_E_INVAL, "max_batch out of range for this game (batch * lanes-per-descent <= 1024, batch <= 64)");
  if (cfg->game_kind == CARO_GAME_MNK && (cfg->k < 2 || cfg->k > cfg->n))
    return fail(CARO_E_INVAL, "k must satisfy 2 <= k <= n");
  if (cfg->stagger < 0) return fail(CARO_E_INVAL, "stagger must be >= 0");
  if (cfg->stagger > 0) {
    if (cfg->max_batch * lpd < 64 || (cfg->max_batch * lpd) % 64 != 0)
      return fail(CARO_E_INVAL, "staggered mode needs whole wavefronts per game (max_batch x lanes per descent a multiple of 64)");
    if (cfg->evict && cfg->max_batch * lpd == 64)
      return fail(CARO_E_INVAL, "staggered mode with eviction: the multi-wavefront kernel only (max_batch x lanes per descent > 64)");
  }
  int ndev = 0;
  if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0)
    return fail(CARO_E_NODEV, "no HIP device: libcaro_hip needs a GPU (there is no CPU fallback)");
  if (cfg->device_id < 0 || cfg->device_id >= ndev) return fail(CARO_E_INVAL, "device_id out of range");
  HIPCHK(hipSetDevice(cfg->device_id));

  caro_engine* h = new caro_engine();
  h->cfg = *cfg;
  h->var = var;
  h->select_pending = 0;
  h->prof_on = 0;
  h->prof_gate = 1;
  h->prof_ctr = 0;
  h->prof_cal = 0;
  h->rows = nullptr;
  h->rows_par = 0;
  {
    const char* e = getenv("CARO_NO_FUSED_TREE");
    h->fused_ok = !(e && e[0] == '1');
  }
  h->ev_used = 0;
  for (int i = 0; i < 8; ++i) { h->prof_ms[i] = 0; h->prof_n[i] = 0; }
  View& v = h->v;
  memset(&v, 0, sizeof v);
  v.gp = make_gp(cfg->game_kind, cfg->n, cfg->k);
  v.G = cfg->n_games;
  v.n_stores = cfg->n_stores;
  v.n_nets = cfg->n_nets;
  v.A = v.gp.A;
  v.HW = v.gp.rows * v.gp.cols;
  v.maxply = v.HW;
  v.maxd = v.HW;
  v.maxB = cfg->max_batch;
  v.ntab = (cfg->evict || cfg->stagger > 0) ? 2 : 1;  // second key table: eviction's target / the clean table a restarted slot moves to
  v.etab = cfg->evict ? 2 : 1;                          // second copy of the action rows: eviction only
  apply_run_params(h, cfg);
  v.cap = cfg->node_cap > 0 ? cfg->node_cap : 4096;
  // packing limits of the minibatch records: path_rec.x = node slot (24 bits) | action << 24 (8 bits), d_rec.x holds
  // the path length and the leaf rank in 8 bits each -- a table of more than 2^24 slots would alias nodes silently
  static_assert(MAXB <= 255, "leaf rank travels in 8 bits of d_rec.x");
  if (v.cap > (1 << 23)) { delete h; return fail(CARO_E_INVAL, "node_cap too large: at most 2^23 nodes per tree (slot ids travel in 24 bits)"); }
  int hc = 64;
  while (hc < 2 * v.cap) hc <<= 1;
  v.hcap = hc;
  if (v.A > 256 || v.maxd > 255) { delete h; return fail(CARO_E_INVAL, "board too large for the path records (actions < 256, depth <= 255)"); }
  {
    // experiment knobs (tools/exp/slot_alias.sh; NOTES round 5: neither changes the tree kernel's time), result-neutral
    const char* e = getenv("CARO_TREE_SKEW");
    const long skew = e ? strtol(e, nullptr, 0) : 0;
    if (skew < 0 || skew > 65536) { delete h; return fail(CARO_E_INVAL, "CARO_TREE_SKEW must be 0 .. 65536 slots"); }
    v.tstride = hc + (int)skew;
    e = getenv("CARO_SLOT_ROT");
    v.slot_rot = e ? (uint32_t)strtoul(e, nullptr, 0) : 0u;
  }
  const int KW = variant_kw(var), AP = variant_ap(var);
  const size_t T = (size_t)v.G * v.n_stores, G = (size_t)v.G;
  int rc = 0;
#define DA(p, n) if ((rc = dalloc(h, &p, (n))) != 0) { caro_engine_destroy(h); return rc; }
  DA(v.tbl, T);
  DA(v.node_key, T * v.ntab * v.tstride * KW);
  DA(v.edges, T * v.etab * v.tstride * 4 * AP);
  DA(v.n_nodes, T);
  DA(v.n_created, T);
  DA(v.root, G * KW);
  DA(v.player, G); DA(v.ply, G); DA(v.step, G); DA(v.uid, G); DA(v.done, G); DA(v.result, G); DA(v.final_r, G);
  DA(v.first, G);
  DA(v.h_key, G * v.maxply * KW);
  DA(v.h_player, G * v.maxply);
  DA(v.h_pi, G * v.maxply * v.A);
  DA(v.path_rec, G * v.maxB * v.maxd);
  DA(v.d_rec, G * v.maxB);
  DA(v.d_key, G * v.maxB * KW);
  DA(v.g_nleaf, G); DA(v.g_off, G); DA(v.g_tree, G); DA(v.g_class, G); DA(v.g_pack, G);
  DA(v.slot_list, 2 * G * v.maxB);
  DA(v.leaf_count, 4);
  DA(v.counters, G * C_N);
  DA(v.counters_sum, C_N);
  DA(v.dr_off, G); DA(v.dr_gidx, G); DA(v.dr_sel, G); DA(v.dr_tot, 2);
  DA(h->scratch, (size_t)v.maxd + 8);
  DA(h->live, 1);
  DA(h->rows, 8);
  if (cfg->stagger > 0) {
    DA(v.lm, G); DA(v.pend, G); DA(v.wait, G); DA(v.dirty, T);
    DA(v.pk_flag, G); DA(v.pk_ply, G); DA(v.pk_final_r, G); DA(v.pk_first, G); DA(v.pk_result, G); DA(v.pk_step, G);
    DA(v.pk_uid, G);
    DA(v.handed, 1);
    DA(v.ph_key, G * v.maxply * KW);
    DA(v.ph_player, G * v.maxply);
    DA(v.ph_pi, G * v.maxply * v.A);
  }
#undef DA
  HIPCHK(hipHostMalloc((void**)&h->pinned, 64, hipHostMallocDefault));
  HIPCHK(hipHostMalloc((void**)&h->pinned64, 128, hipHostMallocDefault));
  *out = h;
  rc = fresh_state(h, nullptr);
  if (rc) { caro_engine_destroy(h); *out = nullptr; return rc; }
  HIPCHK(hipDeviceSynchronize());
  return 0;
}

int caro_engine_restart(caro_engine* h, const caro_config* cfg, void* stream) {
  if (!h || !cfg) return fail(CARO_E_INVAL, "null argument");
  const caro_config& o = h->cfg;
  if (cfg->game_kind != o.game_kind || cfg->n != o.n || cfg->k != o.k || cfg->n_games != o.n_games ||
      cfg->n_stores != o.n_stores || cfg->n_nets != o.n_nets || cfg->max_batch != o.max_batch ||
      cfg->node_cap != o.node_cap || (cfg->evict != 0) != (o.evict != 0) || cfg->device_id != o.device_id ||
      (cfg->stagger > 0) != (o.stagger > 0))
    return fail(CARO_E_INVAL, "caro_engine_restart: the configuration differs from the engine's in a field that shapes its "
                              "memory (game, n_games, n_stores, n_nets, max_batch, node_cap, evict, device, staggered or not)");
  if (cfg->stagger < 0) return fail(CARO_E_INVAL, "stagger must be >= 0");
  if (h->drain_pending) return fail(CARO_E_STATE, "caro_engine_restart with a drain pending (caro_drain_tuples_end first)");
  if (h->select_pending) return fail(CARO_E_STATE, "caro_engine_restart with a pending caro_select");
  HIPCHK(hipSetDevice(o.device_id));
  h->cfg = *cfg;
  apply_run_params(h, cfg);
  return fresh_state(h, (hipStream_t)stream);
}

void caro_engine_destroy(caro_engine* h) {
  if (!h) return;
  for (void* p : h->allocs) (void)hipFree(p);
  for (hipEvent_t e : h->ev) (void)hipEventDestroy(e);
  if (h->drain_ev) (void)hipEventDestroy(h->drain_ev);
  if (h->pinned) (void)hipHostFree(h->pinned);
  if (h->pinned64) (void)hipHostFree(h->pinned64);
  delete h;
}

static int reset_games_impl(caro_engine* h, const int32_t* first_player_dev, void* stream) {
  DISPATCH(h->var, hipLaunchKernelGGL(k_reset<GEO>, dim3(h->v.G), dim3(256), 0, (hipStream_t)stream, h->v,
                                      first_player_dev));
  HIPCHK(hipGetLastError());
  h->select_pending = 0;
  return 0;
}

// A staggered engine carries per-game state the lock-step entry points know nothing about (lm, pend, wait, parked
// records, flipped / dirty key tables): every lock-step mutator refuses it instead of leaving that state stale.
int caro_reset_games(caro_engine* h, const int32_t* first_player_dev, void* stream) {
  if (!h) return fail(CARO_E_INVAL, "null engine");
  if (h->v.stag_S) return fail(CARO_E_STATE, "caro_reset_games: the engine runs in staggered mode (per-game clocks, pending minibatches, parked games); use caro_search_staggered / caro_drain_parked_begin");
  return reset_games_impl(h, first_player_dev, stream);
}

int caro_set_roots(caro_engine* h, const uint64_t* keys, const int32_t* players, void* stream) {
  if (!h || !keys || !players) return fail(CARO_E_INVAL, "null argument");
  if (h->v.stag_S) return fail(CARO_E_STATE, "caro_set_roots: the engine runs in staggered mode (per-game clocks, pending minibatches, parked games); use caro_search_staggered / caro_drain_parked_begin");
  DISPATCH(h->var, hipLaunchKernelGGL(k_set_roots<GEO>, dim3((h->v.G + 255) / 256), dim3(256), 0, (hipStream_t)stream,
                                      h->v, keys, players));
  HIPCHK(hipGetLastError());
  return 0;
}

int caro_select(caro_engine* h, int batch, int mb_index, const double* noise, float* planes, uint64_t* leaf_keys,
                void* stream) {
  if (!h || !planes) return fail(CARO_E_INVAL, "null argument");
  if (batch < 1 || batch > h->v.maxB) return fail(CARO_E_INVAL, "batch exceeds max_batch of the engine");
  if (h->v.stag_S) return fail(CARO_E_STATE, "caro_select: the engine runs in staggered mode (per-game clocks, pending minibatches, parked games); use caro_search_staggered / caro_drain_parked_begin");
  if (h->select_pending) return fail(CARO_E_STATE, "caro_select called twice without caro_expand_backup");
  const int lpd = variant_lpd(h->var);
  hipStream_t st = (hipStream_t)stream;
  const int p0 = prof_begin(h, PK_SELECT, st);
  DISPATCH(h->var, hipLaunchKernelGGL(k_select<GEO>, dim3(h->v.G), dim3(batch * lpd), mail_bytes<GEO>(batch), st, h->v,
                                      batch, mb_index, noise));
  prof_end(h, p0, st);
  const int p1 = prof_begin(h, PK_COMPACT, st);
  DISPATCH(h->var, hipLaunchKernelGGL(k_encode<GEO>, dim3(h->v.G), dim3(128), 0, st, h->v, batch, planes, leaf_keys));
  prof_end(h, p1, st);
  HIPCHK(hipGetLastError());
  h->select_pending = 1;
  return 0;
}

int caro_leaf_counts(caro_engine* h, int32_t counts[2], void* stream) {
  if (!h || !counts) return fail(CARO_E_INVAL, "null argument");
  HIPCHK(hipMemcpyAsync(h->pinned, h->v.leaf_count, 2 * sizeof(int32_t), hipMemcpyDeviceToHost, (hipStream_t)stream));
  HIPCHK(hipStreamSynchronize((hipStream_t)stream));
  counts[0] = h->pinned[0];
  counts[1] = h->pinned[1];
  return 0;
}

int caro_expand_backup(caro_engine* h, const float* probs, const float* values, void* stream) {
  if (!h) return fail(CARO_E_INVAL, "null engine");
  if (!h->select_pending) return fail(CARO_E_STATE, "caro_expand_backup without a pending caro_select");
  const int p0 = prof_begin(h, PK_EXPAND, (hipStream_t)stream);
  DISPATCH(h->var, hipLaunchKernelGGL(k_expand_backup<GEO>, dim3(h->v.G), dim3(64), 0, (hipStream_t)stream, h->v,
                                      probs, values));
  prof_end(h, p0, (hipStream_t)stream);
  HIPCHK(hipGetLastError());
  h->select_pending = 0;
  return 0;
}

// MCTS.search_batch (lib/mcts.py:162-176) for every game with the fused HIP net(s): `searches` minibatches of
// select -> net forward (leaf count read on device) -> expand+backup, enqueued back to back from C.
static int search_batch_impl(caro_engine* h, caro_net* net0, caro_net* net1, int searches, int batch, const double* noise,
                             float* planes, uint64_t* leaf_keys, float* probs, float* values, void* stream, int with_step,
                             const double* uniforms, int32_t* actions, int32_t* done, int32_t* result) {
  if (!h || !net0 || !planes || !probs || !values) return fail(CARO_E_INVAL, "null argument");
  if (h->v.n_nets == 2 && !net1) return fail(CARO_E_INVAL, "engine has two nets, net1 is null");
  if (searches < 1) return fail(CARO_E_INVAL, "searches must be >= 1");
  const int64_t max_rows = (int64_t)h->v.G * batch;
  const size_t noise_stride = (size_t)h->v.G * batch * h->v.A;
  if (batch < 1 || batch > h->v.maxB) return fail(CARO_E_INVAL, "batch exceeds max_batch of the engine");
  if (h->v.stag_S) return fail(CARO_E_STATE, "caro_search_batch: the engine runs in staggered mode (per-game clocks, pending minibatches, parked games); use caro_search_staggered / caro_drain_parked_begin");
  if (h->select_pending) return fail(CARO_E_STATE, "caro_search_batch with a pending caro_select");
  hipStream_t st = (hipStream_t)stream;
  // one 64-lane wavefront per game: the fused tree kernel (k_tree), two launches per minibatch; several whole
  // wavefronts per game: the same fusion as k_tree_mw
  const int bthreads = batch * variant_lpd(h->var);
  const bool fused1 = h->fused_ok && bthreads == 64;
  const bool fused = h->fused_ok && bthreads >= 64 && bthreads % 64 == 0;
  for (int mb = 0; mb < searches; ++mb) {
    // HIP-event timing is SAMPLED: an event pair per kernel costs ~8 % of the step (a pair's barrier packets expose the
    // dispatch latency that back-to-back launches hide).  Every 23rd minibatch of a counter that runs across moves: 23 is
    // coprime to the usual 25 / 20 / 50 / 100 searches per move, so every minibatch index (the first ones after a move
    // carry more leaves) is sampled equally often.  (Every 12th until round 6: measured, 1.2 % of the bench's value --
    // 8.49 against 8.59 M without events, same box, three alternations; every 23rd costs half of that and still gives
    // 21 timed launches of each kernel in the driver's 20 steps.)
    h->prof_gate = (h->prof_ctr++ % PROF_EVERY) == 0;
    int rc = 0;
    const int32_t* counts = h->v.leaf_count;
    if (fused) {
      int32_t* cur = h->rows + 4 * h->rows_par;
      int32_t* nxt = h->rows + 4 * (h->rows_par ^ 1);
      h->rows_par ^= 1;
      counts = cur;
      const int p1 = prof_begin(h, PK_SELECT, st);
      if (fused1) {
        DISPATCH(h->var, hipLaunchKernelGGL(k_tree<GEO>, dim3(h->v.G), dim3(128), 0, st, h->v, batch, mb,
                                            noise ? noise + (size_t)mb * noise_stride : nullptr, probs, values, planes,
                                            leaf_keys, cur, nxt, mb > 0 ? 1 : 0, 1));
      } else {
        DISPATCH(h->var, hipLaunchKernelGGL(k_tree_mw<GEO>, dim3(h->v.G), dim3(bthreads), mail_bytes<GEO>(batch), st,
                                            h->v, batch, mb, noise ? noise + (size_t)mb * noise_stride : nullptr, probs,
                                            values, planes, leaf_keys, cur, nxt, mb > 0 ? 1 : 0, 1, 0,
                                            (const double*)nullptr, (int32_t*)nullptr, (int32_t*)nullptr,
                                            (int32_t*)nullptr));
      }
      prof_end(h, p1, st);
      if (hipGetLastError() != hipSuccess) { h->prof_gate = 1; return fail(CARO_E_HIP, "k_tree launch failed"); }
    } else {
      rc = caro_select(h, batch, mb, noise ? noise + (size_t)mb * noise_stride : nullptr, planes, leaf_keys, stream);
      if (rc) { h->prof_gate = 1; return rc; }
    }
    const int p0 = prof_begin(h, PK_NET, st);
    if (fused)  // (the multi-wave kernel also lists its leaves' slot rows: one-board-per-workgroup net forms take them)
      rc = caro_net_forward_slot_list(net0, h->v.n_nets == 2 ? net1 : nullptr, planes, counts, h->v.g_pack,
                                      fused1 ? nullptr : h->v.slot_list, h->v.G, batch, probs, values, stream);
    else if (h->v.n_nets == 2)
      rc = caro_net_forward_pair_at(net0, net1, planes, counts, -1, max_rows, probs, values, stream);
    else rc = caro_net_forward(net0, planes, counts, 0, max_rows, probs, values, stream);
    prof_end(h, p0, st);
    prof_calibrate(h, st);
    if (!rc && !fused) rc = caro_expand_backup(h, probs, values, stream);
    if (rc) { h->prof_gate = 1; return rc; }
  }
  if (fused) {  // expand + backup of the last minibatch
    int32_t* cur = h->rows + 4 * h->rows_par;
    int32_t* nxt = h->rows + 4 * (h->rows_par ^ 1);
    h->rows_par ^= 1;
    const int p1 = prof_begin(h, PK_EXPAND, st);
    if (fused1) {
      DISPATCH(h->var, hipLaunchKernelGGL(k_tree<GEO>, dim3(h->v.G), dim3(128), 0, st, h->v, batch, searches,
                                          (const double*)nullptr, probs, values, planes, leaf_keys, cur, nxt, 1, 0));
    } else {
      // several wavefronts per game: the ply and the eviction of a caro_search_move ride in this closing launch
      DISPATCH(h->var, hipLaunchKernelGGL(k_tree_mw<GEO>, dim3(h->v.G), dim3(bthreads), 0, st, h->v, batch, searches,
                                          (const double*)nullptr, probs, values, planes, leaf_keys, cur, nxt, 1, 0,
                                          with_step ? 1 : 0, uniforms, actions, done, result));
      if (with_step) with_step = 0;  // done
    }
    prof_end(h, p1, st);
    if (hipGetLastError() != hipSuccess) { h->prof_gate = 1; return fail(CARO_E_HIP, "k_tree launch failed"); }
  }
  h->prof_gate = 1;
  return with_step ? caro_step(h, uniforms, actions, done, result, stream) : 0;
}

int caro_search_batch(caro_engine* h, caro_net* net0, caro_net* net1, int searches, int batch, const double* noise,
                      float* planes, uint64_t* leaf_keys, float* probs, float* values, void* stream) {
  return search_batch_impl(h, net0, net1, searches, batch, noise, planes, leaf_keys, probs, values, stream, 0, nullptr,
                           nullptr, nullptr, nullptr);
}

// search_batch + the ply: caro_search_batch followed by caro_step, with the ply (and the eviction) inside the search's
// closing launch where the geometry has the multi-wave fused kernel
int caro_search_move(caro_engine* h, caro_net* net0, caro_net* net1, int searches, int batch, const double* noise,
                     const double* uniforms, float* planes, uint64_t* leaf_keys, float* probs, float* values,
                     int32_t* actions, int32_t* done, int32_t* result, void* stream) {
  return search_batch_impl(h, net0, net1, searches, batch, noise, planes, leaf_keys, probs, values, stream, 1, uniforms,
                           actions, done, result);
}

// ---- staggered mode (see k_tree_stag)
int caro_search_staggered(caro_engine* h, caro_net* net0, caro_net* net1, int launches, int batch, float* planes,
                          uint64_t* leaf_keys, float* probs, float* values, void* stream) {
  if (!h || !net0 || !planes || !probs || !values) return fail(CARO_E_INVAL, "null argument");
  if (!h->v.stag_S) return fail(CARO_E_STATE, "the engine was not created in staggered mode (caro_config.stagger)");
  if (h->v.n_nets == 2 && !net1) return fail(CARO_E_INVAL, "engine has two nets, net1 is null");
  if (launches < 1) return fail(CARO_E_INVAL, "launches must be >= 1");
  const int bthreads = batch * variant_lpd(h->var);
  if (batch < 1 || batch > h->v.maxB || bthreads < 64 || bthreads % 64 != 0)
    return fail(CARO_E_INVAL, "staggered mode: batch x lanes per descent must be a multiple of 64");
  if (h->cfg.evict && bthreads == 64)
    return fail(CARO_E_INVAL, "staggered mode with eviction: batch x lanes per descent must be above 64");
  if (h->stag_batch && h->stag_batch != batch) return fail(CARO_E_INVAL, "staggered mode: the batch size is fixed by the first call");
  h->stag_batch = batch;
  hipStream_t st = (hipStream_t)stream;
  for (int j = 0; j < launches; ++j) {
    h->prof_gate = (h->prof_ctr++ % PROF_EVERY) == 0;  // sampled HIP-event timing, as caro_search_batch
    int32_t* cur = h->rows + 4 * h->rows_par;
    int32_t* nxt = h->rows + 4 * (h->rows_par ^ 1);
    h->rows_par ^= 1;
    const int p1 = prof_begin(h, PK_SELECT, st);
    if (bthreads == 64) {
      DISPATCH(h->var, hipLaunchKernelGGL(k_tree_stag<GEO>, dim3(h->v.G), dim3(128), 0, st, h->v, batch, probs, values,
                                          planes, leaf_keys, cur, nxt));
    } else {
      DISPATCH(h->var, hipLaunchKernelGGL(k_tree_stag_mw<GEO>, dim3(h->v.G), dim3(bthreads), mail_bytes<GEO>(batch), st,
                                          h->v, batch, probs, values, planes, leaf_keys, cur, nxt));
    }
    prof_end(h, p1, st);
    if (hipGetLastError() != hipSuccess) { h->prof_gate = 1; return fail(CARO_E_HIP, "k_tree_stag launch failed"); }
    const int p0 = prof_begin(h, PK_NET, st);
    const int rc = caro_net_forward_slot_list(net0, h->v.n_nets == 2 ? net1 : nullptr, planes, cur, h->v.g_pack,
                                              bthreads == 64 ? nullptr : h->v.slot_list, h->v.G, batch, probs, values,
                                              stream);
    prof_end(h, p0, st);
    prof_calibrate(h, st);
    if (rc) { h->prof_gate = 1; return rc; }
    // pool form: free slots are handed their next games every fifth launch too (not only at the drain that ends a pass):
    // a finished slot waits two or three launches, a tenth of a ply, instead of half a pass.  (The tables such a slot
    // leaves behind are cleaned at the next drain; no slot ends two games between two drains.)
    if (h->v.stag_pool && j % 5 == 4)
      DISPATCH(h->var, hipLaunchKernelGGL(k_stag_assign<GEO>, dim3(1), dim3(1024), 0, st, h->v));
  }
  h->prof_gate = 1;
  return 0;
}

// drain of the PARKED games: the lock-step drain kernels on a view whose game records and history rows are the
// parked copies (finished flag = pk_flag; a drained record becomes free again)
int caro_drain_parked_begin(caro_engine* h, int64_t cap, uint64_t* states, int32_t* players, double* pi, int32_t* z,
                            int64_t* games, void* stream) {
  if (!h || !states || !players || !pi || !z) return fail(CARO_E_INVAL, "null argument");
  if (!h->v.stag_S) return fail(CARO_E_STATE, "the engine was not created in staggered mode (caro_config.stagger)");
  if (h->drain_pending) return fail(CARO_E_STATE, "caro_drain_parked_begin twice without caro_drain_tuples_end");
  hipStream_t st = (hipStream_t)stream;
  if (!h->drain_ev) HIPCHK(hipEventCreateWithFlags(&h->drain_ev, hipEventDisableTiming));
  View pv = h->v;
  pv.done = h->v.pk_flag; pv.ply = h->v.pk_ply; pv.final_r = h->v.pk_final_r; pv.first = h->v.pk_first;
  pv.result = h->v.pk_result; pv.step = h->v.pk_step; pv.uid = h->v.pk_uid;
  pv.h_key = h->v.ph_key; pv.h_player = h->v.ph_player; pv.h_pi = h->v.ph_pi;
  hipLaunchKernelGGL(k_drain_scan, dim3(1), dim3(1024), 0, st, pv, (long long)cap);
  DISPATCH(h->var, hipLaunchKernelGGL(k_drain_copy<GEO>, dim3(pv.G), dim3(256), 0, st, pv, states, players, pi, z,
                                      games, 0));
  if (h->v.stag_pool)
    DISPATCH(h->var, hipLaunchKernelGGL(k_stag_assign<GEO>, dim3(1), dim3(1024), 0, st, h->v));
  DISPATCH(h->var, hipLaunchKernelGGL(k_stag_clean<GEO>, dim3(h->v.G * h->v.n_stores), dim3(256), 0, st, h->v));
  HIPCHK(hipGetLastError());
  HIPCHK(hipMemcpyAsync(h->pinned64 + 8, h->v.dr_tot, 2 * sizeof(int64_t), hipMemcpyDeviceToHost, st));
  HIPCHK(hipEventRecord(h->drain_ev, st));
  h->drain_pending = 1;
  return 0;
}

int caro_policy(caro_engine* h, double* pi, int32_t* counts, void* stream) {
  if (!h) return fail(CARO_E_INVAL, "null engine");
  DISPATCH(h->var, hipLaunchKernelGGL(k_policy<GEO>, dim3(h->v.G), dim3(64), 0, (hipStream_t)stream, h->v, pi, counts));
  HIPCHK(hipGetLastError());
  return 0;
}

int caro_step(caro_engine* h, const double* uniforms, int32_t* actions, int32_t* done, int32_t* result, void* stream) {
  if (!h) return fail(CARO_E_INVAL, "null engine");
  if (h->v.stag_S) return fail(CARO_E_STATE, "caro_step: the engine runs in staggered mode (per-game clocks, pending minibatches, parked games); use caro_search_staggered / caro_drain_parked_begin");
  if (h->select_pending) return fail(CARO_E_STATE, "caro_step with a pending caro_select");
  const int p0 = prof_begin(h, PK_STEP, (hipStream_t)stream);
  DISPATCH(h->var, hipLaunchKernelGGL(k_step<GEO>, dim3(h->v.G), dim3(64), 0, (hipStream_t)stream, h->v, uniforms,
                                      actions, done, result));
  prof_end(h, p0, (hipStream_t)stream);
  if (h->cfg.evict)  // drop the nodes the move made unreachable
    DISPATCH(h->var, hipLaunchKernelGGL(k_evict<GEO>, dim3(h->v.G), dim3(256), 0, (hipStream_t)stream, h->v));
  HIPCHK(hipGetLastError());
  return 0;
}

// The drain in two halves, so that a host loop never leaves the GPU idle: _begin enqueues the kernels and the copy
// of the two totals (nothing waits), the caller enqueues the next move's search behind it, and _end -- called while
// that search runs -- waits for the totals only.  The output buffers belong to the caller and must stay untouched
// until _end has returned and their rows have been consumed (stream order: anything enqueued before the next _begin).
int caro_drain_tuples_begin(caro_engine* h, int64_t cap, uint64_t* states, int32_t* players, double* pi, int32_t* z,
                            int64_t* games, int recycle, void* stream) {
  if (!h || !states || !players || !pi || !z) return fail(CARO_E_INVAL, "null argument");
  if (h->v.stag_S) return fail(CARO_E_STATE, "caro_drain_tuples_begin: the engine runs in staggered mode (per-game clocks, pending minibatches, parked games); use caro_search_staggered / caro_drain_parked_begin");
  if (h->select_pending) return fail(CARO_E_STATE, "caro_drain_tuples with a pending caro_select");
  if (h->drain_pending) return fail(CARO_E_STATE, "caro_drain_tuples_begin twice without caro_drain_tuples_end");
  hipStream_t st = (hipStream_t)stream;
  if (!h->drain_ev) HIPCHK(hipEventCreateWithFlags(&h->drain_ev, hipEventDisableTiming));
  hipLaunchKernelGGL(k_drain_scan, dim3(1), dim3(1024), 0, st, h->v, (long long)cap);
  DISPATCH(h->var, hipLaunchKernelGGL(k_drain_copy<GEO>, dim3(h->v.G), dim3(256), 0, st, h->v, states, players, pi, z,
                                      games, recycle));
  HIPCHK(hipGetLastError());
  HIPCHK(hipMemcpyAsync(h->pinned64 + 8, h->v.dr_tot, 2 * sizeof(int64_t), hipMemcpyDeviceToHost, st));
  HIPCHK(hipEventRecord(h->drain_ev, st));
  h->drain_pending = 1;
  return 0;
}

int caro_drain_tuples_end(caro_engine* h, int64_t* n_tuples, int64_t* n_games) {
  if (!h || !n_tuples || !n_games) return fail(CARO_E_INVAL, "null argument");
  if (!h->drain_pending) return fail(CARO_E_STATE, "caro_drain_tuples_end without caro_drain_tuples_begin");
  HIPCHK(hipEventSynchronize(h->drain_ev));
  h->drain_pending = 0;
  *n_tuples = h->pinned64[8];
  *n_games = h->pinned64[9];
  return 0;
}

int caro_drain_tuples(caro_engine* h, int64_t cap, uint64_t* states, int32_t* players, double* pi, int32_t* z,
                      int64_t* games, int recycle, int64_t* n_tuples, int64_t* n_games, void* stream) {
  if (!n_tuples || !n_games) return fail(CARO_E_INVAL, "null argument");
  const int rc = caro_drain_tuples_begin(h, cap, states, players, pi, z, games, recycle, stream);
  return rc ? rc : caro_drain_tuples_end(h, n_tuples, n_games);
}

int caro_counters(caro_engine* h, int64_t counters[8], void* stream) {
  if (!h || !counters) return fail(CARO_E_INVAL, "null argument");
  hipLaunchKernelGGL(k_sum_counters, dim3(1), dim3(256), 0, (hipStream_t)stream, h->v);
  HIPCHK(hipMemcpyAsync(h->pinned64, h->v.counters_sum, C_N * sizeof(int64_t), hipMemcpyDeviceToHost,
                        (hipStream_t)stream));
  HIPCHK(hipStreamSynchronize((hipStream_t)stream));
  for (int i = 0; i < 8; ++i) counters[i] = i < C_N ? h->pinned64[i] : 0;
  return 0;
}

int caro_leaf_counts_dev(caro_engine* h, const int32_t** counts_dev) {
  if (!h || !counts_dev) return fail(CARO_E_INVAL, "null argument");
  *counts_dev = h->v.leaf_count;
  return 0;
}

// A HIP stream restricted to CUs [part, part+1) / nparts of the device (hipExtStreamCreateWithCUMask), so that
// independent engines on different streams get disjoint compute units instead of sharing them.
int caro_stream_create_partition(int device_id, int part, int nparts, void** stream_out) {
  if (!stream_out || nparts < 1 || part < 0 || part >= nparts) return fail(CARO_E_INVAL, "bad partition");
  HIPCHK(hipSetDevice(device_id));
  hipDeviceProp_t prop;
  HIPCHK(hipGetDeviceProperties(&prop, device_id));
  const int ncu = prop.multiProcessorCount;
  const int words = (ncu + 31) / 32;
  std::vector<uint32_t> mask(words, 0u);
  const int lo = (int)((long long)ncu * part / nparts), hi = (int)((long long)ncu * (part + 1) / nparts);
  for (int c = lo; c < hi; ++c) mask[c >> 5] |= 1u << (c & 31);
  hipStream_t st = nullptr;
  HIPCHK(hipExtStreamCreateWithCUMask(&st, (uint32_t)words, mask.data()));
  *stream_out = (void*)st;
  return 0;
}
int caro_stream_destroy(void* stream) {
  if (stream) HIPCHK(hipStreamDestroy((hipStream_t)stream));
  return 0;
}

/* diagnostic: allocate (on != 0) or drop the per-game stamp buffer of k_select; read it back with caro_debug_read */
int caro_debug_stamps(caro_engine* h, int on) {
  if (!h) return fail(CARO_E_INVAL, "null engine");
  if (on && !h->v.dbg) {
    void* q = nullptr;
    HIPCHK(hipMalloc(&q, (size_t)h->v.G * 16 * sizeof(unsigned long long)));  // [G][8] block stamps, then [G][8] of expand_body
    HIPCHK(hipMemset(q, 0, (size_t)h->v.G * 16 * sizeof(unsigned long long)));
    h->allocs.push_back(q);
    h->v.dbg = (unsigned long long*)q;
  } else if (!on) {
    h->v.dbg = nullptr;  // the buffer stays owned by the engine
  }
  return 0;
}
int caro_debug_read(caro_engine* h, uint64_t* out_host, int64_t n_u64, void* stream) {
  if (!h || !out_host || !h->v.dbg) return fail(CARO_E_INVAL, "no stamp buffer");
  if (n_u64 > (int64_t)h->v.G * 16) n_u64 = (int64_t)h->v.G * 16;
  HIPCHK(hipStreamSynchronize((hipStream_t)stream));
  HIPCHK(hipMemcpy(out_host, h->v.dbg, n_u64 * sizeof(uint64_t), hipMemcpyDeviceToHost));
  return 0;
}

/* diagnostic: the tree kernels' square root of a visit count against sqrtf on every integer 0..n_max (<= 2^24); the
   number of differing results goes to *bad_host (0 is the only acceptable answer) */
int caro_debug_sqrt_check(uint32_t n_max, uint64_t* bad_host) {
  if (!bad_host || n_max > (1u << 24)) return fail(CARO_E_INVAL, "bad argument");
  unsigned long long* bad = nullptr;
  HIPCHK(hipMalloc(&bad, sizeof(unsigned long long)));
  HIPCHK(hipMemset(bad, 0, sizeof(unsigned long long)));
  hipLaunchKernelGGL(k_sqrt_check, dim3(1024), dim3(256), 0, (hipStream_t) nullptr, n_max, bad);
  HIPCHK(hipGetLastError());
  HIPCHK(hipMemcpy(bad_host, bad, sizeof(unsigned long long), hipMemcpyDeviceToHost));
  HIPCHK(hipFree(bad));
  return 0;
}

int caro_select_cancel(caro_engine* h) {
  if (!h) return fail(CARO_E_INVAL, "null engine");
  h->select_pending = 0;
  return 0;
}

int caro_get_descent(caro_engine* h, int game, int b, int32_t* info_dev, float* value_dev, uint64_t* leaf_key_dev,
                     uint64_t* path_keys_dev, int32_t* path_actions_dev, void* stream) {
  if (!h || !info_dev || !value_dev || !leaf_key_dev || !path_keys_dev || !path_actions_dev)
    return fail(CARO_E_INVAL, "null argument");
  if (!h->select_pending) return fail(CARO_E_STATE, "caro_get_descent needs a pending caro_select");
  if (game < 0 || game >= h->v.G || b < 0 || b >= h->v.maxB) return fail(CARO_E_INVAL, "bad descent index");
  DISPATCH(h->var, hipLaunchKernelGGL(k_get_descent<GEO>, dim3(1), dim3(64), 0, (hipStream_t)stream, h->v, game, b,
                                      info_dev, value_dev, leaf_key_dev, path_keys_dev, path_actions_dev));
  HIPCHK(hipGetLastError());
  return 0;
}

int caro_profile_enable(caro_engine* h, int on) {
  if (!h) return fail(CARO_E_INVAL, "null engine");
  if (!on && h->prof_on) prof_flush(h);
  h->prof_on = on ? 1 : 0;
  return 0;
}

int caro_profile_begin(caro_engine* h, int kind, void* stream) {
  if (!h || kind < 0 || kind >= PK_N) return -1;
  return prof_begin(h, kind, (hipStream_t)stream);
}
void caro_profile_end(caro_engine* h, int slot, void* stream) {
  if (h) prof_end(h, slot, (hipStream_t)stream);
}

int caro_profile_read(caro_engine* h, double ms[8], int64_t launches[8], int reset) {
  if (!h || !ms || !launches) return fail(CARO_E_INVAL, "null argument");
  prof_flush(h);
  for (int i = 0; i < 8; ++i) {
    ms[i] = h->prof_ms[i];
    launches[i] = h->prof_n[i];
    if (reset) { h->prof_ms[i] = 0; h->prof_n[i] = 0; }
  }
  return 0;
}

int caro_live_games(caro_engine* h, int32_t* live, void* stream) {
  if (!h || !live) return fail(CARO_E_INVAL, "null argument");
  hipStream_t st = (hipStream_t)stream;
  HIPCHK(hipMemsetAsync(h->live, 0, sizeof(int32_t), st));
  hipLaunchKernelGGL(k_count_live, dim3(1), dim3(256), 0, st, h->v, h->live);
  HIPCHK(hipMemcpyAsync(h->pinned + 4, h->live, sizeof(int32_t), hipMemcpyDeviceToHost, st));
  HIPCHK(hipStreamSynchronize(st));
  *live = h->pinned[4];
  return 0;
}

int caro_pending_leaves(caro_engine* h, int32_t* pending, void* stream) {
  if (!h || !pending) return fail(CARO_E_INVAL, "null argument");
  hipStream_t st = (hipStream_t)stream;
  HIPCHK(hipMemsetAsync(h->live, 0, sizeof(int32_t), st));
  hipLaunchKernelGGL(k_count_pending, dim3(1), dim3(256), 0, st, h->v, h->select_pending, h->live);
  HIPCHK(hipMemcpyAsync(h->pinned + 4, h->live, sizeof(int32_t), hipMemcpyDeviceToHost, st));
  HIPCHK(hipStreamSynchronize(st));
  *pending = h->pinned[4];
  return 0;
}

int caro_tree_sizes(caro_engine* h, int32_t* out, void* stream) {
  if (!h || !out) return fail(CARO_E_INVAL, "null argument");
  const int T = h->v.G * h->v.n_stores;
  hipLaunchKernelGGL(k_tree_sizes, dim3((T + 255) / 256), dim3(256), 0, (hipStream_t)stream, h->v, out);
  HIPCHK(hipGetLastError());
  return 0;
}

int caro_tree_live(caro_engine* h, int32_t* out, void* stream) {
  if (!h || !out) return fail(CARO_E_INVAL, "null argument");
  const int T = h->v.G * h->v.n_stores;
  hipLaunchKernelGGL(k_tree_live, dim3((T + 255) / 256), dim3(256), 0, (hipStream_t)stream, h->v, out);
  HIPCHK(hipGetLastError());
  return 0;
}

int caro_lookup_nodes(caro_engine* h, int64_t M, const int32_t* game, const int32_t* store, const uint64_t* keys,
                      int32_t* found, int32_t* N, float* W, float* Q, float* P, int32_t* strong, void* stream) {
  if (!h) return fail(CARO_E_INVAL, "null engine");
  if (M <= 0) return 0;
  DISPATCH(h->var, hipLaunchKernelGGL(k_lookup<GEO>, dim3((unsigned)M), dim3(64), 0, (hipStream_t)stream, h->v,
                                      (long long)M, game, store, keys, found, N, W, Q, P, strong));
  HIPCHK(hipGetLastError());
  return 0;
}

int caro_get_roots(caro_engine* h, uint64_t* keys, int32_t* players, int32_t* ply, uint64_t* uid, void* stream) {
  if (!h) return fail(CARO_E_INVAL, "null engine");
  DISPATCH(h->var, hipLaunchKernelGGL(k_get_roots<GEO>, dim3((h->v.G + 255) / 256), dim3(256), 0, (hipStream_t)stream,
                                      h->v, keys, players, ply, uid));
  HIPCHK(hipGetLastError());
  return 0;
}

int caro_poke_nodes(caro_engine* h, int64_t M, const int32_t* game, const int32_t* store, const uint64_t* keys,
                    const int32_t* N, const float* W, const float* Q, const float* P, const int32_t* strong,
                    void* stream) {
  if (!h) return fail(CARO_E_INVAL, "null engine");
  if (M <= 0) return 0;
  DISPATCH(h->var, hipLaunchKernelGGL(k_poke<GEO>, dim3(1), dim3(64), 0, (hipStream_t)stream, h->v, (long long)M, game,
                                      store, keys, N, W, Q, P, strong));
  HIPCHK(hipGetLastError());
  return 0;
}

int caro_backup_path(caro_engine* h, int game, int store, float value, int value_is_f32, int len,
                     const uint64_t* keys, const int32_t* actions, void* stream) {
  if (!h) return fail(CARO_E_INVAL, "null engine");
  if (len < 0 || len > h->v.maxd) return fail(CARO_E_INVAL, "path too long");
  if (game < 0 || game >= h->v.G || store < 0 || store >= h->v.n_stores) return fail(CARO_E_INVAL, "bad tree index");
  if (len == 0) return 0;
  DISPATCH(h->var, hipLaunchKernelGGL(k_backup_one<GEO>, dim3(1), dim3(64), 0, (hipStream_t)stream, h->v, game, store,
                                      value, value_is_f32, len, keys, actions, h->scratch));
  HIPCHK(hipGetLastError());
  return 0;
}

int caro_dump_tree(caro_engine* h, int game, int store, int64_t cap, uint64_t* keys, int32_t* N, float* W, float* Q,
                   float* P, int32_t* strong, int64_t* n_nodes, void* stream) {
  if (!h || !n_nodes) return fail(CARO_E_INVAL, "null argument");
  if (game < 0 || game >= h->v.G || store < 0 || store >= h->v.n_stores) return fail(CARO_E_INVAL, "bad tree index");
  hipStream_t st = (hipStream_t)stream;
  const int t = game * h->v.n_stores + store;
  HIPCHK(hipMemcpyAsync(h->pinned + 5, h->v.n_nodes + t, sizeof(int32_t), hipMemcpyDeviceToHost, st));
  HIPCHK(hipStreamSynchronize(st));
  const int nn = h->pinned[5];
  *n_nodes = nn;
  if (nn == 0 || cap <= 0 || !keys) return 0;
  HIPCHK(hipMemsetAsync(h->live, 0, sizeof(int32_t), st));
  DISPATCH(h->var, hipLaunchKernelGGL(k_dump<GEO>, dim3((unsigned)h->v.hcap), dim3(64), 0, st, h->v, game, store,
                                      (long long)cap, keys, N, W, Q, P, strong, h->live));
  HIPCHK(hipGetLastError());
  HIPCHK(hipStreamSynchronize(st));
  return 0;
}

}  // extern "C"
