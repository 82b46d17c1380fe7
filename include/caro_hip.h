/*
 * caro_hip.h -- C-ABI of libcaro_hip.so, the MI355X (gfx950) self-play engine.
 *
 * The reference (nh273/caro-ai) is pure Python: the path this library replaces
 * sits behind a Python plugin API, not an FFI.  Each entry point below names
 * the reference interface it stands in for (paths relative to the reference
 * tree).  INTEGRATION.md shows the ctypes binding a maintainer of the
 * reference would add (it is what caro_ai_amd/_lib.py does).
 *
 * Conventions
 *   - plain C types only; every `*_dev` pointer is DEVICE memory owned by the
 *     caller (e.g. a torch tensor's data_ptr()) on the engine's device;
 *     `stream` is a hipStream_t passed as void* (NULL = default stream).
 *     Calls enqueue work on `stream` and return without synchronising unless
 *     the comment says otherwise.
 *   - return value: 0 on success, negative CARO_E_* code on failure;
 *     caro_last_error() returns a message for the calling thread.
 *   - no internal threads; one engine per (process, GPU).
 *
 * Board states ("keys"), KW = caro_key_words() 64-bit words per board:
 *   connect four: KW = 1, the reference's own 63-bit state int
 *                 (lib/game/connect_four/connect_four.py:36-56).
 *   m,n,k       : KW = 2*W64, W64 = 1 (n<=8), 2 (n<=11), 4 (n<=15);
 *                 words [0,W64) = bit-plane of token 0, [W64,2*W64) = token 1,
 *                 bit i = square i of lib/game/tictactoe/tictactoe.py:14-24.
 */
#ifndef CARO_HIP_H
#define CARO_HIP_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define CARO_GAME_CONNECT4 0
#define CARO_GAME_MNK 1

#define CARO_E_INVAL (-22)   /* bad argument */
#define CARO_E_NOMEM (-12)   /* allocation failed */
#define CARO_E_HIP (-5)      /* HIP runtime error */
#define CARO_E_NODEV (-19)   /* no usable GPU */
#define CARO_E_STATE (-71)   /* call sequence violated */

typedef struct caro_engine caro_engine;

typedef struct caro_config {
  int32_t game_kind;          /* CARO_GAME_* */
  int32_t n, k;               /* m,n,k board side and run length (TicTacToe(n, k_to_win), tictactoe.py:27) */
  int32_t n_games;            /* G concurrent games on this GPU */
  int32_t n_stores;           /* 1: one tree per game shared by both players (utils.py:60-61);
                                 2: one tree per player (utils.py:58-59, play.py:47) */
  int32_t n_nets;             /* 1: self-play; 2: player p's leaves go to net p (utils.py:64,77-79) */
  int32_t max_batch;          /* largest mcts_batch_size that will be used */
  int32_t node_cap;           /* nodes per tree; 0 = searches_hint * max_batch * max plies bound */
  int32_t steps_before_tau_0; /* utils.py:70,97-99 */
  int32_t first_player_mode;  /* 0: all games start with player 0; 1: player 1; 2: game uid & 1 */
  float c_puct;               /* config.C_PUCT */
  double alpha, explore;      /* config.ALPHA, config.EXPLORE */
  uint64_t seed;              /* key of the generated noise / move uniforms (caro_noise.h) */
  uint64_t uid_base;          /* uid of this engine's game 0 (rank offset in multi-GPU runs) */
  uint64_t uid_stride;        /* uid += uid_stride each time a game slot is recycled (total games in flight) */
  int32_t device_id;
  int32_t evict;              /* 1: after every move drop the nodes that can no longer be reached (boards that do
                                 not contain the new root).  Result-neutral; node_cap then bounds the LIVE nodes. */
  int32_t stagger;            /* > 0: staggered mode with this many minibatches (mcts_searches) per move -- every game
                                 on its own minibatch clock, see caro_search_staggered; 0: lock-step */
  int32_t stagger_recycle;    /* staggered mode, 1: a finished game's slot restarts in-kernel (uid += uid_stride).
                                 2 (with games_limit > 0), the POOL form: a finished slot waits for the next
                                 caro_drain_parked_begin, which hands the free slots the next games of the wanted set that
                                 have not been started yet, in slot order (local index i = uid uid_base + i % n_games +
                                 (i / n_games) * uid_stride: the same set of games) -- the slots stay busy until the
                                 wanted games run out, whatever the lengths of the games a slot happened to get */
  int64_t games_limit;        /* > 0: the engine plays exactly the games with local index k * n_games + g < games_limit
                                 (slot g, its k-th game; uid = uid_base + g + k * uid_stride): a slot whose next game
                                 would lie beyond that stays finished instead of restarting, in either schedule, and
                                 slots g >= games_limit never start.  This is train.py:41-47's `for _ in
                                 range(PLAY_EPISODES)` as a property of the engine: no game outside the wanted set is
                                 ever started, so counters and tuples belong to the wanted games only.  0: no limit */
} caro_config;

const char* caro_last_error(void);
int caro_version(void);

/* ---- geometry of a game kind (host only, no GPU needed) ---- */
int caro_key_words(int game_kind, int n);                    /* KW */
int caro_action_space(int game_kind, int n);                 /* BaseGame.action_space */
int caro_obs_cells(int game_kind, int n);                    /* H*W of BaseGame.obs_shape */

/* ---- host-side single-state rule helpers (API-edge use by the BaseGame shim;
 *      compiled from the same caro_rules.h as the kernels; no GPU needed) ---- */
/* BaseGame.initial_state (connect_four.py:67-74, tictactoe.py:56-63) */
int caro_host_initial(int game_kind, int n, int k, uint64_t* key);
/* BaseGame.move (connect_four.py:241-265, tictactoe.py:210-235): key updated in place, *won set */
int caro_host_move(int game_kind, int n, int k, uint64_t* key, int move, int player, int* won);
/* BaseGame.possible_moves as a byte mask legal[A] (connect_four.py:157-165, tictactoe.py:137-150) */
int caro_host_legal(int game_kind, int n, int k, const uint64_t* key, uint8_t* legal);
/* BaseGame.states_to_training_batch for one state -> float32[2*H*W] */
int caro_host_encode(int game_kind, int n, int k, const uint64_t* key, int who_move, float* planes);
/* one Dirichlet row / one move uniform of the caro_noise.h spec */
int caro_host_noise_row(uint64_t seed, uint64_t uid, uint32_t ply, uint32_t sim, int A, double alpha, double* out);
double caro_host_move_uniform(uint64_t seed, uint64_t uid, uint32_t ply);

/* ---- batched rule kernels (device) : lib/game rules over M independent boards ---- */
/* keys_dev u64[M,KW] in/out, moves_dev i32[M], players_dev i32[M] -> won_dev i32[M], full_dev i32[M] */
int caro_rules_move_batch(int game_kind, int n, int k, int64_t M, uint64_t* keys_dev, const int32_t* moves_dev,
                          const int32_t* players_dev, int32_t* won_dev, int32_t* full_dev, void* stream);
/* legal_dev u8[M,A] */
int caro_rules_legal_batch(int game_kind, int n, int k, int64_t M, const uint64_t* keys_dev, uint8_t* legal_dev,
                           void* stream);
/* planes_dev f32[M,2,H,W] */
int caro_rules_encode_batch(int game_kind, int n, int k, int64_t M, const uint64_t* keys_dev,
                            const int32_t* who_dev, float* planes_dev, void* stream);
/* device form of the noise spec: out_dev f64[M,A] rows keyed (seed, uid[m], ply[m], sim[m]) */
int caro_noise_batch(uint64_t seed, int64_t M, int A, double alpha, const uint64_t* uid_dev, const uint32_t* ply_dev,
                     const uint32_t* sim_dev, double* out_dev, void* stream);

/* ---- engine: G concurrent games = G x play_game (lib/utils.py:25-108) ---- */
/* replaces MCTS.__init__ (lib/mcts.py:27-37) for every tree of every game */
int caro_engine_create(const caro_config* cfg, caro_engine** out);
void caro_engine_destroy(caro_engine* h);
/* A NEW RUN on an existing engine, in place of destroy + create (train.py:185-193 builds its store once and plays
 * every self-play call on it; here the gigabytes of tree tables are kept and only cleared): every game restarts from
 * the initial position with empty trees, zero counters, fresh minibatch clocks and no parked games -- the state
 * caro_engine_create leaves behind, so a restarted engine plays bit for bit what a fresh engine of the same
 * configuration plays.  `cfg` must agree with the engine in everything that shapes its memory (game_kind, n, k,
 * n_games, n_stores, n_nets, max_batch, node_cap, evict, device_id, staggered or not); taken afresh from it are
 * seed, uid_base, uid_stride, games_limit, steps_before_tau_0, first_player_mode, c_puct, alpha, explore, stagger
 * (the number of minibatches per move) and stagger_recycle.  Works on lock-step and staggered engines; refuses
 * (CARO_E_STATE) while a drain or a select is pending.  Enqueues on `stream`, does not synchronise. */
int caro_engine_restart(caro_engine* h, const caro_config* cfg, void* stream);
/* (re)start every game from the initial position with an empty tree: utils.py:58-73 / MCTS.clear (mcts.py:39-43).
 * first_player_dev: i32[G] or NULL (use first_player_mode). */
int caro_reset_games(caro_engine* h, const int32_t* first_player_dev, void* stream);
/* force game positions (tests, MCTS shim): root keys u64[G,KW], players i32[G]; trees are kept */
int caro_set_roots(caro_engine* h, const uint64_t* keys_dev, const int32_t* players_dev, void* stream);

/* One search_minibatch (lib/mcts.py:248-287), first half, for every live game:
 * `batch` find_leaf descents per game on the frozen tree (mcts.py:97-148: root
 * noise :48-62, PUCT :64-84, mask :86-95, first-max argmax :136, game.move :138,
 * terminal values :140-146), de-duplication of new leaves (:272-278), and the
 * NN planes of the unique leaves (game.states_to_training_batch) written as
 * dense rows into planes_dev f32[>= G*batch, 2, H, W]: rows [0,L0) feed net 0,
 * rows [L0, L0+L1) feed net 1.
 * noise_dev: f64[G, batch, A] explicit Dirichlet rows for this minibatch, or
 * NULL to generate them on device from (seed, uid, ply, sim = mb_index*batch + b).
 * leaf_keys_dev (optional, may be NULL): u64[>= G*batch, KW] keys of the rows. */
int caro_select(caro_engine* h, int batch, int mb_index, const double* noise_dev, float* planes_dev,
                uint64_t* leaf_keys_dev, void* stream);
/* MCTS.find_leaf (lib/mcts.py:97-148) of descent `b` of game `game` of the pending select:
 * info_dev i32[4] = (status 0 dropped duplicate / 1 terminal / 2 new leaf, path length, player at the leaf, -),
 * value_dev f32[1] (terminal value), leaf_key_dev u64[KW], path_keys_dev u64[maxd,KW] (the `states` list),
 * path_actions_dev i32[maxd] (the `actions` list); maxd = H*W. */
int caro_get_descent(caro_engine* h, int game, int b, int32_t* info_dev, float* value_dev, uint64_t* leaf_key_dev,
                     uint64_t* path_keys_dev, int32_t* path_actions_dev, void* stream);
/* drop a pending select without expanding (find_leaf alone does not modify the tree) */
int caro_select_cancel(caro_engine* h);
/* Blocks until the select on `stream` has finished; counts[0..1] = L0, L1. */
int caro_leaf_counts(caro_engine* h, int32_t counts[2], void* stream);
/* Device address of the two leaf counts {L0, L1} (i32[2], engine-owned), so that a consumer kernel
 * (caro_net_forward) can read them without a host round trip. */
int caro_leaf_counts_dev(caro_engine* h, const int32_t** counts_dev);
/* Second half (mcts.py:281-287): _create_node (:178-190) for every unique leaf
 * with prior row probs_dev f32[L, A] (softmax ALREADY applied, mcts.py:216) and
 * _backup (:225-246) of terminals (sim order) then new leaves (first-seen
 * order) with values_dev f32[L] (mcts.py:217). Rows are those of caro_select.
 * BUFFER SIZES: probs_dev and values_dev must be ALLOCATED for n_games * max_batch rows (the size of the planes buffer
 * handed to caro_select), whatever L is: the kernel requests a game's value rows before it knows the game's leaf
 * count (one memory latency less per minibatch), so it reads up to max_batch - 1 rows past the last leaf row --
 * never past row n_games * max_batch.  Rows >= L are read and ignored; they need not be initialised. */
int caro_expand_backup(caro_engine* h, const float* probs_dev, const float* values_dev, void* stream);

/* get_policy_value (lib/mcts.py:289-313) of every game's root with the tau the
 * game is in: pi_dev f64[G, A]; counts_dev i32[G, A] (root N) optional. */
int caro_policy(caro_engine* h, double* pi_dev, int32_t* counts_dev, void* stream);
/* One ply of play_game for every live game (utils.py:80-99): pi, history row,
 * np.random.choice via inverse CDF of a uniform (uniforms_dev f64[G], or NULL =
 * generated), game.move, win / draw detection, tau switch.
 * Optional outputs (may be NULL): actions_dev i32[G] (-1 for finished games),
 * done_dev i32[G] (1 once the game is over), result_dev i32[G] (net1_result). */
int caro_step(caro_engine* h, const double* uniforms_dev, int32_t* actions_dev, int32_t* done_dev,
              int32_t* result_dev, void* stream);
/* Replay emission (utils.py:101-106) for finished games, in game order, each
 * game's plies last-to-first exactly as the reference appends them:
 *   states_dev u64[cap,KW], players_dev i32[cap], pi_dev f64[cap,A], z_dev i32[cap]
 * and one record per drained game: games_dev i64[G,4] = (uid, first_player, net1_result, steps).
 * Drained slots restart as new games (uid += uid_stride) when `recycle` != 0,
 * otherwise they stay finished.  Synchronises; *n_tuples / *n_games set on return. */
int caro_drain_tuples(caro_engine* h, int64_t cap, uint64_t* states_dev, int32_t* players_dev, double* pi_dev,
                      int32_t* z_dev, int64_t* games_dev, int recycle, int64_t* n_tuples, int64_t* n_games,
                      void* stream);
/* The same drain in two halves for host loops that must not leave the GPU idle (the reference's loop appends to
 * its deque right away, utils.py:101-106; here the rows of move k are handed over while move k+1 is searched):
 * _begin enqueues the kernels and returns at once; _end waits for the two totals only.  The output buffers must
 * not be touched between the two calls, and their rows are valid for anything enqueued before the next _begin. */
int caro_drain_tuples_begin(caro_engine* h, int64_t cap, uint64_t* states_dev, int32_t* players_dev, double* pi_dev,
                            int32_t* z_dev, int64_t* games_dev, int recycle, void* stream);
int caro_drain_tuples_end(caro_engine* h, int64_t* n_tuples, int64_t* n_games);

/* counters[8] (host array): sims, levels, expansions, terminals, dropped
 * duplicates, overflows, plies, finished games.  Synchronises.
 * `overflows` counts every event after which the engine's games may no longer be the reference's: a minibatch whose
 * new nodes did not fit node_cap (its leaves are dropped), and a ply REFUSED because the root had no visits (one
 * search on an unexpanded root: lib/mcts.py:311 divides by zero there; the game is left where it was).  Every caller
 * in this package treats a non-zero value as an error. */
int caro_counters(caro_engine* h, int64_t counters[8], void* stream);
/* HIP-event timing of the path's kernels on the stream they are launched on (bench.py's live
 * roofline).  Kinds: 0 select, 1 scan+encode, 2 expand+backup, 3 step, 4 net forward (bracketed by the
 * caller with caro_profile_begin/_end around caro_net_forward); 5 / 6 calibrate the pairs themselves: a pair
 * around ONE launch of an empty kernel (E1) and a pair around TWO (E2), recorded behind every fourth sampled net
 * launch of caro_search_batch / caro_search_staggered -- a pair adds o = 2 E1 - E2 (+ the sub-microsecond gap between
 * two dependent launches) to the kernel it brackets, to be subtracted from the other kinds' averages; 7 free.
 * caro_profile_read synchronises on the recorded events; ms[] / launches[] are running totals. */
int caro_profile_enable(caro_engine* h, int on);
int caro_profile_begin(caro_engine* h, int kind, void* stream); /* returns a slot, or -1 when profiling is off */
void caro_profile_end(caro_engine* h, int slot, void* stream);
int caro_profile_read(caro_engine* h, double ms[8], int64_t launches[8], int reset);
/* diagnostics (tools/probe_select.py): per-game cycle stamps of k_select's phases; off unless enabled */
int caro_debug_stamps(caro_engine* h, int on);
int caro_debug_read(caro_engine* h, uint64_t* out_host, int64_t n_u64, void* stream);
/* diagnostic: the device's square root of a visit count (float32, `m.sqrt(sum(counts))` of lib/mcts.py:79 rounded as
   numpy does) compared with sqrtf on every integer 0..n_max (n_max <= 2^24); *bad_host = differing results (must be 0) */
int caro_debug_sqrt_check(uint32_t n_max, uint64_t* bad_host);
/* number of live (unfinished) games; synchronises */
int caro_live_games(caro_engine* h, int32_t* live, void* stream);
/* unique leaves that have been selected but not yet booked as expansions (between caro_select and
 * caro_expand_backup; in staggered mode the pending minibatch of every game).  At any point of a run
 * sims == expansions + terminals + dropped + pending (+ the leaves of overflowed minibatches); synchronises */
int caro_pending_leaves(caro_engine* h, int32_t* pending, void* stream);

/* ---- inspection (tests, MCTS shim: the four public dicts of lib/mcts.py:29-36) ---- */
/* len(MCTS) per tree: out_dev i32[G*n_stores] */
int caro_tree_sizes(caro_engine* h, int32_t* out_dev, void* stream);
/* nodes a tree HOLDS right now (= len(MCTS) without eviction; with caro_config.evict what survived the last
 * k_evict plus what the current move added -- the figure node_cap bounds): out_dev i32[G*n_stores] */
int caro_tree_live(caro_engine* h, int32_t* out_dev, void* stream);
/* look up M (game, store, key) triples: found_dev i32[M]; N i32[M,A]; W,Q,P f32[M,A]; strong i32[M,A]
 * (strong = W has absorbed a float32 net value; 0 = still an exact Python float) */
int caro_lookup_nodes(caro_engine* h, int64_t M, const int32_t* game_dev, const int32_t* store_dev,
                      const uint64_t* keys_dev, int32_t* found_dev, int32_t* N_dev, float* W_dev, float* Q_dev,
                      float* P_dev, int32_t* strong_dev, void* stream);
/* current root key / player / ply / uid of every game: keys u64[G,KW], players i32[G], ply i32[G], uid u64[G] */
int caro_get_roots(caro_engine* h, uint64_t* keys_dev, int32_t* players_dev, int32_t* ply_dev, uint64_t* uid_dev,
                   void* stream);
/* insert / overwrite nodes (MCTS shim attribute setters, lib/test_mcts.py:15-21): same layout as lookup */
int caro_poke_nodes(caro_engine* h, int64_t M, const int32_t* game_dev, const int32_t* store_dev,
                    const uint64_t* keys_dev, const int32_t* N_dev, const float* W_dev, const float* Q_dev,
                    const float* P_dev, const int32_t* strong_dev, void* stream);
/* MCTS._backup (lib/mcts.py:225-246) of ONE path on one tree: keys u64[len,KW], actions i32[len] */
int caro_backup_path(caro_engine* h, int game, int store, float value, int value_is_f32, int len,
                     const uint64_t* keys_dev, const int32_t* actions_dev, void* stream);
/* dump a whole tree (MCTS shim dict views): keys u64[cap,KW], N i32[cap,A], W/Q/P f32[cap,A], strong i32[cap,A];
 * *n_nodes set on return (synchronises). */
int caro_dump_tree(caro_engine* h, int game, int store, int64_t cap, uint64_t* keys_dev, int32_t* N_dev,
                   float* W_dev, float* Q_dev, float* P_dev, int32_t* strong_dev, int64_t* n_nodes, void* stream);

/* ---- fused float32 policy/value net (lib/model.py:10-94, Net.forward in eval mode + F.softmax of
 *      lib/mcts.py:216) for the leaf batch.  Weights: one flat float32 host buffer in the order
 *      conv_in [9 taps][2][64], b[64]; 5 x 9 tap chunks of 4096 floats in the kernel's LDS image order
 *      (caro_ai_amd/net_hip.py packs them, batch-norm folded), b[5][64]; heads [3][64], b[3];
 *      value.0 [20][HW], b[20]; value.2 [20], b[1]; policy.0 [A][2HW], b[A]. ---- */
typedef struct caro_net caro_net;
int64_t caro_net_packed_size(int H, int W, int A);
int caro_net_create(int H, int W, int A, float negative_slope, const float* packed_host, int64_t n_floats,
                    int device_id, caro_net** out);
/* f32w mode: the 3x3 convolutions in row-Winograd F(2,3) form (float32 MFMA, two thirds of the multiplies;
 * results differ from the direct form by float32 rounding only).  ww_host = [5][4][3][4096] floats from
 * caro_ai_amd/net_hip.py:pack_net_w ([layer][transformed tap p][dx], each in the order of the plain tap chunks); the
 * library re-orders them into the chunks its kernel streams ([layer][dx][granule half][p]) at upload.
 * Lowers caro_net_boards_per_workgroup if 128 / (ceil(H/2)*W) is smaller. */
int caro_net_enable_winograd(caro_net* n, const float* ww_host, int64_t n_floats);
/* f32w2 mode for LARGE boards (one board per workgroup: 12x12 .. 15x15): the 3x3 convolutions of lib/model.py:36-47 in
 * 2-D Winograd F(2x2,3x3) form -- 16 / 36 of the direct form's multiplies (the row form of caro_net_enable_winograd:
 * 24 / 36), the same float32 network function within the tolerance of tests/test_gpu_net.py.  ww2_host: the transformed
 * weights in kernel order, caro_net_winograd2d_size() floats (caro_ai_amd/net_hip.py:pack_net_w2).  Mutually exclusive
 * with the other arithmetic modes of a net.  A forward call of such a net is TWO launches on the caller's stream: the
 * trunk (one board per workgroup) and the FC heads + softmax of the whole call, 32 boards per workgroup; the first call
 * allocates the feature rows that travel between them (3 * H * W floats per row of the largest launch seen).  The rows
 * are kept per (net handle, stream) -- the first net's, for a pair -- so launches of one handle on different streams may
 * overlap; a handle serves at most 8 streams, and host calls on one handle are not thread-safe. */
int caro_net_winograd2d_size(void);
int caro_net_winograd2d_supported(int H, int W);
int caro_net_enable_winograd2d(caro_net* n, const float* ww2_host, int64_t n_floats);
/* bf16x3 mode -- an EXTRA arithmetic mode, not the default of any caller and not what bench.py's headline runs: every
 * float32 operand of the residual trunk (lib/model.py:36-47) as the sum of three bfloat16 parts, a product as the six
 * part products of weight 2^-16 and above on v_mfma_f32_16x16x32_bf16, float32 accumulation; conv_in, biases, residual
 * adds, LeakyReLU and the heads in float32 as in the default kernel.  NOT bit-identical to the float32 modes: within
 * the tolerance tests/test_gpu_net.py states for it.  parts_host: caro_net_split_bf16_size() uint16 =
 * [45 (layer, tap)][2 c][3 parts][4 kg][64 co][8 ci] bfloat16 bit patterns, ci = 32 c + 8 kg + 0..7
 * (caro_ai_amd/net_hip.py:pack_net_x3).  Mutually exclusive with the other arithmetic modes of a net.  On boards served one
 * per workgroup (12x12 .. 15x15) a forward call is two launches, as in f32w2 mode: the trunk, then the FC heads + softmax of
 * the whole call 32 boards per workgroup, with the feature rows kept per (net handle, stream) as described below. */
int64_t caro_net_split_bf16_size(void);
int caro_net_enable_split_bf16(caro_net* n, const uint16_t* parts_host, int64_t n_u16);
/* how often a slot of that (handle, stream) table had to change its stream: 0 while a handle serves at most 8 streams;
 * every eviction costs a device synchronisation (the slot's rows are re-used, not re-allocated) */
int64_t caro_net_stream_evictions(const caro_net* n);
void caro_net_destroy(caro_net* n);
int caro_net_boards_per_workgroup(const caro_net* n);
/* rows [row0, row0 + L) of planes_dev f32[max_rows,2,H,W] -> probs_dev f32[.,A] (softmaxed), values_dev f32[.]
 * with L = counts_dev[which] and row0 = which ? counts_dev[0] : 0, both read ON DEVICE. */
int caro_net_forward(caro_net* n, const float* planes_dev, const int32_t* counts_dev, int which, int64_t max_rows,
                     float* probs_dev, float* values_dev, void* stream);

/* slot rows, the form the fused tree kernel of caro_search_batch produces: the j-th unique leaf of game g sits
 * at row g * batch + j of planes_dev (its priors / value come back in the same row of probs_dev / values_dev),
 * gpack_dev i32[n_games] = leaf count | net class << 8, counts_dev = {L0, L1} totals per net, all read ON DEVICE.
 * Every workgroup maps its dense tile of boards onto the slot rows in game order itself, so which leaves share a
 * tile (and the tile size) is a function of the games' states only, never of block arrival order.
 * n1 may be NULL (one net, every game class 0). */
int caro_net_forward_slots(caro_net* n0, caro_net* n1, const float* planes_dev, const int32_t* counts_dev,
                           const int32_t* gpack_dev, int n_games, int batch, float* probs_dev, float* values_dev,
                           void* stream);

/* The same with the dense order of the leaves GIVEN by the producer: slot_list_dev i32[2][n_games * batch], entry
 * [c][i] = slot row of the i-th leaf of net class c (i < counts_dev[c]), in any order -- the multi-wave fused tree kernel
 * of caro_search_batch appends a game's rows when its block gets there.  Only the net forms that serve ONE board per
 * workgroup use it (2-D Winograd, 12x12 .. 15x15: a board's arithmetic does not depend on its dense index, and no
 * workgroup has to derive the map from gpack_dev any more); every other form ignores the list and keeps the game-order
 * map, so which boards share a tile stays a function of the games' states.  slot_list_dev may be NULL. */
int caro_net_forward_slot_list(caro_net* n0, caro_net* n1, const float* planes_dev, const int32_t* counts_dev,
                               const int32_t* gpack_dev, const int32_t* slot_list_dev, int n_games, int batch,
                               float* probs_dev, float* values_dev, void* stream);

/* Table evaluator with the same launch interface as the conv net (leaf counts read on device, dense or slot
 * rows): an exact integer-hash "net" for checking the SEARCH bit for bit -- it stands where lib/mcts.py:212-218
 * calls the net.  With x = the 2*H*W input planes of a row, mix64 = the splitmix64 finaliser of caro_noise.h:
 *   h    = salt + sum over j with x[j] != 0 of (mix64(0x5851f42d4c957f2d + j) | 1)          (mod 2^64)
 *   P[a] = (((mix64(h + 0x9E3779B97F4A7C15 * (a + 1)) >> 20) & 1023) + 1) / 8192            (float32, exact)
 *   v    = ((mix64(h ^ 0xA5A5A5A5A5A5A5A5) >> 20) % 2001 - 1000) / 1024                      (float32, exact)
 * (P is used as is, no softmax).  oracle/caro_oracle.c and tests/synth_net.py hold independent twins. */
int caro_net_create_hash(int H, int W, int A, uint64_t salt, int device_id, caro_net** out);

/* A HIP stream confined to the compute units [part/nparts, (part+1)/nparts) of the device
 * (hipExtStreamCreateWithCUMask): independent engines on such streams run side by side on disjoint CUs. */
int caro_stream_create_partition(int device_id, int part, int nparts, void** stream_out);
int caro_stream_destroy(void* stream);
/* both nets of an arena in ONE launch: rows [0, L0) through n0, rows [L0, L0+L1) through n1 */
int caro_net_forward_pair(caro_net* n0, caro_net* n1, const float* planes_dev, const int32_t* counts_dev,
                          int64_t max_rows, float* probs_dev, float* values_dev, void* stream);
/* the same with n1's rows starting at row1_base instead of L0 (row1_base < 0: at L0) */
int caro_net_forward_pair_at(caro_net* n0, caro_net* n1, const float* planes_dev, const int32_t* counts_dev,
                             int64_t row1_base, int64_t max_rows, float* probs_dev, float* values_dev, void* stream);
/* diagnostic form: also writes per workgroup (total shader cycles, 100 MHz wall ticks, cycles at trunk start, at trunk end) to stamps_dev u64[4*grid],
 * grid = ceil(max_rows / caro_net_boards_per_workgroup); used by tools/probe_clock.py only */
int caro_net_forward_stamped(caro_net* n, const float* planes_dev, const int32_t* counts_dev, int which,
                             int64_t max_rows, float* probs_dev, float* values_dev, uint64_t* stamps_dev,
                             void* stream);
/* diagnostic: from now on every launch of this net's float32 kernels -- the engine's own launches included -- writes the
 * same per-workgroup stamps to stamps_dev u64[4 * grid] (grid <= games * batch / boards per workgroup + 2);
 * NULL switches it off.  tools/probe_engine_net.py only */
int caro_net_debug_stamps(caro_net* n, uint64_t* stamps_dev);
/* MCTS.search_batch (lib/mcts.py:162-176) for every live game with the fused net(s): `searches` x
 * (caro_select -> caro_net_forward per net -> caro_expand_backup) enqueued on `stream` from one call, no host
 * synchronisation.  noise_dev: f64[searches, G, batch, A] or NULL (generated); buffers as for caro_select /
 * caro_expand_backup (G * batch rows); net1 may be NULL when the engine has one net.  With one wavefront per game
 * (batch * lanes-per-descent == 64) the three tree kernels run fused (k_tree), two launches per minibatch, and
 * leaves travel in slot rows (caro_net_forward_slots); with several whole wavefronts per game (a multiple of 64 above
 * 64) the same fusion runs as k_tree_mw; otherwise (less than one wavefront) in the dense rows of caro_select.
 * A move needs searches >= 2 when its root may be unexpanded: the first minibatch on an unexpanded root only expands it
 * (lib/mcts.py:123: every descent returns the root itself, nothing is backed up), so after ONE search no edge has been
 * visited and the policy is 0 / 0 -- the reference raises ZeroDivisionError there (lib/mcts.py:311); caro_policy
 * returns NaN rows for such a game, and caro_step / the staggered ply REFUSE the ply (the game stays where it was,
 * actions_dev = -1, counters[5] is bumped) when tau = 1 or when action 0 -- the reference's argmax of an all-zero
 * row at tau = 0 -- is not a legal move. */
int caro_search_batch(caro_engine* h, caro_net* net0, caro_net* net1, int searches, int batch,
                      const double* noise_dev, float* planes_dev, uint64_t* leaf_keys_dev, float* probs_dev,
                      float* values_dev, void* stream);

/* One whole move of play_game's loop body (lib/utils.py:76-99: search_batch, get_policy_value, the sampled move,
 * game.move, win / draw) for every live game: caro_search_batch followed by caro_step, arguments as for those two.
 * Where several wavefronts serve a game (batch x lanes per descent a multiple of 64 above 64: the 15 x 15 board with 8
 * descents) the ply -- and, with caro_config.evict, the eviction that follows it -- runs inside the search's closing
 * tree launch: one launch per move instead of three.  Results are those of the two calls. */
int caro_search_move(caro_engine* h, caro_net* net0, caro_net* net1, int searches, int batch, const double* noise_dev,
                     const double* uniforms_dev, float* planes_dev, uint64_t* leaf_keys_dev, float* probs_dev,
                     float* values_dev, int32_t* actions_dev, int32_t* done_dev, int32_t* result_dev, void* stream);

/* ---- staggered mode: every game on its own minibatch clock (the hot path of bench.py / train.self_play) ----
 * In lock-step all games reach a move together, and the launches right after a move carry far more new leaves
 * than the rest, so the net launch overflows one round of tiles exactly there.  The reference plays its games one
 * after another (train.py:41-47) -- nothing ties their moves together -- so here each game counts its own
 * minibatches: game g sits out g % searches launches at the start, makes its ply INSIDE the tree kernel when its
 * `searches` minibatches are done (lib/utils.py:80-99), and a finished game is parked (record + history rows copied
 * aside) and its slot restarted at once (uid += uid_stride) when `recycle` != 0.  Every launch then carries the
 * same mix of minibatch indices.  Game by game the results are those of caro_search_batch + caro_step: the same
 * minibatches on the same tree with the same noise keys.  Needs whole wavefronts per game (batch x lanes per descent a
 * multiple of 64: connect four with batch 8 = one wavefront, k_tree_stag; several wavefronts -- TicTacToe with batch 8,
 * 15 x 15 with batch 8 -- k_tree_stag_mw, round 6), generated noise / move uniforms, a fresh or restarted engine.
 * caro_config.evict combines with it where several wavefronts serve a game: the eviction then runs inside the kernel,
 * right behind the game's ply (a finished game drops every node, which also leaves both key tables clean for the restart).
 *   caro_config.stagger     = mcts_searches at caro_engine_create (fixed for the engine's life)
 *   caro_search_staggered   `launches` x (tree kernel -> net kernel); on average every game moves once per
 *                           `searches` launches
 *   caro_drain_parked_begin tuples of the parked games, as caro_drain_tuples_begin (no recycle flag: the slots have
 *                           restarted already); finish with caro_drain_tuples_end
 * A staggered engine is driven by these two calls only: the lock-step mutators (caro_reset_games, caro_set_roots,
 * caro_select, caro_search_batch, caro_step, caro_drain_tuples[_begin]) return CARO_E_STATE on it -- they know nothing
 * of its per-game clocks, pending minibatches and parked records.  Read-only calls (caro_counters, caro_get_roots,
 * caro_policy, caro_lookup_nodes, caro_tree_sizes, caro_pending_leaves ...) work on both kinds. */
int caro_search_staggered(caro_engine* h, caro_net* net0, caro_net* net1, int launches, int batch, float* planes_dev,
                          uint64_t* leaf_keys_dev, float* probs_dev, float* values_dev, void* stream);
int caro_drain_parked_begin(caro_engine* h, int64_t cap, uint64_t* states_dev, int32_t* players_dev, double* pi_dev,
                            int32_t* z_dev, int64_t* games_dev, void* stream);

#ifdef __cplusplus
}
#endif
#endif /* CARO_HIP_H */
