#!/usr/bin/env python3
"""Self-play -> train -> arena loop, the drop-in for the reference's train.py:165-217
on top of the batched HIP engine.

Same stages, names, hyper-parameters (caro_ai_amd/config.py == config.py) and
artefacts as the reference:

  self_play          train.py:25-59    N concurrent games on the GPU instead of PLAY_EPISODES serial ones;
                                       `speed_nodes` / `speed_steps` keep their meaning (train.py:49-54)
  train_neural_net   train.py:62-117   TRAIN_ROUNDS batches of BATCH_SIZE sampled without replacement,
                                       loss = MSE(v, z) + mean(-sum(log_softmax(logits) * pi)), SGD(0.1, 0.9);
                                       the replay buffer lives on the device (tuples never visit the host)
  evaluate           train.py:120-149  EVALUATION_ROUNDS arena games, 20 x 16 sims, tau = 0; win ratio
  checkpoints        train.py:210-217  best_%03d_%05d.dat = torch.save(net.state_dict()) when the ratio > 0.60

Declared deviations: a fresh tree per game instead of one store shared across games,
eval-mode batch-norm during search.  Multi-GPU: games are sharded (caro_ai_amd.parallel), tuples are
all-gathered, rank 0 trains and broadcasts the weights (--ddp: every rank trains on its share of each batch and the
gradients are all-reduced, train_neural_net).

    python -m caro_ai_amd.train -n run -g 0 --cuda --games 256 --iterations 50
"""
import argparse
import collections
import os
import sys
import time

import numpy as np
import torch
import torch.nn.functional as F
import torch.optim as optim

from caro_ai_amd import _lib, parallel
from caro_ai_amd import config as cfg
from caro_ai_amd.lib.game import game_provider
from caro_ai_amd.lib.model import Net, NetWrapper


class DeviceReplayBuffer:
    """`collections.deque(maxlen=REPLAY_BUFFER)` of (state, player, pi, z) tuples (train.py:184) as a ring of
    device tensors.  States are kept packed (the engine's key words); NN planes are produced on the device
    by the batched rules kernel when a batch is sampled."""

    def __init__(self, game, capacity=cfg.REPLAY_BUFFER, device="cuda:0"):
        self.game = game
        self.capacity = int(capacity)
        self.device = torch.device(device)
        self.KW, self.A = game.key_words, game.action_space
        self.states = torch.zeros((self.capacity, self.KW), dtype=torch.int64, device=self.device)
        self.players = torch.zeros(self.capacity, dtype=torch.int32, device=self.device)
        self.pi = torch.zeros((self.capacity, self.A), dtype=torch.float32, device=self.device)
        self.z = torch.zeros(self.capacity, dtype=torch.float32, device=self.device)
        self.size = 0
        self.head = 0  # next write position

    def __len__(self):
        return self.size

    def extend(self, tuples):
        """append a drain's tuples (dict of device tensors) in order; the oldest entries fall out (deque maxlen)"""
        n = int(tuples["z"].shape[0])
        if n == 0:
            return
        if n > self.capacity:
            tuples = {k: v[-self.capacity:] for k, v in tuples.items()}
            n = self.capacity
        idx = (torch.arange(n, device=self.device) + self.head) % self.capacity
        self.states[idx] = tuples["states"].to(self.device)
        self.players[idx] = tuples["players"].to(self.device, dtype=torch.int32)
        self.pi[idx] = tuples["pi"].to(self.device, dtype=torch.float32)
        self.z[idx] = tuples["z"].to(self.device, dtype=torch.float32)
        self.head = (self.head + n) % self.capacity
        self.size = min(self.capacity, self.size + n)

    def sample(self, batch_size, generator=None):
        """random.sample(replay_buffer, BATCH_SIZE) (train.py:77): without replacement"""
        assert self.size >= batch_size
        perm = torch.randperm(self.size, device=self.device, generator=generator)[:batch_size]
        return self.states[perm], self.players[perm], self.pi[perm], self.z[perm]

    def planes(self, states, players):
        """game.states_to_training_batch on the device (lib/game rules kernel)"""
        n = states.shape[0]
        out = torch.empty((n,) + tuple(self.game.obs_shape), dtype=torch.float32, device=self.device)
        if self.device.type == "cuda":
            L = _lib.load()
            st = torch.cuda.current_stream(self.device).cuda_stream
            _lib.check(L.caro_rules_encode_batch(self.game.kind, self.game.n, self.game.k, n,
                                                 states.contiguous().data_ptr(), players.contiguous().data_ptr(),
                                                 out.data_ptr(), st))
        else:  # CPU tensors (tests of the training arithmetic): host-side rules helper
            ints = self.game.from_keys(states.cpu().numpy().view(np.uint64))
            out.copy_(torch.from_numpy(self.game.states_to_training_batch(ints, players.cpu().tolist())))
        return out


def loss_terms(out_logits, out_values, probs, values):
    """train.py:98-106"""
    loss_value = F.mse_loss(out_values.squeeze(-1), values)
    loss_policy = (-F.log_softmax(out_logits, dim=1) * probs).sum(dim=1).mean()
    return loss_policy + loss_value, loss_value, loss_policy


def train_neural_net(game, replay_buffer, net, optimizer, device="cuda:0", train_rounds=cfg.TRAIN_ROUNDS,
                     batch_size=cfg.BATCH_SIZE, generator=None, ddp=False):
    """TRAIN_ROUNDS SGD steps on batches sampled from the replay buffer; returns the mean losses
    (what train.py:113-117 sends to TensorBoard as loss_total / loss_value / loss_policy).

    ddp=True (several ranks, every rank calls this with the SAME buffer content and the same `generator` state -- the
    gathered tuples are identical everywhere): the ranks draw the same batch, each runs forward / backward on its
    share rank::world of it with the loss scaled by share / batch, the gradients are summed over the ranks
    (parallel.allreduce_grads: one flat bucket over RCCL) and every rank takes the same optimizer step, so the weights
    stay identical without a broadcast.  The sum is the gradient of the reference's full-batch loss EXCEPT for the
    batch-norm statistics, which each rank takes over its own share (declared deviation; the default, ddp=False, is the
    reference's single-device step on rank 0 followed by a weight broadcast)."""
    net.train()
    sums = np.zeros(3)
    rank, world = 0, 1
    if ddp and parallel.is_dist():
        rank, _, world = parallel.env_rank()
    for _ in range(train_rounds):
        states, players, probs, values = replay_buffer.sample(batch_size, generator)
        if world > 1:
            states, players, probs, values = (t[rank::world] for t in (states, players, probs, values))
        share = states.shape[0] / batch_size
        optimizer.zero_grad()
        if states.shape[0]:
            x = replay_buffer.planes(states, players)
            out_logits, out_values = net(x)
            loss, loss_value, loss_policy = loss_terms(out_logits, out_values, probs, values)
            (loss * share).backward()
            local = torch.stack([loss.detach(), loss_value.detach(), loss_policy.detach()]).double() * share
        else:
            local = torch.zeros(3, dtype=torch.float64, device=states.device)
        if world > 1:
            for p in net.parameters():  # a rank with an empty share still takes part in the collective
                if p.grad is None:
                    p.grad = torch.zeros_like(p)
            parallel.allreduce_grads(list(net.parameters()))
            parallel.allreduce_sum(local)
        optimizer.step()
        sums += local.cpu().numpy()
    sums /= train_rounds
    return {"loss_total": sums[0], "loss_value": sums[1], "loss_policy": sums[2]}


def staggered_ok(game, batch, evict=False):
    """the geometry staggered mode needs: whole wavefronts per game (batch x lanes per descent a multiple of 64; with
    eviction: above 64)"""
    from caro_ai_amd.engine import staggered_geometry
    return staggered_geometry(game, batch, evict)


# The engines of self_play, kept between calls (the reference builds its MCTS store once, train.py:185, and plays every
# self-play call on it): a call whose shape -- game, slots, batch, node cap, schedule, device -- has been seen before
# restarts the engine in place (caro_engine_restart: trees cleared, not re-allocated; 4.6 GB for 1024 connect-four
# slots) and plays, bit for bit, what a fresh engine plays.  Most recently used last; the oldest is closed.
_ENGINES = collections.OrderedDict()
ENGINE_CACHE = 2


def release_engines():
    """close the cached self-play engines and drop the cached HipNets (frees their device memory)"""
    from caro_ai_amd import net_hip
    while _ENGINES:
        _ENGINES.popitem()[1].close()
    net_hip.release_hipnets()


def _engine_for(game, G, batch, searches, device, stagger, run, hip, reuse, node_cap=None):
    """(engine, reused?) ready to play a run keyed by `run` (SelfPlayEngine.RUN_FIELDS) with the net `hip`"""
    from caro_ai_amd.engine import SelfPlayEngine
    hw = game.obs_shape[1] * game.obs_shape[2]
    # (boards whose no-overflow bound is beyond a default tree run with eviction, as lib.utils.play_games does)
    evict = not node_cap and searches * batch * hw + 64 > SelfPlayEngine.DEFAULT_CAP_LIMIT
    cap = int(node_cap) if node_cap else SelfPlayEngine.default_node_cap(searches, batch, hw, evict)
    stagger = bool(stagger) and staggered_ok(game, batch, evict)
    key = (type(game).__name__, game.kind, game.n, game.k, G, batch, cap, evict, stagger, str(torch.device(device)))
    eng = _ENGINES.pop(key, None) if reuse else None
    if eng is not None and eng.h:
        eng.restart(evaluators=[hip], searches=searches, **run)
        _ENGINES[key] = eng
        return eng, True
    eng = SelfPlayEngine(game, G, evaluators=[hip], max_batch=batch, node_cap=cap, device=device,
                         searches_hint=searches, stagger=stagger, evict=evict, **run)
    if reuse:
        _ENGINES[key] = eng
        while len(_ENGINES) > ENGINE_CACHE:
            _ENGINES.popitem(last=False)[1].close()
    return eng, False


def _forget_engine(eng):
    for k in [k for k, e in _ENGINES.items() if e is eng]:
        del _ENGINES[k]
    try:
        eng.close()
    except Exception:  # (a close that fails after a device error must not mask the error that brought us here)
        pass


class _Drains:
    """what the move loop of a self-play call collects.  Nothing here waits for the GPU -- the next move is already
    enqueued on the stream, and one synchronising read would hold the host until that move is over: counts come from
    shapes, the game records are looked at once, after the loop (`records()`)"""

    def __init__(self):
        self.finished = self.rows = 0
        self._records = []
        self.gatherer = parallel.TupleGatherer(every=1 << 30, pi_dtype=torch.float32)

    def take(self, d):
        if d is None or not int(d["games"].shape[0]):
            return
        self.finished += int(d["games"].shape[0])
        self.rows += int(d["z"].shape[0])
        self._records.append(d["games"])
        self.gatherer.push(d)

    def records(self):
        """(uid, first player, result, steps) of every drained game, on the host"""
        return torch.cat(self._records).cpu().numpy() if self._records else np.zeros((0, 4), np.int64)

    def deliver(self, replay_buffer):
        """the exchange -- ONE collective per call, when every rank has left its rank-local loop -- and the append"""
        out = self.gatherer.flush()
        if out is not None:
            replay_buffer.extend(out)


def _abort(eng):
    """a self-play call failed: the engine goes whatever happens (its trees are gigabytes, its state is unknown).  Under
    several ranks the error must END this rank: the peers are on their way to the collective of `_Drains.deliver` and would
    wait there for the backend's timeout; a rank that exits non-zero is what the launcher's fail-fast path acts on."""
    _forget_engine(eng)
    if parallel.is_dist():
        import traceback
        traceback.print_exc()
        sys.stderr.flush()
        os._exit(13)


def _stats(steps, nodes, dr, t_call, t_ready, t_played, reused, passes):
    dt = time.time() - t_call
    return {"speed_steps": steps / dt, "speed_nodes": nodes / dt, "steps": steps, "nodes": nodes, "games": dr.finished,
            "games_dropped": 0, "rows": dr.rows, "seconds": dt, "seconds_setup": t_ready - t_call,
            "seconds_play": t_played - t_ready, "seconds_gather": time.time() - t_played, "engine_reused": reused,
            "passes": passes, "speed_nodes_play": nodes / max(t_played - t_ready, 1e-9)}


def self_play_stream(game, replay_buffer, net, n_games, device="cuda:0", seed=0, uid_base=0,
                     searches=cfg.MCTS_SEARCHES, batch=cfg.MCTS_BATCH_SIZE, concurrent=None, node_cap=None, net_mode="f32w",
                     streams=1):
    """self_play as a STREAM: the engine is never stopped between calls.  Every slot restarts the moment its game ends
    (uid += stride, in the tree kernel) and a call returns as soon as n_games games have FINISHED since the previous
    call; the games then in flight are not thrown away -- they finish inside the next call and reach the replay buffer
    there.  Between two promotions train.py plays every self-play game with the same best net (train.py:44,185-217),
    so a game started during one iteration and finished during the next is the same game the reference would have
    played; every started game is consumed exactly once, none is dropped, and the GPU never runs the sparse tail of
    a generation -- the pipeline is bench.py's, the rate bench.py's.
    What changes against `self_play`: WHEN a game's tuples arrive (a long game may land one iteration later), not
    which games are played or how.  When the net's weights have changed since the stream was started (a promotion),
    the games in flight belong to the old net: the stream is restarted (they are dropped, at most one game per slot).
    streams=2 (opt-in): the slots as two engines of half the slots on two HIP streams, the float32 net kernel with full
    tiles only -- a half's net launch then takes half the compute units, the halves' launches run side by side and each
    half's tree kernels beside the other half's net launch (bench.py `two_streams`: +3 %, with the bf16x3 kernel +7 %);
    the same uids as one engine.
    Needs the staggered geometry (whole wavefronts per game: `staggered_ok`).  Returns what self_play returns; `nodes` / `speed_nodes`
    count the node-expansions of this call's launches (incl. the part of the in-flight games played in it)."""
    from caro_ai_amd import net_hip
    t_call = time.time()
    rank, _, world = parallel.env_rank() if parallel.is_dist() else (0, 0, 1)
    if not staggered_ok(game, batch):
        raise _lib.CaroError("self_play_stream needs whole wavefronts per game (batch x lanes per descent a multiple of 64)")
    G = max(1, int(concurrent or n_games))
    stride = world * G
    streams = max(1, int(streams))
    if G % streams:
        raise _lib.CaroError("self_play_stream: %d slots do not split over %d streams" % (G, streams))
    hip = net_hip.hipnet_for(net, device, mode=net_mode, split_tiles=streams == 1)
    hw = game.obs_shape[1] * game.obs_shape[2]
    from caro_ai_amd.engine import SelfPlayEngine, StreamedSelfPlay
    cap = int(node_cap) if node_cap else SelfPlayEngine.default_node_cap(searches, batch, hw)
    key = ("stream", type(game).__name__, game.kind, game.n, game.k, G, batch, cap, str(torch.device(device)), streams)
    eng = _ENGINES.pop(key, None)
    ss = getattr(eng, "_stream_state", None) if eng is not None and eng.h else None
    reused = ss is not None and ss["hip"] is hip and ss["searches"] == searches
    if not reused:
        base = uid_base + rank * G
        if ss is not None:  # keep uids unique across restarts: beyond anything the old stream may have started
            base = max(base, ss["base"] + (ss["passes"] // 4 + 2) * stride)
        run = dict(seed=seed, uid_base=base, uid_stride=stride, games_limit=0, stagger_recycle=True,
                   steps_before_tau_0=cfg.STEPS_BEFORE_TAU_0)
        if eng is not None and eng.h:
            eng._drain_open and eng.flush()
            eng.restart(evaluators=[hip], searches=searches, **run)
        elif streams > 1:
            eng = StreamedSelfPlay(game, G, lambda: [hip], n_streams=streams, device=device, partition_cus=False,
                                   max_batch=batch, node_cap=cap, searches_hint=searches, stagger=True, **run)
        else:
            eng = SelfPlayEngine(game, G, evaluators=[hip], max_batch=batch, node_cap=cap, device=device,
                                 searches_hint=searches, stagger=True, **run)
        ss = {"hip": hip, "searches": searches, "base": base, "passes": 0,
              "c": dict.fromkeys(("expansions", "overflows", "plies", "finished"), 0)}
        eng._stream_state = ss
    _ENGINES[key] = eng
    while len(_ENGINES) > ENGINE_CACHE:
        _ENGINES.popitem(last=False)[1].close()
    t_ready = time.time()
    dr = _Drains()
    try:
        max_passes = (hw + 4) * (-(-n_games // G)) + searches + 8
        passes = 0
        # (no flush at the end: the last enqueued pass keeps the GPU busy while the host goes on, its rows are handed
        # out by the first move() of the next call)
        while dr.finished < n_games and passes <= max_passes:
            dr.take(eng.move(searches, batch, recycle=True))
            passes += 1
        ss["passes"] += passes
        t_played = time.time()
        c = eng.counters()
        if c["overflows"] > ss["c"]["overflows"]:
            raise _lib.CaroError("self_play_stream: %d minibatches overflowed the node pool (node_cap=%d)"
                                 % (c["overflows"] - ss["c"]["overflows"], eng.cfg.node_cap))
        if dr.finished < n_games:
            raise _lib.CaroError("self_play_stream: %d of %d games finished after %d passes" % (dr.finished, n_games, passes))
        nodes = c["expansions"] - ss["c"]["expansions"]
        ss["c"] = {k: c[k] for k in ss["c"]}
        recs = dr.records()
        if len(np.unique(recs[:, 0])) != len(recs):
            raise _lib.CaroError("self_play_stream: a game was drained twice")
        steps = int(recs[:, 3].sum())
    except BaseException:
        _abort(eng)
        raise
    try:
        dr.deliver(replay_buffer)
    except BaseException:
        _forget_engine(eng)
        raise
    return _stats(steps, nodes, dr, t_call, t_ready, t_played, reused, passes)


def self_play(game, replay_buffer, net, n_games, device="cuda:0", seed=0, uid_base=0, searches=cfg.MCTS_SEARCHES,
              batch=cfg.MCTS_BATCH_SIZE, concurrent=None, stagger=False, reuse=True, node_cap=None, pool=True, net_mode="f32w"):
    """Play n_games (per rank) with the (best) net against itself, tuples appended on the device.
    Returns speed_steps, speed_nodes, steps, nodes (train.py:49-58) on the wall clock of the WHOLE call -- engine
    construction or restart, weight upload, the games, the tuple exchange --, plus where the time went.

    WHICH games: the uids uid_base + rank*G + g + k * world * G with local index k*G + g < n_games (the engine's
    `games_limit`) -- slot g plays the k-th of them itself, or (staggered, `pool=True`, fewer slots than games) whichever
    slot is free next is handed the next one not started yet, in slot order at every drain: the same set of games either
    way.  A slot with no wanted game left stays finished, so no game outside the wanted set is
    ever started, every counted node-expansion belongs to a wanted game, and the replay buffer never holds a
    length-biased "first to finish" sample.  The same set whether the engine runs lock-step or staggered.
    stagger=True (the CLI's choice where the geometry allows): the engine's staggered mode -- every game on its own
    minibatch clock; each game is the one the lock-step form plays for the same uid, only the ORDER in which games
    reach the replay buffer differs.
    reuse=True: the engine (and the HipNet, while the weights do not change) is kept for the next call of the same
    shape and restarted in place; a reused engine plays the games of a fresh one bit for bit
    (tests/test_gpu_stagger.py::test_reused_self_play_engine_plays_the_fresh_engines_games).
    node_cap: nodes per tree (default: searches x batch x cells, which cannot overflow).
    net_mode: the HipNet arithmetic mode (net_hip.HipNet): "f32w" (default, float32) or the opt-in "bf16x3" (split
    bfloat16 operands, float32 accumulate: 1.3 x the leaves/s, outputs within the float32 kernels' own tolerance, not
    bit-identical to them).
    Raises CaroError if a tree overflowed its node pool (the games would no longer be the reference's)."""
    from caro_ai_amd import net_hip
    t_call = time.time()
    rank, _, world = parallel.env_rank() if parallel.is_dist() else (0, 0, 1)
    G = max(1, min(int(concurrent or n_games), int(n_games)))
    stagger = bool(stagger) and staggered_ok(game, batch)
    restarts = n_games > G
    base, stride = uid_base + rank * G, world * G
    # (staggered with fewer slots than games: the pool form -- a finished slot is handed the next game not started yet at
    # the next drain, so the slots stay busy until the wanted games run out; with each slot tied to its own uids g, g + G, ...
    # the call ended with the longest chain of a slot's games: 111-115 passes for 4 096 games on 1 024 slots against 92-94)
    run = dict(seed=seed, uid_base=base, uid_stride=stride, games_limit=n_games,
               stagger_recycle=(2 if (stagger and pool) else 1) if restarts else 0, steps_before_tau_0=cfg.STEPS_BEFORE_TAU_0)
    hip = net_hip.hipnet_for(net, device, mode=net_mode)
    eng, reused = _engine_for(game, G, batch, searches, device, stagger, run, hip, reuse, node_cap)
    t_ready = time.time()
    dr = _Drains()  # (every drained game is a wanted one: games_limit)
    try:
        # one pass = `searches` launches = one ply per game (staggered: on average; a game sits out fewer than
        # `searches` launches at the start): a bound on the passes that a healthy run never reaches
        hw = game.obs_shape[1] * game.obs_shape[2]
        max_passes = (hw + 4) * (-(-n_games // G)) + 8
        passes = 0
        while dr.finished < n_games and passes <= max_passes:
            dr.take(eng.move(searches, batch, recycle=restarts))  # host-pipelined: hands out the previous pass's rows
            passes += 1
        dr.take(eng.flush())
        t_played = time.time()
        c = eng.counters()
        if c["overflows"]:
            raise _lib.CaroError("self_play: %d minibatches overflowed the node pool (node_cap=%d) or plies were refused "
                                 "on a root without visits: the games are not the reference's" % (c["overflows"], eng.cfg.node_cap))
        if dr.finished < n_games:
            raise _lib.CaroError("self_play: %d of %d games finished after %d passes" % (dr.finished, n_games, passes))
        recs = dr.records()
        off = recs[:, 0] - base
        k, g = off // stride, off % stride
        if not ((off >= 0) & (g < G) & (k * G + g < n_games)).all() or len(np.unique(recs[:, 0])) != n_games:
            raise _lib.CaroError("self_play: the engine drained games outside the wanted set")
        steps = int(recs[:, 3].sum())
    except BaseException:
        _abort(eng)
        raise
    try:
        dr.deliver(replay_buffer)
    except BaseException:
        _forget_engine(eng)
        raise
    if not reuse:
        eng.close()
    return _stats(steps, c["expansions"], dr, t_call, t_ready, t_played, reused, passes)


def evaluate(game, challenger, champion, rounds=cfg.EVALUATION_ROUNDS, device="cuda:0", seed=0,
             reference_stores=False, counts=False, node_cap=None):
    """challenger (net1) vs champion (net2): `rounds` games, 20 x 16 sims, tau = 0 from move 0, one tree per
    player; returns challenger_win / (wins + losses + draws)  (train.py:120-149).
    With several ranks each plays a contiguous share of the rounds (round = game uid, so the set of games is the
    single-rank one) and the three counters are all-reduced: every rank gets the same ratio and takes the same
    promote / keep decision.
    DEVIATIONS from the reference's evaluate (declared): (1) the reference builds ONE pair of
    stores before its loop and reuses it for all rounds (train.py:134-141), so later rounds search on statistics
    left by earlier ones; here every round is an independent game with fresh trees (play.py:47 semantics, SURVEY
    Q3) -- that is what lets the rounds run concurrently and shard across ranks; (2) the reference draws the
    opening side with np.random.choice(2) per round, here it alternates with the round's uid (uid & 1), so a run
    is reproducible.  The promote / keep decision can therefore differ from the reference's for the same nets.

    reference_stores=True: the reference's evaluate itself (train.py:134-149) -- ONE pair [MCTS, MCTS] built before
    the loop and reused by every round, the rounds one after another through the single-game API
    (`lib.utils.play_game`: the opener by np.random.choice(2), one Dirichlet row per descent and one choice per ply
    from numpy's global stream, exactly the reference's draws), trees on the GPU.  Sequential by construction (round
    r searches on what rounds < r left behind), so only rank 0 plays and the counters are shared.  Pinned against
    rounds recorded from the reference: tests/test_gpu_shim.py::test_evaluate_with_reference_stores_*.
    Either form raises (CaroError / MemoryError) if a tree overflowed its node pool (`node_cap`: default = cannot)."""
    rank, _, world = parallel.env_rank() if parallel.is_dist() else (0, 0, 1)
    if reference_stores:
        from caro_ai_amd.lib import mcts as mcts_mod
        from caro_ai_amd.lib.utils import play_game
        res = []
        if rank == 0:
            stores = [mcts_mod.MCTS(game, tree_device=device, node_cap=node_cap),
                      mcts_mod.MCTS(game, tree_device=device, node_cap=node_cap)]
            # eval-mode batch-norm during search (declared deviation Q9; it also lets the stores take the fused one-call
            # search of lib/mcts.py); the nets get their training flags back
            modes = [(n, n.training) for n in (challenger, champion)]
            try:
                for n, _ in modes:
                    n.eval()
                for _ in range(rounds):
                    r, _ = play_game(game, stores, None, challenger, champion, steps_before_tau_0=0, mcts_searches=20,
                                     mcts_batch_size=16, device=device)
                    res.append(r)
            finally:
                for n, was in modes:
                    n.train(was)
        wins, losses, draws = parallel.allreduce_counts((res.count(1), res.count(-1), res.count(0)), device)
        ratio = wins / max(1, wins + losses + draws)
        return (ratio, (wins, losses, draws)) if counts else ratio
    from caro_ai_amd.lib.utils import play_games
    lo, n = parallel.shard_rounds(rounds, rank, world)
    res = []
    if n:
        res = play_games(game, n, None, challenger, champion, steps_before_tau_0=0, mcts_searches=20,
                         mcts_batch_size=16, concurrent=n, seed=seed, uid_base=lo, device=device, node_cap=node_cap)
    wins, losses, draws = parallel.allreduce_counts((res.count(1), res.count(-1), res.count(0)), device)
    ratio = wins / max(1, wins + losses + draws)
    return (ratio, (wins, losses, draws)) if counts else ratio  # counts=True: + (wins, losses, draws) of the challenger


class _NullWriter:
    def add_scalar(self, *a, **k):
        pass

    def close(self):
        pass


def _writer(name):
    try:
        from torch.utils.tensorboard import SummaryWriter
        return SummaryWriter(comment="-" + name)
    except Exception:
        return _NullWriter()


def parse_args(argv=None):
    p = argparse.ArgumentParser()
    p.add_argument("-n", "--name", required=True, help="Name of the run")
    p.add_argument("--cuda", default=False, action="store_true", help="Enable CUDA (the HIP engine needs it)")
    game_provider.add_game_argument(p)
    p.add_argument("--games", type=int, default=256, help="self-play games per iteration (reference: PLAY_EPISODES=1)")
    p.add_argument("--concurrent", type=int, default=0,
                   help="game slots on the GPU (default: one per game, at most 1024); fewer slots than games: slots restart")
    p.add_argument("--iterations", type=int, default=0, help="stop after this many iterations (0 = run for ever)")
    p.add_argument("--exact-self-play", action="store_true",
                   help="every iteration plays exactly its own --games games to the end (the GPU runs the sparse tail "
                        "of the last generation); default: self-play as a stream -- slots restart at once, an iteration "
                        "takes the first --games games that finish, games in flight carry over")
    p.add_argument("--saves", default="saves")
    p.add_argument("--reference-evaluate", action="store_true",
                   help="arena gate with the reference's store semantics: one pair of MCTS stores reused by all "
                        "rounds, rounds played one after another (the default in a single process)")
    p.add_argument("--sharded-evaluate", action="store_true",
                   help="arena gate as independent rounds with fresh trees, concurrent and sharded over the ranks (the "
                        "default under several ranks; a declared deviation from the reference's evaluate)")
    p.add_argument("--net-mode", default="f32w", choices=["f32w", "bf16x3"],
                   help="arithmetic of the self-play net kernel: f32w = float32 (default); bf16x3 = every float32 operand of "
                        "the residual trunk as three bfloat16 parts, float32 accumulate -- 1.3 x the leaves/s, outputs within "
                        "the float32 kernels' own tolerance but not bit-identical to them (the arena gate stays float32)")
    p.add_argument("--streams", type=int, default=1, choices=[1, 2],
                   help="self-play (stream form) as this many engines on separate HIP streams, the float32 net kernel with "
                        "full tiles only: 2 = +3 %% leaves/s (bench.py `two_streams`; +7 %% with --net-mode bf16x3)")
    p.add_argument("--ddp", action="store_true",
                   help="several ranks: every rank trains on its share of each batch, gradients all-reduced "
                        "(default: rank 0 trains, the weights are broadcast)")
    return p.parse_args(argv)


def fit(game, net, device, games, iterations=0, saves_path=None, writer=None, reference_evaluate=None, ddp=False,
        sample_seed=None, stop=None, log=print, concurrent=None, stream=False, net_mode="f32w", streams=1):
    """The reference's training loop (train.py:165-217): self-play with the best net -> replay buffer -> TRAIN_ROUNDS SGD
    steps -> every EVALUATE_EVERY_STEP iterations the arena gate (challenger = the net being trained against the best
    net; promoted when its win ratio exceeds BEST_NET_WIN_RATIO: `NetWrapper.sync`, `best_%03d_%05d.dat`).
    reference_evaluate: the gate with the REFERENCE's evaluate semantics (train.py:134-141: one pair of stores reused by
    all 20 rounds, the opener from np.random.choice(2), rounds one after another; `evaluate(reference_stores=True)`) --
    the default (None) in a single process, where it costs ~2 s every EVALUATE_EVERY_STEP = 100 iterations; False (and
    the default under several ranks): independent rounds with fresh trees, concurrent and sharded over the ranks, a
    declared deviation whose promote / keep decision can differ for the same nets.
    `games` self-play games per iteration (reference: PLAY_EPISODES = 1), `iterations` 0 = for ever.
    sample_seed: seed of the replay sampling (None: torch's global generator, as the reference); stop(history) -> True
    ends the loop early (several ranks: rank 0 decides, the others follow).  `concurrent`: game slots per rank (default:
    one per game).  stream=True: self-play as a stream (`self_play_stream`: slots restart at once, an iteration takes
    the first `games` games that finish, games in flight carry over to the next iteration -- no sparse tail; where the
    geometry has no staggered mode the exact form is used).  net_mode: the self-play net kernel's arithmetic (`self_play`;
    the arena gate always runs float32); streams: the stream form on that many half-engines (`self_play_stream`).  Returns the history: per trained iteration the three losses, per evaluation (iteration, win
    ratio, promoted), the number of promotions, the best net wrapper, and per iteration the seconds each phase took
    (`phases`: self_play -- with its own setup / play / gather split --, train, broadcast, evaluate)."""
    rank, _, world = parallel.env_rank() if parallel.is_dist() else (0, 0, 1)
    if reference_evaluate is None:
        reference_evaluate = world == 1
    writer = writer or _NullWriter()
    best_net = NetWrapper(net)
    optimizer = optim.SGD(net.parameters(), lr=cfg.LEARNING_RATE, momentum=0.9)
    replay_buffer = DeviceReplayBuffer(game, cfg.REPLAY_BUFFER, device)
    hist = {"loss_total": [], "loss_value": [], "loss_policy": [], "evaluations": [], "promotions": 0,
            "best_net": best_net, "speed_nodes": [], "iterations": 0, "phases": []}
    step_idx = best_idx = 0

    def clock():
        if str(device).startswith("cuda"):
            torch.cuda.synchronize(device)
        return time.time()

    while iterations == 0 or step_idx < iterations:
        t0 = clock()
        if stream and staggered_ok(game, cfg.MCTS_BATCH_SIZE):
            sp = self_play_stream(game, replay_buffer, best_net.target_model, games, device=device, seed=0,
                                  uid_base=step_idx * games * world, concurrent=concurrent, net_mode=net_mode,
                                  streams=streams)
        else:
            sp = self_play(game, replay_buffer, best_net.target_model, games, device=device, seed=step_idx,
                           uid_base=step_idx * games * world, stagger=True, concurrent=concurrent, net_mode=net_mode)
        ph = {"self_play": clock() - t0, "self_play_setup": sp["seconds_setup"], "self_play_play": sp["seconds_play"],
              "self_play_gather": sp["seconds_gather"], "engine_reused": sp["engine_reused"], "nodes": sp["nodes"],
              "train": 0.0, "broadcast": 0.0, "evaluate": 0.0}
        hist["phases"].append(ph)
        step_idx += 1
        hist["iterations"] = step_idx
        hist["speed_nodes"].append(sp["speed_nodes"])
        writer.add_scalar("speed_steps", sp["speed_steps"], step_idx)
        writer.add_scalar("speed_nodes", sp["speed_nodes"], step_idx)
        if rank == 0 and log:
            log("Step %d, steps %3d, leaves %4d, steps/s %5.2f, leaves/s %6.2f, best_idx %d, replay %d" % (
                step_idx, sp["steps"], sp["nodes"], sp["speed_steps"], sp["speed_nodes"], best_idx, len(replay_buffer)))
        if len(replay_buffer) < cfg.MIN_REPLAY_TO_TRAIN:
            continue
        t0 = clock()
        gen = None
        if (ddp and world > 1) or sample_seed is not None:
            gen = torch.Generator(device=replay_buffer.device)
            gen.manual_seed((sample_seed or 0) + step_idx)  # ddp: every rank draws the same batches from its (identical) buffer
        if ddp and world > 1:
            losses = train_neural_net(game, replay_buffer, net, optimizer, device, generator=gen, ddp=True)
        elif rank == 0:
            losses = train_neural_net(game, replay_buffer, net, optimizer, device, generator=gen)
        if rank == 0:
            for k, v in losses.items():
                writer.add_scalar(k, v, step_idx)
                hist[k].append(float(v))
        ph["train"] = clock() - t0
        t0 = clock()
        # (ddp: the parameters are already identical; the batch-norm running statistics are each rank's own: rank 0's go out)
        parallel.broadcast_weights(net)
        ph["broadcast"] = clock() - t0
        if step_idx % cfg.EVALUATE_EVERY_STEP == 0:
            t0 = clock()
            win_ratio = evaluate(game, net, best_net.target_model, rounds=cfg.EVALUATION_ROUNDS, device=device,
                                 seed=step_idx, reference_stores=reference_evaluate)
            if rank == 0 and log:
                log("Net evaluated, win ratio = %.2f" % win_ratio)
            writer.add_scalar("eval_win_ratio", win_ratio, step_idx)
            promoted = win_ratio > cfg.BEST_NET_WIN_RATIO
            hist["evaluations"].append((step_idx, win_ratio, promoted))
            if promoted:
                if rank == 0 and log:
                    log("Net is better than cur best, sync")
                best_net.sync()
                best_idx += 1
                hist["promotions"] = best_idx
                if rank == 0 and saves_path:
                    torch.save(net.state_dict(), os.path.join(saves_path, "best_%03d_%05d.dat" % (best_idx, step_idx)))
            ph["evaluate"] = clock() - t0
        if stop is not None:
            # the losses live on rank 0 only, so rank 0 decides and every rank hears it: a rank-local decision would
            # leave the others in the next collective
            flag = torch.tensor([1.0 if (rank == 0 and stop(hist)) else 0.0], dtype=torch.float64, device=device)
            parallel.allreduce_max(flag)
            if flag.item() > 0:
                break
    return hist


def main(argv=None):
    args = parse_args(argv)
    rank, local_rank, world = parallel.init()
    device = parallel.local_device(local_rank)
    saves_path = os.path.join(args.saves, args.name)
    if rank == 0:
        os.makedirs(saves_path, exist_ok=True)
    writer = _writer(args.name) if rank == 0 else _NullWriter()
    game = game_provider.get_game(args)
    net = Net(input_shape=game.obs_shape, actions_n=game.action_space).to(device)
    parallel.broadcast_weights(net)
    fit(game, net, device, args.games, iterations=args.iterations, saves_path=saves_path, writer=writer,
        reference_evaluate=True if args.reference_evaluate else False if args.sharded_evaluate else None, ddp=args.ddp,
        log=lambda m: print(m, flush=True),
        concurrent=args.concurrent or min(args.games, 1024), stream=not args.exact_self_play, net_mode=args.net_mode,
        streams=args.streams)
    writer.close()
    release_engines()  # (the self-play engines are kept between iterations: gigabytes of tree tables)


if __name__ == "__main__":
    main()
