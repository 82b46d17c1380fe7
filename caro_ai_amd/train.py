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


def staggered_ok(game, batch):
    """the geometry staggered mode needs: one wavefront per game (batch x lanes per descent = 64)"""
    A = game.action_space
    lpd = 8 if A == 7 else 16 if A <= 16 else 32 if A <= 32 else 64
    return batch * lpd == 64


def _self_play_loop(eng, game, n_games, G, searches, batch, stagger, restarts, st, slot_gen, last_gen, take):
    """the moves of self_play until the wanted games have finished (separate so that self_play can close its engine,
    and end the rank, on any failure)"""
    if stagger:
        # every pass of `searches` launches is one ply per game on average; a game has at most HW plies and sits out
        # fewer than `searches` launches at the start: a bound on the passes that a healthy run never reaches
        hw = game.obs_shape[1] * game.obs_shape[2]
        max_passes = (hw + 4) * (-(-n_games // G)) + 8
        passes = 0
        while st["finished"] < n_games:
            take(eng.move(searches, batch, recycle=restarts))  # host-pipelined: hands out the previous pass's rows
            passes += 1
            if passes > max_passes:
                raise _lib.CaroError("self_play: %d of %d games finished after %d passes" % (st["finished"], n_games, passes))
        take(eng.flush())
    else:
        while st["finished"] < n_games:
            eng.search(searches, batch)
            eng.step()
            # restart drained slots while some slot still has a wanted generation to begin
            d = eng.drain(recycle=bool(restarts and (slot_gen < last_gen).any()))
            ng = int(d["games"].shape[0])
            take(d)
            if not ng and eng.live_games() == 0:
                break


def self_play(game, replay_buffer, net, n_games, device="cuda:0", seed=0, uid_base=0, searches=cfg.MCTS_SEARCHES,
              batch=cfg.MCTS_BATCH_SIZE, concurrent=None, stagger=False):
    """Play n_games (per rank) with the (best) net against itself, tuples appended on the device.
    Returns speed_steps, speed_nodes, steps, nodes (train.py:49-58).

    WHICH games: slot g of this rank plays uids uid_base + rank*G + g (+ k * world * G for its k-th restart), and the
    games wanted are the first n_games of that sequence (local index k*G + g < n_games) -- the same set whether the
    engine runs lock-step or staggered, whatever finishes first.  With G == n_games (the default) every slot plays
    exactly its own uid to the end and nothing restarts.  With fewer slots than games the slots restart; slots that
    run ahead may start games beyond the wanted set while the last wanted ones finish: those are played but their
    tuples are DROPPED, so the replay buffer never holds a length-biased "first to finish" sample.
    stagger=True (the CLI's choice where the geometry allows): the engine's staggered mode -- every game on its own
    minibatch clock; each game is the one the lock-step form plays for the same uid, only the ORDER in which games
    reach the replay buffer differs."""
    from caro_ai_amd.engine import SelfPlayEngine
    rank, _, world = parallel.env_rank() if parallel.is_dist() else (0, 0, 1)
    G = max(1, min(int(concurrent or n_games), int(n_games)))
    stagger = bool(stagger) and staggered_ok(game, batch)
    restarts = n_games > G
    base, stride = uid_base + rank * G, world * G
    eng = SelfPlayEngine(game, G, net1=net, max_batch=batch, steps_before_tau_0=cfg.STEPS_BEFORE_TAU_0, seed=seed,
                         device=device, searches_hint=searches, uid_base=base, uid_stride=stride,
                         stagger=stagger, stagger_recycle=restarts)
    t0 = time.time()
    st = {"finished": 0, "steps": 0, "dropped": 0}
    slot_gen = np.zeros(G, dtype=np.int64)                  # generation each slot is playing (lock-step bookkeeping)
    last_gen = (n_games - 1 - np.arange(G)) // G             # last wanted generation of each slot
    gatherer = parallel.TupleGatherer(every=1 << 30, pi_dtype=torch.float32)

    def take(d):
        """keep the tuples of the wanted games of one drain"""
        if d is None or not int(d["games"].shape[0]):
            return
        recs = d["games"]
        off = recs[:, 0] - base
        k, g = off // stride, off % stride
        want = (off >= 0) & (g < G) & (k * G + g < n_games)
        nwant = int(want.sum().item())
        st["finished"] += nwant
        st["dropped"] += int(recs.shape[0]) - nwant
        st["steps"] += int(recs[want, 3].sum().item())
        np.maximum.at(slot_gen, g[want].cpu().numpy(), k[want].cpu().numpy() + 1)
        if nwant == int(recs.shape[0]):
            gatherer.push(d)
        elif nwant:
            keep = torch.repeat_interleave(want, recs[:, 3] + 1)  # a game of s steps holds s + 1 rows
            gatherer.push({f: d[f][keep] for f in ("states", "players", "pi", "z")})

    try:
        _self_play_loop(eng, game, n_games, G, searches, batch, stagger, restarts, st, slot_gen, last_gen, take)
    except BaseException:
        # the engine goes whatever happens (its trees are gigabytes).  Under several ranks the error must END this rank:
        # the peers are on their way to the collective in gatherer.flush() and would wait there for the backend's
        # timeout; a rank that exits non-zero is what the launcher's fail-fast path (bench.py / torchrun) acts on.
        eng.close()
        if parallel.is_dist():
            import traceback
            traceback.print_exc()
            sys.stderr.flush()
            os._exit(13)
        raise
    # multi-GPU: the loop above is driven by rank-local counts, so the exchange is ONE collective at the end, when every
    # rank has left its loop
    out = gatherer.flush()
    if out is not None:
        replay_buffer.extend(out)
    c = eng.counters()
    dt = time.time() - t0
    eng.close()
    return {"speed_steps": st["steps"] / dt, "speed_nodes": c["expansions"] / dt, "steps": st["steps"],
            "nodes": c["expansions"], "games": st["finished"], "games_dropped": st["dropped"]}


def evaluate(game, challenger, champion, rounds=cfg.EVALUATION_ROUNDS, device="cuda:0", seed=0,
             reference_stores=False, counts=False):
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
    rounds recorded from the reference: tests/test_gpu_shim.py::test_evaluate_with_reference_stores_*."""
    rank, _, world = parallel.env_rank() if parallel.is_dist() else (0, 0, 1)
    if reference_stores:
        from caro_ai_amd.lib import mcts as mcts_mod
        from caro_ai_amd.lib.utils import play_game
        res = []
        if rank == 0:
            stores = [mcts_mod.MCTS(game, tree_device=device), mcts_mod.MCTS(game, tree_device=device)]
            for _ in range(rounds):
                r, _ = play_game(game, stores, None, challenger, champion, steps_before_tau_0=0, mcts_searches=20,
                                 mcts_batch_size=16, device=device)
                res.append(r)
        wins, losses, draws = parallel.allreduce_counts((res.count(1), res.count(-1), res.count(0)), device)
        ratio = wins / max(1, wins + losses + draws)
        return (ratio, (wins, losses, draws)) if counts else ratio
    from caro_ai_amd.lib.utils import play_games
    lo, n = parallel.shard_rounds(rounds, rank, world)
    res = []
    if n:
        res = play_games(game, n, None, challenger, champion, steps_before_tau_0=0, mcts_searches=20,
                         mcts_batch_size=16, concurrent=n, seed=seed, uid_base=lo, device=device)
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
    p.add_argument("--iterations", type=int, default=0, help="stop after this many iterations (0 = run for ever)")
    p.add_argument("--saves", default="saves")
    p.add_argument("--reference-evaluate", action="store_true",
                   help="arena gate with the reference's store semantics: one pair of MCTS stores reused by all "
                        "rounds, rounds played one after another (default: independent rounds, concurrent, sharded)")
    p.add_argument("--ddp", action="store_true",
                   help="several ranks: every rank trains on its share of each batch, gradients all-reduced "
                        "(default: rank 0 trains, the weights are broadcast)")
    return p.parse_args(argv)


def fit(game, net, device, games, iterations=0, saves_path=None, writer=None, reference_evaluate=False, ddp=False,
        sample_seed=None, stop=None, log=print):
    """The reference's training loop (train.py:165-217): self-play with the best net -> replay buffer -> TRAIN_ROUNDS SGD
    steps -> every EVALUATE_EVERY_STEP iterations the arena gate (challenger = the net being trained against the best
    net; promoted when its win ratio exceeds BEST_NET_WIN_RATIO: `NetWrapper.sync`, `best_%03d_%05d.dat`).
    `games` self-play games per iteration (reference: PLAY_EPISODES = 1), `iterations` 0 = for ever.
    sample_seed: seed of the replay sampling (None: torch's global generator, as the reference); stop(history) -> True
    ends the loop early.  Returns the history: per trained iteration the three losses, per evaluation (iteration, win
    ratio, promoted), the number of promotions, the best net wrapper."""
    rank, _, world = parallel.env_rank() if parallel.is_dist() else (0, 0, 1)
    writer = writer or _NullWriter()
    best_net = NetWrapper(net)
    optimizer = optim.SGD(net.parameters(), lr=cfg.LEARNING_RATE, momentum=0.9)
    replay_buffer = DeviceReplayBuffer(game, cfg.REPLAY_BUFFER, device)
    hist = {"loss_total": [], "loss_value": [], "loss_policy": [], "evaluations": [], "promotions": 0,
            "best_net": best_net, "speed_nodes": [], "iterations": 0}
    step_idx = best_idx = 0
    while iterations == 0 or step_idx < iterations:
        sp = self_play(game, replay_buffer, best_net.target_model, games, device=device, seed=step_idx,
                       uid_base=step_idx * games * world, stagger=True)
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
        # (ddp: the parameters are already identical; the batch-norm running statistics are each rank's own: rank 0's go out)
        parallel.broadcast_weights(net)
        if step_idx % cfg.EVALUATE_EVERY_STEP == 0:
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
        if stop is not None and stop(hist):
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
        reference_evaluate=args.reference_evaluate, ddp=args.ddp, log=lambda m: print(m, flush=True))
    writer.close()


if __name__ == "__main__":
    main()
