"""Game-loop drivers of the self-play path.

`play_game` keeps the reference's signature, return value, replay-buffer record
format and numpy random-stream consumption (lib/utils.py:25-108): it is the
drop-in for train.py:43-47, train.py:139-142 and play.py:47-52 and plays ONE
game through `lib.mcts.MCTS` (tree and search on the GPU).

`play_games` is the same loop for N games at once on the HIP engine (all games
advance in lock-step, leaves of every game share one net batch); it fills the
same `collections.deque` with the same `(state, player, pi, z)` tuples.  Its
random inputs come from the counter-based spec in include/caro_noise.h.
"""
import collections
import time
from typing import Union

import numpy as np

from caro_ai_amd.lib import mcts, model


def update_counts(counts_dict, key, counts):
    """W/L/D bookkeeping of play.py (reference lib/utils.py:9-22)."""
    v = counts_dict.get(key, (0, 0, 0))
    counts_dict[key] = (v[0] + counts[0], v[1] + counts[1], v[2] + counts[2])


class TBMeanTracker:
    """The metric tracker train.py wraps its loop in (`with TBMeanTracker(writer, batch_size=10) as tb_tracker`,
    reference lib/utils.py:111-159): `track(name, value, step)` collects values per name and, every `batch_size`-th
    value of a name, writes their mean to `writer.add_scalar(name, mean, step)`; leaving the block closes the writer.
    Values may be numbers, numpy arrays / scalars or tensors (their mean is what counts)."""

    def __init__(self, writer, batch_size):
        assert isinstance(batch_size, int)
        assert writer is not None
        self.writer = writer
        self.batch_size = batch_size

    def __enter__(self):
        self._batches = collections.defaultdict(list)
        return self

    def __exit__(self, exc_type, exc_val, exc_tb):
        self.writer.close()

    @staticmethod
    def _as_float(value):
        import torch
        assert isinstance(value, (float, int, np.ndarray, np.generic)) or torch.is_tensor(value)
        if torch.is_tensor(value):
            return value.detach().float().mean().item()
        return float(np.mean(value))

    def track(self, param_name, value, iter_index):
        assert isinstance(param_name, str)
        assert isinstance(iter_index, int)
        pending = self._batches[param_name]
        pending.append(self._as_float(value))
        if len(pending) >= self.batch_size:
            self.writer.add_scalar(param_name, np.mean(pending), iter_index)
            pending.clear()


def _first_mover(net1_plays_first):
    """player index (0 = net1) that opens the game; `None` draws it from numpy's global stream (utils.py:65-68)"""
    if net1_plays_first is None:
        return int(np.random.choice(2))
    return 0 if net1_plays_first else 1


def play_game(game, mcts_stores, replay_buffer: Union[collections.deque, None], net1, net2,
              steps_before_tau_0: int, mcts_searches: int, mcts_batch_size: int,
              net1_plays_first: bool = None, device: str = "cpu"):
    """One game; returns (net1_result in {+1, 0, -1}, step).  Interface, assertions, numpy draws (one
    `choice(2)` when the first mover is open, one `choice(A, p=pi)` per ply, one Dirichlet row per descent inside
    `MCTS`) and the replay records are those of the reference's lib/utils.py:25-108."""
    assert isinstance(replay_buffer, (collections.deque, type(None)))
    assert isinstance(mcts_stores, (mcts.MCTS, type(None), list))
    assert isinstance(net1, model.Net)
    assert isinstance(net2, model.Net)
    assert isinstance(steps_before_tau_0, int) and steps_before_tau_0 >= 0
    assert isinstance(mcts_searches, int) and mcts_searches > 0
    assert isinstance(mcts_batch_size, int) and mcts_batch_size > 0

    # one tree per player (none given), one shared tree (a single MCTS), or the caller's pair
    if isinstance(mcts_stores, mcts.MCTS):
        trees = (mcts_stores, mcts_stores)
    else:
        trees = tuple(mcts_stores) if mcts_stores is not None else (mcts.MCTS(game), mcts.MCTS(game))
    brains = (net1, net2)
    mover = _first_mover(net1_plays_first)
    state, step = game.initial_state, 0
    plies = []       # (state, mover, pi) in playing order
    outcome = None   # for the player who made the last move: 1 = won, 0 = board full
    while outcome is None:
        tau = 1 if step < steps_before_tau_0 else 0  # tau = 1 for the first `steps_before_tau_0` plies
        tree = trees[mover]
        tree.search_batch(mcts_searches, mcts_batch_size, state, mover, brains[mover], device=device)
        pi, _ = tree.get_policy_value(state, tau=tau)
        plies.append((state, mover, pi))
        move = int(np.random.choice(game.action_space, p=pi))  # drawn at tau = 0 too, as the reference does
        if move not in game.possible_moves(state):
            print("Impossible action selected")
        state, won = game.move(state, move, mover)
        if won:
            outcome = 1
        elif not game.possible_moves(state):
            outcome = 0
        else:
            mover, step = 1 - mover, step + 1
    net1_result = 0 if outcome == 0 else (1 if mover == 0 else -1)
    if replay_buffer is not None:
        z = outcome  # seen by the last mover; alternates back through the game
        for s, who, pi in reversed(plies):
            replay_buffer.append((s, who, pi, z))
            z = -z
    return net1_result, step


def play_games(game, n_games, replay_buffer, net1, net2=None, steps_before_tau_0=10, mcts_searches=10,
               mcts_batch_size=8, n_stores=None, concurrent=None, seed=0, uid_base=0, device="cuda:0",
               first_player_mode=2, return_stats=False, node_cap=None):
    """Play the `n_games` games with uids uid_base .. uid_base + n_games - 1 on the HIP engine, `concurrent` at a time.

    net2 given -> arena: player 0 is net1, player 1 is net2, one tree per player (play.py:47 semantics,
    n_stores=2); otherwise self-play with one shared tree per game (train.py:43-47).
    Returns the list of net1 results ordered by uid (which games are played, and how each one goes, depends on
    the uids and the seed only -- not on `concurrent`); with return_stats=True also a dict with steps, counters
    and timing.  Exactly the wanted games are played (the engine's games_limit: a slot whose next uid lies beyond the
    range stays finished).  Raises CaroError if a tree overflowed its node pool."""
    from caro_ai_amd import _lib
    from caro_ai_amd.engine import SelfPlayEngine
    arena = net2 is not None and net2 is not net1
    if n_stores is None:
        n_stores = 2 if arena else 1
    G = int(concurrent or min(n_games, 1024))
    G = max(1, min(G, n_games))
    # boards whose per-game node bound (searches x batch x cells) is beyond a default tree: unreachable nodes are dropped
    # after every move (result-neutral), the default cap then bounds the LIVE nodes -- and an overflow raises, below
    hw = game.obs_shape[1] * game.obs_shape[2]
    evict = mcts_searches * mcts_batch_size * hw + 64 > SelfPlayEngine.DEFAULT_CAP_LIMIT
    # One generation of games on a geometry with whole wavefronts per game (connect four with batch 8: play.py's arena;
    # 15 x 15 with batch 8; ...): the engine's staggered mode without restarts -- every game on its own minibatch clock, the same games, evener
    # launches.  Several generations keep the lock-step engine, whose drain restarts the slots.
    from caro_ai_amd.engine import staggered_geometry
    stagger = G == n_games and staggered_geometry(game, mcts_batch_size, evict)
    # slot g plays uids uid_base + g, + G, + 2G, ... while they lie inside the wanted range (games_limit): no game
    # beyond it is ever started
    engine = SelfPlayEngine(game, G, net1=net1, net2=net2 if arena else None, n_stores=n_stores,
                            max_batch=mcts_batch_size, steps_before_tau_0=steps_before_tau_0, seed=seed,
                            uid_base=uid_base, first_player_mode=first_player_mode, device=device,
                            searches_hint=mcts_searches, stagger=stagger, stagger_recycle=False, evict=evict,
                            games_limit=n_games, node_cap=node_cap)
    try:
        t0 = time.time()
        outcome = {}  # uid -> (net1 result, steps)

        def consume(d):
            """tuples of the drained games -> the caller's deque, in the reference's record format (whole arrays at
            once: PackedGame.from_keys, tolist)"""
            states = game.from_keys(d["states"].cpu().numpy().view(np.uint64))
            players = d["players"].cpu().numpy().tolist()
            pis = d["pi"].cpu().numpy().tolist()
            zs = d["z"].cpu().numpy().tolist()
            replay_buffer.extend(zip(states, players, pis, zs))

        # every pass is one ply of every live game (staggered: on average, after at most `searches` launches of waiting):
        # a bound no healthy run reaches -- a run whose trees overflow can stop making plies (a root whose expansion was
        # dropped has no visits, its ply is refused) and must end in the error below, not spin
        moves_left = (hw + 4) * (-(-n_games // G)) + mcts_searches + 8
        while len(outcome) < n_games and moves_left > 0:
            moves_left -= 1
            engine.search(mcts_searches, mcts_batch_size)
            engine.step()
            d = engine.drain(recycle=not stagger)
            if int(d["games"].shape[0]):
                for uid, _first, result, steps in d["games"].cpu().numpy().tolist():
                    assert uid_base <= uid < uid_base + n_games and uid not in outcome, uid
                    outcome[uid] = (int(result), int(steps))
                if replay_buffer is not None:
                    consume(d)
            elif engine.live_games() == 0:
                break
        c1 = engine.counters()
        dt = time.time() - t0
    finally:
        engine.close()
    if c1["overflows"]:
        raise _lib.CaroError("play_games: %d minibatches overflowed the node pool (node_cap=%d, eviction %s) or plies were "
                             "refused on a root without visits: the games are not the reference's"
                             % (c1["overflows"], engine.cfg.node_cap, "on" if evict else "off"))
    if len(outcome) != n_games:
        raise _lib.CaroError("play_games: %d of %d games finished" % (len(outcome), n_games))
    results = [outcome[u][0] for u in sorted(outcome)]
    steps = [outcome[u][1] for u in sorted(outcome)]
    if not return_stats:
        return results
    stats = {"steps": steps, "seconds": dt, "counters": dict(c1),
             "speed_nodes": c1["expansions"] / dt, "speed_steps": sum(steps) / dt}
    return results, stats
