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


def play_game(game, mcts_stores, replay_buffer: Union[collections.deque, None], net1, net2,
              steps_before_tau_0: int, mcts_searches: int, mcts_batch_size: int,
              net1_plays_first: bool = None, device: str = "cpu"):
    """One game; returns (net1_result in {+1, 0, -1}, step)."""
    assert isinstance(replay_buffer, (collections.deque, type(None)))
    assert isinstance(mcts_stores, (mcts.MCTS, type(None), list))
    assert isinstance(net1, model.Net)
    assert isinstance(net2, model.Net)
    assert isinstance(steps_before_tau_0, int) and steps_before_tau_0 >= 0
    assert isinstance(mcts_searches, int) and mcts_searches > 0
    assert isinstance(mcts_batch_size, int) and mcts_batch_size > 0

    if mcts_stores is None:
        mcts_stores = [mcts.MCTS(game), mcts.MCTS(game)]
    elif isinstance(mcts_stores, mcts.MCTS):
        mcts_stores = [mcts_stores, mcts_stores]

    state = game.initial_state
    nets = [net1, net2]
    if net1_plays_first is None:
        cur_player = int(np.random.choice(2))
    else:
        cur_player = 0 if net1_plays_first else 1
    step = 0
    tau = 1 if steps_before_tau_0 > 0 else 0
    history = []
    result = None
    net1_result = None

    while result is None:
        store = mcts_stores[cur_player]
        store.search_batch(mcts_searches, mcts_batch_size, state, cur_player, nets[cur_player], device=device)
        probs, _ = store.get_policy_value(state, tau=tau)
        history.append((state, cur_player, probs))
        action = int(np.random.choice(game.action_space, p=probs))
        if action not in game.possible_moves(state):
            print("Impossible action selected")
        state, won = game.move(state, action, cur_player)
        if won:
            result = 1
            net1_result = 1 if cur_player == 0 else -1
            break
        cur_player = 1 - cur_player
        if len(game.possible_moves(state)) == 0:
            result = 0
            net1_result = 0
            break
        step += 1
        if step >= steps_before_tau_0:
            tau = 0

    if replay_buffer is not None:
        for s, p, probs in reversed(history):
            replay_buffer.append((s, p, probs, result))
            result = -result
    return net1_result, step


def play_games(game, n_games, replay_buffer, net1, net2=None, steps_before_tau_0=10, mcts_searches=10,
               mcts_batch_size=8, n_stores=None, concurrent=None, seed=0, uid_base=0, device="cuda:0",
               first_player_mode=2, engine=None, return_stats=False):
    """Play `n_games` games on the HIP engine, `concurrent` at a time.

    net2 given -> arena: player 0 is net1, player 1 is net2, one tree per player (play.py:47 semantics,
    n_stores=2); otherwise self-play with one shared tree per game (train.py:43-47).
    Returns the list of net1 results (one per finished game, in finishing order); with return_stats=True
    also a dict with steps, counters and timing."""
    from caro_ai_amd.engine import SelfPlayEngine
    arena = net2 is not None and net2 is not net1
    if n_stores is None:
        n_stores = 2 if arena else 1
    G = int(concurrent or min(n_games, 1024))
    own = engine is None
    if own:
        engine = SelfPlayEngine(game, G, net1=net1, net2=net2 if arena else None, n_stores=n_stores,
                                max_batch=mcts_batch_size, steps_before_tau_0=steps_before_tau_0, seed=seed,
                                uid_base=uid_base, first_player_mode=first_player_mode, device=device,
                                searches_hint=mcts_searches)
    results, steps = [], []
    t0 = time.time()
    c0 = engine.counters()

    def consume(d):
        if replay_buffer is None:
            return
        states = game.from_keys(d["states"].cpu().numpy().view(np.uint64))
        players = d["players"].cpu().numpy().tolist()
        pis = d["pi"].cpu().numpy().tolist()
        zs = d["z"].cpu().numpy().tolist()
        for s, p, pi, z in zip(states, players, pis, zs):
            replay_buffer.append((s, p, pi, z))

    started = G
    while len(results) < n_games:
        engine.search(mcts_searches, mcts_batch_size)
        engine.step()
        # recycle finished slots only while more games are still to be started
        recycle = started < n_games
        d = engine.drain(recycle=recycle)
        ng = d["games"].shape[0]
        if ng:
            recs = d["games"].cpu().numpy()
            if recycle and started + ng > n_games:
                pass  # a few extra games may start; their results are simply not collected
            started += ng if recycle else 0
            results.extend(int(r) for r in recs[:, 2])
            steps.extend(int(r) for r in recs[:, 3])
            consume(d)
        if not recycle and engine.live_games() == 0:
            break
    results, steps = results[:n_games], steps[:n_games]
    if not return_stats:
        if own:
            engine.close()
        return results
    c1 = engine.counters()
    dt = time.time() - t0
    stats = {"steps": steps, "seconds": dt, "counters": {k: c1[k] - c0[k] for k in c1},
             "speed_nodes": (c1["expansions"] - c0["expansions"]) / dt, "speed_steps": sum(steps) / dt}
    if own:
        engine.close()
    return results, stats
