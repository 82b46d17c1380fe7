#!/usr/bin/env python3
"""Round-4 golden vectors, again by RUNNING THE REFERENCE (build container only; outputs committed, reference not):
the reference's PERSISTENT-STORE callers, which no earlier fixture exercises --

  persist_selfplay_c4.json.gz   3 consecutive self-play games sharing ONE `MCTS` store, the way train.py runs them
                                (ref train.py:184-193 creates one store, self_play :41-47 passes it to every
                                play_game): STEPS_BEFORE_TAU_0 = 10, MCTS_SEARCHES x MCTS_BATCH_SIZE = 10 x 8
                                (ref config.py), first mover drawn by play_game itself (net1_plays_first=None,
                                lib/utils.py:65-66)
  persist_evaluate_c4.json.gz   4 rounds of the loop of `evaluate` (ref train.py:134-141): ONE pair [MCTS, MCTS]
                                built before the loop and reused by every round, tau = 0, 20 x 16 sims, replay
                                buffer None, challenger != champion (two salted table nets)

Harness as make_golden.py: table net (priors / value = exact integer-hash functions of the planes, softmax
replaced by the identity), np.random.dirichlet / np.random.choice table-driven from include/caro_noise.h keyed
(seed, game uid, ply, sim); the `np.random.choice(2)` that draws the opener returns uid & 1 (recorded).
Every search_batch is traced: root board and player, root N / W / Q (+ which W are float32), len(store).

Usage:  python tests/golden/make_golden_r4.py
"""
import collections
import os
import sys
import time

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
import make_golden as mg  # noqa: E402  (puts /root/reference on sys.path and imports its lib)
from tests.synth_net import synth_numpy  # noqa: E402


class SaltedSynthNet(mg.ref_model.Net):
    """the reference's Net class with the table net's forward (tests/synth_net.py, oracle_synth_net: same bits)"""

    def __init__(self, game, salt):
        super().__init__(game.obs_shape, game.action_space)
        self.A, self.salt = game.action_space, salt

    def forward(self, x):
        P, v = synth_numpy(x.numpy(), self.A, self.salt)
        return torch.from_numpy(P), torch.from_numpy(v).reshape(-1, 1)


class Harness4(mg.Harness):
    """+ the opener draw `np.random.choice(2)` (-> uid & 1) and the root of every search in the trace"""

    def __enter__(self):
        super().__enter__()
        h = self
        inner_choice = np.random.choice
        inner_sb = mg.ref_mcts.MCTS.search_batch
        h.opener_draws = 0
        h.roots = []

        def choice(a, p=None):
            if p is None:
                assert a == 2
                h.opener_draws += 1
                return h.uid & 1
            return inner_choice(a, p=p)

        def search_batch(self_, count, batch_size, state_int, player, net, device="cpu"):
            h.roots.append((str(state_int), int(player)))
            return inner_sb(self_, count, batch_size, state_int, player, net, device)

        np.random.choice = choice
        mg.ref_mcts.MCTS.search_batch = search_batch
        return self


def play(game, stores, rb, net1, net2, sbt0, searches, batch, seed, uid):
    """one reference play_game on the CALLER's store(s), first mover left to play_game (None)"""
    with Harness4(game, seed, uid, True) as h, torch.no_grad():
        r, steps = mg.ref_utils.play_game(game, stores, rb, net1, net2, sbt0, searches, batch)
    assert h.opener_draws == 1
    trace = []
    for (s, p), t in zip(h.roots, h.trace):
        trace.append({"state": s, "player": p, "N": t["N"], "W": t["W"], "W_f32": t["W_f32"], "Q": t["Q"],
                      "nodes": t["nodes"]})
    return {"seed": seed, "uid": uid, "first_player": uid & 1, "result": int(r), "steps": int(steps),
            "plies": len(trace), "trace": trace}


def main():
    t0 = time.time()
    c4 = mg.ConnectFour()
    # (i) train.py's shared self-play store
    net = SaltedSynthNet(c4, 0)
    store = mg.ref_mcts.MCTS(c4)
    rb = collections.deque(maxlen=5000)
    games = []
    for i in range(3):
        n0 = len(rb)
        g = play(c4, store, rb, net, net, 10, 10, 8, 41, 5000 + i)
        new = list(rb)[n0:]  # appended last ply first (lib/utils.py:101-106)
        g["replay"] = {"states": [str(s) for s, _, _, _ in new], "players": [int(p) for _, p, _, _ in new],
                       "pi": [[float(x) for x in pr] for _, _, pr, _ in new], "z": [int(z) for _, _, _, z in new]}
        g["store_len_after"] = len(store)
        games.append(g)
        print("self-play game %d on the shared store: %d plies, result %d, store %d nodes, %.0f s"
              % (i, g["plies"], g["result"], len(store), time.time() - t0), flush=True)
    mg.dump("persist_selfplay_c4.json.gz", {"kind": "c4", "salts": [0, 0], "steps_before_tau_0": 10, "searches": 10,
                                            "batch": 8, "games": games})
    # (ii) evaluate's pair of stores
    challenger, champion = SaltedSynthNet(c4, 0x1111), SaltedSynthNet(c4, 0x2222)
    stores = [mg.ref_mcts.MCTS(c4), mg.ref_mcts.MCTS(c4)]
    rounds = []
    for i in range(4):
        g = play(c4, stores, None, challenger, champion, 0, 20, 16, 44, 6000 + i)
        g["store_len_after"] = [len(stores[0]), len(stores[1])]
        rounds.append(g)
        print("evaluate round %d on the persistent pair: %d plies, result %d, stores %s, %.0f s"
              % (i, g["plies"], g["result"], g["store_len_after"], time.time() - t0), flush=True)
    wins = sum(1 for g in rounds if g["result"] > 0.5)
    losses = sum(1 for g in rounds if g["result"] < -0.5)
    draws = sum(1 for g in rounds if g["result"] == 0)
    mg.dump("persist_evaluate_c4.json.gz", {"kind": "c4", "salts": [0x1111, 0x2222], "steps_before_tau_0": 0,
                                            "searches": 20, "batch": 16, "rounds": rounds,
                                            "win_ratio": wins / (wins + losses + draws)})
    print("done in %.1fs" % (time.time() - t0))


if __name__ == "__main__":
    main()
