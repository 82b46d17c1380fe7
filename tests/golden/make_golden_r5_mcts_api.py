#!/usr/bin/env python3
"""Round-5 golden vector of the reference's `MCTS` CLASS as an API (ref lib/mcts.py:21-313): a
script of calls on one store per game -- find_leaf on an empty and on a grown tree, search_minibatch, search_batch,
is_leaf, len(), the four dicts, get_policy_value at tau = 1 and tau = 0, a walk down the tree with the store kept, clear()
-- run on the REFERENCE in the build container with numpy's global generator seeded (the Dirichlet row of every descent
comes from it) and recorded call by call.  The net is the table net (priors and value exact dyadic functions of a hash of
the planes, tests/synth_net.py) with `F.softmax` replaced by the identity, as in make_golden.py's table-net games, so that
every number is exact on any machine.  tests/test_gpu_shim.py::test_mcts_class_follows_the_reference_call_by_call replays
the script on this package's MCTS (tree on the GPU) and compares every return value.

Usage:  python tests/golden/make_golden_r5_mcts_api.py
"""
import os
import sys

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
import make_golden as mg  # noqa: E402


def node(t, s):
    """the four dict rows of state s as plain numbers (+ which W / Q are numpy float32)"""
    return {"N": [int(x) for x in t.visit_count[s]], "W": [float(x) for x in t.value[s]],
            "Q": [float(x) for x in t.value_avg[s]], "P": [float(x) for x in t.probs[s]],
            "W_f32": [int(isinstance(x, np.float32)) for x in t.value[s]]}


def leaf(ret):
    value, leaf_state, player, states, actions = ret
    return {"value": None if value is None else float(value), "leaf": str(leaf_state), "player": int(player),
            "states": [str(x) for x in states], "actions": [int(a) for a in actions]}


def script(game, seed, B, plies):
    """the calls, in order; `log` is what the test replays and compares"""
    net = mg.SynthNet(game)
    t = mg.ref_mcts.MCTS(game)
    log = []
    np.random.seed(seed)
    s, p = game.initial_state, 0
    log.append(("find_leaf", str(s), p, leaf(t.find_leaf(s, p))))
    log.append(("len", len(t)))
    log.append(("is_leaf", str(s), bool(t.is_leaf(s))))
    for ply in range(plies):
        for k in range(3):
            t.search_minibatch(B, s, p, net)
            log.append(("search_minibatch", B, str(s), p, {"len": len(t), "node": node(t, s)}))
        log.append(("find_leaf", str(s), p, leaf(t.find_leaf(s, p))))
        t.search_batch(4, B, s, p, net)
        log.append(("search_batch", 4, B, str(s), p, {"len": len(t), "node": node(t, s)}))
        for tau in (1, 0):
            pi, q = t.get_policy_value(s, tau=tau)
            log.append(("get_policy_value", str(s), tau, [float(x) for x in pi], [float(x) for x in q]))
        log.append(("is_leaf", str(s), bool(t.is_leaf(s))))
        a = int(np.argmax(t.visit_count[s]))
        s2, won = game.move(s, a, p)
        log.append(("move", str(s), a, p, str(s2), bool(won)))
        if won or not game.possible_moves(s2):
            break
        s, p = s2, 1 - p
    keys = sorted(t.visit_count.keys())
    log.append(("keys", [str(k) for k in keys], int(sum(sum(t.visit_count[k]) for k in keys))))
    t.clear()
    log.append(("clear", len(t), bool(t.is_leaf(game.initial_state))))
    return log


def main():
    torch.set_num_threads(1)
    real = mg.ref_mcts.F.softmax
    mg.ref_mcts.F.softmax = lambda x, dim=1: x
    try:
        with torch.no_grad():
            out = {"c4": {"kind": "c4", "seed": 61, "batch": 8, "log": script(mg.ConnectFour(), 61, 8, 5)},
                   "ttt3": {"kind": "mnk", "n": 3, "k": 3, "seed": 62, "batch": 4, "log": script(mg.TicTacToe(), 62, 4, 6)},
                   "mnk5": {"kind": "mnk", "n": 5, "k": 4, "seed": 63, "batch": 2,
                            "log": script(mg.TicTacToe(5, 4), 63, 2, 4)}}
    finally:
        mg.ref_mcts.F.softmax = real
    for k, v in out.items():
        print(k, len(v["log"]), "calls; nodes at the end:", len(v["log"][-2][1]))
    mg.dump("mcts_api.json.gz", out)


if __name__ == "__main__":
    main()
