#!/usr/bin/env python3
"""Round-5 check, by RUNNING THE REFERENCE's own callers: `self_play` (ref train.py:24-59) and `evaluate`
(ref train.py:120-149) themselves -- not a restatement of their loops -- under the harness of make_golden_r4.py, and the
assertion that they do exactly what the round-4 fixtures recorded (persist_selfplay_c4 / persist_evaluate_c4 were made
by calling `play_game` in a hand-written copy of the two loops' call pattern).  train.py is imported as
make_golden_r5_train.py describes (never-called stand-in for the tensorboardX import; the globals `best_net`, `step_idx`,
`best_idx` that `__main__` sets are set on the module).  The one binding that is replaced is the module's
`play_game` name: a wrapper that opens the table-noise harness for the game (uid = base + call number) and then calls the
reference's play_game with the arguments it was given.

Output callers_check.json.gz: per call the result / steps / store lengths the REAL functions produced, the replay rows'
digest, evaluate's return value, the names the functions hand to their tracker.  tests/test_oracle_golden.py compares it
with the round-4 fixtures (which the oracle and the GPU shim reproduce bit for bit).

Usage:  python tests/golden/make_golden_r5_callers.py
"""
import collections
import hashlib
import os
import sys

import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
import make_golden as mg  # noqa: E402
import make_golden_r4 as r4  # noqa: E402
from make_golden_r5_train import Tracker, import_reference_train  # noqa: E402
from tests.conftest import load_golden  # noqa: E402


class Holder:
    """what ref train.py's `best_net` is to self_play: an object with a `target_model`"""

    def __init__(self, net):
        self.target_model = net


def rows_digest(rows):
    h = hashlib.sha256()
    for s, p, pi, z in rows:
        h.update(("%d|%d|%s|%d;" % (int(s), int(p), ",".join(repr(float(x)) for x in pi), int(z))).encode())
    return h.hexdigest()


def main():
    ref_train = import_reference_train()
    c4 = mg.ConnectFour()
    real_play_game = ref_train.play_game
    calls = []

    def harnessed(seed, uid_base):
        def play_game(*a, **k):
            uid = uid_base + len([c for c in calls if c["base"] == uid_base])
            with r4.Harness4(c4, seed, uid, True) as h, torch.no_grad():
                r, steps = real_play_game(*a, **k)
            assert h.opener_draws == 1
            calls.append({"base": uid_base, "uid": uid, "result": int(r), "steps": int(steps), "plies": len(h.trace),
                          "root_N": [t["N"] for t in h.trace], "nodes": [t["nodes"] for t in h.trace]})
            return r, steps
        return play_game

    out = {}
    # ---- self_play: one store for all games (ref train.py:184-193), PLAY_EPISODES = 1 game per call
    fx = load_golden("persist_selfplay_c4.json.gz")
    assert (ref_train.cfg.PLAY_EPISODES, ref_train.cfg.MCTS_SEARCHES, ref_train.cfg.MCTS_BATCH_SIZE,
            ref_train.cfg.STEPS_BEFORE_TAU_0) == (1, fx["searches"], fx["batch"], fx["steps_before_tau_0"])
    net = r4.SaltedSynthNet(c4, fx["salts"][0])
    ref_train.best_net, ref_train.best_idx = Holder(net), 0
    ref_train.play_game = harnessed(fx["games"][0]["seed"], fx["games"][0]["uid"])
    store = mg.ref_mcts.MCTS(c4)
    rb = collections.deque(maxlen=ref_train.cfg.REPLAY_BUFFER)
    tracker = Tracker()
    sp = []
    for i, want in enumerate(fx["games"]):
        ref_train.step_idx = i
        n0 = len(rb)
        ref_train.self_play(c4, store, rb, net, tracker, "cpu")
        c = calls[-1]
        new = list(rb)[n0:]
        assert (c["uid"], c["result"], c["steps"], c["plies"]) == (want["uid"], want["result"], want["steps"], want["plies"])
        assert c["root_N"] == [t["N"] for t in want["trace"]] and c["nodes"] == [t["nodes"] for t in want["trace"]]
        assert len(store) == want["store_len_after"]
        assert [str(s) for s, _, _, _ in new] == want["replay"]["states"] and [z for _, _, _, z in new] == want["replay"]["z"]
        assert [[float(x) for x in pr] for _, _, pr, _ in new] == want["replay"]["pi"]
        sp.append({"uid": c["uid"], "result": c["result"], "steps": c["steps"], "store_len_after": len(store),
                   "replay_rows": len(new), "replay_sha256": rows_digest(new)})
    out["self_play"] = {"calls": sp, "tracked": sorted(tracker.seen)}
    print("\nself_play: the reference's function reproduces persist_selfplay_c4.json.gz", sp)
    # ---- evaluate: its own pair of stores, built inside the function and reused by every round
    fx = load_golden("persist_evaluate_c4.json.gz")
    ch, cp = r4.SaltedSynthNet(c4, fx["salts"][0]), r4.SaltedSynthNet(c4, fx["salts"][1])
    n_before = len(calls)
    ref_train.play_game = harnessed(fx["rounds"][0]["seed"], fx["rounds"][0]["uid"])
    ratio = ref_train.evaluate(c4, ch, cp, rounds=len(fx["rounds"]), device="cpu")
    ev = calls[n_before:]
    for c, want in zip(ev, fx["rounds"]):
        assert (c["uid"], c["result"], c["steps"], c["plies"]) == (want["uid"], want["result"], want["steps"], want["plies"])
        assert c["root_N"] == [t["N"] for t in want["trace"]] and c["nodes"] == [t["nodes"] for t in want["trace"]]
    assert ratio == fx["win_ratio"]
    out["evaluate"] = {"rounds": [{"uid": c["uid"], "result": c["result"], "steps": c["steps"]} for c in ev],
                       "win_ratio": ratio}
    print("evaluate: the reference's function reproduces persist_evaluate_c4.json.gz, win ratio", ratio)
    ref_train.play_game = real_play_game
    mg.dump("callers_check.json.gz", out)


if __name__ == "__main__":
    main()
