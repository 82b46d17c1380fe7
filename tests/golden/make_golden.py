#!/usr/bin/env python3
"""Generate the golden vectors under tests/golden/ by RUNNING THE REFERENCE.

Runs only in the build container (needs /root/reference); the outputs are
committed, the reference is not.  Usage:  python tests/golden/make_golden.py

Harness rules:
  * net.eval() + torch.no_grad() + torch.set_num_threads(1)      (declared deviation Q9)
  * fresh MCTS stores per game                                    (Q3)
  * net1_plays_first always given
  * np.random.dirichlet / np.random.choice are replaced for the duration of a
    game by table-driven versions: noise row = include/caro_noise.h row keyed
    (seed, game uid, ply, sim); move = inverse CDF of caro_move_uniform(seed,
    uid, ply) (numpy's own legacy-choice algorithm).
  * "synth" games additionally replace F.softmax inside lib.mcts by the
    identity and use a fake Net that returns the synthetic hash net's P and v
    (exact dyadic float32), so the search is isolated from conv numerics.
"""
import gzip
import hashlib
import json
import os
import sys
import time

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.abspath(os.path.join(HERE, "..", ".."))
REF = "/root/reference"
sys.dont_write_bytecode = True
sys.path.insert(0, REF)
sys.path.insert(0, ROOT)

from lib import mcts as ref_mcts, utils as ref_utils, model as ref_model  # noqa: E402
from lib.game.connect_four.connect_four import ConnectFour  # noqa: E402
from lib.game.tictactoe.tictactoe import TicTacToe  # noqa: E402
from oracle import oracle as orc  # noqa: E402  (host form of the noise spec)

torch.set_num_threads(1)
MASK = (1 << 64) - 1


def mix64(z):
    z &= MASK
    z ^= z >> 30
    z = (z * 0xbf58476d1ce4e5b9) & MASK
    z ^= z >> 27
    z = (z * 0x94d049bb133111eb) & MASK
    z ^= z >> 31
    return z


def synth_eval(planes, A):
    """numpy twin of oracle_synth_net / tests/synth_net.py"""
    L = planes.shape[0]
    flat = planes.reshape(L, -1) != 0
    coef = np.array([mix64(0x5851f42d4c957f2d + j) | 1 for j in range(flat.shape[1])], dtype=np.uint64)
    P = np.zeros((L, A), np.float32)
    v = np.zeros(L, np.float32)
    for i in range(L):
        h = int(coef[flat[i]].sum(dtype=np.uint64)) & MASK if flat[i].any() else 0
        for a in range(A):
            ha = mix64(h + 0x9E3779B97F4A7C15 * (a + 1))
            P[i, a] = np.float32(((ha >> 20) & 1023) + 1) / np.float32(8192.0)
        hv = mix64(h ^ 0xA5A5A5A5A5A5A5A5)
        v[i] = np.float32(int((hv >> 20) % 2001) - 1000) / np.float32(1024.0)
    return P, v


class SynthNet(ref_model.Net):
    def __init__(self, game):
        super().__init__(game.obs_shape, game.action_space)
        self.A = game.action_space

    def forward(self, x):
        P, v = synth_eval(x.numpy(), self.A)
        return torch.from_numpy(P), torch.from_numpy(v).reshape(-1, 1)


class Harness:
    """Context manager patching the reference's random inputs for one game."""

    def __init__(self, game, seed, uid, identity_softmax):
        self.A = game.action_space
        self.seed, self.uid = seed, uid
        self.identity_softmax = identity_softmax
        self.ply = -1
        self.sim = -1
        self.noise_log = []
        self.uniform_log = []
        self.trace = []  # per ply: root N, store len

    def __enter__(self):
        h = self
        self._dir, self._choice = np.random.dirichlet, np.random.choice
        self._softmax = ref_mcts.F.softmax
        self._find_leaf = ref_mcts.MCTS.find_leaf
        self._search_batch = ref_mcts.MCTS.search_batch

        def dirichlet(alpha):
            row = orc.noise_row(h.seed, h.uid, h.ply, h.sim, len(alpha), alpha[0])
            h.noise_log.append(row)
            return row

        def choice(a, p=None):
            u = orc.move_uniform(h.seed, h.uid, h.ply)
            h.uniform_log.append(u)
            cdf = np.cumsum(np.asarray(p, dtype=np.float64))
            cdf /= cdf[-1]
            return int(np.searchsorted(cdf, u, side="right"))

        def find_leaf(self_, state_int, player):
            h.sim += 1
            return h._find_leaf(self_, state_int, player)

        def search_batch(self_, count, batch_size, state_int, player, net, device="cpu"):
            h.ply += 1
            h.sim = -1
            r = h._search_batch(self_, count, batch_size, state_int, player, net, device)
            h.trace.append({"N": list(map(int, self_.visit_count[state_int])),
                            "W": [float(x) for x in self_.value[state_int]],
                            "W_f32": [int(isinstance(x, np.float32)) for x in self_.value[state_int]],
                            "Q": [float(x) for x in self_.value_avg[state_int]],
                            "nodes": len(self_)})
            return r

        np.random.dirichlet, np.random.choice = dirichlet, choice
        ref_mcts.MCTS.find_leaf, ref_mcts.MCTS.search_batch = find_leaf, search_batch
        if self.identity_softmax:
            ref_mcts.F.softmax = lambda x, dim=1: x
        return self

    def __exit__(self, *exc):
        np.random.dirichlet, np.random.choice = self._dir, self._choice
        ref_mcts.MCTS.find_leaf, ref_mcts.MCTS.search_batch = self._find_leaf, self._search_batch
        ref_mcts.F.softmax = self._softmax


def play_reference(game, net1, net2, n_stores, sbt0, searches, batch, first_player, seed, uid, synth):
    import collections
    rb = collections.deque()
    stores = None if n_stores == 2 else ref_mcts.MCTS(game)
    with Harness(game, seed, uid, synth) as h, torch.no_grad():
        r, steps = ref_utils.play_game(game, stores, rb, net1, net2, sbt0, searches, batch,
                                       net1_plays_first=(first_player == 0))
    hist = list(rb)[::-1]  # forward ply order
    return {
        "seed": seed, "uid": uid, "n_stores": n_stores, "steps_before_tau_0": sbt0,
        "searches": searches, "batch": batch, "first_player": first_player,
        "result": int(r), "steps": int(steps), "plies": len(hist),
        "states": [str(s) for s, _, _, _ in hist],
        "players": [int(p) for _, p, _, _ in hist],
        "pi": [[float(x) for x in pr] for _, _, pr, _ in hist],
        "z": [int(z) for _, _, _, z in hist],
        "trace": h.trace,
        "noise_rows": len(h.noise_log),
        "noise_sha1": hashlib.sha1(np.asarray(h.noise_log, np.float64).tobytes()).hexdigest(),
        "_noise": h.noise_log, "_uniform": h.uniform_log,
    }


def strip(g, keep_tables):
    g = dict(g)
    noise, uni = g.pop("_noise"), g.pop("_uniform")
    if keep_tables:
        g["noise_table"] = [[float(x) for x in r] for r in noise]
        g["uniform_table"] = [float(u) for u in uni]
    return g


def rules_vectors(game, n_games, rng, max_plies=10**9):
    """Random playouts through the reference's rules."""
    recs = []
    for _ in range(n_games):
        s = game.initial_state
        p = int(rng.integers(2))
        for _ply in range(max_plies):
            legal = game.possible_moves(s)
            if not legal:
                break
            mv = int(legal[int(rng.integers(len(legal)))])
            s2, won = game.move(s, mv, p)
            planes = game.states_to_training_batch([s2], [1 - p])[0]
            recs.append({"s": str(s), "m": mv, "p": p, "s2": str(s2), "won": bool(won),
                         "legal": [int(x) for x in legal],
                         "planes": np.packbits(planes.astype(np.uint8).reshape(-1)).tobytes().hex()})
            s, p = s2, 1 - p
            if won:
                break
    return recs


def dump(name, obj):
    path = os.path.join(HERE, name)
    with gzip.open(path, "wt", compresslevel=9) as f:
        json.dump(obj, f, separators=(",", ":"))
    print("wrote", name, os.path.getsize(path), "bytes")


def load_net(game, path):
    net = ref_model.Net(game.obs_shape, game.action_space)
    net.load_state_dict(torch.load(path, map_location="cpu"))
    net.eval()
    return net


def main():
    t0 = time.time()
    rng = np.random.default_rng(20261003)
    c4, ttt = ConnectFour(), TicTacToe()
    g55, g15 = TicTacToe(5, 4), TicTacToe(15, 5)

    # ---- G1: rules ----
    dump("rules_c4.json.gz", {"kind": "c4", "recs": rules_vectors(c4, 120, rng)})
    dump("rules_ttt3.json.gz", {"kind": "mnk", "n": 3, "k": 3, "recs": rules_vectors(ttt, 150, rng)})
    dump("rules_mnk5.json.gz", {"kind": "mnk", "n": 5, "k": 4, "recs": rules_vectors(g55, 40, rng)})
    dump("rules_mnk15.json.gz", {"kind": "mnk", "n": 15, "k": 5, "recs": rules_vectors(g15, 6, rng)})

    # ---- G2: tree walk, synthetic table net ----
    games = []
    for i, (fp, sbt0) in enumerate([(0, 10), (1, 10), (0, 0), (1, 3)]):
        g = play_reference(c4, SynthNet(c4), SynthNet(c4), 1, sbt0, 25, 8, fp, 7, 100 + i, True)
        games.append(strip(g, keep_tables=(i == 0)))
    # arena-shaped: two stores, batch 16 (train.evaluate's 20x16)
    g = play_reference(c4, SynthNet(c4), SynthNet(c4), 2, 0, 20, 16, 0, 7, 150, True)
    games.append(strip(g, False))
    dump("synth_c4.json.gz", {"kind": "c4", "games": games})

    games = []
    for i, (fp, sbt0, s, b) in enumerate([(0, 10, 25, 1), (1, 10, 25, 1), (0, 2, 10, 8), (1, 0, 25, 4)]):
        g = play_reference(ttt, SynthNet(ttt), SynthNet(ttt), 1, sbt0, s, b, fp, 11, 200 + i, True)
        games.append(strip(g, keep_tables=(i == 0)))
    g = play_reference(ttt, SynthNet(ttt), SynthNet(ttt), 2, 0, 10, 8, 1, 11, 250, True)
    games.append(strip(g, False))
    dump("synth_ttt3.json.gz", {"kind": "mnk", "n": 3, "k": 3, "games": games})

    games = [strip(play_reference(g55, SynthNet(g55), SynthNet(g55), 1, 4, 12, 8, i & 1, 13, 300 + i, True), False)
             for i in range(2)]
    dump("synth_mnk5.json.gz", {"kind": "mnk", "n": 5, "k": 4, "games": games})

    games = [strip(play_reference(g15, SynthNet(g15), SynthNet(g15), 1, 6, 6, 8, 0, 17, 400, True), False)]
    dump("synth_mnk15.json.gz", {"kind": "mnk", "n": 15, "k": 5, "games": games})

    # ---- G3: end to end, shipped weights (CPU float32, eval mode) ----
    w26 = os.path.join(REF, "saves/trained_connect4/best_026_12000.dat")
    w25 = os.path.join(REF, "saves/trained_connect4/best_025_10600.dat")
    wt5 = os.path.join(REF, "saves/trained_tictactoe/best_005_00900.dat")
    n26, n25, nt5 = load_net(c4, w26), load_net(c4, w25), load_net(ttt, wt5)
    games = [strip(play_reference(c4, n26, n26, 1, 10, 25, 8, i & 1, 23, 500 + i, False), False) for i in range(3)]
    dump("real_c4.json.gz", {"kind": "c4", "weights": "best_026_12000.dat", "games": games})
    # config 1 (BASELINE.json): TicTacToe, 1 game, 25x1 sims
    games = [strip(play_reference(ttt, nt5, nt5, 1, 10, 25, 1, 0, 1234, 600, False), False)]
    dump("real_ttt3.json.gz", {"kind": "mnk", "n": 3, "k": 3, "weights": "best_005_00900.dat", "games": games})
    # ---- G5: arena, two nets, two stores, tau = 0 ----
    games = [strip(play_reference(c4, n26, n25, 2, 0, 10, 8, i & 1, 29, 700 + i, False), False) for i in range(4)]
    dump("arena_c4.json.gz", {"kind": "c4", "weights": ["best_026_12000.dat", "best_025_10600.dat"], "games": games})

    # ---- G4: Net.forward on fixed boards ----
    recs = json.load(gzip.open(os.path.join(HERE, "rules_c4.json.gz"), "rt"))["recs"][:256]
    states = [int(r["s2"]) for r in recs]
    who = [1 - r["p"] for r in recs]
    x = torch.tensor(c4.states_to_training_batch(states, who))
    with torch.no_grad():
        lg, vl = n26(x)
    np.savez_compressed(os.path.join(HERE, "net_c4_forward.npz"), states=np.array(states, dtype=np.uint64),
                        who=np.array(who, np.int32), logits=lg.numpy(), values=vl.numpy())
    print("done in %.1fs" % (time.time() - t0))


if __name__ == "__main__":
    main()
