#!/usr/bin/env python3
"""Round-3 additions to the golden vectors, again by RUNNING THE REFERENCE (build container only; the outputs
are committed, the reference is not).  Same harness as make_golden.py (imported from it).  SURVEY 8(c) G3 for
BASELINE config 4 -- the m,n,k game at 15 x 15, k = 5, 50 x 8 = 400 sims/move:

  real_mnk15.json.gz       N_REAL games of the reference's TicTacToe(15, 5) at 50 x 8 with the conv net: the
                           state_dict is produced by THIS repo's Net under torch.manual_seed(SEED_W) and loaded
                           into the reference's Net (the seed is committed, not the weights; the fixture carries a
                           SHA-256 of the tensors so that a test can tell it rebuilt the same ones)
  synth_mnk15_400.json.gz  N_SYNTH games at 50 x 8 with the synthetic table net (search isolated from conv
                           numerics; everything bit-exact)

tau = 1 for 10 plies (config.STEPS_BEFORE_TAU_0), first player = uid & 1.

Usage:  python tests/golden/make_golden_r3.py
"""
import hashlib
import os
import sys
import time

import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
import make_golden as mg  # noqa: E402  (puts /root/reference on sys.path and imports its lib)

SEED_W = 0
N_REAL = 8
N_SYNTH = 3


def own_state_dict(game, seed):
    """the repo's own Net, seeded; tests rebuild it with the same two lines"""
    from caro_ai_amd.lib.model import Net
    torch.manual_seed(seed)
    return Net(game.obs_shape, game.action_space).state_dict()


def state_dict_sha256(sd):
    h = hashlib.sha256()
    for k in sorted(sd):
        h.update(k.encode())
        h.update(sd[k].detach().cpu().contiguous().numpy().tobytes())
    return h.hexdigest()


def slim(g):
    g = mg.strip(g, False)
    g["trace"] = [{"N": t["N"], "nodes": t["nodes"]} for t in g["trace"]]
    return g


def main():
    t0 = time.time()
    g15 = mg.TicTacToe(15, 5)
    sd = own_state_dict(g15, SEED_W)
    net = mg.ref_model.Net(g15.obs_shape, g15.action_space)
    net.load_state_dict(sd)
    net.eval()
    games = []
    for i in range(N_REAL):
        games.append(slim(mg.play_reference(g15, net, net, 1, 10, 50, 8, i & 1, 41, 4000 + i, False)))
        print("conv-net game %d: %d plies, result %d, %d nodes, %.0f s"
              % (i, games[-1]["plies"], games[-1]["result"], games[-1]["trace"][-1]["nodes"], time.time() - t0),
              flush=True)
    mg.dump("real_mnk15.json.gz", {"kind": "mnk", "n": 15, "k": 5, "weights_seed": SEED_W,
                                   "weights_sha256": state_dict_sha256(sd), "games": games})
    games = []
    for i in range(N_SYNTH):
        sn = mg.SynthNet(g15)
        games.append(mg.strip(mg.play_reference(g15, sn, sn, 1, 10, 50, 8, i & 1, 43, 4100 + i, True), False))
        print("table-net game %d: %d plies, result %d, %d nodes, %.0f s"
              % (i, games[-1]["plies"], games[-1]["result"], games[-1]["trace"][-1]["nodes"], time.time() - t0),
              flush=True)
    mg.dump("synth_mnk15_400.json.gz", {"kind": "mnk", "n": 15, "k": 5, "games": games})
    print("done in %.1fs" % (time.time() - t0))


if __name__ == "__main__":
    main()
