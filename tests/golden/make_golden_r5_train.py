#!/usr/bin/env python3
"""Round-5 golden vector for the TRAINING STEP (DESIGN called f1 "parity-unpinned"): the
reference's own `train_neural_net` (ref train.py:62-117) RUN in the build container and recorded.

How a function of a script that SURVEY 8(c) lists as not importable is run all the same -- nothing of it is copied or
restated here, the reference's file is imported as it lies under /root/reference:
  * `train.py` imports `tensorboardX` at its top for `SummaryWriter`, which only its `__main__` block instantiates.  The
    container has no tensorboardX; a module of that name holding a `SummaryWriter` that RAISES when called is registered
    before the import.  It is never called: no arithmetic depends on it.
  * `train_neural_net` reads the module globals `net` and `step_idx` that `__main__` would have set; they are
    set as attributes of the imported module, which is where the function looks them up.
  * `random.sample(replay_buffer, BATCH_SIZE)` (ref train.py:77) is wrapped to LOG the indices it draws: the wrapper draws
    `random.sample(range(len(buffer)), k)` -- the same calls into the generator, the same elements -- and returns the
    buffer's items at those indices.
Inputs: the (state, player, pi, z) tuples of the 32 games of tests/golden/real_c4_x32.json.gz (recorded from the
reference in round 2) as the replay deque, games in file order, plies forward; the shipped best_026_12000.dat as the net;
SGD(lr = LEARNING_RATE, momentum 0.9) as ref train.py:182; `random.seed(2026)`; the reference's own TRAIN_ROUNDS = 10 and
BATCH_SIZE = 256.  Output train_step_c4.json.gz: the drawn indices, the three loss means the function hands to its
tracker, and per tensor of the net's state_dict afterwards a SHA-256 of its bytes + sum / abs-max in float64.

Usage:  python tests/golden/make_golden_r5_train.py
"""
import collections
import hashlib
import os
import random
import sys
import types

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
import make_golden as mg  # noqa: E402  (puts /root/reference first on sys.path and imports its lib)
from tests.conftest import load_golden  # noqa: E402


def import_reference_train():
    stub = types.ModuleType("tensorboardX")

    class SummaryWriter:  # only ref train.py's __main__ block would instantiate it
        def __init__(self, *a, **k):
            raise RuntimeError("tensorboardX stand-in: SummaryWriter must never be instantiated by the golden script")

    stub.SummaryWriter = SummaryWriter
    sys.modules["tensorboardX"] = stub
    import train as ref_train
    assert os.path.abspath(ref_train.__file__) == os.path.join(mg.REF, "train.py")
    return ref_train


class Tracker:
    def __init__(self):
        self.seen = {}

    def track(self, name, value, step):
        self.seen[name] = float(value)


def tensor_digest(t):
    a = t.detach().cpu().contiguous().numpy()
    return {"sha256": hashlib.sha256(a.tobytes()).hexdigest(), "sum": float(a.astype(np.float64).sum()),
            "absmax": float(np.abs(a.astype(np.float64)).max()) if a.size else 0.0, "shape": list(a.shape),
            "dtype": str(a.dtype)}


def main():
    torch.set_num_threads(1)
    ref_train = import_reference_train()
    cfg = ref_train.cfg
    game = mg.ConnectFour()
    d = load_golden("real_c4_x32.json.gz")
    rb = collections.deque(maxlen=cfg.REPLAY_BUFFER)
    for gm in d["games"]:
        for s, p, pi, z in zip(gm["states"], gm["players"], gm["pi"], gm["z"]):
            rb.append((int(s), int(p), [float(x) for x in pi], int(z)))
    assert len(rb) >= cfg.BATCH_SIZE
    net = mg.ref_model.Net(game.obs_shape, game.action_space)
    net.load_state_dict(torch.load(os.path.join(mg.REF, "saves/trained_connect4/best_026_12000.dat"), map_location="cpu"))
    optimizer = torch.optim.SGD(net.parameters(), lr=cfg.LEARNING_RATE, momentum=0.9)
    ref_train.net = net        # what ref train.py's __main__ sets (Q15)
    ref_train.step_idx = 1
    drawn = []
    real_sample = random.sample

    def sample(population, k):
        idx = real_sample(range(len(population)), k)
        drawn.append([int(i) for i in idx])
        return [population[i] for i in idx]

    tracker = Tracker()
    random.seed(2026)
    random.sample = sample
    try:
        ref_train.train_neural_net(game, rb, optimizer, tracker, "cpu")
    finally:
        random.sample = real_sample
    assert len(drawn) == cfg.TRAIN_ROUNDS and all(len(r) == cfg.BATCH_SIZE for r in drawn)
    # the wrapper draws what the plain call draws: same seed, plain random.sample on the deque, same elements
    random.seed(2026)
    assert [rb[i] for i in drawn[0]] == random.sample(rb, cfg.BATCH_SIZE)
    out = {"kind": "c4", "tuples_from": "real_c4_x32.json.gz", "n_tuples": len(rb), "weights": "best_026_12000.dat",
           "seed": 2026, "train_rounds": cfg.TRAIN_ROUNDS, "batch_size": cfg.BATCH_SIZE, "lr": cfg.LEARNING_RATE,
           "momentum": 0.9, "indices": drawn, "losses": tracker.seen,
           "state_dict": {k: tensor_digest(v) for k, v in net.state_dict().items()},
           "torch": torch.__version__}
    print("losses", tracker.seen)
    mg.dump("train_step_c4.json.gz", out)


if __name__ == "__main__":
    main()
