"""Does the self-play -> train -> arena loop LEARN?  (VERDICT r4 task 4; ref train.py:62-117,120-149,205-217.)
Runs caro_ai_amd.train's own functions from a random net with fixed seeds and the reference's real gate
(BEST_NET_WIN_RATIO 0.60) and prints, per iteration, the losses and every evaluation's win ratio.
    python tools/learn_probe.py [ttt|c4] [iterations] [games per iteration] [evaluate every]
"""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from caro_ai_amd import config as cfg, train  # noqa: E402
from caro_ai_amd.lib.game.connect_four import ConnectFour  # noqa: E402
from caro_ai_amd.lib.game.tictactoe import TicTacToe  # noqa: E402
from caro_ai_amd.lib.model import Net, NetWrapper  # noqa: E402


def run(kind="ttt", iterations=12, games=256, every=2, seed=0, verbose=True):
    game = TicTacToe() if kind == "ttt" else ConnectFour()
    dev = "cuda:0"
    torch.manual_seed(seed)
    net = Net(game.obs_shape, game.action_space).to(dev)
    best = NetWrapper(net)
    initial = NetWrapper(net).target_model  # the random net every challenger is ALSO measured against
    opt = torch.optim.SGD(net.parameters(), lr=cfg.LEARNING_RATE, momentum=0.9)
    rb = train.DeviceReplayBuffer(game, cfg.REPLAY_BUFFER, dev)
    hist = {"loss_total": [], "loss_value": [], "loss_policy": [], "gate": [], "vs_initial": [], "promotions": 0}
    t0 = time.time()
    for it in range(1, iterations + 1):
        sp = train.self_play(game, rb, best.target_model, games, device=dev, seed=seed * 1000 + it,
                             uid_base=it * games, stagger=True)
        if len(rb) < cfg.MIN_REPLAY_TO_TRAIN:
            continue
        gen = torch.Generator(device=dev)
        gen.manual_seed(seed * 1000 + it)
        for _ in range(1):
            ls = train.train_neural_net(game, rb, net, opt, dev, generator=gen)
        for k in ("loss_total", "loss_value", "loss_policy"):
            hist[k].append(ls[k])
        msg = "it %2d  replay %5d  loss %.4f (v %.4f  p %.4f)  %.0f exp/s" % (
            it, len(rb), ls["loss_total"], ls["loss_value"], ls["loss_policy"], sp["speed_nodes"])
        if it % every == 0:
            r, wld = train.evaluate(game, net, best.target_model, rounds=cfg.EVALUATION_ROUNDS, device=dev, seed=it,
                                    counts=True)
            r0, wld0 = train.evaluate(game, net, initial, rounds=cfg.EVALUATION_ROUNDS, device=dev, seed=1000 + it,
                                      counts=True)
            hist["gate"].append(r)
            hist["vs_initial"].append(r0)
            hist.setdefault("wld_vs_initial", []).append(wld0)
            msg += "  | vs best %.2f %s  vs initial %.2f %s" % (r, wld, r0, wld0)
            if r > cfg.BEST_NET_WIN_RATIO:
                best.sync()
                hist["promotions"] += 1
                msg += "  PROMOTED"
        if verbose:
            print(msg, " [%.0f s]" % (time.time() - t0), flush=True)
    hist["seconds"] = time.time() - t0
    return hist


if __name__ == "__main__":
    a = sys.argv[1:]
    h = run(a[0] if a else "ttt", int(a[1]) if len(a) > 1 else 12, int(a[2]) if len(a) > 2 else 256,
            int(a[3]) if len(a) > 3 else 2)
    print({k: (v if not isinstance(v, list) else [round(float(x), 3) if not isinstance(x, tuple) else x for x in v])
           for k, v in h.items()})
