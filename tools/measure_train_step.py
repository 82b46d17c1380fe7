"""Wall time of train_neural_net on the GPU (train.py:62-117: TRAIN_ROUNDS x BATCH_SIZE from a device replay buffer).
The first call pays torch's kernel selection for the convolutions; the later ones are what an iteration costs."""
import time, sys, os
import numpy as np, torch
sys.path.insert(0, os.getcwd())
from caro_ai_amd.lib.game.connect_four import ConnectFour
from caro_ai_amd.lib.model import Net
from caro_ai_amd.train import DeviceReplayBuffer, train_neural_net
from caro_ai_amd import config as cfg
game = ConnectFour()
dev = "cuda:0"
net = Net(game.obs_shape, 7).to(dev)
opt = torch.optim.SGD(net.parameters(), lr=0.1, momentum=0.9)
buf = DeviceReplayBuffer(game, 30000, dev)
rng = np.random.default_rng(0)
states, players = [], []
s, pl = game.initial_state, 1
while len(states) < 4000:
    mv = game.possible_moves(s)
    s2, won = game.move(s, int(rng.choice(mv)), pl)
    states.append(s2); players.append(1 - pl)
    s, pl = (game.initial_state, 1) if won or not game.possible_moves(s2) else (s2, 1 - pl)
n = len(states)
pi = rng.random((n, 7)); pi /= pi.sum(1, keepdims=True)
buf.extend({"states": torch.from_numpy(game.to_keys(states).astype(np.int64)).reshape(n, -1).to(dev),
            "players": torch.tensor(players, dtype=torch.int32, device=dev), "pi": torch.from_numpy(pi).to(dev),
            "z": torch.from_numpy(rng.integers(-1, 2, n)).to(dev)})
for rep in range(4):
    torch.cuda.synchronize(); t = time.time()
    out = train_neural_net(game, buf, net, opt, dev)
    torch.cuda.synchronize()
    print("train_neural_net (%d rounds x batch %d): %.1f ms" % (cfg.TRAIN_ROUNDS, cfg.BATCH_SIZE, (time.time() - t) * 1e3), out, flush=True)
