"""Soak of the stream form of self-play (train.self_play_stream): 60 consecutive calls of 4 096 connect-four games on 1 024
slots at 25 x 8 with the shipped net; prints one JSON line (rate over everything, overflows, games).
`python tools/stream_soak.py [net_mode]` -- net_mode f32w (default) or bf16x3."""
import json, os, sys, time, torch
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from caro_ai_amd import train
from caro_ai_amd.data import weights_path
from caro_ai_amd.lib.game.connect_four import ConnectFour
from caro_ai_amd.lib.model import Net
g = ConnectFour()
net = Net(g.obs_shape, g.action_space)
net.load_state_dict(torch.load(weights_path("best_026_12000.dat"), map_location="cpu"))
net = net.to("cuda:0").eval()
mode = sys.argv[1] if len(sys.argv) > 1 else "f32w"
rb = train.DeviceReplayBuffer(g, 1 << 18, "cuda:0")
t0 = time.time(); nodes = games = 0
for i in range(60):
    sp = train.self_play_stream(g, rb, net, 4096, device="cuda:0", searches=25, batch=8, concurrent=1024, uid_base=0, net_mode=mode)
    nodes += sp["nodes"]; games += sp["games"]
torch.cuda.synchronize(); dt = time.time() - t0
eng = next(iter(train._ENGINES.values()))
c = eng.counters()
print(json.dumps({"what": "60 consecutive train.self_play_stream calls of 4 096 connect-four games on 1 024 slots, 25 x 8, shipped net, net_mode " + mode,
                  "seconds": dt, "games": games, "node_expansions": nodes, "speed_nodes": nodes / dt, "overflows": c["overflows"],
                  "finished_by_the_engine": c["finished"], "replay_rows": len(rb)}))
train.release_engines()
