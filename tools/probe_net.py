"""Accuracy of the inference forms vs a float64 forward, on boards from recorded games."""
import gzip, json, os, sys
import numpy as np, torch
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from caro_ai_amd.lib.model import Net, GemmNet
from caro_ai_amd.lib.game.connect_four import ConnectFour
from caro_ai_amd.net_hip import HipNet
g = ConnectFour()
net = Net(g.obs_shape, 7); net.load_state_dict(torch.load("tests/golden/weights/best_026_12000.dat", map_location="cpu")); net.eval()
recs = json.load(gzip.open("tests/golden/rules_c4.json.gz", "rt"))["recs"][:600]
x = torch.from_numpy(g.states_to_training_batch([int(r["s2"]) for r in recs], [1 - r["p"] for r in recs]))
with torch.no_grad():
    lg, vl = net(x); p32 = torch.softmax(lg, 1)
    lg64, vl64 = net.double()(x.double()); p64 = torch.softmax(lg64, 1); net.float()
    gn = GemmNet(net).cuda().eval(); lgg, vlg = gn(x.cuda()); pg = torch.softmax(lgg, 1).cpu(); vlg = vlg.cpu()
hn = HipNet(net, "cuda:0"); ph, vh = hn(x.cuda()); torch.cuda.synchronize(); ph, vh = ph.cpu(), vh.cpu()
h3 = HipNet(net, "cuda:0", mode="3xbf16"); p3, v3 = h3(x.cuda()); torch.cuda.synchronize(); p3, v3 = p3.cpu(), v3.cpu()
def rep(name, p, v):
    dp = (p.double() - p64).abs(); dv = (v.double().reshape(-1) - vl64.reshape(-1)).abs()
    print("%-12s |dP| max %.3e mean %.3e   |dv| max %.3e mean %.3e" % (name, dp.max(), dp.mean(), dv.max(), dv.mean()))
rep("cpu fp32", p32, vl); rep("gemm gpu", pg, vlg); rep("hip f32", ph, vh); rep("hip 3xbf16", p3, v3)
import time
xs = x.cuda()[:600].repeat(3, 1, 1, 1)[:1430].contiguous()
for name, n_ in (("hip f32", hn), ("hip 3xbf16", h3)):
    for _ in range(50): n_(xs)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(200): n_(xs)
    torch.cuda.synchronize(); print("%-12s %.1f us per forward of 1430 leaves" % (name, (time.perf_counter() - t0) / 200 * 1e6))
print("logit range", lg64.min().item(), lg64.max().item())
