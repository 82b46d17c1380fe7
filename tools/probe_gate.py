import os, sys, torch
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from tests.test_gpu_net import _net, _boards
from caro_ai_amd.net_hip import HipNet
for wf in ("best_025_10600.dat", "best_026_12000.dat"):
    net = _net((2, 6, 7), 7, wf)
    hns = {m: HipNet(net, "cuda:0", mode=m) for m in ("f32", "f32w", "bf16x3")}
    for L in (1, 7, 300):
        worst = {m: 0.0 for m in hns}; over = {m: 0 for m in hns}
        n = 200 if L == 1 else 40
        for seed in range(n):
            x = _boards(L, (2, 6, 7), 1000 + seed)
            with torch.no_grad():
                lg, vl = net(x); p_ref = torch.softmax(lg, 1)
                lg64, vl64 = net.double()(x.double()); p64 = torch.softmax(lg64, 1)
            net.float()
            e_ref = (p_ref.double() - p64).abs().max().item()
            for m, hn in hns.items():
                p, v = hn(x.cuda()); torch.cuda.synchronize()
                e = (p.cpu().double() - p64).abs().max().item()
                r = e / max(e_ref, 2.5e-7)
                worst[m] = max(worst[m], r); over[m] += e >= max(4 * e_ref, 1e-6)
        print(wf, "L", L, {m: "worst e/e_ref %.2f, over the gate %d/%d" % (worst[m], over[m], n) for m in hns}, flush=True)
