"""k_net_forward launch time vs grid size at the same L: does the max-sized grid (early-exit workgroups) cost time?"""
import os, sys
import numpy as np, torch
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from caro_ai_amd.lib.model import Net
from caro_ai_amd.net_hip import HipNet
net = Net((2, 6, 7), 7); net.load_state_dict(torch.load("tests/golden/weights/best_026_12000.dat", map_location="cpu"))
hn = HipNet(net, "cuda:0")
L = 1430
x = (torch.rand((8192, 2, 6, 7), device="cuda") < 0.3).float()
counts = torch.tensor([L, 0], dtype=torch.int32, device="cuda")
probs = torch.empty((8192, 7), device="cuda"); vals = torch.empty(8192, device="cuda")
for max_rows in (1434, 1536, 2048, 3072, 8192):
    for _ in range(200):
        hn.forward_dev(x, counts.data_ptr(), 0, max_rows, probs, vals, None)
    torch.cuda.synchronize()
    ts = []
    for _ in range(200):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(); hn.forward_dev(x, counts.data_ptr(), 0, max_rows, probs, vals, None); e1.record()
        torch.cuda.synchronize(); ts.append(e0.elapsed_time(e1) * 1e3)
    print("L=%d max_rows=%5d grid=%4d: median %.1f us  min %.1f" % (L, max_rows, (max_rows + 5) // 6, np.median(ts), np.min(ts)))
