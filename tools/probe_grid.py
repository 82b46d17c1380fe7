"""Cost of the early-exit workgroups of k_net_forward_w: the same 1434 (or 1300) leaves launched with a grid
sized for the leaves and with the grid of the fused path (G*B = 8192 rows -> 1366 workgroups)."""
import os, sys
import torch
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from caro_ai_amd import _lib
from caro_ai_amd.lib.model import Net
from caro_ai_amd.net_hip import HipNet
L = _lib.load()
net = Net((2, 6, 7), 7); net.load_state_dict(torch.load("caro_ai_amd/data/weights/best_026_12000.dat", map_location="cpu"))
hn = HipNet(net, "cuda:0")
cap = 8192
x = (torch.rand((cap, 2, 6, 7), device="cuda") < 0.3).float()
probs = torch.empty((cap, 7), device="cuda"); vals = torch.empty(cap, device="cuda")
for rows in (1434, 1300, 600):
    counts = torch.tensor([rows, 0], dtype=torch.int32, device="cuda")
    for max_rows in (rows, 1536, 2048, 3072, 4096, cap):
        if max_rows < rows: continue
        for _ in range(200): hn.forward_dev(x, counts.data_ptr(), 0, max_rows, probs, vals, None)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        torch.cuda.synchronize(); e0.record()
        for _ in range(500): hn.forward_dev(x, counts.data_ptr(), 0, max_rows, probs, vals, None)
        e1.record(); torch.cuda.synchronize()
        print("leaves %5d  grid for %5d rows (%4d workgroups): %.1f us per launch" % (
            rows, max_rows, (max_rows + 5) // 6, e0.elapsed_time(e1) * 2.0))
