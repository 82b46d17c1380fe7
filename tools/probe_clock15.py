"""Workgroup clock of the net kernel on 15x15 boards by form: python tools/probe_clock15.py [f32w2 f32w1 f32 ...]
(per-workgroup stamps of caro_net_forward_stamped: total | conv_in | trunk | heads cycles, and the launch time of 7 600
boards -- config 4's leaf count per launch -- by HIP events)."""
import os, sys
import numpy as np, torch
sys.path.insert(0, os.getcwd())
from caro_ai_amd import _lib
from caro_ai_amd.lib.model import Net
from caro_ai_amd.net_hip import HipNet
L = _lib.load()
torch.manual_seed(0)
net = Net((2, 15, 15), 225).eval()
rows = 7600
x = (torch.rand((rows, 2, 15, 15), device="cuda") < 0.3).float()
counts = torch.tensor([rows, 0], dtype=torch.int32, device="cuda")
probs = torch.empty((rows, 225), device="cuda"); vals = torch.empty(rows, device="cuda")
ref = None
for mode in (sys.argv[1:] or ["f32w2", "f32w1"]):
    hn = HipNet(net, "cuda:0", mode=mode)
    stamps = torch.zeros(4 * rows, dtype=torch.int64, device="cuda")
    for _ in range(3): hn.forward_dev(x, counts.data_ptr(), 0, rows, probs, vals, None)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(10): hn.forward_dev(x, counts.data_ptr(), 0, rows, probs, vals, None)
    e1.record(); torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / 10
    _lib.check(L.caro_net_forward_stamped(hn.h, x.data_ptr(), counts.data_ptr(), 0, rows, probs.data_ptr(), vals.data_ptr(), stamps.data_ptr(), None))
    torch.cuda.synchronize()
    s = stamps.cpu().numpy().reshape(-1, 4).astype(np.float64); s = s[s[:, 0] > 0]
    p = probs.cpu()
    if ref is None: ref = p
    print("%-6s 15x15 x %d boards: launch %.3f ms | workgroups %d, cycles median %.0f: conv_in %.0f | trunk %.0f | heads %.0f | max|dP| vs first form %.2e"
          % (mode, rows, ms, len(s), np.median(s[:, 0]), np.median(s[:, 2]), np.median(s[:, 3] - s[:, 2]), np.median(s[:, 0] - s[:, 3]), (p - ref).abs().max().item()))
    hn.close()
