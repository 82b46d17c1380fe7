"""heads_f32 phase cycles on a 15x15 board (tools/exp/build_exp.py 20; CARO_HIP_LIB=tools/exp/_build/libcaro_exp20.so)."""
import os, sys
import numpy as np, torch
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from caro_ai_amd import _lib
from caro_ai_amd.lib.model import Net
from caro_ai_amd.net_hip import HipNet
L = _lib.load()
torch.manual_seed(0)
hn = HipNet(Net((2, 15, 15), 225).eval(), "cuda:0")
rows = 2048
x = (torch.rand((rows, 2, 15, 15), device="cuda") < 0.3).float()
counts = torch.tensor([rows, 0], dtype=torch.int32, device="cuda")
probs = torch.empty((rows, 225), device="cuda"); vals = torch.empty(rows, device="cuda")
stamps = torch.zeros(4 * rows, dtype=torch.int64, device="cuda")
for _ in range(10): hn.forward_dev(x, counts.data_ptr(), 0, rows, probs, vals, None)
_lib.check(L.caro_net_forward_stamped(hn.h, x.data_ptr(), counts.data_ptr(), 0, rows, probs.data_ptr(), vals.data_ptr(), stamps.data_ptr(), None))
torch.cuda.synchronize()
s = stamps.cpu().numpy().reshape(-1, 4); s = s[s[:, 0] > 0]
ph = s[:, 1]
print("15x15 heads %.0f cycles: 1x1 conv (+ staging) %.0f | FC value + policy %.0f | tanh + softmax terms + sum %.0f" % (
    np.median(s[:, 0] - s[:, 3]), np.median(ph & 0xFFFFF), np.median((ph >> 20) & 0xFFFFF), np.median((ph >> 40) & 0xFFFFF)))
