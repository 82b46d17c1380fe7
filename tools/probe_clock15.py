import os, sys
import numpy as np, torch
sys.path.insert(0, "/root/repo")
sys.path.insert(0, os.getcwd())
from caro_ai_amd import _lib
from caro_ai_amd.lib.model import Net
from caro_ai_amd.net_hip import HipNet
L = _lib.load()
torch.manual_seed(0)
net = Net((2, 15, 15), 225).eval()
hn = HipNet(net, "cuda:0")
rows = 2048
x = (torch.rand((rows, 2, 15, 15), device="cuda") < 0.3).float()
counts = torch.tensor([rows, 0], dtype=torch.int32, device="cuda")
probs = torch.empty((rows, 225), device="cuda"); vals = torch.empty(rows, device="cuda")
stamps = torch.zeros(4 * rows, dtype=torch.int64, device="cuda")
for _ in range(20): hn.forward_dev(x, counts.data_ptr(), 0, rows, probs, vals, None)
_lib.check(L.caro_net_forward_stamped(hn.h, x.data_ptr(), counts.data_ptr(), 0, rows, probs.data_ptr(), vals.data_ptr(), stamps.data_ptr(), None))
torch.cuda.synchronize()
s = stamps.cpu().numpy().reshape(-1, 4).astype(np.float64); s = s[s[:, 0] > 0]
print("15x15: workgroups %d, cycles median %.0f: conv_in %.0f | trunk %.0f | heads %.0f" % (len(s), np.median(s[:, 0]), np.median(s[:, 2]), np.median(s[:, 3] - s[:, 2]), np.median(s[:, 0] - s[:, 3])))
