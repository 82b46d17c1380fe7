"""Phase cycles of heads_f32 (build with tools/exp/build_exp.py 20, run with CARO_HIP_LIB=tools/exp/_build/libcaro_exp20.so)."""
import os, sys
import numpy as np, torch
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from caro_ai_amd import _lib
from caro_ai_amd.lib.model import Net
from caro_ai_amd.net_hip import HipNet
L = _lib.load()
net = Net((2, 6, 7), 7); net.load_state_dict(torch.load("caro_ai_amd/data/weights/best_026_12000.dat", map_location="cpu"))
hn = HipNet(net, "cuda:0")
rows = 1434
x = (torch.rand((rows, 2, 6, 7), device="cuda") < 0.3).float()
counts = torch.tensor([rows, 0], dtype=torch.int32, device="cuda")
probs = torch.empty((rows, 7), device="cuda"); vals = torch.empty(rows, device="cuda")
stamps = torch.zeros(4 * 1024, dtype=torch.int64, device="cuda")
for _ in range(500): hn.forward_dev(x, counts.data_ptr(), 0, rows, probs, vals, None)
_lib.check(L.caro_net_forward_stamped(hn.h, x.data_ptr(), counts.data_ptr(), 0, rows, probs.data_ptr(), vals.data_ptr(), stamps.data_ptr(), None))
torch.cuda.synchronize()
s = stamps.cpu().numpy().reshape(-1, 4); s = s[s[:, 0] > 0]
ph = s[:, 1]
print("workgroups %d: total %.0f, conv_in %.0f, trunk %.0f, heads %.0f" % (len(s), np.median(s[:, 0]), np.median(s[:, 2]), np.median(s[:, 3] - s[:, 2]), np.median(s[:, 0] - s[:, 3])))
print("heads phases (cycles, median): 1x1 conv %.0f | FC value + policy %.0f | tanh + softmax stats %.0f" % (
    np.median(ph & 0xFFFFF), np.median((ph >> 20) & 0xFFFFF), np.median((ph >> 40) & 0xFFFFF)))
