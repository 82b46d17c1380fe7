"""Per-layer clock of the 2-D Winograd trunk on 15x15 boards (experiment build 24):
python tools/exp/build_exp.py 24 && CARO_HIP_LIB=tools/exp/_build/libcaro_exp24.so python tools/probe_layers15.py"""
import ctypes as C, os, sys
import numpy as np, torch
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from caro_ai_amd import _lib
from caro_ai_amd.lib.model import Net
from caro_ai_amd.net_hip import HipNet
L = _lib.load()
L.caro_exp_read_lst.argtypes = [C.c_void_p]
torch.manual_seed(0)
net = Net((2, 15, 15), 225).eval()
hn = HipNet(net, "cuda:0", mode=(sys.argv[1] if len(sys.argv) > 1 else "f32w2"))
names = ["phase 0 main loop", "fold + zero", "phase 1 main loop", "fold + old/bias requests", "barrier A (inputs read)",
         "exchange write + barrier B", "finish + write"]
rows = 7600
x = (torch.rand((rows, 2, 15, 15), device="cuda") < 0.3).float()
counts = torch.tensor([rows, 0], dtype=torch.int32, device="cuda")
probs = torch.empty((rows, 225), device="cuda"); vals = torch.empty(rows, device="cuda")
for _ in range(5):
    hn.forward_dev(x, counts.data_ptr(), 0, rows, probs, vals, None)
torch.cuda.synchronize()
out = np.zeros(64 * 5 * 8, np.uint64)
assert L.caro_exp_read_lst(out.ctypes.data) == 0
st = out.reshape(64, 5, 8).astype(np.float64)
d = np.diff(st, axis=2)  # [wg, layer, phase]
ok = d[:, 0, 0] > 0
med = np.median(d[ok], axis=0)
print("wave 0 of %d workgroups: cycles per phase, median" % ok.sum())
for i, n in enumerate(names):
    print("   %-28s" % n, " ".join("%8.0f" % med[l, i] for l in range(5)), "| sum %8.0f" % med[:, i].sum())
gap = np.median(st[ok][:, 1:, 0] - st[ok][:, :-1, 7], axis=0)
print("   %-28s" % "closing barrier -> next layer", " ".join("%8.0f" % g for g in gap))
print("   %-28s" % "layer (start -> start)", " ".join("%8.0f" % g for g in np.median(st[ok][:, 1:, 0] - st[ok][:, :-1, 0], axis=0)))
