"""What a short light-load gap between two net launches does to the clock the net launch runs at (experiment).
Between stamped launches of k_net_forward_w (1434 rows): nothing | torch.cuda._sleep (one spinning thread) of ~25 us |
a small elementwise kernel | an fp32 matmul of ~25 us.  GHz = workgroup cycles / (100 MHz ticks x 10)."""
import ctypes as C, os, sys
import numpy as np, torch
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
from caro_ai_amd import _lib
from caro_ai_amd.lib.model import Net
from caro_ai_amd.net_hip import HipNet
L = _lib.load()
net = Net((2, 6, 7), 7); net.load_state_dict(torch.load("tests/golden/weights/best_026_12000.dat", map_location="cpu"))
hn = HipNet(net, "cuda:0", mode="f32w")
rows = 1434
x = (torch.rand((rows, 2, 6, 7), device="cuda") < 0.3).float()
counts = torch.tensor([rows, 0], dtype=torch.int32, device="cuda")
probs = torch.empty((rows, 7), device="cuda"); vals = torch.empty(rows, device="cuda")
stamps = torch.zeros(4 * 2048, dtype=torch.int64, device="cuda")
a = torch.randn(1024, 1024, device="cuda"); b = torch.randn(1024, 1024, device="cuda")
small = torch.zeros(4096, device="cuda")
def gap_none(): pass
def gap_sleep(): torch.cuda._sleep(60000)
def gap_small():
    for _ in range(5): small.add_(1.0)
def gap_mm(): torch.mm(a, b)
def gap_sleep_long(): torch.cuda._sleep(600000)
x64 = torch.rand(1024 * 128 * 8, dtype=torch.float64, device="cuda") + 1.0
def gap_f64():
    for _ in range(3): torch.lgamma(x64)
x32 = torch.rand(1024 * 128 * 8, device="cuda") + 1.0
def gap_f32():
    for _ in range(3): torch.lgamma(x32)
for name, gap in (("back to back", gap_none), ("one spinning thread ~25 us", gap_sleep), ("5 small elementwise kernels", gap_small),
                  ("fp32 matmul 1024^3", gap_mm), ("one spinning thread ~250 us", gap_sleep_long), ("3 float64 lgamma kernels over 1 M elements", gap_f64), ("3 float32 lgamma kernels over 1 M elements", gap_f32)):
    for _ in range(200):
        gap(); hn.forward_dev(x, counts.data_ptr(), 0, rows, probs, vals, None)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize(); e0.record()
    for _ in range(200):
        gap(); hn.forward_dev(x, counts.data_ptr(), 0, rows, probs, vals, None)
    e1.record(); torch.cuda.synchronize()
    per = e0.elapsed_time(e1) * 1e3 / 200
    gap()
    _lib.check(L.caro_net_forward_stamped(hn.h, x.data_ptr(), counts.data_ptr(), 0, rows, probs.data_ptr(),
                                          vals.data_ptr(), stamps.data_ptr(), None))
    torch.cuda.synchronize()
    s = stamps.cpu().numpy().reshape(-1, 4).astype(np.float64); s = s[s[:, 0] > 0]
    print("%-32s: pair %.1f us | workgroup cycles %.0f | wall %.1f us | GHz %.2f" % (name, per, np.median(s[:, 0]), np.median(s[:, 1]) / 100.0, np.median(s[:, 0] / (s[:, 1] * 10.0))), flush=True)
