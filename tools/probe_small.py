"""In-kernel clock of k_net_forward_w in the small-launch regime (K-split tiles): per-workgroup cycles by phase,
and the launch time by HIP events, for leaf counts around config 5's 667 and a half of config 2's 1434."""
import ctypes as C, os, sys
import numpy as np, torch
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from caro_ai_amd import _lib
from caro_ai_amd.lib.model import Net
from caro_ai_amd.net_hip import HipNet
L = _lib.load()
net = Net((2, 6, 7), 7); net.load_state_dict(torch.load("caro_ai_amd/data/weights/best_026_12000.dat", map_location="cpu"))
hn = HipNet(net, "cuda:0", mode="f32w")
rows_list = [int(a) for a in sys.argv[1:]] or [8, 64, 200, 255, 400, 667, 717, 765, 900, 1434, 1536]
for rows in rows_list:
    x = (torch.rand((rows, 2, 6, 7), device="cuda") < 0.3).float()
    counts = torch.tensor([rows, 0], dtype=torch.int32, device="cuda")
    probs = torch.empty((rows, 7), device="cuda"); vals = torch.empty(rows, device="cuda")
    stamps = torch.zeros(4 * 2048, dtype=torch.int64, device="cuda")
    for _ in range(300):
        hn.forward_dev(x, counts.data_ptr(), 0, rows, probs, vals, None)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize()
    e0.record()
    for _ in range(200):
        hn.forward_dev(x, counts.data_ptr(), 0, rows, probs, vals, None)
    e1.record()
    torch.cuda.synchronize()
    us = e0.elapsed_time(e1) * 1e3 / 200
    _lib.check(L.caro_net_forward_stamped(hn.h, x.data_ptr(), counts.data_ptr(), 0, rows, probs.data_ptr(),
                                          vals.data_ptr(), stamps.data_ptr(), None))
    torch.cuda.synchronize()
    s = stamps.cpu().numpy().reshape(-1, 4).astype(np.float64)
    s = s[s[:, 0] > 0]
    cyc = s[:, 0]
    print("rows %5d: launch %.1f us back-to-back | workgroups %d, cycles median %.0f max %.0f | conv_in %.0f trunk %.0f heads %.0f | GHz %.2f"
          % (rows, us, s.shape[0], np.median(cyc), cyc.max(), np.median(s[:, 2]), np.median(s[:, 3] - s[:, 2]),
             np.median(s[:, 0] - s[:, 3]), np.median(cyc / (s[:, 1] * 10.0))), flush=True)
