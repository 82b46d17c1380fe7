"""Per-layer clock of the row-Winograd trunk (experiment build 24: tools/exp/build_exp.py 24, then
CARO_HIP_LIB=tools/exp/_build/libcaro_exp24.so python tools/probe_layers.py [rows ...])."""
import ctypes as C, os, sys
import numpy as np, torch
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from caro_ai_amd import _lib
from caro_ai_amd.lib.model import Net
from caro_ai_amd.net_hip import HipNet
L = _lib.load()
L.caro_exp_read_lst.argtypes = [C.c_void_p]
L.caro_exp_read_pst.argtypes = [C.c_void_p]
net = Net((2, 6, 7), 7); net.load_state_dict(torch.load("caro_ai_amd/data/weights/best_026_12000.dat", map_location="cpu"))
hn = HipNet(net, "cuda:0", mode="f32w")
names = ["main loop", "output transform", "barrier (inputs read)", "K-split exchange", "bias+leaky+write", "vmcnt+barrier"]
for rows in [int(a) for a in sys.argv[1:]] or [200, 717, 1434]:
    x = (torch.rand((rows, 2, 6, 7), device="cuda") < 0.3).float()
    counts = torch.tensor([rows, 0], dtype=torch.int32, device="cuda")
    probs = torch.empty((rows, 7), device="cuda"); vals = torch.empty(rows, device="cuda")
    for _ in range(50):
        hn.forward_dev(x, counts.data_ptr(), 0, rows, probs, vals, None)
    torch.cuda.synchronize()
    out = np.zeros(64 * 5 * 8, np.uint64)
    assert L.caro_exp_read_lst(out.ctypes.data) == 0
    d = np.diff(out.reshape(64, 5, 8)[:, :, :7].astype(np.float64), axis=2)  # [wg, layer, phase]
    ok = d[:, 0, 0] > 0
    med = np.median(d[ok], axis=0)  # [layer, phase]
    print("rows %d (%d workgroups stamped): cycles per phase, median over workgroups" % (rows, ok.sum()))
    for i, n in enumerate(names):
        print("   %-24s" % n, " ".join("%8.0f" % med[l, i] for l in range(5)), "| sum %8.0f" % med[:, i].sum())
    print("   %-24s" % "layer total", " ".join("%8.0f" % med[l].sum() for l in range(5)), "| sum %8.0f" % med.sum())
    pst = np.zeros(64 * 16, np.uint64)
    assert L.caro_exp_read_pst(pst.ctypes.data) == 0
    q = pst.reshape(64, 16).astype(np.float64)[ok]
    pn = ["leaf count known", "LDS zeroed + fetches issued", "conv_in weights arrived", "slot rows mapped (tile_rows)",
          "conv_in", "chunks arrived + barrier"]
    print("   prologue:", " | ".join("%s %.0f" % (n, np.median(q[:, i + 1] - q[:, i])) for i, n in enumerate(pn)),
          "| total %.0f" % np.median(q[:, 6] - q[:, 0]))
    hnames = ["1x1 convolutions", "FC stage", "softmax + stores", "tail"]
    print("   heads:", " | ".join("%s %.0f" % (n, np.median(q[:, 9 + i] - q[:, 8 + i])) for i, n in enumerate(hnames)),
          "| total %.0f" % np.median(q[:, 12] - q[:, 8]))
