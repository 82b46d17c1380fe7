"""In-kernel clock of k_net_forward: per-workgroup shader cycles / 100 MHz ticks, after warm-up."""
import ctypes as C, os, sys
import numpy as np, torch
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from caro_ai_amd import _lib
from caro_ai_amd.lib.model import Net
from caro_ai_amd.net_hip import HipNet
L = _lib.load()
net = Net((2, 6, 7), 7); net.load_state_dict(torch.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "caro_ai_amd/data/weights/") + "best_026_12000.dat", map_location="cpu"))
mode = sys.argv[1] if len(sys.argv) > 1 else "f32"
hn = HipNet(net, "cuda:0", mode=mode)
# MFMA-pipe cycles per SIMD: f32 45 taps x 64 x (2 of 8 waves per SIMD) x 64 cycles; f32w 60 transformed taps x 32 k-steps;
# bf16x3 45 taps x 48 v_mfma_f32_32x32x16_bf16 x 32 cycles
mfma_cycles = {"f32": 23040 * 16, "bf16x3": 45 * 48 * 2 * 32}.get(mode, 15360 * 16)
for rows in (1434, 600) + ((1700, 2300) if mode == "f32w" else ()):
    x = (torch.rand((rows, 2, 6, 7), device="cuda") < 0.3).float()
    counts = torch.tensor([rows, 0], dtype=torch.int32, device="cuda")
    probs = torch.empty((rows, 7), device="cuda"); vals = torch.empty(rows, device="cuda")
    grid = (rows + 5) // 6
    stamps = torch.zeros(4 * max(grid, 1024), dtype=torch.int64, device="cuda")  # split-tile launches use up to 512 workgroups
    for _ in range(2000):   # ~0.5 s of back-to-back launches
        hn.forward_dev(x, counts.data_ptr(), 0, rows, probs, vals, None)
    _lib.check(L.caro_net_forward_stamped(hn.h, x.data_ptr(), counts.data_ptr(), 0, rows, probs.data_ptr(),
                                          vals.data_ptr(), stamps.data_ptr(), None))
    torch.cuda.synchronize()
    s = stamps.cpu().numpy().reshape(-1, 4).astype(np.float64)
    s = s[s[:, 0] > 0]
    if rows > 1536:
        print("   overflow tiles (workgroups >= 256): %d, cycles median %.0f" % ((s.shape[0] - 256), np.median(s[256:, 0])))
        s = s[:256]
    cyc, rt = s[:, 0], s[:, 1]
    if mode == "bf16x3":  # this kernel packs the cycles of its five epilogues above bit 20 of the tick field
        raw = stamps.cpu().numpy().reshape(-1, 4)[:s.shape[0], 1]
        rt = (raw & 0xFFFFF).astype(np.float64)
        print("   the five epilogues (wave 0): %.0f cycles (median)" % np.median((raw >> 20).astype(np.float64)))
    ghz = cyc / (rt * 10.0)
    print("rows %d: workgroups %d, cycles median %.0f, wall us median %.1f, clock GHz median %.3f (min %.3f max %.3f)" % (
        rows, grid, np.median(cyc), np.median(rt) / 100.0, np.median(ghz), ghz.min(), ghz.max()))
    print("   phases (cycles, median): zero+conv_in %.0f | trunk (taps + 5 epilogues) %.0f | heads+softmax %.0f" % (
        np.median(s[:, 2]), np.median(s[:, 3] - s[:, 2]), np.median(s[:, 0] - s[:, 3])))
    print("   MFMA-bound cycles per workgroup and SIMD = %d -> %.1f %% of the measured cycles" % (
        mfma_cycles, 100 * mfma_cycles / np.median(cyc)))
    if mode == "bf16x3" and os.environ.get("CARO_X3_TIMERS"):  # diagnostic build -DCARO_X3_TIMERS=1: per-phase cycles of wave 0, summed over the 45 taps
        t = stamps.cpu().numpy()[4 * 512:4 * 512 + 8 * grid].reshape(-1, 8).astype(np.float64)
        names = ["top: staging loads issued", "segment (c0, blocks 0-1) + reads of (c0, 2-3)", "segment (c0, 2-3) + reads of c1",
                 "segment (c1, 0-1) + reads of (c1, 2-3)", "segment (c1, 2-3) + staged writes + next tap's first sets", "closing barrier"]
        for q, nm in enumerate(names):
            print("      %-52s %8.0f cycles (median), %6.0f per tap" % (nm, np.median(t[:, q]), np.median(t[:, q]) / 45))
