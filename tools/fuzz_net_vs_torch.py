#!/usr/bin/env python3
"""GPU check of the net kernels beyond the suite's shapes: every square board 3x3 .. 15x15 and 6x7, random batch-norm
statistics and weights, launch sizes drawn across the tile-class boundaries of each shape, every HipNet mode the shape
supports (f32 direct, f32w = the Winograd form the board gets, f32w1 row form, f32w2 where supported) against the torch
CPU float32 forward of the same `Net` (|dP| < 1e-4, |dv| < 1e-4 -- tests/test_gpu_net.py's stated tolerance -- and no
further from the float64 forward than 4x torch's own float32 error).

    python tools/fuzz_net_vs_torch.py [cases per shape] [seed]
"""
import os
import sys

import numpy as np
import torch

ROOT = os.path.abspath(os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
sys.path.insert(0, ROOT)
from caro_ai_amd import _lib  # noqa: E402
from caro_ai_amd.net_hip import HipNet  # noqa: E402
from tests.test_gpu_net import _boards, _net  # noqa: E402


def main():
    per_shape = int(sys.argv[1]) if len(sys.argv) > 1 else 6
    seed = int(sys.argv[2]) if len(sys.argv) > 2 else 1
    rng = np.random.default_rng(seed)
    L_ = _lib.load()
    torch.set_num_threads(8)
    worst = {}
    n_cases = 0
    for shape in [(2, 6, 7)] + [(2, n, n) for n in range(3, 16)]:
        A = 7 if shape == (2, 6, 7) else shape[1] * shape[2]
        net = _net(shape, A, None, seed=int(rng.integers(1 << 20)))
        modes = ["f32", "f32w", "f32w1", "bf16x3"] + (["f32w2"] if L_.caro_net_winograd2d_supported(shape[1], shape[2]) else [])
        hn = {m: HipNet(net, "cuda:0", mode=m) for m in modes}
        tb = {m: int(L_.caro_net_boards_per_workgroup(hn[m].h)) for m in modes}
        sizes = {1, 2, int(rng.integers(3, 40))}
        for m in modes:  # around one round of the chip in each mode's tile, and a little above
            sizes |= {max(1, 256 * tb[m] + int(rng.integers(-3, 4))), max(1, 128 * tb[m] + int(rng.integers(-2, 3)))}
        sizes = sorted(sizes)[:per_shape + 3] if shape[1] >= 12 else sorted(sizes)
        for L in sizes:
            L = min(L, 1700 if shape[1] >= 12 else 4000)
            x = _boards(L, shape, int(rng.integers(1 << 20)))
            with torch.no_grad():
                lg, vl = net(x)
                p_ref = torch.softmax(lg, dim=1)
                lg64, _ = net.double()(x.double())
                p64 = torch.softmax(lg64, dim=1)
            net.float()
            e_ref = (p_ref.double() - p64).abs().max().item()
            for m in modes:
                p, v = hn[m](x.to("cuda:0"))
                torch.cuda.synchronize()
                p, v = p.cpu(), v.cpu()
                dp, dv = (p - p_ref).abs().max().item(), (v - vl[:, 0]).abs().max().item()
                e_hip = (p.double() - p64).abs().max().item()
                ok = dp < 1e-4 and dv < 1e-4 and e_hip < max(4 * e_ref, 1e-6) and bool(torch.isfinite(p).all())
                key = (shape[1], shape[2], m)
                worst[key] = max(worst.get(key, 0.0), dp)
                n_cases += 1
                if not ok:
                    print("FAIL shape %s mode %s L %d: dP %.3g dv %.3g e_hip %.3g e_ref %.3g" % (shape, m, L, dp, dv, e_hip, e_ref), flush=True)
                    raise SystemExit(1)
        for h in hn.values():
            h.close()
        print("%dx%d: %d launch sizes x %s ok, worst |dP| %.2e" % (shape[1], shape[2], len(sizes), modes,
                                                                  max(worst[(shape[1], shape[2], m)] for m in modes)), flush=True)
    print("net kernels == torch float32 within tolerance on %d launches (seed %d)" % (n_cases, seed))


if __name__ == "__main__":
    main()
