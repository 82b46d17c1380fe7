"""Time of the Dirichlet rows of one minibatch (caro_noise_batch = the lanes, functions and bits the tree kernels use):
M rows of A actions, HIP events around repeated launches.  python tools/bench_noise.py [A] [M]"""
import os, sys
import numpy as np, torch
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from caro_ai_amd import _lib
A = int(sys.argv[1]) if len(sys.argv) > 1 else 225
M = int(sys.argv[2]) if len(sys.argv) > 2 else 8192
L = _lib.load()
rng = np.random.default_rng(1)
uid = torch.from_numpy(rng.integers(0, 2**40, M).astype(np.int64)).cuda()
ply = torch.from_numpy(rng.integers(0, 200, M).astype(np.int32)).cuda()
sim = torch.from_numpy(rng.integers(0, 400, M).astype(np.int32)).cuda()
out = torch.zeros((M, A), dtype=torch.float64, device="cuda")
def run():
    _lib.check(L.caro_noise_batch(7, M, A, 0.3, uid.data_ptr(), ply.data_ptr(), sim.data_ptr(), out.data_ptr(), None))
for _ in range(5): run()
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(50): run()
e1.record(); torch.cuda.synchronize()
import hashlib
print("lib %s: A=%d M=%d: %.1f us per launch, sha %s, row sum %.17g" % (os.path.basename(_lib.LIB_PATH), A, M, e0.elapsed_time(e1) * 1e3 / 50,
      hashlib.sha256(out.cpu().numpy().tobytes()).hexdigest()[:16], float(out[0].sum())))
