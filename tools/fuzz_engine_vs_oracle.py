#!/usr/bin/env python3
"""GPU check beyond the test-suite's fixed configurations: the ENGINE against the ORACLE on randomly drawn ones --
connect four and m,n,k boards of 3x3 .. 15x15 with any k, searches 2 .. 12, batch 1 .. 16 (non powers of two too), one
store or one per player, one or two table nets, tau switch 0 .. 8, 1 .. 24 concurrent games with recycling, both launch
forms (step-wise kernels / the fused path where the geometry allows), the staggered schedule where one wavefront serves a
game, eviction on or off, the moves through caro_search_batch + caro_step or through caro_search_move, on a fresh engine
or on one restarted in place after another run (round 6) -- every finished game must equal
the oracle's game of the same uid (tests/test_gpu_engine.py::_check_against_oracle: result, steps, boards, players, float64
pi, z), no overflow.

    python tools/fuzz_engine_vs_oracle.py [configurations] [seed]
"""
import os
import sys
import time

import numpy as np

ROOT = os.path.abspath(os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
sys.path.insert(0, ROOT)
from tests.test_gpu_engine import _check_against_oracle  # noqa: E402


def main():
    n_cfg = int(sys.argv[1]) if len(sys.argv) > 1 else 100
    seed = int(sys.argv[2]) if len(sys.argv) > 2 else 1
    rng = np.random.default_rng(seed)
    t0 = time.time()
    done = games = 0
    for i in range(n_cfg):
        if rng.random() < 0.25:
            d, cells = {"kind": "c4"}, 42
        else:
            n = int(rng.choice([3, 3, 4, 4, 5, 5, 6, 7, 8, 9, 10, 12, 15]))
            k = int(rng.integers(3 if n > 3 else 3, min(n, 6) + 1))
            d, cells = {"kind": "mnk", "n": n, "k": k}, n * n
        B = int(rng.choice([1, 2, 3, 4, 5, 8, 8, 16]))
        if rng.random() < 0.35:  # steer a third of the draws onto the one-wavefront geometries (fused k_tree / k_tree_stag)
            B = 8 if d["kind"] == "c4" else 4 if cells <= 16 else 2 if cells <= 32 else 1
        S = int(rng.integers(2, 13))
        if cells >= 100:  # keep the oracle's share in seconds
            S, B = min(S, 5), min(B, 8)
        ns = int(rng.integers(1, 3))
        two_nets = ns == 2 and rng.random() < 0.5
        G = int(rng.integers(1, 25 if cells < 100 else 7))
        n_fin = G + int(rng.integers(0, G + 1))
        form = "fused" if rng.random() < 0.6 else "stepwise"
        kw = {}
        A = 7 if d["kind"] == "c4" else cells
        lpd = 8 if A == 7 else 16 if A <= 16 else 32 if A <= 32 else 64
        if rng.random() < 0.3:
            kw["evict"] = True
        if form == "fused" and B * lpd >= 64 and (B * lpd) % 64 == 0 and (B * lpd > 64 or "evict" not in kw) and rng.random() < 0.5:
            kw.update(stagger=True, searches_hint=S)   # every game on its own minibatch clock: k_tree_stag / k_tree_stag_mw
        if form == "fused" and rng.random() < 0.5:
            kw["one_call"] = True  # search + ply through caro_search_move (the multi-wave kernel's closing launch makes the ply)
        if rng.random() < 0.3:     # exactly n_finish games (caro_config.games_limit): nothing beyond them is started
            kw["games_limit"] = -1  # (filled in below, once n_finish is drawn)
        if rng.random() < 0.25:    # the checked run on an engine RESTARTED in place after another run (caro_engine_restart)
            kw["dirty_first"] = (int(rng.integers(1, 1 << 30)), int(rng.integers(0, 1 << 20)), int(rng.integers(1, 12)))
        cfg = dict(d=d, G=G, n_finish=n_fin, sbt0=int(rng.integers(0, 9)), S=S, B=B, n_stores=ns, seed=int(rng.integers(1, 1 << 30)),
                   uid_base=int(rng.integers(0, 1 << 20)), form=form, salts=(0x1111, 0x2222) if two_nets else None, **kw)
        if cfg.get("games_limit"):
            cfg["games_limit"] = cfg["n_finish"]
            if cfg.get("stagger") and rng.random() < 0.5:
                cfg["stagger_recycle"] = 2  # the pool form: free slots are handed the next unstarted games (k_stag_assign)
        try:
            c, ref, g = _check_against_oracle(**cfg)
        except Exception:
            print("MISMATCH / ERROR at configuration %d: %r" % (i, cfg), flush=True)
            raise
        done += 1
        games += len(g)
        if (i + 1) % 10 == 0:
            print("%d configurations, %d games equal (%.0f s)" % (done, games, time.time() - t0), flush=True)
    print("engine == oracle on %d random configurations, %d whole games (seed %d)" % (done, games, seed))


if __name__ == "__main__":
    main()
