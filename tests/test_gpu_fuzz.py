"""A randomised sweep inside the GPU suite: the engine against the oracle on configurations the fixed tests do not name
(tools/fuzz_engine_vs_oracle.py is the long form: 1 500 configurations, 25 231 games, profiles/r05_engine_fuzz.txt)."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("seed", [11, 12, 13])
def test_engine_equals_oracle_on_random_configurations(seed):
    """40 configurations per seed -- connect four and m,n,k 3x3 .. 15x15, any k <= 6, searches 2 .. 12, batch 1 / 2 / 3 / 4 /
    5 / 8 / 16, one store or one per player, one or two table nets, tau switch 0 .. 8, 1 .. 24 concurrent games with
    recycling, step-wise kernels or the fused path (one or several wavefronts per game), the staggered schedule where one
    wavefront serves a game, eviction on or off, moves through caro_search_batch + caro_step or caro_search_move, a fresh
    engine or one restarted in place after another run: every finished game equals the oracle's game of the same uid,
    nothing overflows."""
    from tests.test_gpu_engine import _check_against_oracle
    rng = np.random.default_rng(seed)
    games = 0
    for i in range(40):
        if rng.random() < 0.25:
            d, cells = {"kind": "c4"}, 42
        else:
            n = int(rng.choice([3, 3, 4, 4, 5, 5, 6, 7, 8, 9, 10, 12, 15]))
            d, cells = {"kind": "mnk", "n": n, "k": int(rng.integers(3, min(n, 6) + 1))}, n * n
        B = int(rng.choice([1, 2, 3, 4, 5, 8, 8, 16]))
        if rng.random() < 0.35:  # a third of the draws on the one-wavefront geometries (fused k_tree / k_tree_stag)
            B = 8 if d["kind"] == "c4" else 4 if cells <= 16 else 2 if cells <= 32 else 1
        S = int(rng.integers(2, 13))
        if cells >= 100:
            S, B = min(S, 5), min(B, 8)
        ns = int(rng.integers(1, 3))
        two_nets = ns == 2 and rng.random() < 0.5
        G = int(rng.integers(1, 25 if cells < 100 else 7))
        form = "fused" if rng.random() < 0.6 else "stepwise"
        A = 7 if d["kind"] == "c4" else cells
        lpd = 8 if A == 7 else 16 if A <= 16 else 32 if A <= 32 else 64
        kw = {}
        if rng.random() < 0.3:
            kw["evict"] = True
        if form == "fused" and B * lpd >= 64 and (B * lpd) % 64 == 0 and (B * lpd > 64 or "evict" not in kw) and rng.random() < 0.5:
            kw.update(stagger=True, searches_hint=S)  # one wavefront per game: k_tree_stag; several: k_tree_stag_mw (+ eviction)
        if form == "fused" and rng.random() < 0.5:
            kw["one_call"] = True  # search + ply through caro_search_move
        if rng.random() < 0.3:     # exactly n_finish games (caro_config.games_limit): nothing beyond them is started
            kw["games_limit"] = -1  # (filled in below, once n_finish is drawn)
        if rng.random() < 0.25:    # on an engine restarted in place after another run (caro_engine_restart)
            kw["dirty_first"] = (int(rng.integers(1, 1 << 30)), int(rng.integers(0, 1 << 20)), int(rng.integers(1, 12)))
        cfg = dict(d=d, G=G, n_finish=G + int(rng.integers(0, G + 1)), sbt0=int(rng.integers(0, 9)), S=S, B=B, n_stores=ns,
                   seed=int(rng.integers(1, 1 << 30)), uid_base=int(rng.integers(0, 1 << 20)), form=form,
                   salts=(0x1111, 0x2222) if two_nets else None, **kw)
        if cfg.get("games_limit"):
            cfg["games_limit"] = cfg["n_finish"]
            if cfg.get("stagger"):  # in-kernel restarts with the slot's own next uid, or the pool form (k_stag_assign)
                cfg["stagger_recycle"] = 2 if rng.random() < 0.5 else 1
        try:
            c, ref, g = _check_against_oracle(**cfg)
        except Exception:
            print("configuration %d of seed %d: %r" % (i, seed, cfg))
            raise
        games += len(g)
    assert games > 300
