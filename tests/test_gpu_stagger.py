"""Staggered mode (include/caro_hip.h `caro_stagger_enable`, k_tree_stag): every game on its own minibatch clock,
the ply inside the tree kernel, finished games parked and their slots restarted in place.  Per game nothing may
change: every finished game must equal the oracle's game of the same uid -- result, steps, boards, pi (float64
bits), z -- exactly as in lock-step mode, with one net and with two, with restarts and without."""
import numpy as np
import pytest
import torch

from tests.test_gpu_engine import DEV, _check_against_oracle, _engine, _game_of, _oracle_games, _synth

pytestmark = pytest.mark.gpu


def test_connect4_staggered_games_vs_oracle():
    """config 2's per-game settings (25 x 8, tau = 1 for 10 plies), 64 slots restarted until 160 games finished"""
    c, ref, games = _check_against_oracle({"kind": "c4"}, 64, 160, 10, 25, 8, 1, seed=3, uid_base=1000, form="fused",
                                          stagger=True, searches_hint=25)
    assert len(games) >= 160 and c["overflows"] == 0
    assert games[:, 0].max() >= 1000 + 128  # third-generation games: slots restarted in-kernel at least twice


def test_connect4_arena_staggered_two_nets_two_stores_vs_oracle():
    """config 5's shape: two nets, one tree per player, tau = 0 from move 0"""
    _check_against_oracle({"kind": "c4"}, 32, 80, 0, 12, 8, 2, seed=6, uid_base=7000, form="fused",
                          salts=(0x1111, 0x2222), stagger=True, searches_hint=12)


def test_tictactoe_staggered_with_draws_vs_oracle():
    """another one-wavefront geometry (16 lanes x 4 descents) and drawn games"""
    c, ref, games = _check_against_oracle({"kind": "mnk", "n": 3, "k": 3}, 64, 300, 2, 25, 4, 1, seed=9, uid_base=0,
                                          form="fused", stagger=True, searches_hint=25)
    assert (games[:, 2] == 0).any()


def test_staggered_without_restart_every_slot_plays_one_game():
    d = {"kind": "c4"}
    game = _game_of(d)
    G, S, B, seed = 48, 10, 8, 44
    eng = _engine(game, G, [_synth(game, "fused")], max_batch=B, steps_before_tau_0=4, seed=seed, uid_base=500,
                  stagger=True, stagger_recycle=False, searches_hint=S)
    tuples, games = eng.play_until(S, B, recycle=False)
    assert eng.live_games() == 0
    c = eng.counters()
    eng.close()
    assert sorted(games[:, 0].tolist()) == list(range(500, 500 + G)) and c["finished"] == G
    ref = _oracle_games(d, games[:, 0], seed, 4, S, B, 1)
    PI = np.concatenate([t["pi"] for t in tuples])
    off = 0
    for uid, first, result, steps in games.tolist():
        r = ref[uid]
        assert (first, result, steps) == (r["first"], r["result"], r["steps"]), uid
        assert np.array_equal(PI[off:off + r["plies"]], r["pi"][::-1]), uid
        off += r["plies"]
    # the oracle's totals over complete games: the staggered engine does the same work, only at other times
    tot = {k: sum(r["counters"][k] for r in ref.values()) for k in ["sims", "levels", "expansions", "terminals", "dropped"]}
    for k in tot:
        assert c[k] == tot[k], k


def test_staggered_conv_net_games_equal_lock_step_games():
    """the real net (fused HIP kernel): a staggered engine and a lock-step engine of the same size play the same
    games bit for bit (32 slots: every launch of either stays in the smallest tile class of the net kernel)"""
    import os
    from caro_ai_amd.engine import SelfPlayEngine
    from caro_ai_amd.lib.model import Net
    from tests.conftest import GOLDEN
    game = _game_of({"kind": "c4"})
    net = Net(game.obs_shape, game.action_space)
    net.load_state_dict(torch.load(os.path.join(GOLDEN, "weights", "best_026_12000.dat"), map_location="cpu"))
    net = net.to(DEV).eval()
    out = []
    for stagger in (False, True):
        eng = SelfPlayEngine(game, 32, net1=net, max_batch=8, seed=17, device=DEV, searches_hint=25, stagger=stagger)
        tuples, games = eng.play_until(25, 8, n_finished=64)
        eng.close()
        recs, off = {}, 0
        PI = np.concatenate([t["pi"] for t in tuples]); ST = np.concatenate([t["states"] for t in tuples])
        for uid, first, result, steps in games.tolist():
            n = steps + 1
            recs[uid] = (first, result, steps, ST[off:off + n].tobytes(), PI[off:off + n].tobytes())
            off += n
        out.append(recs)
    common = set(out[0]) & set(out[1])
    assert len(common) >= 48
    for uid in common:
        assert out[0][uid] == out[1][uid], uid


def test_staggered_launches_carry_an_even_leaf_count():
    """what the mode is for: at 1024 games the lock-step engine's first minibatches of a move overflow one round of
    net tiles (> 1536 leaves); staggered, expansions per launch stay within a few per cent of their mean"""
    game = _game_of({"kind": "c4"})
    S, B = 25, 8
    eng = _engine(game, 1024, [_synth(game, "fused")], max_batch=B, steps_before_tau_0=10, seed=2, stagger=True,
                  searches_hint=S)
    for _ in range(3):  # warm up: every game has started, clocks are spread
        eng.search(S, B)
    import ctypes as C
    from caro_ai_amd import _lib
    per_launch = []
    prev = eng.counters()["expansions"]
    nets = [e.h for e in eng.evaluators] + [None]
    for _ in range(2 * S):
        _lib.check(eng.L.caro_search_staggered(eng.h, nets[0], nets[1], 1, B, C.c_void_p(eng.planes.data_ptr()),
                                               C.c_void_p(eng.leaf_keys.data_ptr()), C.c_void_p(eng._probs.data_ptr()),
                                               C.c_void_p(eng._values.data_ptr()), eng._stream()))
        now = eng.counters()["expansions"]
        per_launch.append(now - prev)
        prev = now
    eng.close()
    per_launch = np.array(per_launch[1:])  # expansions are booked one launch after their leaves were selected
    print("expansions per launch: mean %.0f min %d max %d" % (per_launch.mean(), per_launch.min(), per_launch.max()))
    assert per_launch.max() < 1.12 * per_launch.mean() and per_launch.min() > 0.88 * per_launch.mean()


def test_staggered_mode_refuses_what_it_cannot_do():
    from caro_ai_amd import _lib
    game = _game_of({"kind": "c4"})
    with pytest.raises(_lib.CaroError):  # eviction uses the second key table
        _engine(game, 8, [_synth(game, "fused")], max_batch=8, stagger=True, searches_hint=5, evict=True, node_cap=256)
    eng = _engine(game, 8, [_synth(game, "fused")], max_batch=8, stagger=True, searches_hint=5)
    with pytest.raises(_lib.CaroError):  # batch x lanes per descent must be 64
        eng.L.caro_search_staggered.restype  # (binding exists)
        _lib.check(eng.L.caro_search_staggered(eng.h, eng.evaluators[0].h, None, 1, 4, eng.planes.data_ptr(), None,
                                               eng._probs.data_ptr(), eng._values.data_ptr(), None))
    eng.close()
    g15 = _game_of({"kind": "mnk", "n": 15, "k": 5})
    with pytest.raises(_lib.CaroError):  # 15 x 15 with batch 8 is eight wavefronts per game
        _engine(g15, 4, [_synth(g15, "fused")], max_batch=8, stagger=True, searches_hint=5)
    lock = _engine(game, 8, [_synth(game, "fused")], max_batch=8, searches_hint=5)
    assert lock.L.caro_search_staggered(lock.h, lock.evaluators[0].h, None, 1, 8, lock.planes.data_ptr(), None,
                                        lock._probs.data_ptr(), lock._values.data_ptr(), None) == -71  # lock-step engine
    lock.close()


def test_train_self_play_staggered_fills_the_replay_buffer():
    """train.self_play(stagger=True) -- the CLI's throughput form (train.py:25-59's loop, all games at once): at least
    n_games finished games reach the device replay buffer with well-formed rows, and -- connect four, batch 8 -- the
    staggered engine is really what ran.  The same call on a geometry without one wavefront per game falls back to
    lock-step."""
    from caro_ai_amd import train
    from caro_ai_amd.lib.model import Net
    game = _game_of({"kind": "c4"})
    assert train.staggered_ok(game, 8) and not train.staggered_ok(_game_of({"kind": "mnk", "n": 3, "k": 3}), 8)
    torch.manual_seed(1)
    net = Net(game.obs_shape, game.action_space).to(DEV).eval()
    rb = train.DeviceReplayBuffer(game, 20000, DEV)
    sp = train.self_play(game, rb, net, 96, device=DEV, seed=3, searches=6, batch=8, concurrent=64, stagger=True)
    assert sp["games"] >= 96 and sp["steps"] >= 7 * 96 and sp["speed_nodes"] > 0
    n = len(rb)
    assert n == sp["steps"] + sp["games"]  # a game of s steps contributes s + 1 rows
    pi = rb.pi[:n].cpu().numpy()
    z = rb.z[:n].cpu().numpy()
    assert np.allclose(pi.sum(1), 1.0, atol=1e-5) and set(np.unique(z).tolist()) <= {-1.0, 0.0, 1.0}
    ttt = _game_of({"kind": "mnk", "n": 3, "k": 3})
    torch.manual_seed(2)
    net3 = Net(ttt.obs_shape, ttt.action_space).to(DEV).eval()
    rb3 = train.DeviceReplayBuffer(ttt, 4096, DEV)
    sp3 = train.self_play(ttt, rb3, net3, 32, device=DEV, seed=3, searches=4, batch=8, concurrent=16, stagger=True)
    assert sp3["games"] >= 32 and len(rb3) == sp3["steps"] + sp3["games"]


def test_staggered_real_weights_vs_reference_recorded_games():
    """The staggered path with the real net on the GPU (what bench.py runs) against the 32 self-play games RECORDED
    FROM THE REFERENCE at config 2's per-game settings (tests/golden/real_c4_x32.json.gz: shipped best_026_12000.dat,
    25 x 8 sims/move, tau = 1 for 10 plies).  A staggered engine cannot be stopped after every ply to read root N, but
    its replay rows carry pi of every ply -- N / sum N for the first ten plies, the one-hot of the first maximum
    afterwards: each game is compared ply by ply for as long as it follows the recorded boards.  Stated tolerance as
    for the lock-step comparison: >= 99 % of the compared plies carry the reference's pi exactly, and a
    game that matched at every ply ends with the recorded result and step count."""
    import os
    from caro_ai_amd.engine import SelfPlayEngine
    from caro_ai_amd.lib.model import Net
    from tests.conftest import GOLDEN, load_golden
    d = load_golden("real_c4_x32.json.gz")
    game = _game_of(d)
    games = d["games"]
    g0 = games[0]
    assert [gm["uid"] for gm in games] == list(range(g0["uid"], g0["uid"] + 32))
    net = Net(game.obs_shape, game.action_space)
    net.load_state_dict(torch.load(os.path.join(GOLDEN, "weights", d["weights"]), map_location="cpu"))
    net = net.to(DEV).eval()
    eng = SelfPlayEngine(game, 32, net1=net, max_batch=g0["batch"], seed=g0["seed"], uid_base=g0["uid"], device=DEV,
                         steps_before_tau_0=g0["steps_before_tau_0"], searches_hint=g0["searches"], stagger=True,
                         stagger_recycle=False)
    tuples, recs = eng.play_until(g0["searches"], g0["batch"], recycle=False)
    assert eng.live_games() == 0 and eng.counters()["overflows"] == 0
    eng.close()
    ST = np.concatenate([t["states"] for t in tuples])
    PI = np.concatenate([t["pi"] for t in tuples])
    by_uid, off = {}, 0
    for uid, first, result, steps in recs.tolist():
        n = steps + 1
        by_uid[uid] = (first, result, steps, game.from_keys(ST[off:off + n].view(np.uint64))[::-1], PI[off:off + n][::-1])
        off += n
    total = same = whole = 0
    for gm in games:
        first, result, steps, states, pis = by_uid[gm["uid"]]
        assert first == gm["first_player"]
        ok = True
        for ply in range(min(len(states), gm["plies"])):
            if str(states[ply]) != gm["states"][ply]:
                break  # another move was played: later plies are another game
            total += 1
            if pis[ply].tolist() == gm["pi"][ply]:
                same += 1
            else:
                ok = False
        else:
            if ok and len(states) == gm["plies"]:
                assert (result, steps) == (gm["result"], gm["steps"]), gm["uid"]
                whole += 1
    print("staggered vs reference-recorded games: identical pi on %d / %d plies, %d / 32 whole games" % (same, total, whole))
    assert total >= 400 and same / total >= 0.99 and whole >= 24


@pytest.mark.parametrize("S", [2, 3])
def test_staggered_tiny_search_counts(S):
    """edge cases of the per-game clock: two minibatches per move (the ply is due at every second launch; with ONE
    the reference itself divides by a zero visit total, lib/mcts.py:304-311) and three"""
    _check_against_oracle({"kind": "c4"}, 16, 40, 2, S, 8, 1, seed=90 + S, uid_base=0, form="fused", stagger=True,
                          searches_hint=S)
