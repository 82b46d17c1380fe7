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
    with pytest.raises(_lib.CaroError):  # eviction inside the ply exists in the multi-wavefront kernel only (batch x lanes > 64)
        _engine(game, 8, [_synth(game, "fused")], max_batch=8, stagger=True, searches_hint=5, evict=True, node_cap=256)
    eng = _engine(game, 8, [_synth(game, "fused")], max_batch=8, stagger=True, searches_hint=5)
    with pytest.raises(_lib.CaroError):  # batch x lanes per descent must be a multiple of 64 (4 x 8 = 32 is not)
        eng.L.caro_search_staggered.restype  # (binding exists)
        _lib.check(eng.L.caro_search_staggered(eng.h, eng.evaluators[0].h, None, 1, 4, eng.planes.data_ptr(), None,
                                               eng._probs.data_ptr(), eng._values.data_ptr(), None))
    eng.close()
    g5 = _game_of({"kind": "mnk", "n": 5, "k": 4})
    with pytest.raises(_lib.CaroError):  # 5 x 5 with batch 3: 96 lanes, not whole wavefronts
        _engine(g5, 4, [_synth(g5, "fused")], max_batch=3, stagger=True, searches_hint=5)
    g15 = _game_of({"kind": "mnk", "n": 15, "k": 5})
    ok = _engine(g15, 4, [_synth(g15, "fused")], max_batch=8, stagger=True, searches_hint=5, evict=True, node_cap=512)
    ok.close()  # (eight wavefronts per game with eviction: accepted since round 6)
    lock = _engine(game, 8, [_synth(game, "fused")], max_batch=8, searches_hint=5)
    assert lock.L.caro_search_staggered(lock.h, lock.evaluators[0].h, None, 1, 8, lock.planes.data_ptr(), None,
                                        lock._probs.data_ptr(), lock._values.data_ptr(), None) == -71  # lock-step engine
    lock.close()


def test_train_self_play_staggered_fills_the_replay_buffer():
    """train.self_play(stagger=True) -- the CLI's throughput form (train.py:25-59's loop, all games at once): at least
    n_games finished games reach the device replay buffer with well-formed rows, and -- connect four, batch 8 -- the
    staggered engine is really what ran.  The same call on TicTacToe with the reference's batch of 8 (two wavefronts per
    game: the multi-wavefront staggered kernel since round 6)."""
    from caro_ai_amd import train
    from caro_ai_amd.lib.model import Net
    game = _game_of({"kind": "c4"})
    assert train.staggered_ok(game, 8) and train.staggered_ok(_game_of({"kind": "mnk", "n": 3, "k": 3}), 8)  # (3x3: two wavefronts)
    assert not train.staggered_ok(_game_of({"kind": "mnk", "n": 3, "k": 3}), 3) and not train.staggered_ok(game, 8, evict=True)
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


@pytest.mark.parametrize("inference", ["hipw", "hipx3"])  # hipx3: the opt-in bf16x3 net kernel (bench.py's labelled extra leg)
def test_staggered_real_weights_vs_reference_recorded_games(inference):
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
                         stagger_recycle=False, inference=inference)
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
    print("staggered (%s) vs reference-recorded games: identical pi on %d / %d plies, %d / 32 whole games" % (inference, same, total, whole))
    assert total >= 400 and same / total >= 0.99 and whole >= 24


@pytest.mark.parametrize("S", [2, 3])
def test_staggered_tiny_search_counts(S):
    """edge cases of the per-game clock: two minibatches per move (the ply is due at every second launch; with ONE
    the reference itself divides by a zero visit total, lib/mcts.py:304-311) and three"""
    _check_against_oracle({"kind": "c4"}, 16, 40, 2, S, 8, 1, seed=90 + S, uid_base=0, form="fused", stagger=True,
                          searches_hint=S)


# ------------------------------------------------------------------ the launches bench.py times, at the sizes it times them
def _full_size_staggered(d, G, S, B, sbt0, n_stores, salts, seed, n_finish, keep_every, max_moves):
    """A staggered engine at a BASELINE configuration's full size with the exact table net(s) on the device (the
    launches are bench.py's -- k_tree_stag + slot rows + a device evaluator that reads the leaf counts itself --, only
    the evaluator is the exact one), slots restarted in place until `n_finish` games have finished.  Checks the
    size-independent properties (nothing overflows, the sims identity incl. the pending minibatches, every finished
    game a legal game with a well-formed z, every uid played once) and returns the kept sample {uid: record} --
    first and later generations of a slot alike -- with the oracle's games of the same uids."""
    game = _game_of(d)
    evs = [_synth(game, "fused", s) for s in salts]
    eng = _engine(game, G, evs, n_stores=n_stores, max_batch=B, steps_before_tau_0=sbt0, seed=seed, uid_base=0,
                  stagger=True, searches_hint=S)
    kept, games, moves, finished = {}, [], 0, 0
    while finished < n_finish:
        eng.search(S, B)   # S launches: on average every game makes one ply
        moves += 1
        dr = eng.drain(recycle=True)
        ng = dr["games"].shape[0]
        if ng:
            gr = dr["games"].cpu().numpy()
            finished += ng
            games.append(gr)
            z = dr["z"].cpu().numpy()
            st = pi = pl = None
            off = 0
            for uid, first, result, steps in gr.tolist():
                n = steps + 1
                zz = z[off:off + n]
                # z alternates back from the last mover: +1 / -1 for a win, all 0 for a draw (utils.py:101-106)
                assert zz[0] == (1 if result != 0 else 0) and (np.abs(zz) == abs(int(zz[0]))).all(), uid
                assert (zz[1::2] == -zz[0]).all() and (zz[0::2] == zz[0]).all(), uid
                if uid % keep_every == 3:
                    if st is None:
                        st, pi, pl = dr["states"].cpu().numpy(), dr["pi"].cpu().numpy(), dr["players"].cpu().numpy()
                    kept[uid] = (first, result, steps, st[off:off + n].copy(), pi[off:off + n].copy(), zz.copy(),
                                 pl[off:off + n].copy())
                off += n
            assert off == z.shape[0]
        assert moves < max_moves
    c = eng.counters()
    pending = eng.pending_leaves()
    eng.close()
    games = np.concatenate(games)
    assert c["overflows"] == 0
    assert c["sims"] % B == 0 and c["sims"] <= moves * S * G * B
    assert c["expansions"] + c["terminals"] + c["dropped"] + pending == c["sims"]
    assert c["finished"] >= len(games) >= n_finish and len(set(games[:, 0].tolist())) == len(games)
    hw = game.obs_shape[1] * game.obs_shape[2]
    assert (games[:, 3] >= 6).all() and (games[:, 3] < hw).all()  # four in a row: no game ends before the 7th ply
    ref = _oracle_games(d, kept.keys(), seed, sbt0, S, B, n_stores, salts=salts if len(salts) == 2 else (salts[0], salts[0]))
    for uid, (first, result, steps, st, pi, z, pl) in kept.items():
        r = ref[uid]
        assert (first, result, steps) == (r["first"], r["result"], r["steps"]), uid
        assert game.from_keys(st.view(np.uint64)) == r["states"][::-1], uid
        assert pl.tolist() == r["players"][::-1].tolist(), uid
        assert np.array_equal(pi, r["pi"][::-1]), uid
        assert z.tolist() == r["z"][::-1].tolist(), uid
    return c, games, kept, moves


def test_config2_full_size_staggered_whole_games_vs_oracle():
    """BASELINE config 2 as bench.py runs it: 1024 connect-four games, 25 x 8 sims/move, tau = 1
    for 10 plies, staggered, slots restarted in place until >= 2048 games have finished; the sampled finished games --
    first, second and third generation of their slots -- equal the oracle's game of the same uid ply by ply (boards,
    players, float64 pi, z, result, steps)."""
    c, games, kept, moves = _full_size_staggered({"kind": "c4"}, 1024, 25, 8, 10, 1, (0,), seed=21, n_finish=2048,
                                                 keep_every=64, max_moves=400)
    gens = sorted({uid // 1024 for uid in kept})
    print("config 2 staggered, full size: %d passes of 25 launches, %d games finished, %d sampled (generations %s), "
          "sims %d, expansions %d" % (moves, len(games), len(kept), gens, c["sims"], c["expansions"]))
    assert len(kept) >= 16 and gens[0] == 0 and gens[-1] >= 1


def test_config5_full_size_staggered_two_nets_whole_games_vs_oracle():
    """BASELINE config 5's shape as bench.py runs it: 512 arena matches, two (salted table) nets,
    one tree per player, 100 x 8 sims/move, tau = 0 from move 0, staggered with restarts until >= 512 matches have
    finished; sampled matches equal the oracle's ply by ply."""
    c, games, kept, moves = _full_size_staggered({"kind": "c4"}, 512, 100, 8, 0, 2, (0x1111, 0x2222), seed=5,
                                                 n_finish=512, keep_every=48, max_moves=200)
    print("config 5 staggered, full size: %d passes of 100 launches, %d matches finished, %d sampled, sims %d, "
          "expansions %d" % (moves, len(games), len(kept), c["sims"], c["expansions"]))
    assert len(kept) >= 8


def _two_real_nets(game, names):
    import os
    from caro_ai_amd.lib.model import Net
    from tests.conftest import GOLDEN
    nets = []
    for nm in names:
        net = Net(game.obs_shape, game.action_space)
        net.load_state_dict(torch.load(os.path.join(GOLDEN, "weights", nm), map_location="cpu"))
        nets.append(net.to(DEV).eval())
    return nets


def _games_by_uid(game, tuples, recs):
    ST = np.concatenate([t["states"] for t in tuples])
    PI = np.concatenate([t["pi"] for t in tuples])
    by_uid, off = {}, 0
    for uid, first, result, steps in recs.tolist():
        n = steps + 1
        by_uid[uid] = (first, result, steps, ST[off:off + n][::-1].copy(), PI[off:off + n][::-1].copy())
        off += n
    return by_uid


def test_staggered_two_real_nets_vs_reference_recorded_arena_800_sims():
    """Config 5's launches with the two REAL nets on the GPU: best_026 vs best_025, 100 x 8
    sims/move, tau = 0, one tree per player, staggered -- against the 8 arena games RECORDED FROM THE REFERENCE
    (tests/golden/arena_c4_800.json.gz).  A staggered engine cannot be stopped after a ply, but its replay rows carry
    pi of every ply (tau = 0: the one-hot of the first maximum of root N = the move played): each game is compared ply
    by ply for as long as it follows the recorded boards.  Tolerance as for the lock-step comparison
    (tests/test_gpu_shim.py::test_real_weights_gpu_net_arena_800_sims): >= 99 % of the compared plies carry the
    reference's pi, and a game that matched at every ply ends with the recorded result and step count.
    Beside it, exactly: the staggered engine's games are the lock-step engine's games bit for bit (same size, same
    nets: every launch of either stays in the net kernel's smallest tile class)."""
    from caro_ai_amd.engine import SelfPlayEngine
    from tests.conftest import load_golden
    d = load_golden("arena_c4_800.json.gz")
    game = _game_of(d)
    games = d["games"]
    g0 = games[0]
    n = len(games)
    assert [gm["uid"] for gm in games] == list(range(g0["uid"], g0["uid"] + n))
    n1, n2 = _two_real_nets(game, d["weights"])
    out = []
    for stagger in (True, False):
        eng = SelfPlayEngine(game, n, net1=n1, net2=n2, n_stores=2, max_batch=8, seed=g0["seed"], uid_base=g0["uid"],
                             steps_before_tau_0=0, first_player_mode=2, device=DEV, searches_hint=100, stagger=stagger,
                             stagger_recycle=False)
        tuples, recs = eng.play_until(100, 8, recycle=False)
        assert eng.live_games() == 0 and eng.counters()["overflows"] == 0
        eng.close()
        out.append(_games_by_uid(game, tuples, recs))
    stag, lock = out
    assert sorted(stag) == sorted(lock) == [gm["uid"] for gm in games]
    for uid in stag:
        a, b = stag[uid], lock[uid]
        assert a[:3] == b[:3] and a[3].tobytes() == b[3].tobytes() and a[4].tobytes() == b[4].tobytes(), uid
    total = same = whole = 0
    for gm in games:
        first, result, steps, st, pis = stag[gm["uid"]]
        assert first == gm["first_player"]
        states = game.from_keys(st.view(np.uint64))
        ok = True
        for ply in range(min(len(states), gm["plies"])):
            if str(states[ply]) != gm["states"][ply]:
                break
            total += 1
            if pis[ply].tolist() == gm["pi"][ply]:
                same += 1
            else:
                ok = False
        else:
            if ok and len(states) == gm["plies"]:
                assert (result, steps) == (gm["result"], gm["steps"]), gm["uid"]
                whole += 1
    print("staggered two-net arena vs reference-recorded games: identical pi on %d / %d plies, %d / %d whole games"
          % (same, total, whole, n))
    assert total >= 100 and same / total >= 0.99 and whole >= 6


def test_lock_step_mutators_refuse_a_staggered_engine():
    """a staggered engine carries per-game clocks, pending minibatches and parked records the lock-step
    entry points know nothing about -- they return CARO_E_STATE instead of leaving that state stale"""
    from caro_ai_amd import _lib
    game = _game_of({"kind": "c4"})
    eng = _engine(game, 8, [_synth(game, "fused")], max_batch=8, stagger=True, searches_hint=5)
    eng.search(5, 8)
    for call in (lambda: eng.reset(), lambda: eng.set_roots([game.initial_state] * 8, [0] * 8),
                 lambda: _lib.check(eng.L.caro_step(eng.h, None, None, None, None, None)),
                 lambda: _lib.check(eng.L.caro_search_batch(eng.h, eng.evaluators[0].h, None, 1, 8, None,
                                                            eng.planes.data_ptr(), None, eng._probs.data_ptr(),
                                                            eng._values.data_ptr(), None)),
                 lambda: _lib.check(eng.L.caro_select(eng.h, 8, 0, None, eng.planes.data_ptr(), None, None))):
        with pytest.raises(_lib.CaroError, match="-71|staggered"):
            call()
    # the engine is still usable and its games still the oracle's
    tuples, games = eng.play_until(5, 8, n_finished=8)
    assert eng.counters()["overflows"] == 0
    eng.close()
    ref = _oracle_games({"kind": "c4"}, games[:, 0], 0, 10, 5, 8, 1)
    for uid, first, result, steps in games.tolist():
        assert (first, result, steps) == (ref[uid]["first"], ref[uid]["result"], ref[uid]["steps"]), uid


@pytest.mark.parametrize("n_games,concurrent", [(48, None), (80, 32)])
def test_train_self_play_plays_the_same_games_staggered_and_lock_step(n_games, concurrent):
    """train.self_play must put the SAME set of games into the replay buffer whichever schedule runs
    -- the first n_games uids of the rank's sequence, each played to its end -- not "whatever finishes first" (which is
    biased towards short games).  With the real net, 48 slots x one game each and 32 slots x 80 games: the replay rows
    of the two schedules are equal as multisets (boards, players, float32 pi, z), and so are the step totals."""
    import os
    from caro_ai_amd import train
    from caro_ai_amd.lib.model import Net
    from tests.conftest import GOLDEN
    game = _game_of({"kind": "c4"})
    net = Net(game.obs_shape, game.action_space)
    net.load_state_dict(torch.load(os.path.join(GOLDEN, "weights", "best_026_12000.dat"), map_location="cpu"))
    net = net.to(DEV).eval()
    rows = []
    for stagger in (False, True):
        rb = train.DeviceReplayBuffer(game, 20000, DEV)
        sp = train.self_play(game, rb, net, n_games, device=DEV, seed=7, uid_base=300, searches=8, batch=8,
                             concurrent=concurrent, stagger=stagger)
        assert sp["games"] == n_games and len(rb) == sp["steps"] + n_games
        if concurrent is None:
            assert sp["games_dropped"] == 0  # one slot per game: nothing beyond the wanted set is ever started
        n = len(rb)
        rec = np.concatenate([rb.states[:n].cpu().numpy().view(np.uint8).reshape(n, -1),
                              rb.players[:n].cpu().numpy().view(np.uint8).reshape(n, -1),
                              rb.pi[:n].cpu().numpy().view(np.uint8).reshape(n, -1),
                              rb.z[:n].cpu().numpy().view(np.uint8).reshape(n, -1)], axis=1)
        rows.append((sorted(map(bytes, rec)), sp["steps"]))
    assert rows[0][1] == rows[1][1]
    assert rows[0][0] == rows[1][0]
    # the empty board opens every game: exactly n_games such rows (a dropped or half-played game would show here)
    empty = np.array([game.to_keys([game.initial_state])[0]]).view(np.uint8).tobytes()
    assert sum(1 for r in rows[1][0] if r.startswith(empty)) == n_games


# ------------------------------------------------------------------ round 6: several wavefronts per game, eviction
@pytest.mark.parametrize("d,G,n_fin,S,B,ns,kw", [
    ({"kind": "mnk", "n": 15, "k": 5}, 6, 9, 5, 8, 1, {"evict": True}),    # config 4's geometry: eight wavefronts, eviction
    ({"kind": "mnk", "n": 15, "k": 5}, 5, 7, 4, 4, 2, {}),                  # four wavefronts, one tree per player, no eviction
    ({"kind": "mnk", "n": 3, "k": 3}, 48, 200, 9, 8, 1, {}),                # TicTacToe with the reference's batch of 8: two wavefronts, draws
    ({"kind": "mnk", "n": 10, "k": 5}, 8, 12, 5, 4, 1, {"evict": True}),    # two actions per lane
    ({"kind": "c4"}, 24, 60, 6, 16, 2, {"evict": True}),                    # connect four with 16 descents: two wavefronts
])
def test_staggered_multi_wave_games_vs_oracle(d, G, n_fin, S, B, ns, kw):
    """the staggered schedule on geometries with several wavefronts per game (k_tree_stag_mw; round 6), with the eviction
    inside the kernel's ply where it is on: every finished game -- first generation of its slot or a later one -- equals
    the oracle's game of the same uid (result, steps, boards, float64 pi, z)"""
    c, ref, games = _check_against_oracle(d, G, n_fin, 3, S, B, ns, seed=31, uid_base=400, form="fused",
                                          salts=(0x1111, 0x2222) if ns == 2 else None, stagger=True, searches_hint=S, **kw)
    assert len(games) >= n_fin and c["overflows"] == 0
    assert games[:, 0].max() >= 400 + G  # a slot restarted in-kernel
    if d.get("n") == 3:
        assert (games[:, 2] == 0).any()  # drawn games among them


def test_staggered_eviction_keeps_the_live_nodes_small_and_equals_lock_step():
    """config 4's shape in small: a staggered engine with eviction and a lock-step engine with eviction play the same
    games (uids, tuples) -- and the staggered one's trees hold what survives its own plies, never more than the cap"""
    d = {"kind": "mnk", "n": 15, "k": 5}
    game = _game_of(d)
    G, S, B = 6, 6, 8
    kw = dict(max_batch=B, steps_before_tau_0=4, seed=8, uid_base=90, searches_hint=S, evict=True, node_cap=600)
    lock = _engine(game, G, [_synth(game, "fused")], **kw)
    tl, gl = lock.play_until(S, B, n_finished=8)
    lock.close()
    stag = _engine(game, G, [_synth(game, "fused")], stagger=True, **kw)
    ts, gs = stag.play_until(S, B, n_finished=8)
    live = stag.tree_live()
    c = stag.counters()
    stag.close()
    assert c["overflows"] == 0 and live.max() <= 600

    def by_uid(tuples, games):
        out, off = {}, 0
        PI = np.concatenate([t["pi"] for t in tuples]); ST = np.concatenate([t["states"] for t in tuples])
        for uid, first, result, steps in games.tolist():
            n = steps + 1
            out[uid] = (first, result, steps, ST[off:off + n].tobytes(), PI[off:off + n].tobytes())
            off += n
        return out
    a, b = by_uid(tl, gl), by_uid(ts, gs)
    common = sorted(set(a) & set(b))
    assert len(common) >= 6
    for uid in common:
        assert a[uid] == b[uid], uid
