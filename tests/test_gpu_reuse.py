"""Round 6: a self-play engine that is RESTARTED in place (caro_engine_restart) instead of re-created, the engine's
`games_limit` (exactly the wanted games are played), refused plies on a root without visits, and node-pool overflows
surfaced by every caller (VERDICT r5 tasks 1 and 3)."""
import os
import subprocess
import sys

import numpy as np
import pytest
import torch

from tests.conftest import ROOT, WEIGHTS
from tests.test_gpu_engine import DEV, _engine, _game_of, _oracle_games, _synth

pytestmark = pytest.mark.gpu


def _collect(tuples, games):
    out, off = {}, 0
    if not len(games):
        return out
    PI = np.concatenate([t["pi"] for t in tuples])
    ST = np.concatenate([t["states"] for t in tuples])
    Z = np.concatenate([t["z"] for t in tuples])
    for uid, first, result, steps in games.tolist():
        n = steps + 1
        out[uid] = (first, result, steps, ST[off:off + n].tobytes(), PI[off:off + n].tobytes(), Z[off:off + n].tobytes())
        off += n
    return out


@pytest.mark.parametrize("stagger", [False, True])
def test_restarted_engine_plays_the_fresh_engines_games(stagger):
    """caro_engine_restart leaves the state caro_engine_create leaves: after a first run that filled the trees, flipped
    key tables (staggered restarts) and parked games, a restart with run B's key plays B exactly as a fresh engine
    does -- same uids, same tuples bit for bit, same counters -- and both equal the oracle's games."""
    d = {"kind": "c4"}
    game = _game_of(d)
    G, S, B = 24, 6, 8
    kw = dict(max_batch=B, steps_before_tau_0=4, searches_hint=S, stagger=stagger)
    run_a = dict(seed=5, uid_base=100, uid_stride=G)
    run_b = dict(seed=17, uid_base=9000, uid_stride=G)
    fresh = _engine(game, G, [_synth(game, "fused")], **kw, **run_b)
    want = _collect(*fresh.play_until(S, B, n_finished=60))
    c_want = fresh.counters()
    fresh.close()
    eng = _engine(game, G, [_synth(game, "fused")], **kw, **run_a)
    eng.play_until(S, B, n_finished=40)  # dirty: nodes everywhere, restarted slots, a drain behind it
    eng.restart(**run_b)
    got = _collect(*eng.play_until(S, B, n_finished=60))
    assert eng.counters() == c_want
    # and once more on the same engine, with the staggered mode's recycling switched off and a games_limit
    eng.restart(stagger_recycle=False, games_limit=G, **run_b) if stagger else eng.restart(games_limit=G, **run_b)
    t, g = eng.play_until(S, B, recycle=not stagger) if stagger else eng.play_until(S, B, n_finished=G)
    one_gen = _collect(t, g)
    eng.close()
    assert sorted(got) == sorted(want) and len(got) >= 60
    for uid in want:
        assert got[uid] == want[uid], uid
    assert sorted(one_gen) == list(range(9000, 9000 + G))
    for uid in one_gen:
        assert one_gen[uid] == want[uid], uid
    ref = _oracle_games(d, sorted(want)[:12], 17, 4, S, B, 1)
    for uid, r in ref.items():
        assert want[uid][:3] == (r["first"], r["result"], r["steps"]), uid


def test_restart_refuses_another_shape_and_a_pending_drain():
    from caro_ai_amd import _lib
    game = _game_of({"kind": "c4"})
    eng = _engine(game, 8, [_synth(game, "fused")], max_batch=8, searches_hint=4)
    import ctypes as C
    c = _lib.CaroConfig.from_buffer_copy(eng.cfg)
    c.n_games = 16
    assert eng.L.caro_engine_restart(eng.h, C.byref(c), None) == -22
    eng.search(4, 8)
    eng.step()
    eng.drain_begin(True)
    assert eng.L.caro_engine_restart(eng.h, C.byref(eng.cfg), None) == -71
    eng.drain_end()
    with pytest.raises(AssertionError):
        eng.restart(node_cap=5)
    eng.restart(seed=3)
    assert eng.counters()["sims"] == 0
    eng.close()


@pytest.mark.parametrize("stagger", [False, True])
@pytest.mark.parametrize("n_games", [20, 37])
def test_games_limit_plays_exactly_the_wanted_games(stagger, n_games):
    """slot g plays its k-th game only while k * G + g < games_limit: with 8 slots and 20 (37) wanted games exactly
    the uids base + 0 .. base + 19 (36) are played, in either schedule, and afterwards every slot is finished"""
    d = {"kind": "c4"}
    game = _game_of(d)
    G, S, B, seed, base = 8, 5, 8, 23, 4000
    eng = _engine(game, G, [_synth(game, "fused")], max_batch=B, steps_before_tau_0=4, seed=seed, uid_base=base,
                  searches_hint=S, stagger=stagger, stagger_recycle=True, games_limit=n_games)
    tuples, games = eng.play_until(S, B, n_finished=n_games, max_moves=60 * 6)
    for _ in range(3):  # nothing more comes
        eng.search(S, B)
        eng.step()
        assert int(eng.drain(recycle=True)["games"].shape[0]) == 0
    assert eng.live_games() == 0
    c = eng.counters()
    eng.close()
    assert sorted(games[:, 0].tolist()) == list(range(base, base + n_games))
    assert c["finished"] == n_games and c["overflows"] == 0
    ref = _oracle_games(d, games[:, 0], seed, 4, S, B, 1)
    got = _collect(tuples, games)
    for uid, r in ref.items():
        assert got[uid][:3] == (r["first"], r["result"], r["steps"]), uid
    # every counted simulation belongs to a wanted game
    assert c["sims"] == sum(r["counters"]["sims"] for r in ref.values())
    assert c["expansions"] == sum(r["counters"]["expansions"] for r in ref.values())


def test_fewer_wanted_games_than_slots():
    game = _game_of({"kind": "c4"})
    eng = _engine(game, 16, [_synth(game, "fused")], max_batch=8, searches_hint=4, uid_base=50, games_limit=5)
    assert eng.live_games() == 5
    _, games = eng.play_until(4, 8, recycle=True, max_moves=60)
    eng.close()
    assert sorted(games[:, 0].tolist()) == [50, 51, 52, 53, 54]


@pytest.mark.parametrize("stagger", [False, True])
def test_reused_self_play_engine_plays_the_fresh_engines_games(stagger):
    """train.self_play keeps its engine and its HipNet between calls: the second call of a shape restarts the cached
    engine in place, and its replay rows are, bit for bit and in order, those of a call on a fresh engine"""
    from caro_ai_amd import net_hip, train
    from caro_ai_amd.lib.model import Net
    game = _game_of({"kind": "c4"})
    net = Net(game.obs_shape, game.action_space)
    net.load_state_dict(torch.load(os.path.join(WEIGHTS, "best_026_12000.dat"), map_location="cpu"))
    net = net.to(DEV).eval()
    kw = dict(device=DEV, searches=6, batch=8, concurrent=32, stagger=stagger)

    def rows(rb):
        n = len(rb)
        return [t[:n].cpu().numpy().tobytes() for t in (rb.states, rb.players, rb.pi, rb.z)]

    train.release_engines()
    rb0 = train.DeviceReplayBuffer(game, 20000, DEV)
    sp0 = train.self_play(game, rb0, net, 80, seed=4, uid_base=640, reuse=False, **kw)
    assert not sp0["engine_reused"] and not train._ENGINES
    rb1 = train.DeviceReplayBuffer(game, 20000, DEV)
    train.self_play(game, rb1, net, 80, seed=9, uid_base=0, **kw)  # another run first: leaves the engine dirty
    hip = net_hip.hipnet_for(net, DEV)
    rb2 = train.DeviceReplayBuffer(game, 20000, DEV)
    sp2 = train.self_play(game, rb2, net, 80, seed=4, uid_base=640, **kw)
    assert sp2["engine_reused"] and len(train._ENGINES) == 1
    assert net_hip.hipnet_for(net, DEV) is hip  # same weights: the same packed net
    assert (sp2["games"], sp2["steps"], sp2["nodes"]) == (80, sp0["steps"], sp0["nodes"]) and sp2["games_dropped"] == 0
    assert rows(rb2) == rows(rb0) and rows(rb1) != rows(rb0)
    # new weights: a new HipNet, the engine stays
    with torch.no_grad():
        net.conv_in[0].weight.mul_(1.01)
    rb3 = train.DeviceReplayBuffer(game, 20000, DEV)
    sp3 = train.self_play(game, rb3, net, 80, seed=4, uid_base=640, **kw)
    assert sp3["engine_reused"] and net_hip.hipnet_for(net, DEV) is not hip
    assert rows(rb3) != rows(rb0)
    train.release_engines()
    assert not train._ENGINES


# ------------------------------------------------------------------ a root without visits is not played from
@pytest.mark.parametrize("form", ["stepwise", "fused"])
def test_one_search_on_an_unexpanded_root_refuses_the_ply(form):
    """ONE minibatch on an unexpanded root only expands it (lib/mcts.py:123): no edge has a visit and the reference's
    get_policy_value divides by zero at tau = 1 (mcts.py:311).  The engine refuses the ply -- action -1, the game stays
    where it was, the `overflows` tally every caller checks is bumped -- instead of moving on a NaN policy; the next
    search finds an expanded root and the game goes on."""
    game = _game_of({"kind": "c4"})
    G = 8
    eng = _engine(game, G, [_synth(game, form)], max_batch=8, steps_before_tau_0=10, seed=2, searches_hint=4)
    roots0 = eng.roots()[0].copy()
    eng.search(1, 8)
    a, done, res = eng.step()
    assert a.cpu().tolist() == [-1] * G and done.cpu().tolist() == [0] * G
    r = eng.roots()
    assert np.array_equal(r[0], roots0) and r[2].tolist() == [0] * G
    c = eng.counters()
    assert c["overflows"] == G and c["plies"] == 0
    eng.search(2, 8)
    a, done, _ = eng.step()
    assert all(0 <= x < 7 for x in a.cpu().tolist()) and eng.counters()["plies"] == G
    eng.close()


def test_staggered_ply_on_a_root_without_visits_is_refused_once_and_the_game_goes_on():
    """the same inside the staggered tree kernel (step_body's one-wave form): with ONE minibatch per move the ply after
    the root's own expansion finds no visits, is refused, the game searches again and then moves"""
    game = _game_of({"kind": "c4"})
    G = 8
    eng = _engine(game, G, [_synth(game, "fused")], max_batch=8, steps_before_tau_0=10, seed=2, searches_hint=1,
                  stagger=True, stagger_recycle=False)
    eng.search(1, 8)
    eng.search(1, 8)  # launch 2: expand of the root, the ply is due -> refused
    c = eng.counters()
    assert c["overflows"] == G and c["plies"] == 0
    eng.search(1, 8)  # launch 3: a minibatch with descents below the root has been backed up -> the ply is made
    c = eng.counters()
    assert c["overflows"] == G and c["plies"] == G and eng.roots()[2].tolist() == [1] * G
    eng.close()


def test_tau0_zero_visit_root_plays_the_references_action_zero():
    """at tau = 0 the reference does not divide: argmax of the all-zero visit row is action 0 (mcts.py:305-307), played
    if legal -- the engine does the same"""
    game = _game_of({"kind": "c4"})
    eng = _engine(game, 4, [_synth(game, "fused")], max_batch=8, steps_before_tau_0=0, seed=2, searches_hint=4)
    eng.search(1, 8)
    a, _, _ = eng.step()
    assert a.cpu().tolist() == [0] * 4 and eng.counters()["overflows"] == 0
    eng.close()


# ------------------------------------------------------------------ overflows are errors in every caller
def _c4_nets():
    from caro_ai_amd.lib.model import Net
    game = _game_of({"kind": "c4"})
    nets = []
    for w in ("best_026_12000.dat", "best_025_10600.dat"):
        n = Net(game.obs_shape, game.action_space)
        n.load_state_dict(torch.load(os.path.join(WEIGHTS, w), map_location="cpu"))
        nets.append(n.to(DEV).eval())
    return game, nets


def test_self_play_raises_on_overflow():
    from caro_ai_amd import _lib, train
    game, (net, _) = _c4_nets()
    rb = train.DeviceReplayBuffer(game, 4096, DEV)
    with pytest.raises(_lib.CaroError, match="overflow"):
        train.self_play(game, rb, net, 16, device=DEV, searches=10, batch=8, stagger=True, node_cap=24)
    assert not train._ENGINES  # the failed engine is not kept
    with pytest.raises(_lib.CaroError, match="overflow"):
        train.self_play(game, rb, net, 16, device=DEV, searches=10, batch=8, stagger=False, node_cap=24)
    train.release_engines()


def test_play_games_and_evaluate_raise_on_overflow():
    import collections
    from caro_ai_amd import _lib, train
    from caro_ai_amd.lib.utils import play_games
    game, (a, b) = _c4_nets()
    with pytest.raises(_lib.CaroError, match="overflow"):
        play_games(game, 8, collections.deque(), a, None, mcts_searches=10, mcts_batch_size=8, node_cap=24)
    with pytest.raises(_lib.CaroError, match="overflow"):
        train.evaluate(game, a, b, rounds=4, device=DEV, node_cap=24)
    with pytest.raises(MemoryError):
        train.evaluate(game, a, b, rounds=1, device=DEV, node_cap=24, reference_stores=True)
    assert 0.0 <= train.evaluate(game, a, b, rounds=4, device=DEV) <= 1.0  # the default cap cannot overflow


def test_play_cli_and_bench_fail_on_overflow():
    from caro_ai_amd import _lib, play
    a, b = (os.path.join(WEIGHTS, w) for w in ("best_026_12000.dat", "best_025_10600.dat"))
    with pytest.raises(_lib.CaroError, match="overflow"):
        play.main(["-g", "0", "--cuda", a, b, "-r", "2", "--node-cap", "24"])
    env = dict(os.environ, PYTHONPATH=ROOT)
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--steps", "3", "--warmup", "1", "--games", "64",
                        "--node-cap", "40", "--no-cpu-baseline", "--no-extra-configs", "--sustained-moves", "0",
                        "--train-loop-games", "0"], env=env, capture_output=True, text=True, timeout=600)
    assert r.returncode == 3, (r.returncode, r.stderr[-2000:])
    assert "overflowed the node pool" in r.stderr


def test_play_games_large_board_defaults_to_eviction():
    """15 x 15 at 50 x 8: the no-overflow bound (90 064 nodes of 4 KB) is beyond a default tree, so play_games runs with
    eviction and the default live-node cap; the games are played, nothing overflows"""
    from caro_ai_amd.lib.game.tictactoe import TicTacToe
    from caro_ai_amd.lib.model import Net
    from caro_ai_amd.lib.utils import play_games
    game = TicTacToe(15, 5)
    torch.manual_seed(0)
    net = Net(game.obs_shape, game.action_space).to(DEV).eval()
    res, stats = play_games(game, 2, None, net, None, mcts_searches=50, mcts_batch_size=8, steps_before_tau_0=4,
                            return_stats=True)
    assert len(res) == 2 and stats["counters"]["overflows"] == 0 and stats["counters"]["finished"] == 2


def test_play_games_fills_a_deque_from_1024_gomoku_games():
    """f3 at scale (VERDICT r5 task 4): 1 024 games on the 15 x 15 board -> reference-format tuples in a deque through
    the vectorised conversions; a sample is compared with the oracle's games of the same uids"""
    import collections
    import time
    from caro_ai_amd.lib.game.tictactoe import TicTacToe
    from caro_ai_amd.lib.utils import play_games
    from caro_ai_amd.net_hip import HashNet
    game = TicTacToe(15, 5)
    d = {"kind": "mnk", "n": 15, "k": 5}
    S, B, seed = 3, 8, 12
    from caro_ai_amd.engine import SelfPlayEngine
    eng = SelfPlayEngine(game, 1024, evaluators=[HashNet(game, device=DEV)], max_batch=B, steps_before_tau_0=6, seed=seed,
                         device=DEV, searches_hint=S, games_limit=1024)
    dq = collections.deque()
    by_uid = {}
    t_conv = 0.0
    for _ in range(260):  # (a 15 x 15 game has at most 225 plies)
        if not eng.live_games():
            break
        eng.search(S, B)
        eng.step()
        dd = eng.drain(recycle=True)
        if int(dd["games"].shape[0]):
            keys = dd["states"].cpu().numpy().view(np.uint64)
            t0 = time.perf_counter()
            states = game.from_keys(keys)
            t_conv += time.perf_counter() - t0
            rows = list(zip(states, dd["players"].cpu().numpy().tolist(), dd["pi"].cpu().numpy().tolist(),
                            dd["z"].cpu().numpy().tolist()))
            dq.extend(rows)
            off = 0
            for uid, first, result, steps in dd["games"].cpu().numpy().tolist():
                by_uid[uid] = rows[off:off + steps + 1]
                off += steps + 1
    assert eng.counters()["overflows"] == 0
    eng.close()
    assert len(by_uid) == 1024 and len(dq) > 1024 * 9
    assert t_conv / len(dq) < 20e-6, t_conv / len(dq)  # measured ~1.5 us per state (was 46 us through the per-bit loop)
    back = game.to_keys([r[0] for r in list(dq)[:2000]])
    assert game.from_keys(back) == [r[0] for r in list(dq)[:2000]]
    sample = [0, 1, 511, 1023]
    ref = _oracle_games(d, sample, seed, 6, S, B, 1)
    for uid in sample:
        r = ref[uid]
        rows = by_uid[uid]
        assert len(rows) == r["plies"]
        assert [x[0] for x in rows] == [int(s) for s in r["states"][::-1]], uid
        assert np.array_equal(np.array([x[2] for x in rows]), r["pi"][::-1]), uid


def test_self_play_stream_consumes_every_started_game_exactly_once():
    """train.self_play_stream: the engine is not stopped between calls -- three calls of 48 games on 32 slots: each call
    hands over at least the games asked for, the second and third continue the first one's engine, no game is drained
    twice (checked inside), and every game the engine has finished is in the replay buffer or in the one open drain (no
    started game is dropped).  New weights restart the stream with uids beyond everything the old one may have started."""
    from caro_ai_amd import train
    from caro_ai_amd.lib.model import Net
    d = {"kind": "c4"}
    game = _game_of(d)
    torch.manual_seed(5)
    net = Net(game.obs_shape, game.action_space).to(DEV).eval()
    train.release_engines()
    G, S, B = 32, 5, 8
    rb = train.DeviceReplayBuffer(game, 1 << 16, DEV)
    uids, rows = [], 0
    for i in range(3):
        sp = train.self_play_stream(game, rb, net, 48, device=DEV, seed=3, uid_base=1000, searches=S, batch=B, concurrent=G)
        assert sp["games"] >= 48 and sp["engine_reused"] == (i > 0) and sp["nodes"] > 0
        rows += sp["rows"]
        eng = next(iter(train._ENGINES.values()))
        assert eng._stream_state["passes"] > 0
    assert len(rb) == rows
    c = eng.counters()
    assert c["overflows"] == 0
    # the engine's own tally: everything finished so far has been handed out except what the open drain holds
    fin = c["finished"]
    d_open = eng.flush()
    n_open = 0 if d_open is None else int(d_open["games"].shape[0])
    n = len(rb)
    st = rb.states[:n].cpu().numpy()
    empty = game.to_keys([game.initial_state])[0]
    assert int((st == empty).all(axis=1).sum()) + n_open == fin  # one opening position per finished game
    with torch.no_grad():
        net.conv_in[0].weight.mul_(1.01)
    sp = train.self_play_stream(game, rb, net, 48, device=DEV, seed=3, uid_base=1000, searches=S, batch=B, concurrent=G)
    assert not sp["engine_reused"]
    assert next(iter(train._ENGINES.values()))._stream_state["base"] > 1000 + G
    train.release_engines()


def test_bench_selfcheck_two_ranks_on_one_gpu():
    """bench.py --gpus 2 --selfcheck as two gloo ranks sharing the one GPU: the checklist passes (incl. the engine whose
    gathered tuples equal the same uids played on rank 0 alone) and its record is in the JSON line; with one rank's
    engine tuples sabotaged every rank ends with exit code 4 and the check's name"""
    import json
    env = dict(os.environ, PYTHONPATH=ROOT, CARO_SHARE_GPU="1", CARO_DIST_BACKEND="gloo")
    for k in ("WORLD_SIZE", "RANK", "LOCAL_RANK"):
        env.pop(k, None)
    cmd = [sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--selfcheck", "--steps", "2", "--warmup", "1",
           "--games", "64", "--no-cpu-baseline"]
    r = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stderr[-3000:]
    line = json.loads([l for l in r.stdout.splitlines() if l.startswith("{")][-1])
    assert line["selfcheck"]["checks"][-1] == "engine_tuples" and line["selfcheck"]["world_size"] == 2
    assert line["n_gpus"] == 2 and line["dist"]["world_size"] == 2
    r = subprocess.run(cmd, env=dict(env, CARO_SELFCHECK_FAULT="engine_tuples"), capture_output=True, text=True, timeout=900)
    assert r.returncode == 4, (r.returncode, r.stderr[-3000:])
    assert "selfcheck FAILED" in r.stderr and "engine_tuples:" in r.stderr


@pytest.mark.parametrize("d,B", [({"kind": "mnk", "n": 15, "k": 5}, 8), ({"kind": "mnk", "n": 10, "k": 5}, 4),
                                 ({"kind": "c4"}, 16)])
def test_multi_wave_fused_kernel_and_search_move_equal_search_plus_step(d, B):
    """Geometries with several wavefronts per game (15 x 15 with 8 descents = config 4's: eight; 10 x 10 with 4; connect
    four with 16 descents: two) take the fused multi-wave tree kernel: no k_encode / k_expand_backup launches between a
    move's minibatches.  And caro_search_move (the ply + the eviction inside the search's closing launch) leaves, move
    after move, exactly what caro_search_batch + caro_step leave: roots, plies, counters, live nodes, drained tuples --
    which equal the oracle's games."""
    game = _game_of(d)
    G, S, seed = 6, 5, 19
    kw = dict(max_batch=B, steps_before_tau_0=3, seed=seed, searches_hint=S, evict=True, node_cap=512, uid_base=70)
    a = _engine(game, G, [_synth(game, "fused")], **kw)
    b = _engine(game, G, [_synth(game, "fused")], **kw)
    a.profile(True)
    ta, tb, ga, gb = [], [], [], []
    for move in range(14):
        a.search(S, B)
        a.step()
        b.search_step(S, B)
        ra, rb = a.roots(), b.roots()
        for x, y in zip(ra, rb):
            assert np.array_equal(x, y), move
        assert a.counters() == b.counters() and np.array_equal(a.tree_live(), b.tree_live()), move
        for eng, t, g in ((a, ta, ga), (b, tb, gb)):
            dd = eng.drain(recycle=True)
            if int(dd["games"].shape[0]):
                t.append({k: v.cpu().numpy() for k, v in dd.items() if k != "games"})
                g.append(dd["games"].cpu().numpy())
    prof = a.profile_read()
    assert prof["select"][1] > 0 and prof["compact"][1] == 0, prof  # the fused kernel ran, k_encode never did
    assert a.counters()["overflows"] == 0
    a.close(); b.close()
    if d["kind"] == "c4":  # (the short boards finish games inside 14 moves)
        assert ga
    if ga:
        A_, B_ = _collect(ta, np.concatenate(ga)), _collect(tb, np.concatenate(gb))
        assert A_ == B_
        ref = _oracle_games(d, sorted(A_)[:3], seed, 3, S, B, 1)
        for uid, r in ref.items():
            assert A_[uid][:3] == (r["first"], r["result"], r["steps"]), uid


def test_bench_under_torchrun_two_ranks_on_one_gpu():
    """the DRIVER's launch form for N > 1 -- python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr
    127.0.0.1 --master-port P bench.py --gpus N --steps K --warmup W -- as two gloo ranks sharing the one GPU: one JSON
    line from rank 0, n_gpus 2, weak scaling, the dist record with both ranks, value = the sum of what the ranks played
    over the max-over-ranks time"""
    import json
    import socket
    sk = socket.socket()
    sk.bind(("127.0.0.1", 0))
    port = sk.getsockname()[1]
    sk.close()
    env = dict(os.environ, PYTHONPATH=ROOT, CARO_SHARE_GPU="1", CARO_DIST_BACKEND="gloo", OMP_NUM_THREADS="1")
    for k in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT"):
        env.pop(k, None)
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "3", "--warmup", "2",
           "--games", "128"]
    r = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=900, cwd=ROOT)
    assert r.returncode == 0, r.stderr[-3000:]
    lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, lines  # rank 0 alone prints
    d = json.loads(lines[0])
    assert d["n_gpus"] == 2 and d["scaling"] == "weak" and d["steps"] == 3 and d["warmup"] == 2
    assert d["dist"]["world_size"] == 2 and d["dist"]["backend"] == "gloo" and len(d["dist"]["ranks"]) == 2
    assert d["cpu_baseline"] is None and d["train_loop"] is None  # N = 1 only
    assert d["value"] > 0 and d["overflows"] == 0 and abs(d["per_gpu"] * 2 - d["value"]) < 1e-6 * d["value"]


@pytest.mark.parametrize("d,B,kw", [({"kind": "c4"}, 8, {}), ({"kind": "mnk", "n": 3, "k": 3}, 8, {}),
                                    ({"kind": "mnk", "n": 10, "k": 5}, 4, {"evict": True})])
def test_staggered_pool_mode_hands_free_slots_the_next_games(d, B, kw):
    """caro_config.stagger_recycle = 2 with games_limit: a finished slot is handed the next game of the wanted set not
    started yet (at the drain, in slot order) instead of its own next uid.  Exactly the wanted games are played, each
    equal to the oracle's game of its uid; slots play games that are not "theirs"; and a second run is the first one
    bit for bit (the assignment is a function of the games' progress, not of timing)."""
    game = _game_of(d)
    G, S, seed, base, n_games = 8, 5, 37, 300, 44
    runs = []
    for _ in range(2):
        eng = _engine(game, G, [_synth(game, "fused")], max_batch=B, steps_before_tau_0=3, seed=seed, uid_base=base,
                      searches_hint=S, stagger=True, stagger_recycle=2, games_limit=n_games,
                      node_cap=S * B * game.obs_shape[1] * game.obs_shape[2] + 64, **kw)
        slots = {}
        tuples, games = [], []
        for _ in range(400):
            eng.search(S, B)
            dd = eng.drain(recycle=True)
            if int(dd["games"].shape[0]):
                tuples.append({k: v.cpu().numpy() for k, v in dd.items() if k != "games"})
                games.append(dd["games"].cpu().numpy())
            for g, u in enumerate(eng.roots()[3].tolist()):
                slots.setdefault(int(u), g)
            if eng.live_games() == 0 and sum(len(x) for x in games) >= n_games:
                break
        c = eng.counters()
        eng.close()
        games = np.concatenate(games)
        runs.append((tuples, games, c, slots))
    tuples, games, c, slots = runs[0]
    assert sorted(games[:, 0].tolist()) == list(range(base, base + n_games)) and c["finished"] == n_games and c["overflows"] == 0
    assert any((u - base) % G != g for u, g in slots.items() if base <= u < base + n_games)  # a slot played another slot's uid
    ref = _oracle_games(d, games[:, 0], seed, 3, S, B, 1)
    got = _collect(tuples, games)
    for uid, r in ref.items():
        assert got[uid][:3] == (r["first"], r["result"], r["steps"]), uid
    assert c["sims"] == sum(r["counters"]["sims"] for r in ref.values())
    assert np.array_equal(runs[1][1], games) and runs[1][2] == c and _collect(runs[1][0], runs[1][1]) == got


def test_self_play_pool_plays_the_same_games():
    from caro_ai_amd import train
    game, (net, _) = _c4_nets()

    def rows(rb):
        n = len(rb)
        rec = np.concatenate([t[:n].cpu().numpy().view(np.uint8).reshape(n, -1) for t in (rb.states, rb.players, rb.pi, rb.z)], axis=1)
        return sorted(map(bytes, rec))
    out = {}
    for pool in (False, True):
        rb = train.DeviceReplayBuffer(game, 60000, DEV)
        sp = train.self_play(game, rb, net, 256, device=DEV, seed=2, uid_base=50, searches=6, batch=8, concurrent=32,
                             stagger=True, pool=pool, reuse=False)
        out[pool] = (rows(rb), sp["steps"], sp["passes"])
        assert sp["games"] == 256
    assert out[True][0] == out[False][0] and out[True][1] == out[False][1]  # the same games, row for row
    # (the passes: with many short games per slot the pool form ends sooner -- bench.py's train_loop: 95 against 113 for
    # 4 096 games on 1 024 slots --; with 8 long games per slot, as here, the two are about even)
    assert out[True][2] <= out[False][2] * 1.1


def test_self_play_with_the_split_bf16_net_mode():
    """train.self_play / self_play_stream with net_mode="bf16x3" (the opt-in arithmetic of k_net_forward_x3): complete
    games of the wanted uids, no overflow, tuples of the usual shape; and the cached HipNet of the default mode is a
    different object (a mode change never reuses the other mode's weights image)."""
    import torch
    from caro_ai_amd import net_hip, train
    from caro_ai_amd.lib.game.connect_four import ConnectFour
    from caro_ai_amd.lib.model import Net
    from caro_ai_amd.train import DeviceReplayBuffer
    g = ConnectFour()
    net = Net(g.obs_shape, g.action_space)
    net.load_state_dict(torch.load(os.path.join(WEIGHTS, "best_026_12000.dat"), map_location="cpu"))
    net = net.to("cuda:0").eval()
    assert net_hip.hipnet_for(net, "cuda:0", mode="bf16x3") is not net_hip.hipnet_for(net, "cuda:0")
    assert net_hip.hipnet_for(net, "cuda:0", mode="bf16x3").mode == "bf16x3"
    rb = DeviceReplayBuffer(g, 20000, "cuda:0")
    sp = train.self_play(g, rb, net, 48, seed=3, searches=5, batch=8, concurrent=16, stagger=True, net_mode="bf16x3")
    assert sp["games"] == 48 and sp["nodes"] > 0 and sp["steps"] >= 48 * 7 and len(rb) >= sp["steps"]
    st = train.self_play_stream(g, rb, net, 32, seed=3, searches=5, batch=8, concurrent=16, net_mode="bf16x3")
    assert st["games"] >= 32 and st["nodes"] > 0
    train.release_engines()
    net_hip.release_hipnets()


def test_self_play_stream_on_two_half_engines():
    """train.self_play_stream(streams=2): the slots as two engines on two HIP streams (full net tiles only), three
    consecutive calls: every call delivers complete games with unique uids out of the single engine's uid set, nothing
    overflows, the engine is kept between the calls; and the one-engine form still works beside it (its own cache key)."""
    import torch
    from caro_ai_amd import net_hip, train
    from caro_ai_amd.engine import StreamedSelfPlay
    from caro_ai_amd.lib.game.connect_four import ConnectFour
    from caro_ai_amd.lib.model import Net
    from caro_ai_amd.train import DeviceReplayBuffer
    g = ConnectFour()
    net = Net(g.obs_shape, g.action_space)
    net.load_state_dict(torch.load(os.path.join(WEIGHTS, "best_026_12000.dat"), map_location="cpu"))
    net = net.to("cuda:0").eval()
    rb = DeviceReplayBuffer(g, 40000, "cuda:0")
    seen = set()
    for call in range(3):
        st = train.self_play_stream(g, rb, net, 64, seed=5, searches=5, batch=8, concurrent=32, streams=2)
        assert st["games"] >= 64 and st["nodes"] > 0 and st["engine_reused"] == (call > 0)
    eng = next(e for e in train._ENGINES.values() if isinstance(e, StreamedSelfPlay))
    assert len(eng.parts) == 2 and eng.counters()["overflows"] == 0
    # uids: slot g of part k is global slot 16 k + g, stride 32 -- the single engine's layout
    assert [e.cfg.uid_base for e in eng.parts] == [eng.parts[0].cfg.uid_base, eng.parts[0].cfg.uid_base + 16]
    assert all(e.cfg.uid_stride == 32 for e in eng.parts)
    one = train.self_play_stream(g, rb, net, 32, seed=5, searches=5, batch=8, concurrent=32)
    assert one["games"] >= 32 and not one["engine_reused"]
    assert len(rb) > 0
    train.release_engines()
    net_hip.release_hipnets()
