"""Replay tuples on their way out of the engine, and run-to-run reproducibility.

* A drain hands out tensors of its own: tuples kept across later drains (what `parallel.TupleGatherer` and
  `train.self_play` do) must not change when the engine's staging buffer is rewritten.
* The replay rows `train.self_play` stores are exactly the drained tuples, in drain order.
* Two runs of the same seeded configuration give the same bits, at the full 1024-game size where the
  net kernel mixes full and K-split tiles (the tile a leaf meets is a function of the games' states only).
"""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

DEV = "cuda:0"


def _host(d):
    return {k: v.cpu().numpy().copy() for k, v in d.items()}


def test_drained_tuples_survive_later_drains_and_the_gatherer_keeps_them():
    from caro_ai_amd import parallel
    from caro_ai_amd.engine import SelfPlayEngine
    from caro_ai_amd.lib.game.tictactoe import TicTacToe
    from caro_ai_amd.net_hip import HashNet
    from oracle.oracle import Oracle
    game = TicTacToe()
    S, B, seed = 6, 4, 13
    eng = SelfPlayEngine(game, 32, evaluators=[HashNet(game, device=DEV)], max_batch=B, steps_before_tau_0=2, seed=seed,
                         device=DEV)
    tg = parallel.TupleGatherer(every=7)
    kept, copies, gathered = [], [], []
    for _ in range(30):
        eng.search(S, B)
        eng.step()
        d = eng.drain(recycle=True)
        kept.append(d)                      # device tensors, NOT cloned by the caller
        copies.append(_host(d))             # what they held when they were handed out
        out = tg.push(d)
        if out is not None:
            gathered.append(_host(out))
    out = tg.flush()
    if out is not None:
        gathered.append(_host(out))
    eng.close()
    n_rows = sum(c["z"].shape[0] for c in copies)
    assert n_rows > 200 and sum(c["z"].shape[0] > 0 for c in copies) > 10  # many non-empty drains
    for d, c in zip(kept, copies):
        for k in c:
            np.testing.assert_array_equal(d[k].cpu().numpy(), c[k], err_msg=k)
    for k in ("states", "players", "z"):
        np.testing.assert_array_equal(np.concatenate([g[k] for g in gathered]), np.concatenate([c[k] for c in copies]))
    np.testing.assert_array_equal(np.concatenate([g["pi"] for g in gathered]),
                                  np.concatenate([c["pi"] for c in copies]).astype(np.float32))
    # and the rows are the games the oracle plays for the same uids
    off = 0
    S_all = np.concatenate([c["states"] for c in copies])
    Z_all = np.concatenate([c["z"] for c in copies])
    PI_all = np.concatenate([c["pi"] for c in copies])
    for c in copies:
        for uid, first, result, steps in c["games"].tolist():
            o = Oracle(Oracle.MNK, 3, 3)
            o.use_synth_net()
            o.set_stream(seed, int(uid))
            r = o.play_game(2, S, B, int(uid) & 1)
            n = r["plies"]
            assert (result, steps) == (r["result"], r["steps"])
            assert game.from_keys(S_all[off:off + n].view(np.uint64)) == r["states"][::-1]
            assert Z_all[off:off + n].tolist() == r["z"][::-1].tolist()
            assert np.array_equal(PI_all[off:off + n], r["pi"][::-1])
            off += n
    assert off == n_rows


def test_pipelined_move_hands_out_the_same_rows_one_move_later():
    """SelfPlayEngine.move() (caro_drain_tuples_begin / _end: this move's drain is enqueued, the PREVIOUS move's
    totals are collected while the GPU searches) against the plain search / step / drain loop: the same drains,
    in the same order, one move later; flush() hands out the last one.  Calling _end without _begin, or _begin
    twice, is refused."""
    from caro_ai_amd import _lib
    from caro_ai_amd.engine import SelfPlayEngine
    from caro_ai_amd.lib.game.tictactoe import TicTacToe
    from caro_ai_amd.net_hip import HashNet
    game = TicTacToe()
    S, B, seed, moves = 6, 4, 21, 24

    def make():
        return SelfPlayEngine(game, 48, evaluators=[HashNet(game, device=DEV)], max_batch=B, steps_before_tau_0=2,
                              seed=seed, device=DEV)

    eng = make()
    plain = []
    for _ in range(moves):
        eng.search(S, B)
        eng.step()
        plain.append(_host(eng.drain(recycle=True)))
    c_plain = eng.counters()
    eng.close()
    eng = make()
    piped = []
    for i in range(moves):
        out = eng.move(S, B)
        assert (out is None) == (i == 0)
        if out is not None:
            piped.append(_host(out))
    with pytest.raises(_lib.CaroError):
        eng.drain_begin()  # the last move's drain is still open
    piped.append(_host(eng.flush()))
    assert eng.flush() is None
    with pytest.raises(_lib.CaroError):
        eng.drain_end()
    assert eng.counters() == c_plain
    eng.close()
    assert len(piped) == len(plain) and sum(p["z"].shape[0] for p in plain) > 200
    for a, b in zip(plain, piped):
        for k in a:
            np.testing.assert_array_equal(a[k], b[k], err_msg=k)


def test_self_play_replay_rows_are_the_drained_tuples():
    from caro_ai_amd import config as cfg
    from caro_ai_amd import train
    from caro_ai_amd.engine import SelfPlayEngine
    from caro_ai_amd.lib.game.tictactoe import TicTacToe
    from caro_ai_amd.lib.model import Net
    game = TicTacToe()
    torch.manual_seed(3)
    net = Net(game.obs_shape, game.action_space).to(DEV).eval()
    n_games, G, S, B = 96, 32, 5, 4
    rb = train.DeviceReplayBuffer(game, 4096, DEV)
    sp = train.self_play(game, rb, net, n_games, device=DEV, seed=5, uid_base=0, searches=S, batch=B, concurrent=G)
    assert sp["games"] == n_games
    # the same games again, every drain copied to the host at once.  self_play's rule: the wanted games are uids
    # 0 .. n_games - 1 (slot g plays g, g + G, g + 2G); drained slots restart while some slot still has a wanted
    # generation to begin; games beyond the wanted set are played but their rows dropped
    eng = SelfPlayEngine(game, G, net1=net, max_batch=B, steps_before_tau_0=cfg.STEPS_BEFORE_TAU_0, seed=5, device=DEV,
                         searches_hint=S, uid_base=0, uid_stride=G)
    rows, finished = [], 0
    slot_gen = np.zeros(G, dtype=np.int64)
    last_gen = (n_games - 1 - np.arange(G)) // G
    while finished < n_games:
        eng.search(S, B)
        eng.step()
        d = eng.drain(recycle=bool((slot_gen < last_gen).any()))
        ng = int(d["games"].shape[0])
        if ng:
            h = _host(d)
            recs = h["games"]
            want = recs[:, 0] < n_games
            np.maximum.at(slot_gen, recs[want, 0] % G, recs[want, 0] // G + 1)
            finished += int(want.sum())
            keep = np.repeat(want, recs[:, 3] + 1)
            rows.append({k: h[k][keep] for k in ("states", "players", "pi", "z")})
        elif eng.live_games() == 0:
            break
    eng.close()
    n = sum(r["z"].shape[0] for r in rows)
    assert len(rb) == n and n > 5 * n_games
    np.testing.assert_array_equal(rb.states[:n].cpu().numpy(), np.concatenate([r["states"] for r in rows]))
    np.testing.assert_array_equal(rb.players[:n].cpu().numpy(), np.concatenate([r["players"] for r in rows]))
    np.testing.assert_array_equal(rb.z[:n].cpu().numpy(), np.concatenate([r["z"] for r in rows]).astype(np.float32))
    np.testing.assert_array_equal(rb.pi[:n].cpu().numpy(), np.concatenate([r["pi"] for r in rows]).astype(np.float32))
    # several different games, not one drain repeated
    assert len(set(map(bytes, np.concatenate([r["states"] for r in rows])))) > n // 4


@pytest.mark.parametrize("stagger", [False, True])
@pytest.mark.parametrize("inference", ["hipw", "hip"])
def test_same_seed_same_bits_at_1024_games(inference, stagger):
    """config 2 size: full and K-split net tiles are both in play (lock-step: the first minibatches of a move carry
    more leaves than one round of full tiles; staggered -- bench.py's launches --: every launch sits at the edge of one
    round); every root row, pi and replay row must still repeat bit for bit"""
    import os
    from caro_ai_amd.engine import SelfPlayEngine
    from caro_ai_amd.lib.game.connect_four import ConnectFour
    from caro_ai_amd.lib.model import Net
    from tests.conftest import GOLDEN
    game = ConnectFour()
    net = Net(game.obs_shape, game.action_space)
    net.load_state_dict(torch.load(os.path.join(GOLDEN, "weights", "best_026_12000.dat"), map_location="cpu"))
    net = net.to(DEV).eval()
    sample = list(range(0, 1024, 7))

    def run():
        eng = SelfPlayEngine(game, 1024, net1=net, max_batch=8, seed=99, device=DEV, searches_hint=25,
                             inference=inference, stagger=stagger)
        out = []
        if stagger:
            # the plies happen inside the launches: what can be compared are the replay rows of the games that finish
            # (every ply's board and float64 pi), the game records, and the trees' root rows at the end
            n_games = 0
            for _ in range(30):
                eng.search(25, 8)
                d = eng.drain()
                n_games += int(d["games"].shape[0])
                out.append(tuple(d[k].cpu().numpy().tobytes() for k in ("states", "players", "pi", "z", "games")))
            assert n_games >= 1024
            passes = 30
        else:
            passes = 3
        for _ in range(0 if stagger else 3):
            eng.search(25, 8)
            pi, counts = eng.policy()
            keys = eng.roots()[0]
            nd = eng.lookup(sample, [0] * len(sample), [game.from_key(k) for k in keys[::7]])
            out.append((pi.cpu().numpy().tobytes(), counts.cpu().numpy().tobytes(), nd["W"].tobytes(),
                        nd["P"].tobytes()))
            eng.step()
        keys, players, plies, uids = eng.roots()
        nd = eng.lookup(sample, [0] * len(sample), [game.from_key(k) for k in keys[::7]])
        out.append((keys.tobytes(), plies.tobytes(), uids.tobytes(), nd["N"].tobytes(), nd["W"].tobytes(),
                    nd["P"].tobytes()))
        c = eng.counters()
        eng.close()
        return out, c, passes

    a, ca, passes = run()
    b, cb, _ = run()
    assert ca == cb and ca["overflows"] == 0
    # launches around one round of full tiles (256 x 6 boards); staggered: the games sit out g % 25 launches at the start
    assert ca["expansions"] / (passes * 25) > 1536 * (0.85 if stagger else 0.9)
    for x, y in zip(a, b):
        assert x == y


def test_same_seed_same_bits_15x15_conv_net_in_the_engine():
    """config 4's launches (2-D Winograd net kernel on slot... dense rows of the step-wise tree kernels, eviction on) are
    deterministic: two runs of 64 games x 3 moves at 50 x 8 give the same root rows, pi and counters bit for bit"""
    from caro_ai_amd.engine import SelfPlayEngine
    from caro_ai_amd.lib.game.tictactoe import TicTacToe
    from caro_ai_amd.lib.model import Net
    game = TicTacToe(15, 5)
    torch.manual_seed(0)
    net = Net(game.obs_shape, game.action_space).to(DEV).eval()

    def run():
        eng = SelfPlayEngine(game, 64, net1=net, max_batch=8, seed=5, device=DEV, searches_hint=50, node_cap=4096,
                             evict=True, inference="hipw")
        assert eng.evaluators[0].mode == "f32w2"
        out = []
        for _ in range(3):
            eng.search(50, 8)
            pi, counts = eng.policy()
            out.append((pi.cpu().numpy().tobytes(), counts.cpu().numpy().tobytes()))
            eng.step()
        c = eng.counters()
        eng.close()
        return out, c

    a, ca = run()
    b, cb = run()
    assert ca == cb and ca["overflows"] == 0 and ca["expansions"] > 0.8 * ca["sims"]
    assert a == b


def test_kept_drains_do_not_pin_the_staging_block_on_large_boards(monkeypatch):
    """a drain's rows are views of a staging block sized for every game finishing at its longest
    (G * H*W rows); on 15 x 15 that is 1.8 KB x 225 x G per drain, and whoever keeps the tuples of every move
    (TupleGatherer, a replay buffer) would pin all of it.  Above `VIEW_LIMIT_BYTES` a drain hands out copies of the
    finished rows: kept drains own what they hold, memory stays bounded, and the rows are the plain path's."""
    from caro_ai_amd.engine import SelfPlayEngine
    from caro_ai_amd.lib.game.tictactoe import TicTacToe
    from caro_ai_amd.net_hip import HashNet
    game = TicTacToe(15, 3)  # k = 3 on the 15 x 15 board: 225 actions per row as config 4, games that end early and
    G, S, B = 64, 2, 8       # at different times (a few per move: the case in which a kept view pins the most)

    def run(limit):
        if limit is not None:
            monkeypatch.setattr(SelfPlayEngine, "VIEW_LIMIT_BYTES", limit)
        eng = SelfPlayEngine(game, G, evaluators=[HashNet(game, device=DEV)], max_batch=B, steps_before_tau_0=400,
                             seed=5, device=DEV, node_cap=2048)
        torch.cuda.synchronize()
        base = torch.cuda.memory_allocated()
        kept, peak = [], 0
        for _ in range(120):
            eng.search(S, B)
            eng.step()
            kept.append(eng.drain(recycle=True))
            peak = max(peak, torch.cuda.memory_allocated() - base)
        rows = sum(int(d["z"].shape[0]) for d in kept)
        held = sum({v.untyped_storage().data_ptr(): v.untyped_storage().nbytes()
                    for d in kept for v in d.values() if v.numel()}.values())
        out = [_host(d) for d in kept]
        eng.close()
        return rows, held, peak, out

    staging = G * 225 * (8 * game.key_words + 8 * 225 + 8)
    assert staging > SelfPlayEngine.VIEW_LIMIT_BYTES
    rows, held, peak, out = run(None)
    assert rows > 300 and sum(d["z"].shape[0] > 0 for d in out) >= 5
    payload = rows * (8 * game.key_words + 8 * 225 + 8)
    assert held < 2 * payload + (1 << 20), (held, payload)          # kept drains own their rows, not 260 staging blocks
    assert peak < 3 * staging + 2 * payload, (peak, staging, payload)
    rows_v, held_v, _, out_v = run(1 << 40)                            # the view form, for comparison: same rows ...
    assert rows_v == rows
    for a, b in zip(out, out_v):
        for k in a:
            np.testing.assert_array_equal(a[k], b[k], err_msg=k)
    assert held_v > 10 * held                                         # ... and every non-empty drain pins its block


def test_gomoku15_games_do_not_depend_on_the_stream_split():
    """bench.py's config-4 leg plays its 1024 games as two engines of 512 on two HIP streams sharing ONE net handle
    (per-stream feature rows).  The 15 x 15 net kernel evaluates one board per workgroup, so a leaf's priors do not
    depend on which launch or tile carries it: 16 games as one engine == the same uids as 2 x 8 on two streams, root
    boards and visit counts bit for bit at every move, conv net (2-D Winograd form), eviction on as config 4 runs."""
    from caro_ai_amd.engine import SelfPlayEngine, StreamedSelfPlay
    from caro_ai_amd.lib.game.tictactoe import TicTacToe
    from caro_ai_amd.lib.model import Net
    from caro_ai_amd.net_hip import HipNet
    game = TicTacToe(15, 5)
    torch.manual_seed(0)
    net = Net(game.obs_shape, game.action_space).to(DEV).eval()
    hip = HipNet(net, DEV)
    assert hip.mode == "f32w2"
    kw = dict(max_batch=8, steps_before_tau_0=10, seed=3, device=DEV, searches_hint=6, evict=True, node_cap=1024)
    one = SelfPlayEngine(game, 16, evaluators=[hip], **kw)
    two = StreamedSelfPlay(game, 16, lambda: [hip], n_streams=2, partition_cus=False, **kw)
    for mv in range(12):
        one.search(6, 8)
        two.search(6, 8)
        torch.cuda.synchronize()
        pi1, n1 = one.policy()
        n2 = []
        for e, st in two._each():
            with torch.cuda.stream(st):
                n2.append(e.policy()[1])
        torch.cuda.synchronize()
        assert torch.equal(n1, torch.cat(n2)), mv
        one.step()
        two.step()
        torch.cuda.synchronize()
        k1 = one.roots()[0]
        k2 = np.concatenate([e.roots()[0] for e in two.parts])
        assert np.array_equal(np.asarray(k1), k2), mv
    assert int(n1.sum()) > 16 * 40
    c1, c2 = one.counters(), [e.counters() for e in two.parts]
    assert c1["expansions"] == sum(c["expansions"] for c in c2) and c1["overflows"] == 0
    one.close()
    two.close()
