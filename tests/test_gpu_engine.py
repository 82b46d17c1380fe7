"""GPU parity tests: the HIP engine, called through the C-ABI (libcaro_hip.so),
against the pinned oracle and the vectors recorded from the reference.

Bar: every integer (boards, actions, visit counts, z, results, counters) and
every float64 pi bit exact with the synthetic table net; W/Q float32 bit exact.
Run on the GPU box:  python -m pytest tests -m gpu
"""
import numpy as np
import pytest
import torch

from tests.conftest import load_golden

pytestmark = pytest.mark.gpu

DEV = "cuda:0"


def _game_of(d):
    from caro_ai_amd.lib.game.connect_four import ConnectFour
    from caro_ai_amd.lib.game.tictactoe import TicTacToe
    return ConnectFour() if d["kind"] == "c4" else TicTacToe(d["n"], d["k"])


def _oracle_of(d, n_stores=1):
    from oracle.oracle import Oracle
    return Oracle(Oracle.C4, n_stores=n_stores) if d["kind"] == "c4" else Oracle(Oracle.MNK, d["n"], d["k"],
                                                                                 n_stores=n_stores)


FORMS = ["stepwise", "fused"]


@pytest.fixture(params=FORMS)
def form(request):
    """How the table net is evaluated, i.e. which launches the engine runs.
    stepwise: the torch twin (tests/synth_net.py); leaf counts cross to the host, the engine issues
              k_select / k_encode / k_expand_backup one by one (the form torch evaluators use).
    fused:    the device table evaluator (`caro_net_create_hash`, leaf counts read on the device); a whole
              search_batch is enqueued by `caro_search_batch`: k_tree + slot rows when one wavefront serves a
              game (batch x lanes-per-descent = 64), dense rows otherwise -- the launches bench.py and train.py run."""
    return request.param


def _synth(game, form="stepwise", salt=0):
    if form == "fused":
        from caro_ai_amd.net_hip import HashNet
        return HashNet(game, device=DEV, salt=salt)
    from tests.synth_net import SynthNet
    return SynthNet(2 * game.obs_shape[1] * game.obs_shape[2], game.action_space, DEV, salt)


def _engine(game, G, evaluators, **kw):
    from caro_ai_amd.engine import SelfPlayEngine
    return SelfPlayEngine(game, G, evaluators=evaluators, device=DEV, **kw)


# ------------------------------------------------------------------ noise spec
@pytest.mark.parametrize("A", [7, 9, 25, 64, 100, 225])
def test_noise_device_equals_host_bits(A):
    from caro_ai_amd import _lib
    from oracle.oracle import noise_row
    L = _lib.load()
    M = 512
    rng = np.random.default_rng(A)
    uid = rng.integers(0, 2**62, M, dtype=np.uint64)
    ply = rng.integers(0, 225, M).astype(np.uint32)
    sim = rng.integers(0, 800, M).astype(np.uint32)
    d_uid = torch.from_numpy(uid.view(np.int64)).to(DEV)
    d_ply = torch.from_numpy(ply.view(np.int32)).to(DEV)
    d_sim = torch.from_numpy(sim.view(np.int32)).to(DEV)
    out = torch.zeros((M, A), dtype=torch.float64, device=DEV)
    _lib.check(L.caro_noise_batch(12345, M, A, 0.3, d_uid.data_ptr(), d_ply.data_ptr(), d_sim.data_ptr(),
                                  out.data_ptr(), None))
    torch.cuda.synchronize()
    got = out.cpu().numpy()
    exp = np.stack([noise_row(12345, int(uid[i]), int(ply[i]), int(sim[i]), A) for i in range(M)])
    assert np.array_equal(got.view(np.uint64), exp.view(np.uint64))
    assert np.all(got > 0) and np.allclose(got.sum(1), 1.0, atol=1e-12)


def test_visit_count_square_root_is_sqrtf_on_every_count():
    """the descent's sqrt(sum N) (lib/mcts.py:79, float32 as numpy rounds it) is a trimmed form of sqrtf: equal on every
    integer a visit count can be, 0 .. 2^24"""
    import ctypes as C
    from caro_ai_amd import _lib
    bad = C.c_uint64(12345)
    _lib.check(_lib.load().caro_debug_sqrt_check(1 << 24, C.byref(bad)))
    assert bad.value == 0


# ------------------------------------------------------------------ batched rules vs the reference's vectors
@pytest.mark.parametrize("name", ["rules_c4.json.gz", "rules_ttt3.json.gz", "rules_mnk5.json.gz",
                                  "rules_mnk15.json.gz"])
def test_rules_kernels_vs_reference(name):
    from caro_ai_amd import _lib
    L = _lib.load()
    d = load_golden(name)
    game = _game_of(d)
    recs = d["recs"]
    M, A, KW = len(recs), game.action_space, game.key_words
    HW = game.obs_shape[1] * game.obs_shape[2]
    keys = torch.from_numpy(game.to_keys([int(r["s"]) for r in recs]).view(np.int64)).to(DEV)
    moves = torch.tensor([r["m"] for r in recs], dtype=torch.int32, device=DEV)
    players = torch.tensor([r["p"] for r in recs], dtype=torch.int32, device=DEV)
    legal = torch.zeros((M, A), dtype=torch.uint8, device=DEV)
    _lib.check(L.caro_rules_legal_batch(game.kind, game.n, game.k, M, keys.data_ptr(), legal.data_ptr(), None))
    won = torch.zeros(M, dtype=torch.int32, device=DEV)
    full = torch.zeros(M, dtype=torch.int32, device=DEV)
    _lib.check(L.caro_rules_move_batch(game.kind, game.n, game.k, M, keys.data_ptr(), moves.data_ptr(),
                                       players.data_ptr(), won.data_ptr(), full.data_ptr(), None))
    who = (1 - players).contiguous()
    planes = torch.zeros((M, 2 * HW), dtype=torch.float32, device=DEV)
    _lib.check(L.caro_rules_encode_batch(game.kind, game.n, game.k, M, keys.data_ptr(), who.data_ptr(),
                                         planes.data_ptr(), None))
    torch.cuda.synchronize()
    legal, won, keys2, planes = legal.cpu().numpy(), won.cpu().numpy(), keys.cpu().numpy().view(np.uint64), planes.cpu().numpy()
    new_states = game.from_keys(keys2)
    for i, r in enumerate(recs):
        assert np.flatnonzero(legal[i]).tolist() == r["legal"]
        assert new_states[i] == int(r["s2"])
        assert bool(won[i]) == r["won"]
        assert np.packbits(planes[i].astype(np.uint8)).tobytes().hex() == r["planes"]


@pytest.mark.parametrize("name", ["c4", "ttt3", "mnk15", "mnk5", "mnk8", "mnk10"])
def test_rules_kernels_vs_reference_digests(name):
    """SURVEY 8(c) G1 at its stated size (tests/golden/make_golden_r5.py: 10^5 random connect-four plies, 10^4 each for
    3x3 and 15x15 k=5 played through the REFERENCE's game classes, one SHA-256 per 1000-ply block over next state, won,
    legal mask, planes).  The blocks are walked through this package's BaseGame shim (the host helpers of
    caro_rules.h); everything the digest absorbs comes from the batched rule KERNELS on all plies of the set at once."""
    import hashlib
    from caro_ai_amd import _lib
    from tests import rules_digest as rd
    L = _lib.load()
    d = load_golden("rules_digest.json.gz")
    s = d["sets"][name]
    game = _game_of(s)
    A, HW, nb, bl = game.action_space, game.obs_shape[1] * game.obs_shape[2], len(s["blocks"]), d["block"]
    recs, tallies = [], []
    for b in range(nb):
        tallies.append(rd.playout_block(game, d["seed"], b, bl, A, lambda s_, lg, m, p, s2, won: recs.append((s_, m, p))))
    M = len(recs)
    assert M == nb * bl == {"c4": 100000, "ttt3": 10000, "mnk15": 10000}.get(name, 5000)
    keys = torch.from_numpy(game.to_keys([r[0] for r in recs]).view(np.int64)).to(DEV)
    moves = torch.tensor([r[1] for r in recs], dtype=torch.int32, device=DEV)
    players = torch.tensor([r[2] for r in recs], dtype=torch.int32, device=DEV)
    legal = torch.zeros((M, A), dtype=torch.uint8, device=DEV)
    _lib.check(L.caro_rules_legal_batch(game.kind, game.n, game.k, M, keys.data_ptr(), legal.data_ptr(), None))
    won = torch.zeros(M, dtype=torch.int32, device=DEV)
    full = torch.zeros(M, dtype=torch.int32, device=DEV)
    _lib.check(L.caro_rules_move_batch(game.kind, game.n, game.k, M, keys.data_ptr(), moves.data_ptr(),
                                       players.data_ptr(), won.data_ptr(), full.data_ptr(), None))
    who = (1 - players).contiguous()
    planes = torch.zeros((M, 2 * HW), dtype=torch.float32, device=DEV)
    _lib.check(L.caro_rules_encode_batch(game.kind, game.n, game.k, M, keys.data_ptr(), who.data_ptr(),
                                         planes.data_ptr(), None))
    torch.cuda.synchronize()
    legal, won = legal.cpu().numpy(), won.cpu().numpy()
    planes = planes.cpu().numpy().astype(np.uint8)
    new_states = game.from_keys(keys.cpu().numpy().view(np.uint64))
    for b, want in enumerate(s["blocks"]):
        h = hashlib.sha256()
        for i in range(b * bl, (b + 1) * bl):
            rd.absorb(h, new_states[i], bool(won[i]), np.flatnonzero(legal[i]), A, planes[i])
        assert {"sha256": h.hexdigest(), "wins": tallies[b][0], "draws": tallies[b][1]} == want, (name, b)


# ------------------------------------------------------------------ engine vs recorded reference games
def _play_and_check_golden(d, g, form, explicit_noise=False, **engine_kw):
    from oracle.oracle import noise_row
    game = _game_of(d)
    A = game.action_space
    engine_kw.setdefault("node_cap", g["searches"] * g["batch"] * g["plies"] + 64)
    eng = _engine(game, 1, [_synth(game, form)], n_stores=g["n_stores"], max_batch=g["batch"], steps_before_tau_0=g["steps_before_tau_0"],
                  seed=g["seed"], uid_base=g["uid"], **engine_kw)
    eng.reset([g["first_player"]])
    S, B = g["searches"], g["batch"]
    for ply in range(g["plies"]):
        keys, players, plies, uid = eng.roots()
        assert str(game.from_key(keys[0])) == g["states"][ply]
        assert int(players[0]) == g["players"][ply] and int(plies[0]) == ply
        noise = None
        if explicit_noise:
            noise = np.stack([[[noise_row(g["seed"], g["uid"], ply, mb * B + b, A) for b in range(B)]]
                              for mb in range(S)])  # [S, G=1, B, A]
        eng.search(S, B, noise)
        store = g["players"][ply] if g["n_stores"] == 2 else 0
        nd = eng.lookup([0], [store], [int(g["states"][ply])])
        tr = g["trace"][ply]
        assert nd["found"][0] == 1
        assert nd["N"][0].tolist() == tr["N"], ply
        assert nd["W"][0].astype(np.float64).tolist() == tr["W"], ply  # float32 sums, queue order
        assert nd["strong"][0].tolist() == tr["W_f32"], ply
        # Q: float32 when strong, exact python ratio otherwise (the kernel recomputes it at the root)
        for a in range(A):
            if tr["W_f32"][a]:
                assert float(nd["Q"][0][a]) == tr["Q"][a]
            else:
                assert np.float32(tr["Q"][a]) == nd["Q"][0][a]
        assert eng.tree_sizes()[0][store] == tr["nodes"], ply
        pi, counts = eng.policy()
        assert pi[0].cpu().numpy().tolist() == g["pi"][ply], ply
        eng.step()
    dr = eng.drain(recycle=False)
    assert dr["games"].shape[0] == 1
    uid, first, result, steps = dr["games"][0].cpu().numpy().tolist()
    assert (uid, first, result, steps) == (g["uid"], g["first_player"], g["result"], g["steps"])
    n = g["plies"]
    assert dr["z"].cpu().numpy().tolist() == g["z"][::-1]
    assert dr["players"].cpu().numpy().tolist() == g["players"][::-1]
    assert [str(s) for s in game.from_keys(dr["states"].cpu().numpy().view(np.uint64))] == g["states"][::-1]
    assert dr["pi"].cpu().numpy().tolist() == g["pi"][::-1]
    assert n == dr["z"].shape[0]
    c = eng.counters()
    assert c["overflows"] == 0 and c["finished"] == 1 and c["plies"] == n
    eng.close()


@pytest.mark.parametrize("name", ["synth_c4.json.gz", "synth_ttt3.json.gz", "synth_mnk5.json.gz",
                                  "synth_mnk15.json.gz"])
def test_engine_replays_reference_games(name, form):
    """G2 on the GPU: the engine reproduces games recorded from the REFERENCE
    (synthetic table net, noise generated on device from the spec)."""
    d = load_golden(name)
    for g in d["games"]:
        _play_and_check_golden(d, g, form)


def test_engine_replays_reference_drawn_games(form):
    """whole games recorded from the reference that end in a DRAW (tests/golden/make_golden_r5.py; ref
    lib/utils.py:86-96, lib/mcts.py:144-146), one store and one store per player: root N / W / Q / nodes / pi / z"""
    d = load_golden("draws_ttt3.json.gz")
    assert len(d["games"]) >= 3
    for g in d["games"]:
        assert g["result"] == 0 and set(g["z"]) == {0}
        _play_and_check_golden(d, g, form)


def test_engine_replays_reference_games_on_mid_size_boards(form):
    """whole table-net games recorded from the reference on 6x6 k4, 8x8 k5 (one and two stores; 8x8 with batch 1 is a
    one-wavefront geometry: the fused tree kernel) and 10x10 k5 (two actions per lane): root N / W / Q / nodes / pi / z"""
    d = load_golden("synth_mid.json.gz")
    assert len(d["games"]) == 5
    for g in d["games"]:
        _play_and_check_golden({"kind": "mnk", "n": g["n"], "k": g["k"]}, g, form)


def test_config4_table_net_games_400_sims_with_eviction(form):
    """BASELINE config 4's per-game settings -- TicTacToe(15, 5), 50 x 8 = 400 sims/move, tau = 1 for 10 plies --
    with the engine as config 4 runs it: eviction on, 4 096 live nodes per tree.  Whole games recorded from the
    REFERENCE (65-75 plies, 6-8 k nodes created; tests/golden/make_golden_r3.py): root N / W / Q / dtype flag,
    nodes ever created, pi, moves, z, result -- all bit exact (ref lib/game/tictactoe/tictactoe.py:210-235,
    lib/utils.py:25-108)."""
    d = load_golden("synth_mnk15_400.json.gz")
    assert len(d["games"]) >= 2
    for g in d["games"]:
        assert (g["searches"], g["batch"]) == (50, 8) and g["plies"] > 30
        _play_and_check_golden(d, g, form, node_cap=4096, evict=True)


def test_explicit_noise_table_path(form):
    d = load_golden("synth_ttt3.json.gz")
    _play_and_check_golden(d, d["games"][2], form, explicit_noise=True)
    d = load_golden("synth_c4.json.gz")
    _play_and_check_golden(d, d["games"][2], form, explicit_noise=True)


# ------------------------------------------------------------------ many concurrent games vs the oracle
def _oracle_games(d, uids, seed, sbt0, S, B, n_stores, first_mode=2, salts=(0, 0)):
    out = {}
    for uid in uids:
        o = _oracle_of(d, n_stores)
        o.use_synth_net(*salts)
        o.set_stream(seed, int(uid))
        fp = int(uid) & 1 if first_mode == 2 else first_mode
        r = o.play_game(sbt0, S, B, fp)
        r["counters"] = o.counters()
        r["first"] = fp
        out[int(uid)] = r
    return out


def _check_against_oracle(d, G, n_finish, sbt0, S, B, n_stores, seed, uid_base, form="stepwise", salts=None,
                          one_call=False, dirty_first=None, **engine_kw):
    """salts = (s0, s1): player 0's leaves go to table net s0, player 1's to net s1 (n_nets = 2, play.py arena).
    one_call: every move through caro_search_move (search + ply from one call).  dirty_first = (seed, uid_base, moves):
    the engine first plays another run for so many moves and is then RESTARTED in place (caro_engine_restart) for the
    run that is checked."""
    game = _game_of(d)
    evs = [_synth(game, form)] if salts is None else [_synth(game, form, salts[0]), _synth(game, form, salts[1])]
    cap = S * B * game.obs_shape[1] * game.obs_shape[2] + 64
    if dirty_first is not None:
        eng = _engine(game, G, evs, n_stores=n_stores, max_batch=B, steps_before_tau_0=sbt0, seed=dirty_first[0],
                      uid_base=dirty_first[1], node_cap=cap, **engine_kw)
        eng.play_until(S, B, max_moves=dirty_first[2], one_call=one_call)
        eng.restart(seed=seed, uid_base=uid_base)
    else:
        eng = _engine(game, G, evs, n_stores=n_stores, max_batch=B, steps_before_tau_0=sbt0, seed=seed,
                      uid_base=uid_base, node_cap=cap, **engine_kw)
    tuples, games = eng.play_until(S, B, n_finished=n_finish, one_call=one_call)
    c = eng.counters()
    assert c["overflows"] == 0
    if engine_kw.get("games_limit"):  # exactly the wanted games: slot g's k-th game while k * G + g < games_limit
        lim = engine_kw["games_limit"]
        assert eng.live_games() == 0 and len(games) == lim and c["finished"] == lim
        assert sorted(games[:, 0].tolist()) == sorted(uid_base + (i % G) + (i // G) * G for i in range(lim))
    eng.close()
    uids = games[:, 0]
    assert len(set(uids.tolist())) == len(uids)
    ref = _oracle_games(d, uids, seed, sbt0, S, B, n_stores, salts=salts or (0, 0))
    # per-game records
    for uid, first, result, steps in games.tolist():
        r = ref[uid]
        assert (first, result, steps) == (r["first"], r["result"], r["steps"]), uid
    # tuples: each drain emits its games in slot order, plies last-to-first
    S_all = np.concatenate([t["states"] for t in tuples])
    P_all = np.concatenate([t["players"] for t in tuples])
    PI_all = np.concatenate([t["pi"] for t in tuples])
    Z_all = np.concatenate([t["z"] for t in tuples])
    off = 0
    for uid in uids.tolist():
        r = ref[uid]
        n = r["plies"]
        assert game.from_keys(S_all[off:off + n].view(np.uint64)) == r["states"][::-1], uid
        assert P_all[off:off + n].tolist() == r["players"][::-1].tolist()
        assert np.array_equal(PI_all[off:off + n], r["pi"][::-1]), uid
        assert Z_all[off:off + n].tolist() == r["z"][::-1].tolist()
        off += n
    assert off == len(Z_all)
    return c, ref, games


def test_connect4_64_games_vs_oracle(form):
    """64 concurrent connect-four games, 25x8 sims/move (config 2's S x B), slots
    recycled until 96 games finished: every game identical to the oracle's."""
    c, ref, games = _check_against_oracle({"kind": "c4"}, 64, 96, 10, 25, 8, 1, seed=3, uid_base=1000, form=form)
    assert len(games) >= 96


def test_connect4_arena_two_stores_vs_oracle(form):
    _check_against_oracle({"kind": "c4"}, 16, 16, 0, 10, 16, 2, seed=5, uid_base=5000, form=form)


def test_connect4_arena_two_nets_two_stores_vs_oracle(form):
    """config 5's shape (play.py:47): two different nets, one tree per player, tau = 0 from move 0; B = 8 is the
    one-wavefront-per-game geometry, so form=fused runs k_tree with both nets' rows in one launch"""
    _check_against_oracle({"kind": "c4"}, 32, 48, 0, 12, 8, 2, seed=6, uid_base=7000, form=form,
                          salts=(0x1111, 0x2222))
    _check_against_oracle({"kind": "c4"}, 8, 8, 0, 6, 16, 2, seed=7, uid_base=7100, form=form, salts=(3, 0))


def test_tictactoe_games_vs_oracle_with_draws(form):
    c, ref, games = _check_against_oracle({"kind": "mnk", "n": 3, "k": 3}, 64, 200, 2, 25, 4, 1, seed=9, uid_base=0,
                                          form=form)
    assert (games[:, 2] == 0).any(), "no drawn game in the sample"  # draw path covered


def test_gomoku15_games_vs_oracle(form):
    _check_against_oracle({"kind": "mnk", "n": 15, "k": 5}, 8, 8, 6, 6, 8, 1, seed=17, uid_base=40, form=form)


def test_mnk_mid_sizes_vs_oracle(form):
    _check_against_oracle({"kind": "mnk", "n": 6, "k": 4}, 8, 8, 3, 8, 8, 1, seed=21, uid_base=0, form=form)
    _check_against_oracle({"kind": "mnk", "n": 10, "k": 5}, 4, 4, 3, 6, 8, 1, seed=22, uid_base=0, form=form)


@pytest.mark.parametrize("d,B,two_nets", [({"kind": "mnk", "n": 4, "k": 3}, 4, False),    # 16 lanes x 4 descents
                                          ({"kind": "mnk", "n": 5, "k": 4}, 2, False),    # 32 x 2
                                          ({"kind": "mnk", "n": 8, "k": 5}, 1, False),    # 64 x 1
                                          ({"kind": "mnk", "n": 10, "k": 5}, 1, False),   # 64 x 1, 2 actions per lane
                                          ({"kind": "mnk", "n": 15, "k": 5}, 1, True),    # 64 x 1, 4 actions per lane
                                          ({"kind": "mnk", "n": 3, "k": 3}, 4, True)])
def test_one_wavefront_geometries_vs_oracle(d, B, two_nets, form):
    """every geometry in which caro_search_batch runs the fused tree kernel (batch x lanes-per-descent = 64),
    with one and with two nets"""
    n = d["n"]
    S = 24 // B if n <= 5 else 10
    _check_against_oracle(d, 8, 12 if n <= 5 else 8, 2, S, B, 2 if two_nets else 1, seed=80 + n, uid_base=300, form=form,
                          salts=(9, 10) if two_nets else None)


def test_counters_match_oracle_totals(form):
    """sims / levels / expansions / terminals / dropped summed over complete
    games equal the oracle's (the roofline's byte formula is built on these)."""
    d = {"kind": "c4"}
    game = _game_of(d)
    G, S, B = 32, 25, 8
    eng = _engine(game, G, [_synth(game, form)], max_batch=B, steps_before_tau_0=10, seed=77, uid_base=0)
    tuples, games = eng.play_until(S, B, recycle=False)
    c = eng.counters()
    eng.close()
    assert len(games) == G and c["finished"] == G
    ref = _oracle_games(d, range(G), 77, 10, S, B, 1)
    tot = {k: sum(r["counters"][k] for r in ref.values()) for k in ["sims", "levels", "expansions", "terminals", "dropped"]}
    # finished games keep idling in lock-step runs only until every game is done; they add no sims
    for k in tot:
        assert c[k] == tot[k], k
    assert c["plies"] == sum(r["plies"] for r in ref.values())


def test_full_size_1024_games_invariants(form):
    """Config 2 geometry (1024 games, 25x8): size-independent properties."""
    d = {"kind": "c4"}
    game = _game_of(d)
    G, S, B = 1024, 25, 8
    eng = _engine(game, G, [_synth(game, form)], max_batch=B, steps_before_tau_0=10, seed=1, uid_base=0)
    for _ in range(3):
        eng.search(S, B)
        pi, counts = eng.policy()
        cn = counts.cpu().numpy()
        # root visit counts: every sim except the dropped duplicates and the first expansion is backed up
        assert (cn.sum(1) <= S * B * 3).all() and (cn.sum(1) > 0).all()
        p = pi.cpu().numpy()
        assert np.allclose(p.sum(1), 1.0)
        eng.step()
    c = eng.counters()
    assert c["sims"] == 3 * G * S * B and c["overflows"] == 0
    assert c["expansions"] + c["terminals"] + c["dropped"] == c["sims"]
    # a sample of the 1024 games equals the oracle ply by ply
    keys, players, plies, uid = eng.roots()
    from oracle.oracle import Oracle
    for gidx in [0, 1, 511, 1023]:
        o = Oracle(Oracle.C4)
        o.use_synth_net()
        o.set_stream(1, gidx)
        s, pl = o.initial_state, gidx & 1
        for ply in range(3):
            o.search_batch(S, B, s, pl, ply=ply)
            prob = o.get_policy(s, 1)
            from oracle.oracle import move_uniform, sample_index
            a = sample_index(prob, move_uniform(1, gidx, ply))
            s, won = o.move(s, a, pl)
            pl = 1 - pl
            assert not won
        assert game.from_key(keys[gidx]) == s
    eng.close()


def test_config4_full_size_whole_games_with_eviction():
    """BASELINE config 4 at its real size: 1024 concurrent 15 x 15 k = 5 games, 50 x 8 sims/move, eviction with a
    4 096-node cap, slots recycled until >= 1024 games have finished (table net on the device: the launches
    are bench.py's, only the evaluator is the exact one).  Size-independent properties -- nothing overflows at
    any point of a whole game, the sims identity holds, every finished game is a legal game -- and a sample of
    the finished games equals the oracle's game of the same uid ply by ply (states, pi, z, result, steps)."""
    d = {"kind": "mnk", "n": 15, "k": 5}
    game = _game_of(d)
    G, S, B, seed = 1024, 50, 8, 9
    eng = _engine(game, G, [_synth(game, "fused")], max_batch=B, steps_before_tau_0=10, seed=seed, uid_base=0,
                  node_cap=4096, evict=True, searches_hint=S)
    kept, games, moves, finished = {}, [], 0, 0
    live_max = 0
    while finished < 1024:
        eng.search(S, B)
        if moves % 8 == 0:
            live_max = max(live_max, int(eng.tree_live().max()))  # before the move's eviction: the peak
        eng.step()
        moves += 1
        dr = eng.drain(recycle=True)
        ng = dr["games"].shape[0]
        if ng:
            gr = dr["games"].cpu().numpy()
            finished += ng
            games.append(gr)
            off = 0
            z, st = dr["z"].cpu().numpy(), None
            for uid, first, result, steps in gr.tolist():
                n = steps + 1
                # z alternates back from the last mover: +1 / -1 for a win, all 0 for a draw (utils.py:101-106)
                zz = z[off:off + n]
                assert zz[0] == (1 if result != 0 else 0) and (np.abs(zz) == abs(int(zz[0]))).all()
                if uid % 64 == 0:  # first and later generations of a slot alike
                    st = dr["states"].cpu().numpy() if st is None else st
                    kept[uid] = (first, result, steps, st[off:off + n].copy(),
                                 dr["pi"][off:off + n].cpu().numpy(), zz.copy())
                off += n
            assert off == z.shape[0]
        assert moves < 600
    c = eng.counters()
    eng.close()
    games = np.concatenate(games)
    assert c["overflows"] == 0 and live_max <= 4096
    assert c["sims"] == moves * G * S * B
    assert c["expansions"] + c["terminals"] + c["dropped"] == c["sims"]
    assert c["finished"] == len(games) >= 1024 and len(set(games[:, 0].tolist())) == len(games)
    assert (games[:, 3] >= 8).all() and (games[:, 3] < 225).all()  # k = 5: no game ends before the 9th ply
    print("config 4 full size: %d moves, %d games finished, peak live nodes %d, sims %d, expansions %d"
          % (moves, len(games), live_max, c["sims"], c["expansions"]))
    assert len(kept) >= 12
    ref = _oracle_games(d, kept.keys(), seed, 10, S, B, 1)
    for uid, (first, result, steps, st, pi, z) in kept.items():
        r = ref[uid]
        assert (first, result, steps) == (r["first"], r["result"], r["steps"]), uid
        assert game.from_keys(st.view(np.uint64)) == r["states"][::-1], uid
        assert np.array_equal(pi, r["pi"][::-1]), uid
        assert z.tolist() == r["z"][::-1].tolist(), uid


# ------------------------------------------------------------------ node eviction is result-neutral
def _check_evict(d, G, n_finish, sbt0, S, B, n_stores, seed, uid_base, cap, form):
    game = _game_of(d)
    eng = _engine(game, G, [_synth(game, form)], n_stores=n_stores, max_batch=B, steps_before_tau_0=sbt0, seed=seed,
                  uid_base=uid_base, node_cap=cap, evict=True)
    tuples, games = eng.play_until(S, B, n_finished=n_finish)
    c = eng.counters()
    eng.close()
    assert c["overflows"] == 0
    ref = _oracle_games(d, games[:, 0], seed, sbt0, S, B, n_stores)
    exp_total = 0
    for uid, first, result, steps in games.tolist():
        r = ref[uid]
        assert (first, result, steps) == (r["first"], r["result"], r["steps"]), uid
    PI_all = np.concatenate([t["pi"] for t in tuples])
    off = 0
    for uid in games[:, 0].tolist():
        r = ref[uid]
        assert np.array_equal(PI_all[off:off + r["plies"]], r["pi"][::-1]), uid
        off += r["plies"]
    return c


def test_eviction_connect4_small_cap(form):
    """cap 1024 live nodes per tree is far below what a whole game creates (~600-2600) once trees are shared
    across 20+ plies without eviction: with eviction nothing overflows and every game still equals the oracle."""
    _check_evict({"kind": "c4"}, 32, 48, 10, 25, 8, 1, seed=31, uid_base=0, cap=1024, form=form)
    _check_evict({"kind": "c4"}, 8, 8, 0, 10, 16, 2, seed=32, uid_base=100, cap=512, form=form)


def test_eviction_gomoku15_and_ttt(form):
    _check_evict({"kind": "mnk", "n": 15, "k": 5}, 4, 4, 6, 6, 8, 1, seed=33, uid_base=0, cap=256, form=form)
    _check_evict({"kind": "mnk", "n": 15, "k": 5}, 4, 4, 4, 12, 1, 1, seed=35, uid_base=0, cap=256, form=form)
    _check_evict({"kind": "mnk", "n": 3, "k": 3}, 32, 64, 2, 25, 4, 1, seed=34, uid_base=0, cap=128, form=form)


def test_without_eviction_the_small_cap_overflows():
    game = _game_of({"kind": "c4"})
    eng = _engine(game, 8, [_synth(game)], max_batch=8, steps_before_tau_0=10, seed=31, node_cap=128)
    eng.play_until(25, 8, max_moves=12, recycle=False)
    assert eng.counters()["overflows"] > 0  # counted, never silent
    eng.close()


# ------------------------------------------------------------------ sharding does not change the games
def test_games_do_not_depend_on_sharding(form):
    """8 slots in one engine == 2 'ranks' of 4 slots (parallel.shard layout) == StreamedSelfPlay with 2 parts:
    every uid gives the same game (result, steps, tuples), whatever plays it."""
    from caro_ai_amd import parallel
    from caro_ai_amd.engine import StreamedSelfPlay
    game = _game_of({"kind": "c4"})
    S, B, seed = 10, 8, 41

    def collect(tuples, games):
        out, off = {}, 0
        PI = np.concatenate([t["pi"] for t in tuples]); ST = np.concatenate([t["states"] for t in tuples])
        for uid, first, result, steps in games.tolist():
            n = steps + 1
            out[uid] = (first, result, steps, ST[off:off + n].tobytes(), PI[off:off + n].tobytes())
            off += n
        return out

    eng = _engine(game, 8, [_synth(game, form)], max_batch=B, seed=seed)
    t, g = eng.play_until(S, B, n_finished=16)
    eng.close()
    ref = collect(t, g)
    got = {}
    for rank in range(2):
        e = _engine(game, 4, [_synth(game, form)], max_batch=B, seed=seed, **parallel.shard(4, rank, 2))
        t, g = e.play_until(S, B, n_finished=8)
        e.close()
        got.update(collect(t, g))
    common = set(ref) & set(got)
    assert len(common) >= 12
    for uid in common:
        assert ref[uid] == got[uid], uid
    sp = StreamedSelfPlay(game, 8, lambda: [_synth(game, form)], n_streams=2, max_batch=B, seed=seed)
    seen = {}
    for _ in range(40):
        sp.search(S, B); sp.step()
        d = sp.drain()
        if d["games"].shape[0]:
            seen.update(collect([{k: v.cpu().numpy() for k, v in d.items() if k != "games"}], d["games"].cpu().numpy()))
    sp.close()
    common = set(ref) & set(seen)
    assert len(common) >= 8
    for uid in common:
        assert ref[uid] == seen[uid], uid


# ------------------------------------------------------------------ searches from recorded mid / late game positions
def _search_from_positions(d, recs, S, B, seed, form):
    """One engine slot per recorded position (set_roots on empty trees), S x B sims, then root N / W / Q /
    strong flag / tree size / pi against the oracle searching the same position with the same noise key."""
    from oracle.oracle import Oracle
    game = _game_of(d)
    states = [int(r["s2"]) for r in recs]
    players = [1 - r["p"] for r in recs]
    G = len(states)
    eng = _engine(game, G, [_synth(game, form)], max_batch=B, steps_before_tau_0=10, seed=seed, uid_base=0,
                  node_cap=S * B + 8)
    eng.set_roots(states, players)
    eng.search(S, B)
    nd = eng.lookup(list(range(G)), [0] * G, states)
    sizes = eng.tree_sizes()[:, 0]
    pi = eng.policy()[0].cpu().numpy()
    assert eng.counters()["overflows"] == 0
    eng.close()
    o = _oracle_of(d)
    o.use_synth_net()
    checked = terminal_roots = 0
    for g in range(G):
        if not game.possible_moves(states[g]):
            continue  # a full board is never searched by play_game
        o.clear()
        o.set_stream(seed, g)
        o.search_batch(S, B, states[g], players[g], ply=0)
        ref = o.get_node(states[g])
        assert nd["found"][g] == 1
        assert nd["N"][g].tolist() == ref["N"].tolist(), (g, states[g])
        assert nd["W"][g].astype(np.float64).tolist() == ref["W"].tolist(), g
        assert nd["strong"][g].tolist() == ref["W_is_f32"].tolist(), g
        for a in range(game.action_space):
            if ref["W_is_f32"][a]:
                assert float(nd["Q"][g][a]) == ref["Q"][a]
            else:
                assert np.float32(ref["Q"][a]) == nd["Q"][g][a]
        assert sizes[g] == o.store_len(0), g
        if ref["N"].sum() > 0:
            assert pi[g].tolist() == o.get_policy(states[g], 1).tolist()
        checked += 1
        terminal_roots += int(o.counters()["terminals"] > 0)
    return checked, terminal_roots


def test_connect4_search_from_2000_recorded_positions(form):
    d = load_golden("rules_c4.json.gz")
    recs = [r for r in d["recs"] if not r["won"]][:2000]
    checked, with_terminals = _search_from_positions({"kind": "c4"}, recs, 6, 8, seed=51, form=form)
    assert checked >= 1900 and with_terminals > 200  # late-game roots: wins, full columns and draws inside the search


def test_tictactoe_and_gomoku_search_from_recorded_positions(form):
    d = load_golden("rules_ttt3.json.gz")
    recs = [r for r in d["recs"] if not r["won"]]
    checked, with_terminals = _search_from_positions({"kind": "mnk", "n": 3, "k": 3}, recs, 8, 4, seed=52, form=form)
    assert checked > 700 and with_terminals > 300
    d = load_golden("rules_mnk15.json.gz")
    recs = [r for r in d["recs"] if not r["won"]][::4]
    checked, _ = _search_from_positions({"kind": "mnk", "n": 15, "k": 5}, recs, 3, 8, seed=53, form=form)
    assert checked > 100
    d = load_golden("rules_mnk5.json.gz")
    recs = [r for r in d["recs"] if not r["won"]]
    checked, with_terminals = _search_from_positions({"kind": "mnk", "n": 5, "k": 4}, recs, 5, 8, seed=54, form=form)
    checked, with_terminals = _search_from_positions({"kind": "mnk", "n": 5, "k": 4}, recs, 12, 2, seed=55, form=form)
    assert checked > 500
    assert checked > 500 and with_terminals > 20


@pytest.mark.parametrize("B", [1, 2, 16, 64])
def test_batch_size_edge_cases_connect4(B, form):
    """mcts_batch_size from 1 (TicTacToe plumbing config) to 64 (the kernel's block = 64 x 8 lanes)"""
    _check_against_oracle({"kind": "c4"}, 8, 8, 4, 200 // B if B <= 16 else 3, B, 1, seed=60 + B, uid_base=0, form=form)


@pytest.mark.parametrize("n,k", [(4, 3), (8, 5), (11, 5), (12, 6), (4, 4)])
def test_mnk_geometry_variants(n, k, form):
    """every lane geometry: A = 16 (no padding lanes), 64 (full wave), 121 / 144 (2 and 4 actions per lane), k = n"""
    _check_against_oracle({"kind": "mnk", "n": n, "k": k}, 4, 4, 3, 5, 8, 1, seed=70 + n, uid_base=0, form=form)


def test_capi_rejects_bad_arguments():
    import ctypes as C
    from caro_ai_amd import _lib
    L = _lib.load()
    h = C.c_void_p()

    def cfg(**kw):
        c = _lib.CaroConfig()
        c.game_kind, c.n, c.k, c.n_games, c.n_stores, c.n_nets, c.max_batch, c.node_cap = 0, 0, 0, 4, 1, 1, 8, 64
        c.c_puct, c.alpha, c.explore = 1.0, 0.3, 0.25
        for a, b in kw.items():
            setattr(c, a, b)
        return c

    for bad in (dict(n_games=0), dict(n_stores=3), dict(n_nets=0), dict(max_batch=0), dict(max_batch=200),
                dict(game_kind=1, n=16, k=5), dict(game_kind=1, n=5, k=6), dict(game_kind=7), dict(device_id=99)):
        c = cfg(**bad)
        assert L.caro_engine_create(C.byref(c), C.byref(h)) == -22, bad  # CARO_E_INVAL
        assert L.caro_last_error()
    c = cfg()
    assert L.caro_engine_create(C.byref(c), C.byref(h)) == 0
    planes = torch.zeros((32, 2, 6, 7), device=DEV)
    assert L.caro_select(h, 9, 0, None, planes.data_ptr(), None, None) == -22       # batch > max_batch
    assert L.caro_expand_backup(h, planes.data_ptr(), planes.data_ptr(), None) == -71  # no pending select
    assert L.caro_select(h, 8, 0, None, planes.data_ptr(), None, None) == 0
    assert L.caro_select(h, 8, 1, None, planes.data_ptr(), None, None) == -71       # select twice
    assert L.caro_step(h, None, None, None, None, None) == -71
    assert L.caro_select_cancel(h) == 0
    torch.cuda.synchronize()
    L.caro_engine_destroy(h)


# ------------------------------------------------------------------ G1 at scale: 1e5 random plies, device rules vs oracle
@pytest.mark.parametrize("d,n_games", [({"kind": "c4"}, 3000), ({"kind": "mnk", "n": 3, "k": 3}, 2500),
                                       ({"kind": "mnk", "n": 7, "k": 4}, 300), ({"kind": "mnk", "n": 15, "k": 5}, 60)])
def test_rules_kernels_vs_oracle_random_playouts(d, n_games):
    """Random playouts generated with the oracle's rules (CPU); every transition is then replayed through the
    batched device kernels: next state, won, board-full, legal mask and NN planes must all agree."""
    from caro_ai_amd import _lib
    L = _lib.load()
    game = _game_of(d)
    o = _oracle_of(d)
    rng = np.random.default_rng(99)
    S, M_, P_, S2, WON, FULL = [], [], [], [], [], []
    for _ in range(n_games):
        s, p = o.initial_state, int(rng.integers(2))
        while True:
            legal = o.possible_moves(s)
            if not legal:
                break
            m = int(legal[int(rng.integers(len(legal)))])
            s2, won = o.move(s, m, p)
            S.append(s); M_.append(m); P_.append(p); S2.append(s2); WON.append(won)
            FULL.append(len(o.possible_moves(s2)) == 0)
            if won:
                break
            s, p = s2, 1 - p
    M = len(S)
    A, HW = game.action_space, game.obs_shape[1] * game.obs_shape[2]
    keys = torch.from_numpy(game.to_keys(S).view(np.int64)).to(DEV)
    moves = torch.tensor(M_, dtype=torch.int32, device=DEV)
    players = torch.tensor(P_, dtype=torch.int32, device=DEV)
    legal = torch.zeros((M, A), dtype=torch.uint8, device=DEV)
    _lib.check(L.caro_rules_legal_batch(game.kind, game.n, game.k, M, keys.data_ptr(), legal.data_ptr(), None))
    won = torch.zeros(M, dtype=torch.int32, device=DEV)
    full = torch.zeros(M, dtype=torch.int32, device=DEV)
    _lib.check(L.caro_rules_move_batch(game.kind, game.n, game.k, M, keys.data_ptr(), moves.data_ptr(),
                                       players.data_ptr(), won.data_ptr(), full.data_ptr(), None))
    who = (1 - players).contiguous()
    planes = torch.zeros((M, 2 * HW), dtype=torch.float32, device=DEV)
    _lib.check(L.caro_rules_encode_batch(game.kind, game.n, game.k, M, keys.data_ptr(), who.data_ptr(),
                                         planes.data_ptr(), None))
    torch.cuda.synchronize()
    assert np.array_equal(keys.cpu().numpy().view(np.uint64), game.to_keys(S2))
    assert won.cpu().numpy().astype(bool).tolist() == WON
    assert full.cpu().numpy().astype(bool).tolist() == FULL
    lg = legal.cpu().numpy()
    pl = planes.cpu().numpy()
    idx = rng.choice(M, size=min(M, 3000), replace=False)  # the per-state oracle calls are the slow part
    for i in idx:
        assert np.flatnonzero(lg[i]).tolist() == o.possible_moves(S[i])
        ref = o.states_to_training_batch([S2[i]], [1 - P_[i]])[0].reshape(-1)
        assert np.array_equal(pl[i], ref)
    assert M > (50000 if d["kind"] == "c4" else 3000)


def test_engine_create_refuses_tables_the_path_records_cannot_address():
    """path records carry the node slot in 24 bits; a node_cap that needs more than 2^24 slots per tree is
    refused instead of aliasing nodes"""
    from caro_ai_amd import _lib
    from caro_ai_amd.lib.game.connect_four import ConnectFour
    with pytest.raises(_lib.CaroError, match="node_cap too large"):
        _engine(ConnectFour(), 1, [_synth(ConnectFour())], node_cap=(1 << 23) + 1)
