"""GPU tests of the reference-API shim (lib.mcts.MCTS, lib.utils.play_game /
play_games), written the way the reference's own tests are."""
import collections
import os

import numpy as np
import pytest
import torch

from tests.conftest import GOLDEN, load_golden

pytestmark = pytest.mark.gpu


@pytest.fixture
def tree():
    """lib/test_mcts.py:8-22 on a real game: three chained tic-tac-toe states, actions 0 and 1 used."""
    from caro_ai_amd.lib.game.tictactoe import TicTacToe
    from caro_ai_amd.lib.mcts import MCTS
    game = TicTacToe()
    t = MCTS(game)
    s1 = game.initial_state
    s2, _ = game.move(s1, 1, 0)
    s3, _ = game.move(s2, 0, 1)
    t.visit_count = {s1: [0, 1], s2: [1, 0], s3: [0, 0]}
    t.value = {s1: [0.0, 0.5], s2: [0.6, 0.0], s3: [0.0, 0.0]}
    t.value_avg = {s1: [0.0, 0.5], s2: [0.6, 0.0], s3: [0.0, 0.0]}
    t.probs = {s1: [0.1, 0.9], s2: [0.8, 0.2], s3: [0.7, 0.3]}
    return t, (s1, s2, s3)


class TestBackup:
    def test_back_up(self, tree):
        t, (s1, s2, s3) = tree
        assert len(t) == 3
        t._backup(0.2, [s1, s2, s3], [1, 0, 0])
        vc, val, avg = t.visit_count, t.value, t.value_avg
        assert [vc[s][:2] for s in (s1, s2, s3)] == [[0, 2], [2, 0], [1, 0]]
        # the tree stores W and Q as float32 (poked python floats are rounded): 1e-7 relative
        np.testing.assert_allclose([val[s][:2] for s in (s1, s2, s3)], [[0.0, 0.3], [0.8, 0.0], [-0.2, 0.0]],
                                   rtol=2e-7, atol=1e-8)
        np.testing.assert_allclose([avg[s][:2] for s in (s1, s2, s3)], [[0.0, 0.15], [0.4, 0.0], [-0.2, 0.0]],
                                   rtol=2e-7, atol=1e-8)
        np.testing.assert_allclose(t.probs[s2][:2], [0.8, 0.2], rtol=1e-7)
        t.clear()
        assert len(t) == 0 and t.is_leaf(s1)


def _game_for(d):
    from caro_ai_amd.lib.game.connect_four import ConnectFour
    from caro_ai_amd.lib.game.tictactoe import TicTacToe
    return ConnectFour() if d["kind"] == "c4" else TicTacToe(d["n"], d["k"])


def _synth_module(game):
    """A lib.model.Net whose forward is the synthetic table net (logits = log P so softmax returns ~P)."""
    from caro_ai_amd.lib.model import Net
    from tests.synth_net import SynthNet

    class M(Net):
        def forward(self, x):
            P, v = SynthNet(int(np.prod(x.shape[1:])), self.actions_n, x.device)(x)
            return torch.log(P), v.reshape(-1, 1)

    return M(game.obs_shape, game.action_space)


def test_find_leaf_and_first_visit_order():
    """Q4/Q6: unexpanded root returns itself with an empty path; two sims from the empty
    connect-four board give N = [1,0,...] (first legal action wins all-equal scores)."""
    from caro_ai_amd.lib.game.connect_four import ConnectFour
    from caro_ai_amd.lib.mcts import MCTS
    g = ConnectFour()
    net = _synth_module(g)
    t = MCTS(g)
    s = g.initial_state
    value, leaf, player, states, actions = t.find_leaf(s, 0)
    assert value is None and leaf == s and player == 0 and states == [] and actions == []
    assert len(t) == 0
    np.random.seed(0)
    t.search_minibatch(1, s, 0, net, device="cuda:0")
    assert len(t) == 1 and not t.is_leaf(s)
    t.search_minibatch(1, s, 0, net, device="cuda:0")
    assert t.visit_count[s] == [1, 0, 0, 0, 0, 0, 0]
    value, leaf, player, states, actions = t.find_leaf(s, 0)
    assert states[0] == s and len(states) == len(actions) >= 1 and player in (0, 1)
    probs, values = t.get_policy_value(s, tau=0)
    assert probs == [1.0, 0, 0, 0, 0, 0, 0] and len(values) == 7


def test_play_game_numpy_stream_matches_oracle():
    """play_game consumes numpy's global RNG exactly like the reference (dirichlet per descent, choice per
    ply): replaying the recorded draws through the oracle's explicit tables gives the same game."""
    from caro_ai_amd.lib.game.connect_four import ConnectFour
    from caro_ai_amd.lib.utils import play_game
    from oracle.oracle import Oracle
    g = ConnectFour()
    net = _synth_module(g)
    rec_noise, rec_u = [], []
    real_dir, real_choice = np.random.dirichlet, np.random.choice

    def dir_(alpha):
        r = real_dir(alpha)
        rec_noise.append(r)
        return r

    def choice_(a, p=None):
        if p is None:
            return real_choice(a)
        u = np.random.random()
        rec_u.append(u)
        cdf = np.cumsum(np.asarray(p, dtype=np.float64))
        cdf /= cdf[-1]
        return int(np.searchsorted(cdf, u, side="right"))

    np.random.dirichlet, np.random.choice = dir_, choice_
    try:
        np.random.seed(123)
        rb = collections.deque()
        r, steps = play_game(g, None, rb, net, net, 4, 6, 8, net1_plays_first=True, device="cuda:0")
    finally:
        np.random.dirichlet, np.random.choice = real_dir, real_choice
    # synthetic P passes through log + softmax on the GPU: compare with the oracle fed the SAME softmaxed P
    from tests.synth_net import SynthNet

    def fn(planes, states, players):
        P, v = SynthNet(84, 7, "cuda:0")(torch.from_numpy(np.ascontiguousarray(planes)).to("cuda:0"))
        return torch.softmax(torch.log(P), dim=1).cpu().numpy(), v.cpu().numpy()

    o = Oracle(Oracle.C4, n_stores=2)
    o.set_net(0, fn)
    o.set_net(1, fn)
    o.set_noise_table(np.array(rec_noise))
    o.set_uniform_table(np.array(rec_u))
    ref = o.play_game(4, 6, 8, 0)
    assert (r, steps) == (ref["result"], ref["steps"])
    hist = list(rb)[::-1]
    assert [h[0] for h in hist] == ref["states"]
    assert [h[1] for h in hist] == ref["players"].tolist()
    assert [h[3] for h in hist] == ref["z"].tolist()
    assert np.array_equal(np.array([h[2] for h in hist]), ref["pi"])
    assert o.noise_pos() == len(rec_noise)


def test_play_game_fused_path_numpy_stream_matches_oracle():
    """The low-latency form of the single-game API: a real `Net` in eval mode on the GPU makes
    `MCTS.search_batch` take ONE `caro_search_batch` call per move (k_tree -> fused net kernel) with the Dirichlet
    rows of the whole search pre-drawn from numpy.  The draws must be the reference's in number and order
    (none for a minibatch that meets an unexpanded root, lib/mcts.py:123,131): replaying the recorded draws
    through the oracle -- fed by the same HIP net on the same leaf boards -- gives the same game bit for bit."""
    from caro_ai_amd.lib.game.connect_four import ConnectFour
    from caro_ai_amd.lib.mcts import MCTS
    from caro_ai_amd.lib.utils import play_game
    from caro_ai_amd.net_hip import HipNet
    from oracle.oracle import Oracle
    g = ConnectFour()
    net, = _real_nets(g, ["best_026_12000.dat"])
    rec_noise, rec_u = [], []
    real_dir, real_choice = np.random.dirichlet, np.random.choice

    def dir_(alpha, size=None):
        r = real_dir(alpha, size)
        rec_noise.extend(np.atleast_2d(r))
        return r

    def choice_(a, p=None):
        if p is None:
            return real_choice(a)
        u = np.random.random()
        rec_u.append(u)
        cdf = np.cumsum(np.asarray(p, dtype=np.float64))
        cdf /= cdf[-1]
        return int(np.searchsorted(cdf, u, side="right"))

    stores = [MCTS(g), MCTS(g)]
    np.random.dirichlet, np.random.choice = dir_, choice_
    try:
        np.random.seed(321)
        rb = collections.deque()
        r, steps = play_game(g, stores, rb, net, net, 4, 10, 8, net1_plays_first=False, device="cuda:0")
    finally:
        np.random.dirichlet, np.random.choice = real_dir, real_choice
    assert all(len(t._hip) == 1 for t in stores), "the fused path was not taken"
    hip = HipNet(net, "cuda:0")

    def fn(planes, states, players):
        P, v = hip(torch.from_numpy(np.ascontiguousarray(planes)).to("cuda:0"))
        return P.cpu().numpy(), v.cpu().numpy()

    o = Oracle(Oracle.C4, n_stores=2)
    o.set_net(0, fn)
    o.set_net(1, fn)
    o.set_noise_table(np.array(rec_noise))
    o.set_uniform_table(np.array(rec_u))
    ref = o.play_game(4, 10, 8, 1)
    assert (r, steps) == (ref["result"], ref["steps"])
    hist = list(rb)[::-1]
    assert [h[0] for h in hist] == ref["states"]
    assert [h[1] for h in hist] == ref["players"].tolist()
    assert [h[3] for h in hist] == ref["z"].tolist()
    assert np.array_equal(np.array([h[2] for h in hist]), ref["pi"])
    assert o.noise_pos() == len(rec_noise)  # every pre-drawn row is one the reference would have drawn
    assert sum(len(t) for t in stores) == o.store_len(0) + o.store_len(1)


def test_fused_path_follows_weight_updates_and_falls_back():
    """HipNet cache: rebuilt when the weights change in place; train-mode nets, CPU devices and subclasses with
    their own forward keep the step-wise (reference-sequence) path"""
    from caro_ai_amd.lib.game.connect_four import ConnectFour
    from caro_ai_amd.lib.mcts import MCTS
    g = ConnectFour()
    net, = _real_nets(g, ["best_026_12000.dat"])
    t = MCTS(g)
    s = g.initial_state
    np.random.seed(1)
    t.search_batch(3, 8, s, 0, net, device="cuda:0")
    first = t._hip[id(net)][1]
    t.search_batch(3, 8, s, 0, net, device="cuda:0")
    assert t._hip[id(net)][1] is first
    p_before = t.probs[s].copy()
    with torch.no_grad():
        net.policy[0].bias[0] += 1000.0  # the shipped net's logits on the empty board are hundreds apart
    t.clear()
    t.search_batch(3, 8, s, 0, net, device="cuda:0")
    assert t._hip[id(net)][1] is not first
    assert p_before[0] < 0.01 and t.probs[s][0] > 0.99
    t2 = MCTS(g)
    assert t2._fused_net(net.train(), "cuda:0") is None
    assert t2._fused_net(net.eval(), "cpu") is None
    assert t2._fused_net(_synth_module(g).eval(), "cuda:0") is None


def test_play_games_fills_replay_buffer_like_the_reference():
    from caro_ai_amd.lib.game.tictactoe import TicTacToe
    from caro_ai_amd.lib.model import Net
    from caro_ai_amd.lib.utils import play_games
    g = TicTacToe()
    net = Net(g.obs_shape, g.action_space)
    net.load_state_dict(torch.load(os.path.join(GOLDEN, "weights", "best_005_00900.dat"), map_location="cpu"))
    rb = collections.deque(maxlen=5000)
    results, stats = play_games(g, 40, rb, net, steps_before_tau_0=10, mcts_searches=10, mcts_batch_size=8,
                                concurrent=16, seed=5, return_stats=True)
    assert len(results) == 40 and set(results) <= {1, 0, -1}
    assert len(rb) >= 40 * 5
    for state, player, pi, z in rb:
        assert isinstance(state, int) and player in (0, 1) and z in (1, 0, -1)
        assert len(pi) == 9 and abs(sum(pi) - 1.0) < 1e-9
        legal = g.possible_moves(state)
        assert all(p == 0 for a, p in enumerate(pi) if a not in legal)
    assert stats["speed_nodes"] > 0 and stats["counters"]["overflows"] == 0


def test_arena_real_weights_statistics():
    """Config 5 shape, reduced: best_026 vs best_025, tau = 0, two stores, two nets."""
    from caro_ai_amd.lib.game.connect_four import ConnectFour
    from caro_ai_amd.lib.model import Net
    from caro_ai_amd.lib.utils import play_games
    g = ConnectFour()
    nets = []
    for w in ("best_026_12000.dat", "best_025_10600.dat"):
        n = Net(g.obs_shape, g.action_space)
        n.load_state_dict(torch.load(os.path.join(GOLDEN, "weights", w), map_location="cpu"))
        nets.append(n)
    res, stats = play_games(g, 32, None, nets[0], nets[1], steps_before_tau_0=0, mcts_searches=10,
                            mcts_batch_size=8, concurrent=32, seed=29, uid_base=700, return_stats=True)
    assert len(res) == 32 and stats["counters"]["overflows"] == 0
    w, l, d = res.count(1), res.count(-1), res.count(0)
    assert w + l + d == 32


def _cpu_torch_evaluator(net):
    """The reference's own net arithmetic: torch CPU float32 forward + F.softmax on the leaf batch
    (lib/mcts.py:212-218), results copied back to the GPU tree."""
    net = net.cpu().eval()

    def fn(planes):
        with torch.no_grad():
            lg, vl = net(planes.cpu())
            return (torch.softmax(lg, dim=1).to(planes.device).contiguous(),
                    vl[:, 0].to(planes.device).contiguous())

    return fn


@pytest.mark.parametrize("name", ["real_c4.json.gz", "real_ttt3.json.gz", "arena_c4.json.gz"])
def test_real_weights_exact_with_reference_net_arithmetic(name):
    """G3 / G5 / BASELINE config 1, exact: HIP tree walk + the reference's net arithmetic (same torch CPU
    forward on the same leaf batches) reproduces the games recorded from the reference bit for bit --
    root N every ply, pi (float64), actions, z, result, steps."""
    from caro_ai_amd.engine import SelfPlayEngine
    from caro_ai_amd.lib.game.connect_four import ConnectFour
    from caro_ai_amd.lib.game.tictactoe import TicTacToe
    from caro_ai_amd.lib.model import Net
    torch.set_num_threads(1)
    d = load_golden(name)
    g = ConnectFour() if d["kind"] == "c4" else TicTacToe(d["n"], d["k"])
    ws = d["weights"] if isinstance(d["weights"], list) else [d["weights"]]
    nets = []
    for w in ws:
        n = Net(g.obs_shape, g.action_space)
        n.load_state_dict(torch.load(os.path.join(GOLDEN, "weights", w), map_location="cpu"))
        nets.append(n)
    for gm in d["games"]:
        eng = SelfPlayEngine(g, 1, evaluators=[_cpu_torch_evaluator(n) for n in nets], n_stores=gm["n_stores"],
                             max_batch=gm["batch"], steps_before_tau_0=gm["steps_before_tau_0"], seed=gm["seed"],
                             uid_base=gm["uid"], node_cap=4096)
        eng.reset([gm["first_player"]])
        for ply in range(gm["plies"]):
            assert str(g.from_key(eng.roots()[0][0])) == gm["states"][ply]
            eng.search(gm["searches"], gm["batch"])
            pi, counts = eng.policy()
            assert counts[0].cpu().tolist() == gm["trace"][ply]["N"], (gm["uid"], ply)
            assert pi[0].cpu().numpy().tolist() == gm["pi"][ply]
            eng.step()
        dr = eng.drain(recycle=False)
        uid, first, result, steps = dr["games"][0].cpu().tolist()
        assert (result, steps) == (gm["result"], gm["steps"])
        assert dr["z"].cpu().tolist() == gm["z"][::-1]
        eng.close()


def _real_nets(g, names):
    from caro_ai_amd.lib.model import Net
    nets = []
    for w in names:
        n = Net(g.obs_shape, g.action_space)
        n.load_state_dict(torch.load(os.path.join(GOLDEN, "weights", w), map_location="cpu"))
        nets.append(n.to("cuda:0").eval())
    return nets


def _compare_with_recorded_games(d, eng, g, n_moves):
    """All recorded games of `d` run as ONE engine (slot i = game i).  A game is compared ply by ply for as long as
    its root is the recorded one (PUCT argmax is discontinuous: once a different move is played the tree persists
    (Q2) and later plies are another game).  Returns (plies compared, plies with the reference's root N vector,
    max |d pi| on the others, games that followed the recorded moves to the end, mismatch log)."""
    games = d["games"]
    G = len(games)
    alive = [True] * G
    exact = [True] * G  # the reference's N at every ply so far => the same pi => the same sampled moves
    total = same = 0
    max_dpi = 0.0
    log = []
    for ply in range(n_moves):
        keys = eng.roots()[0]
        for i, gm in enumerate(games):
            if alive[i] and ply < gm["plies"] and str(g.from_key(keys[i])) != gm["states"][ply]:
                alive[i] = False
        eng.search(games[0]["searches"], games[0]["batch"])
        pi, counts = eng.policy()
        pi, counts = pi.cpu().numpy(), counts.cpu().numpy()
        for i, gm in enumerate(games):
            if not alive[i] or ply >= gm["plies"]:
                continue
            total += 1
            if counts[i].tolist() == gm["trace"][ply]["N"]:
                same += 1
                assert pi[i].tolist() == gm["pi"][ply]  # same N, same tau => the same float64 pi
            else:
                exact[i] = False
                log.append("uid %d ply %d: N=%s ref=%s" % (gm["uid"], ply, counts[i].tolist(), gm["trace"][ply]["N"]))
                max_dpi = max(max_dpi, float(np.abs(pi[i] - np.array(gm["pi"][ply])).max()))
        eng.step()
    dr = eng.drain(recycle=False)
    recs = {int(r[0]): r for r in dr["games"].cpu().numpy().tolist()}
    followed = 0
    for i, gm in enumerate(games):
        r = recs.get(gm["uid"])
        ended_alike = r is not None and (r[1], r[2], r[3]) == (gm["first_player"], gm["result"], gm["steps"])
        if alive[i] and exact[i]:  # identical search results at every ply: the whole game must be identical
            assert ended_alike, (gm["uid"], r)
        followed += int(alive[i] and ended_alike)
    return total, same, max_dpi, followed, log


# the net arithmetic under test: fused HIP kernel (row-Winograd; the same forced onto full 128-row tiles, the
# tile bench.py's 1024-game launches use; direct form) and the torch GEMM form
NET_FORMS = ["hipw", "hipw-fulltiles", "hip", "gemm", "hipx3"]  # hipx3: the extra bf16x3 form under the same tolerance


@pytest.mark.parametrize("inference", NET_FORMS)
def test_real_weights_gpu_net_32_games(inference, monkeypatch):
    """G3 at BASELINE config 2's per-game settings (25 x 8 sims/move, tau = 1 for 10 plies, shipped
    best_026_12000.dat): 32 games recorded from the reference (tests/golden/make_golden_r2.py) against the engine
    with the net on the GPU, one 32-game engine through `caro_search_batch`.  Stated tolerance:
    >= 99 % of the compared plies carry the reference's root visit vector; whatever differs stays within
    |d pi| <= 0.15; every game that reproduced all recorded moves ends with the recorded result and step count."""
    from caro_ai_amd.engine import SelfPlayEngine
    from caro_ai_amd.lib.game.connect_four import ConnectFour
    if inference == "hipw-fulltiles":
        monkeypatch.setenv("CARO_NO_SPLIT_TILES", "1")
        inference = "hipw"
    d = load_golden("real_c4_x32.json.gz")
    g = ConnectFour()
    net, = _real_nets(g, [d["weights"]])
    gm0 = d["games"][0]
    assert [gm["uid"] for gm in d["games"]] == list(range(gm0["uid"], gm0["uid"] + 32))
    eng = SelfPlayEngine(g, 32, net1=net, max_batch=gm0["batch"], inference=inference, seed=gm0["seed"],
                         steps_before_tau_0=gm0["steps_before_tau_0"], uid_base=gm0["uid"], first_player_mode=2)
    assert eng.async_net == (inference != "gemm")
    total, same, max_dpi, followed, log = _compare_with_recorded_games(d, eng, g, max(gm["plies"] for gm in d["games"]))
    eng.close()
    print("\n".join(log))
    print("%s: identical root-N plies %d / %d (%.2f %%), max |dpi| elsewhere %.4f, %d / 32 games followed to the end"
          % (inference, same, total, 100.0 * same / total, max_dpi, followed))
    assert total >= 400
    assert same / total >= 0.99 and max_dpi <= 0.15
    assert followed >= 24


@pytest.mark.parametrize("inference", ["hipw", "hip", "hipx3"])
def test_real_weights_gpu_net_arena_800_sims(inference):
    """G5 at BASELINE config 5's per-game settings: best_026 vs best_025, 100 x 8 sims/move, tau = 0 from move 0,
    one tree per player; 8 games recorded from the reference, both nets in one launch (two-net k_tree)."""
    from caro_ai_amd.engine import SelfPlayEngine
    from caro_ai_amd.lib.game.connect_four import ConnectFour
    d = load_golden("arena_c4_800.json.gz")
    g = ConnectFour()
    n1, n2 = _real_nets(g, d["weights"])
    gm0 = d["games"][0]
    eng = SelfPlayEngine(g, 8, net1=n1, net2=n2, n_stores=2, max_batch=8, inference=inference, seed=gm0["seed"],
                         steps_before_tau_0=0, uid_base=gm0["uid"], first_player_mode=2, searches_hint=100)
    total, same, max_dpi, followed, log = _compare_with_recorded_games(d, eng, g, max(gm["plies"] for gm in d["games"]))
    eng.close()
    print("\n".join(log))
    print("%s arena: identical root-N plies %d / %d, max |dpi| elsewhere %.4f, %d / 8 games followed to the end"
          % (inference, same, total, max_dpi, followed))
    assert total >= 100 and same / total >= 0.99 and max_dpi <= 0.15
    # tau = 0: the move is the argmax of N, so a ply with the reference's N plays the reference's move
    assert followed >= 6


@pytest.mark.parametrize("name,inference", [("arena_c4_320_x16.json.gz", "hipw"), ("arena_c4_800_x16.json.gz", "hipw"),
                                            ("arena_c4_800_x16.json.gz", "hip"), ("arena_c4_320_x16.json.gz", "hipx3"),
                                            ("arena_c4_800_x16.json.gz", "hipx3")])
def test_arena_32_recorded_games_gpu_net(name, inference):
    """SURVEY 8(c) G5 (tests/golden/make_golden_r5.py): 16 + 16 seeded tau = 0 arena games best_026 vs best_025 recorded
    from the reference at play.py's 40 x 8 and config 5's 100 x 8 sims/move, all 16 of a set in one engine, both nets
    in one launch.  Stated tolerance as above; and the W / L / D tally is the reference's whenever every game
    followed the recorded moves."""
    from caro_ai_amd.engine import SelfPlayEngine
    from caro_ai_amd.lib.game.connect_four import ConnectFour
    d = load_golden(name)
    g = ConnectFour()
    n1, n2 = _real_nets(g, d["weights"])
    gm0 = d["games"][0]
    assert [gm["uid"] for gm in d["games"]] == list(range(gm0["uid"], gm0["uid"] + 16))
    eng = SelfPlayEngine(g, 16, net1=n1, net2=n2, n_stores=2, max_batch=8, inference=inference, seed=gm0["seed"],
                         steps_before_tau_0=0, uid_base=gm0["uid"], first_player_mode=2, searches_hint=gm0["searches"])
    total, same, max_dpi, followed, log = _compare_with_recorded_games(d, eng, g, max(gm["plies"] for gm in d["games"]))
    eng.close()
    print("\n".join(log))
    print("%s %s: identical root-N plies %d / %d, max |dpi| elsewhere %.4f, %d / 16 games followed to the end"
          % (name, inference, same, total, max_dpi, followed))
    assert total >= 200 and same / total >= 0.99 and max_dpi <= 0.15
    assert followed >= 14


def test_arena_32_recorded_games_exact_with_reference_net_arithmetic():
    """the same 32 games, the first four of each set EXACTLY: HIP tree walk (two stores, two evaluators) + the
    reference's net arithmetic (torch CPU forward on the same leaf batches) -- root N, pi, z, result, steps"""
    from caro_ai_amd.engine import SelfPlayEngine
    from caro_ai_amd.lib.game.connect_four import ConnectFour
    from caro_ai_amd.lib.model import Net
    torch.set_num_threads(1)
    g = ConnectFour()
    for name in ("arena_c4_320_x16.json.gz", "arena_c4_800_x16.json.gz"):
        d = load_golden(name)
        nets = []
        for w in d["weights"]:
            n = Net(g.obs_shape, g.action_space)
            n.load_state_dict(torch.load(os.path.join(GOLDEN, "weights", w), map_location="cpu"))
            nets.append(n)
        res = []
        for gm in d["games"][:4]:
            eng = SelfPlayEngine(g, 1, evaluators=[_cpu_torch_evaluator(n) for n in nets], n_stores=2,
                                 max_batch=gm["batch"], steps_before_tau_0=0, seed=gm["seed"], uid_base=gm["uid"],
                                 node_cap=gm["searches"] * gm["batch"] * gm["plies"] + 64)
            eng.reset([gm["first_player"]])
            for ply in range(gm["plies"]):
                assert str(g.from_key(eng.roots()[0][0])) == gm["states"][ply]
                eng.search(gm["searches"], gm["batch"])
                pi, counts = eng.policy()
                assert counts[0].cpu().tolist() == gm["trace"][ply]["N"], (gm["uid"], ply)
                assert pi[0].cpu().numpy().tolist() == gm["pi"][ply]
                eng.step()
            dr = eng.drain(recycle=False)
            uid, first, result, steps = dr["games"][0].cpu().tolist()
            assert (result, steps) == (gm["result"], gm["steps"])
            assert dr["z"].cpu().tolist() == gm["z"][::-1]
            eng.close()


# ------------------------------------------------------------------ BASELINE config 4: 15 x 15, k = 5, 50 x 8 sims/move
def _seeded_net_15(d):
    """the repo's own Net under the committed seed -- the state_dict the reference's Net was
    loaded with when tests/golden/make_golden_r3.py recorded the games (the fixture's SHA-256 proves it)"""
    from tests.test_oracle_golden import seeded_net_15, state_dict_sha256
    net = seeded_net_15(d["weights_seed"])
    assert state_dict_sha256(net.state_dict()) == d["weights_sha256"]
    return net


def test_config4_conv_net_exact_with_reference_net_arithmetic():
    """G3 at config 4's per-game settings, exact: the reference's TicTacToe(15, 5) games at 50 x 8 sims/move with
    the conv net (ref lib/game/tictactoe/tictactoe.py:210-235, lib/utils.py:25-108) against the HIP tree walk --
    eviction on, 4 096 live nodes per tree, as config 4 runs -- with the reference's net arithmetic (the same torch
    CPU forward on the same leaf batches): root N every ply, pi (float64), z, result, steps bit for bit."""
    from caro_ai_amd.engine import SelfPlayEngine
    from caro_ai_amd.lib.game.tictactoe import TicTacToe
    torch.set_num_threads(1)
    d = load_golden("real_mnk15.json.gz")
    g = TicTacToe(d["n"], d["k"])
    net = _seeded_net_15(d)
    for gm in d["games"][:3]:
        assert (gm["searches"], gm["batch"]) == (50, 8)
        eng = SelfPlayEngine(g, 1, evaluators=[_cpu_torch_evaluator(net)], n_stores=gm["n_stores"],
                             max_batch=gm["batch"], steps_before_tau_0=gm["steps_before_tau_0"], seed=gm["seed"],
                             uid_base=gm["uid"], node_cap=4096, evict=True)
        eng.reset([gm["first_player"]])
        for ply in range(gm["plies"]):
            assert str(g.from_key(eng.roots()[0][0])) == gm["states"][ply]
            eng.search(gm["searches"], gm["batch"])
            pi, counts = eng.policy()
            assert counts[0].cpu().tolist() == gm["trace"][ply]["N"], (gm["uid"], ply)
            assert pi[0].cpu().numpy().tolist() == gm["pi"][ply]
            assert eng.tree_sizes()[0][0] == gm["trace"][ply]["nodes"]  # len(MCTS): nodes ever created
            assert eng.tree_live()[0][0] <= 4096
            eng.step()
        dr = eng.drain(recycle=False)
        uid, first, result, steps = dr["games"][0].cpu().tolist()
        assert (result, steps) == (gm["result"], gm["steps"])
        assert dr["z"].cpu().tolist() == gm["z"][::-1]
        assert eng.counters()["overflows"] == 0
        eng.close()


@pytest.mark.parametrize("inference", ["hipw", "hipw1", "hip", "hipx3"])  # hipx3: the opt-in bf16x3 form (+ k_net_heads here)
def test_config4_conv_net_on_gpu_vs_reference_games(inference):
    """The same recorded games against the engine with the net ON THE GPU (fused HIP kernel: hipw = the 2-D Winograd
    form F(2x2,3x3) that config 4 runs from round 4 on, hipw1 = the row form, hip = direct convolutions),
    all games in one engine, eviction + 4 096-node cap.  Stated tolerance as for Connect4: >= 99 %
    of the compared plies carry the reference's root visit vector, |d pi| <= 0.15 elsewhere, and a game that
    matched at every ply ends with the recorded result and step count."""
    from caro_ai_amd.engine import SelfPlayEngine
    from caro_ai_amd.lib.game.tictactoe import TicTacToe
    d = load_golden("real_mnk15.json.gz")
    g = TicTacToe(d["n"], d["k"])
    net = _seeded_net_15(d).to("cuda:0")
    games = d["games"]
    gm0 = games[0]
    G = len(games)
    assert [gm["uid"] for gm in games] == list(range(gm0["uid"], gm0["uid"] + G))
    eng = SelfPlayEngine(g, G, net1=net, max_batch=gm0["batch"], inference=inference, seed=gm0["seed"],
                         steps_before_tau_0=gm0["steps_before_tau_0"], uid_base=gm0["uid"], first_player_mode=2,
                         node_cap=4096, evict=True, searches_hint=gm0["searches"])
    total, same, max_dpi, followed, log = _compare_with_recorded_games(d, eng, g, max(gm["plies"] for gm in games))
    assert eng.counters()["overflows"] == 0
    eng.close()
    print("\n".join(log))
    print("%s 15x15: identical root-N plies %d / %d (%.2f %%), max |dpi| elsewhere %.4f, %d / %d games followed to "
          "the end" % (inference, same, total, 100.0 * same / total, max_dpi, followed, G))
    assert total >= 200
    assert same / total >= 0.99 and max_dpi <= 0.15
    assert followed >= G - 2


def test_play_cli_round_robin(capsys):
    """play.py drop-in: two checkpoints, every ordered pair, W/L/D bookkeeping adds up (play.py:40-76)."""
    from caro_ai_amd import play
    a = os.path.join(GOLDEN, "weights", "best_026_12000.dat")
    b = os.path.join(GOLDEN, "weights", "best_025_10600.dat")
    agents, pairs = play.main(["-g", "0", "--cuda", a, b, "-r", "8"])
    out = capsys.readouterr().out
    assert "Leaderboard:" in out and out.count(" vs ") == 2
    assert sum(pairs[(a, b)]) == 8 and sum(pairs[(b, a)]) == 8
    wa, la, da = agents[a]
    wb, lb, db = agents[b]
    assert (wa, la, da) == (lb, wb, db) and wa + la + da == 16
    # the printed lines have the SHAPE of what the reference's play.py printed (tests/golden/play_script_c4.json.gz:
    # recorded by running that script): the same lines once names and numbers are masked
    import re
    ref = load_golden("play_script_c4.json.gz")

    def shape(lines, names):
        out = []
        for ln in lines:
            for k, nm in enumerate(names):
                ln = ln.replace(nm, "<%d>" % k)
            out.append(re.sub(r"\d+", "N", ln))
        return out

    mine = shape(out.strip().splitlines(), [a, b])
    theirs = shape(ref["stdout"], ref["models"])
    assert mine[:3] == theirs[:3] and sorted(mine[3:]) == sorted(theirs[3:]), (mine, theirs)


def test_play_session_bot_moves():
    """lib/play_session.py: the bot answers a human move; state, legality and the value read-out behave."""
    from caro_ai_amd.lib.game.connect_four import ConnectFour
    from caro_ai_amd.lib.play_session import Session
    np.random.seed(0)
    g = ConnectFour()
    s = Session(g, os.path.join(GOLDEN, "weights", "best_026_12000.dat"), player_moves_first=True)
    assert s.is_valid_move(3) and not s.is_draw()
    assert s.move_player(3) is False
    won = s.move_bot()
    assert won is False and len(s.moves) == 2 and s.moves[1] in range(7)
    assert s.value is not None and "<pre>" in s.render()
    assert len(s.mcts_store) > 10


@pytest.mark.parametrize("two_nets", [False, True])
def test_fused_tree_kernel_equals_stepwise_kernels(two_nets, monkeypatch):
    """caro_search_batch runs the fused k_tree (expand+backup, select, row reservation + planes in one launch)
    when a game is one wavefront; CARO_NO_FUSED_TREE=1 keeps the four-launch form.  Net rows land in another
    order, nothing else may change: per-ply root visit counts, roots and the drained tuples are identical."""
    from caro_ai_amd.engine import SelfPlayEngine
    from caro_ai_amd.lib.game.connect_four import ConnectFour
    from caro_ai_amd.lib.model import Net
    g = ConnectFour()
    nets = []
    for w in ("best_026_12000.dat", "best_025_10600.dat"):
        n = Net(g.obs_shape, g.action_space)
        n.load_state_dict(torch.load(os.path.join(GOLDEN, "weights", w), map_location="cpu"))
        nets.append(n.to("cuda:0"))
    runs = []
    for no_fused in ("1", "0"):
        monkeypatch.setenv("CARO_NO_FUSED_TREE", no_fused)
        kw = dict(n_stores=2, first_player_mode=2, steps_before_tau_0=0) if two_nets else dict(steps_before_tau_0=4)
        eng = SelfPlayEngine(g, 96, net1=nets[0], net2=nets[1] if two_nets else None, max_batch=8, seed=5, **kw)
        trace = []
        for ply in range(12):
            eng.search(9, 8)
            pi, counts = eng.policy()
            trace.append((counts.cpu().numpy().copy(), eng.roots()[0].copy()))
            eng.step()
            d = eng.drain(recycle=True)
            trace.append(tuple(d[k].cpu().numpy().copy() for k in ("games", "states", "players", "pi", "z")))
        c = eng.counters()
        eng.close()
        runs.append((trace, c))
    (ta, ca), (tb, cb) = runs
    assert ca == cb and ca["expansions"] > 0
    for x, y in zip(ta, tb):
        for u, w_ in zip(x, y):
            assert np.array_equal(u, w_)


# ------------------------------------------------------------------ SURVEY Q3 through the Level-1 shim: the caller's stores persist
class _TableDraws:
    """What tests/golden/make_golden_r4.py did to the reference, done to the shim: np.random.dirichlet /
    np.random.choice replaced by the counter-based tables keyed (seed, game uid, ply, sim), the opener draw
    `np.random.choice(2)` -> uid & 1, F.softmax -> identity (the table net returns priors, not logits).  The reference
    counts sims per find_leaf call; the shim draws its rows per minibatch (`batch` rows when the root is expanded,
    none otherwise -- lib/mcts.py:123,131), so sim = minibatch index * batch + row.  Also records, after every
    search_batch, what the reference's trace holds: root board / player, root N / W / Q, len(store)."""

    def __init__(self):
        self.seed = self.uid = 0
        self.trace = []

    def new_game(self, seed, uid):
        self.seed, self.uid, self.ply = seed, uid, -1
        self.trace = []

    def __enter__(self):
        import torch.nn.functional as F
        from caro_ai_amd.lib.mcts import MCTS
        from oracle.oracle import move_uniform, noise_row
        h = self
        self._saved = (np.random.dirichlet, np.random.choice, F.softmax, MCTS.search_batch, MCTS.search_minibatch)
        sb, smb = MCTS.search_batch, MCTS.search_minibatch

        def dirichlet(alpha, size=None):
            assert size is None  # the step-wise path draws row by row
            row = noise_row(h.seed, h.uid, h.ply, h.mb * h.batch + h.b, len(alpha), alpha[0])
            h.b += 1
            return row

        def choice(a, p=None):
            if p is None:
                assert a == 2
                return h.uid & 1
            u = move_uniform(h.seed, h.uid, h.ply)
            cdf = np.cumsum(np.asarray(p, dtype=np.float64))
            cdf /= cdf[-1]
            return int(np.searchsorted(cdf, u, side="right"))

        def search_batch(self_, count, batch_size, state_int, player, net, device="cpu"):
            h.ply += 1
            h.mb, h.batch = -1, batch_size
            r = sb(self_, count, batch_size, state_int, player, net, device)
            nd = self_._lookup(state_int)
            h.trace.append({"state": str(state_int), "player": int(player), "N": [int(x) for x in nd["N"][0]],
                            "W": [float(np.float32(x)) for x in nd["W"][0]],
                            "W_f32": [int(x) for x in nd["strong"][0]],
                            "Q": [float(x) for x in self_._q_list(nd)], "nodes": len(self_)})
            return r

        def search_minibatch(self_, batch_size, state_int, player, net, device="cpu"):
            h.mb += 1
            h.b = 0
            return smb(self_, batch_size, state_int, player, net, device)

        np.random.dirichlet, np.random.choice = dirichlet, choice
        F.softmax = lambda x, dim=1: x
        MCTS.search_batch, MCTS.search_minibatch = search_batch, search_minibatch
        return self

    def __exit__(self, *exc):
        import torch.nn.functional as F
        from caro_ai_amd.lib.mcts import MCTS
        np.random.dirichlet, np.random.choice, F.softmax, MCTS.search_batch, MCTS.search_minibatch = self._saved


def _table_module(game, salt):
    """a lib.model.Net whose forward IS the table net (priors straight out: the softmax is the identity here)"""
    from caro_ai_amd.lib.model import Net
    from tests.synth_net import SynthNet

    class M(Net):
        def forward(self, x):
            P, v = SynthNet(int(np.prod(x.shape[1:])), self.actions_n, x.device, salt)(x)
            return P, v.reshape(-1, 1)

    return M(game.obs_shape, game.action_space)


def _check_trace(got, want, tag):
    assert len(got) == len(want), tag
    for ply, (a, b) in enumerate(zip(got, want)):
        for k in ("state", "player", "N", "W_f32", "W", "Q", "nodes"):
            assert a[k] == b[k], (tag, ply, k, a[k], b[k])


def test_play_game_on_a_shared_store_across_games_vs_reference():
    """INTEGRATION.md Level 1 promises that `play_game(game, mcts_store, ...)` runs unchanged with the CALLER's store
    surviving across calls.  ref train.py:184-193 + :41-47: three consecutive self-play games on ONE `MCTS`
    (10 x 8 sims, tau = 1 for 10 plies, opener drawn by play_game), recorded from the reference
    (tests/golden/persist_selfplay_c4.json.gz): root N / W / Q after every search, len(store), the replay rows,
    result and steps -- exact."""
    from caro_ai_amd.lib.game.connect_four import ConnectFour
    from caro_ai_amd.lib.mcts import MCTS
    from caro_ai_amd.lib.utils import play_game
    d = load_golden("persist_selfplay_c4.json.gz")
    g = ConnectFour()
    net = _table_module(g, d["salts"][0])
    store = MCTS(g)
    rb = collections.deque(maxlen=5000)
    with _TableDraws() as h:
        for gm in d["games"]:
            h.new_game(gm["seed"], gm["uid"])
            n0 = len(rb)
            r, steps = play_game(g, store, rb, net, net, d["steps_before_tau_0"], d["searches"], d["batch"],
                                 device="cuda:0")
            assert (r, steps) == (gm["result"], gm["steps"]), gm["uid"]
            _check_trace(h.trace, gm["trace"], gm["uid"])
            assert len(store) == gm["store_len_after"]
            new = list(rb)[n0:]
            assert [str(s) for s, _, _, _ in new] == gm["replay"]["states"]
            assert [p for _, p, _, _ in new] == gm["replay"]["players"]
            assert [list(pi) for _, _, pi, _ in new] == gm["replay"]["pi"]
            assert [z for _, _, _, z in new] == gm["replay"]["z"]
    store.clear()  # ref train.py:217: a new best net empties the store
    assert len(store) == 0


def test_play_game_on_the_evaluate_pair_across_rounds_vs_reference():
    """ref train.py:134-141: one pair [MCTS, MCTS] reused by every round of `evaluate` (20 x 16 sims, tau = 0, no
    replay buffer, challenger != champion): four rounds recorded from the reference, through the shim with the
    caller's two persistent stores -- exact"""
    from caro_ai_amd.lib.game.connect_four import ConnectFour
    from caro_ai_amd.lib.mcts import MCTS
    from caro_ai_amd.lib.utils import play_game
    d = load_golden("persist_evaluate_c4.json.gz")
    g = ConnectFour()
    challenger, champion = _table_module(g, d["salts"][0]), _table_module(g, d["salts"][1])
    stores = [MCTS(g), MCTS(g)]
    with _TableDraws() as h:
        for gm in d["rounds"]:
            h.new_game(gm["seed"], gm["uid"])
            r, steps = play_game(g, stores, None, challenger, champion, 0, d["searches"], d["batch"], device="cuda:0")
            assert (r, steps) == (gm["result"], gm["steps"]), gm["uid"]
            _check_trace(h.trace, gm["trace"], gm["uid"])
            assert [len(stores[0]), len(stores[1])] == gm["store_len_after"]


def test_evaluate_with_reference_stores_reproduces_the_reference_rounds():
    """train.evaluate(..., reference_stores=True) is the reference's evaluate: its rounds are the recorded ones (the
    per-round games are checked through the trace hook), its return value the recorded win ratio; the default
    (independent rounds) is a different, declared, experiment and need not agree"""
    from caro_ai_amd import train
    from caro_ai_amd.lib.game.connect_four import ConnectFour
    d = load_golden("persist_evaluate_c4.json.gz")
    g = ConnectFour()
    challenger, champion = _table_module(g, d["salts"][0]), _table_module(g, d["salts"][1])
    rounds = d["rounds"]
    with _TableDraws() as h:
        import caro_ai_amd.lib.utils as U
        inner, seen = U.play_game, []

        def play_game(*a, **k):  # evaluate knows nothing of uids: hand the harness each round's key
            gm = rounds[len(seen)]
            h.new_game(gm["seed"], gm["uid"])
            r = inner(*a, **k)
            _check_trace(h.trace, gm["trace"], gm["uid"])
            seen.append(r)
            return r

        U.play_game = play_game
        try:
            ratio = train.evaluate(g, challenger, champion, rounds=len(rounds), device="cuda:0", reference_stores=True)
        finally:
            U.play_game = inner
    assert [r for r, _ in seen] == [gm["result"] for gm in rounds]
    assert ratio == d["win_ratio"]


def test_reference_play_script_loop_on_this_packages_play_game():
    """INTEGRATION level 1 at script level.  The reference's arena script play.py (ref play.py:15-76) was RUN as
    `__main__` in the build container (tests/golden/make_golden_r5_play.py: numpy seeded, the shipped checkpoints, two
    rounds per ordered pair, its own 40 x 8 sims per move, fresh stores per game; harness: eval-mode nets) and what it
    printed was recorded.  Here the loop of that script runs on THIS package's modules -- `lib.model.Net`,
    `lib.utils.play_game`, `lib.utils.update_counts`, `config` -- with the same seed (play_game draws the opener, the
    Dirichlet rows and the moves from numpy's global generator exactly where the reference does) and the nets on the
    CPU (so the net arithmetic is the reference's; the trees are on the GPU): every game's (result, steps) and every
    printed line are the reference's."""
    from caro_ai_amd import config as cfg
    from caro_ai_amd.lib import model, utils
    from caro_ai_amd.lib.game.connect_four import ConnectFour
    torch.set_num_threads(1)
    fx = load_golden("play_script_c4.json.gz")
    assert (cfg.PLAY_MCTS_SEARCHES, cfg.PLAY_MCTS_BATCH_SIZE) == (fx["searches"], fx["batch"])
    game = ConnectFour()
    nets = []
    for fname in fx["models"]:
        net = model.Net(game.obs_shape, game.action_space)
        net.load_state_dict(torch.load(os.path.join(GOLDEN, "weights", fname), map_location=lambda storage, loc: storage))
        nets.append((fname, net.eval()))
    lines, games = [], []
    total_agent = {}
    np.random.seed(fx["seed"])
    for idx1, n1 in enumerate(nets):
        for idx2, n2 in enumerate(nets):
            if idx1 == idx2:
                continue
            wins, losses, draws = 0, 0, 0
            for _ in range(fx["rounds"]):
                r, steps = utils.play_game(game=game, mcts_stores=None, replay_buffer=None, net1=n1[1], net2=n2[1],
                                           steps_before_tau_0=0, mcts_searches=cfg.PLAY_MCTS_SEARCHES,
                                           mcts_batch_size=cfg.PLAY_MCTS_BATCH_SIZE, device="cpu")
                games.append([int(r), int(steps)])
                if r > 0.5:
                    wins += 1
                elif r < -0.5:
                    losses += 1
                else:
                    draws += 1
            lines.append("%s vs %s -> w=%d, l=%d, d=%d" % (n1[0], n2[0], wins, losses, draws))
            utils.update_counts(total_agent, n1[0], (wins, losses, draws))
            utils.update_counts(total_agent, n2[0], (losses, wins, draws))
    leaders = sorted(total_agent.items(), reverse=True, key=lambda p: p[1][0])
    lines.append("Leaderboard:")
    for name, (w, l, d) in leaders:
        lines.append("%s: \t w=%d, l=%d, d=%d" % (name, w, l, d))
    assert games == fx["games"], (games, fx["games"])
    assert lines == fx["stdout"], (lines, fx["stdout"])


@pytest.mark.parametrize("kind", ["c4", "ttt3"])
def test_play_session_follows_the_reference_session(kind):
    """SURVEY 8(f) rank 4, pinned: the reference's `lib/play_session.Session` (ref lib/play_session.py:7-49) was driven
    through whole games in the build container (tests/golden/make_golden_r5_session.py: a scripted human, the bot's
    `move_bot` = 40 x 8 sims on the session's persistent store, tau = 0, numpy seeded; harness: eval-mode net).  This
    package's Session with the same seed and the same human -- the net on the CPU, so that its arithmetic is the
    reference's, the tree on the GPU -- makes the same moves, reports the same position values (to 1e-5), renders the
    same boards and ends with a store of the same size."""
    from caro_ai_amd import config as cfg
    from caro_ai_amd.lib.game.connect_four import ConnectFour
    from caro_ai_amd.lib.game.tictactoe import TicTacToe
    from caro_ai_amd.lib.play_session import Session
    torch.set_num_threads(1)
    fx = load_golden("session.json.gz")
    assert (cfg.BOT_MCTS_SEARCHES, cfg.BOT_MCTS_BATCH_SIZE) == (fx["searches"], fx["batch"])
    game = ConnectFour() if kind == "c4" else TicTacToe()
    for gm in fx[kind]["games"]:
        np.random.seed(gm["seed"])
        s = Session(game, os.path.join(GOLDEN, "weights", fx[kind]["weights"]), gm["player_moves_first"], device="cuda:0")
        s.model.cpu()      # the net where the reference's runs ...
        s.device = "cpu"   # ... the tree stays on the GPU (MCTS(tree_device="cuda:0"))
        turns, outcome, turn = [], None, 0
        while outcome is None:
            if gm["player_moves_first"] or turns:
                legal = game.possible_moves(s.state)
                mv = int(legal[(3 * turn + 1) % len(legal)])
                assert s.is_valid_move(mv)
                if s.move_player(mv):
                    outcome = "human"
                    break
                if s.is_draw():
                    outcome = "draw"
                    break
            won = s.move_bot()
            want = gm["turns"][len(turns)]
            got = {"move": int(s.moves[-1]), "value": float(s.value), "state": str(s.state), "render": s.render()}
            # moves, boards and the rendered board exactly; the reported value (a float32 Q = W / N whose W sums the
            # net's values) to 1e-5: torch's CPU convolutions differ in the last bit between the build container's
            # processor and the GPU box's, which no visit count of these games notices but a printed float does
            assert (got["move"], got["state"]) == (want["move"], want["state"]), (kind, gm["seed"], len(turns), got, want)
            assert abs(got["value"] - want["value"]) <= 1e-5, (got["value"], want["value"])
            board = lambda r: r[r.index("<pre>"):]
            assert board(got["render"]) == board(want["render"]) and got["render"].startswith("Position evaluation: ")
            turns.append(got)
            if won:
                outcome = "bot"
            elif s.is_draw():
                outcome = "draw"
            turn += 1
        assert outcome == gm["outcome"] and [int(m) for m in s.moves] == gm["moves"] and len(turns) == len(gm["turns"])
        assert len(s.mcts_store) == gm["store_len"]


@pytest.mark.parametrize("name", ["c4", "ttt3", "mnk5"])
def test_mcts_class_follows_the_reference_call_by_call(name, monkeypatch):
    """SURVEY 8(a) rows a1-a12 at the level of the CLASS: a script of calls on the reference's `MCTS` (find_leaf on an
    empty and on a grown tree, search_minibatch, search_batch, is_leaf, len(), the four dicts, get_policy_value at
    tau = 1 / 0, a walk down the tree on the same store, clear()) was run on the reference with numpy seeded and
    recorded call by call (tests/golden/make_golden_r5_mcts_api.py; table net, softmax = identity: every number exact
    on any machine).  The same calls on this package's MCTS, tree on the GPU, the Dirichlet rows drawn from numpy's
    global generator where the reference draws them: every return value equal -- visit counts, float32 W, priors, pi
    and the leaf tuples exactly; Q exactly where W has absorbed a float32 value, otherwise to float32 (the reference
    holds python floats there)."""
    import torch.nn.functional as F
    from caro_ai_amd.lib.model import Net
    from caro_ai_amd.lib.mcts import MCTS
    from tests.synth_net import SynthNet
    fx = load_golden("mcts_api.json.gz")[name]
    game = _game_for(fx)
    monkeypatch.setattr(F, "softmax", lambda x, dim=1: x)

    class TableNet(Net):
        def forward(self, x):
            P, v = SynthNet(int(np.prod(x.shape[1:])), self.actions_n, x.device)(x)
            return P, v.reshape(-1, 1)

    net = TableNet(game.obs_shape, game.action_space)
    t = MCTS(game)
    np.random.seed(fx["seed"])

    def same_q(got, want, strong):
        return all(float(g) == w if f else np.float32(g) == np.float32(w) for g, w, f in zip(got, want, strong))

    def check_node(s, want):
        assert [int(x) for x in t.visit_count[s]] == want["N"]
        assert [float(x) for x in t.value[s]] == want["W"]
        assert [float(x) for x in t.probs[s]] == want["P"]
        assert same_q(t.value_avg[s], want["Q"], want["W_f32"]), (t.value_avg[s], want["Q"])

    strong_of = {}
    for i, call in enumerate(fx["log"]):
        op = call[0]
        if op == "find_leaf":
            value, leaf, player, states, actions = t.find_leaf(int(call[1]), call[2])
            got = {"value": None if value is None else float(value), "leaf": str(leaf), "player": int(player),
                   "states": [str(x) for x in states], "actions": [int(a) for a in actions]}
            assert got == call[3], (i, got, call[3])
        elif op == "len":
            assert len(t) == call[1]
        elif op == "is_leaf":
            assert bool(t.is_leaf(int(call[1]))) == call[2], i
        elif op == "search_minibatch":
            t.search_minibatch(call[1], int(call[2]), call[3], net, device="cuda:0")
            assert len(t) == call[4]["len"], i
            check_node(int(call[2]), call[4]["node"])
        elif op == "search_batch":
            t.search_batch(call[1], call[2], int(call[3]), call[4], net, device="cuda:0")
            assert len(t) == call[5]["len"], i
            check_node(int(call[3]), call[5]["node"])
            strong_of[call[3]] = call[5]["node"]["W_f32"]
        elif op == "get_policy_value":
            pi, q = t.get_policy_value(int(call[1]), tau=call[2])
            assert [float(x) for x in pi] == call[3], (i, pi, call[3])
            assert same_q(q, call[4], strong_of[call[1]]), (i, q, call[4])
        elif op == "move":
            assert game.move(int(call[1]), call[2], call[3]) == (int(call[4]), call[5])
        elif op == "keys":
            assert sorted(str(k) for k in t.visit_count.keys()) == sorted(call[1])
            assert sum(sum(int(x) for x in v) for v in t.visit_count.values()) == call[2]
        elif op == "clear":
            t.clear()
            assert len(t) == call[1] and bool(t.is_leaf(game.initial_state)) == call[2]
        else:
            raise AssertionError(op)
