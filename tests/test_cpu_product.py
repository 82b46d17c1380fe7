"""CPU-side checks of the product: the C-ABI library loads and exports every
symbol include/caro_hip.h declares, host-side rule helpers and game shims agree
with the reference's vectors, the torch twin of the synthetic net equals the C
one, and the engine refuses to run without a GPU (no fallback)."""
import ctypes as C
import os
import re

import numpy as np
import pytest
import torch

from tests.conftest import ROOT, load_golden


def test_library_exports_every_declared_symbol():
    from caro_ai_amd import _lib
    lib = _lib.load()
    hdr = open(os.path.join(ROOT, "include", "caro_hip.h")).read()
    declared = set(re.findall(r"\b(caro_[a-z0-9_]+)\s*\(", hdr))
    declared -= {"caro_engine", "caro_config"}
    assert declared, "no declarations parsed"
    for name in sorted(declared):
        assert hasattr(lib, name), "libcaro_hip.so does not export %s" % name
    assert declared == set(_lib.EXPORTS), declared ^ set(_lib.EXPORTS)
    assert lib.caro_version() >= 100


def test_config_struct_layout_matches_header():
    from caro_ai_amd import _lib
    assert C.sizeof(_lib.CaroConfig) == 112
    assert _lib.CaroConfig.alpha.offset == 48 and _lib.CaroConfig.seed.offset == 64
    assert _lib.CaroConfig.stagger_recycle.offset == 100 and _lib.CaroConfig.games_limit.offset == 104


def test_default_node_cap_is_the_no_overflow_bound_or_refused():
    """VERDICT r5 task 3: the default cap is searches x batch x cells (+ slack) -- it cannot overflow -- and where that is
    beyond a default tree the constructor refuses (no silent clamp) unless eviction is on"""
    from caro_ai_amd import _lib
    from caro_ai_amd.engine import SelfPlayEngine as E
    assert E.default_node_cap(25, 8, 42) == 25 * 8 * 42 + 64
    assert E.default_node_cap(100, 8, 42) == 100 * 8 * 42 + 64
    assert E.default_node_cap(10, 8, 225) == 10 * 8 * 225 + 64
    with pytest.raises(_lib.CaroError, match="evict=True"):
        E.default_node_cap(50, 8, 225)  # 90 064 nodes of 4 KB: round 5 clamped this to 65 536 without a word
    assert E.default_node_cap(50, 8, 225, evict=True) == E.EVICT_DEFAULT_CAP


@pytest.mark.parametrize("n", [3, 4, 5, 8, 10, 11, 12, 15])
def test_array_conversions_equal_the_per_state_codec(n):
    """f3 (VERDICT r5 task 4): to_keys / from_keys over whole arrays == the per-state to_key / from_key, on random boards
    incl. leading zeros (tokens of player 0 in the first cells), the empty and the full board"""
    import random
    import time
    from caro_ai_amd.lib.game._packed import PackedGame
    from caro_ai_amd.lib.game.tictactoe import TicTacToe
    g = TicTacToe(n, min(n, 5))
    rnd = random.Random(n)
    states = [int("".join(rnd.choice("012") for _ in range(n * n))) for _ in range(400)]
    states += [g.initial_state, int("0" * (n * n)), int("1" * (n * n)), int("0" * (n * n - 1) + "1")]
    t0 = time.perf_counter()
    keys = g.to_keys(states)
    back = g.from_keys(keys)
    dt = (time.perf_counter() - t0) / len(states)
    assert back == states
    assert np.array_equal(keys, PackedGame.to_keys(g, states)) and PackedGame.from_keys(g, keys) == states
    assert keys.dtype == np.uint64 and keys.shape == (len(states), g.key_words)
    assert g.to_keys([]).shape == (0, g.key_words) and g.from_keys(np.zeros((0, g.key_words), np.uint64)) == []
    assert dt < 40e-6  # both ways; measured 2.7 us at 15 x 15 (the per-bit loops: 140 us)
    with pytest.raises(AssertionError):
        g.to_keys([int("1" + "0" * (n * n))])  # more digits than cells


def test_connect_four_array_conversions():
    from caro_ai_amd.lib.game.connect_four import ConnectFour
    g = ConnectFour()
    s, states = g.initial_state, []
    for i in range(20):
        states.append(s)
        s, _ = g.move(s, i % 7, i & 1)
    keys = g.to_keys(states)
    assert keys.shape == (20, 1) and keys.dtype == np.uint64 and g.from_keys(keys) == states
    assert all(isinstance(x, int) for x in g.from_keys(keys))


@pytest.mark.skipif(torch.cuda.is_available(), reason="checks the no-GPU failure mode")
def test_engine_fails_loudly_without_gpu():
    from caro_ai_amd import _lib
    from caro_ai_amd.engine import SelfPlayEngine
    from caro_ai_amd.lib.game.connect_four import ConnectFour
    with pytest.raises(_lib.CaroError):
        SelfPlayEngine(ConnectFour(), 4, evaluators=[lambda x: x])
    cfg = _lib.CaroConfig()
    cfg.game_kind, cfg.n_games, cfg.n_stores, cfg.n_nets, cfg.max_batch = 0, 4, 1, 1, 8
    h = C.c_void_p()
    rc = _lib.load().caro_engine_create(C.byref(cfg), C.byref(h))
    assert rc == -19 and not h.value  # CARO_E_NODEV
    assert b"no CPU fallback" in _lib.load().caro_last_error()


def _game_of(d):
    from caro_ai_amd.lib.game.connect_four import ConnectFour
    from caro_ai_amd.lib.game.tictactoe import TicTacToe
    return ConnectFour() if d["kind"] == "c4" else TicTacToe(d["n"], d["k"])


@pytest.mark.parametrize("name", ["rules_c4.json.gz", "rules_ttt3.json.gz", "rules_mnk5.json.gz",
                                  "rules_mnk15.json.gz"])
def test_game_shims_vs_reference_vectors(name):
    d = load_golden(name)
    game = _game_of(d)
    for r in d["recs"]:
        s = int(r["s"])
        assert game.possible_moves(s) == r["legal"]
        assert sorted(game.invalid_moves(s) + r["legal"]) == list(range(game.action_space))
        s2, won = game.move(s, r["m"], r["p"])
        assert s2 == int(r["s2"]) and won == r["won"]
        assert game.from_key(game.to_key(s2)) == s2
        planes = game.states_to_training_batch([s2], [1 - r["p"]])[0]
        assert planes.dtype == np.float32 and planes.shape == tuple(game.obs_shape)
        assert np.packbits(planes.astype(np.uint8).reshape(-1)).tobytes().hex() == r["planes"]


def test_connect_four_known_answers_through_the_shim():
    """lib/game/connect_four/test_connect_four.py:28-141 restated against the shim."""
    from caro_ai_amd.lib.game.connect_four import ConnectFour
    g = ConnectFour()
    empty = 0b000000000000000000000000000000000000000000110110110110110110110
    full1 = 0b111111111111111111111111111111111111111111000000000000000000000
    assert g.initial_state == empty
    assert g.encode_lists([[]] * 7) == empty and g.encode_lists([[1] * 6] * 7) == full1
    assert g.encode_lists([[0] * 6] * 7) == 0
    assert g.decode_binary(empty) == [[]] * 7 and g.decode_binary(full1) == [[1] * 6] * 7
    assert g.decode_binary(0) == [[0] * 6] * 7
    assert g.possible_moves(0) == [] and g.possible_moves(full1) == []
    assert g.possible_moves(empty) == [0, 1, 2, 3, 4, 5, 6]
    f = g.encode_lists([[0, 1, 1], [1, 0], [0, 1], [0, 0, 1], [0, 0], [1, 1, 1, 0], []])
    s, won = g.move(f, 4, 0)
    assert won is True and s == 3531389463375529686
    f = g.encode_lists([[0, 0, 0, 1], [0, 0, 1], [0], [1], [], [], []])
    assert g.move(f, 2, 1)[1] is True and g.move(f, 2, 0)[1] is False
    with pytest.raises(AssertionError):
        g.move(full1, 0, 1)
    with pytest.raises(AssertionError):
        g.move(empty, 7, 1)
    assert g.obs_shape == (2, 6, 7) and g.action_space == 7


def test_tictactoe_known_answers_through_the_shim():
    """lib/game/tictactoe/test_tictactoe.py:41-144 restated against the shim."""
    from caro_ai_amd.lib.game.tictactoe import TicTacToe
    g = TicTacToe(3, 3)
    assert g.initial_state == 222222222
    assert g.possible_moves(int("010220011")) == [3, 4]
    assert g.invalid_moves(int("010220011")) == [0, 1, 2, 5, 6, 7, 8]
    assert g.possible_moves(int("212220012")) == [0, 2, 3, 4, 8]
    for board, mv, pl, exp in [("002112122", 2, 0, "000112122"), ("021012212", 6, 0, "021012012"),
                               ("021102212", 8, 0, "021102210"), ("120122012", 4, 0, "120102012"),
                               ("120102222", 6, 1, "120102122")]:
        nb, won = g.move(int(board), mv, pl)
        assert won is True and nb == int(exp)
    b = g.states_to_training_batch([int("001010221"), int("101222001")], [1, 0])
    np.testing.assert_equal(b[0], [[[0, 0, 1], [0, 1, 0], [0, 0, 1]], [[1, 1, 0], [1, 0, 1], [0, 0, 0]]])


def test_host_noise_equals_oracle_noise():
    from caro_ai_amd import _lib
    from oracle.oracle import move_uniform, noise_row
    L = _lib.load()
    for A in (7, 9, 225):
        out = np.zeros(A)
        _lib.check(L.caro_host_noise_row(5, 77, 3, 41, A, 0.3, out.ctypes.data))
        assert np.array_equal(out, noise_row(5, 77, 3, 41, A))
    assert L.caro_host_move_uniform(5, 77, 3) == move_uniform(5, 77, 3)


def test_synth_net_twins_agree():
    """torch int64 twin (tests/synth_net.py) == C twin (oracle_synth_net) == the
    numpy twin used when the goldens were recorded."""
    from oracle.oracle import Oracle
    from tests.synth_net import synth_numpy
    rng = np.random.default_rng(0)
    for o in (Oracle(Oracle.C4), Oracle(Oracle.MNK, 15, 5)):
        planes = (rng.random((17, 2, o.rows, o.cols)) < 0.3).astype(np.float32)
        P, v = synth_numpy(planes, o.A)
        Pc = np.zeros((17, o.A), np.float32)
        vc = np.zeros(17, np.float32)
        o.L.oracle_synth_net.argtypes = [C.c_void_p] + [C.c_int] + [C.c_void_p] * 5
        o.L.oracle_synth_net(o.h, 17, planes.ctypes.data, None, None, Pc.ctypes.data, vc.ctypes.data)
        assert np.array_equal(P, Pc) and np.array_equal(v, vc)
        assert P.min() > 0 and abs(v).max() <= 1000 / 1024
        # salted form (the two nets of an arena): torch twin == C twin, and the salt changes the outputs
        o.L.oracle_synth_eval.argtypes = [C.c_void_p, C.c_uint64, C.c_int] + [C.c_void_p] * 3
        for salt in (0, 0x2222, (1 << 64) - 5):
            Ps, vs = synth_numpy(planes, o.A, salt)
            o.L.oracle_synth_eval(o.h, salt, 17, planes.ctypes.data, Pc.ctypes.data, vc.ctypes.data)
            assert np.array_equal(Ps, Pc) and np.array_equal(vs, vc)
            assert (salt == 0) == np.array_equal(Ps, P)


def test_batched_dirichlet_consumes_numpy_stream_like_single_calls():
    """lib/mcts.py `_noise_table` draws all rows of a search_batch with ONE np.random.dirichlet(alpha, size=n);
    the reference draws them one call per descent (lib/mcts.py:56).  Same global stream, same rows, same state
    afterwards -- for every action count in use."""
    import numpy as np
    for A in (7, 9, 225):
        np.random.seed(1234 + A)
        one = np.stack([np.random.dirichlet([0.3] * A) for _ in range(40)])
        after_one = np.random.random()
        np.random.seed(1234 + A)
        many = np.random.dirichlet([0.3] * A, size=40)
        after_many = np.random.random()
        assert np.array_equal(one.view(np.uint64), many.view(np.uint64)) and after_one == after_many


def test_net_kernel_touches_m0_only_in_front_of_its_lds_dma():
    """caro_net.hip sets M0 by hand in front of every `global_load_lds_dwordx4` (inline asm, M0 not on the clobber
    list: ADVICE r2).  That is safe as long as the COMPILER never keeps a value of its own in M0 -- checked on the
    built gfx950 code object: every instruction that names m0 is an `s_mov_b32 m0, ...` whose next memory instruction
    is the LDS DMA it belongs to."""
    import shutil
    import subprocess
    import tempfile
    llvm = "/opt/rocm/lib/llvm/bin"
    obj = os.path.join(ROOT, "caro_ai_amd", "csrc", "caro_net.hip.o")
    if not (os.path.exists(obj) and os.path.exists(os.path.join(llvm, "llvm-objdump"))):
        pytest.skip("needs the built object and the ROCm llvm tools")
    with tempfile.TemporaryDirectory() as tmp:
        fat, co = os.path.join(tmp, "net.fatbin"), os.path.join(tmp, "net.co")
        subprocess.check_call([os.path.join(llvm, "llvm-objcopy"), "--dump-section", ".hip_fatbin=" + fat, obj,
                               os.path.join(tmp, "copy.o")])
        subprocess.check_call([os.path.join(llvm, "clang-offload-bundler"), "--unbundle", "--type=o", "--input=" + fat,
                               "--targets=hipv4-amdgcn-amd-amdhsa--gfx950", "--output=" + co])
        dis = subprocess.check_output([os.path.join(llvm, "llvm-objdump"), "-d", co], text=True)
    ins = [ln.split("//")[0].split() for ln in dis.splitlines() if ln.startswith("\t") or ln.startswith("  ")]
    ins = [i for i in ins if i]
    uses = [k for k, i in enumerate(ins) if any(tok.strip(",") == "m0" for tok in i[1:])]
    assert len(uses) > 20, "no M0 traffic found: did the kernel change its DMA form?"
    for k in uses:
        assert ins[k][0] == "s_mov_b32" and ins[k][1].strip(",") == "m0", " ".join(ins[k])
        nxt = next(i[0] for i in ins[k + 1:k + 6] if i[0] != "s_nop")
        assert nxt == "global_load_lds_dwordx4", (" ".join(ins[k]), nxt)


def test_winograd2d_predicate_matches_what_the_heads_kernel_is_sized_for():
    """k_net_heads stages 228 floats per feature plane and 20 x 225 value-head weights -- boards of more
    than 225 cells (15x16: 8 x 8 tiles, 240 cells) must not be offered the 2-D Winograd form (host-side predicate)"""
    from caro_ai_amd import _lib
    L = _lib.load()
    ok = {(h, w) for h in range(1, 20) for w in range(1, 20) if L.caro_net_winograd2d_supported(h, w)}
    assert (15, 15) in ok and (12, 12) in ok and (14, 14) in ok and (13, 15) in ok
    assert all(128 <= h * w <= 225 and h <= 16 and w <= 16 for h, w in ok)
    assert (15, 16) not in ok and (16, 15) not in ok and (16, 16) not in ok and (11, 11) not in ok and (6, 7) not in ok


def _code_object_metadata(obj_name):
    """kernel name -> metadata dict (.vgpr_count, .vgpr_spill_count, .private_segment_fixed_size ...) of the gfx950 code
    object inside a built .o"""
    import subprocess
    import tempfile
    import yaml
    llvm = "/opt/rocm/lib/llvm/bin"
    obj = os.path.join(ROOT, "caro_ai_amd", "csrc", obj_name)
    if not (os.path.exists(obj) and os.path.exists(os.path.join(llvm, "llvm-readelf"))):
        pytest.skip("needs the built object and the ROCm llvm tools")
    with tempfile.TemporaryDirectory() as tmp:
        fat, co = os.path.join(tmp, "x.fatbin"), os.path.join(tmp, "x.co")
        subprocess.check_call([os.path.join(llvm, "llvm-objcopy"), "--dump-section", ".hip_fatbin=" + fat, obj,
                               os.path.join(tmp, "copy.o")])
        subprocess.check_call([os.path.join(llvm, "clang-offload-bundler"), "--unbundle", "--type=o", "--input=" + fat,
                               "--targets=hipv4-amdgcn-amd-amdhsa--gfx950", "--output=" + co])
        notes = subprocess.check_output([os.path.join(llvm, "llvm-readelf"), "--notes", co], text=True)
    doc = notes[notes.index("---"):notes.rindex("...")]
    return {k[".name"]: k for k in yaml.safe_load(doc)["amdhsa.kernels"]}


def test_hot_kernels_do_not_spill():
    """k_net_forward_w2 ran with 14 spilled registers and 60 bytes of scratch per lane (reloads inside
    its MFMA loop).  The hot kernels of every bench leg keep everything in registers: no spilled VGPRs, no scratch."""
    net = _code_object_metadata("caro_net.hip.o")
    eng = _code_object_metadata("caro_engine.hip.o")
    seen = 0
    # (mangled names: "11k_tree_stagI" is k_tree_stag alone, not k_tree_stag_mw; "9k_tree_mwI": config 4's tree kernel)
    for md, wanted in ((net, ("k_net_forward_w2", "k_net_forward_wE", "k_net_heads", "k_net_forward_x3")),
                       (eng, ("11k_tree_stagI", "6k_treeI", "9k_tree_mwI"))):
        for name, k in md.items():
            if any(w in name for w in wanted):
                assert k[".vgpr_spill_count"] == 0 and k[".vgpr_count"] <= 256, (name, k[".vgpr_spill_count"])
                # (the m,n,k tree kernels index small per-thread arrays: scratch, no spill; so does the multi-wave kernel's
                # general-form ply on connect four, which no bench leg runs)
                if md is net or ("C4Rules" in name and "k_tree_mw" not in name):
                    seen += 1
                    assert k[".private_segment_fixed_size"] == 0, (name, k[".private_segment_fixed_size"])
    assert seen >= 5


def test_render_is_the_references_text():
    """`BaseGame.render` (what the chat front end shows; ref connect_four.py / tictactoe.py:237-259) on positions the
    reference rendered (tests/golden/make_golden_r5_session.py): the boards of two whole sessions per game, and random
    positions on 6x7, 3x3, 5x5 and 15x15 -- character for character (the m,n,k marks are the reference's cross / circle)"""
    from caro_ai_amd.lib.game.connect_four import ConnectFour
    from caro_ai_amd.lib.game.tictactoe import TicTacToe
    from tests.conftest import load_golden
    fx = load_golden("session.json.gz")
    n = 0
    for kind, game in (("c4", ConnectFour()), ("ttt3", TicTacToe())):
        for gm in fx[kind]["games"]:
            for t in gm["turns"]:
                r = t["render"]
                assert game.render(int(t["state"])) == r[r.index("<pre>") + 5:-len("</pre>")]
                n += 1
    for r in fx["renders"]:
        game = ConnectFour() if r["kind"] == "c4" else TicTacToe(r["n"], r["k"])
        assert game.render(int(r["state"])) == r["render"], r
        n += 1
    assert n > 30


def test_tictactoe_helpers_known_answers_and_reference_digests():
    """the reference's line helpers (ref lib/game/tictactoe/tictactoe_helpers.py:7-179) under their own names in
    caro_ai_amd.lib.game.tictactoe.tictactoe_helpers: the reference's known answers (test_tictactoe_helpers.py:14-53),
    and on 1 500 random boards of 3x3 .. 15x15 the same outputs as the reference's functions, digest for digest
    (tests/golden/make_golden_r5_helpers.py ran those)."""
    from caro_ai_amd.lib.game.tictactoe import tictactoe_helpers as th
    from tests.conftest import load_golden
    from tests.rules_digest import helpers_digest as digest
    b = [[1, -1, 1], [-1, -1, 0], [0, -1, 1]]
    assert th.get_col(b, [0, 0]) == [1, -1, 0] == th.get_col(b, [1, 0])
    assert th.get_col(b, [2, 1]) == [-1, -1, -1] and th.get_col(b, [1, 2]) == [1, 0, 1]
    assert th.get_diag(b, [0, 0]) == [1, -1, 1] == th.get_diag(b, [1, 1])
    assert th.get_diag(b, [1, 0]) == [-1, -1] == th.get_diag(b, [2, 1]) and th.get_diag(b, [1, 2]) == [-1, 0]
    assert th.get_antidiag(b, [0, 0]) == [1] and th.get_antidiag(b, [1, 0]) == [-1, -1]
    assert th.get_antidiag(b, [2, 1]) == [-1, 0] == th.get_antidiag(b, [1, 2]) and th.get_antidiag(b, [1, 1]) == [0, -1, 1]
    assert th.get_row(b, [1, 2]) == [-1, -1, 0]
    for arr, k, tok, want in [([1, 1, 1], 3, 1, True), ([-1, -1, -1], 3, -1, True), ([1, 0, 1], 3, 1, False),
                              ([-1, -1, 1], 3, -1, False), ([1, 1, 1, 0], 3, 1, True), ([0, -1, -1, -1], 3, -1, True),
                              ([1, 0, 1, 1], 3, 1, False), ([-1, 1, -1, 1], 3, -1, False), ([1, 1], 3, 1, False)]:
        assert th.k_in_a_row(arr, k, tok) is want, (arr, k, tok)
    with pytest.raises(AssertionError):
        th.k_in_a_row([1, 1], 1, 1)
    assert th.check_win([[1, -1, 1], [0, -1, 0], [0, -1, 1]], (1, 1), 3, -1) and not th.check_win(b, (0, 0), 3, 1)
    assert th.check_win([[0, -1, 1], [0, 1, 0], [1, -1, 1]], (2, 0), 3, 1)
    fx = load_golden("helpers_digest.json.gz")
    for n, want in fx["digests"].items():
        assert digest(th, int(n), fx["seed"], fx["cases"]) == want, n


def test_codec_helpers_of_the_game_classes_equal_the_references():
    """the list / matrix views a caller of the reference can reach beside the BaseGame interface (SURVEY 8(a) rows a14 /
    a18): ConnectFour.bits_to_int / int_to_bits / encode_lists / decode_binary / convert_mcts_state_to_nn_state (ref
    connect_four.py:94-155), TicTacToe.flatten_nested_list / _pad_mcts_state / encode_game_state /
    convert_mcts_state_to_list_state (ref tictactoe.py:44-135) -- known answers, and on random positions the digests the
    REFERENCE's classes produced (tests/golden/make_golden_r5_helpers.py); the modules sit where the reference's do
    (lib.game.connect_four.connect_four, lib.game.tictactoe.tictactoe, lib.game.tictactoe.tictactoe_helpers)"""
    from caro_ai_amd.lib.game.connect_four.connect_four import ConnectFour
    from caro_ai_amd.lib.game.tictactoe.tictactoe import TicTacToe
    from tests.conftest import load_golden
    from tests.rules_digest import codec_digest
    c4, t3 = ConnectFour(), TicTacToe()
    assert c4.int_to_bits(5, 3) == [1, 0, 1] and c4.int_to_bits(9, 3) == [0, 0, 1] and c4.bits_to_int([1, 0, 1, 1]) == 11
    assert c4.convert_mcts_state_to_nn_state(c4.initial_state) == [[]] * 7
    assert t3.convert_mcts_state_to_list_state(int("120020100")) == [[1, 2, 0], [0, 2, 0], [1, 0, 0]]
    assert t3.convert_mcts_state_to_list_state(int("000120100")) == [[0, 0, 0], [1, 2, 0], [1, 0, 0]]
    assert t3.encode_game_state([[0, 0, 0], [1, 2, 0], [1, 0, 0]]) == int("000120100")
    assert t3.flatten_nested_list([[1, 2], [3]]) == [1, 2, 3] and t3._pad_mcts_state("12") == "000000012"
    fx = load_golden("helpers_digest.json.gz")
    assert codec_digest(c4, fx["seed"], fx["cases"]) == fx["codec"]["c4"]
    for n, k in ((3, 3), (5, 4), (15, 5)):
        assert codec_digest(TicTacToe(n, k), fx["seed"], fx["cases"]) == fx["codec"]["mnk%d" % n], n


def test_mcts_host_side_helpers_and_tb_tracker():
    """names a caller of the reference can reach that the kernels made redundant, kept as host-side functions with the
    reference's results (compared with the reference's own methods in the build container, round 5): MCTS._add_noise
    (one numpy Dirichlet draw, all A actions, lib/mcts.py:48-62), _calculate_upper_bound (:64-84; NEP-50 types: float32
    in, float32 out), _mask_invalid_actions (:86-95), and lib.utils.TBMeanTracker (lib/utils.py:111-159)."""
    import math
    from caro_ai_amd import config as cfg
    from caro_ai_amd.lib.game.connect_four import ConnectFour
    from caro_ai_amd.lib.mcts import MCTS
    from caro_ai_amd.lib.utils import TBMeanTracker
    g = ConnectFour()
    t = MCTS(g)  # no engine until the first search: these helpers need no GPU
    P = [np.float32(x) for x in (0.1, 0.2, 0.05, 0.25, 0.15, 0.05, 0.2)]
    np.random.seed(3)
    got = t._add_noise(P)
    np.random.seed(3)
    nz = np.random.dirichlet([cfg.ALPHA] * 7)
    assert got == [(1 - cfg.EXPLORE) * p + cfg.EXPLORE * n for p, n in zip(P, nz)] and all(type(x) is np.float64 for x in got)
    Q, N = [np.float32(0.5), np.float32(-0.25)] + [np.float32(0)] * 5, [3, 1, 0, 0, 0, 0, 0]
    ub = t._calculate_upper_bound(Q, P, N)
    assert all(type(x) is np.float32 for x in ub)
    assert ub[0] == np.float32(0.5) + np.float32(cfg.C_PUCT) * P[0] * np.float32(math.sqrt(4)) / np.float32(4)
    assert t._calculate_upper_bound([0.0] * 7, P, [0] * 7) == [0.0] * 7   # no +1 under the root: all zero at a fresh node (Q4)
    s = g.initial_state
    for _ in range(6):
        s, _ = g.move(s, 2, 0)
    scores = [1.0] * 7
    t._mask_invalid_actions(scores, s)
    assert scores == [1.0, 1.0, -np.inf, 1.0, 1.0, 1.0, 1.0]

    class Writer:
        rows, closed = [], False

        def add_scalar(self, name, value, step):
            self.rows.append((name, float(value), step))

        def close(self):
            self.closed = True

    w = Writer()
    with TBMeanTracker(w, 3) as tb:
        for i, v in enumerate([1.0, 2, np.float32(3.5), np.array([1.0, 3.0]), torch.tensor([2.0, 4.0]), 7]):
            tb.track("x", v, i)
    assert w.rows == [("x", 2.1666666666666665, 2), ("x", 4.0, 5)] and w.closed   # what the reference's tracker writes


def test_argument_checks_are_the_references():
    """differential run against the reference's classes in the build container (round 5): which calls assert, which
    raise something else, which go through -- connect_four.py:250-255 (types are `int`, numpy integers are refused; a
    full column asserts), tictactoe.py:226-227 (the bound is off by one: the cell just behind the board is an IndexError,
    an occupied square is NOT refused)"""
    from caro_ai_amd.lib.game.connect_four import ConnectFour
    from caro_ai_amd.lib.game.tictactoe import TicTacToe
    c4, t3 = ConnectFour(), TicTacToe()
    full = c4.initial_state
    for _ in range(3):
        full, _ = c4.move(full, 0, 0)
        full, _ = c4.move(full, 0, 1)
    for bad in (lambda: c4.move(full, 0, 1), lambda: c4.move(c4.initial_state, 7, 1), lambda: c4.move(c4.initial_state, -1, 1),
                lambda: c4.move(c4.initial_state, 0, 2), lambda: c4.move(float(c4.initial_state), 0, 1),
                lambda: c4.move(c4.initial_state, np.int64(3), 1), lambda: c4.move(c4.initial_state, "3", 1),
                lambda: c4.possible_moves(1.0), lambda: c4.possible_moves(np.int64(c4.initial_state)),
                lambda: c4.decode_binary(1.5), lambda: c4.encode_lists([[]] * 6), lambda: c4.encode_lists(tuple([[]] * 7)),
                lambda: t3.move(t3.initial_state, 10, 0), lambda: t3.move(t3.initial_state, -1, 0),
                lambda: t3.move(t3.initial_state, 0, 2)):
        with pytest.raises(AssertionError):
            bad()
    with pytest.raises(IndexError):
        t3.move(t3.initial_state, 9, 0)
    assert t3.move(int("212222222"), 1, 0) == (202222222, False)   # an occupied square is overwritten, as in the reference
    assert sorted(c4.invalid_moves(full)) == [0] and t3.possible_moves(int("010101010")) == []
    assert c4.states_to_training_batch([full], [0, 1]).shape == (1, 2, 6, 7)   # a longer who_moves list is tolerated
