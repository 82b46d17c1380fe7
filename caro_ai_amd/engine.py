"""Host driver of the HIP self-play engine: G concurrent `play_game`s on one GPU.

The loop below is the reference's lib/utils.py:76-99 + lib/mcts.py:162-176,
248-287 turned inside out: instead of one game calling the net with <= 8 rows,
every phase runs for all G games at once,

    for each move:
        for mb in range(mcts_searches):           # search_batch
            select       (HIP)   G x batch descents, unique leaves -> dense planes
            net forward  (torch) one batch of L0 (+ L1) rows      [lib/model.py]
            expand+backup(HIP)
        step             (HIP)   pi, sample, game.move, win/draw, history
        drain            (HIP)   finished games -> (s, player, pi, z) tuples, slots recycled

PyTorch is used for device memory, streams and the conv net only.
"""
import ctypes as C

import numpy as np
import torch

from caro_ai_amd import _lib
from caro_ai_amd import config as cfg

COUNTER_NAMES = ["sims", "levels", "expansions", "terminals", "dropped", "overflows", "plies", "finished"]


# inference= values served by the fused HIP net kernel -> HipNet mode
# hipw: Winograd, the form chosen by the board (row form F(2,3); 2-D form F(2x2,3x3) from 13x13 up); hipw1 / hipw2 force one
# hipx3: the extra bf16x3 split-operand form (net_hip.HipNet mode "bf16x3"): never a default, not bit-identical to fp32
HIP_NET_MODES = {"hip": "f32", "hipw": "f32w", "hipw1": "f32w1", "hipw2": "f32w2", "hipx3": "bf16x3"}

def lanes_per_descent(game):
    """lane geometry of the tree kernels for a game (csrc/caro_variants.h): lanes that share one descent"""
    A = game.action_space
    return 8 if A == 7 else 16 if A <= 16 else 32 if A <= 32 else 64


def staggered_geometry(game, batch, evict=False):
    """can the staggered schedule (every game on its own minibatch clock, caro_search_staggered) run this geometry: whole
    wavefronts per game -- batch x lanes per descent a multiple of 64 --, and with eviction the multi-wavefront kernel
    (above 64)"""
    t = int(batch) * lanes_per_descent(game)
    return t >= 64 and t % 64 == 0 and (t > 64 or not evict)


def _ptr(t):
    return C.c_void_p(t.data_ptr()) if t is not None else None


def torch_evaluator(net, form="gemm"):
    """leaf planes -> (P float32[L,A] softmaxed, v float32[L]); mcts.py:212-218 on the device.
    A `Net` is run in its GEMM inference form (lib/model.py GemmNet: no per-shape kernel search),
    form="net" keeps the module as is."""
    from caro_ai_amd.lib.model import GemmNet, Net
    net.eval()
    if form == "gemm" and isinstance(net, Net):
        dev = next(net.parameters()).device
        net = GemmNet(net).to(dev).eval()

    @torch.no_grad()
    def fn(planes):
        logits, values = net(planes)
        return torch.softmax(logits.float(), dim=1).contiguous(), values.float().reshape(-1).contiguous()

    return fn


class SelfPlayEngine:
    VIEW_LIMIT_BYTES = 16 << 20  # drains whose staging block is larger hand out copies of the finished rows (drain_end)
    DEFAULT_CAP_LIMIT = 1 << 16  # largest node_cap chosen without being asked (15 x 15: 4 KB per node and slot)
    EVICT_DEFAULT_CAP = 1 << 13  # default cap of LIVE nodes with eviction on, where the no-eviction bound is above the limit

    def __init__(self, game, n_games, net1=None, net2=None, evaluators=None, n_stores=1, max_batch=None,
                 node_cap=None, steps_before_tau_0=cfg.STEPS_BEFORE_TAU_0, first_player_mode=2,
                 c_puct=cfg.C_PUCT, alpha=cfg.ALPHA, explore=cfg.EXPLORE, seed=0, uid_base=0, uid_stride=None,
                 device="cuda:0", searches_hint=cfg.MCTS_SEARCHES, inference="hipw", evict=False, stagger=False,
                 stagger_recycle=True, games_limit=0):
        if not torch.cuda.is_available():
            raise _lib.CaroError("SelfPlayEngine needs a GPU (torch.cuda.is_available() is False); "
                                 "there is no CPU fallback")
        self.L = _lib.load()
        self.game = game
        self.device = torch.device(device)
        self.G = int(n_games)
        self.A = game.action_space
        self.KW = game.key_words
        self.obs_shape = tuple(game.obs_shape)
        self.HW = self.obs_shape[1] * self.obs_shape[2]
        self.max_batch = int(max_batch or cfg.MCTS_BATCH_SIZE)
        if evaluators is None:
            nets = [net1] if net2 is None or net2 is net1 else [net1, net2]
            if inference in HIP_NET_MODES:
                from caro_ai_amd.net_hip import HipNet
                mode = HIP_NET_MODES[inference]
                evaluators = [HipNet(n, str(self.device), mode=mode) for n in nets]
            else:
                evaluators = [torch_evaluator(n.to(self.device), form=inference) for n in nets]
        self.evaluators = list(evaluators)
        # evaluators that read the leaf count on the device let a whole move be enqueued without a host sync
        self.async_net = all(getattr(e, "device_counts", False) for e in self.evaluators)
        self.n_nets = len(self.evaluators)
        assert self.n_nets in (1, 2)
        if node_cap is None:
            node_cap = self.default_node_cap(searches_hint, self.max_batch, self.HW, evict)
        c = _lib.CaroConfig()
        c.game_kind, c.n, c.k = game.kind, game.n, game.k
        c.n_games, c.n_stores, c.n_nets = self.G, n_stores, self.n_nets
        c.max_batch, c.node_cap = self.max_batch, int(node_cap)
        c.steps_before_tau_0, c.first_player_mode = steps_before_tau_0, first_player_mode
        c.c_puct, c.alpha, c.explore = c_puct, alpha, explore
        c.seed, c.uid_base = seed, uid_base
        c.uid_stride = uid_stride if uid_stride is not None else self.G
        c.device_id = self.device.index or 0
        c.evict = 1 if evict else 0
        # staggered mode (include/caro_hip.h, caro_search_staggered): every game on its own minibatch clock, the ply
        # inside the tree kernel, finished games parked and restarted in place; search() then runs `searches`
        # launches, step() has nothing left to do and drain() hands out the parked games
        self.stagger = bool(stagger)
        self.stag_S = int(searches_hint)
        if self.stagger and not self.async_net:
            raise _lib.CaroError("stagger=True needs device-side evaluators (the fused HIP net or HashNet)")
        c.stagger = self.stag_S if self.stagger else 0
        # (2 = pool mode, with games_limit: finished slots are handed the next games not started yet at every drain
        # instead of restarting with their own next uid -- caro_config.stagger_recycle)
        c.stagger_recycle = int(stagger_recycle)
        c.games_limit = int(games_limit or 0)
        self.stagger_recycle = bool(stagger_recycle)
        self.cfg = c
        self.n_stores = n_stores
        torch.cuda.set_device(self.device)
        h = C.c_void_p()
        _lib.check(self.L.caro_engine_create(C.byref(c), C.byref(h)))
        self.h = h
        rows = self.G * self.max_batch  # one row per descent at most, whichever net it goes to
        self.planes = torch.zeros((rows,) + self.obs_shape, dtype=torch.float32, device=self.device)
        self.leaf_keys = torch.zeros((rows, self.KW), dtype=torch.int64, device=self.device)
        self._probs = torch.zeros((rows, self.A), dtype=torch.float32, device=self.device)
        self._values = torch.zeros(rows, dtype=torch.float32, device=self.device)
        self._counts = (C.c_int32 * 2)()
        cd = C.c_void_p()
        _lib.check(self.L.caro_leaf_counts_dev(self.h, C.byref(cd)))
        self._counts_dev = cd
        self.maxply = self.HW
        self._row_bytes = 8 * self.KW + 8 * self.A + 8  # one drained tuple: state words, float64 pi, player, z
        self.net_rows = 0
        self.net_calls = 0
        self._prof = False
        self._drain_open = False

    @classmethod
    def default_node_cap(cls, searches, max_batch, cells, evict=False):
        """node_cap when the caller names none.  A tree never holds more than searches x batch new nodes per move over
        at most `cells` moves (lib/mcts.py:248-287: one minibatch adds at most `batch` nodes): that bound is the
        default.  Where it is larger than a default tree may be (15 x 15: 4 KB per node and slot) a smaller cap would
        let a long game overflow and silently leave the reference's games, so -- without eviction -- the constructor
        REFUSES instead of clamping; with eviction (node_cap bounds the LIVE nodes) the default is EVICT_DEFAULT_CAP
        and an overflow, should one happen, is an error in every caller of this package."""
        bound = int(searches) * int(max_batch) * int(cells) + 64
        if bound <= cls.DEFAULT_CAP_LIMIT:
            return bound
        if not evict:
            raise _lib.CaroError(
                "SelfPlayEngine: %d searches x %d descents x %d cells needs up to %d nodes per tree, more than the %d a "
                "default tree holds; pass evict=True (unreachable nodes are dropped after every move, result-neutral: "
                "node_cap then bounds the LIVE nodes) or an explicit node_cap"
                % (searches, max_batch, cells, bound, cls.DEFAULT_CAP_LIMIT))
        return cls.EVICT_DEFAULT_CAP

    def close(self):
        if getattr(self, "h", None):
            self.L.caro_engine_destroy(self.h)
            self.h = None

    RUN_FIELDS = ("seed", "uid_base", "uid_stride", "games_limit", "steps_before_tau_0", "first_player_mode", "c_puct",
                  "alpha", "explore", "stagger_recycle")

    def restart(self, evaluators=None, searches=None, **run):
        """A new run on this engine in place of close() + a new engine (caro_engine_restart): every game back at the
        initial position, trees empty, counters zero, fresh minibatch clocks -- what a fresh engine of the same
        configuration starts from, so it plays the same games bit for bit -- with the tree tables kept.  `run` may
        reset any of RUN_FIELDS; `searches` the staggered mode's minibatches per move; `evaluators` the nets."""
        bad = set(run) - set(self.RUN_FIELDS)
        assert not bad, "restart() cannot change %s: these shape the engine's memory" % sorted(bad)
        if self._drain_open:
            self.flush()
        c = self.cfg
        for k, val in run.items():
            setattr(c, k, int(val) if k == "stagger_recycle" else val)
        if searches is not None and self.stagger:
            self.stag_S = int(searches)
            c.stagger = self.stag_S
        self.stagger_recycle = bool(c.stagger_recycle)
        if evaluators is not None:
            evaluators = list(evaluators)
            assert len(evaluators) == self.n_nets
            assert all(getattr(e, "device_counts", False) for e in evaluators) == self.async_net
            self.evaluators = evaluators
        _lib.check(self.L.caro_engine_restart(self.h, C.byref(c), self._stream()))
        self.net_rows = self.net_calls = 0

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def _stream(self):
        return C.c_void_p(torch.cuda.current_stream(self.device).cuda_stream)

    # ------------------------------------------------------------ phases
    def reset(self, first_players=None):
        fp = None
        if first_players is not None:
            fp = torch.as_tensor(first_players, dtype=torch.int32).to(self.device).contiguous()
        _lib.check(self.L.caro_reset_games(self.h, _ptr(fp), self._stream()))
        if fp is not None:
            torch.cuda.current_stream(self.device).synchronize()

    def set_roots(self, states, players):
        keys = torch.from_numpy(self.game.to_keys(states).view(np.int64)).to(self.device)
        pl = torch.as_tensor(players, dtype=torch.int32).to(self.device)
        _lib.check(self.L.caro_set_roots(self.h, _ptr(keys), _ptr(pl), self._stream()))
        torch.cuda.current_stream(self.device).synchronize()

    def minibatch(self, batch, mb_index, noise=None):
        """one search_minibatch for all games; returns (L0, L1)"""
        st = self._stream()
        nz = None
        if noise is not None:
            nz = noise if torch.is_tensor(noise) else torch.as_tensor(np.asarray(noise, dtype=np.float64))
            nz = nz.to(self.device, dtype=torch.float64).contiguous()
            assert nz.numel() == self.G * batch * self.A
        _lib.check(self.L.caro_select(self.h, batch, mb_index, _ptr(nz), _ptr(self.planes), _ptr(self.leaf_keys), st))
        if self.async_net:
            # fused HIP net: L is read on the device, nothing waits on the host
            slot = self.L.caro_profile_begin(self.h, 4, st) if self._prof else -1
            for which, ev in enumerate(self.evaluators):
                ev.forward_dev(self.planes, self._counts_dev, which, self.G * batch, self._probs, self._values, st)
            if slot >= 0:
                self.L.caro_profile_end(self.h, slot, st)
            self.net_calls += self.n_nets
            _lib.check(self.L.caro_expand_backup(self.h, _ptr(self._probs), _ptr(self._values), st))
            return None
        _lib.check(self.L.caro_leaf_counts(self.h, self._counts, st))
        l0, l1 = self._counts[0], self._counts[1]
        if l0:
            p, v = self.evaluators[0](self.planes[:l0])
            self._probs[:l0].copy_(p)
            self._values[:l0].copy_(v)
        if l1:
            p, v = self.evaluators[1](self.planes[l0:l0 + l1])
            self._probs[l0:l0 + l1].copy_(p)
            self._values[l0:l0 + l1].copy_(v)
        self.net_rows += l0 + l1
        self.net_calls += (l0 > 0) + (l1 > 0)
        _lib.check(self.L.caro_expand_backup(self.h, _ptr(self._probs), _ptr(self._values), st))
        return l0, l1

    def search(self, searches, batch, noise=None):
        """search_batch (mcts.py:162-176) for all games. noise: optional [searches, G, batch, A] rows."""
        if self.stagger:
            assert noise is None and searches == self.stag_S, "staggered mode: generated noise, searches = searches_hint"
            nets = [e.h for e in self.evaluators] + [None]
            _lib.check(self.L.caro_search_staggered(self.h, nets[0], nets[1], searches, batch, _ptr(self.planes),
                                                    _ptr(self.leaf_keys), _ptr(self._probs), _ptr(self._values),
                                                    self._stream()))
            self.net_calls += searches * self.n_nets
            return
        if self.async_net:
            # whole search_batch enqueued by one C call (no Python / ctypes work per launch)
            nz = None
            if noise is not None:
                nz = noise if torch.is_tensor(noise) else torch.as_tensor(np.asarray(noise, dtype=np.float64))
                nz = nz.to(self.device, dtype=torch.float64).contiguous()
                assert nz.numel() == searches * self.G * batch * self.A
            nets = [e.h for e in self.evaluators] + [None]
            _lib.check(self.L.caro_search_batch(self.h, nets[0], nets[1], searches, batch, _ptr(nz), _ptr(self.planes),
                                                _ptr(self.leaf_keys), _ptr(self._probs), _ptr(self._values),
                                                self._stream()))
            self.net_calls += searches * self.n_nets
            return
        for mb in range(searches):
            self.minibatch(batch, mb, None if noise is None else noise[mb])

    def policy(self):
        pi = torch.empty((self.G, self.A), dtype=torch.float64, device=self.device)
        counts = torch.empty((self.G, self.A), dtype=torch.int32, device=self.device)
        _lib.check(self.L.caro_policy(self.h, _ptr(pi), _ptr(counts), self._stream()))
        return pi, counts

    def step(self, uniforms=None):
        if self.stagger:  # the plies are made inside the tree kernel, each game in its own time
            assert uniforms is None
            return None
        u = None
        if uniforms is not None:
            u = torch.as_tensor(np.asarray(uniforms, dtype=np.float64)).to(self.device).contiguous()
        actions = torch.empty(self.G, dtype=torch.int32, device=self.device)
        done = torch.empty(self.G, dtype=torch.int32, device=self.device)
        result = torch.empty(self.G, dtype=torch.int32, device=self.device)
        _lib.check(self.L.caro_step(self.h, _ptr(u), _ptr(actions), _ptr(done), _ptr(result), self._stream()))
        return actions, done, result

    def _staging(self, cap):
        """fresh output buffers for one drain, carved from ONE allocation (the caching allocator hands back a block a
        dropped drain released: no GPU work): the rows a drain returns are views of them, so no copy kernels follow the
        drain kernels on the stream and tuples kept across moves never alias a later drain"""
        cap = int(cap or self.G * self.maxply)
        KW, A, G = self.KW, self.A, self.G
        sizes = (cap * KW * 8, cap * A * 8, G * 4 * 8, cap * 4, cap * 4)  # states, pi, games (8-byte types first), players, z
        buf = torch.empty(sum(sizes), dtype=torch.uint8, device=self.device)
        o = [0]
        for n in sizes:
            o.append(o[-1] + n)
        return cap, (buf[o[0]:o[1]].view(torch.int64).view(cap, KW),
                     buf[o[3]:o[4]].view(torch.int32),
                     buf[o[1]:o[2]].view(torch.float64).view(cap, A),
                     buf[o[4]:o[5]].view(torch.int32),
                     buf[o[2]:o[3]].view(torch.int64).view(G, 4))

    def drain_begin(self, recycle=True, cap=None):
        """first half of drain(): the kernels are enqueued, nothing waits (see caro_drain_tuples_begin)"""
        cap, bufs = self._staging(cap)
        s, p, pi, z, games = bufs
        if self.stagger:  # the parked games; their slots have restarted already (or not: stagger_recycle)
            assert bool(recycle) == self.stagger_recycle, \
                "staggered mode restarts slots in-kernel: recycle is fixed by stagger_recycle at construction"
            _lib.check(self.L.caro_drain_parked_begin(self.h, cap, _ptr(s), _ptr(p), _ptr(pi), _ptr(z), _ptr(games),
                                                      self._stream()))
        else:
            _lib.check(self.L.caro_drain_tuples_begin(self.h, cap, _ptr(s), _ptr(p), _ptr(pi), _ptr(z), _ptr(games),
                                                      1 if recycle else 0, self._stream()))
        self._dr = bufs  # (only once the call was accepted: a refused begin leaves the open drain's buffers in place)

    def drain_end(self):
        """second half: waits for the totals, hands out the rows (views of this drain's own buffers, see _staging)"""
        s, p, pi, z, games = self._dr
        nt, ng = C.c_int64(0), C.c_int64(0)
        _lib.check(self.L.caro_drain_tuples_end(self.h, C.addressof(nt), C.addressof(ng)))
        nt, ng = nt.value, ng.value
        if nt == 0 and ng == 0:  # (fresh zero-size tensors: a zero-row VIEW would keep the staging block alive as well)
            self._dr = None
            return {"states": s.new_empty((0, self.KW)), "players": p.new_empty((0,)), "pi": pi.new_empty((0, self.A)),
                    "z": z.new_empty((0,)), "games": games.new_empty((0, 4))}
        out = {"states": s[:nt], "players": p[:nt], "pi": pi[:nt], "z": z[:nt], "games": games[:ng]}
        # A view keeps the WHOLE staging allocation alive.  Connect four: 3 MB, nothing.  15 x 15: G * 225 rows of
        # 1.8 KB = 106 MB per drain at 256 games, of which a move's finished games fill a few percent -- a consumer
        # that keeps its tuples (TupleGatherer, a replay buffer) would pin gigabytes.  There the rows are copied out
        # (a move of that board takes > 100 ms: five small copies do not show) and the staging block goes back.
        if s.shape[0] * self._row_bytes > self.VIEW_LIMIT_BYTES and 4 * nt < s.shape[0]:
            out = {k: v.clone() for k, v in out.items()}
            self._dr = None
        return out

    def drain(self, recycle=True, cap=None):
        """Finished games -> tuples (device tensors of their own: later drains do not touch them), in the
        reference's append order."""
        self.drain_begin(recycle, cap)
        return self.drain_end()

    def search_step(self, searches, batch):
        """search() + step() (generated noise and move uniforms): one C call, caro_search_move, for a lock-step engine
        with device-side evaluators -- where several wavefronts serve a game the ply and the eviction ride in the
        search's closing launch"""
        if self.async_net and not self.stagger:
            nets = [e.h for e in self.evaluators] + [None]
            _lib.check(self.L.caro_search_move(self.h, nets[0], nets[1], searches, batch, None, None, _ptr(self.planes),
                                               _ptr(self.leaf_keys), _ptr(self._probs), _ptr(self._values), None, None,
                                               None, self._stream()))
            self.net_calls += searches * self.n_nets
        else:
            self.search(searches, batch)
            self.step()

    def move(self, searches, batch, recycle=True):
        """One move of every game with the host loop software-pipelined: search + ply + the drain kernels of THIS
        move are enqueued, and only then the totals of the PREVIOUS move's drain are waited for -- while the GPU is
        busy with this move's search -- so nothing on the host sits between two moves on the GPU.  Returns the
        tuples of the previous move (None on the first call); flush() hands out the last ones."""
        self.search_step(searches, batch)
        out = self.drain_end() if self._drain_open else None
        self.drain_begin(recycle)
        self._drain_open = True
        return out

    def flush(self):
        if not self._drain_open:
            return None
        self._drain_open = False
        return self.drain_end()

    # ------------------------------------------------------------ inspection
    def counters(self):
        out = (C.c_int64 * 8)()
        _lib.check(self.L.caro_counters(self.h, out, self._stream()))
        return dict(zip(COUNTER_NAMES, list(out)))

    def profile(self, on=True):
        _lib.check(self.L.caro_profile_enable(self.h, 1 if on else 0))
        self._prof = bool(on)

    def profile_read(self, reset=True):
        """{kernel: (total ms, launches)} measured with HIP events on the launch stream"""
        ms = (C.c_double * 8)()
        n = (C.c_int64 * 8)()
        _lib.check(self.L.caro_profile_read(self.h, ms, n, 1 if reset else 0))
        return {k: (ms[i], n[i]) for i, k in enumerate(["select", "compact", "expand_backup", "step", "net", "null1", "null2"])}

    def live_games(self):
        out = C.c_int32(0)
        _lib.check(self.L.caro_live_games(self.h, C.addressof(out), self._stream()))
        return out.value

    def pending_leaves(self):
        """unique leaves selected but not booked as expansions yet (staggered mode: every game's pending minibatch):
        sims == expansions + terminals + dropped + pending at any point of a run without overflows"""
        out = C.c_int32(0)
        _lib.check(self.L.caro_pending_leaves(self.h, C.addressof(out), self._stream()))
        return out.value

    def tree_sizes(self):
        out = torch.empty(self.G * self.n_stores, dtype=torch.int32, device=self.device)
        _lib.check(self.L.caro_tree_sizes(self.h, _ptr(out), self._stream()))
        return out.cpu().numpy().reshape(self.G, self.n_stores)

    def tree_live(self):
        """nodes each tree holds now (what node_cap bounds; tree_sizes() = nodes ever created = len(MCTS))"""
        out = torch.empty(self.G * self.n_stores, dtype=torch.int32, device=self.device)
        _lib.check(self.L.caro_tree_live(self.h, _ptr(out), self._stream()))
        return out.cpu().numpy().reshape(self.G, self.n_stores)

    def roots(self):
        keys = torch.empty((self.G, self.KW), dtype=torch.int64, device=self.device)
        pl = torch.empty(self.G, dtype=torch.int32, device=self.device)
        ply = torch.empty(self.G, dtype=torch.int32, device=self.device)
        uid = torch.empty(self.G, dtype=torch.int64, device=self.device)
        _lib.check(self.L.caro_get_roots(self.h, _ptr(keys), _ptr(pl), _ptr(ply), _ptr(uid), self._stream()))
        return (keys.cpu().numpy().view(np.uint64), pl.cpu().numpy(), ply.cpu().numpy(),
                uid.cpu().numpy().view(np.uint64))

    def lookup(self, games, stores, states):
        """node rows of (game, store, state) triples -> dict of numpy arrays"""
        M = len(states)
        dev = self.device
        g = torch.as_tensor(games, dtype=torch.int32).to(dev)
        s = torch.as_tensor(stores, dtype=torch.int32).to(dev)
        keys = torch.from_numpy(self.game.to_keys(states).view(np.int64)).to(dev)
        found = torch.zeros(M, dtype=torch.int32, device=dev)
        N = torch.zeros((M, self.A), dtype=torch.int32, device=dev)
        strong = torch.zeros((M, self.A), dtype=torch.int32, device=dev)
        W = torch.zeros((M, self.A), dtype=torch.float32, device=dev)
        Q = torch.zeros_like(W)
        P = torch.zeros_like(W)
        _lib.check(self.L.caro_lookup_nodes(self.h, M, _ptr(g), _ptr(s), _ptr(keys), _ptr(found), _ptr(N), _ptr(W),
                                            _ptr(Q), _ptr(P), _ptr(strong), self._stream()))
        return {"found": found.cpu().numpy(), "N": N.cpu().numpy(), "W": W.cpu().numpy(), "Q": Q.cpu().numpy(),
                "P": P.cpu().numpy(), "strong": strong.cpu().numpy()}

    # ------------------------------------------------------------ whole games
    def play_until(self, searches, batch, n_finished=None, max_moves=None, recycle=True, on_tuples=None,
                   one_call=False):
        """Run move steps until `n_finished` games have been drained (or `max_moves`).
        one_call: search + ply through search_step() (caro_search_move) instead of search() and step().
        Returns (list of tuple dicts on the host unless on_tuples consumes them, game records int64[n,4])."""
        out, games = [], []
        finished = 0
        moves = 0
        while True:
            if one_call:
                self.search_step(searches, batch)
            else:
                self.search(searches, batch)
                self.step()
            moves += 1
            d = self.drain(recycle=recycle)
            ng = d["games"].shape[0]
            if ng:
                finished += ng
                games.append(d["games"].cpu().numpy().copy())
                if on_tuples is not None:
                    on_tuples(d)
                else:
                    out.append({k: v.cpu().numpy().copy() for k, v in d.items() if k != "games"})
            if n_finished is not None and finished >= n_finished:
                break
            if max_moves is not None and moves >= max_moves:
                break
            if not recycle and self.live_games() == 0:
                break
        games = np.concatenate(games) if games else np.zeros((0, 4), np.int64)
        return out, games


class StreamedSelfPlay:
    """The same G games split over `n_streams` independent engines, each on its own HIP stream.

    The tree kernels of one part (latency bound, a handful of waves per CU) run underneath the net kernel of
    another part (MFMA bound), and no part's leaf batch exceeds one round of net workgroups.  Game uids are
    laid out exactly as in a single engine (slot g of part k is global slot k*G/n + g), so the set of games
    played does not depend on n_streams.  Interface = the subset of SelfPlayEngine that bench.py / play loops use.
    """

    def __init__(self, game, n_games, make_evaluators, n_streams=2, uid_base=0, uid_stride=None, device="cuda:0",
                 partition_cus=True, **kw):
        assert n_games % n_streams == 0
        self.G = n_games
        self.device = torch.device(device)
        per = n_games // n_streams
        stride = uid_stride if uid_stride is not None else n_games
        self._raw_streams = []
        if partition_cus:
            # each part gets its own slice of the CUs: the parts run side by side instead of sharing CUs
            L = _lib.load()
            self.streams = []
            for k in range(n_streams):
                ptr = C.c_void_p()
                _lib.check(L.caro_stream_create_partition(self.device.index or 0, k, n_streams, C.byref(ptr)))
                self._raw_streams.append(ptr)
                self.streams.append(torch.cuda.ExternalStream(ptr.value, device=self.device))
        else:
            self.streams = [torch.cuda.Stream(device=self.device) for _ in range(n_streams)]
        self.parts = []
        for k in range(n_streams):
            with torch.cuda.stream(self.streams[k]):
                self.parts.append(SelfPlayEngine(game, per, evaluators=make_evaluators(), uid_base=uid_base + k * per,
                                                 uid_stride=stride, device=device, **kw))
        torch.cuda.synchronize(self.device)

    def _each(self):
        return zip(self.parts, self.streams)

    # what train.self_play_stream asks of an engine
    @property
    def h(self):
        return all(e.h for e in self.parts)

    @property
    def cfg(self):
        return self.parts[0].cfg

    _drain_open = False  # (this class's move() drains synchronously per part: nothing stays open between calls)

    def restart(self, evaluators=None, searches=None, **run):
        """SelfPlayEngine.restart on every part; `uid_base` is the whole engine's (part k starts k * G / n further)"""
        per = self.G // len(self.parts)
        for k, (e, st) in enumerate(self._each()):
            with torch.cuda.stream(st):
                r = dict(run)
                if "uid_base" in r:
                    r["uid_base"] = r["uid_base"] + k * per
                e.restart(evaluators=evaluators, searches=searches, **r)
                e._primed = False
        torch.cuda.synchronize(self.device)

    def search(self, searches, batch):
        for e, st in self._each():
            with torch.cuda.stream(st):
                e.search(searches, batch)

    def step(self):
        for e, st in self._each():
            with torch.cuda.stream(st):
                e.step()

    def drain(self, recycle=True):
        outs = []
        for e, st in self._each():
            with torch.cuda.stream(st):
                outs.append(e.drain(recycle=recycle))
        for st in self.streams:
            st.synchronize()
        return {k: torch.cat([o[k] for o in outs]) for k in outs[0]}

    def move(self, searches, batch, recycle=True):
        """One move of every part, software-pipelined on the host: for each part in turn, drain its PREVIOUS
        move (the only host sync) and immediately enqueue its next search + step, so the other parts keep the
        GPU busy while this one is being waited for.  Returns the tuples drained this call (those of the
        previous move; the first call returns none)."""
        outs = []
        for e, st in self._each():
            with torch.cuda.stream(st):
                if getattr(e, "_primed", False):
                    outs.append(e.drain(recycle=recycle))
                e.search_step(searches, batch)
                e._primed = True
        if not outs:
            return None
        return {k: torch.cat([o[k] for o in outs]) for k in outs[0]}

    def flush(self, recycle=True):
        """drain the last enqueued move of every part"""
        outs = []
        for e, st in self._each():
            with torch.cuda.stream(st):
                if getattr(e, "_primed", False):
                    outs.append(e.drain(recycle=recycle))
                    e._primed = False
        for st in self.streams:
            st.synchronize()
        return {k: torch.cat([o[k] for o in outs]) for k in outs[0]} if outs else None

    def counters(self):
        tot = {}
        for e, st in self._each():
            with torch.cuda.stream(st):
                for k, v in e.counters().items():
                    tot[k] = tot.get(k, 0) + v
        return tot

    def profile(self, on=True):
        for e in self.parts:
            e.profile(on)

    def profile_read(self, reset=True):
        tot = {}
        for e in self.parts:
            for k, (ms, n) in e.profile_read(reset).items():
                a = tot.get(k, (0.0, 0))
                tot[k] = (a[0] + ms, a[1] + n)
        return tot

    def synchronize(self):
        for st in self.streams:
            st.synchronize()

    def close(self):
        for e in self.parts:
            e.close()
        for ptr in self._raw_streams:
            _lib.load().caro_stream_destroy(ptr)
        self._raw_streams = []
