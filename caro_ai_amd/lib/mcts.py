"""Monte-Carlo tree search behind the reference's `MCTS` interface
(lib/mcts.py:21-313): same constructor, methods, argument meaning and the four
public dict attributes, with the tree living in GPU memory and every search
step executed by the HIP engine (one game slot, one tree).

This is the single-game, latency-shaped entry (play.py, play_session,
train.evaluate call it once per move); throughput comes from running many
games through `caro_ai_amd.engine.SelfPlayEngine` / `lib.utils.play_games`.

Two ways through `search_batch`:
  fused      `net` is this package's `lib.model.Net` in eval mode and `device` is a GPU: the net runs as the
             fused HIP kernel (`HipNet`, cached per net and weight version), the Dirichlet rows of the WHOLE
             search_batch are drawn from numpy up front -- in the reference's order and number -- and ONE
             `caro_search_batch` call enqueues every minibatch (k_tree -> net kernel); the host synchronises once.
  step-wise  any other evaluator (a foreign module, a subclass with its own forward, a net in train mode, a net
             on the CPU): per minibatch `caro_select` -> leaf count -> `net(planes)` -> `caro_expand_backup`,
             which is the reference's own sequence (lib/mcts.py:248-287).

Randomness follows the reference exactly: one `np.random.dirichlet([ALPHA]*A)`
draw per descent that starts at an expanded root (lib/mcts.py:131-132,56-57),
taken from numpy's global state in the same order and handed to the kernel as
an explicit noise table, so a seeded numpy stream is consumed identically.

`device` keeps the reference's meaning: where the NET runs (leaf planes are
moved there and the priors back, lib/mcts.py:214-218).  The tree itself is
always on the GPU; there is no CPU search path.
"""
import ctypes as C
import math as m
from typing import List, Optional, Tuple

import numpy as np
import torch

from caro_ai_amd import _lib
from caro_ai_amd import config as cfg
from caro_ai_amd.engine import SelfPlayEngine

StateInt = int


def _net_evaluator(net, device):
    import torch.nn.functional as F

    def fn(planes):
        with torch.no_grad():
            logits, values = net(planes.to(device))  # batch_tensor.to(device), mcts.py:214
            probs = F.softmax(logits.float(), dim=1)  # mcts.py:216
        return probs.to(planes.device).contiguous(), values.float().reshape(-1).to(planes.device).contiguous()

    return fn


class MCTS:
    """Statistics for every state met during the search, keyed by the state int."""

    NODE_CAP = {"default": 1 << 17, "big": 1 << 15}

    def __init__(self, game, tree_device="cuda:0", node_cap=None):
        self.c_puct = cfg.C_PUCT
        self.game = game
        self._tree_device = tree_device
        self._node_cap = node_cap
        self._eng = None
        self._hip = {}  # id(net) -> (weights version, HipNet)

    # ------------------------------------------------------------ engine plumbing
    def _engine(self) -> SelfPlayEngine:
        if self._eng is None:
            A = self.game.action_space
            lpd = 8 if A == 7 else 16 if A <= 16 else 32 if A <= 32 else 64
            cap = self._node_cap or (self.NODE_CAP["big"] if A > 64 else self.NODE_CAP["default"])
            self._eng = SelfPlayEngine(self.game, 1, evaluators=[None], n_stores=1, max_batch=min(64, 1024 // lpd),
                                       node_cap=cap, c_puct=self.c_puct, alpha=cfg.ALPHA, explore=cfg.EXPLORE,
                                       device=self._tree_device)
            self._seen_overflow = 0
        return self._eng

    def _check_overflow(self):
        if self._eng.counters()["overflows"] > self._seen_overflow:
            raise MemoryError("MCTS node pool exhausted (node_cap=%d); clear() the store or raise node_cap"
                              % self._eng.cfg.node_cap)

    def _lookup(self, state_int):
        r = self._engine().lookup([0], [0], [state_int])
        return r if r["found"][0] else None

    # ------------------------------------------------------------ reference API
    def clear(self):
        if self._eng is not None:
            self._eng.reset([0])

    def __len__(self):
        return 0 if self._eng is None else int(self._eng.tree_sizes()[0][0])

    def is_leaf(self, state_int: StateInt) -> bool:
        return self._lookup(state_int) is None

    def _noise_rows(self, batch_size, state_int):
        """np.random.dirichlet per descent, only when the root is already expanded (mcts.py:123,131)."""
        A = self.game.action_space
        if self.is_leaf(state_int):
            return np.zeros((1, batch_size, A))
        return np.stack([np.random.dirichlet([cfg.ALPHA] * A) for _ in range(batch_size)])[None]

    def _select(self, batch_size, state_int, player):
        eng = self._engine()
        assert batch_size <= eng.max_batch, "mcts_batch_size above %d is not supported for this game" % eng.max_batch
        noise = self._noise_rows(batch_size, state_int)
        eng.set_roots([state_int], [player])
        nz = torch.as_tensor(noise, dtype=torch.float64).to(eng.device).contiguous()
        st = eng._stream()
        _lib.check(eng.L.caro_select(eng.h, batch_size, 0, C.c_void_p(nz.data_ptr()),
                                     C.c_void_p(eng.planes.data_ptr()), C.c_void_p(eng.leaf_keys.data_ptr()), st))
        _lib.check(eng.L.caro_leaf_counts(eng.h, eng._counts, st))
        return eng._counts[0]

    def find_leaf(self, state_int: StateInt, player: int) -> Tuple[Optional[float], StateInt, int, List, List]:
        """One descent (value, leaf_state, player, states, actions); the tree is not modified."""
        eng = self._engine()
        self._select(1, state_int, player)
        dev = eng.device
        info = torch.zeros(4, dtype=torch.int32, device=dev)
        value = torch.zeros(1, dtype=torch.float32, device=dev)
        leaf = torch.zeros(eng.KW, dtype=torch.int64, device=dev)
        pkeys = torch.zeros((eng.HW, eng.KW), dtype=torch.int64, device=dev)
        pact = torch.zeros(eng.HW, dtype=torch.int32, device=dev)
        _lib.check(eng.L.caro_get_descent(eng.h, 0, 0, info.data_ptr(), value.data_ptr(), leaf.data_ptr(),
                                          pkeys.data_ptr(), pact.data_ptr(), eng._stream()))
        _lib.check(eng.L.caro_select_cancel(eng.h))
        status, length, leaf_player, _ = info.cpu().tolist()
        states = self.game.from_keys(pkeys[:length].cpu().numpy().view(np.uint64)) if length else []
        actions = pact[:length].cpu().tolist()
        val = float(value.item()) if status == 1 else None
        return val, self.game.from_key(leaf.cpu().numpy().view(np.uint64)), leaf_player, states, actions

    # ------------------------------------------------------------ fused path (one C call per search_batch)
    @staticmethod
    def _weights_version(net):
        """changes whenever a parameter or batch-norm buffer is written in place (optimizer step, load_state_dict)
        or replaced (`.to()`): tensor version counters + storage addresses"""
        from caro_ai_amd.net_hip import weights_version
        return weights_version(net)

    def _fused_net(self, net, device):
        """the HipNet of `net` if this call may take the fused path, else None"""
        from caro_ai_amd.lib.model import Net
        if not isinstance(net, Net) or type(net).forward is not Net.forward or net.training:
            return None
        if not str(device).startswith("cuda"):
            return None
        eng = self._engine()
        if torch.device(device).index not in (None, eng.device.index):
            return None
        ver = self._weights_version(net)
        hit = self._hip.get(id(net))
        if hit is None or hit[0] != ver:
            from caro_ai_amd.net_hip import HipNet
            if hit is not None:
                hit[1].close()
            hit = (ver, HipNet(net, str(eng.device)))
            self._hip[id(net)] = hit
        return hit[1]

    def _noise_table(self, count, batch_size, state_int):
        """The Dirichlet rows a whole search_batch consumes, drawn from numpy's global stream exactly as the
        reference draws them: one row per descent that starts at an EXPANDED root (lib/mcts.py:123,131-132), i.e.
        none for a first minibatch that finds the root unexpanded (its descents return the root itself, Q6) and
        `batch_size` for every other minibatch.  `dirichlet(alpha, size=n)` fills its rows one after another from
        the same gamma stream, so it equals n single calls (tests/test_cpu_product.py checks that)."""
        A = self.game.action_space
        table = np.zeros((count, 1, batch_size, A))
        skip = 1 if self.is_leaf(state_int) else 0
        n = (count - skip) * batch_size
        if n > 0:
            table[skip:] = np.random.dirichlet([cfg.ALPHA] * A, size=n).reshape(count - skip, 1, batch_size, A)
        return table

    def search_batch(self, count: int, batch_size: int, state_int: StateInt, player: int, net, device: str = "cpu"):
        hip = self._fused_net(net, device) if count > 0 else None
        if hip is None:
            for _ in range(count):
                self.search_minibatch(batch_size, state_int, player, net, device)
            return
        eng = self._engine()
        assert batch_size <= eng.max_batch, "mcts_batch_size above %d is not supported for this game" % eng.max_batch
        noise = self._noise_table(count, batch_size, state_int)
        eng.set_roots([state_int], [player])
        nz = torch.from_numpy(noise).to(eng.device)
        _lib.check(eng.L.caro_search_batch(eng.h, hip.h, None, count, batch_size, C.c_void_p(nz.data_ptr()),
                                           C.c_void_p(eng.planes.data_ptr()), C.c_void_p(eng.leaf_keys.data_ptr()),
                                           C.c_void_p(eng._probs.data_ptr()), C.c_void_p(eng._values.data_ptr()),
                                           eng._stream()))
        self._check_overflow()  # reads the counters: the one host synchronisation of the search

    def search_minibatch(self, batch_size: int, state_int: StateInt, player: int, net, device: str = "cpu") -> None:
        eng = self._engine()
        n_leaf = self._select(batch_size, state_int, player)
        if n_leaf:
            p, v = _net_evaluator(net, device)(eng.planes[:n_leaf])
            eng._probs[:n_leaf].copy_(p)
            eng._values[:n_leaf].copy_(v)
        _lib.check(eng.L.caro_expand_backup(eng.h, C.c_void_p(eng._probs.data_ptr()),
                                            C.c_void_p(eng._values.data_ptr()), eng._stream()))
        self._check_overflow()

    def get_policy_value(self, state_int: StateInt, tau: int = 1) -> Tuple[List[float], List[float]]:
        nd = self._lookup(state_int)
        if nd is None:
            raise KeyError(state_int)
        counts = [int(c) for c in nd["N"][0]]
        if tau == 0:
            probs = [0.0] * self.game.action_space
            probs[int(np.argmax(counts))] = 1.0
        else:
            adj = [c ** (1.0 / tau) for c in counts]
            total = sum(adj)
            probs = [c / total for c in adj]
        return probs, self._q_list(nd)

    # ------------------------------------------------------------ the four public dicts
    @staticmethod
    def _q_list(nd, i=0):
        out = []
        for n, w, q, strong in zip(nd["N"][i], nd["W"][i], nd["Q"][i], nd["strong"][i]):
            out.append(np.float32(q) if strong else (float(w) / int(n) if n else 0.0))
        return out

    def _dump(self):
        eng = self._engine()
        n = len(self) if self._eng is not None else 0
        dev, A, KW = eng.device, eng.A, eng.KW
        cap = max(1, n)
        keys = torch.zeros((cap, KW), dtype=torch.int64, device=dev)
        N = torch.zeros((cap, A), dtype=torch.int32, device=dev)
        strong = torch.zeros((cap, A), dtype=torch.int32, device=dev)
        W = torch.zeros((cap, A), dtype=torch.float32, device=dev)
        Q = torch.zeros_like(W)
        P = torch.zeros_like(W)
        nn = C.c_int64(0)
        _lib.check(eng.L.caro_dump_tree(eng.h, 0, 0, cap, keys.data_ptr(), N.data_ptr(), W.data_ptr(), Q.data_ptr(),
                                        P.data_ptr(), strong.data_ptr(), C.addressof(nn), eng._stream()))
        n = min(nn.value, cap)
        states = self.game.from_keys(keys[:n].cpu().numpy().view(np.uint64)) if n else []
        return states, {"N": N[:n].cpu().numpy(), "W": W[:n].cpu().numpy(), "Q": Q[:n].cpu().numpy(),
                        "P": P[:n].cpu().numpy(), "strong": strong[:n].cpu().numpy()}

    def _view(self, field):
        states, nd = self._dump()
        out = {}
        for i, s in enumerate(states):
            if field == "N":
                out[s] = [int(x) for x in nd["N"][i]]
            elif field == "W":
                out[s] = [np.float32(w) if st else float(w) for w, st in zip(nd["W"][i], nd["strong"][i])]
            elif field == "Q":
                out[s] = self._q_list(nd, i)
            else:
                out[s] = nd["P"][i].copy()
        return out

    def _assign(self, field, mapping):
        """dict assignment as in lib/test_mcts.py:15-21: (re)writes that field of the given states."""
        eng = self._engine()
        A = eng.A
        states = list(mapping.keys())
        if not states:
            return
        cur = eng.lookup([0] * len(states), [0] * len(states), states)
        N, W, Q, P, strong = (cur[k].copy() for k in ("N", "W", "Q", "P", "strong"))
        for i, s in enumerate(states):
            row = list(mapping[s]) + [0] * (A - len(mapping[s]))
            if field == "N":
                N[i] = row
            elif field == "W":
                W[i] = row
                strong[i] = [int(isinstance(x, np.float32)) for x in row]
            elif field == "Q":
                Q[i] = row
            else:
                P[i] = row
        dev = eng.device
        keys = torch.from_numpy(self.game.to_keys(states).view(np.int64)).to(dev)
        z = torch.zeros(len(states), dtype=torch.int32, device=dev)
        t = [torch.from_numpy(np.ascontiguousarray(x)).to(dev) for x in
             (N.astype(np.int32), W.astype(np.float32), Q.astype(np.float32), P.astype(np.float32),
              strong.astype(np.int32))]
        _lib.check(eng.L.caro_poke_nodes(eng.h, len(states), z.data_ptr(), z.data_ptr(), keys.data_ptr(),
                                         t[0].data_ptr(), t[1].data_ptr(), t[2].data_ptr(), t[3].data_ptr(),
                                         t[4].data_ptr(), eng._stream()))
        torch.cuda.current_stream(dev).synchronize()

    visit_count = property(lambda self: self._view("N"), lambda self, m: self._assign("N", m))
    value = property(lambda self: self._view("W"), lambda self, m: self._assign("W", m))
    value_avg = property(lambda self: self._view("Q"), lambda self, m: self._assign("Q", m))
    probs = property(lambda self: self._view("P"), lambda self, m: self._assign("P", m))

    # ------------------------------------------------------------ the reference's per-step helpers, host side
    # The search runs in the kernels (descend_level: noise, PUCT, mask, argmax in one pass); these are the same
    # formulas on Python lists for callers (and tests) that reach into the reference's MCTS -- same names, arguments,
    # results and numpy draws as lib/mcts.py:48-95, 192-223.  The engine never calls them.
    def _add_noise(self, probs):
        """(1 - EXPLORE) * P + EXPLORE * Dirichlet(ALPHA) over ALL actions; one draw from numpy's global stream"""
        noise = np.random.dirichlet([cfg.ALPHA] * self.game.action_space)
        keep = 1 - cfg.EXPLORE
        return [keep * p + cfg.EXPLORE * n for p, n in zip(probs, noise)]

    def _calculate_upper_bound(self, values_avg, probs, counts):
        """Q + c_puct * P * sqrt(sum N) / (1 + N) per action (no +1 under the root: all zero at a fresh node)"""
        root = m.sqrt(sum(counts))
        return [q + self.c_puct * p * root / (1 + n) for q, p, n in zip(values_avg, probs, counts)]

    def _mask_invalid_actions(self, scores, cur_state):
        """-inf on the illegal actions of `cur_state`, in place"""
        for a in self.game.invalid_moves(cur_state):
            scores[a] = -np.inf

    def _expand_tree(self, expand_states, expand_players, expand_queue, backup_queue, net, device="cpu"):
        """evaluate the queued leaves in one batch, create their nodes, queue their backups (lib/mcts.py:192-223)"""
        import torch.nn.functional as F
        planes = torch.tensor(self.game.states_to_training_batch(expand_states, expand_players)).to(device)
        with torch.no_grad():
            logits, values = net(planes)
        priors = F.softmax(logits, dim=1).cpu().numpy()
        for (leaf_state, states, actions), value, prior in zip(expand_queue, values.cpu().numpy()[:, 0], priors):
            self._create_node(leaf_state, prior)
            backup_queue.append((value, states, actions))

    # ------------------------------------------------------------ pieces the reference's tests poke
    def _create_node(self, leaf_state: int, prob):
        A = self.game.action_space
        self._assign("P", {leaf_state: list(prob)})
        self._assign("N", {leaf_state: [0] * A})
        self._assign("W", {leaf_state: [0.0] * A})
        self._assign("Q", {leaf_state: [0.0] * A})

    def _backup(self, value: float, states: List[StateInt], actions: List[int]):
        eng = self._engine()
        if not states:
            return
        dev = eng.device
        keys = torch.from_numpy(self.game.to_keys(states).view(np.int64)).to(dev)
        act = torch.as_tensor(actions, dtype=torch.int32).to(dev)
        _lib.check(eng.L.caro_backup_path(eng.h, 0, 0, float(value), int(isinstance(value, np.float32)), len(states),
                                          keys.data_ptr(), act.data_ptr(), eng._stream()))
        torch.cuda.current_stream(dev).synchronize()
