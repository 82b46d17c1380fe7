"""Multi-GPU self-play: one process per GPU, games sharded by id, no data-path
collective inside the search.  The reference has no distributed code at all
(SURVEY.md section 2); this is the sharding of its independent `play_game` calls
(train.py:41-47 runs them one after another).

Exchanges (torch.distributed; backend "nccl" = RCCL over xGMI on the GPU box,
"gloo" in the CPU tests):
  gather_tuples      variable-length all-gather of (state, player, pi, z) replay
                     tuples: one int64 count all-gather + one padded
                     all_gather_into_tensor per field group.
  TupleGatherer      the same exchange batched over several moves: messages are
                     KB..MB, i.e. latency bound (xGMI ring all-gather of a few
                     hundred KB ~ tens of us plus launch and host overhead), and a
                     move lasts ~6 ms, so tuples wait on the device and every
                     `every`-th move ONE count all-gather and ONE byte-packed payload
                     all-gather move them all.
  broadcast_weights  state_dict broadcast from rank 0 after a training step.
  allreduce_grads    optional data-parallel training step: one flat gradient bucket summed over the ranks.
  allreduce_sum      arena W/L/D counters, expansion counters.
"""
import os

import torch
import torch.distributed as dist


def env_rank():
    return int(os.environ.get("RANK", "0")), int(os.environ.get("LOCAL_RANK", "0")), int(
        os.environ.get("WORLD_SIZE", "1"))


def init(backend=None):
    """Initialise torch.distributed from the torchrun environment (no-op for world size 1)."""
    rank, local_rank, world = env_rank()
    if world > 1 and not dist.is_initialized():
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29500")
        if backend is None:
            # CARO_DIST_BACKEND=gloo lets several ranks share one GPU (rehearsal of the N > 1 path on a 1-GPU box)
            backend = os.environ.get("CARO_DIST_BACKEND") or ("nccl" if torch.cuda.is_available() else "gloo")
        if backend == "nccl":
            torch.cuda.set_device(local_rank)
            # binding the process group to its GPU up front makes barrier() / the first collective use it
            try:
                dist.init_process_group(backend=backend, rank=rank, world_size=world,
                                        device_id=torch.device("cuda", local_rank))
            except TypeError:  # a torch without the device_id argument
                if not dist.is_initialized():
                    dist.init_process_group(backend=backend, rank=rank, world_size=world)
        else:
            dist.init_process_group(backend=backend, rank=rank, world_size=world)
    return rank, local_rank, world


def local_device(local_rank=None):
    """the GPU of this rank: cuda:<local_rank>, wrapped onto the GPUs that exist when several ranks share one
    (gloo rehearsal of the N > 1 path on a 1-GPU box, CARO_DIST_BACKEND=gloo / CARO_SHARE_GPU=1)"""
    if local_rank is None:
        local_rank = env_rank()[1]
    n = torch.cuda.device_count()
    if os.environ.get("CARO_SHARE_GPU"):
        return "cuda:0"
    return "cuda:%d" % (local_rank % n if n > 0 else local_rank)


def shard(n_games_per_rank, rank, world):
    """uid layout: game slot g of rank r starts as uid r*G + g and is recycled with
    stride world*G, so every uid is played exactly once whatever the world size."""
    return {"uid_base": rank * n_games_per_rank, "uid_stride": world * n_games_per_rank}


def shard_rounds(n, rank, world):
    """(first, count) of the contiguous share of n independent rounds that `rank` plays; the shares tile
    [0, n) whatever the world size (the first n % world ranks take one more)"""
    per, rest = divmod(int(n), int(world))
    lo = rank * per + min(rank, rest)
    return lo, per + (1 if rank < rest else 0)


def allreduce_counts(counts, device="cpu"):
    """arena W / L / D (or any small tuple of ints) summed over the ranks -> tuple of ints on every rank
    (train.py:120-149 run sharded: SURVEY 8(e) "all_reduce(SUM) of 3 ints")"""
    if not is_dist():
        return tuple(int(c) for c in counts)
    dev = torch.device("cpu") if dist.get_backend() == "gloo" else torch.device(device)
    t = torch.tensor([int(c) for c in counts], dtype=torch.int64, device=dev)
    dist.all_reduce(t, op=dist.ReduceOp.SUM)
    return tuple(int(x) for x in t.tolist())


def _via_host(t):
    """gloo collectives run on host memory; nccl (RCCL) takes the device tensors as they are"""
    return dist.get_backend() == "gloo" and t.is_cuda


def is_dist():
    return dist.is_available() and dist.is_initialized() and dist.get_world_size() > 1


def _allreduce(t, op):
    if is_dist():
        if _via_host(t):
            h = t.cpu()
            dist.all_reduce(h, op=op)
            t.copy_(h)
        else:
            dist.all_reduce(t, op=op)
    return t


def allreduce_sum(t):
    return _allreduce(t, dist.ReduceOp.SUM)


def allreduce_max(t):
    return _allreduce(t, dist.ReduceOp.MAX)


def gather_tuples(tuples, pi_dtype=torch.float32):
    """All-gather one drain's tuples.  `tuples`: dict with states int64[n,KW],
    players int32[n], pi float[n,A], z int32[n] on this rank's device.  Returns the
    same dict holding every rank's rows, rank-major (deterministic order).
    pi travels as float32 (what the trainer consumes, train.py:92)."""
    if not is_dist():
        return {"states": tuples["states"], "players": tuples["players"], "pi": tuples["pi"].to(pi_dtype),
                "z": tuples["z"]}
    world = dist.get_world_size()
    out_dev = tuples["states"].device
    if _via_host(tuples["states"]):
        tuples = {k: v.cpu() for k, v in tuples.items()}
    dev = tuples["states"].device
    n = tuples["states"].shape[0]
    KW, A = tuples["states"].shape[1], tuples["pi"].shape[1]
    counts = torch.zeros(world, dtype=torch.int64, device=dev)
    mine = torch.tensor([n], dtype=torch.int64, device=dev)
    dist.all_gather_into_tensor(counts, mine)
    counts_h = counts.cpu().tolist()
    m = max(counts_h)
    if m == 0:
        return {"states": tuples["states"][:0], "players": tuples["players"][:0],
                "pi": tuples["pi"][:0].to(pi_dtype), "z": tuples["z"][:0]}
    # one integer record [KW states | player | z] and one float record [pi]
    ints = torch.zeros((m, KW + 2), dtype=torch.int64, device=dev)
    ints[:n, :KW] = tuples["states"]
    ints[:n, KW] = tuples["players"].to(torch.int64)
    ints[:n, KW + 1] = tuples["z"].to(torch.int64)
    flt = torch.zeros((m, A), dtype=pi_dtype, device=dev)
    flt[:n] = tuples["pi"].to(pi_dtype)
    all_i = torch.empty((world * m, KW + 2), dtype=torch.int64, device=dev)
    all_f = torch.empty((world * m, A), dtype=pi_dtype, device=dev)
    dist.all_gather_into_tensor(all_i, ints)
    dist.all_gather_into_tensor(all_f, flt)
    keep = torch.cat([torch.arange(r * m, r * m + c, device=dev) for r, c in enumerate(counts_h)])
    all_i, all_f = all_i[keep], all_f[keep]
    return {"states": all_i[:, :KW].contiguous().to(out_dev), "players": all_i[:, KW].to(torch.int32).to(out_dev),
            "pi": all_f.to(out_dev), "z": all_i[:, KW + 1].to(torch.int32).to(out_dev)}


class TupleGatherer:
    """Collects the drains of several moves on the device and exchanges them in one go.

        tg = TupleGatherer(every=8)
        for each move:  out = tg.push(engine.drain())     # None, or every rank's rows since the last exchange
        out = tg.flush()                                  # at the end (collective: every rank calls it)

    Rows come back rank-major, each rank's rows in the order they were pushed (deterministic).
    One record = [KW int64 states | int32 player | int32 z | A float32 pi] as bytes, so the payload is a
    single all_gather_into_tensor whatever the field types."""

    FIELDS = ("states", "players", "pi", "z")

    def __init__(self, every=8, pi_dtype=torch.float32):
        self.every = max(1, int(every))
        self.pi_dtype = pi_dtype
        self.pending = []
        self.moves = 0

    def push(self, tuples):
        if tuples is not None and int(tuples["z"].shape[0]) > 0:
            self.pending.append({k: tuples[k] for k in self.FIELDS})
        self.moves += 1
        return self.flush() if self.moves % self.every == 0 else None

    def _local(self):
        if not self.pending:
            return None
        out = {"states": torch.cat([d["states"] for d in self.pending]),
               "players": torch.cat([d["players"] for d in self.pending]).to(torch.int32),
               "pi": torch.cat([d["pi"] for d in self.pending]).to(self.pi_dtype),
               "z": torch.cat([d["z"] for d in self.pending]).to(torch.int32)}
        self.pending = []
        return out

    def flush(self):
        mine = self._local()
        if not is_dist():
            return mine
        world = dist.get_world_size()
        # shapes must be known on every rank, with or without local rows: agree on them through the count message
        n = 0 if mine is None else int(mine["z"].shape[0])
        KW = 0 if mine is None else int(mine["states"].shape[1])
        A = 0 if mine is None else int(mine["pi"].shape[1])
        dev = torch.device("cpu") if dist.get_backend() == "gloo" else torch.device("cuda", torch.cuda.current_device())
        out_dev = dev if mine is None else mine["z"].device
        head = torch.tensor([n, KW, A], dtype=torch.int64, device=dev)
        heads = torch.zeros(world * 3, dtype=torch.int64, device=dev)
        dist.all_gather_into_tensor(heads, head)
        heads_h = heads.cpu().reshape(world, 3)
        counts = heads_h[:, 0].tolist()
        m = max(counts)
        if m == 0:
            return None
        KW, A = int(heads_h[:, 1].max()), int(heads_h[:, 2].max())
        isz = torch.empty((), dtype=self.pi_dtype).element_size()
        rec = 8 * KW + 8 + isz * A
        buf = torch.zeros((m, rec), dtype=torch.uint8, device=dev)
        if n:
            buf[:n, :8 * KW] = mine["states"].contiguous().to(dev).view(torch.uint8).reshape(n, 8 * KW)
            buf[:n, 8 * KW:8 * KW + 4] = mine["players"].contiguous().to(dev).view(torch.uint8).reshape(n, 4)
            buf[:n, 8 * KW + 4:8 * KW + 8] = mine["z"].contiguous().to(dev).view(torch.uint8).reshape(n, 4)
            buf[:n, 8 * KW + 8:] = mine["pi"].contiguous().to(dev).view(torch.uint8).reshape(n, isz * A)
        allb = torch.empty((world * m, rec), dtype=torch.uint8, device=dev)
        dist.all_gather_into_tensor(allb, buf)
        keep = torch.cat([torch.arange(r * m, r * m + c, device=dev) for r, c in enumerate(counts)])
        allb = allb[keep]
        tot = allb.shape[0]
        def field(lo, hi, dtype, shape):
            # (reshape(-1) first: a ONE-row slice counts as contiguous as it is and keeps the record stride, which a
            # view as a wider type refuses -- found by selfcheck() on the RCCL group)
            return allb[:, lo:hi].contiguous().reshape(-1).view(dtype).reshape(shape).to(out_dev)
        return {"states": field(0, 8 * KW, torch.int64, (tot, KW)),
                "players": field(8 * KW, 8 * KW + 4, torch.int32, (tot,)),
                "z": field(8 * KW + 4, 8 * KW + 8, torch.int32, (tot,)),
                "pi": field(8 * KW + 8, rec, self.pi_dtype, (tot, A))}


def allreduce_grads(params):
    """data-parallel training step (SURVEY 8(f)1 "optional DDP grad all-reduce"): SUM of every parameter's gradient
    over the ranks, as ONE flat bucket (the net is 0.3-1.2 MB: a single collective, latency bound over xGMI).  The
    caller has already scaled its loss by its share of the global batch, so the sum IS the full-batch gradient."""
    if not is_dist():
        return
    grads = [p.grad for p in params if p.grad is not None]
    if not grads:
        return
    flat = torch.cat([g.reshape(-1) for g in grads])
    allreduce_sum(flat)
    off = 0
    for g in grads:
        n = g.numel()
        g.copy_(flat[off:off + n].view_as(g))
        off += n


def broadcast_weights(net, src=0):
    """state_dict of rank `src` to every rank as ONE flat byte buffer (62 tensors, 0.76-1.2 MB: one latency-bound
    collective instead of 62; float and integer buffers -- num_batches_tracked -- travel side by side as bytes)"""
    if not is_dist():
        return net
    tensors = [t for t in net.state_dict().values() if t.numel()]
    if not tensors:
        return net
    via_host = _via_host(tensors[0])
    dev = torch.device("cpu") if via_host else tensors[0].device
    flat = torch.cat([t.detach().contiguous().reshape(-1).view(torch.uint8).to(dev) for t in tensors])
    dist.broadcast(flat, src=src)
    off = 0
    with torch.no_grad():
        for t in tensors:
            n = t.numel() * t.element_size()
            # (clone: a view of another element size needs an aligned storage offset)
            t.copy_(flat[off:off + n].clone().view(t.dtype).reshape(t.shape).to(t.device))
            off += n
    return net


# ------------------------------------------------------------------ GPUs visible to a launcher, without the runtime
def _kfd_gpu_nodes(root="/sys/class/kfd/kfd/topology/nodes"):
    """KFD topology nodes that are GPUs (simd_count > 0; CPU nodes have none), in node order"""
    out = []
    try:
        names = sorted(os.listdir(root), key=lambda x: int(x) if x.isdigit() else 1 << 30)
    except OSError:
        return out
    for nm in names:
        try:
            props = dict(line.split(None, 1) for line in open(os.path.join(root, nm, "properties")) if " " in line)
        except OSError:
            continue  # a node this process may not read (cgroup device filter): not a GPU it can use
        if int(props.get("simd_count", "0").strip() or 0) > 0:
            out.append(nm)
    return out


def visible_gpu_count(root=None, env=None):
    """Number of GPUs a child process would see, WITHOUT touching the HIP runtime (a launcher must not initialise the
    GPU before it starts its ranks: no `torch.cuda.*` call here): the KFD topology's GPU nodes, narrowed by
    ROCR_VISIBLE_DEVICES and then HIP_VISIBLE_DEVICES / CUDA_VISIBLE_DEVICES the way the runtime applies them (a list
    of indices -- or UUIDs, counted as one device each -- cut at the first invalid entry)."""
    env = os.environ if env is None else env
    if root is None:  # CARO_KFD_TOPOLOGY: another topology directory (tests)
        root = env.get("CARO_KFD_TOPOLOGY", "/sys/class/kfd/kfd/topology/nodes")
    n = len(_kfd_gpu_nodes(root))

    def narrow(n, var):
        v = env.get(var)
        if v is None:
            return n
        cnt = 0
        for tok in v.split(","):
            tok = tok.strip()
            if not tok:
                break
            if tok.lstrip("-").isdigit():
                if not 0 <= int(tok) < n:
                    break
            cnt += 1
        return min(cnt, n)

    n = narrow(n, "ROCR_VISIBLE_DEVICES")
    hip = "HIP_VISIBLE_DEVICES" if "HIP_VISIBLE_DEVICES" in env else "CUDA_VISIBLE_DEVICES"
    return narrow(n, hip)


# ------------------------------------------------------------------ first-contact checklist of a multi-GPU node, as code
class SelfcheckError(RuntimeError):
    """a check of parallel.selfcheck failed; `.check` names it"""

    def __init__(self, check, detail):
        super().__init__("%s: %s" % (check, detail))
        self.check = check


def _synthetic_rows(rank, n, KW, A):
    """n replay rows whose content is a function of (rank, row) alone: every rank can compute every other rank's"""
    i = torch.arange(n, dtype=torch.int64)
    states = (i[:, None] * 1000003 + rank * 7919 + torch.arange(KW, dtype=torch.int64)[None, :] * 104729) * 2654435761
    pi = ((i[:, None] * 31 + torch.arange(A)[None, :] * 17 + rank * 13) % 97).to(torch.float32) / 97.0
    return {"states": states, "players": ((i + rank) % 2).to(torch.int32), "pi": pi,
            "z": ((i + rank) % 3 - 1).to(torch.int32)}


def selfcheck(device="cpu", engine_check=None, fault=None, KW=1, A=7, log=None, force=False):
    """What a first run on a real multi-GPU node can still trip over, exercised ONCE before any timed work (the
    reference has no distributed code; everything below is this package's own N > 1 layer, which so far has only run
    over gloo and on a one-rank RCCL group): every collective the run will issue, at the dtypes and shapes it will use,
    each with contents every rank can predict, so a wrong answer is caught and NAMED instead of showing up as a hang or
    as wrong tuples an hour later.

      identity               all_gather_object of (host, device, PCI bus id, uuid): the ranks sit on distinct GPUs
                             (unless CARO_SHARE_GPU says the sharing is deliberate)
      header_all_gather      the int64[3] count / shape message of TupleGatherer.flush
      payload_all_gather     TupleGatherer's byte-packed uint8 payload, with ZERO rows on some ranks, through the class itself
      empty_flush            a flush with no rows anywhere (returns None on every rank)
      gather_tuples          the per-field form (one int64 + one float32 all_gather_into_tensor)
      allreduce_float64      SUM and MAX of float64 (bench.py's totals and max-over-ranks time)
      allreduce_counts       int64 SUM (arena W / L / D)
      broadcast_weights      the flat uint8 weight broadcast: every rank ends with rank 0's state_dict bit for bit
      allreduce_grads        the one flat float32 gradient bucket of the data-parallel step
      engine_tuples          (GPU) `engine_check(rank, world)` plays this rank's shard of world x 16 table-net games and
                             returns its tuples; they are exchanged with TupleGatherer and compared, as a multiset of rows,
                             with the same uids played by rank 0 ALONE (results must not depend on the sharding)

    Every check ends with an all-reduce of its pass / fail flag, so all ranks leave together: SelfcheckError(check,
    detail) on every rank if any rank saw a mismatch or an exception.  `fault` = name of a check to sabotage on the last
    rank (its own contribution is corrupted): the failure path of the tests.  Returns {"checks": [...], "seconds": s}.
    World size 1: the collectives degenerate to local copies; force=True issues them all the same on an initialised
    one-rank group (what a 1-GPU box can show of RCCL: tests/test_parallel_gloo.py's nccl worker)."""
    import socket
    import time as _time
    t_start = _time.time()
    rank, world = (dist.get_rank(), dist.get_world_size()) if (dist.is_available() and dist.is_initialized()) else (0, 1)
    on = world > 1 or (force and dist.is_available() and dist.is_initialized())
    dev = torch.device(device)
    coll_dev = torch.device("cpu") if (on and dist.get_backend() == "gloo") else dev
    last = world - 1
    done = []

    def verdict(name, ok, detail=""):
        """collective: 1.0 if any rank failed"""
        flag = torch.tensor([0.0 if ok else 1.0], dtype=torch.float64, device=coll_dev)
        if on:
            dist.all_reduce(flag, op=dist.ReduceOp.MAX)
        if flag.item() > 0:
            raise SelfcheckError(name, detail or "failed on another rank")
        done.append(name)
        if log:
            log("[selfcheck] %s ok" % name)

    def run(name, fn):
        ok, detail = True, ""
        try:
            r = fn(fault == name and rank == last)
            if r is not None and r is not True:
                ok, detail = False, str(r)
        except SelfcheckError:
            raise
        except Exception as e:  # a collective that raises on this rank: report it under the check's name
            ok, detail = False, "%s: %s" % (type(e).__name__, e)
        verdict(name, ok, detail)

    # -- identity
    def identity(bad):
        mine = {"rank": rank, "host": socket.gethostname(), "device": str(dev), "pci_bus_id": None, "uuid": None,
                "payload": list(range(rank + 1))}
        if dev.type == "cuda":
            p = torch.cuda.get_device_properties(dev)
            mine["pci_bus_id"], mine["uuid"] = getattr(p, "pci_bus_id", None), str(getattr(p, "uuid", "")) or None
        if bad:
            mine["payload"] = [-1]
        allr = [None] * world
        if on:
            dist.all_gather_object(allr, mine)
        else:
            allr = [mine]
        for r, m in enumerate(allr):
            if m["rank"] != r or m["payload"] != list(range(r + 1)):
                return "all_gather_object returned %r in slot %d" % (m, r)
        ident = [(m["host"], m["device"], m["pci_bus_id"], m["uuid"]) for m in allr]
        if dev.type == "cuda" and len(set(ident)) != world and not os.environ.get("CARO_SHARE_GPU"):
            return "two ranks on one GPU: %s" % ident
    run("identity", identity)

    counts = [(r * 3 + 1) % 5 for r in range(world)]  # rows per rank: some ranks have none
    if world > 1:
        counts[0] = 0

    # -- the count / shape header
    def header(bad):
        head = torch.tensor([counts[rank] + (1 if bad else 0), KW, A], dtype=torch.int64, device=coll_dev)
        heads = torch.zeros(world * 3, dtype=torch.int64, device=coll_dev)
        if on:
            dist.all_gather_into_tensor(heads, head)
        else:
            heads.copy_(head)
        want = torch.tensor([[c, KW, A] for c in counts], dtype=torch.int64).reshape(-1)
        if not torch.equal(heads.cpu(), want):
            return "int64 all_gather_into_tensor gave %s, expected %s" % (heads.cpu().tolist(), want.tolist())
    run("header_all_gather", header)

    def expect_rows():
        parts = [_synthetic_rows(r, counts[r], KW, A) for r in range(world)]
        return {k: torch.cat([p[k] for p in parts]) for k in parts[0]}

    def same_rows(got, want, what):
        if got is None:
            return "%s returned no rows, expected %d" % (what, int(want["z"].shape[0]))
        for k in ("states", "players", "pi", "z"):
            g = got[k].cpu()
            if g.shape != want[k].shape or not torch.equal(g, want[k].to(g.dtype)):
                return "%s: field %r differs (shape %s vs %s)" % (what, k, tuple(g.shape), tuple(want[k].shape))

    # -- the byte-packed payload, through the product class
    def payload(bad):
        mine = {k: v.to(dev) for k, v in _synthetic_rows(rank, counts[rank], KW, A).items()}
        if bad and counts[rank]:
            mine["pi"] = mine["pi"] + 1.0
        elif bad:
            mine = {k: v.to(dev) for k, v in _synthetic_rows(rank, 1, KW, A).items()}
        tg = TupleGatherer(every=1)
        return same_rows(tg.push(mine), expect_rows(), "TupleGatherer")
    run("payload_all_gather", payload)

    def empty(bad):
        tg = TupleGatherer(every=1)
        d = {k: v.to(dev) for k, v in _synthetic_rows(rank, 1 if bad else 0, KW, A).items()}
        out = tg.push(d)
        if out is not None:
            return "a flush without rows returned %d rows" % int(out["z"].shape[0])
    run("empty_flush", empty)

    def per_field(bad):
        mine = {k: v.to(dev) for k, v in _synthetic_rows(rank, counts[rank], KW, A).items()}
        if bad:
            mine["states"] = mine["states"] + 1
            if not counts[rank]:
                mine = {k: v.to(dev) for k, v in _synthetic_rows(rank, 2, KW, A).items()}
        return same_rows(gather_tuples(mine), expect_rows(), "gather_tuples")
    run("gather_tuples", per_field)

    # -- all-reduces
    def reduce_f64(bad):
        t = torch.tensor([rank + 1.0, 2.0 ** -rank, 1e15 + rank], dtype=torch.float64, device=dev)
        if bad:
            t = t + 1.0
        s = allreduce_sum(t.clone()).cpu()
        m = allreduce_max(t.clone()).cpu()
        ws = torch.tensor([sum(r + 1.0 for r in range(world)), sum(2.0 ** -r for r in range(world)),
                           sum(1e15 + r for r in range(world))], dtype=torch.float64)
        wm = torch.tensor([float(world), 1.0, 1e15 + world - 1], dtype=torch.float64)
        if not torch.equal(s, ws) or not torch.equal(m, wm):
            return "float64 all_reduce: SUM %s (expected %s), MAX %s (expected %s)" % (s.tolist(), ws.tolist(), m.tolist(), wm.tolist())
    run("allreduce_float64", reduce_f64)

    def reduce_counts(bad):
        got = allreduce_counts((rank + (5 if bad else 0), 2 * rank + 1, 7), dev)
        want = (sum(range(world)), sum(2 * r + 1 for r in range(world)), 7 * world)
        if tuple(got) != want:
            return "int64 all_reduce gave %s, expected %s" % (got, want)
    run("allreduce_counts", reduce_counts)

    # -- weights and gradients
    def small_net(seed):
        torch.manual_seed(seed)
        net = torch.nn.Sequential(torch.nn.Conv2d(2, 8, 3, padding=1), torch.nn.BatchNorm2d(8), torch.nn.Flatten(),
                                  torch.nn.Linear(8 * 9, 5))
        with torch.no_grad():
            net[1].running_mean.add_(seed + 0.5)
            net[1].num_batches_tracked.add_(seed + 3)
        return net

    def bcast(bad):
        rng = torch.get_rng_state()
        net = small_net(100 + rank).to(dev)
        ref = small_net(100)
        torch.set_rng_state(rng)
        broadcast_weights(net)
        if bad:
            with torch.no_grad():
                net[3].bias.add_(1.0)
        for (k, a), b in zip(net.state_dict().items(), ref.state_dict().values()):
            if not torch.equal(a.cpu(), b):
                return "broadcast_weights: tensor %r is not rank 0's" % k
    run("broadcast_weights", bcast)

    def grads(bad):
        ps = [torch.nn.Parameter(torch.zeros(n, device=dev)) for n in (5, 64, 3)]
        for j, p in enumerate(ps):
            p.grad = torch.full_like(p, float(rank + 1) * (j + 1) + (1.0 if bad else 0.0))
        allreduce_grads(ps)
        tot = sum(r + 1.0 for r in range(world))
        for j, p in enumerate(ps):
            if not torch.equal(p.grad.cpu(), torch.full((p.numel(),), tot * (j + 1))):
                return "allreduce_grads: bucket %d holds %s, expected %s" % (j, p.grad.flatten()[:3].tolist(), tot * (j + 1))
    run("allreduce_grads", grads)

    # -- the engine's tuples: sharded == played alone
    if engine_check is not None:
        def engine(bad):
            mine, alone = engine_check(rank, world)  # this rank's tuples; (rank 0 only) all uids played alone
            if bad:
                mine = dict(mine, z=mine["z"] + 1)
            tg = TupleGatherer(every=1, pi_dtype=torch.float64)
            allrows = tg.push(mine)
            if rank != 0:
                return None

            def multiset(d):
                n = int(d["z"].shape[0])
                rec = torch.cat([d["states"].cpu().contiguous().view(torch.uint8).reshape(n, -1),
                                 d["players"].cpu().to(torch.int32).contiguous().view(torch.uint8).reshape(n, -1),
                                 d["pi"].cpu().to(torch.float64).contiguous().view(torch.uint8).reshape(n, -1),
                                 d["z"].cpu().to(torch.int32).contiguous().view(torch.uint8).reshape(n, -1)], dim=1).numpy()
                return sorted(map(bytes, rec))
            if allrows is None or multiset(allrows) != multiset(alone):
                return ("the tuples gathered from %d ranks differ from the same uids played on rank 0 alone (%s vs %d rows)"
                        % (world, "none" if allrows is None else int(allrows["z"].shape[0]), int(alone["z"].shape[0])))
        run("engine_tuples", engine)
    return {"checks": done, "seconds": _time.time() - t_start, "world_size": world,
            "backend": dist.get_backend() if on else None}


def engine_selfcheck(device, games_per_rank=16, searches=4, batch=8, seed=77):
    """the `engine_check` of selfcheck(): connect four, table net (exact integers: any difference is a real one),
    `games_per_rank` slots per rank, every slot plays ONE game"""
    def play(rank, world):
        from caro_ai_amd.engine import SelfPlayEngine
        from caro_ai_amd.lib.game.connect_four import ConnectFour
        from caro_ai_amd.net_hip import HashNet
        game = ConnectFour()

        def games(G, **uids):
            eng = SelfPlayEngine(game, G, evaluators=[HashNet(game, device=str(device))], max_batch=batch, seed=seed,
                                 device=str(device), searches_hint=searches, games_limit=G, steps_before_tau_0=4, **uids)
            rows = []
            for _ in range(64):  # (a connect-four game has at most 42 plies)
                if not eng.live_games():
                    break
                eng.search(searches, batch)
                eng.step()
                d = eng.drain(recycle=True)
                if int(d["z"].shape[0]):
                    rows.append({k: d[k] for k in ("states", "players", "pi", "z")})
            assert eng.counters()["overflows"] == 0 and eng.live_games() == 0
            eng.close()
            return {k: torch.cat([r[k] for r in rows]) for k in rows[0]}
        mine = games(games_per_rank, **shard(games_per_rank, rank, world))
        alone = games(games_per_rank * world, uid_base=0, uid_stride=games_per_rank * world) if rank == 0 else None
        return mine, alone
    return play
