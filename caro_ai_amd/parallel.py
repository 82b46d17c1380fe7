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
        return {"states": allb[:, :8 * KW].contiguous().view(torch.int64).reshape(tot, KW).to(out_dev),
                "players": allb[:, 8 * KW:8 * KW + 4].contiguous().view(torch.int32).reshape(tot).to(out_dev),
                "z": allb[:, 8 * KW + 4:8 * KW + 8].contiguous().view(torch.int32).reshape(tot).to(out_dev),
                "pi": allb[:, 8 * KW + 8:].contiguous().view(self.pi_dtype).reshape(tot, A).to(out_dev)}


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
