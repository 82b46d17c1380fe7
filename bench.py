#!/usr/bin/env python3
"""Headline benchmark: self-play MCTS node-expansions/sec/GPU, Connect4,
200 sims/move (25 x 8), 1024 concurrent games per GPU (BASELINE.json configs[1]).

    python bench.py [--gpus N] [--steps K] [--warmup W]
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...
    python bench.py --gpus N          # WORLD_SIZE unset: starts its N ranks itself (one child process per GPU)

A "step" is one move of every one of the G concurrent games on a rank: 25 launch pairs (fused tree kernel -> net
forward), the plies, and the drain of finished games (plus, for N > 1, the all-gather of the drained (s, pi, z)
tuples every few moves).  Default schedule = STAGGERED (include/caro_hip.h, caro_search_staggered): every game runs
its own minibatch clock, so a step is 25 launches in which every game makes one move on average, each at its own
launch, and finished games restart in place; `--stagger 0` = lock-step (all games move together).  Game by game the
two schedules play the same games.  value = node-expansions (reference: `_create_node` calls, lib/mcts.py:178-190 =
train.py's "leaves") summed over ranks / max-over-ranks wall time of the K timed steps.

One JSON line on stdout (rank 0).  `roofline` is the dominant kernel (the fused net forward; HIP events on its
launch stream, inside the timed region; `mfma_busy_pmc` / `traffic` from the committed PMC passes of the same
configuration), `roofline_tree` the tree kernel; `sustained` = 200 further moves of the same engine on their own
clock; `dist` = what the collective layer is (backend, world size, each rank's device and own value);
`cpu_baseline` is the oracle (CPU port of the reference algorithm) driving the same net on the host cores for a
bounded sample, rank 0, N = 1 only.  At N = 1 the line also carries `config5` (512-match arena) and `config4`
(15 x 15, measured in mid-game after --config4-warmup moves at full size), run in the same process after the
headline loop; `extras_rc` != 0 says one of them failed.
"""
import argparse
import json
import math
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0       # MI355X_MICROARCH.md: 8.0 TB/s spec
MFMA_F32_PEAK_TFS = 157.3   # MI355X_MICROARCH.md: f32-input MFMA = 157.3 TFLOP/s dense
MFMA_BF16_PEAK_TFS = 2500.0  # MI355X_MICROARCH.md: bf16 MFMA ~2.5 PFLOP/s dense (the extra bf16x3 leg only)
PMC_FILE = os.path.join("profiles", "pmc_r06.json")
CPU_RATIO_FILE = os.path.join("profiles", "cpu_ratio_r03.json")
NET_KERNEL = {"hip": "k_net_forward", "hipw": "k_net_forward_w", "hipx3": "k_net_forward_x3"}


def host_cores():
    """cores this process may really use: the affinity mask, capped by the cgroup CPU quota (cpu.max / cfs quota)"""
    n = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    try:
        quota, period = open("/sys/fs/cgroup/cpu.max").read().split()[:2]
        if quota != "max":
            n = min(n, max(1, int(int(quota) / int(period))))
    except Exception:
        try:
            q = int(open("/sys/fs/cgroup/cpu/cpu.cfs_quota_us").read())
            per = int(open("/sys/fs/cgroup/cpu/cpu.cfs_period_us").read())
            if q > 0:
                n = min(n, max(1, q // per))
        except Exception:
            pass
    return n


def pmc_status():
    """(commit the PMC passes were taken on, None | why the counters may not be quoted).  The committed PMC file
    carries the SHA-256 of the kernel sources it was measured on (tools/profile_r06.sh); counters of other sources
    are not quoted: a kernel change without a new PMC pass must not carry stale numbers."""
    try:
        d = json.load(open(os.path.join(ROOT, PMC_FILE)))
    except Exception as e:
        return None, "no PMC file (%s)" % e
    want = d.get("source_sha256")
    if not want:
        return d.get("pmc_source_head"), "the PMC file does not say which sources it was taken on"
    sys.path.insert(0, os.path.join(ROOT, "tools"))
    from source_sha import source_sha256
    have = source_sha256(ROOT)
    changed = sorted(f for f in set(want) | set(have) if want.get(f) != have.get(f))
    if changed:
        return d.get("pmc_source_head"), "kernel sources changed since the PMC pass (%s): counters not quoted" % ", ".join(changed)
    return d.get("pmc_source_head"), None


def load_pmc(section=None):
    """HBM bytes per launch and MFMA-busy fraction from the committed PMC passes (separate rocprofv3 --pmc runs of
    this command, tools/profile_r06.sh), by kernel; section = None (headline) | "config5" | "config4" | "net_bf16x3" (the headline's configuration with --net hipx3).  Empty when
    the kernel sources are not the ones the passes were taken on (pmc_status)."""
    if pmc_status()[1] is not None:
        return {}
    try:
        d = json.load(open(os.path.join(ROOT, PMC_FILE)))
        d = d[section] if section else d
        out = {k: {"hbm": v["hbm_bytes_per_launch"]} for k, v in d["kernels"].items()}
        for k, v in d.get("mfma_utilisation", {}).items():
            out.setdefault(k, {})["mfma_busy"] = v["mfma_busy_fraction"]
            out[k]["us_under_profiler"] = v["avg_duration_us_under_the_profiler"]
        return out
    except Exception:
        return {}


def load_net(game, device, weights):
    from caro_ai_amd.lib.model import Net
    torch.manual_seed(0)
    net = Net(game.obs_shape, game.action_space)
    tag = "random-init(seed 0)"
    if weights and os.path.exists(weights):
        net.load_state_dict(torch.load(weights, map_location="cpu"))
        tag = os.path.basename(weights)
    return net.to(device).eval(), tag


# ------------------------------------------------------------------ CPU baseline
def _cpu_worker(args):
    """one host core: the oracle (oracle/caro_oracle.c) playing whole games with the same net,
    torch CPU float32 forward, 1 thread, for ~`seconds`"""
    game_name, S, B, sbt0, weights, seconds, wid = args
    torch.set_num_threads(1)
    from caro_ai_amd.lib.model import Net
    from oracle.oracle import Oracle
    o = Oracle(Oracle.C4) if game_name == "c4" else Oracle(Oracle.MNK, 15, 5)
    torch.manual_seed(0)
    net = Net((2, o.rows, o.cols), o.A)
    if weights and os.path.exists(weights):
        net.load_state_dict(torch.load(weights, map_location="cpu"))
    net.eval()

    def fn(planes, states, players):
        with torch.no_grad():
            lg, vl = net(torch.from_numpy(np.ascontiguousarray(planes)))
            return torch.softmax(lg, dim=1).numpy(), vl.numpy()[:, 0]

    o.set_net(0, fn)
    o.set_net(1, fn)
    t0 = time.perf_counter()
    games = 0
    while time.perf_counter() - t0 < seconds:
        o.set_stream(0, wid * 100000 + games)
        o.play_game(sbt0, S, B, games & 1)
        games += 1
    c = o.counters()
    return games, c["expansions"], c["sims"], c["net_rows"], c["net_calls"], time.perf_counter() - t0


def cpu_baseline(game_name, S, B, sbt0, weights, seconds, procs):
    """CPU port of the reference algorithm on `procs` host cores, one single-threaded process each
    (fork: must run before this process touches the GPU)."""
    import multiprocessing as mp
    procs = max(1, procs)
    work = [(game_name, S, B, sbt0, weights, seconds, w) for w in range(procs)]
    if procs == 1:
        res = [_cpu_worker(work[0])]
    else:
        with mp.get_context("fork").Pool(procs) as pool:
            res = pool.map(_cpu_worker, work)
    games = sum(r[0] for r in res)
    value = sum(r[1] / r[5] for r in res)
    sims = sum(r[2] for r in res)
    rows, calls = sum(r[3] for r in res), sum(r[4] for r in res)
    out = {"value": value, "unit": "node-expansions/s", "cores": procs, "kind": "port",
           "per_core": value / procs,
           "sample": "%d whole games, %d sims, %.1f s wall on %d single-threaded processes, oracle/caro_oracle.c + "
                     "torch CPU fp32 forward (eval-mode BN, %.2f rows/net call)"
                     % (games, sims, max(r[5] for r in res), procs, rows / max(1, calls))}
    try:  # how the port relates to the reference's own Python (both timed on one core of the build container)
        out["ratio_vs_reference"] = json.load(open(os.path.join(ROOT, CPU_RATIO_FILE)))
    except Exception:
        out["ratio_vs_reference"] = None
    return out


# ------------------------------------------------------------------ N > 1 without torchrun
def _free_port():
    import socket
    sk = socket.socket()
    sk.bind(("127.0.0.1", 0))
    port = sk.getsockname()[1]
    sk.close()
    return port


def self_launch(n, argv=None, script=None, poll_s=0.2, timeout_s=1500.0):
    """`python bench.py --gpus N` with WORLD_SIZE unset: start the N ranks ourselves (one child process per GPU,
    the environment torchrun would give them), relay what they print, return the worst exit code.  This parent
    never touches the GPU runtime: the GPUs are counted from the KFD topology in sysfs (parallel.visible_gpu_count,
    honouring ROCR_/HIP_VISIBLE_DEVICES), not through torch.cuda; it skips the CPU baseline.
    All children are polled together: the first one that fails (or the deadline) ends the others at once instead
    of leaving them in init_process_group / a collective until their own time-outs."""
    import subprocess
    from caro_ai_amd import parallel
    have = parallel.visible_gpu_count()
    if have < n and not os.environ.get("CARO_SHARE_GPU"):
        print("[bench] --gpus %d but %d visible GPU(s) in the KFD topology; set CARO_SHARE_GPU=1 (+ "
              "CARO_DIST_BACKEND=gloo) to rehearse on fewer" % (n, have), file=sys.stderr)
        return 2
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", MASTER_PORT=str(_free_port()), WORLD_SIZE=str(n),
               LOCAL_WORLD_SIZE=str(n))
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    env.setdefault("OMP_NUM_THREADS", "1")
    cmd = [sys.executable, script or os.path.abspath(__file__)] + list(sys.argv[1:] if argv is None else argv)
    procs = []
    for r in range(n):  # stdout / stderr are inherited: rank 0's JSON line lands on our stdout as it is
        procs.append(subprocess.Popen(cmd, env=dict(env, RANK=str(r), LOCAL_RANK=str(r))))
    rcs = [None] * n
    deadline = time.monotonic() + timeout_s
    failed = None
    while any(rc is None for rc in rcs):
        for i, p in enumerate(procs):
            if rcs[i] is None:
                rcs[i] = p.poll()
                if rcs[i] not in (None, 0) and failed is None:
                    failed = i
        if failed is not None or time.monotonic() > deadline:
            break
        time.sleep(poll_s)
    if any(rc is None for rc in rcs):  # a rank failed or the deadline passed: end the survivors
        why = "rank %d exited with code %s" % (failed, rcs[failed]) if failed is not None else "deadline passed"
        print("[bench] %s: terminating the other ranks" % why, file=sys.stderr, flush=True)
        for i, p in enumerate(procs):
            if rcs[i] is None:
                p.terminate()
        for i, p in enumerate(procs):
            if rcs[i] is None:
                try:
                    p.wait(timeout=10)
                except subprocess.TimeoutExpired:
                    p.kill()
                    p.wait()
                rcs[i] = 124 if failed is None else 143
    worst = rcs[failed] if failed is not None else max(rcs, key=abs)
    if worst:
        print("[bench] rank exit codes: %s" % rcs, file=sys.stderr)
    return worst


def dist_record(world, device, value_local, want_world):
    """what the collective layer really is in this run: backend, world size, each rank's device and own value.
    Raises if the run is not what --gpus asked for: world size != --gpus, or two ranks on one device without
    CARO_SHARE_GPU (a wrong LOCAL_RANK -> device mapping would otherwise pass as an N-GPU number)."""
    import socket
    import torch.distributed as dist
    props = torch.cuda.get_device_properties(device)
    mine = {"rank": int(os.environ.get("RANK", "0")), "device": str(device),
            "device_name": torch.cuda.get_device_name(device),
            # distinguishes physical GPUs even when every rank calls its own `cuda:0` (per-rank HIP_VISIBLE_DEVICES)
            "device_uuid": str(getattr(props, "uuid", "")) or None,
            "pci_bus_id": getattr(props, "pci_bus_id", None),
            "pid": os.getpid(), "host": socket.gethostname(), "value": value_local}
    if world == 1:
        rec = {"backend": None, "world_size": 1, "ranks": [mine]}
    else:
        ranks = [None] * world
        dist.all_gather_object(ranks, mine)
        rec = {"backend": dist.get_backend(), "world_size": dist.get_world_size(),
               "collectives": "RCCL over xGMI" if dist.get_backend() == "nccl" else "gloo (rehearsal, host memory)",
               "ranks": ranks}
    vals = [r["value"] for r in rec["ranks"]]
    rec["value_min"], rec["value_max"] = min(vals), max(vals)
    rec["shared_gpu"] = bool(os.environ.get("CARO_SHARE_GPU"))
    if rec["world_size"] != want_world:
        raise SystemExit("[bench] world size %d != --gpus %d" % (rec["world_size"], want_world))
    # two ranks sit on one GPU only if host, device index, PCI bus id and uuid ALL agree
    ident = [(r["host"], r["device"], r["pci_bus_id"], r["device_uuid"]) for r in rec["ranks"]]
    if len(set(ident)) != len(ident) and not rec["shared_gpu"]:
        raise SystemExit("[bench] two ranks on one GPU without CARO_SHARE_GPU: %s" % ident)
    return rec


def run_selfcheck(device, engine=True, fault=None):
    """--selfcheck: caro_ai_amd.parallel.selfcheck before any timed work -- every collective of the run once, with
    predictable contents, and a 16-games-per-rank engine whose gathered tuples must equal the same uids played on rank 0
    alone.  A failure ends EVERY rank with exit code 4 and the failing check's name on stderr (the launcher then reports
    4); returns the record that goes into the JSON line."""
    from caro_ai_amd import parallel
    fault = fault or os.environ.get("CARO_SELFCHECK_FAULT") or None
    rank = int(os.environ.get("RANK", "0"))
    try:
        rec = parallel.selfcheck(device, engine_check=parallel.engine_selfcheck(device) if engine else None, fault=fault,
                                 log=(lambda m: print(m, file=sys.stderr, flush=True)) if rank == 0 else None)
    except parallel.SelfcheckError as e:
        print("[bench] selfcheck FAILED (rank %d): %s" % (rank, e), file=sys.stderr, flush=True)
        sys.exit(4)
    return rec


# ------------------------------------------------------------------ one timed configuration
class Leg:
    """One BASELINE.json configuration: engine + nets + the move loop."""

    def __init__(self, args, game_name, G, S, B, arena, rank, world, device, evict=None, node_cap=0, streams=None,
                 stream_mask=None, net=None, split_tiles=True):
        from caro_ai_amd import parallel
        from caro_ai_amd.engine import SelfPlayEngine, StreamedSelfPlay, torch_evaluator
        from caro_ai_amd.lib.game.connect_four import ConnectFour
        from caro_ai_amd.lib.game.tictactoe import TicTacToe
        from caro_ai_amd.lib.model import GemmNet
        self.args, self.game_name, self.G, self.S, self.B, self.arena = args, game_name, G, S, B, arena
        self.rank, self.world, self.device = rank, world, device
        self.net = net or args.net  # inference form of this leg (the labelled bf16x3 leg overrides the run's)
        if game_name == "c4":
            self.game, weights, self.sbt0 = ConnectFour(), args.weights, 10
        else:
            self.game, weights, self.sbt0 = TicTacToe(15, 5), None, 10
        net, self.wtag = load_net(self.game, device, weights)
        extra = {}
        if node_cap:
            extra["node_cap"] = node_cap
        self.evict = int(game_name == "gomoku15") if evict is None or evict < 0 else evict
        if self.evict:
            extra["evict"] = True
            extra.setdefault("node_cap", 4096)
        if arena:
            assert game_name == "c4" and self.net in NET_KERNEL
            self.sbt0 = 0
            net2, wtag2 = load_net(self.game, device, os.path.join(os.path.dirname(weights), "best_025_10600.dat"))
            self.wtag += " vs " + wtag2
            extra.update(n_stores=2, first_player_mode=2)
        self.is_hip = self.net in NET_KERNEL
        self.hipnet = None
        if self.is_hip:
            from caro_ai_amd.net_hip import HipNet
            mode = {"hip": "f32", "hipw": "f32w", "hipx3": "bf16x3"}[self.net]
            self.hipnet = HipNet(net, str(device), mode=mode, split_tiles=split_tiles)
            hipnets = [self.hipnet] + ([HipNet(net2, str(device), mode=mode, split_tiles=split_tiles)] if arena else [])
            make_evaluators = lambda: list(hipnets)
        else:
            fnet = GemmNet(net).to(device).eval()
            make_evaluators = lambda: [torch_evaluator(fnet, form="net")]
        self.n_streams = (args.streams if streams is None else streams) if self.is_hip else 1
        stream_mask = args.stream_mask if stream_mask is None else stream_mask
        common = dict(max_batch=B, steps_before_tau_0=self.sbt0, seed=0, device=str(device), searches_hint=S)
        # staggered mode (every game on its own minibatch clock: even leaf counts per launch, include/caro_hip.h):
        # wherever one wavefront serves a game and nothing needs the second key table
        # (--stagger 2: wherever the geometry allows -- since round 6 also several wavefronts per game with eviction,
        # config 4; measured there and not faster: a 15x15 net launch is thirty rounds of workgroups whatever the leaf count)
        from caro_ai_amd.engine import staggered_geometry
        self.stagger = bool(args.stagger and self.is_hip and staggered_geometry(self.game, B, bool(self.evict))
                            and (args.stagger >= 2 or (game_name == "c4" and B == 8)))
        # a small copy of the same configuration: played to completion before the clock starts, it takes the
        # first-use costs (code objects, torch's clone / cat / cast kernels, allocator growth of the drain path)
        self._warm = SelfPlayEngine(self.game, 16, evaluators=make_evaluators(), uid_base=1 << 40, uid_stride=16,
                                    **{**common, **extra, "searches_hint": 2})
        if self.n_streams > 1:
            self.eng = StreamedSelfPlay(self.game, G, make_evaluators, n_streams=self.n_streams,
                                        partition_cus=bool(stream_mask), stagger=self.stagger, **common, **extra,
                                        **parallel.shard(G, rank, world))
        else:
            self.eng = SelfPlayEngine(self.game, G, evaluators=make_evaluators(), stagger=self.stagger, **common,
                                      **extra, **parallel.shard(G, rank, world))
        self.gatherer = parallel.TupleGatherer(every=args.gather_every)
        self.n_tuples = 0

    def _count(self, d):
        if d is not None:
            self.n_tuples += int(d["z"].shape[0])

    def prewarm(self, moves=48):
        """fixed number of moves (every rank runs the same collectives), 2 x B sims each: games end and drain"""
        from caro_ai_amd import parallel
        tg = parallel.TupleGatherer(every=2)
        w = self._warm
        for _ in range(moves):
            w.search(2, self.B)
            w.step()
            self._count(tg.push(w.drain(recycle=True)))
        self._count(tg.flush())
        w.counters()
        w.close()
        self._warm = None

    def one_step(self):
        # host-pipelined (one engine or several parts): this move is enqueued, then the tuples of the PREVIOUS move's
        # drain are collected while the GPU searches -- no host work sits between two moves on the GPU
        d = self.eng.move(self.S, self.B)
        self._count(self.gatherer.push(d))

    def barrier(self):
        self._count(self.gatherer.push(self.eng.flush()))  # the last enqueued move belongs to the timed region
        self._count(self.gatherer.flush())
        torch.cuda.synchronize(self.device)
        if self.world > 1:
            torch.distributed.barrier()
        torch.cuda.synchronize(self.device)

    def run(self, steps, warmup, profile=True, label="bench"):
        from caro_ai_amd import parallel
        rank = self.rank
        self.prewarm()
        if profile:
            self.eng.profile(True)  # on from the warm-up: the event pool is created before the clock starts
        for i in range(warmup):
            self.one_step()
            if rank == 0 and i % 5 == 0:
                print("[%s] warmup step %d" % (label, i), file=sys.stderr, flush=True)
        self.barrier()
        c0 = self.eng.counters()
        if profile:
            self.eng.profile_read(reset=True)
        self.n_tuples = 0
        stamps = []
        t0 = time.perf_counter()
        for i in range(steps):
            self.one_step()
            stamps.append(time.perf_counter())
        self.barrier()
        dt = time.perf_counter() - t0
        c1 = self.eng.counters()
        prof = self.eng.profile_read(reset=True) if profile else None
        self.eng.profile(False)
        if rank == 0:  # host-side step times (each ends with the drain's sync): shows where a slow step sits
            ms = np.diff(np.array([t0] + stamps)) * 1e3
            print("[%s] ms per step: " % label + " ".join("%.2f" % x for x in ms) + " | closing barrier %.2f"
                  % ((t0 + dt - stamps[-1]) * 1e3), file=sys.stderr, flush=True)
        delta = {k: c1[k] - c0[k] for k in c1}
        tot = torch.tensor([delta["expansions"], delta["sims"], delta["levels"], delta["plies"], delta["finished"]],
                           dtype=torch.float64, device=self.device)
        tmax = torch.tensor([dt], dtype=torch.float64, device=self.device)
        parallel.allreduce_sum(tot)
        parallel.allreduce_max(tmax)
        rep = self._report(steps, warmup, float(tmax.item()), [float(x) for x in tot.tolist()], delta, prof)
        rep["value_local"] = delta["expansions"] / dt  # this rank's own rate on its own clock
        return rep

    def _report(self, steps, warmup, dt, tot, delta, prof):
        args, game, G, S, B = self.args, self.game, self.G, self.S, self.B
        exp_all, sims_all, levels_all, plies_all, fin_all = tot
        n_streams = self.n_streams
        section = "net_bf16x3" if (self.net == "hipx3" and self.game_name == "c4" and G == 1024 and not self.arena and S == 25) else \
            None if (self.game_name == "c4" and G == 1024 and not self.arena and S == 25) else \
            "config5" if (self.arena and G == 512 and S == 100) else \
            "config4" if (self.game_name == "gomoku15" and G == 1024 and S == 50) else "none"
        pmc = load_pmc(section) if section != "none" else {}
        A, KW, HW = game.action_space, game.key_words, game.obs_shape[1] * game.obs_shape[2]
        bytes_per_level = 12 * A + 8 * KW + 28            # N,Q,P rows + key probe + backup RMW
        bytes_per_exp = 16 * HW + 20 * A + 8 * KW + 12
        # FLOPs of one leaf through lib/model.py Net (2 x MAC): conv_in, 5 residual 3x3 convs, 1x1 heads, FC heads
        flops_per_leaf = 2.0 * (HW * 64 * 18 + 5 * HW * 64 * 576 + HW * 3 * 64 + 20 * HW + 20 + 2 * HW * A)
        traffic_note = ("HBM bytes per launch from %s (separate rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes of this "
                        "configuration, tools/profile_r06.sh, corrected as MI355X_MICROARCH.md prescribes: read side "
                        "x2); not measured in this run" % PMC_FILE)
        kernel_us = 0.0
        roofline = roofline_tree = None
        pmc_head, pmc_stale = pmc_status()
        # What an event pair adds to the kernel it brackets (the dispatch behind an event's barrier packet; rocprofv3
        # sees the kernel alone): calibrated in the run itself.  Behind every fourth sampled net launch the engine
        # brackets ONE launch of an empty kernel (E1 = K0 + o) and TWO (E2 = 2 K0 + g + o):  o = 2 E1 - E2 (+ g, the
        # sub-microsecond gap between two dependent launches: o is underestimated, the kernels' times stay conservative).
        gap_s, ncal, e1, e2 = 0.0, 0, 0.0, 0.0
        if prof is not None and prof.get("null1", (0, 0))[1] > 0 and prof.get("null2", (0, 0))[1] > 0:
            e1 = prof["null1"][0] * 1e-3 / prof["null1"][1]
            e2 = prof["null2"][0] * 1e-3 / prof["null2"][1]
            gap_s, ncal = min(max(2.0 * e1 - e2, 0.0), e1), prof["null1"][1]
        timing_note = ("HIP-event pairs on the launch stream around a sample of the launches (every 23rd minibatch), minus "
                       "what a pair adds to the kernel it brackets: %.2f us = 2 E1 - E2 of pairs around one / two launches "
                       "of an empty kernel recorded beside them (E1 %.2f us, E2 %.2f us, %d calibrations)"
                       % (gap_s * 1e6, e1 * 1e6, e2 * 1e6, ncal))
        if prof is not None and prof["select"][1] > 0:
            ms, n = prof["select"]
            avg_s = max(ms * 1e-3 / n - gap_s, 1e-9)  # timed on a sample of the launches (every 23rd minibatch, all indices equally)
            n_launches = steps * S * n_streams
            levels_per_launch = delta["levels"] / n_launches
            achieved = levels_per_launch * bytes_per_level / avg_s / 1e9
            fused = prof.get("compact", (0, 0))[1] == 0
            lpd = 8 if A == 7 else 16 if A <= 16 else 32 if A <= 32 else 64
            tname = (("k_tree_stag_mw" if B * lpd > 64 else "k_tree_stag") if self.stagger else
                     "k_tree_mw" if B * lpd > 64 else "k_tree") if fused else "k_select"
            others = {k: (max(v[0] * 1e3 / v[1] - gap_s * 1e6, 0.0) if v[1] else None) for k, v in prof.items()
                      if k not in ("select", "net", "null1", "null2")}
            roofline_tree = {"bound": "hbm", "kernel": tname, "achieved": achieved, "peak": HBM_PEAK_GBS,
                             "unit": "GB/s", "frac": achieved / HBM_PEAK_GBS, "traffic": pmc.get(tname, {}).get("hbm"),
                             "traffic_source": traffic_note if pmc.get(tname) else None,
                             "algorithmic_bytes_per_launch": levels_per_launch * bytes_per_level,
                             "avg_launch_us": avg_s * 1e6, "launches": n_launches, "launches_timed": n,
                             "event_pair_gap_us": gap_s * 1e6, "timing": timing_note,
                             "pmc_source_head": pmc_head, "pmc_stale": pmc_stale,
                             "levels_per_launch": levels_per_launch, "bytes_per_level": bytes_per_level,
                             "other_kernels_us": others}
            kernel_us += S * avg_s * 1e6 + sum(v for v in others.values() if v)
        if prof is not None and prof.get("net", (0, 0))[1] > 0:
            ms, n = prof["net"]
            avg_s = max(ms * 1e-3 / n - gap_s, 1e-9)
            leaves_per_launch = delta["expansions"] / (steps * S * n_streams)
            achieved = leaves_per_launch * flops_per_leaf / avg_s / 1e12
            peak = MFMA_BF16_PEAK_TFS if self.net == "hipx3" else MFMA_F32_PEAK_TFS
            kname = NET_KERNEL[self.net]
            if self.net == "hipw" and self.hipnet.mode == "f32w2":
                kname = "k_net_forward_w2"  # large boards: the 2-D Winograd form
            # What the matrix pipe EXECUTES per launch.  hipw, row-Winograd F(2,3): 60 transformed taps x 32 MFMAs x
            # 8 waves x 4096 flop per workgroup of TB boards, tile padding and the last partial tile included -- 2/3 of
            # the 3x3 multiplies of the direct form; hipw on large boards, 2-D Winograd F(2x2,3x3): 80 taps x 4 blocks
            # of 32 MFMAs per board -- 4/9.  hip: the direct form executes its algorithmic count.
            executed = None
            if self.net in ("hipw", "hipx3"):
                tb = self.hipnet.L.caro_net_boards_per_workgroup(self.hipnet.h)
                wg_flops = self.hipnet.workgroup_mfma_flops()
                executed = math.ceil(leaves_per_launch / tb) * wg_flops / avg_s / 1e12
            ex = executed if executed is not None else achieved
            roofline = {"bound": "mfma", "kernel": kname,
                        # `achieved` / `frac` = flops the MFMA pipe executes per launch / launch time
                        # (a utilisation figure, <= 1 by construction, comparable with the PMC busy fraction);
                        # the direct-convolution count is kept beside it as algorithmic_*
                        "achieved": ex, "peak": peak, "unit": "TFLOP/s", "frac": ex / peak,
                        "algorithmic_achieved": achieved, "algorithmic_frac": achieved / peak,
                        "note": ("bf16x3 leg: flops = the six bf16 part products per multiply the pipe executes, peak = "
                                 "the bf16 MFMA peak; a utilisation figure of the pipe, not comparable with the fp32 legs' "
                                 "-- compare avg_launch_us.  " if self.net == "hipx3" else "") +
                                "achieved = flops the matrix pipe executes per launch (Winograd form: fewer multiplies "
                                "than the direct convolution; tile padding included) / launch time measured with HIP "
                                "events in this run; algorithmic_* prices SURVEY 8(d)'s direct-convolution flops per "
                                "leaf over the same time and can exceed 1 for a Winograd kernel; mfma_busy_pmc is the "
                                "hardware's own count from the committed --pmc pass"
                                + ("; a launch here = k_net_forward_w2 (trunk, 1x1 convolutions) + k_net_heads (the FC "
                                   "heads of the launch, 32 boards per workgroup): avg_launch_us spans both"
                                   if kname == "k_net_forward_w2" else ""),
                        "traffic": pmc.get(kname, {}).get("hbm"),
                        "traffic_source": (traffic_note + "; the net's algorithmic bytes per launch are planes + "
                                           "priors + the weights once (%.2f MB): the measured figure is higher because "
                                           "every XCD's L2 reads the weight taps itself -- harmless for a kernel bound "
                                           "by the matrix pipe"
                                           % ((leaves_per_launch * (8 * HW + 4 * A + 4)
                                               + 4.0 * ((45 * 6144 if self.net == "hipx3" else 60 * 4096)  # residual weights: 45 taps x 24 576 B of bf16 parts | 60 transformed taps of 4096 floats
                                                        + 18 * 64 + 64 + 5 * 64 + 3 * 64 + 3 + 20 * HW + 41
                                                        + A * 2 * HW + A)) / 1e6))
                        if pmc.get(kname) else None,
                        "mfma_busy_pmc": pmc.get(kname, {}).get("mfma_busy"),
                        "mfma_busy_note": "SQ_VALU_MFMA_BUSY_CYCLES / (kernel cycles x 1024 SIMDs), rocprofv3 --pmc pass of "
                                          "this configuration (%s)" % PMC_FILE
                        if pmc.get(kname, {}).get("mfma_busy") else None,
                        "avg_launch_us": avg_s * 1e6, "launches_timed": n, "leaves_per_launch": leaves_per_launch,
                        "flops_per_leaf": flops_per_leaf,
                        "event_pair_gap_us": gap_s * 1e6, "timing": timing_note,
                        "pmc_source_head": pmc_head, "pmc_stale": pmc_stale,
                        "clock_note": "peak is the guide's figure at 2.4 GHz; in steady state the device holds 2.36-2.37 GHz "
                                      "(workgroup clocks stamped inside the engine's own launches, "
                                      "profiles/r04_net_launch_clock.txt; not measured in this run), the first milliseconds "
                                      "after an idle gap (a host synchronisation) run at 2.1 GHz and climb."
                                      + (" A full-tile workgroup of this leg is 312 k cycles = 132 us at that clock; "
                                         "avg_launch_us is above it because it averages in the 7.5 % of launches whose leaf "
                                         "count exceeds one round of full tiles (1 536) and pay a second, short round "
                                         "(mean 190 us)." if section is None and kname == "k_net_forward_w" else "")}
            kernel_us += S * avg_s * 1e6
        if roofline is None:  # torch evaluators: the net is not our kernel; the tree walk is the dominant own kernel
            roofline = roofline_tree
        ms_per_step = dt * 1e3 / steps
        what = "arena matches (two nets, one tree per player)" if self.arena else "self-play games"
        board = "Connect4 6x7" if self.game_name == "c4" else "m,n,k 15x15 k=5"
        # named after the kernel this leg really launches (HipNet.mode: "hipw" picks the row or the 2-D form by board)
        if self.hipnet is not None:
            netdesc = {"f32": "fused HIP MFMA kernel k_net_forward, direct 3x3 convs,",
                       "f32w1": "fused HIP MFMA kernel k_net_forward_w, 3x3 convs in row-Winograd F(2,3) form,",
                       "f32w2": "fused HIP MFMA kernels k_net_forward_w2 + k_net_heads, 3x3 convs in 2-D Winograd "
                                "F(2x2,3x3) form, FC heads batched 32 boards per workgroup,",
                       "bf16x3": "fused HIP MFMA kernel k_net_forward_x3, direct 3x3 convs, bf16x3 split operands, fp32 "
                                 "accumulate (NOT bit-identical to the fp32 forms: within tests/test_gpu_net.py's tolerance);"
                                 " conv_in, epilogues and heads in"}[self.hipnet.mode]
        else:
            netdesc = {"gemm": "torch gather+GEMM"}[self.net]
        return {
            "value": exp_all / dt, "unit": "node-expansions/s", "steps": steps, "warmup": warmup,
            "ms_per_step": ms_per_step,
            "data": "synthetic (self-play from empty boards; net weights: %s)" % self.wtag,
            "config": {"workload": "%s %d concurrent %s/GPU, %dx%d = %d sims/move, tau=1 for %d plies"
                                   % (board, G, what, S, B, S * B, self.sbt0),
                       "games_per_gpu": G, "searches": S, "batch": B, "net": "lib/model.py Net, %s fp32" % netdesc,
                       "streams_per_gpu": n_streams,
                       "schedule": "staggered: every game on its own minibatch clock, ply inside the tree kernel, "
                                   "finished games parked and restarted in place" if self.stagger else "lock-step",
                       "parallelism": "games sharded x%d, tuples all-gathered every %d moves"
                                      % (self.world, args.gather_every)},
            "per_gpu": exp_all / dt / self.world,
            "sims_per_s": sims_all / dt, "plies_per_s": plies_all / dt, "games_per_s": fin_all / dt,
            "games_finished": int(fin_all), "live_nodes": self.live_nodes(),
            "net_rows_per_s": exp_all / dt, "mean_depth": levels_all / max(1.0, sims_all),
            "expansions_per_sim": exp_all / max(1.0, sims_all),
            "algorithmic_GBps": (levels_all * bytes_per_level + exp_all * bytes_per_exp) / dt / 1e9,
            "overflows": delta["overflows"], "evict": bool(self.evict),
            # per step: the kernels' own time (HIP events, sampled) against the wall clock
            "kernel_ms_per_step": kernel_us / 1e3 if kernel_us else None,
            "idle_frac": (1.0 - kernel_us / 1e3 / ms_per_step) if kernel_us else None,
            "roofline": roofline, "roofline_tree": roofline_tree,
        }

    def live_nodes(self):
        """nodes held per tree right now (after eviction, if on): mean / max over the games"""
        if self.n_streams > 1:
            return None
        ts = self.eng.tree_live()
        return {"mean": float(ts.mean()), "max": int(ts.max()), "cap": int(self.eng.cfg.node_cap)}

    def sustained(self, moves):
        """`moves` further moves of the same engine, finished games recycled: steady-state rate on its own clock"""
        self.barrier()
        c0 = self.eng.counters()
        t0 = time.perf_counter()
        for i in range(moves):
            self.one_step()
        self.barrier()
        dt = time.perf_counter() - t0
        c1 = self.eng.counters()
        d = {k: c1[k] - c0[k] for k in c1}
        return {"moves": moves, "seconds": dt, "value": d["expansions"] / dt, "unit": "node-expansions/s",
                "ms_per_step": dt * 1e3 / moves, "games_finished": d["finished"], "games_per_s": d["finished"] / dt,
                "plies": d["plies"], "overflows": d["overflows"], "mean_depth": d["levels"] / max(1, d["sims"]),
                "note": "continues the headline engine after the timed steps; not part of `value`"}

    def close(self):
        self.eng.close()


def train_loop_record(args, device, headline, games=4096, concurrent=1024):
    """The number a user of train.py sees (VERDICT r5 task 1): `speed_nodes` of
    caro_ai_amd.train.self_play -- the reference's self-play phase, train.py:25-59 -- for `games` connect-four games on
    `concurrent` slots at the headline's 25 x 8 sims/move, on the wall clock of the WHOLE call: HipNet packing and
    upload, engine construction (4.6 GB of tree tables), the games with their ramp and tail (exactly `games` games:
    no slot restarts beyond the wanted set), the gather into the replay buffer.  Called twice: on a cold cache
    (`speed_nodes`: what the first iteration of train.fit pays) and again (`reused`: engine and HipNet restarted in
    place -- every later iteration)."""
    from caro_ai_amd import train
    from caro_ai_amd.lib.game.connect_four import ConnectFour
    game = ConnectFour()
    net, wtag = load_net(game, device, args.weights)
    rb = train.DeviceReplayBuffer(game, 1 << 18, device)
    train.release_engines()
    # (first-use costs of this process -- torch's cat / index kernels behind the gather and the replay ring, code objects --
    # are taken by a 16-game call on a throw-away engine, as the headline's prewarm does; construction of the real engine
    # and of its HipNet stay inside the cold call)
    train.self_play(game, train.DeviceReplayBuffer(game, 4096, device), net, 16, device=str(device), searches=2,
                    batch=args.batch, stagger=True, reuse=False)
    from caro_ai_amd import net_hip
    net_hip.release_hipnets()
    torch.cuda.synchronize(device)
    kw = dict(device=str(device), searches=args.searches, batch=args.batch, concurrent=concurrent, stagger=True)
    calls = []
    for i in range(2):
        torch.cuda.synchronize(device)
        t0 = time.perf_counter()
        sp = train.self_play(game, rb, net, games, seed=i, uid_base=i * games, **kw)
        torch.cuda.synchronize(device)
        sp["seconds_wall"] = time.perf_counter() - t0  # with the closing synchronisation: >= sp["seconds"]
        sp["speed_nodes_wall"] = sp["nodes"] / sp["seconds_wall"]
        calls.append(sp)
    # the same as a STREAM (train.self_play_stream, the CLI's default): slots restart at once, a call takes the first
    # `games` games that finish, games in flight carry over -- three calls, the first one starts the stream
    train.release_engines()
    stream = []
    for i in range(3):
        torch.cuda.synchronize(device)
        t0 = time.perf_counter()
        sp = train.self_play_stream(game, rb, net, games, device=str(device), searches=args.searches, batch=args.batch,
                                    concurrent=concurrent, uid_base=(2 + i) * games)
        torch.cuda.synchronize(device)
        sp["seconds_wall"] = time.perf_counter() - t0
        sp["speed_nodes_wall"] = sp["nodes"] / sp["seconds_wall"]
        stream.append(sp)
    train.release_engines()
    # ... and the stream on two half-engines (train.self_play_stream(streams=2), `python -m caro_ai_amd.train --streams 2`):
    # the product form of this line's `two_streams` record
    stream2 = []
    for i in range(3):
        torch.cuda.synchronize(device)
        t0 = time.perf_counter()
        sp = train.self_play_stream(game, rb, net, games, device=str(device), searches=args.searches, batch=args.batch,
                                    concurrent=concurrent, uid_base=(6 + i) * games, streams=2)
        torch.cuda.synchronize(device)
        sp["seconds_wall"] = time.perf_counter() - t0
        sp["speed_nodes_wall"] = sp["nodes"] / sp["seconds_wall"]
        stream2.append(sp)
    train.release_engines()
    torch.cuda.empty_cache()
    keep = ("speed_nodes_wall", "speed_nodes_play", "nodes", "steps", "games", "rows", "seconds_wall", "seconds_setup",
            "seconds_play", "seconds_gather", "engine_reused", "passes")
    cold, warm = ({k: c[k] for k in keep} for c in calls)
    return {"speed_nodes": cold["speed_nodes_wall"], "unit": "node-expansions/s", "frac_of_headline": cold["speed_nodes_wall"] / headline,
            # what `python -m caro_ai_amd.train` itself runs (its default is the stream form): the FIRST call -- engine
            # construction, HipNet, the stream's ramp inside -- and a later one
            "cli_default": {"form": "stream", "speed_nodes_first_call": stream[0]["speed_nodes_wall"],
                            "frac_of_headline_first_call": stream[0]["speed_nodes_wall"] / headline,
                            "speed_nodes": stream[-1]["speed_nodes_wall"],
                            "frac_of_headline": stream[-1]["speed_nodes_wall"] / headline},
            "reused": dict(warm, speed_nodes=warm["speed_nodes_wall"], frac_of_headline=warm["speed_nodes_wall"] / headline),
            "cold": cold,
            "stream": {"speed_nodes": stream[-1]["speed_nodes_wall"], "frac_of_headline": stream[-1]["speed_nodes_wall"] / headline,
                       "calls": [{k: c[k] for k in keep} for c in stream],
                       "what": "train.self_play_stream, three consecutive calls of %d games (the first starts the stream): "
                               "no call plays a sparse tail, the games in flight at its end finish in the next call" % games},
            "stream_two_engines": {"speed_nodes": stream2[-1]["speed_nodes_wall"],
                                   "frac_of_headline": stream2[-1]["speed_nodes_wall"] / headline,
                                   "calls": [{k: c[k] for k in keep} for c in stream2],
                                   "what": "the same with streams=2 (`train.py --streams 2`, opt-in): the slots as two engines "
                                           "of half the slots on two HIP streams, full net tiles only"},
            "tail_note": "exact form: the wanted games are played to the end and nothing beyond them is started, so the "
                         "last passes of a call carry few live games (`passes` against games x mean plies / slots) and "
                         "each of those passes costs its 25 launch pairs at their small-launch floor; the stream form "
                         "has no such passes",
            "games": games, "concurrent": concurrent, "searches": args.searches, "batch": args.batch,
            "weights": wtag,
            "what": "caro_ai_amd.train.self_play(n_games=%d, concurrent=%d, searches=%d, batch=%d, stagger=True): "
                    "node-expansions / wall seconds of the whole call incl. engine construction (cold) or in-place restart "
                    "(reused), HipNet upload, the games' ramp and tail, and the gather into the device replay buffer; "
                    "seconds_play = the move loop alone" % (games, concurrent, args.searches, args.batch)}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=40)
    ap.add_argument("--warmup", type=int, default=20)
    ap.add_argument("--games", type=int, default=1024, help="concurrent games per GPU")
    ap.add_argument("--searches", type=int, default=25)
    ap.add_argument("--batch", type=int, default=8)
    ap.add_argument("--game", default="c4", choices=["c4", "gomoku15"])
    ap.add_argument("--arena", action="store_true",
                    help="BASELINE config 5: best_026 vs best_025, one tree per player, tau = 0 from move 0 "
                         "(use with --games 512 --searches 100)")
    ap.add_argument("--node-cap", type=int, default=0, help="nodes per tree (0 = searches*batch*cells bound)")
    ap.add_argument("--evict", type=int, default=-1,
                    help="drop unreachable nodes after every move (result-neutral); default: on for gomoku15")
    ap.add_argument("--weights", default=os.path.join(ROOT, "caro_ai_amd", "data", "weights", "best_026_12000.dat"))
    ap.add_argument("--net", default="hipw", choices=["hipw", "hip", "gemm", "hipx3"],
                    help="inference form of lib/model.py Net: hipw = fused HIP fp32 MFMA kernel, 3x3 convs in Winograd form "
                         "(row form F(2,3); 2-D form F(2x2,3x3) on 13x13 .. 15x15 boards) -- the default and the only form "
                         "that is tuned; the other two are A/B baselines: hip = the same kernel structure with direct 3x3 "
                         "convs, gemm = PyTorch-ROCm gather + GEMM (leaf counts cross to the host); hipx3 = the extra bf16x3 "
                         "split-operand form (every fp32 trunk operand as three bfloat16 parts, six part products on the bf16 "
                         "MFMA, fp32 accumulate): not bit-identical to fp32, the line then says so in `dtype`")
    ap.add_argument("--streams", type=int, default=1,
                    help="split the games of a GPU over this many engines on separate HIP streams (tree kernels of "
                         "one part overlap the net kernel of another)")
    ap.add_argument("--stream-mask", type=int, default=1, help="with --streams > 1: confine each part to its own CU slice")
    ap.add_argument("--stagger", type=int, default=1,
                    help="1: staggered mode for connect four with batch 8 (the headline, config 5); 2: wherever the geometry has "
                         "whole wavefronts per game (also config 4); 0: lock-step")
    ap.add_argument("--gather-every", type=int, default=8,
                    help="N > 1: all-gather the finished games' tuples every this many moves (one payload message)")
    ap.add_argument("--cpu-seconds", type=float, default=12.0)
    ap.add_argument("--cpu-procs", type=int, default=0, help="host cores for the CPU baseline (0 = all this process may use: affinity and cgroup quota, capped at 32)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-profile", action="store_true", help="do not record HIP events around the kernels")
    ap.add_argument("--no-extra-configs", action="store_true",
                    help="skip the config4 (15x15) / config5 (arena) legs that follow the headline loop at N = 1")
    ap.add_argument("--sustained-moves", type=int, default=200,
                    help="N = 1 headline: moves of the `sustained` sub-record that follows the timed loop (0 = skip)")
    ap.add_argument("--train-loop-games", type=int, default=4096,
                    help="N = 1 headline: games of the `train_loop` sub-record (train.self_play on --games slots; 0 = skip)")
    ap.add_argument("--config4-warmup", type=int, default=40,
                    help="moves played at full size before config4's timed moves (mid-game measurement)")
    ap.add_argument("--config4-steps", type=int, default=8)
    ap.add_argument("--selfcheck", action="store_true",
                    help="before the timed loop: every collective of the run once with predictable contents, distinct GPUs per "
                         "rank, and a small engine whose gathered tuples must equal the same games played on rank 0 alone "
                         "(caro_ai_amd.parallel.selfcheck); any mismatch ends all ranks with exit code 4 and the check's name")
    args = ap.parse_args()

    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        # not under torchrun: become the launcher (nothing in this process has touched the GPU yet)
        sys.exit(self_launch(args.gpus))

    from caro_ai_amd import parallel

    # CPU baseline first (N = 1 only): it forks worker processes, which must happen before the GPU is touched
    cpu_line = None
    headline = args.game == "c4" and not args.arena
    if int(os.environ.get("WORLD_SIZE", "1")) == 1 and not args.no_cpu_baseline and not args.arena:
        procs = args.cpu_procs or min(32, host_cores())
        w = args.weights if args.game == "c4" else None
        print("[bench] cpu baseline on %d cores for %.0f s" % (procs, args.cpu_seconds), file=sys.stderr, flush=True)
        cpu_line = cpu_baseline(args.game, args.searches, args.batch, 10, w, args.cpu_seconds, procs)

    rank, local_rank, world = parallel.init()
    if world != args.gpus:
        print("[bench] --gpus %d but WORLD_SIZE=%d: launch with torch.distributed.run --nproc-per-node %d, or run "
              "`python bench.py --gpus %d` with WORLD_SIZE unset (it starts the ranks itself)"
              % (args.gpus, world, args.gpus, args.gpus), file=sys.stderr)
        sys.exit(2)
    # one rank per GPU; CARO_SHARE_GPU=1 maps every rank to cuda:0 (rehearsal of the N > 1 path on a 1-GPU box)
    device = torch.device("cuda", 0 if os.environ.get("CARO_SHARE_GPU") else local_rank)
    torch.cuda.set_device(device)
    selfcheck = run_selfcheck(device) if args.selfcheck else None

    leg = Leg(args, args.game, args.games, args.searches, args.batch, args.arena, rank, world, device,
              evict=args.evict, node_cap=args.node_cap)
    res = leg.run(args.steps, args.warmup, profile=not args.no_profile)
    dist_rec = dist_record(world, device, res.pop("value_local"), args.gpus)
    sustained = None
    if world == 1 and headline and args.sustained_moves > 0:
        # the timed region above is ~0.1 s; the same engine keeps playing: a steady-state figure with finished games
        # recycled all along (its own clock, not part of `value`)
        sustained = leg.sustained(args.sustained_moves)
    leg.close()
    del leg
    torch.cuda.empty_cache()

    extras, extras_rc = {}, 0
    two_streams = None
    if world == 1 and headline and args.net == "hipw" and args.streams == 1 and args.stagger and args.games % 2 == 0:
        # The same games as two engines of half the games on two HIP streams, the net kernel with FULL tiles only (no
        # K-split tiles: a half's launch then takes half the compute units and the two halves' launches run side by
        # side, each half's tree kernels beside the other half's net launch).  A SCHEDULING variant reported beside
        # `value`, which stays the one-engine figure every round has reported and the one the kernels' launch times and
        # rooflines of this line belong to.  With K-split tiles two streams lose (each half's launch fills the chip).
        try:
            x = Leg(args, game_name="c4", G=args.games, S=args.searches, B=args.batch, arena=False, rank=rank, world=world,
                    device=device, streams=2, stream_mask=0, split_tiles=False)
            r2 = x.run(args.steps, args.warmup, profile=False, label="two_streams")
            x.close()
            del x
            torch.cuda.empty_cache()
            if r2["overflows"]:
                raise RuntimeError("two_streams: %d minibatches overflowed the node pool" % r2["overflows"])
            two_streams = {"value": r2["value"], "unit": r2["unit"], "ms_per_step": r2["ms_per_step"],
                           "vs_value": r2["value"] / res["value"], "streams_per_gpu": 2, "overflows": r2["overflows"],
                           "games_finished": r2["games_finished"],
                           "note": "the headline's games as two engines of half the games on two HIP streams (no CU "
                                   "mask), k_net_forward_w with full tiles only (HipNet(split_tiles=False)); a scheduling "
                                   "variant, not a kernel speed-up: `value` and the rooflines of this line are one engine's"}
        except Exception as e:
            import traceback
            traceback.print_exc()
            two_streams = {"error": repr(e)}
            extras_rc = 1
    train_loop = None
    if world == 1 and headline and args.train_loop_games > 0 and args.net == "hipw" and args.stagger:
        try:
            train_loop = train_loop_record(args, device, res["value"], games=args.train_loop_games,
                                           concurrent=min(args.games, args.train_loop_games))
        except Exception as e:
            import traceback
            traceback.print_exc()
            train_loop = {"error": repr(e)}
            extras_rc = 1
    if world == 1 and headline and not args.no_extra_configs and args.net in NET_KERNEL and args.streams == 1:
        # BASELINE.json configs 5 and 4 at full size (parity of both is tests/ business).  config 4 is measured in
        # MID-GAME: --config4-warmup moves at full size first, so that trees are deep, eviction has work to do and
        # games finish inside the timed moves.
        keep = ("value", "unit", "steps", "warmup", "ms_per_step", "config", "data", "sims_per_s", "games_per_s",
                "games_finished", "mean_depth", "expansions_per_sim", "overflows", "evict", "live_nodes",
                "kernel_ms_per_step", "roofline", "roofline_tree")

        def side_leg(key, spec, st, wu, profile):
            x = Leg(args, rank=rank, world=world, device=device, **spec)
            r = x.run(st, wu, profile=profile, label=key)
            x.close()
            del x
            torch.cuda.empty_cache()
            if r["overflows"]:  # the games are no longer the reference's: the leg fails (extras_rc)
                raise RuntimeError("%s: %d minibatches overflowed the node pool" % (key, r["overflows"]))
            return {k: r[k] for k in keep}

        c4 = dict(game_name="gomoku15", G=1024, S=50, B=8, arena=False)
        try:
            extras["config5"] = side_leg("config5", dict(game_name="c4", G=512, S=100, B=8, arena=True), 6, 3,
                                         not args.no_profile)
        except Exception as e:  # the headline line survives a failing side leg, but says so: extras_rc != 0
            import traceback
            traceback.print_exc()
            extras["config5"] = {"error": repr(e)}
            extras_rc = 1
        try:
            # config 4 twice.  (1) ONE stream, HIP events on: this is the leg's `value` -- the form BASELINE's configuration
            # names and every earlier round's one-stream figure compares with -- and its launches do not overlap, so
            # `roofline` / `roofline_tree` / `kernel_ms_per_step` price single launches.  (2) the same games as two engines
            # of 512 on two streams (no CU mask), events off: a 15x15 net launch is 30 rounds of workgroups, the last one
            # partly filled, and the other half's tree kernels run there -- reported beside it as `two_streams` (ADVICE r5:
            # a scheduling change, not to be read as a kernel speed-up).  One board per net workgroup: a game's bits do not
            # depend on the split (tests/test_gpu_tuples.py::test_gomoku15_games_do_not_depend_on_the_stream_split).
            one = side_leg("config4", c4, args.config4_steps, args.config4_warmup, not args.no_profile)
            two = side_leg("config4-2streams", dict(c4, streams=2, stream_mask=0), args.config4_steps,
                           args.config4_warmup, False)
            one["two_streams"] = {"value": two["value"], "unit": two["unit"], "ms_per_step": two["ms_per_step"],
                                  "streams_per_gpu": 2, "games_finished": two["games_finished"],
                                  "overflows": two["overflows"],
                                  "note": "the same 1024 games as two engines of 512 on two HIP streams, in this run: one "
                                          "half's tree kernels run in the partly filled last round of the other half's net "
                                          "launch; bit-identical per game"}
            if args.net == "hipw":
                # config 4 with the opt-in bf16x3 kernel (k_net_forward_x3 + k_net_heads): labelled, beside the leg's value
                x4 = side_leg("config4-bf16x3", dict(c4, net="hipx3"), args.config4_steps, args.config4_warmup, False)
                one["net_bf16x3"] = {"value": x4["value"], "unit": x4["unit"], "ms_per_step": x4["ms_per_step"],
                                     "vs_config4": x4["value"] / one["value"], "games_finished": x4["games_finished"],
                                     "overflows": x4["overflows"],
                                     "note": "extra, not this leg's value: the same configuration with lib/model.py Net in "
                                             "bf16x3 split-operand arithmetic, fp32 accumulate (direct 3x3 form, "
                                             "k_net_forward_x3 + k_net_heads); not bit-identical to the fp32 kernels"}
            extras["config4"] = one
        except Exception as e:
            import traceback
            traceback.print_exc()
            extras["config4"] = {"error": repr(e)}
            extras_rc = 1
        if args.net == "hipw" and args.game == "c4":
            # A LABELLED EXTRA LEG, never the headline: the headline's configuration with the residual trunk in bf16x3
            # split-operand arithmetic (caro_net_enable_split_bf16).  Its games are not bit-identical to the fp32 kernels'
            # (the net's outputs differ within tests/test_gpu_net.py's tolerance), its roofline is priced against the
            # bf16 MFMA peak, and `dtype` of this line stays "f32".
            try:
                x3 = side_leg("net_bf16x3", dict(game_name="c4", G=args.games, S=args.searches, B=args.batch, arena=False,
                                                 net="hipx3"), 10, 5, not args.no_profile)
                x3["vs_headline"] = x3["value"] / res["value"]
                # the same leg as two engines of half the games on two HIP streams (no CU mask): with this kernel a
                # half's tree kernels overlap the other half's net launch to a gain (the fp32 kernel loses that way)
                x3b = side_leg("net_bf16x3-2streams", dict(game_name="c4", G=args.games, S=args.searches, B=args.batch,
                                                           arena=False, net="hipx3", streams=2, stream_mask=0), 10, 5, False)
                x3["two_streams"] = {"value": x3b["value"], "unit": x3b["unit"], "ms_per_step": x3b["ms_per_step"],
                                     "vs_headline": x3b["value"] / res["value"], "streams_per_gpu": 2,
                                     "overflows": x3b["overflows"]}
                x3["note"] = ("extra leg, not the headline: lib/model.py Net with bf16x3 split operands, fp32 accumulate "
                              "(k_net_forward_x3); 10 moves after 5 of warm-up, the headline's games and seeds")
                extras["net_bf16x3"] = x3
            except Exception as e:
                import traceback
                traceback.print_exc()
                extras["net_bf16x3"] = {"error": repr(e)}
                extras_rc = 1

    if rank == 0:
        out = {"metric": "self-play MCTS node-expansions/sec/GPU (Connect4, 200 sims/move); 1->8 GPU scaling"
               if args.game == "c4" else "self-play MCTS node-expansions/sec/GPU (15x15 k=5)",
               "value": res["value"], "unit": res["unit"], "n_gpus": world, "steps": args.steps,
               "warmup": args.warmup, "ms_per_step": res["ms_per_step"], "higher_is_better": True,
               "scaling": "weak", "vs_baseline": None,
               "dtype": "f32" if args.net != "hipx3" else "f32 (residual trunk: bf16x3 split operands, f32 accumulate)"}
        out.update({k: v for k, v in res.items() if k not in out})
        out["dist"] = dist_rec
        out["selfcheck"] = selfcheck
        out["sustained"] = sustained
        out["two_streams"] = two_streams
        out["train_loop"] = train_loop
        out["cpu_baseline"] = cpu_line
        out.update(extras)
        out["extras_rc"] = extras_rc
        print(json.dumps(out), flush=True)
    if world > 1:
        torch.distributed.barrier()
        torch.distributed.destroy_process_group()
    bad = res["overflows"] + (sustained["overflows"] if sustained else 0)
    if bad:  # a tree ran out of nodes inside the timed games: the number above is not a number of the reference's games
        print("[bench] %d minibatches overflowed the node pool: exit 3" % bad, file=sys.stderr)
        sys.exit(3)


if __name__ == "__main__":
    main()
