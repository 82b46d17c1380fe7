#!/usr/bin/env python3
"""Headline benchmark: self-play MCTS node-expansions/sec/GPU, Connect4,
200 sims/move (25 x 8), 1024 concurrent games per GPU (BASELINE.json configs[1]).

    python bench.py [--gpus N] [--steps K] [--warmup W]
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...

A "step" is one move of every one of the G concurrent games on a rank: 25
search minibatches (select -> net forward -> expand+backup), the ply itself,
and the drain / recycling of finished games (plus, for N > 1, the all-gather of
the drained (s, pi, z) tuples).  value = node-expansions (reference:
`_create_node` calls, lib/mcts.py:178-190 = train.py's "leaves") summed over
ranks / max-over-ranks wall time of the K timed steps.

One JSON line on stdout (rank 0).  `roofline` is the select kernel (HIP events
on its launch stream, inside the timed region); `cpu_baseline` is the oracle
(CPU port of the reference algorithm) driving the same net on one host core for
a bounded sample, rank 0, N = 1 only.
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
MFMA_BF16_PEAK_TFS = 2500.0 # MI355X_MICROARCH.md: ~2.5 PFLOP/s dense bf16


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


def load_pmc():
    """HBM bytes per launch from the committed PMC passes (profiles/pmc_r01.json), keyed by kernel."""
    path = os.path.join(ROOT, "profiles", "pmc_r01.json")
    try:
        return {k: v["hbm_bytes_per_launch"] for k, v in json.load(open(path))["kernels"].items()}
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
    return {"value": value, "unit": "node-expansions/s", "cores": procs, "kind": "port",
            "per_core": value / procs,
            "sample": "%d whole games, %d sims, %.1f s wall on %d single-threaded processes, oracle/caro_oracle.c + "
                      "torch CPU fp32 forward (eval-mode BN, %.2f rows/net call)"
                      % (games, sims, max(r[5] for r in res), procs, rows / max(1, calls))}


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
    ap.add_argument("--weights", default=os.path.join(ROOT, "tests", "golden", "weights", "best_026_12000.dat"))
    ap.add_argument("--net", default="hipw", choices=["hip", "hipw", "hip3x", "gemm", "folded", "net"],
                    help="inference form of lib/model.py Net: hipw = fused HIP fp32 MFMA kernel, 3x3 convs in row-Winograd "
                         "F(2,3) form (default); hip = the same with direct 3x3 convs; hip3x = direct convs on the bf16 "
                         "MFMA pipe via three-way split operands (opt-in); torch "
                         "gather+GEMM; BN-folded conv2d; or the module as is")
    ap.add_argument("--streams", type=int, default=1,
                    help="split the games of a GPU over this many engines on separate HIP streams (tree kernels of "
                         "one part overlap the net kernel of another)")
    ap.add_argument("--stream-mask", type=int, default=1, help="with --streams > 1: confine each part to its own CU slice")
    ap.add_argument("--gather-every", type=int, default=8,
                    help="N > 1: all-gather the finished games' tuples every this many moves (one payload message)")
    ap.add_argument("--cpu-seconds", type=float, default=12.0)
    ap.add_argument("--cpu-procs", type=int, default=0, help="host cores for the CPU baseline (0 = all this process may use: affinity and cgroup quota, capped at 32)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-profile", action="store_true", help="do not record HIP events around the kernels")
    args = ap.parse_args()

    from caro_ai_amd import parallel
    from caro_ai_amd.engine import SelfPlayEngine, StreamedSelfPlay, torch_evaluator
    from caro_ai_amd.lib.game.connect_four import ConnectFour
    from caro_ai_amd.lib.game.tictactoe import TicTacToe
    from caro_ai_amd.lib.model import FoldedNet, GemmNet

    # CPU baseline first (N = 1 only): it forks worker processes, which must happen before the GPU is touched
    cpu_line = None
    if int(os.environ.get("WORLD_SIZE", "1")) == 1 and not args.no_cpu_baseline and not args.arena:
        procs = args.cpu_procs or min(32, host_cores())
        w = args.weights if args.game == "c4" else None
        print("[bench] cpu baseline on %d cores for %.0f s" % (procs, args.cpu_seconds), file=sys.stderr, flush=True)
        cpu_line = cpu_baseline(args.game, args.searches, args.batch, 10, w, args.cpu_seconds, procs)

    rank, local_rank, world = parallel.init()
    assert world == args.gpus, "launch with torch.distributed.run --nproc-per-node %d" % args.gpus
    # one rank per GPU; CARO_SHARE_GPU=1 maps every rank to cuda:0 (rehearsal of the N > 1 path on a 1-GPU box)
    device = torch.device("cuda", 0 if os.environ.get("CARO_SHARE_GPU") else local_rank)
    torch.cuda.set_device(device)

    if args.game == "c4":
        game, weights, sbt0 = ConnectFour(), args.weights, 10
    else:
        game, weights, sbt0 = TicTacToe(15, 5), None, 10
    net, wtag = load_net(game, device, weights)
    G, S, B = args.games, args.searches, args.batch
    extra = {}
    if args.node_cap:
        extra["node_cap"] = args.node_cap
    evict = args.evict if args.evict >= 0 else int(args.game == "gomoku15")
    if evict:
        extra["evict"] = True
        extra.setdefault("node_cap", 4096)
    if args.arena:
        assert args.game == "c4" and args.net in ("hip", "hipw", "hip3x")
        sbt0 = 0
        net2, wtag2 = load_net(game, device, os.path.join(os.path.dirname(weights), "best_025_10600.dat"))
        wtag = wtag + " vs " + wtag2
        extra.update(n_stores=2, first_player_mode=2)
    is_hip = args.net in ("hip", "hipw", "hip3x")
    if is_hip:
        from caro_ai_amd.net_hip import HipNet
        mode = {"hip": "f32", "hipw": "f32w", "hip3x": "3xbf16"}[args.net]
        hipnet = HipNet(net, str(device), mode=mode)
        hipnets = [hipnet] + ([HipNet(net2, str(device), mode=mode)] if args.arena else [])
        make_evaluators = lambda: list(hipnets)
    else:
        fnet = {"gemm": GemmNet, "folded": FoldedNet, "net": lambda n: n}[args.net](net).to(device).eval()
        make_evaluators = lambda: [torch_evaluator(fnet, form="net")]
    n_streams = args.streams if is_hip else 1
    if n_streams > 1:
        eng = StreamedSelfPlay(game, G, make_evaluators, n_streams=n_streams, partition_cus=bool(args.stream_mask),
                               max_batch=B, steps_before_tau_0=sbt0,
                               seed=0, device=str(device), searches_hint=S, **extra, **parallel.shard(G, rank, world))
    else:
        eng = SelfPlayEngine(game, G, evaluators=make_evaluators(), max_batch=B, steps_before_tau_0=sbt0,
                             seed=0, device=str(device), searches_hint=S, **extra, **parallel.shard(G, rank, world))

    n_tuples = 0
    # N > 1: tuples wait on the device and are all-gathered every 8th move (and at the end of the timed region)
    gatherer = parallel.TupleGatherer(every=args.gather_every)

    def count(d):
        nonlocal n_tuples
        if d is not None:
            n_tuples += int(d["z"].shape[0])

    def one_step():
        if n_streams > 1:
            d = eng.move(S, B)      # host-pipelined over the parts: drains the previous move of each part
        else:
            eng.search(S, B)
            eng.step()
            d = eng.drain(recycle=True)
        count(gatherer.push(d))

    def barrier():
        if n_streams > 1:
            count(gatherer.push(eng.flush()))  # the last enqueued move of every part belongs to the timed region
        count(gatherer.flush())
        torch.cuda.synchronize(device)
        if world > 1:
            torch.distributed.barrier()
        torch.cuda.synchronize(device)

    for i in range(args.warmup):
        one_step()
        if rank == 0 and i % 5 == 0:
            print("[bench] warmup step %d" % i, file=sys.stderr, flush=True)
    barrier()
    c0 = eng.counters()
    if not args.no_profile:
        eng.profile(True)
        eng.profile_read(reset=True)
    n_tuples = 0
    t0 = time.perf_counter()
    for i in range(args.steps):
        one_step()
        if rank == 0 and i % 10 == 0:
            print("[bench] step %d" % i, file=sys.stderr, flush=True)
    barrier()
    dt = time.perf_counter() - t0
    c1 = eng.counters()
    prof = eng.profile_read(reset=True) if not args.no_profile else None
    eng.profile(False)

    delta = {k: c1[k] - c0[k] for k in c1}
    tot = torch.tensor([delta["expansions"], delta["sims"], delta["levels"], delta["plies"], delta["finished"],
                        delta["expansions"]], dtype=torch.float64, device=device)
    tmax = torch.tensor([dt], dtype=torch.float64, device=device)
    parallel.allreduce_sum(tot)
    parallel.allreduce_max(tmax)
    dt_max = float(tmax.item())
    exp_all, sims_all, levels_all, plies_all, fin_all, rows_all = [float(x) for x in tot.tolist()]

    if rank == 0:
        pmc = load_pmc() if args.game == "c4" and G == 1024 else {}
        A, KW, HW = game.action_space, game.key_words, game.obs_shape[1] * game.obs_shape[2]
        bytes_per_level = 12 * A + 8 * KW + 28            # N,Q,P rows + key probe + backup RMW
        bytes_per_exp = 16 * HW + 20 * A + 8 * KW + 12
        # FLOPs of one leaf through lib/model.py Net (2 x MAC): conv_in, 5 residual 3x3 convs, 1x1 heads, FC heads
        flops_per_leaf = 2.0 * (HW * 64 * 18 + 5 * HW * 64 * 576 + HW * 3 * 64 + 20 * HW + 20 + 2 * HW * A)
        roofline = roofline_tree = None
        if prof is not None and prof["select"][1] > 0:
            ms, n = prof["select"]
            avg_s = ms * 1e-3 / n               # timed on a sample of the launches (every 12th minibatch, all minibatch indices equally)
            n_launches = args.steps * S * n_streams
            levels_per_launch = delta["levels"] / n_launches
            achieved = levels_per_launch * bytes_per_level / avg_s / 1e9
            # one net + one wavefront per game: caro_search_batch runs the fused k_tree (expand+backup of the
            # previous minibatch, select, row reservation + planes); otherwise k_select / k_encode / k_expand_backup
            fused = prof.get("compact", (0, 0))[1] == 0
            tname = "k_tree" if fused else "k_select"
            roofline_tree = {"bound": "hbm", "kernel": tname, "achieved": achieved, "peak": HBM_PEAK_GBS,
                             "unit": "GB/s", "frac": achieved / HBM_PEAK_GBS, "traffic": pmc.get(tname),
                             "avg_launch_us": avg_s * 1e6, "launches": n_launches, "launches_timed": n,
                             "levels_per_launch": levels_per_launch, "bytes_per_level": bytes_per_level,
                             "other_kernels_us": {k: (v[0] * 1e3 / v[1] if v[1] else None) for k, v in prof.items()
                                                  if k not in ("select", "net")}}
        if prof is not None and prof.get("net", (0, 0))[1] > 0:
            ms, n = prof["net"]
            avg_s = ms * 1e-3 / n
            # the net is timed on a sample of the launches; one net launch per select launch
            leaves_per_launch = delta["expansions"] / (args.steps * S * n_streams)
            achieved = leaves_per_launch * flops_per_leaf / avg_s / 1e12
            # hip3x issues 6 bf16 MFMA flops per algorithmic flop: priced against the dense bf16 peak / 6
            peak = MFMA_F32_PEAK_TFS if args.net != "hip3x" else MFMA_BF16_PEAK_TFS / 6.0
            kname = {"hip": "k_net_forward", "hipw": "k_net_forward_w", "hip3x": "k_net_forward_3x"}[args.net]
            roofline = {"bound": "mfma", "kernel": kname,
                        "achieved": achieved, "peak": peak,
                        "unit": "TFLOP/s", "frac": achieved / peak, "traffic": pmc.get(kname),
                        "avg_launch_us": avg_s * 1e6, "launches_timed": n, "leaves_per_launch": leaves_per_launch,
                        "flops_per_leaf": flops_per_leaf}
            if args.net == "hipw":
                # the Winograd form issues 2/3 of the 3x3-conv multiplies: what the MFMA pipe itself executes
                # (60 taps x 32 MFMAs x 8 waves x 4096 flop per workgroup of TB boards, padding included)
                tb = hipnet.L.caro_net_boards_per_workgroup(hipnet.h)
                issued = math.ceil(leaves_per_launch / tb) * 60 * 32 * 8 * 4096.0 / avg_s / 1e12
                roofline["note"] = ("achieved = algorithmic (direct-convolution) flops per launch / launch time; "
                                    "mfma_issued = flops the MFMA pipe executes in the F(2,3) form")
                roofline["mfma_issued"] = {"achieved": issued, "frac": issued / peak}
        if roofline is None:  # torch evaluators: the net is not our kernel; the tree walk is the dominant own kernel
            roofline = roofline_tree
        out = {
            "metric": "self-play MCTS node-expansions/sec/GPU (Connect4, 200 sims/move); 1->8 GPU scaling"
            if args.game == "c4" else "self-play MCTS node-expansions/sec/GPU (15x15 k=5)",
            "value": exp_all / dt_max,
            "unit": "node-expansions/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": dt_max * 1e3 / args.steps,
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": "f32" if args.net != "hip3x" else "f32 (3x3 conv products as 3-way split bf16, f32 accumulate)",
            "data": "synthetic (self-play from empty boards; net weights: %s)" % wtag,
            "config": {"workload": "%s %d concurrent %s/GPU, %dx%d = %d sims/move, tau=1 for %d plies"
                                   % ("Connect4 6x7" if args.game == "c4" else "m,n,k 15x15 k=5", G,
                                      "arena matches (two nets, one tree per player)" if args.arena else "self-play games",
                                      S, B, S * B, sbt0),
                       "games_per_gpu": G, "searches": S, "batch": B, "net": "lib/model.py Net, %s fp32" % {"hip": "fused HIP MFMA kernel", "hipw": "fused HIP MFMA kernel, 3x3 convs in row-Winograd F(2,3) form,", "hip3x": "fused HIP kernel, 3x3 convs as 3-way split bf16 MFMA with f32 accumulate,", "gemm": "torch gather+GEMM", "folded": "torch conv2d BN-folded", "net": "torch module"}[args.net],
                       "streams_per_gpu": n_streams,
                       "parallelism": "games sharded x%d, tuples all-gathered every %d moves" % (world, args.gather_every)},
            "per_gpu": exp_all / dt_max / world,
            "sims_per_s": sims_all / dt_max, "plies_per_s": plies_all / dt_max, "games_per_s": fin_all / dt_max,
            "net_rows_per_s": rows_all / dt_max, "mean_depth": levels_all / max(1.0, sims_all),
            "expansions_per_sim": exp_all / max(1.0, sims_all),
            "algorithmic_GBps": (levels_all * bytes_per_level + exp_all * bytes_per_exp) / dt_max / 1e9,
            "overflows": delta["overflows"], "evict": bool(evict),
            "roofline": roofline,
            "roofline_tree": roofline_tree,
        }
        out["cpu_baseline"] = cpu_line
        print(json.dumps(out))
    eng.close()
    if world > 1:
        torch.distributed.barrier()
        torch.distributed.destroy_process_group()


if __name__ == "__main__":
    main()
