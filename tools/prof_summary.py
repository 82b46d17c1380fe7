#!/usr/bin/env python3
"""Summarise the rocprofv3 runs of tools/profile_r06.sh (gpurun_out/prof_r06/) into profiles/.

    python tools/prof_summary.py gpurun_out/prof_r06 r06_a pmc_r06.json

bench.py plays a 16-game copy of its configuration before the clock starts (first-use costs); its small launches
are in the traces too.  Only the launches of the full-size engine are summarised: per kernel, the dispatches with
the largest grid.  Outputs:
  profiles/<tag>_kernel_stats.csv   per-kernel calls / avg / min / max / share, from the kernel trace (--stats run)
  profiles/pmc_r02.json             HBM bytes per launch (FETCH_SIZE x2 on the read side, the gfx950 correction of
                                    MI355X_MICROARCH.md; WRITE_SIZE as is; KiB units) and the MFMA-busy fraction
"""
import csv
import glob
import json
import os
import sys
from collections import defaultdict

SHORT = ["k_tree_stag_mw", "k_tree_stag", "k_tree_mw", "k_tree", "k_net_heads", "k_net_forward_w2", "k_net_forward_w", "k_net_forward_x3", "k_net_forward", "k_select", "k_expand_backup", "k_encode",
         "k_step", "k_drain_copy", "k_drain_scan", "k_evict", "k_net_hash"]


def short(name):
    for k in SHORT:
        if k in name:
            return k
    return None


def rows(dirname, pattern):
    for f in glob.glob(os.path.join(dirname, "**", pattern), recursive=True):
        yield from csv.DictReader(open(f))


def full_size(recs, grid_key):
    """keep, per kernel, the dispatches with that kernel's largest grid"""
    big = defaultdict(int)
    for r in recs:
        big[r["k"]] = max(big[r["k"]], int(r[grid_key]))
    return [r for r in recs if int(r[grid_key]) == big[r["k"]]]


def main():
    src, tag = sys.argv[1], sys.argv[2]
    root = os.path.abspath(os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
    # ---- kernel trace of the --stats run
    def stats(sub, out_tag):
        recs = []
        for r in rows(os.path.join(src, sub), "*kernel_trace.csv"):
            k = short(r["Kernel_Name"])
            if k:
                recs.append({"k": k, "g": r["Grid_Size_X"], "ns": int(r["End_Timestamp"]) - int(r["Start_Timestamp"]),
                             "t0": int(r["Start_Timestamp"]), "t1": int(r["End_Timestamp"])})
        if not recs:
            return
        recs = full_size(recs, "g")
        per = defaultdict(list)
        for r in recs:
            per[r["k"]].append(r["ns"])
        total = sum(sum(v) for v in per.values())
        span = (max(r["t1"] for r in recs) - min(r["t0"] for r in recs)) if recs else 0
        cmd_file = os.path.join(src, sub + ".cmd")  # the command line as the profile script ran it
        cmd = open(cmd_file).read().strip() if os.path.exists(cmd_file) else "rocprofv3 --kernel-trace --stats -- python3 bench.py ..."
        out = os.path.join(root, "profiles", out_tag + "_kernel_stats.csv")
        with open(out, "w") as f:
            f.write("# %s; launches of the full-size engine only (largest grid per kernel); "
                    "busy = sum of kernel time / span first..last launch = %.3f\n" % (cmd, total / span if span else 0))
            f.write("kernel,calls,total_us,avg_us,min_us,max_us,share\n")
            for k, v in sorted(per.items(), key=lambda kv: -sum(kv[1])):
                f.write("%s,%d,%.1f,%.2f,%.2f,%.2f,%.4f\n" % (k, len(v), sum(v) / 1e3, sum(v) / len(v) / 1e3, min(v) / 1e3,
                                                              max(v) / 1e3, sum(v) / total))
        print(open(out).read())

    stats("stats", tag)
    stats("stats_config4", tag + "_config4")
    stats("stats_x3", tag + "_x3")  # the labelled extra leg (--net hipx3)
    # ---- counters
    pmc_name = sys.argv[3] if len(sys.argv) > 3 else "pmc_r02.json"

    def counters(sub, names):
        acc = []
        for r in rows(os.path.join(src, sub), "*counter_collection.csv"):
            k = short(r["Kernel_Name"])
            if k and r["Counter_Name"] in names:
                acc.append({"k": k, "g": r["Grid_Size"], "c": r["Counter_Name"], "v": float(r["Counter_Value"]),
                            "ns": int(r["End_Timestamp"]) - int(r["Start_Timestamp"])})
        return full_size(acc, "g")

    def section(prefix):
        fe, wr = counters(prefix + "fetch", {"FETCH_SIZE"}), counters(prefix + "write", {"WRITE_SIZE"})
        kernels = {}
        for k in sorted({r["k"] for r in fe} | {r["k"] for r in wr}):
            f = [r["v"] for r in fe if r["k"] == k]
            w = [r["v"] for r in wr if r["k"] == k]
            fa, wa = sum(f) / max(1, len(f)), sum(w) / max(1, len(w))
            kernels[k] = {"launches_fetch_pass": len(f), "launches_write_pass": len(w), "FETCH_SIZE_KiB_per_launch": fa,
                          "WRITE_SIZE_KiB_per_launch": wa, "hbm_bytes_per_launch_raw": (fa + wa) * 1024.0,
                          "hbm_bytes_per_launch": (2.0 * fa + wa) * 1024.0}
        mf = counters(prefix + "mfma", {"SQ_VALU_MFMA_BUSY_CYCLES", "SQ_BUSY_CU_CYCLES", "GRBM_GUI_ACTIVE"})
        mfma = {}
        for k in sorted({r["k"] for r in mf}):
            get = lambda c: [r["v"] for r in mf if r["k"] == k and r["c"] == c]
            busy, gui = get("SQ_VALU_MFMA_BUSY_CYCLES"), get("GRBM_GUI_ACTIVE")
            dur = [r["ns"] for r in mf if r["k"] == k and r["c"] == "GRBM_GUI_ACTIVE"]
            if not busy or not gui:
                continue
            b, g = sum(busy) / len(busy), sum(gui) / len(gui)
            # GRBM_GUI_ACTIVE is summed over the 8 XCDs: kernel cycles = g / 8; 1024 SIMDs each with one MFMA pipe
            mfma[k] = {"launches": len(busy), "SQ_VALU_MFMA_BUSY_CYCLES_per_launch": b,
                       "GRBM_GUI_ACTIVE_per_launch_sum_of_8_XCDs": g,
                       "avg_duration_us_under_the_profiler": sum(dur) / len(dur) / 1e3,
                       "mfma_busy_fraction": b / (g / 8.0 * 1024.0)}
        for k, v in kernels.items():
            print("%s%-18s fetch %10.1f KiB  write %10.1f KiB  -> %12.0f B/launch  (%d launches)" %
                  (prefix, k, v["FETCH_SIZE_KiB_per_launch"], v["WRITE_SIZE_KiB_per_launch"], v["hbm_bytes_per_launch"],
                   v["launches_fetch_pass"]))
        for k, v in mfma.items():
            print("%s%-18s MFMA busy %.3f (%.1f us under the profiler)" % (prefix, k, v["mfma_busy_fraction"],
                                                                           v["avg_duration_us_under_the_profiler"]))
        return {"kernels": kernels, "mfma_utilisation": mfma}

    out_json = {"source": "rocprofv3 --pmc FETCH_SIZE | --pmc WRITE_SIZE | --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CU_CYCLES "
                          "GRBM_GUI_ACTIVE (separate passes, each with --kernel-trace only) -- python3 bench.py "
                          "--no-cpu-baseline --no-extra-configs --no-profile (steps / warm-up: see the script that "
                          "made the directory, tools/profile_rNN.sh); launches of the full-size engine",
                "correction": "read side x2 (gfx950 FETCH_SIZE reports 1/2 of wide coalesced reads); KiB units",
                "mfma_note": "busy cycles of the MFMA pipe summed over the 1024 SIMDs / (kernel cycles x 1024)"}
    out_json.update(section("pmc_"))
    for extra in ("config5", "config4"):  # profile_r03.sh: the same three passes on BASELINE configs 5 and 4
        if os.path.isdir(os.path.join(src, extra + "_fetch")):
            out_json[extra] = section(extra + "_")
    if os.path.isdir(os.path.join(src, "x3_fetch")):  # the headline's configuration with --net hipx3 (the labelled extra leg)
        out_json["net_bf16x3"] = section("x3_")
    # what the passes were taken on (tools/profile_r06.sh leaves the kernel sources' hashes beside the counters), and
    # the commit that holds exactly those sources -- bench.py refuses to quote the counters once the sources differ
    sha_file = os.path.join(src, "source_sha256.json")
    if os.path.exists(sha_file):
        import subprocess
        sys.path.insert(0, os.path.join(root, "tools"))
        from source_sha import source_sha256
        out_json["source_sha256"] = json.load(open(sha_file))
        same = out_json["source_sha256"] == source_sha256(root)
        clean = subprocess.run(["git", "-C", root, "diff", "--quiet", "HEAD", "--", "caro_ai_amd/csrc", "include"]).returncode == 0
        head = subprocess.run(["git", "-C", root, "rev-parse", "--short=12", "HEAD"], capture_output=True, text=True).stdout.strip()
        out_json["pmc_source_head"] = head if (same and clean) else None
        print("PMC passes taken on the sources of commit", out_json["pmc_source_head"])
    json.dump(out_json, open(os.path.join(root, "profiles", pmc_name), "w"), indent=1)


if __name__ == "__main__":
    main()
