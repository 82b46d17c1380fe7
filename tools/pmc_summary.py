#!/usr/bin/env python3
"""Summarise rocprofv3 --pmc passes into profiles/pmc_rNN.json.

    python tools/pmc_summary.py gpurun_out/pmc_fetch gpurun_out/pmc_write profiles/pmc_r01.json

Units and corrections follow /opt/skills/guides/MI355X_MICROARCH.md (HBM section): FETCH_SIZE / WRITE_SIZE
are in KiB; on gfx950 FETCH_SIZE reports half of the bytes of a wide coalesced read, so the read side is
doubled ("double it before comparing with a byte count"); other access widths are uncalibrated, so both
the raw and the corrected figure are kept.  hbm_bytes_per_launch = (2*FETCH_SIZE + WRITE_SIZE) * 1024.
"""
import csv
import glob
import json
import os
import sys
from collections import defaultdict

# longest names first: the first key found in the (mangled) kernel name wins
SHORT = {"k_tree": "k_tree", "k_net_forward_w": "k_net_forward_w", "k_select": "k_select",
         "k_net_forward": "k_net_forward", "k_expand_backup": "k_expand_backup",
         "k_encode": "k_encode", "k_scan": "k_scan", "k_step": "k_step", "k_drain_copy": "k_drain_copy"}


def collect(dirname, counter):
    acc = defaultdict(list)
    for f in glob.glob(os.path.join(dirname, "**", "*counter_collection.csv"), recursive=True):
        for r in csv.DictReader(open(f)):
            if r["Counter_Name"] != counter:
                continue
            for key, short in SHORT.items():
                if key in r["Kernel_Name"]:
                    acc[short].append(float(r["Counter_Value"]))
                    break
    return acc


def main():
    fetch_dir, write_dir, out = sys.argv[1:4]
    fe, wr = collect(fetch_dir, "FETCH_SIZE"), collect(write_dir, "WRITE_SIZE")
    kernels = {}
    for k in sorted(set(fe) | set(wr)):
        f = sum(fe[k]) / max(1, len(fe[k]))
        w = sum(wr[k]) / max(1, len(wr[k]))
        kernels[k] = {"launches_fetch_pass": len(fe[k]), "launches_write_pass": len(wr[k]),
                      "FETCH_SIZE_KiB_per_launch": f, "WRITE_SIZE_KiB_per_launch": w,
                      "hbm_bytes_per_launch_raw": (f + w) * 1024.0,
                      "hbm_bytes_per_launch": (2.0 * f + w) * 1024.0}
    json.dump({"source": "rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE (separate passes) -- python3 bench.py "
                         "--steps 3 --warmup 2 --no-cpu-baseline --no-profile",
               "correction": "read side x2 (gfx950 FETCH_SIZE reports 1/2 of wide coalesced reads); KiB units",
               "kernels": kernels}, open(out, "w"), indent=1)
    for k, v in kernels.items():
        print("%-18s fetch %10.1f KiB  write %10.1f KiB  -> %12.0f B/launch" %
              (k, v["FETCH_SIZE_KiB_per_launch"], v["WRITE_SIZE_KiB_per_launch"], v["hbm_bytes_per_launch"]))


if __name__ == "__main__":
    main()
