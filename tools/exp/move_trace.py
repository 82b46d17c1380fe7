"""What one move of the headline bench puts on the stream (from a rocprofv3 kernel trace): kernels between two
consecutive k_stag_clean launches.  python tools/exp/move_trace.py <kernel_trace.csv> [which]"""
import csv, sys, collections
rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
which = int(sys.argv[2]) if len(sys.argv) > 2 else -3
idx = [i for i, r in enumerate(rows) if "k_stag_clean" in r["Kernel_Name"]]
seq = rows[idx[which]:idx[which + 1]]
cnt = collections.Counter(); dur = collections.Counter()
for r in seq:
    n = r["Kernel_Name"].split("<")[0].split("(")[0][-44:]
    cnt[n] += 1; dur[n] += (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1000
tot = (int(seq[-1]["End_Timestamp"]) - int(seq[0]["Start_Timestamp"])) / 1000
print("one move: span %.1f us, %d kernels, kernel time %.1f us" % (tot, len(seq), sum(dur.values())))
for n in cnt:
    print("  %-46s x%3d  %8.1f us" % (n, cnt[n], dur[n]))
