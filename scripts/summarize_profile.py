"""Turns the rocprofv3 CSVs of scripts/profile_round.sh into a short text summary and <dir>/traffic.json
(per-kernel mean duration; per-launch HBM bytes from the PMC passes, corrected as MI355X_MICROARCH.md prescribes:
the counters are in KiB; bytes_read = FETCH_SIZE * 1024 * 2 (gfx950 reports half of a wide coalesced stream --
k_mask4, which reads the input exactly once, calibrates this), bytes_written = WRITE_SIZE * 1024)."""
import collections
import csv
import glob
import json
import os
import re
import sys

root, wl = sys.argv[1], (sys.argv[2] if len(sys.argv) > 2 else "")


def rows(pattern):
    for f in glob.glob(os.path.join(root, pattern), recursive=True):
        with open(f) as fh:
            for r in csv.DictReader(fh):
                yield r


def short(n):
    m = re.search(r"(k_l2win<\d+>|k_gmc7<\w+>|k_[a-zA-Z0-9_]+)", n)  # k_l2win<10> keeps its radius, k_gmc7<false> its step kind, k_rows<8, 256> is k_rows
    return m.group(1) if m else None


print("== %s: rocprofv3 --kernel-trace --stats (mean duration per launch, us) ==" % wl)
dur = {}
for r in rows("trace/**/*kernel_stats.csv"):
    k = short(r["Name"])
    if k:
        dur[k] = float(r["AverageNs"]) / 1e3
        print("%-12s calls %4s  mean %9.2f us  min %9.2f  max %9.2f" % (k, r["Calls"], dur[k], float(r["MinNs"]) / 1e3, float(r["MaxNs"]) / 1e3))
traffic = {"kernels": {}}
for name, pat, scale, note, key in (("FETCH_SIZE", "pmc_fetch/**/*counter_collection.csv", 2.0, "x1024 B x2 (gfx950 half-count correction)", "fetch_bytes"),
                                    ("WRITE_SIZE", "pmc_write/**/*counter_collection.csv", 1.0, "x1024 B", "write_bytes")):
    acc = collections.defaultdict(list)
    for r in rows(pat):
        if r.get("Counter_Name") == name:
            k = short(r["Kernel_Name"])
            if k:
                acc[k].append(float(r["Counter_Value"]))
    print("== %s per launch (%s) ==" % (name, note))
    for k, v in sorted(acc.items()):
        m = sum(v) / len(v)
        print("%-12s launches %4d  raw %12.1f  -> %8.2f MB" % (k, len(v), m, m * 1024 * scale / 1e6))
        traffic["kernels"].setdefault(k, {})[key] = round(m * 1024 * scale)
# (k_stats is bench.py's one-off look at the routing state after the timed loops -- dtfill_pass_stats -- not a kernel of the pass)
tot = sum(v.get("fetch_bytes", 0) + v.get("write_bytes", 0) for k, v in traffic["kernels"].items() if k != "k_stats")
print("== pass: %.1f us of kernels, %.1f MB of HBM traffic ==" % (sum(v for k, v in dur.items() if k != "k_stats"), tot / 1e6))
traffic["pass_bytes"] = tot
traffic["kernel_us"] = {k: round(v, 2) for k, v in dur.items()}
json.dump(traffic, open(os.path.join(root, "traffic.json"), "w"), indent=1)
