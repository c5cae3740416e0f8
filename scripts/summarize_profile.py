"""Turns the rocprofv3 CSVs of scripts/profile_round.sh into a short text summary (per-kernel mean
duration; per-launch HBM bytes from the PMC passes, corrected as MI355X_MICROARCH.md prescribes:
the counters are in KiB; bytes_read = FETCH_SIZE * 1024 * 2 (gfx950 reports half of a wide coalesced
stream -- k_mask, which reads the 54.8 MB input exactly once, calibrates this: raw 26.8 k -> 54.9 MB),
bytes_written = WRITE_SIZE * 1024)."""
import csv, glob, os, sys, collections

root = sys.argv[1]
def rows(pattern):
    for f in glob.glob(os.path.join(root, pattern), recursive=True):
        with open(f) as fh:
            for r in csv.DictReader(fh):
                yield r
def short(n):
    n = n.replace("(anonymous namespace)::", "").replace("void ", "")
    return n.split("(")[0]
print("== kernel trace (mean duration per launch, us) ==")
st = list(rows("trace/**/*kernel_stats.csv"))
for r in st:
    if "k_" in r["Name"]:
        print("%-12s calls %4s  mean %9.2f us  min %9.2f  max %9.2f" % (short(r["Name"]), r["Calls"], float(r["AverageNs"]) / 1e3, float(r["MinNs"]) / 1e3, float(r["MaxNs"]) / 1e3))
for name, pat, scale, note in (("FETCH_SIZE", "pmc_fetch/**/*counter_collection.csv", 2.0, "x1024 B x2 (gfx950 half-count correction)"),
                               ("WRITE_SIZE", "pmc_write/**/*counter_collection.csv", 1.0, "x1024 B")):
    acc = collections.defaultdict(list)
    for r in rows(pat):
        if r.get("Counter_Name") == name:
            acc[short(r["Kernel_Name"])].append(float(r["Counter_Value"]))
    print("== %s per launch (%s) ==" % (name, note))
    for k, v in acc.items():
        if k.startswith("k_"):
            m = sum(v) / len(v)
            print("%-12s launches %4d  raw %12.1f  -> %8.2f MB" % (k, len(v), m, m * 1024 * scale / 1e6))
