"""Per-dispatch averages of the SQ counters scripts/dev/pmc_script.sh collected: python scripts/dev/pmc_read.py <tag> <kernel substring>"""
import collections, glob, sqlite3, sys
tag, sub = sys.argv[1], sys.argv[2]
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for f in sorted(glob.glob("gpurun_out/pmc_%s/p*/run_results.db" % tag)):
    c = sqlite3.connect(f)
    for kn, cn, v in c.execute("select kernel_name,counter_name,value from counters_collection"):
        if sub in kn:
            acc[kn.replace("(anonymous namespace)::", "").replace("void ", "").split("(")[0][-40:]][cn].append(v)
for k, v in acc.items():
    print(k)
    for a, b in sorted(v.items()):
        print("   %-24s %14.0f  (%d dispatches)" % (a, sum(b) / len(b), len(b)))
