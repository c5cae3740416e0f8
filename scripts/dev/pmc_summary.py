"""Per-kernel SQ counter summary of a scripts/pmc_general.sh run (newest CSV of every group only)."""
import csv, glob, collections, re, os, sys
root = sys.argv[1] if len(sys.argv) > 1 else "gpurun_out/pmc_general"
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for g in sorted(glob.glob(root + "/g*/")):
    fs = sorted(glob.glob(g + "**/*counter_collection.csv", recursive=True), key=os.path.getmtime)
    if not fs:
        continue
    for r in csv.DictReader(open(fs[-1])):
        m = re.search(r"(k_[a-zA-Z0-9_]+(<[^>]*>)?)", r["Kernel_Name"])
        if m:
            acc[m.group(1)][r["Counter_Name"]].append(float(r["Counter_Value"]))
for k in sorted(acc):
    c = {n: sum(v) / len(v) for n, v in acc[k].items()}
    w = max(c.get("SQ_WAVES", 1), 1)
    g = lambda n: c.get(n, 0.0)
    print("%-12s waves %6d VALU/w %6.0f SALU/w %5.0f VMEMRD/w %5.1f VMEMWR/w %5.1f LDS/w %6.1f wavecyc/w %7.0f wait_any %2.0f%% GUI/8 %7.0f cyc  VALU-issue %7.0f cyc" % (
        k, w, g("SQ_INSTS_VALU") / w, g("SQ_INSTS_SALU") / w, g("SQ_INSTS_VMEM_RD") / w, g("SQ_INSTS_VMEM_WR") / w,
        g("SQ_INSTS_LDS") / w, g("SQ_WAVE_CYCLES") / w, 100 * g("SQ_WAIT_ANY") / max(g("SQ_WAVE_CYCLES"), 1),
        g("GRBM_GUI_ACTIVE") / 8, g("SQ_INSTS_VALU") * 4 / 1024))
