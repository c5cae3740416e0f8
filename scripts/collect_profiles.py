"""Copies what scripts/profile_round.sh left under gpurun_out/prof_<tag>/ into profiles/<tag>/ (summaries + kernel stats) and
rebuilds profiles/traffic.json (per workload, and <workload>_l2 for the Euclidean mode: per-kernel HBM bytes per launch, pass
total, the batch they were measured on) -- bench.py reads roofline.traffic from there."""
import glob
import importlib
import json
import os
import shutil
import sys

tag = sys.argv[1] if len(sys.argv) > 1 else "r02"
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, root)
synth = importlib.import_module("distancetransform-depthcompletion_amd.synth")
src, dst = os.path.join(root, "gpurun_out", "prof_" + tag), os.path.join(root, "profiles", tag)
os.makedirs(dst, exist_ok=True)
traffic = {"source": "profiles/%s/<workload>_summary.txt: rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE in separate passes "
                     "(scripts/profile_round.sh; METRIC=l2 for the <workload>_l2 entries); bytes per launch; FETCH_SIZE x1024 x2 "
                     "per MI355X_MICROARCH.md (calibrated on k_mask4: it reads the input exactly once), WRITE_SIZE x1024" % tag}
for d in sorted(glob.glob(os.path.join(src, "*", ""))):
    name = os.path.basename(d.rstrip("/"))
    if not os.path.exists(os.path.join(d, "traffic.json")) or name.startswith("side_"):  # (the neighbours' kernels: below)
        continue
    shutil.copy(os.path.join(d, "summary.txt"), os.path.join(dst, name + "_summary.txt"))
    shutil.copy(os.path.join(d, "kernel_stats.csv"), os.path.join(dst, name + "_kernel_stats.csv"))
    t = json.load(open(os.path.join(d, "traffic.json")))
    t["batch"] = synth.CONFIGS[name[:-3] if name.endswith("_l2") else name]["B"]
    traffic[name] = t
json.dump(traffic, open(os.path.join(root, "profiles", "traffic.json"), "w"), indent=1)
for f in ("bench_default.json", "bench_l2.json", "bench_outlier.txt", "bench_gmc.txt"):
    p = os.path.join(root, "gpurun_out", f)
    if os.path.exists(p):
        shutil.copy(p, os.path.join(dst, f))
for f in glob.glob(os.path.join(src, "side_*_kernel_stats.csv")):
    shutil.copy(f, os.path.join(dst, os.path.basename(f)))
for d in glob.glob(os.path.join(src, "side_*", "")):  # scripts/profile_side.sh: trace + HBM counters of the neighbours' kernels
    if os.path.exists(os.path.join(d, "summary.txt")):
        shutil.copy(os.path.join(d, "summary.txt"), os.path.join(dst, os.path.basename(d.rstrip("/")) + "_summary.txt"))
print("profiles/%s:" % tag, sorted(os.listdir(dst)))
