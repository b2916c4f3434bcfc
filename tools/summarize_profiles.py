#!/usr/bin/env python3
"""Condenses the rocprofv3 output of tools/profile_round.sh into the small files kept under
profiles/:  <prefix>_<wl>_kernel_stats.csv, <prefix>_<wl>_pmc.csv, <prefix>_bench_<wl>.json and
traffic.json (HBM bytes per launch of the dominant kernel: (2*FETCH_SIZE + WRITE_SIZE) KiB,
FETCH_SIZE doubled as gfx950 tallies 128-byte requests at 64 bytes).

    python tools/summarize_profiles.py gpurun_out/<tag> profiles r01
"""
import collections
import csv
import glob
import json
import os
import shutil
import sys

src, dst, prefix = sys.argv[1], sys.argv[2], sys.argv[3]
os.makedirs(dst, exist_ok=True)
traffic_path = os.path.join(dst, "traffic.json")
traffic = json.load(open(traffic_path)) if os.path.exists(traffic_path) else {}


def dominant(acc):
    """The frame kernel with the most dispatches (a frame that measures tile costs runs another
    instantiation of k_mega once in a while)."""
    best, best_n = None, -1
    for n, counters in acc.items():
        if "k_mega" in n:
            d = max(len(v) for v in counters.values())
            if d > best_n:
                best, best_n = n, d
    return best


def newest_per_dir(files):
    """A pass run again into the same directory leaves the earlier run's file (named by process id)
    beside the new one: only the newest file of each directory counts."""
    by_dir = {}
    for f in files:
        d = os.path.dirname(f)
        if d not in by_dir or os.path.getmtime(f) > os.path.getmtime(by_dir[d]):
            by_dir[d] = f
    return sorted(by_dir.values())


for bench in sorted(glob.glob(os.path.join(src, "bench_*.json"))):
    wl = os.path.basename(bench)[6:-5]
    shutil.copy(bench, os.path.join(dst, f"{prefix}_bench_{wl}.json"))
    stats = newest_per_dir(glob.glob(os.path.join(src, f"trace_{wl}", "**", "*kernel_stats.csv"), recursive=True))
    if stats:
        rows = list(csv.DictReader(open(stats[0])))
        with open(os.path.join(dst, f"{prefix}_{wl}_kernel_stats.csv"), "w", newline="") as f:
            w = csv.DictWriter(f, fieldnames=list(rows[0].keys()))
            w.writeheader()
            for r in rows[:12]:
                w.writerow(r)
    acc = collections.defaultdict(lambda: collections.defaultdict(list))
    for f in newest_per_dir(glob.glob(os.path.join(src, f"pmc_{wl}_*", "**", "*counter_collection.csv"), recursive=True) +
                            glob.glob(os.path.join(src, f"sq_{wl}", "**", "*counter_collection.csv"), recursive=True)):
        for r in csv.DictReader(open(f)):
            acc[r["Kernel_Name"].split("(")[0]][r["Counter_Name"]].append(float(r["Counter_Value"]))
    kern = dominant(acc)
    if kern:
        with open(os.path.join(dst, f"{prefix}_{wl}_pmc.csv"), "w", newline="") as f:
            w = csv.writer(f)
            w.writerow(["kernel", "counter", "mean_per_dispatch", "dispatches"])
            for c in sorted(acc[kern]):
                v = acc[kern][c]
                w.writerow([kern, c, sum(v) / len(v), len(v)])
        v = acc[kern]
        if "FETCH_SIZE" in v and "WRITE_SIZE" in v:
            fetch = sum(v["FETCH_SIZE"]) / len(v["FETCH_SIZE"])
            write = sum(v["WRITE_SIZE"]) / len(v["WRITE_SIZE"])
            traffic[wl] = (2 * fetch + write) * 1024
trav_path = os.path.join(dst, "traversal.json")
trav = json.load(open(trav_path)) if os.path.exists(trav_path) else {}
for f in sorted(glob.glob(os.path.join(src, "cnt_*.json"))):
    wl = os.path.basename(f)[4:-5]
    try:
        t = json.load(open(f)).get("traversal")
    except Exception:
        t = None
    if t:
        trav[wl] = {"nodes_per_ray": t["nodes_per_ray"], "prims_per_ray": t["prims_per_ray"]}
trav["_how"] = ("4-wide nodes entered and primitives tested per intersection query of the workload's frame, counted by a "
                "-DVRT_COUNT_NODES=1 build of the library (tools/build_variant.sh cnt -DVRT_COUNT_NODES=1)")
json.dump(trav, open(trav_path, "w"), indent=1)
traffic["_how"] = ("dominant render kernel of each workload: (2*FETCH_SIZE + WRITE_SIZE)*1024 bytes per dispatch "
                   "from separate rocprofv3 --pmc passes (FETCH_SIZE doubled per the gfx950 correction)")
json.dump(traffic, open(traffic_path, "w"), indent=1)
print("wrote", sorted(os.listdir(dst)))
