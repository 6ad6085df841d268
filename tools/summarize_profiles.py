#!/usr/bin/env python3
"""Condense rocprofv3 output directories (gpurun_out/<round>/...) into small files under profiles/.

    python tools/summarize_profiles.py gpurun_out/r1 r1

writes profiles/<tag>_kernel_stats.csv (rocprofv3 --kernel-trace --stats, verbatim),
profiles/<tag>_pmc_per_kernel.csv (per-kernel averages of every collected counter) and
profiles/<tag>_traffic.json (fabric bytes per launch of each kernel: (2 * FETCH_SIZE + WRITE_SIZE) * 1024 -- on gfx950 FETCH_SIZE
tallies a 128-byte line at 64 bytes (MI355X_MICROARCH.md, HBM section); profiles/r3_tcc_xcd_ab.csv shows that every fabric read of
these kernels is such a line: TCC_EA0_RDREQ_32B = 0, FETCH_SIZE * 1024 = TCC_EA0_RDREQ * 64 = TCC_MISS * 64).
"""
import collections
import csv
import glob
import json
import os
import shutil
import sys

src, tag = sys.argv[1], sys.argv[2]
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
out = os.path.join(root, "profiles")
os.makedirs(out, exist_ok=True)


def short(name):
    name = name.replace("void ", "").replace("hn::(anonymous namespace)::", "").replace("hn::", "")
    return name.split("(")[0]


def newest(paths):
    """gpurun MERGES a run's files into the local directory: a directory that was used twice holds both runs' files."""
    return max(paths, key=os.path.getmtime) if paths else None


ks = newest(glob.glob(os.path.join(src, "kt", "*", "*_kernel_stats.csv")))
if ks:
    shutil.copy(ks, os.path.join(out, f"{tag}_kernel_stats.csv"))
kt = newest(glob.glob(os.path.join(src, "kt_train", "*", "*_kernel_stats.csv")))
if kt:
    shutil.copy(kt, os.path.join(out, f"{tag}_train_kernel_stats.csv"))
agg = collections.defaultdict(lambda: collections.defaultdict(float))
launches = collections.defaultdict(lambda: collections.defaultdict(set))
dur = collections.defaultdict(lambda: collections.defaultdict(float))
by_run = collections.defaultdict(list)
for path in glob.glob(os.path.join(src, "*", "*", "*_counter_collection.csv")):
    by_run[path.split(os.sep)[-3]].append(path)
for run, paths in by_run.items():
    path = newest(paths)   # one file per counter pass: the latest run's
    for r in csv.DictReader(open(path)):
        k = short(r["Kernel_Name"])
        agg[k][r["Counter_Name"]] += float(r["Counter_Value"])
        if r["Dispatch_Id"] not in launches[k][r["Counter_Name"]]:
            launches[k][r["Counter_Name"]].add(r["Dispatch_Id"])
            dur[k][run] += int(r["End_Timestamp"]) - int(r["Start_Timestamp"])
counters = sorted({c for k in agg for c in agg[k]})
with open(os.path.join(out, f"{tag}_pmc_per_kernel.csv"), "w", newline="") as f:
    w = csv.writer(f)
    w.writerow(["kernel", "launches"] + counters)
    for k in sorted(agg, key=lambda k: -sum(dur[k].values())):
        if k.startswith("at::") or k.startswith("__amd"):
            continue
        n = max(len(v) for v in launches[k].values())
        w.writerow([k, n] + [round(agg[k][c] / max(1, len(launches[k][c])), 1) if c in agg[k] else "" for c in counters])
traffic = {}
for k in agg:
    if "FETCH_SIZE" in agg[k] and "WRITE_SIZE" in agg[k]:
        fb = agg[k]["FETCH_SIZE"] / len(launches[k]["FETCH_SIZE"]) * 1024 * 2    # 128-byte lines tallied at 64 bytes (see the docstring)
        wb = agg[k]["WRITE_SIZE"] / len(launches[k]["WRITE_SIZE"]) * 1024
        traffic[k] = {"fetch_bytes": round(fb), "write_bytes": round(wb), "hbm_bytes_per_launch": round(fb + wb),
                      "fetch_counter_rule": "2 x FETCH_SIZE x 1024"}
json.dump(traffic, open(os.path.join(out, f"{tag}_traffic.json"), "w"), indent=1, sort_keys=True)
print("wrote", sorted(os.listdir(out)))
