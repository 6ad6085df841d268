#!/usr/bin/env python3
"""Per-kernel averages of the L2 / fabric counters collected by tools/profile_r3_counters.sh for the XCD-aware and the
-DHN_NO_XCD build, side by side (writes CSV to stdout)."""
import collections, csv, glob, os, sys

src = sys.argv[1]


def short(name):
    name = name.replace("void ", "").replace("hn::(anonymous namespace)::", "").replace("hn::", "")
    return name.split("(")[0]


agg = collections.defaultdict(lambda: collections.defaultdict(float))
cnt = collections.defaultdict(lambda: collections.defaultdict(set))
for path in glob.glob(os.path.join(src, "*", "*", "*_counter_collection.csv")):
    variant = path.split(os.sep)[-3].split("_")[0]
    for r in csv.DictReader(open(path)):
        k = short(r["Kernel_Name"])
        if k.startswith("at::") or k.startswith("__amd"):
            continue
        key = (k, r.get("Grid_Size", ""))
        agg[(variant, key)][r["Counter_Name"]] += float(r["Counter_Value"])
        cnt[(variant, key)][r["Counter_Name"]].add(r["Dispatch_Id"])
counters = sorted({c for v in agg.values() for c in v})
w = csv.writer(sys.stdout)
w.writerow(["kernel", "grid", "build"] + counters + ["l2_hit_rate", "fetch_MB(x1024)", "rdreq_x64B_MB", "rdreq_32B_share"])
keys = sorted({k for (_, k) in agg}, key=lambda k: -agg[("xcd", k)].get("FETCH_SIZE", 0))
for key in keys:
    for variant in ("xcd", "noxcd"):
        a = agg.get((variant, key))
        if not a:
            continue
        avg = {c: a[c] / max(1, len(cnt[(variant, key)][c])) for c in a}
        hit = avg.get("TCC_HIT_sum", 0.0)
        miss = avg.get("TCC_MISS_sum", 0.0)
        rd, rd32 = avg.get("TCC_EA0_RDREQ_sum", 0.0), avg.get("TCC_EA0_RDREQ_32B_sum", 0.0)
        w.writerow([key[0], key[1], variant] + [round(avg.get(c, 0.0), 1) for c in counters] +
                   [round(hit / (hit + miss), 4) if hit + miss else "", round(avg.get("FETCH_SIZE", 0.0) * 1024 / 1e6, 2),
                    round(rd * 64 / 1e6, 2), round(rd32 / rd, 4) if rd else ""])
