"""Summarise a rocprofv3 --kernel-trace of tools/bench_train.py: per-kernel totals of the LAST step, busy time per queue, overlap."""
import csv, glob, sys, collections
f = glob.glob(sys.argv[1] + "/*/*_kernel_trace.csv")[0]
rows = list(csv.DictReader(open(f)))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
def nm(r):
    return r["Kernel_Name"].replace("void ", "").replace("hn::(anonymous namespace)::", "").split("(")[0][:44]
# steps are delimited by k_adam launches
ends = [i for i, r in enumerate(rows) if nm(r).startswith("k_adam")]
lo, hi = ends[-2] + 1, ends[-1] + 1
step = rows[lo:hi]
t0, t1 = int(step[0]["Start_Timestamp"]), int(step[-1]["End_Timestamp"])
print(f"last step: {len(step)} launches, span {(t1 - t0) / 1e6:.3f} ms")
tot = collections.defaultdict(lambda: [0, 0.0])
for r in step:
    d = (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3
    tot[nm(r)][0] += 1; tot[nm(r)][1] += d
for k, (c, d) in sorted(tot.items(), key=lambda kv: -kv[1][1])[:26]:
    print(f"  {k:46s} {c:4d} launches {d / 1e3:8.3f} ms  avg {d / c:7.1f} us")
print(f"  sum of kernel durations {sum(v[1] for v in tot.values()) / 1e3:.3f} ms")
# union of busy intervals, and per queue
ivs = sorted((int(r["Start_Timestamp"]), int(r["End_Timestamp"])) for r in step)
busy, cur_s, cur_e = 0, *ivs[0]
for s, e in ivs[1:]:
    if s > cur_e: busy += cur_e - cur_s; cur_s, cur_e = s, e
    else: cur_e = max(cur_e, e)
busy += cur_e - cur_s
print(f"  union busy {busy / 1e6:.3f} ms; idle gaps {(t1 - t0 - busy) / 1e6:.3f} ms")
q = collections.defaultdict(float)
for r in step: q[r.get("Queue_Id", "?")] += (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e6
print("  per queue (ms):", dict(q))
