"""Summarise a rocprofv3 --kernel-trace of tools/bench_trainer_loop.py: in a steady-state Trainer.training_step (between two k_adam launches of the
first loop), the kernels that are NOT the library's -- the replay buffer's gathers / scatters, the refill mask, the fresh maps -- and the idle gaps."""
import collections, csv, glob, sys
fs = glob.glob(sys.argv[1] + "/*/*_kernel_trace.csv")
if not fs:
    sys.exit("no kernel trace found under " + sys.argv[1])
rows = list(csv.DictReader(open(fs[0])))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
def nm(r):
    return r["Kernel_Name"].replace("void ", "").replace("hn::(anonymous namespace)::", "hn:").split("(")[0][:70]
ends = [i for i, r in enumerate(rows) if "k_adam" in r["Kernel_Name"]]
# the tool runs 5 warm-up + 12 timed training_steps, then 3 + 12 bare library calls: take the 10th step of the first loop
lo, hi = ends[9] + 1, ends[10] + 1
step = rows[lo:hi]
t0, t1 = int(rows[ends[9]]["End_Timestamp"]), int(step[-1]["End_Timestamp"])
print(f"one training_step: {len(step)} launches, adam-to-adam {(t1 - t0) / 1e6:.3f} ms")
tot = collections.defaultdict(lambda: [0, 0.0])
for r in step:
    d = (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3
    k = nm(r)
    tot["library (hn::)" if k.startswith("hn:") else k][0] += 1
    tot["library (hn::)" if k.startswith("hn:") else k][1] += d
for k, (c, d) in sorted(tot.items(), key=lambda kv: -kv[1][1])[:24]:
    print(f"  {k:72s} {c:4d} launches {d:9.1f} us")
other = sum(v[1] for k, v in tot.items() if not k.startswith("library"))
print(f"  non-library kernels: {sum(v[0] for k, v in tot.items() if not k.startswith('library'))} launches, {other:.1f} us")
ivs = sorted((int(r["Start_Timestamp"]), int(r["End_Timestamp"])) for r in step)
busy, cs, ce = 0, t0, t0
gaps = []
for s, e in ivs:
    if s > ce:
        busy += ce - cs; gaps.append((s - ce, s)); cs, ce = s, e
    else:
        ce = max(ce, e)
busy += ce - cs
print(f"  union busy {busy / 1e6:.3f} ms; idle {(t1 - t0 - busy) / 1e6:.3f} ms; largest gaps (us): {[round(g / 1e3, 1) for g, _ in sorted(gaps, reverse=True)[:8]]}")
# the same for a bare library step (train_grad + adam) of the second loop
lo2, hi2 = ends[-3] + 1, ends[-2] + 1
s2 = rows[lo2:hi2]
print(f"bare library step: {len(s2)} launches, adam-to-adam {(int(s2[-1]['End_Timestamp']) - int(rows[ends[-3]]['End_Timestamp'])) / 1e6:.3f} ms")
