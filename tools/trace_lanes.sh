cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/tl; rm -rf $O; mkdir -p $O
rocprofv3 --kernel-trace --output-format csv -d $O -- python3 $R/bench.py --steps 40 --warmup 12 --no-cpu-baseline --no-secondary --lanes 2 "$@" > $O/log.txt 2>&1
python3 - <<PY
import csv, glob
rows = []
for f in glob.glob("$O/*/*kernel_trace.csv"):
    rows += list(csv.DictReader(open(f)))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
def short(n): return n.replace("void ", "").replace("hn::(anonymous namespace)::", "").split("(")[0][:26]
idx = [i for i, r in enumerate(rows) if "<2, 2, 2, 0" in r["Kernel_Name"]]
a, b = idx[-9], idx[-5]
t0 = int(rows[a]["Start_Timestamp"])
for r in rows[a:b + 1]:
    s, e = int(r["Start_Timestamp"]) - t0, int(r["End_Timestamp"]) - t0
    q = r.get("Queue_Id", "?")
    print(f"{s / 1e3:8.1f} {e / 1e3:8.1f} {(e - s) / 1e3:6.1f} q{q} " + "    " * (int(q) % 6) + short(r["Kernel_Name"]))
PY
