# timeline of one steady-state iteration (two of them), pair-kernel era: gpurun -- bash tools/trace_iter5.sh [bench args]
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/${TM:-tm}; rm -rf $O; mkdir -p $O
rocprofv3 --kernel-trace --output-format csv -d $O -- python3 $R/bench.py --steps 40 --warmup 12 --no-cpu-baseline --no-secondary "$@" > $O/log.txt 2>&1
python3 - <<PY | tee $O/timeline.txt
import csv, glob
rows = []
for f in glob.glob("$O/*/*kernel_trace.csv"):
    rows += list(csv.DictReader(open(f)))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
def short(n): return n.replace("void ", "").replace("hn::(anonymous namespace)::", "").replace("hn::", "").split("(")[0][:30]
idx = [i for i, r in enumerate(rows) if "k_dc_asm_pair" in r["Kernel_Name"]]
for k in (-6, -4):
    a, b = idx[k], idx[k + 1]
    t0 = int(rows[a]["Start_Timestamp"])
    for r in rows[a:b + 1]:
        s, e = int(r["Start_Timestamp"]) - t0, int(r["End_Timestamp"]) - t0
        q = r.get("Queue_Id", "?")
        print(f"{s / 1e3:8.1f} {e / 1e3:8.1f} {(e - s) / 1e3:6.1f} q{q} " + "    " * (int(q) % 6) + short(r["Kernel_Name"]) + f"  grid {r.get('Grid_Size_X', '?')}")
    print()
PY
