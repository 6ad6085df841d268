# timeline of one steady-state iteration: gpurun -- bash tools/trace_iter.sh [bench args]
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/ti; rm -rf $O; mkdir -p $O
rocprofv3 --kernel-trace --output-format csv -d $O -- python3 $R/bench.py --steps 40 --warmup 12 --no-cpu-baseline --no-secondary "$@" > $O/log.txt 2>&1
python3 - <<PY
import csv, glob
rows = []
for f in glob.glob("$O/*/*kernel_trace.csv"):
    rows += list(csv.DictReader(open(f)))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
def short(n): return n.replace("void ", "").replace("hn::(anonymous namespace)::", "").split("(")[0][:34]
# find the last-but-3 launch of the inc kernel (k_dc_valu<2, 2, 2 or k_dc_mfma_s<2, 2, 2)
idx = [i for i, r in enumerate(rows) if "<2, 2, 2, 0" in r["Kernel_Name"]]
a, b = idx[-4], idx[-3]
t0 = int(rows[a]["Start_Timestamp"])
for r in rows[a:b + 1]:
    s, e = int(r["Start_Timestamp"]) - t0, int(r["End_Timestamp"]) - t0
    print(f"{s / 1e3:8.1f} {e / 1e3:8.1f} {(e - s) / 1e3:7.1f}  q={r.get('Queue_Id', '?'):>3} {short(r['Kernel_Name'])}")
PY
