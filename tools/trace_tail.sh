cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/tt; rm -rf $O; mkdir -p $O
rocprofv3 --kernel-trace --output-format csv -d $O -- python3 $R/bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu-baseline --no-secondary "$@" > $O/log.txt 2>&1
python3 - <<PY
import csv, glob
rows = []
for f in glob.glob("$O/*/*kernel_trace.csv"):
    rows += list(csv.DictReader(open(f)))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
dec = [(int(r["Start_Timestamp"]), int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) for r in rows if "k_dc_valu<8, 8, 0, 1" in r["Kernel_Name"] or "k_dc_mfma_s<8, 8, 0, 1" in r["Kernel_Name"]]
print(len(dec))
last = dec[-60:]
t0 = last[0][0]
print(" ".join(f"{(s - t0) / 1e6:.1f}ms:{d / 1e3:.1f}" for s, d in last))
PY
