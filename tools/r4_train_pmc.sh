cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r4tp
rm -rf $O; mkdir -p $O
rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT --output-format csv -d $O/pmc -- python3 $R/tools/bench_train.py --steps 3 > /dev/null 2>&1
python3 - <<PY
import csv, glob, collections
acc = collections.defaultdict(lambda: collections.defaultdict(float)); cnt = collections.defaultdict(set); dur = collections.defaultdict(float)
for f in glob.glob("$O/pmc/*/*counter_collection.csv"):
    for r in csv.DictReader(open(f)):
        n = r["Kernel_Name"].replace("void ", "").replace("hn::(anonymous namespace)::", "").split("(")[0]
        acc[n][r["Counter_Name"]] += float(r["Counter_Value"])
        if r["Dispatch_Id"] not in cnt[n]:
            cnt[n].add(r["Dispatch_Id"]); dur[n] += (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3
for n, c in sorted(acc.items(), key=lambda kv: -dur[kv[0]])[:16]:
    k = len(cnt[n]); wc = c["SQ_WAVE_CYCLES"] or 1
    print(f"{n:40s} n {k:4d} avg {dur[n]/k:7.1f} us  VALU insts {c['SQ_INSTS_VALU']/k:.3g}  LDS insts {c['SQ_INSTS_LDS']/k:.3g}  bank-conflict/LDS {c['SQ_LDS_BANK_CONFLICT']/max(1,c['SQ_INSTS_LDS']):.2f}  "
          f"active-VALU {c['SQ_ACTIVE_INST_VALU']/wc:.3f}  wait-any {c['SQ_WAIT_ANY']/wc:.3f}  wait-inst {c['SQ_WAIT_INST_ANY']/wc:.3f}  busy-cyc/launch {c['SQ_BUSY_CYCLES']/k:.3g}")
PY
find $O -name "*kernel_trace.csv" -delete; find $O -name "*agent_info.csv" -delete
