# SQ counters of the final training step's kernels incl. the matrix-core backward DoubleConvs (second --pmc pass: MFMA counters): bash tools/r4_train_pmc2.sh
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r4tp2
rm -rf $O; mkdir -p $O
timeout 300 rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_INSTS_VALU SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT --output-format csv -d $O/a -- python3 $R/tools/bench_train.py --steps 3 > /dev/null 2>&1
timeout 300 rocprofv3 --kernel-trace --pmc SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU_MFMA_MOPS_F32 SQ_BUSY_CYCLES --output-format csv -d $O/b -- python3 $R/tools/bench_train.py --steps 3 > /dev/null 2>&1
timeout 60 python3 - <<PY
import csv, glob, collections
acc = collections.defaultdict(lambda: collections.defaultdict(float)); cnt = collections.defaultdict(lambda: collections.defaultdict(set)); dur = collections.defaultdict(float)
for tag in "ab":
    for f in glob.glob("$O/" + tag + "/*/*counter_collection.csv"):
        for r in csv.DictReader(open(f)):
            n = r["Kernel_Name"].replace("void ", "").replace("hn::(anonymous namespace)::", "").split("(")[0]
            acc[n][r["Counter_Name"]] += float(r["Counter_Value"])
            if r["Dispatch_Id"] not in cnt[n][tag]:
                cnt[n][tag].add(r["Dispatch_Id"])
                if tag == "a": dur[n] += (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3
for n, c in sorted(acc.items(), key=lambda kv: -dur[kv[0]])[:18]:
    ka, kb = max(1, len(cnt[n]["a"])), max(1, len(cnt[n]["b"])); wc = c["SQ_WAVE_CYCLES"] or 1
    mf = c["SQ_INSTS_MFMA"] / kb
    print(f"{n:40s} n {ka:4d} avg {dur[n]/ka:7.1f} us  VALU {c['SQ_INSTS_VALU']/ka:.3g}  LDS {c['SQ_INSTS_LDS']/ka:.3g}  conflicts/LDS {c['SQ_LDS_BANK_CONFLICT']/max(1,c['SQ_INSTS_LDS']):.2f}  "
          f"wait-any {c['SQ_WAIT_ANY']/wc:.3f}  wait-inst {c['SQ_WAIT_INST_ANY']/wc:.3f}  MFMA insts {mf:.3g}  MFMA-busy cycles {c['SQ_VALU_MFMA_BUSY_CYCLES']/kb:.3g}")
PY
rm -rf $O
