# SQ counters of the Winograd level-0 kernels next to the direct ones (same command, dc_wino = 11 / 0)
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r4wp
rm -rf $O; mkdir -p $O
for m in 11 0; do
  rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_INSTS_LDS --output-format csv -d $O/pmc$m -- python3 $R/bench.py --steps 24 --warmup 12 --no-cpu-baseline --no-secondary --opt dc_wino=$m > /dev/null 2>&1
done
python3 - <<PY
import csv, glob, collections
for m in (11, 0):
    acc = collections.defaultdict(lambda: collections.defaultdict(float)); cnt = collections.defaultdict(set)
    for f in glob.glob("$O/pmc%d/*/*counter_collection.csv" % m):
        for r in csv.DictReader(open(f)):
            n = r["Kernel_Name"].replace("void ", "").replace("hn::(anonymous namespace)::", "").split("(")[0]
            if not n.startswith("k_dc_"): continue
            acc[n][r["Counter_Name"]] += float(r["Counter_Value"]); cnt[n].add(r["Dispatch_Id"])
    print("== dc_wino =", m)
    for n, c in sorted(acc.items()):
        k = len(cnt[n])
        wc = c["SQ_WAVE_CYCLES"] / k
        print(f"{n:34s} launches {k:4d}  wave-cycles {wc:.3g}  VALU insts {c['SQ_INSTS_VALU']/k:.3g}  LDS insts {c['SQ_INSTS_LDS']/k:.3g}  "
              f"active-VALU/wave-cycles {c['SQ_ACTIVE_INST_VALU']/c['SQ_WAVE_CYCLES']:.3f}  wait-any {c['SQ_WAIT_ANY']/c['SQ_WAVE_CYCLES']:.3f}  wait-inst {c['SQ_WAIT_INST_ANY']/c['SQ_WAVE_CYCLES']:.3f}  busy-cycles {c['SQ_BUSY_CYCLES']/k:.3g}")
PY
find $O -name "*kernel_trace.csv" -delete; find $O -name "*agent_info.csv" -delete
