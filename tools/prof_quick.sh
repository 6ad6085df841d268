# quick per-kernel timing (+ a few PMC counters) of the default bench: gpurun -- bash tools/prof_quick.sh [bench args]
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/quick
rm -rf $O; mkdir -p $O
BENCH="python3 $R/bench.py --steps 40 --warmup 12 --no-cpu-baseline --no-secondary $*"
rocprofv3 --kernel-trace --stats --output-format csv -d $O/kt -- $BENCH > $O/kt.log 2>&1
python3 $R/tools/kstats.py $O/kt | sort -k5 -n -r | head -24
rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_LDS_BANK_CONFLICT SQ_INSTS_VALU SQ_INSTS_LDS SQ_BUSY_CYCLES --output-format csv -d $O/pmc -- $BENCH > /dev/null 2>&1
python3 - <<PY
import csv, glob, collections
acc = collections.defaultdict(lambda: collections.defaultdict(float)); cnt = collections.Counter()
for f in glob.glob("$O/pmc/*/*counter_collection.csv"):
    for r in csv.DictReader(open(f)):
        n = r["Kernel_Name"].replace("void ", "").replace("hn::(anonymous namespace)::", "").split("(")[0]
        acc[n][r["Counter_Name"]] += float(r["Counter_Value"]); 
        if r["Counter_Name"] == "SQ_WAVE_CYCLES": cnt[n] += 1
for n, c in acc.items():
    if "k_dc" in n or "down" in n or "up" in n:
        k = cnt[n] or 1
        print(f"{n:40s} " + " ".join(f"{a.replace('SQ_','')}={v/k:.3g}" for a, v in sorted(c.items())))
PY
