"""Print average kernel durations (us) from a rocprofv3 --kernel-trace --stats output directory."""
import csv, glob, sys
pat = sys.argv[2:] if len(sys.argv) > 2 else None
for f in glob.glob(sys.argv[1] + "/*/*_kernel_stats.csv"):
    for r in csv.DictReader(open(f)):
        name = r["Name"].replace("void ", "").replace("hn::(anonymous namespace)::", "").split("(")[0]
        if pat and not any(p in name for p in pat):
            continue
        print(f"{name:45s} calls {int(r['Calls']):4d} avg {float(r['AverageNs'])/1e3:8.1f} us  min {float(r['MinNs'])/1e3:8.1f}  max {float(r['MaxNs'])/1e3:8.1f}")
