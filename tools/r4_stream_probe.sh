# the side-stream picker (hn_internal.h: SidePick) across the scenarios that broke fixed choices: bash tools/r4_stream_probe.sh
R=$GRAFT_REPO_ROOT
export HN_DEBUG_PICK=1
for i in 1 2; do timeout 400 python3 $R/bench.py --steps 20 --warmup 5 --no-cpu-baseline 2>/tmp/err.txt | tail -1 | python3 -c "
import sys,json
d=json.loads(sys.stdin.read()); print('bench.py', d['value'], [ (s['value']) for s in d['secondary']])"; grep helmnet_hip /tmp/err.txt | sort | uniq -c; done
for a in "" "--side-stream"; do echo -n "bench_train [$a]: "; timeout 100 python3 $R/tools/bench_train.py --steps 20 $a 2>/tmp/err.txt | tail -1 | cut -c115-135; grep helmnet_hip /tmp/err.txt | sort | uniq -c; done
timeout 200 python3 $R/tools/bench_caller_stream.py 2>/tmp/err.txt | tail -1; grep helmnet_hip /tmp/err.txt | sort | uniq -c
