# it/s of variant libraries, interleaved repetitions: gpurun -- bash tools/exp_mix.sh <reps> <names...>   (base = built library, mfma = base with --opt dc_valu=0)
R=$GRAFT_REPO_ROOT; cd $R
reps=$1; shift
B="--steps 300 --warmup 30 --no-cpu-baseline --no-secondary"
for r in $(seq $reps); do for l in "$@"; do
  if [ $l = base ]; then L=helmnet_amd/libhelmnet_hip.so; O=""; elif [ $l = mfma ]; then L=helmnet_amd/libhelmnet_hip.so; O="--opt dc_valu=0"; else L=tools/lib_$l.so; O=""; fi
  python tools/run_with_lib.py $L $B $O 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(\"$l\", d[\"value\"])"
done; done | python -c "
import sys, collections
v = collections.defaultdict(list)
for line in sys.stdin:
    k, x = line.split(); v[k].append(float(x))
for k, xs in v.items(): print(f'{k:6s} median {sorted(xs)[len(xs)//2]:8.1f}   ' + ' '.join(f'{x:.0f}' for x in xs))
"
