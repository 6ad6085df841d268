# weight-gradient block cap (pixels per block) after the matrix-core backward DoubleConvs: bash tools/r4_train_cumask.sh
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
cat > /tmp/bt.py <<PY
import sys, os, runpy
sys.path.insert(0, "$R")
from helmnet_amd import _lib
_lib._LIB_PATH = os.path.abspath(sys.argv[1])
sys.argv = ["bench_train.py"] + sys.argv[2:]
runpy.run_path("$R/tools/bench_train.py", run_name="__main__")
PY
for rep in 1 2; do
 for L in $R/helmnet_amd/libhelmnet_hip.so $R/tools/lib_capdiv250.so $R/tools/lib_capdiv350.so $R/tools/lib_capdiv600.so; do
  for shape in "96 32" "128 32"; do set -- $shape
   echo -n "$(basename $L) n=$1 batch=$2: "; python3 /tmp/bt.py $L --steps 20 --n $1 --batch $2 2>/dev/null | tail -1 | python3 -c "import sys,json; print(round(json.loads(sys.stdin.read())['ms_per_step'],3))"
  done
 done
done
