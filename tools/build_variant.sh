# build an experiment variant of the library: tools/build_variant.sh <name> <source.hip> [extra hipcc flags]  -> tools/lib_<name>.so
set -e
R=$(cd $(dirname $0)/.. && pwd)
name=$1; src=$2; shift 2
python3 -c "import sys; sys.path.insert(0, '$R'); from helmnet_amd.build import build; build()"
O=$R/helmnet_amd/build
/opt/rocm/bin/hipcc -O3 --offload-arch=gfx950 -fPIC -std=c++17 -I$R/include -I$R/helmnet_amd/csrc "$@" -c $R/helmnet_amd/csrc/$src -o /tmp/var_$name.o
objs=""
for f in $O/*.o; do b=$(basename $f .o); if [ "$b.hip" = "$src" ]; then objs="$objs /tmp/var_$name.o"; else objs="$objs $f"; fi; done
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o $R/tools/lib_$name.so $objs
echo $R/tools/lib_$name.so
