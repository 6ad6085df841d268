# tools/lib_<name>.so with extra hipcc flags on SEVERAL sources: tools/build_multi_variant.sh <name> "<flags>" a.hip b.hip ...
set -e
R=$(cd $(dirname $0)/.. && pwd)
name=$1; flags=$2; shift 2
python3 -c "import sys; sys.path.insert(0, '$R'); from helmnet_amd.build import build; build()"
O=$R/helmnet_amd/build
for src in "$@"; do
  /opt/rocm/bin/hipcc -O3 --offload-arch=gfx950 -fPIC -std=c++17 -I$R/include -I$R/helmnet_amd/csrc $flags -c $R/helmnet_amd/csrc/$src -o /tmp/mv_${name}_$(basename $src .hip).o &
done
wait
objs=""
for f in $O/*.o; do b=$(basename $f .o); if [ -f /tmp/mv_${name}_$b.o ] && echo " $* " | grep -q " $b.hip "; then objs="$objs /tmp/mv_${name}_$b.o"; else objs="$objs $f"; fi; done
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o $R/tools/lib_$name.so $objs
echo $R/tools/lib_$name.so
