# tools/lib_repeat.so: the library with -DHN_EXP_REPEAT (hn_internal.h: HN_REP; tools/energy_probe.py)
set -e
R=$(cd $(dirname $0)/.. && pwd)
python3 -c "import sys; sys.path.insert(0, '$R'); from helmnet_amd.build import build; build()"
O=$R/helmnet_amd/build
F="-O3 --offload-arch=gfx950 -fPIC -std=c++17 -I$R/include -I$R/helmnet_amd/csrc -DHN_EXP_REPEAT"
/opt/rocm/bin/hipcc $F -c $R/helmnet_amd/csrc/hn_unet.hip -o /tmp/rep_unet.o &
/opt/rocm/bin/hipcc $F -c $R/helmnet_amd/csrc/hn_api.hip -o /tmp/rep_api.o &
wait
objs=""
for f in $O/*.o; do b=$(basename $f .o); case $b in hn_unet) objs="$objs /tmp/rep_unet.o";; hn_api) objs="$objs /tmp/rep_api.o";; *) objs="$objs $f";; esac; done
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o $R/tools/lib_repeat.so $objs
echo $R/tools/lib_repeat.so
