# in-line (side_stream=0) and side-stream durations of the hidden-state kernels, streaming (hn_cs.hip) vs general: gpurun -- bash tools/cs_prof.sh
cd /tmp && export TMPDIR=/tmp; R=$GRAFT_REPO_ROOT
for cfg in "--size 512 --batch 16" ""; do for sk in 1 0; do for ss in 0 1; do
  O=$R/gpurun_out/csp; rm -rf $O; mkdir -p $O
  rocprofv3 --kernel-trace --stats --output-format csv -d $O -- python3 $R/bench.py --steps 30 --warmup 10 --no-cpu-baseline --no-secondary $cfg --opt state_kernel=$sk --opt side_stream=$ss > /dev/null 2>&1
  echo "== cfg [$cfg] state_kernel=$sk side_stream=$ss"; python3 $R/tools/kstats.py $O k_conv_state k_double_conv
done; done; done
