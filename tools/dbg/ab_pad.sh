# A/B of occupancy-limited raster kernels (dynamic-LDS padding) on the tracking closure: GSX_PAD_F / GSX_PAD_B in KiB
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
for cfg in "0 0" "64 64" "40 40" "26 26" "96 96"; do
  set -- $cfg
  export GSX_PAD_F=$1 GSX_PAD_B=$2
  rm -rf gpurun_out/prof_pad
  rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_pad -o p -- python3 tools/prof_closure.py --frames 3 > gpurun_out/prof_pad.log 2>&1
  echo "== pad_f=$1 pad_b=$2"
  python tools/show_stats.py $(find gpurun_out/prof_pad -name "p_kernel_stats.csv") 3
done
