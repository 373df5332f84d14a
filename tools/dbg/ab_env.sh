# usage: bash tools/dbg/ab_env.sh "VAR=a VAR2=b" "VAR=c" ...   (each argument = one environment for a profiled run)
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
for cfg in "$@"; do
  rm -rf gpurun_out/prof_ab
  env $cfg rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_ab -o p -- python3 tools/prof_closure.py --frames 3 > gpurun_out/prof_ab.log 2>&1
  echo "== $cfg"
  python tools/show_stats.py $(find gpurun_out/prof_ab -name "p_kernel_stats.csv") ${NSHOW:-9}
done
