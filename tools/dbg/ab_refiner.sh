cd $GRAFT_REPO_ROOT
for i in 1 2; do
for v in device host; do
  echo -n "$v: "; GSX_POSE_REFINER=$v timeout -k 10 300 python tools/run_slam.py 2>&1 | tail -1 | cut -c1-160
done
done
