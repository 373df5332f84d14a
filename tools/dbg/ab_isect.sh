cd $GRAFT_REPO_ROOT
for v in base items1 items2; do
  if [ $v != base ]; then cp gslam_amd/libgsx.so /tmp/libgsx_base.so 2>/dev/null; cp gslam_amd/_ab/libgsx_$v.so gslam_amd/libgsx.so; fi
  echo "== $v"; timeout -k 10 200 python tools/ab_raster.py 100000 500000 C=1 2>&1 | grep "N=\|isect_bin"
  if [ $v != base ]; then cp /tmp/libgsx_base.so gslam_amd/libgsx.so; fi
done
