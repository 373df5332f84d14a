cd $GRAFT_REPO_ROOT
for v in 0 1; do
  GSX_BIN_PRESORT=$v timeout -k 10 300 python tools/ab_raster.py 100000 250000 1000000 C=1 C=8 2>&1 | grep -E "^N=|isect_bin_sort" | sed -e 's/ 640x480.*//' -e 's/.*sync-free): //' | paste -sd' ' | sed "s/^/presort=$v: /"
done
