cd $GRAFT_REPO_ROOT
for it in 4 8 16 32; do echo "== items $it"; GSX_BIN_ITEMS=$it timeout -k 10 200 python tools/ab_raster.py 500000 C=1 C=8 2>&1 | grep "N=\|isect_bin"; done
echo "== auto"; timeout -k 10 200 python tools/ab_raster.py 100000 500000 C=1 C=8 2>&1 | grep "N=\|isect_bin"
