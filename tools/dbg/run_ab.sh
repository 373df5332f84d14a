set -e
cd $GRAFT_REPO_ROOT
timeout -k 10 300 python tools/ab_raster.py 100000 500000 C=1 C=8 2>&1 | grep -E "^N=|tile order"
