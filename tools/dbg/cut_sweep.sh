cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
for m in 0.02 0.05 0.15 0.4; do
python3 tools/prof_closure.py --frames 6 --cut-margin $m 2>&1 | grep "tile sort"
done
