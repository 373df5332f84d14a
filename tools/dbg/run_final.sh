set -e
cd $GRAFT_REPO_ROOT
timeout -k 10 300 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -2
( time timeout -k 10 600 python bench.py ) 2>&1 | tail -6 | cut -c1-1500
