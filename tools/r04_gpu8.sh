cd $GRAFT_REPO_ROOT
python tools/c5mix_clock_watch.py C5mix 30 2>&1 | cut -c1-330
