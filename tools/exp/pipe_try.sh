set -e
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
for k in dynamic fixed; do timeout -k 5 120 python tools/time_run.py $k 16384 2>&1 | grep -E "units"; done
timeout -k 5 90 python tools/exp/pipe_dbg.py dynamic 4096 1 2>&1 | tail -2
