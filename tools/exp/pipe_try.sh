set -e
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
timeout -k 5 200 python tools/stats_run.py dynamic 8192 2>&1 | grep -E "kernel=|hdr"
for k in dynamic fixed; do timeout -k 5 120 python tools/time_run.py $k 16384 2>&1 | grep -E "units"; done
timeout -k 10 900 python -m pytest tests/test_inflate_gpu.py tests/test_decoder_gpu.py -x -q -m gpu 2>&1 | tail -5
