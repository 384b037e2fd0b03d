set -e
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
make -s -C oracle
timeout -k 10 1500 python -m pytest tests/test_inflate_gpu.py tests/test_decoder_gpu.py -x -q -m gpu 2>&1 | tail -3
