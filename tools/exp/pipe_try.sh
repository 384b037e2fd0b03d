set -e
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
make -s -C oracle
timeout -k 5 200 python tools/time_zstd.py 8192 2>&1 | grep -E "frames"
timeout -k 10 800 python -m pytest tests/test_zstd_gpu.py -x -q -m gpu 2>&1 | tail -3
timeout -k 10 600 python tools/fuzz_gpu.py 3 77 2>&1 | grep -v amdgpu | tail -4
