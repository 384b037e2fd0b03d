set -e
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
make -s -C oracle
timeout -k 10 500 python tools/fuzz_stream.py 120 31 2>&1 | grep -v amdgpu | tail -3
timeout -k 10 600 python tools/fuzz_zstd_stream.py 60 17 2>&1 | grep -v amdgpu | tail -3
timeout -k 10 500 python tools/fuzz_gpu.py 6 501 2>&1 | grep -v amdgpu | tail -2
timeout -k 10 300 python tools/exp/try_dyn.py 2>&1 | tail -2
