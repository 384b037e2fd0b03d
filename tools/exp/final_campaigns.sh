set -e
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
make -s -C oracle
timeout -k 10 900 python tools/fuzz_gpu.py 10 777 2>&1 | grep -v amdgpu | tail -2
timeout -k 10 500 python tools/fuzz_stream.py 120 41 2>&1 | grep -v amdgpu | tail -2
timeout -k 10 600 python tools/fuzz_zstd_stream.py 80 29 2>&1 | grep -v amdgpu | tail -1
CHIP_INFLATE_PIPE=1 timeout -k 10 500 python tools/fuzz_gpu.py 3 778 2>&1 | grep -v amdgpu | tail -1
timeout -k 10 300 python tools/exp/try_dyn.py 2>&1 | tail -1
