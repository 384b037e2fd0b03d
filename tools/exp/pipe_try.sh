set -e
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
make -s -C oracle
for l in libcompu_hip.so libcompu_hip_prev.so libcompu_hip.so libcompu_hip_prev.so; do COMPU_HIP_LIB=$PWD/compu_amd/$l timeout -k 5 200 python tools/time_zstd.py 8192 2>&1 | grep -E "frames"; done
timeout -k 10 800 python -m pytest tests/test_zstd_gpu.py -x -q -m gpu 2>&1 | tail -3
