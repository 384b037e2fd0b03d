set -e
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
tag=r04
COMPU_HIP_LIB=$PWD/compu_amd/libcompu_hip_NO_CAND.so rocprofv3 --kernel-trace --pmc FETCH_SIZE -d gpurun_out/${tag}_encode_noglob_pmc_fetch -o run --output-format csv -- python3 bench.py --workload encode --steps 3 --warmup 1 --no-cpu --extra 0 > gpurun_out/r4_enc_nocand.json 2>> gpurun_out/${tag}_bench.log || true
for l in libcompu_hip.so libcompu_hip_NO_CAND.so; do COMPU_HIP_LIB=$PWD/compu_amd/$l timeout -k 5 200 python tools/time_encode.py 16384 1 2>&1 | grep -E "units"; done
timeout -k 10 900 python -m pytest tests/test_inflate_gpu.py -x -q -m gpu -k "pipeline" 2>&1 | tail -3
