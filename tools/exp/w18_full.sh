cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
for l in ${LIBS:-libcompu_hip.so libcompu_hip_w18.so libcompu_hip_g320.so libcompu_hip.so libcompu_hip_w18.so}; do
  CHIP_DEBUG_GRID=1 COMPU_HIP_LIB=$PWD/compu_amd/$l python bench.py --workload ${WL:-dynamic} --steps 10 --warmup 2 --no-cpu --extra 0 2>gpurun_out/w18_err.txt | python3 -c "import sys,json; b=json.loads(sys.stdin.read()); print('$l', b['roofline']['kernel_ms_avg'], b['verified_bit_exact'] if 'verified_bit_exact' in b else '')"
grep -h "waves per CU" gpurun_out/w18_err.txt | tail -1; done
