cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
export COMPU_HIP_LIB=compu_amd/libcompu_hip_r2.so
rocprofv3 --kernel-trace --pmc SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_SCA SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS -d gpurun_out/pmcB2 -o runc --output-format csv -- python3 tools/prof_run.py dynamic 8192 3 > gpurun_out/pmcB2.log 2>&1
rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU -d gpurun_out/pmcA2 -o runc --output-format csv -- python3 tools/prof_run.py dynamic 8192 3 > gpurun_out/pmcA2.log 2>&1
