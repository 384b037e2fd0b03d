# PMC passes over the encoder kernel: bash tools/pmc_encode.sh [kind=encode] [tag=enc]   (on the GPU box, from the repo root)
# (A third pass that asked for TA_*/TCP_* counters is not run: rocprofv3 aborted at the first dispatch, before any product kernel ran --
# rocprofiler_create_counter_config, error 38 "Request exceeds the capabilities of the hardware to collect", i.e. more TA/TCP counters than
# one pass can hold (round 2's gpurun_out/encC.log).  Those counters are not collected; the candidate-load amplification is read off the
# ISA and FETCH_SIZE instead.)
kind=${1:-encode}; tag=${2:-enc}
cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU -d gpurun_out/${tag}A -o runc --output-format csv -- python3 tools/prof_run.py $kind 8192 3 > gpurun_out/${tag}A.log 2>&1 &&
rocprofv3 --kernel-trace --pmc SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_SCA SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_INST_CYCLES_VMEM_RD -d gpurun_out/${tag}B -o runc --output-format csv -- python3 tools/prof_run.py $kind 8192 3 > gpurun_out/${tag}B.log 2>&1
for p in A B; do python3 tools/pmc_summary.py gpurun_out/${tag}$p/runc_counter_collection.csv 8192; done > gpurun_out/${tag}_pmc.txt 2>&1
