#!/bin/bash
# Instruction counters of chip::zstd_kernel on 8 192 frames of the bench payload (GPU box).  usage: bash tools/pmc_zstd.sh <tag>
set -eo pipefail
tag="${1:-r03}"
cd /tmp && export TMPDIR=/tmp
cd "$GRAFT_REPO_ROOT"
rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU -d gpurun_out/${tag}_pmc_zstd -o run --output-format csv -- python3 tools/time_zstd.py 8192 > gpurun_out/${tag}_pmc_zstd.log 2>&1
python3 tools/pmc_summary.py gpurun_out/${tag}_pmc_zstd/run_counter_collection.csv 8192 > gpurun_out/${tag}_pmc_zstd.txt
rocprofv3 --kernel-trace --pmc SQ_WAVES SQ_INST_CYCLES_SALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_SCA SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_INST_LDS -d gpurun_out/${tag}_pmc_zstd2 -o run --output-format csv -- python3 tools/time_zstd.py 8192 >> gpurun_out/${tag}_pmc_zstd.log 2>&1 || echo "second pass refused"
python3 tools/pmc_summary.py gpurun_out/${tag}_pmc_zstd2/run_counter_collection.csv 8192 >> gpurun_out/${tag}_pmc_zstd.txt || true
cat gpurun_out/${tag}_pmc_zstd.txt
