#!/bin/bash
# Round profile recipe (GPU box): kernel trace + stats of the default bench (which covers every BASELINE config), then, per
# workload, FETCH_SIZE / WRITE_SIZE in passes of their own (never combined with other trace domains), then the instruction
# counters of the inflate kernel.  Outputs land under gpurun_out/; tools/profiles_summarize.py turns them into the files kept
# under profiles/.
#   usage: bash tools/profile_round.sh <tag>
set -eo pipefail
tag="${1:-r04}"
cd /tmp && export TMPDIR=/tmp
cd "$GRAFT_REPO_ROOT"
make -s -C oracle
rocprofv3 --kernel-trace --stats -d gpurun_out/${tag}_kt -o run --output-format csv -- python3 bench.py > gpurun_out/${tag}_bench.json 2> gpurun_out/${tag}_bench.log
echo "[profile] default bench traced"
for wl in dynamic stored fixed mixed encode; do
  rocprofv3 --kernel-trace --pmc FETCH_SIZE -d gpurun_out/${tag}_${wl}_pmc_fetch -o run --output-format csv -- python3 bench.py --workload $wl --steps 3 --warmup 1 --no-cpu --extra 0 > /dev/null 2>> gpurun_out/${tag}_bench.log
  rocprofv3 --kernel-trace --pmc WRITE_SIZE -d gpurun_out/${tag}_${wl}_pmc_write -o run --output-format csv -- python3 bench.py --workload $wl --steps 3 --warmup 1 --no-cpu --extra 0 > /dev/null 2>> gpurun_out/${tag}_bench.log
  echo "[profile] $wl PMC passes done"
done
# the LZ77 source fetches, by ablation: the same FETCH_SIZE pass on a build that leaves them out (-DCHIP_EXP_NO_GLOB; its output is wrong, its
# other traffic is the product's).  The difference is counted as it stands -- a scattered 16-byte load is one 64-byte request, which
# FETCH_SIZE reports exactly (tools/exp/fetch_probe.hip) -- while the streamed rest is doubled.
(cd compu_amd/csrc && hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -shared -DCHIP_EXP_NO_GLOB -o ../libcompu_hip_NO_GLOB.so *.hip 2>/dev/null)
for wl in dynamic fixed; do
  COMPU_HIP_LIB=$PWD/compu_amd/libcompu_hip_NO_GLOB.so rocprofv3 --kernel-trace --pmc FETCH_SIZE -d gpurun_out/${tag}_${wl}_noglob_pmc_fetch -o run --output-format csv -- python3 bench.py --workload $wl --steps 3 --warmup 1 --no-cpu --extra 0 > /dev/null 2>> gpurun_out/${tag}_bench.log || true
done
# the encoder's candidate gathers, the same way (-DCHIP_EXP_NO_CAND: every candidate load reads the lane's own position)
(cd compu_amd/csrc && hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -shared -DCHIP_EXP_NO_CAND -o ../libcompu_hip_NO_CAND.so *.hip 2>/dev/null)
COMPU_HIP_LIB=$PWD/compu_amd/libcompu_hip_NO_CAND.so rocprofv3 --kernel-trace --pmc FETCH_SIZE -d gpurun_out/${tag}_encode_noglob_pmc_fetch -o run --output-format csv -- python3 bench.py --workload encode --steps 3 --warmup 1 --no-cpu --extra 0 > /dev/null 2>> gpurun_out/${tag}_bench.log || true
echo "[profile] source-fetch ablation done"
hipcc --offload-arch=gfx950 -O3 -o /tmp/fetch_probe tools/exp/fetch_probe.hip 2>/dev/null
rocprofv3 --kernel-trace --pmc FETCH_SIZE -d gpurun_out/${tag}_fetch_probe -o run --output-format csv -- /tmp/fetch_probe > gpurun_out/${tag}_fetch_probe.log 2>&1 || true
rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU -d gpurun_out/${tag}_pmc_insts -o run --output-format csv -- python3 tools/prof_run.py dynamic 8192 3 > gpurun_out/${tag}_pmc_insts.log 2>&1
python3 tools/pmc_summary.py gpurun_out/${tag}_pmc_insts/run_counter_collection.csv 8192 > gpurun_out/${tag}_pmc_insts.txt || true
# the same instruction counters for the zstd kernel (8 192 frames) and the level-1 encoder (8 192 units)
rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU -d gpurun_out/${tag}_pmc_insts_zstd -o run --output-format csv -- python3 tools/time_zstd.py 8192 > gpurun_out/${tag}_pmc_insts_zstd.log 2>&1
python3 tools/pmc_summary.py gpurun_out/${tag}_pmc_insts_zstd/run_counter_collection.csv 8192 > gpurun_out/${tag}_pmc_insts_zstd.txt || true
rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU -d gpurun_out/${tag}_pmc_insts_encode -o run --output-format csv -- python3 tools/prof_run.py encode 8192 3 > gpurun_out/${tag}_pmc_insts_encode.log 2>&1
python3 tools/pmc_summary.py gpurun_out/${tag}_pmc_insts_encode/run_counter_collection.csv 8192 > gpurun_out/${tag}_pmc_insts_encode.txt || true
echo "[profile] instruction counters done"
# cfg5 at its per-GPU size (1 M frames over 8 GPUs = 131 072 per GPU): the line is kept as evidence that the size runs on one GPU
python3 bench.py --workload mixed --units 131072 --steps 3 --warmup 1 --no-cpu --extra 0 > gpurun_out/${tag}_mixed_131072.json 2>> gpurun_out/${tag}_bench.log || true
echo "[profile] done"
