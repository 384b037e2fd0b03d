set -e
export CHIP_INFLATE_PIPE=1
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
timeout -k 5 60 python tools/time_run.py dynamic 256 2>&1 | grep -E "units"
timeout -k 5 90 python tools/exp/pipe_dbg.py dynamic 8192 2 2>&1 | tail -3
timeout -k 5 120 python tools/exp/x_stats.py dynamic 8192 2>&1 | tail -13
rocprofv3 --kernel-trace --stats -d gpurun_out/r4_pipe_kt -o run --output-format csv -- python3 tools/time_run.py dynamic 16384 > gpurun_out/r4_tr.txt 2>&1
grep units gpurun_out/r4_tr.txt
python3 - <<PY
import csv,glob
f=glob.glob("gpurun_out/r4_pipe_kt/**/*kernel_stats.csv", recursive=True)[0]
for r in csv.DictReader(open(f)):
    if "chip" in r["Name"]: print(r["Name"][:60], r["Calls"], r["AverageNs"])
PY
rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU -d gpurun_out/r4_pmcA -o run --output-format csv -- python3 tools/prof_run.py dynamic 8192 3 > gpurun_out/r4_pmcA.log 2>&1
python3 tools/pmc_summary.py gpurun_out/r4_pmcA/run_counter_collection.csv 8192 | grep -A8 lz77
