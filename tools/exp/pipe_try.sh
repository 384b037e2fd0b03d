set -e
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
make -s -C oracle
# encoder: hash-table size against occupancy (timing only; the streams of the smaller tables differ from the oracle's)
for l in libcompu_hip.so libcompu_hip_hash11.so libcompu_hip_hash10.so libcompu_hip.so; do COMPU_HIP_LIB=$PWD/compu_amd/$l timeout -k 5 200 python tools/time_encode.py 16384 1 2>&1 | grep -E "units"; done
# the pipeline's traffic: FETCH_SIZE and WRITE_SIZE of its kernels, 16 384 dynamic units
for c in FETCH_SIZE WRITE_SIZE; do
  CHIP_INFLATE_PIPE=1 timeout -k 10 300 rocprofv3 --kernel-trace --pmc $c -d gpurun_out/r4_pipe_$c -o run --output-format csv -- python3 tools/time_run.py dynamic 16384 > gpurun_out/r4_pipe_$c.log 2>&1 || true
  python3 - $c <<'PY'
import csv, sys, collections
c = sys.argv[1]
rows = [r for r in csv.DictReader(open(f"gpurun_out/r4_pipe_{c}/run_counter_collection.csv")) if r["Counter_Name"] == c]
by = collections.defaultdict(list)
for r in rows: by[r["Kernel_Name"][:50]].append(float(r["Counter_Value"]))
for k, v in by.items():
    if "chip::" in k: print(f"pipeline {c} {k:50s} launches {len(v)} KiB per launch {sum(v)/len(v):12.0f} per unit KB {sum(v)/len(v)*1024/16384/1000:8.1f}")
PY
done
# the campaign through the pipeline
CHIP_INFLATE_PIPE=1 timeout -k 10 600 python tools/fuzz_gpu.py 4 421 2>&1 | tail -4
