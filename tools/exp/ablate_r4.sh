# Round-4 measurement call (GPU box): ablation builds timed beside the product on one box, FETCH/WRITE for product and NO_GLOB,
# the fetch probe, the op-rate probe.
set -e
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
for kind in dynamic fixed; do
for l in libcompu_hip.so libcompu_hip_NO_GLOB.so libcompu_hip_NO_FLUSH.so libcompu_hip.so; do COMPU_HIP_LIB=$PWD/compu_amd/$l python tools/time_run.py $kind 16384; done 2>&1 | grep units
done
for l in libcompu_hip.so libcompu_hip_NO_GLOB.so libcompu_hip_NO_FLUSH.so; do
  for c in FETCH_SIZE WRITE_SIZE; do
    COMPU_HIP_LIB=$PWD/compu_amd/$l rocprofv3 --kernel-trace --pmc $c -d gpurun_out/abl_${l}_$c -o run --output-format csv -- python3 tools/time_run.py dynamic 16384 > /dev/null 2>&1 || true
    python3 - <<PY
import csv,glob
f=glob.glob("gpurun_out/abl_${l}_$c/**/*counter_collection.csv", recursive=True)[0]
v=[float(r["Counter_Value"]) for r in csv.DictReader(open(f)) if "inflate_kernel" in r["Kernel_Name"] and r["Counter_Name"]=="$c"]
print("$l $c KiB per launch of 16384:", round(v[-1]), " per unit KB:", round(v[-1]/16384*1.024,1))
PY
  done
done
hipcc --offload-arch=gfx950 -O3 -o /tmp/fetch_probe tools/exp/fetch_probe.hip 2>/dev/null
rocprofv3 --kernel-trace --pmc FETCH_SIZE -d gpurun_out/fetch_probe -o run --output-format csv -- /tmp/fetch_probe
python3 - <<PY
import csv,glob
f=glob.glob("gpurun_out/fetch_probe/**/*counter_collection.csv", recursive=True)[0]
for r in csv.DictReader(open(f)):
    if r["Counter_Name"]=="FETCH_SIZE": print(r["Kernel_Name"][:40], "FETCH_SIZE KiB", r["Counter_Value"], "bytes per load", float(r["Counter_Value"])*1024/(256*8*256*256))
PY
hipcc --offload-arch=gfx950 -O3 -o /tmp/op_rate tools/exp/op_rate.hip 2>/dev/null
/tmp/op_rate
