set -e
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
for l in libcompu_hip_prev.so libcompu_hip.so libcompu_hip_prev.so libcompu_hip.so; do COMPU_HIP_LIB=$PWD/compu_amd/$l python tools/time_run.py dynamic 16384; done 2>&1 | grep units
for l in libcompu_hip_prev.so libcompu_hip.so; do
  for c in FETCH_SIZE WRITE_SIZE; do
    COMPU_HIP_LIB=$PWD/compu_amd/$l rocprofv3 --kernel-trace --pmc $c -d gpurun_out/nt_${l}_$c -o run --output-format csv -- python3 tools/prof_run.py dynamic 16384 3 > /dev/null 2>&1
    python3 - <<PY
import csv,glob
f=glob.glob("gpurun_out/nt_${l}_$c/**/*counter_collection.csv", recursive=True)[0]
v=[float(r["Counter_Value"]) for r in csv.DictReader(open(f)) if "inflate_kernel" in r["Kernel_Name"] and r["Counter_Name"]=="$c"]
print("$l $c KiB per launch of 16384:", round(v[-1]), " per unit KB:", round(v[-1]/16384*1.024,1))
PY
  done
done
