set -e
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -1
timeout -k 10 1700 python -m pytest tests -x -q -m gpu 2>&1 | tail -3
python bench.py > gpurun_out/r04_bench_final_build.json 2> gpurun_out/r04_bench_final_build.log; python3 -c "
import json; b=json.load(open('gpurun_out/r04_bench_final_build.json')); print(b['value'], b['ms_per_step'], b['roofline']['frac']); print({k:(v['kernel_ms_avg'], v['verified']) for k,v in b['workloads'].items()})"
