set -e
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -2
timeout -k 10 1700 python -m pytest tests -x -q -m gpu 2>&1 | tail -4
python bench.py > gpurun_out/r04_bench_final_build.json 2> gpurun_out/r04_bench_final_build.log; tail -c 600 gpurun_out/r04_bench_final_build.json
