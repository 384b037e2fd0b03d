#!/bin/bash
# Round profile recipe (GPU box): kernel trace of the default bench, then FETCH_SIZE / WRITE_SIZE in their own
# passes (never combined with other trace domains).  Outputs land under gpurun_out/; tools/profiles_summarize.py
# turns them into the files kept under profiles/.   usage: bash tools/profile_round.sh <tag>
set -eo pipefail
tag="${1:-r01}"
cd /tmp && export TMPDIR=/tmp
cd "$GRAFT_REPO_ROOT"
make -s -C oracle
rocprofv3 --kernel-trace --stats -d gpurun_out/${tag}_kt -o run --output-format csv -- python3 bench.py > gpurun_out/${tag}_bench.json 2> gpurun_out/${tag}_bench.log
rocprofv3 --kernel-trace --pmc FETCH_SIZE -d gpurun_out/${tag}_pmc_fetch -o run --output-format csv -- python3 bench.py --steps 3 --warmup 1 --no-cpu > /dev/null 2> gpurun_out/${tag}_pmc_fetch.log
rocprofv3 --kernel-trace --pmc WRITE_SIZE -d gpurun_out/${tag}_pmc_write -o run --output-format csv -- python3 bench.py --steps 3 --warmup 1 --no-cpu > /dev/null 2> gpurun_out/${tag}_pmc_write.log
python3 bench.py --workload mixed --no-cpu > gpurun_out/${tag}_mixed.json 2>> gpurun_out/${tag}_bench.log
python3 bench.py --workload encode --no-cpu > gpurun_out/${tag}_encode.json 2>> gpurun_out/${tag}_bench.log
