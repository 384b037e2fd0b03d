#!/bin/bash
# Same-box A/B timing of variant builds (GPU box; boxes differ by several per cent, so variants are only compared inside one call).
#   usage: bash tools/exp/ab_time.sh "<timing tool and its arguments>" lib1.so lib2.so ...
#   e.g.   bash tools/exp/ab_time.sh "tools/time_zstd.py 8192" libcompu_hip.so libcompu_hip_variant.so libcompu_hip.so
# (Variants that change the resident waves of a persistent grid: compare at the full launch size instead, tools/exp/w18_full.sh.)
# The libraries are looked up under compu_amd/ (built there with hipcc ... -D<variant flag> -o ../libcompu_hip_<name>.so *.hip).
set -e
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
tool="$1"; shift
for l in "$@"; do
  COMPU_HIP_LIB=$PWD/compu_amd/$l timeout -k 5 300 python $tool 2>&1 | grep -E "units|frames" || true
done
