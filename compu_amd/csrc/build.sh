#!/bin/bash
# Builds compu_amd/libcompu_hip.so for gfx950 (cross-compiles without a GPU).
set -euo pipefail
here="$(cd "$(dirname "${BASH_SOURCE[0]}")" && pwd)"
out="$here/../libcompu_hip.so"
HIPCC="${HIPCC:-/opt/rocm/bin/hipcc}"
"$HIPCC" --offload-arch=gfx950 -O3 -std=c++17 -fPIC -shared -Wall \
    -o "$out" "$here"/*.hip "$@"
echo "built $out"
if [ "${CHIP_BUILD_STATS:-0}" = "1" ]; then
    "$HIPCC" --offload-arch=gfx950 -O3 -std=c++17 -fPIC -shared -Wall -DCHIP_STATS \
        -o "$here/../libcompu_hip_stats.so" "$here"/*.hip
    echo "built $here/../libcompu_hip_stats.so (diagnostic)"
fi
# C++ replay of the reference's integration tests over the C ABI (runs on the GPU box only)
g++ -O1 -std=c++17 -Wall -o "$here/../../tests/cpp/test_reference" "$here/../../tests/cpp/test_reference.cpp" \
    -L"$here/.." -lcompu_hip -Wl,-rpath,'$ORIGIN/../../compu_amd' -L/opt/rocm/lib -Wl,-rpath,/opt/rocm/lib -lamdhip64
echo "built tests/cpp/test_reference"
