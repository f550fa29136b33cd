#!/bin/bash
# Build libtaco_hip.so for MI355X (gfx950).  hipcc cross-compiles without a GPU.
set -e
cd "$(dirname "$0")"
HIPCC=${HIPCC:-/opt/rocm/bin/hipcc}
$HIPCC --offload-arch=gfx950 -O3 -std=c++17 -shared -fPIC -o libtaco_hip.so taco_lib.hip
echo "built $(pwd)/libtaco_hip.so"
