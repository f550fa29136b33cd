#!/bin/bash
# Build libtaco_hip.so for MI355X (gfx950).  hipcc cross-compiles without a GPU.
# The register / scratch usage of every kernel is kept in kernel_resources.txt (tools/check_kernel_resources.py reads it and fails
# the build when a persistent kernel has scratch).
set -e
cd "$(dirname "$0")"
HIPCC=${HIPCC:-/opt/rocm/bin/hipcc}
if ! $HIPCC --offload-arch=gfx950 -O3 -std=c++17 -shared -fPIC -Rpass-analysis=kernel-resource-usage -o libtaco_hip.so taco_lib.hip 2> kernel_resources.txt; then
  grep -B2 -A8 -E "error:|fatal" kernel_resources.txt >&2 || cat kernel_resources.txt >&2
  exit 1
fi
grep -A3 "warning:" kernel_resources.txt >&2 || true
echo "built $(pwd)/libtaco_hip.so"
python3 ../../tools/check_kernel_resources.py kernel_resources.txt
