#!/bin/bash
# Builds libiwae_amd.so for gfx950 (MI355X) in-tree.  hipcc cross-compiles without a GPU.
set -euo pipefail
cd "$(dirname "$0")"
HIPCC=${HIPCC:-/opt/rocm/bin/hipcc}
# -fno-slp-vectorize: packed v_pk_*_f32 next to MFMAs costs more issue slots than it saves (plus v_mov shuffles)
FLAGS="--offload-arch=gfx950 -O3 -std=c++17 -fPIC -Wall -Wno-unused-function -fno-slp-vectorize"
# STAMPS=1 ./build.sh: diagnostic library with per-phase cycle stamps inside dense_kernel (never the shipped build)
if [ -n "${STAMPS:-}" ]; then FLAGS="$FLAGS -DIWAE_DENSE_STAMPS"; fi
$HIPCC $FLAGS -c kernels.hip -o kernels.o
$HIPCC $FLAGS -c model.hip -o model.o
$HIPCC $FLAGS -c fp32_kernels.hip -o fp32_kernels.o
$HIPCC --offload-arch=gfx950 -shared -fPIC -o ../libiwae_amd.so kernels.o model.o fp32_kernels.o -ldl
echo "built $(cd .. && pwd)/libiwae_amd.so"
