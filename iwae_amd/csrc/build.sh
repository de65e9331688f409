#!/bin/bash
# Builds libiwae_amd.so for gfx950 (MI355X) in-tree.  hipcc cross-compiles without a GPU.
set -euo pipefail
cd "$(dirname "$0")"
HIPCC=${HIPCC:-/opt/rocm/bin/hipcc}
# -fno-slp-vectorize: packed v_pk_*_f32 next to MFMAs costs more issue slots than it saves (plus v_mov shuffles)
FLAGS="--offload-arch=gfx950 -O3 -std=c++17 -fPIC -Wall -Wno-unused-function -fno-slp-vectorize"
# DIAG=1 ./build.sh: diagnostic library (never the shipped build): the weight-gradient ablation branches (WgradPArgs.dbg), the
# phase-stamp instantiation of out_bwd_pair_kernel and the diagnostic names of iwae_set_option.  STAMPS=1: DIAG plus per-phase
# cycle stamps inside dense_kernel / bern_pipe_kernel / dec_bwd_kernel.
# Build id = sha256 over the sources this library is made of (the same list, order and framing as iwae_amd/_capi.py::source_build_id):
# iwae_build_id() returns it, tests/conftest.py rebuilds when the binary under test does not match the tree, bench.py stamps its line and
# refuses profile artefacts taken on another build.
BUILD_ID=$(for f in build.sh fp32_kernels.hip kernels.h kernels.hip layout.h model.hip ../../include/iwae_amd.h; do echo "== $(basename $f)"; cat "$f"; done | sha256sum | cut -c1-16)
if [ -n "${DIAG:-}${STAMPS:-}" ]; then BUILD_ID="${BUILD_ID}-diag"; fi
if [ -n "${DIAG:-}${STAMPS:-}" ]; then FLAGS="$FLAGS -DIWAE_DIAG"; fi
if [ -n "${STAMPS:-}" ]; then FLAGS="$FLAGS -DIWAE_DENSE_STAMPS"; fi
# the three translation units compile side by side (kernels.hip is the long one)
$HIPCC $FLAGS -c kernels.hip -o kernels.o & p1=$!
$HIPCC $FLAGS -DIWAE_BUILD_ID="\"$BUILD_ID\"" -c model.hip -o model.o & p2=$!
$HIPCC $FLAGS -c fp32_kernels.hip -o fp32_kernels.o & p3=$!
wait $p1; wait $p2; wait $p3
$HIPCC --offload-arch=gfx950 -shared -fPIC -o ../libiwae_amd.so kernels.o model.o fp32_kernels.o -ldl
echo "built $(cd .. && pwd)/libiwae_amd.so (build id $BUILD_ID)"
