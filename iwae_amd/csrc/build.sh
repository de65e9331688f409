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
if [ -n "${DIAG:-}${STAMPS:-}" ]; then FLAGS="$FLAGS -DIWAE_DIAG"; fi
if [ -n "${STAMPS:-}" ]; then FLAGS="$FLAGS -DIWAE_DENSE_STAMPS"; fi
$HIPCC $FLAGS -c kernels.hip -o kernels.o
$HIPCC $FLAGS -c model.hip -o model.o
$HIPCC $FLAGS -c fp32_kernels.hip -o fp32_kernels.o
$HIPCC --offload-arch=gfx950 -shared -fPIC -o ../libiwae_amd.so kernels.o model.o fp32_kernels.o -ldl
echo "built $(cd .. && pwd)/libiwae_amd.so"
