#!/bin/bash
# Developer script (GPU box): builds the DIAG library IN the box's copy of the tree (the shipped .so is replaced there only) and runs
# tools/dev/ab.sh on it -- for the timing-only ablation options (abl_skip, fake_s, wg_debug).  usage: tools/dev/ab_diag.sh "abl_skip=1" ...
set -u
: "${GRAFT_REPO_ROOT:?run on the GPU box}"
cd "$GRAFT_REPO_ROOT" && DIAG=1 bash iwae_amd/csrc/build.sh > gpurun_out/diag_build.log 2>&1 || { tail gpurun_out/diag_build.log; exit 1; }
tools/dev/ab.sh "$@"
