#!/bin/bash
# Builds libxicsrt_hip.so for gfx950 (MI355X) in-tree.  hipcc cross-compiles
# without a GPU.  -ffp-contract=off: the reference (NumPy) evaluates a*b+c with
# two roundings; fused multiply-adds appear only where written explicitly.
# -instcombine-max-copied-from-constant-users: large by-value kernel arguments (staged kernels) are read
# through scalar loads; above 300 uses LLVM would otherwise keep a per-lane private copy.
# -disable-machine-licm: MachineLICM hoists the materialisation of every FP64 literal (polynomial
# coefficients of sincos / acos / exp ...) out of the tile loop into vector registers and then spills them
# to scratch; without it the fused lean kernel needs 89 instead of 128 VGPRs and no scratch at all.
set -e
cd "$(dirname "$0")"
HIPCC=${HIPCC:-/opt/rocm/bin/hipcc}
$HIPCC --offload-arch=gfx950 -O3 -std=c++17 -fPIC -shared \
    -ffp-contract=off -fno-fast-math \
    -mllvm -instcombine-max-copied-from-constant-users=10000 \
    -mllvm -disable-machine-licm \
    -Wall -Wno-unused-function -Werror -Wno-error=unused-command-line-argument \
    ${XRT_EXTRA_FLAGS} \
    -o ${XRT_OUT:-libxicsrt_hip.so} xrt_kernels.hip
echo "built $(pwd)/${XRT_OUT:-libxicsrt_hip.so}"
