#!/bin/bash
# builds tools/lab/gemm_lab (gfx950) against the in-tree library
set -e
cd "$(dirname "$0")"
CSRC=../../robot-3dlotus_amd/csrc
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -mllvm -amdgpu-mfma-vgpr-form -Wno-unused-result -o gemm_lab gemm_lab.hip -L$CSRC -llotus_hip -Wl,-rpath,'$ORIGIN/'$CSRC "$@"
