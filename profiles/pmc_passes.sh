#!/bin/bash
# The three rocprofv3 counter passes behind profiles/rNN_pmc.json / rNN_sq.md (run on the GPU box, from the repo root):
#   bash profiles/pmc_passes.sh r02
# Separate --pmc passes with --kernel-trace only (FETCH_SIZE and WRITE_SIZE do not fit one pass; MI355X_MICROARCH.md
# "rocprofv3 PMC slots").  Outputs land in gpurun_out/pmc_<tag>_{fetch,write,sq}/.
set -u
TAG=${1:-r04}
export TMPDIR=/tmp
ARGS="--steps 3 --warmup 2 --no-cpu-baseline --no-roofline --no-other-modes --no-fresh-batches ${BENCH_ARGS:-}"  # e.g. BENCH_ARGS="--workload peract" with tag r03p
mkdir -p gpurun_out
rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d gpurun_out/pmc_${TAG}_fetch -o f -- python bench.py $ARGS > gpurun_out/pmc_${TAG}_fetch.log 2>&1
rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d gpurun_out/pmc_${TAG}_write -o w -- python bench.py $ARGS > gpurun_out/pmc_${TAG}_write.log 2>&1
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_ANY SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d gpurun_out/pmc_${TAG}_sq -o s -- python bench.py $ARGS > gpurun_out/pmc_${TAG}_sq.log 2>&1
sha256sum robot-3dlotus_amd/csrc/liblotus_hip.so | cut -c1-16 > gpurun_out/pmc_${TAG}_so.sha
ls gpurun_out/pmc_${TAG}_fetch gpurun_out/pmc_${TAG}_write gpurun_out/pmc_${TAG}_sq
