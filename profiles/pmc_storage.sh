#!/bin/bash
# HBM bytes per training step of the PerAct workload (BASELINE configs[4]) with bf16 and with fp32 activation storage:
#   bash profiles/pmc_storage.sh r03     (on the GPU box, from the repo root; then python profiles/pmc_storage.py r03)
# Separate --pmc passes (FETCH_SIZE / WRITE_SIZE) with --kernel-trace only, as profiles/pmc_passes.sh.
set -u
TAG=${1:-r03}
export TMPDIR=/tmp
ARGS="--workload peract --steps 3 --warmup 2 --no-cpu-baseline --no-roofline --no-other-modes --no-fresh-batches"
mkdir -p gpurun_out
for S in bf16 fp32; do
  rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d gpurun_out/pmc_${TAG}_peract_${S}_fetch -o f -- python bench.py $ARGS --act-storage $S > gpurun_out/pmc_${TAG}_peract_${S}_fetch.log 2>&1
  rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d gpurun_out/pmc_${TAG}_peract_${S}_write -o w -- python bench.py $ARGS --act-storage $S > gpurun_out/pmc_${TAG}_peract_${S}_write.log 2>&1
done
ls gpurun_out | grep pmc_${TAG}_peract
