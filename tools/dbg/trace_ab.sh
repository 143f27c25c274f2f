#!/bin/bash
# kernel trace of the step with the LDS-DMA dense kernels on and off (one gpurun call): per-kernel tables and timelines
export TMPDIR=/tmp
mkdir -p gpurun_out/prof
for v in 1 0; do
  rm -rf gpurun_out/prof/dma${v}*
  LOTUS_GEMM_DMA=$v rocprofv3 --kernel-trace --stats -d gpurun_out/prof -o dma$v -- python bench.py --steps 20 --warmup 10 --no-cpu-baseline --no-other-modes --no-roofline --no-fresh-batches --no-side-workloads > gpurun_out/prof/dma$v.log 2>&1
  python profiles/summarize.py $(ls gpurun_out/prof/dma${v}_results.db gpurun_out/prof/*/dma${v}_results.db 2>/dev/null | head -1) 31 > gpurun_out/dma${v}_kernels.md
done
head -30 gpurun_out/dma1_kernels.md | cut -c1-200
head -30 gpurun_out/dma0_kernels.md | cut -c1-200
