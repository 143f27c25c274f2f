#!/bin/bash
# current in-step timeline + kernel table of the headline step (no fresh batches): gpurun_out/now_timeline.txt, now_kernels.md
export TMPDIR=/tmp; mkdir -p gpurun_out/prof; rm -f gpurun_out/prof/now_*
rocprofv3 --kernel-trace --stats -d gpurun_out/prof -o now -- python bench.py --steps 20 --warmup 10 --no-cpu-baseline --no-other-modes --no-roofline --no-fresh-batches --no-side-workloads > gpurun_out/prof/now.log 2>&1
python profiles/timeline.py gpurun_out/prof/now_results.db > gpurun_out/now_timeline.txt
python profiles/summarize.py gpurun_out/prof/now_results.db 20 > gpurun_out/now_kernels.md
python tools/step_sequence.py gpurun_out/prof/now_results.db > gpurun_out/now_sequence.txt 2>&1
rm -rf gpurun_out/prof
head -50 gpurun_out/now_timeline.txt
